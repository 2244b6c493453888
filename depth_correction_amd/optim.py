"""torch.optim.Adam for the handful of small parameter tensors this path optimises (model weights [1,P], pose corrections
[S,6]; train.py:138-160): the single-tensor update of torch.optim.Adam (no amsgrad / maximize) as ONE launch per fp64 GPU
parameter (dc_adam_step_device: step counter on the device, so a captured iteration replays correctly) instead of ~10 tensor operations and their Python dispatch -- 80 us of host time per step for two
parameters otherwise, more than the GPU needs for a whole C2 iteration.  Same arithmetic in the same order
(tests/test_gpu_api.py::test_dc_adam_equals_torch_adam); parameters that are not contiguous fp64 GPU tensors take the
tensor expressions of the same formulas."""
from __future__ import annotations

import math

import torch

__all__ = ['Adam']


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError('invalid Adam hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                st = self.state[p]
                native = (p.is_cuda and p.dtype == torch.float64 and g.dtype == torch.float64 and p.is_contiguous()
                          and g.is_contiguous())
                if not st:
                    # the native path counts its steps on the device, so that a captured iteration replays correctly
                    st['step'] = torch.zeros((), dtype=torch.int64, device=p.device) if native else 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format if native else torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(st['exp_avg'])
                m, v = st['exp_avg'], st['exp_avg_sq']
                if native and isinstance(st['step'], torch.Tensor):
                    from ._native import lib, check, ptr, stream_ptr
                    with torch.cuda.device(p.device):
                        check(lib().dc_adam_step_device(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(st['step']), 1.0, float(lr),
                                                        float(b1), float(b2), float(eps), float(wd), stream_ptr()),
                              'dc_adam_step_device')
                    torch.autograd.graph.increment_version(p)         # written through its pointer
                    continue
                if isinstance(st['step'], torch.Tensor):               # the parameter stopped qualifying: count on the host from here
                    st['step'] = int(st['step'].item())
                st['step'] += 1
                t = st['step']
                if wd != 0.0:
                    g = g.add(p, alpha=wd)
                m.lerp_(g, 1.0 - b1)
                v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                denom = (v.sqrt() / math.sqrt(1.0 - b2 ** t)).add_(eps)
                p.addcdiv_(m, denom, value=-lr / (1.0 - b1 ** t))
        return loss
