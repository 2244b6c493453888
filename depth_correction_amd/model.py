"""Depth correction models with the reference's class API (model.py:70-354).

``d' = f(d, gamma)`` with gamma the incidence angle; only points selected by the cloud's mask are corrected
(BaseModel.forward, model.py:76-78).  The classes are ordinary ``torch.nn.Module``s (``w`` is ``[1,P]`` float64,
state-dict compatible with the reference); their forward is the reference's elementwise expression.  Inside the
fused training path (plan.SequencePlan) the same formula and its gradient are evaluated by the HIP point kernels
(dc_points_fwd / dc_consistency_bwd) straight from ``model.w`` / ``model.exponent``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .depth_cloud import DepthCloud

__all__ = ['BaseModel', 'InvCos', 'Linear', 'load_model', 'model_by_name', 'Polynomial', 'ScaledPolynomial',
           'ScaledInvCos']


class BaseModel(torch.nn.Module):
    def __init__(self, device=torch.device('cpu')):
        super().__init__()
        self.device = device

    def forward(self, dc: DepthCloud) -> DepthCloud:
        return self.correct_depth(dc, dc.mask)

    def correct_depth(self, dc: DepthCloud, mask=None) -> DepthCloud:
        return dc

    def inverse(self, dc: DepthCloud, mask=None) -> DepthCloud:
        return dc

    def _apply_to_depth(self, dc, mask, fun, op=None):
        """New cloud whose depth is fun(depth, inc_angles) on the masked points (never in place).  ``op``: the same function as
        a dc_correct_depth code, used when nothing here needs a gradient (the online node, evaluation under no_grad) -- one
        kernel instead of the boolean gather, pow, GEMM, elementwise passes and index_put of the tensor expression."""
        assert dc.inc_angles is not None
        out = dc.copy()
        if op is not None and self._no_grad_on_device(dc, mask):
            w, e = self.kernel_params()
            out.depth = ops.correct_depth(dc.depth.contiguous(), dc.inc_angles.contiguous(), mask, w, e, op)
            return out
        if mask is None:
            out.depth = fun(dc.depth, dc.inc_angles)
        else:
            depth = dc.depth.clone()
            # float32 clouds with the (always float64) weights: the reference's index_put raises on the dtype mismatch here;
            # the corrected depths are rounded to the cloud's dtype instead
            depth[mask] = fun(dc.depth[mask], dc.inc_angles[mask]).to(depth.dtype)
            out.depth = depth
        return out

    def _no_grad_on_device(self, dc, mask):
        d, g = dc.depth, dc.inc_angles
        if not (d.is_cuda and d.dtype in (torch.float32, torch.float64) and g.dtype == d.dtype and g.device == d.device
                and d.dim() == 2 and d.shape[1] == 1 and g.shape == d.shape):
            return False
        if mask is not None and not (mask.dtype == torch.bool and mask.shape == d.shape[:1] and mask.device == d.device):
            return False
        w, e = self.kernel_params()
        if w.device != d.device or w.dtype != torch.float64 or e.dtype != torch.float64:
            return False
        return not (torch.is_grad_enabled() and any(t.requires_grad for t in (d, g, w, e)))

    # ---- the fused HIP kernels' view of a model: its kind and its parameters as one [1, P] tensor + exponents ------
    kernel_kind = None        # name understood by the point kernels (_native.MODEL_KINDS); None: tensor path only

    def kernel_params(self):
        """(w [1,P], exponent [1,P]) for the kernels; ``w`` stays connected to the model's parameters by autograd."""
        raise NotImplementedError()

    def _zero_exponent(self, n, device):
        """The constant exponent vector of the models that have none: ONE tensor per (model, device), so that the plans' basis
        rows -- keyed by the identity and version of the exponent tensor -- survive from one evaluation to the next."""
        z = self.__dict__.get('_zero_exp')
        if z is None or z.shape[1] != n or z.device != device:
            z = torch.zeros((1, n), dtype=torch.float64, device=device)
            self.__dict__['_zero_exp'] = z
        return z

    def __str__(self):
        return 'BaseModel()'

    def construct(self, *args, **kwargs):
        return type(self)(*args, **kwargs)

    def detach(self):
        return self.construct(**{k: v.detach() for k, v in self.named_parameters()})

    def clone(self):
        return self.construct(**{k: v.clone() for k, v in self.named_parameters()})


class Linear(BaseModel):
    kernel_kind = 'Linear'

    def kernel_params(self):
        w = torch.stack([self.w0.reshape(()), self.w1.reshape(()), self.b.reshape(())]).reshape(1, 3)
        return w, self._zero_exponent(3, w.device)

    def __init__(self, w0=1.0, w1=0.0, b=0.0, uniform_weights=False, device=torch.device('cpu')):
        super().__init__(device=device)
        if uniform_weights:
            w0 = torch.nn.init.uniform_(torch.as_tensor(w0), -0.95, 1.05)
            w1 = torch.nn.init.uniform_(torch.as_tensor(w1), -0.05, 0.05)
            b = torch.nn.init.uniform_(torch.as_tensor(b), -0.05, 0.05)
        self.w0 = torch.nn.Parameter(torch.as_tensor(w0, device=device))
        self.w1 = torch.nn.Parameter(torch.as_tensor(w1, device=device))
        self.b = torch.nn.Parameter(torch.as_tensor(b, device=device))

    def correct_depth(self, dc, mask=None):
        return self._apply_to_depth(dc, mask, lambda d, g: self.w0 * d + self.w1 * g + self.b)

    def inverse(self, dc, mask=None):
        raise NotImplementedError()

    def __str__(self):
        return 'Linear(%.6g, %.6g, %.6g)' % (self.w0.item(), self.w1.item(), self.b.item())


class _PolynomialBase(BaseModel):
    """bias(gamma) = sum_k w_k gamma^e_k with fixed or learnable exponents (model.py:151-179, 220-248)."""
    def kernel_params(self):
        return self.w, self.exponent

    def __init__(self, p0=None, p1=None, w=None, exponent=None, learnable_exponents=False, device=torch.device('cpu')):
        super().__init__(device=device)
        self.legacy = exponent is None
        if exponent is None:
            assert w is None, w
            exponent, w = [2.0, 4.0], [p0 or 0.0, p1 or 0.0]
        if w is None:
            w = [0.0] * len(exponent)
        elif isinstance(w, float):
            w = [w]
        w = torch.as_tensor(w, dtype=torch.float64, device=device).view((1, -1))
        assert w.numel() == len(exponent), (w, exponent)
        self.w = torch.nn.Parameter(w)
        exponent = torch.as_tensor(exponent, dtype=torch.float64, device=device).view((1, -1))
        self.exponent = torch.nn.Parameter(exponent) if learnable_exponents else exponent

    def bias(self, inc_angles):
        assert inc_angles.dim() == 2 and inc_angles.shape[1] == 1
        return torch.matmul(torch.pow(inc_angles, self.exponent), self.w.t()).view((-1, 1))

    def to(self, *args, **kwargs):
        ret = super().to(*args, **kwargs)
        if not isinstance(ret.exponent, torch.nn.Parameter):
            ret.exponent = ret.exponent.to(*args, **kwargs)
        return ret

    def __str__(self):
        terms = ', '.join('%.6gx^%.6g' % (w, e) for w, e in zip(self.w.detach().flatten().tolist(), self.exponent.detach().flatten().tolist()))
        return '%s(%s)' % (type(self).__name__, terms)


class Polynomial(_PolynomialBase):
    """d' = d - bias(gamma)  (model.py:149-215)."""
    kernel_kind = 'Polynomial'

    def correct_depth(self, dc, mask=None):
        return self._apply_to_depth(dc, mask, lambda d, g: d - self.bias(g), op=0)

    def inverse(self, dc, mask=None):
        if mask is None:       # the reference's two branches differ (model.py:201-213); kept as is
            return self._apply_to_depth(dc, None, lambda d, g: d / (1. - self.bias(g)), op=3)
        return self._apply_to_depth(dc, mask, lambda d, g: d + self.bias(g), op=1)


class ScaledPolynomial(_PolynomialBase):
    """d' = d (1 - bias(gamma))  (model.py:218-286)."""
    kernel_kind = 'ScaledPolynomial'

    def correct_depth(self, dc, mask=None):
        return self._apply_to_depth(dc, mask, lambda d, g: d * (1. - self.bias(g)), op=2)

    def inverse(self, dc, mask=None):
        return self._apply_to_depth(dc, mask, lambda d, g: d / (1. - self.bias(g)), op=3)


class InvCos(BaseModel):
    kernel_kind = 'InvCos'

    def kernel_params(self):
        w = self.p0.reshape(1, 1)
        return w, self._zero_exponent(1, w.device)

    def __init__(self, p0=0.0, device=torch.device('cpu')):
        super().__init__(device=device)
        self.p0 = torch.nn.Parameter(torch.as_tensor(p0, device=device))

    def correct_depth(self, dc, mask=None):
        return self._apply_to_depth(dc, mask, lambda d, g: d - self.p0 / torch.cos(g))

    def inverse(self, dc, mask=None):
        raise NotImplementedError()

    def __str__(self):
        return 'InvCos(%.6g)' % (self.p0.item(),)


class ScaledInvCos(BaseModel):
    kernel_kind = 'ScaledInvCos'

    def kernel_params(self):
        w = self.p0.reshape(1, 1)
        return w, self._zero_exponent(1, w.device)

    def __init__(self, p0=0.0, device=torch.device('cpu')):
        super().__init__(device=device)
        self.p0 = torch.nn.Parameter(torch.as_tensor(p0, device=device))

    def correct_depth(self, dc, mask=None):
        return self._apply_to_depth(dc, mask, lambda d, g: d * (1. - self.p0 / torch.cos(g).abs()))

    def inverse(self, dc, mask=None):
        return self._apply_to_depth(dc, mask, lambda d, g: d / (1. - self.p0 / torch.cos(g).abs()))

    def __str__(self):
        return 'ScaledInvCos(%.6g)' % (self.p0.item(),)


def model_by_name(name):
    classes = {c.__name__: c for c in (BaseModel, InvCos, Linear, Polynomial, ScaledInvCos, ScaledPolynomial)}
    assert name in classes, name
    return classes[name]


def load_model(class_name=None, model_args=None, model_kwargs=None, state_dict=None, device=None, cfg=None,
               eval_mode=True):
    """Model factory with the reference's precedence rules (model.py:19-67)."""
    if cfg is not None:
        class_name = cfg.model_class if class_name is None else class_name
        model_args = (cfg.model_args[:] if cfg.model_args else []) if model_args is None else model_args
        model_kwargs = (cfg.model_kwargs.copy() if cfg.model_kwargs else {}) if model_kwargs is None else model_kwargs
        state_dict = cfg.model_state_dict if state_dict is None else state_dict
        device = cfg.device if device is None else device
    model_args, model_kwargs = model_args or [], dict(model_kwargs or {})
    if isinstance(state_dict, str) and state_dict:
        print('Loading model state from %s.' % state_dict)
        state_dict = torch.load(state_dict)
    if isinstance(device, str):
        device = torch.device(device)
    model_kwargs.setdefault('device', device if device is not None else torch.device('cpu'))
    model = model_by_name(class_name)(*model_args, **model_kwargs)
    if state_dict:
        model.load_state_dict(state_dict)
    if eval_mode:
        model.eval()
    if device is not None:
        model.to(device)
    return model
