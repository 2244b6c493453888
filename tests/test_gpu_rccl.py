"""The RCCL path on the GPU box: a real `nccl` process group (one rank: the box has one GPU) with the collective forced
(DC_FORCE_DIST=1), driving the multi-rank form of the native step -- evaluation -> reduction -> all-reduce of
[sum loss, count, dL/dw] -> Adam riding in the next launch -- against the single-process trainer.  Reference split:
train.py:166-175 (sequences are independent), loss.py:205-213 (pooled mean)."""
import os
import socket

import numpy as np
import pytest
import torch

from helpers import npy
from test_gpu_api import _cfg, _setup

pytestmark = pytest.mark.gpu


@pytest.fixture()
def nccl_group(monkeypatch):
    import torch.distributed as dist
    assert not dist.is_initialized()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    monkeypatch.setenv('DC_FORCE_DIST', '1')
    monkeypatch.setenv('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device('cuda:0'))
    try:
        yield dist
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def test_sequence_trainer_over_rccl_equals_single_process(golden, nccl_group):
    from depth_correction_amd.plan import SequencePlan, SequenceTrainer
    dist = nccl_group
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    calls = []
    real = dist.all_reduce

    def counting(tensor, *a, **kw):
        calls.append(tensor.numel())
        return real(tensor, *a, **kw)
    dist.all_reduce = counting
    try:
        g = golden('room_k10')
        cfg = _cfg(g, float_type='float32')
        clouds, poses, _, ns, mask = _setup(g, cfg)
        plan = SequencePlan(clouds, poses, ns[0], mask)
        single = SequenceTrainer([plan], g['w'], g['exponent'], [poses], lr=1e-2)
        multi = SequenceTrainer([plan], g['w'], g['exponent'], [poses], lr=1e-2, distributed=True, chained=True)
        assert multi.update_in_next and not multi.chained and multi.count == single.count
        n0 = len(calls)
        for it in range(8):
            a, b = npy(multi.step()).copy(), npy(single.step()).copy()
            assert a[1] == b[1]
            np.testing.assert_allclose(a[0], b[0], rtol=1e-11)
            np.testing.assert_allclose(a[2:], b[2:], rtol=1e-9, atol=1e-12 * np.abs(b[2:]).max())
        multi.flush()
        torch.cuda.synchronize()
        assert len(calls) - n0 == 8 and set(calls[n0:]) == {2 + 2}          # ONE collective of [sum, count, dL/dw] per step
        np.testing.assert_allclose(npy(multi.w), npy(single.w), rtol=1e-10)
        assert abs(npy(multi.w)[0] - g['w'][0]) > 1e-3
    finally:
        dist.all_reduce = real


def test_grad_reducer_and_object_gather_over_rccl(nccl_group):
    """distributed.GradReducer (train()'s one collective) and the checkpoint-time object gather on a real RCCL group."""
    from depth_correction_amd.distributed import GradReducer, gather_objects
    w = torch.nn.Parameter(torch.tensor([[1.0, 2.0]], dtype=torch.float64, device='cuda:0'))
    local = torch.nn.Parameter(torch.ones((3, 6), dtype=torch.float64, device='cuda:0'))
    loss = (w.sum() * 3.0 + local.sum()) * 4.0                    # weighted local loss, weight 4
    loss.backward()
    red = GradReducer([w], [local])
    mean, total = red.reduce(loss, 4.0)
    torch.cuda.synchronize()
    assert float(total) == 4.0 and abs(float(mean) - float(loss) / 4.0) < 1e-12
    np.testing.assert_allclose(npy(w.grad), np.full((1, 2), 3.0), rtol=1e-15)
    np.testing.assert_allclose(npy(local.grad), np.ones((3, 6)), rtol=1e-15)
    # world size 1 short-cuts the object gather; the device argument is what a multi-rank launcher relies on
    out = gather_objects(([0], [torch.ones(2)]), device='cuda:0')
    assert len(out) == 1 and out[0][0] == [0]
