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


def test_two_local_sequences_keep_the_update_in_next_path(golden, nccl_group):
    """Config 3 with more sequences than GPUs: a rank that owns SEVERAL sequences still takes the Adam update of step t in the
    first launch of step t + 1 (dc_sequence_eval_after_update on its first sequence; the others follow on the stream and read
    the published weights) -- weights and sums follow the plain trainer (evaluate all -> sum -> all-reduce -> dc_adam_step)."""
    from depth_correction_amd.plan import SequencePlan, SequenceTrainer
    g = golden('room_k10')
    cfg = _cfg(g, float_type='float32')
    clouds, poses, _, ns, mask = _setup(g, cfg)
    mask2 = mask & (torch.arange(len(mask), device=mask.device) % 3 != 0)
    mk = lambda: [SequencePlan(clouds, poses, ns[0], mask), SequencePlan(clouds, poses, ns[0].clone(), mask2)]
    plain = SequenceTrainer(mk(), g['w'], g['exponent'], [poses, poses], lr=1e-2, distributed=True)
    fast = SequenceTrainer(mk(), g['w'], g['exponent'], [poses, poses], lr=1e-2, distributed=True, chained=True)
    assert fast.update_in_next and not plain.update_in_next
    for it in range(8):
        a, b = npy(fast.step()).copy(), npy(plain.step()).copy()
        assert fast.update_in_next
        assert a[1] == b[1] == float(mask.sum() + mask2.sum())
        np.testing.assert_allclose(a[0], b[0], rtol=1e-10)
        np.testing.assert_allclose(a[2:], b[2:], rtol=1e-8, atol=1e-12 * np.abs(b[2:]).max())
    fast.flush()
    np.testing.assert_allclose(npy(fast.w), npy(plain.w), rtol=1e-9)
    assert abs(npy(fast.w)[0] - float(g['w'].reshape(-1)[0])) > 1e-3


def _train_rank(rank, world, port, log_dir, result):
    """One RCCL rank of test_train_two_rccl_ranks: its own process (spawned by a parent that never touched the GPU from this
    test's point of view: the child initialises the device itself), GPU `rank`."""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    try:
        hist = _train_sequences('cuda', log_dir, None)          # index-less device: the launcher's set_device decides
        result[rank] = hist
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _native_run(mode, device, log_dir, distributed, loop_batch, n_it=9):
    """train() with the default (no-op) callbacks -- the loops without a host synchronisation per iteration -- on small sequences;
    returns ([(train loss, validation loss) per iteration] parsed from the progress lines (rank 0 prints them), best config)."""
    import contextlib
    import io
    import re
    from depth_correction_amd.config import Config, Loss, PoseCorrection
    from depth_correction_amd.dataset import KittiLikeDataset, RoomBoxDataset
    from depth_correction_amd.preproc import filtered_cloud
    from depth_correction_amd import train as T
    if mode == 'icp':
        cfg = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0, max_depth=12.0,
                     vp_dispersion_bounds=[], n_opt_iters=n_it, lr=2e-3, log_dir=log_dir, float_type='float64', device=device,
                     distributed=distributed, loop_batch=loop_batch, model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]})
        seqs = [[(filtered_cloud(c, cfg), p) for c, p in KittiLikeDataset(n_poses=3, n_rings=96, n_azimuth=384, seed_base=2000 + 100 * q)]
                for q in range(2)]
        vals = []
    else:
        cfg = Config(pose_correction=PoseCorrection.pose if mode == 'pose' else PoseCorrection.none, nn_k=8, nn_r=None, min_depth=0.0,
                     max_depth=float('inf'), grid_res=0.0, vp_dispersion_bounds=[], n_opt_iters=n_it, lr=2e-3, device=device,
                     float_type='float64', log_dir=log_dir, distributed=distributed, loop_batch=loop_batch,
                     model_kwargs={'w': [1e-3, -2e-3], 'exponent': [2.0, 4.0]})
        seqs = [list(RoomBoxDataset(n_pts=3000, n_poses=3, seed_base=1000 + 100 * q, dtype=np.float64)) for q in range(3)]
        seqs, vals = seqs[:2], seqs[2:]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        best = T.train(cfg, train_datasets=seqs, val_datasets=vals)
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith('It. ')]
    hist = [tuple(float(x) for x in re.findall(r'(?:train loss|val\.): (-?[0-9]+\.[0-9]+|nan)', ln)) for ln in lines]
    return np.asarray(hist), best


@pytest.mark.parametrize('mode', ['model', 'pose', 'icp'])
def test_train_sharded_native_loops_equal_the_plain_loop(tmp_path, monkeypatch, nccl_group, mode):
    """train() with sharded sequences (here: one RCCL rank with the collectives forced) takes the loops WITHOUT a host
    synchronisation per iteration -- model only: evaluation (+ the previous Adam update in its launch) -> joint sums -> ONE
    all-reduce (train._native_shared_loop); per-pose corrections on the map-consistency loss and on the ICP loss (BASELINE
    config 4's shape): evaluations -> joint sums -> ONE all-reduce -> finishing launches (train._native_pose_loop) -- and follows
    the reference's per-iteration loop of an unsharded run: losses per iteration, final weights and corrections."""
    from depth_correction_amd import train as train_mod
    dist = nccl_group
    took, calls = [], []
    for name in ('_native_loop', '_native_shared_loop', '_native_pose_loop', '_batched_loop'):
        fn = getattr(train_mod, name)
        monkeypatch.setattr(train_mod, name, (lambda f, n: (lambda *a, **k: (took.append(n), f(*a, **k))[1]))(fn, name))
    real = dist.all_reduce
    monkeypatch.setattr(dist, 'all_reduce', lambda t, *a, **k: (calls.append(t.numel()), real(t, *a, **k))[1])
    (tmp_path / 'plain').mkdir()
    (tmp_path / 'sharded').mkdir()
    ref, b0 = _native_run(mode, 'cuda:0', str(tmp_path / 'plain'), False, 1)
    assert took == [] and calls == []
    got, b1 = _native_run(mode, 'cuda:0', str(tmp_path / 'sharded'), None, 4)
    assert took == ['_native_shared_loop' if mode == 'model' else '_native_pose_loop'], took
    # one collective per iteration: [training | validation] x {loss, divisor, dL/dw} (+ a few at set-up: counts, which loop)
    per_it = [c for c in calls if c == 2 * (2 + 2)]
    assert len(per_it) == 9 and len(calls) <= 9 + 4, calls
    assert ref.shape == got.shape == (9, 2) and np.isfinite(ref).all()
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-12)
    assert ref[-1, 0] != ref[0, 0]
    sa, sb = torch.load(b0.model_state_dict), torch.load(b1.model_state_dict)
    for k in sa:
        np.testing.assert_allclose(sb[k].cpu().numpy(), sa[k].cpu().numpy(), rtol=1e-8, atol=1e-13)
    da, db = torch.load(b0.train_pose_deltas), torch.load(b1.train_pose_deltas)
    assert len(da) == len(db) == (0 if mode == 'model' else 2)
    for x, y in zip(da, db):
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=1e-7, atol=1e-10)


def _train_sequences(device, log_dir, distributed):
    """train() on three small room sequences with per-sequence pose corrections and the min-eigenvalue loss; per iteration
    [global train loss, w...]."""
    from depth_correction_amd.config import Config, PoseCorrection
    from depth_correction_amd.dataset import RoomBoxDataset
    from depth_correction_amd import train as T
    cfg = Config(pose_correction=PoseCorrection.sequence, nn_k=8, nn_r=None, min_depth=0.0, max_depth=float('inf'), grid_res=0.0,
                 vp_dispersion_bounds=[], n_opt_iters=6, lr=1e-3, device=device, float_type='float64', log_dir=log_dir,
                 distributed=distributed, model_kwargs={'w': [1e-3, -2e-3], 'exponent': [2.0, 4.0]})
    seqs = [list(RoomBoxDataset(n_pts=3000, n_poses=3, seed_base=1000 + 100 * q, dtype=np.float64)) for q in range(3)]
    hist = []

    class CB(T.TrainCallbacks):
        def train_loss(self, it, model, clouds, pose_deltas, poses, masks, loss_):
            hist.append([float(loss_.detach())] + model.w.detach().reshape(-1).tolist())

    T.train(cfg, callbacks=CB(), train_datasets=seqs, val_datasets=[])
    return np.asarray(hist)


def _native_rank(rank, world, port, log_dir, mode, result):
    """One RCCL rank of test_train_native_loops_on_two_rccl_ranks (its own spawned process, GPU `rank`)."""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    try:
        hist, best = _native_run(mode, 'cuda', log_dir, None, 4)
        result[rank] = (hist, best.train_pose_deltas)
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (the one-GPU box collects and skips it)')
@pytest.mark.parametrize('mode', ['model', 'pose', 'icp'])
def test_train_native_loops_on_two_rccl_ranks(tmp_path, mode):
    """The loops of test_train_sharded_native_loops_equal_the_plain_loop on TWO RCCL ranks (sequence q on rank q mod 2; the
    validation sequence of the min-eigenvalue modes lives on rank 0 only, so rank 1 joins the all-reduce with zeros): rank 0's
    progress lines follow the unsharded per-iteration loop, and its checkpoint holds both sequences' corrections."""
    import torch.multiprocessing as mp
    ref, _ = _native_run(mode, 'cuda:0', str(tmp_path / 'single'), False, 1)
    ctx = mp.get_context('spawn')
    result = ctx.Manager().dict()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_native_rank, args=(r, 2, port, str(tmp_path / 'sharded'), mode, result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    np.testing.assert_allclose(result[0][0], ref, rtol=1e-8, atol=1e-12)
    assert len(result[1][0]) == 0                                  # (rank 1 replays the bookkeeping silently)
    assert len(torch.load(result[0][1])) == (0 if mode == 'model' else 2)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (the one-GPU box collects and skips it)')
def test_train_two_rccl_ranks(tmp_path):
    """train() itself on TWO RCCL ranks (BASELINE config 3 / 4 in code: sequence q on rank q mod 2, per-sequence pose corrections
    with their owner, the model weights in one packed all-reduce per iteration) reaches its checkpoints and follows the
    single-process run: global loss and weights per iteration on both ranks, rank 0's checkpoint with every sequence's
    corrections."""
    import glob
    import torch.multiprocessing as mp
    ref = _train_sequences('cuda:0', str(tmp_path / 'single'), False)
    ctx = mp.get_context('spawn')
    result = ctx.Manager().dict()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_train_rank, args=(r, 2, port, str(tmp_path / 'sharded'), result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    for r in range(2):
        np.testing.assert_allclose(result[r], ref, rtol=1e-8, atol=1e-12)
    ckpt = sorted(glob.glob(os.path.join(str(tmp_path / 'sharded'), '*_pose_deltas.pth')))
    assert ckpt and len(torch.load(ckpt[-1])) == 3


def test_train_sequences_single_process(tmp_path):
    """The single-process arm of test_train_two_rccl_ranks on a one-GPU box (so the arm that is skipped here does not rot): six
    iterations over three sequences with per-sequence pose corrections; the model moves and a checkpoint with three corrections
    is written."""
    import glob
    hist = _train_sequences('cuda:0', str(tmp_path), False)
    assert hist.shape == (6, 3) and np.isfinite(hist).all() and not np.allclose(hist[0, 1:], hist[-1, 1:])
    ckpt = sorted(glob.glob(os.path.join(str(tmp_path), '*_pose_deltas.pth')))
    assert ckpt and len(torch.load(ckpt[-1])) == 3
