"""The N > 1 path on the CPU: world_size-2 gloo processes, sequences sharded round robin, ONE all-reduce of
[sum loss, count, dL/dw] per step, identical Adam steps on every rank.  The per-sequence evaluation is injected
from the oracle (the HIP evaluator needs a GPU); what is tested is the sharding / reduction / optimiser logic of
plan.SequenceTrainer and distributed.py, against a single-process run over all sequences."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SEQ, N_STEPS = 3, 4


class _Seq:
    """Stand-in for a SequencePlan: holds the oracle inputs of one small sequence."""

    def __init__(self, q):
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import dc_oracle as O
        from depth_correction_amd.dataset import RoomBoxDataset
        self.O = O
        ds = RoomBoxDataset(n_pts=300, n_poses=2, seed_base=1000 + 100 * q)
        self.scans, poses = [], []
        for cloud, pose in ds:
            pts = torch.as_tensor(np.stack([cloud[f] for f in 'xyz'], 1))
            depth = pts.norm(dim=-1, keepdim=True)
            dirs = pts / depth
            _, ind = O.knn_bruteforce(pts.numpy(), 6)
            f = O.features(pts, torch.as_tensor(ind), dirs)
            self.scans.append(dict(vps=torch.zeros_like(pts), dirs=dirs, depth=depth, inc=f['inc_angles'], mask=None))
            poses.append(pose)
        self.poses = torch.as_tensor(np.stack(poses))
        x0 = torch.cat([O.points_from(*O.transform_cloud(s['vps'], s['dirs'], T), s['depth'])
                        for s, T in zip(self.scans, self.poses)])
        self.nbr = torch.as_tensor(O.knn_bruteforce(x0.numpy(), 6)[1])
        self.mask = torch.arange(len(x0)) % (q + 2) != 0
        self.count, self.n_scans = float(self.mask.sum()), len(self.scans)


def _oracle_eval(seq, w, e, poses, out):
    wt = w.detach().clone().reshape(1, -1).requires_grad_(True)
    s, _ = seq.O.eval_sequence(seq.scans, poses, wt, e.reshape(1, -1), seq.nbr, seq.mask, reduction='sum')
    s.backward()
    out.zero_()
    out[0], out[1] = s.detach(), seq.count
    out[2:2 + w.numel()] = wt.grad.reshape(-1)


def _make_trainer(seqs, distributed):
    from depth_correction_amd.plan import SequenceTrainer
    opt_state = {}

    def adam(grad_sum):          # torch.optim.Adam itself as the reference for the native kernel's semantics
        tr = opt_state['trainer']
        if 'opt' not in opt_state:
            opt_state['p'] = torch.nn.Parameter(tr.w)
            opt_state['opt'] = torch.optim.Adam([opt_state['p']], lr=tr.lr)
        opt_state['p'].grad = grad_sum / tr.count
        opt_state['opt'].step()

    tr = SequenceTrainer(seqs, [1e-3, 2e-3], [2.0, 4.0], [s.poses for s in seqs], lr=1e-2, distributed=distributed,
                         evaluate=_oracle_eval, adam=adam, device=torch.device('cpu'))
    opt_state['trainer'] = tr
    return tr


def _worker(rank, world, port, result):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from depth_correction_amd.distributed import shard_sequences, world_info
    assert world_info() == (rank, world)
    mine = shard_sequences(N_SEQ, rank, world)
    tr = _make_trainer([_Seq(q) for q in mine], distributed=True)
    hist = []
    for _ in range(N_STEPS):
        acc = tr.step()
        hist.append(acc.clone())
    result[rank] = (mine, torch.stack(hist), tr.w.detach().clone(), tr.count)
    dist.destroy_process_group()


def test_two_ranks_equal_single_process():
    torch.set_num_threads(2)
    single = _make_trainer([_Seq(q) for q in range(N_SEQ)], distributed=False)
    ref = torch.stack([single.step().clone() for _ in range(N_STEPS)])
    ctx = mp.get_context('spawn')
    result = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert sorted(result[0][0] + result[1][0]) == list(range(N_SEQ)) and result[0][0] == [0, 2]
    for r in range(2):
        _, hist, w, count = result[r]
        assert count == single.count                                   # global masked-point count on every rank
        torch.testing.assert_close(hist, ref, rtol=1e-10, atol=1e-14)   # [sum loss, count, dL/dw] after the all-reduce
        torch.testing.assert_close(w, single.w.detach(), rtol=1e-12, atol=0)
    assert torch.equal(result[0][2], result[1][2])                      # ranks stay bit-identical
    assert not torch.equal(ref[0, 2:], ref[-1, 2:])                     # and the parameters actually moved


def test_shard_sequences():
    from depth_correction_amd.distributed import shard_sequences
    assert [shard_sequences(10, r, 4) for r in range(4)] == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    assert shard_sequences(2, 3, 8) == []
