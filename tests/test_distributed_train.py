"""train() with the sequences sharded over two gloo ranks against the same train() in one process, on the CPU.

What is under test is the rank-aware part of depth_correction_amd.train / distributed.GradReducer (BASELINE configs 3
and 4 in code): round-robin ownership of sequences, per-sequence pose corrections staying with their owner, the model
weights and a PoseCorrection.common 6-vector travelling in ONE packed all-reduce together with the weighted loss, a
rank without validation sequences, rank-0 checkpoints holding every rank's pose corrections.  The per-sequence
arithmetic is injected from the oracle (module attributes of ``train`` are replaced inside every process), because
the HIP evaluators need a GPU; the optimisers, the pose chain (``create_corrected_poses``) and the model are the real ones.
"""
import glob
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TRAIN, N_VAL, N_ITERS = 3, 1, 4


class _Obj(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _install_oracle_backend():
    """Replace the GPU-backed set-up / evaluation functions the training loop calls by CPU ones built on the oracle."""
    for p in (ROOT, os.path.join(ROOT, 'oracle')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dc_oracle as O
    from depth_correction_amd import train as T
    from depth_correction_amd.eval import create_corrected_poses

    def load_sequences(datasets, cfg):
        all_clouds, all_poses = [], []
        for ds in datasets:
            scans, poses = [], []
            for cloud, pose in ds:
                pts = torch.as_tensor(np.stack([cloud[f] for f in 'xyz'], 1).astype(np.float64))
                depth = pts.norm(dim=-1, keepdim=True)
                dirs = pts / depth
                nbr = torch.as_tensor(O.knn_bruteforce(pts.numpy(), 8)[1])
                f = O.features(pts, nbr, dirs)
                scans.append(dict(vps=torch.zeros_like(pts), dirs=dirs, depth=depth, inc=f['inc_angles'],
                                  normals=f['normals'], mask=None))
                poses.append(pose)
            all_clouds.append(scans)
            all_poses.append(torch.as_tensor(np.stack(poses)))
        return all_clouds, all_poses

    def world_points(scans, poses):
        return [O.points_from(*O.transform_cloud(s['vps'], s['dirs'], P), s['depth']) for s, P in zip(scans, poses)]

    def global_cloud(clouds=None, poses=None, **kw):
        x = torch.cat(world_points(clouds, poses))
        return _Obj(points=x, mask=None)

    def establish_neighborhoods(cloud=None, cfg=None, **kw):
        return torch.as_tensor(O.knn_bruteforce(cloud.points.numpy(), cfg.nn_k)[1]), None

    def global_cloud_mask(cloud, mask, cfg):
        return torch.arange(len(cloud.points)) % 7 != 0

    def icp_masks(all_clouds, all_poses, ratio):
        out = []
        for scans, poses in zip(all_clouds, all_poses):
            x = world_points(scans, poses)
            seq = []
            for j in range(len(scans) - 1):
                m1, i2, _ = O.nn1_correspondences(x[j].numpy(), x[j + 1].numpy(), ratio)
                seq.append((torch.as_tensor(m1), torch.as_tensor(i2)))
            out.append(seq)
        return out

    def p2plane64(points, normals, masks):
        """O.point_to_plane's formula without its cast of the points to float32 (loss.py:436-437).  Through that cast
        the backward carries the upstream factor in float32, so the gradients of a sharded and of a single-process run
        differ by ~1e-9 absolute; Adam turns that into O(lr) differences wherever a gradient component nearly cancels.
        This test is about the sharding logic, so its stand-in evaluator stays in fp64."""
        total = 0.0
        for i, (mask1, mask2) in enumerate(masks):
            a, b = points[i][mask1], points[i + 1][mask2]
            n1, n2 = normals[i][mask1], normals[i + 1][mask2]
            d12 = torch.linalg.norm((n1 * (b - a)).sum(-1, keepdim=True) * n1, dim=-1).mean()
            d21 = torch.linalg.norm((n2 * (a - b)).sum(-1, keepdim=True) * n2, dim=-1).mean()
            total = total + 0.5 * (d12 + d21)
        return total / len(masks)

    def eval_loss_clouds(clouds, poses, pose_deltas, masks, ns, model, loss_fun, cfg):
        poses_upd = create_corrected_poses(poses, pose_deltas, cfg)
        if cfg.loss == 'icp_loss':
            total = 0.0
            for scans, P, m in zip(clouds, poses_upd, masks):
                pts, nrm = [], []
                for s, Ts in zip(scans, P):
                    d = O.model_apply(s['depth'], s['inc'], s['mask'], model.w, model.exponent)
                    v, r, n = O.transform_cloud(s['vps'], s['dirs'], Ts, normals=s['normals'])
                    pts.append(O.points_from(v, r, d)), nrm.append(n)
                total = total + p2plane64(pts, nrm, m)
            return total / len(clouds), [None] * len(clouds), poses_upd, None
        total, count, views = 0.0, 0.0, []
        for scans, P, nn, m in zip(clouds, poses_upd, ns, masks):
            s, _ = O.eval_sequence(scans, P, model.w, model.exponent, nn[0], m, reduction='sum')
            total, count = total + s, count + float(m.sum())
            views.append(_Obj(count=float(m.sum())))
        return total / count, views, poses_upd, views

    T._load_sequences, T.global_cloud, T.establish_neighborhoods = load_sequences, global_cloud, establish_neighborhoods
    T.global_cloud_mask, T._icp_masks, T.eval_loss_clouds = global_cloud_mask, icp_masks, eval_loss_clouds
    return T


def _datasets():
    from depth_correction_amd.dataset import RoomBoxDataset
    mk = lambda q: list(RoomBoxDataset(n_pts=250, n_poses=3, seed_base=1000 + 100 * q))
    return [mk(q) for q in range(N_TRAIN)], [mk(10 + q) for q in range(N_VAL)]


def _cfg(loss, pose_correction, log_dir, distributed):
    from depth_correction_amd.config import Config
    return Config(loss=loss, pose_correction=pose_correction, nn_k=6, nn_r=None, n_opt_iters=N_ITERS, lr=2e-4, device='cpu',
                  float_type='float64', log_dir=log_dir, distributed=distributed,
                  model_kwargs={'w': [1e-3, -2e-3], 'exponent': [2.0, 4.0]})


def _run(loss, pose_correction, log_dir, distributed):
    T = _install_oracle_backend()
    torch.set_num_threads(1)
    hist, last = [], {}

    class CB(T.TrainCallbacks):
        def train_loss(self, it, model, clouds, pose_deltas, poses, masks, loss_):
            hist.append([float(loss_.detach())] + model.w.detach().reshape(-1).tolist())
            last['deltas'] = [d.detach().clone() for d in pose_deltas]      # this rank's sequences, before the step

        def val_loss(self, it, model, clouds, pose_deltas, poses, masks, loss_):
            hist[-1].append(float(loss_.detach()))

    train_ds, val_ds = _datasets()
    best = T.train(_cfg(loss, pose_correction, log_dir, distributed), callbacks=CB(), train_datasets=train_ds, val_datasets=val_ds)
    return np.asarray(hist), best, last['deltas']


def _worker(rank, world, port, loss, pose_correction, log_dir, result):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    hist, best, deltas = _run(loss, pose_correction, log_dir, None)
    result[rank] = (hist, deltas)
    dist.destroy_process_group()


@pytest.mark.parametrize('loss,pose_correction', [('icp_loss', 'pose'), ('icp_loss', 'common'), ('min_eigval_loss', 'sequence')])
def test_sharded_train_equals_single_process(tmp_path, loss, pose_correction):
    ref_dir, dist_dir = str(tmp_path / 'single'), str(tmp_path / 'sharded')
    ref, ref_best, ref_deltas = _run(loss, pose_correction, ref_dir, False)
    assert ref.shape == (N_ITERS, 4) and not np.allclose(ref[0, 1:3], ref[-1, 1:3])         # the model moved
    ctx = mp.get_context('spawn')
    result = ctx.Manager().dict()
    port = 29500 + (os.getpid() * 7 + len(loss) + len(pose_correction)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, loss, pose_correction, dist_dir, result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    rtol, atol = 1e-10, 1e-14
    for r in range(2):
        # per iteration: global train loss, model weights before the step, global validation loss -- on EVERY rank
        np.testing.assert_allclose(result[r][0], ref, rtol=rtol, atol=atol)
    assert np.array_equal(result[0][0], result[1][0])                                          # replicas in lock step
    # the pose corrections entering the last iteration: every sequence's, on the rank that owns it (q mod 2)
    for q in range(N_TRAIN):
        torch.testing.assert_close(result[q % 2][1][q // 2], ref_deltas[q], rtol=1e-8, atol=1e-12)
    assert all(float(d.abs().max()) > 0 for d in ref_deltas)                                   # every correction moved
    # rank 0's checkpoint carries the pose corrections of all sequences, in sequence order
    ref_ckpt = sorted(glob.glob(os.path.join(ref_dir, '*_pose_deltas.pth')))[-1]
    got_ckpt = sorted(glob.glob(os.path.join(dist_dir, '*_pose_deltas.pth')))[-1]
    assert os.path.basename(ref_ckpt) == os.path.basename(got_ckpt)
    want, got = torch.load(ref_ckpt), torch.load(got_ckpt)
    assert len(got) == len(want) == N_TRAIN
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-8, atol=1e-12)


def test_grad_reducer_single_process_is_identity():
    from depth_correction_amd.distributed import GradReducer
    w = torch.nn.Parameter(torch.tensor([[1.0, 2.0]], dtype=torch.float64))
    d = torch.nn.Parameter(torch.zeros((2, 6), dtype=torch.float64))
    loss = (w ** 2).sum() + (d + 1.0).sum()
    (3.0 * loss).backward()
    mean, total = GradReducer([w], [d]).reduce(3.0 * loss, 3.0)
    assert float(total) == 3.0 and abs(float(mean) - float(loss)) < 1e-15
    torch.testing.assert_close(w.grad, 2 * w.detach())
    torch.testing.assert_close(d.grad, torch.ones_like(d))
