import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from depth_correction_amd.dataset import RoomBoxDataset
from depth_correction_amd.filters import filter_grid
from depth_correction_amd.pipeline import build_sequence
ds = RoomBoxDataset(n_pts=200_000, n_poses=10, seed_base=1000, dtype=np.float32)
scans = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
poses = np.stack([p for _, p in ds])
rng = np.random.default_rng(135)
kept = [filter_grid(s, 0.2, keep='random', rng=rng) for s in scans]
for r in (0.25, 0.4):
    plan, info = build_sequence(kept, poses, k=None, r=r, dtype=torch.float32)
    w = torch.tensor([1e-3, 2e-3], dtype=torch.float64, device='cuda:0'); e = torch.tensor([2.0, 4.0], dtype=torch.float64, device='cuda:0')
    out = torch.zeros(2 + 4 + 12 * plan.n_scans, dtype=torch.float64, device='cuda:0')
    plan.eval_native(w, e, plan.poses12(info['poses']), out, want_grad=True, want_pose=True)
    torch.cuda.synchronize()
    t = plan._pose_table[1]
    ws = t['wseg'].cpu().numpy().astype(np.int64).reshape(-1, 4, plan.n_scans + 1)
    print('r', r, 'n', plan.n, 'slot rows padded/orig', plan.pose_slot_rows, 'max_rows', plan.fwd_table.max_rows,
          'mean wave slots', ws[:, :, -1].mean(), 'nonempty scans per wave', (np.diff(ws, axis=2) > 0).sum(2).mean())
