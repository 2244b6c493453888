import sys, json, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from depth_correction_amd.dataset import RoomBoxDataset
from depth_correction_amd.pipeline import build_sequence
ds = RoomBoxDataset(n_pts=200000, n_poses=10, seed_base=1000, dtype=np.float32)
scans_xyz = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
poses = np.stack([p for _, p in ds])
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    plan, info = build_sequence(scans_xyz, poses, k=10, dtype=torch.float32, device='cuda:0', stage_times=(rep>0))
    torch.cuda.synchronize(); t1=time.perf_counter()
    print(rep, round((t1-t0)*1e3,1), {k:round(v,1) for k,v in info['setup_ms'].items()})
    del plan, info
