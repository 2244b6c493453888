#!/usr/bin/env python3
"""Host-side cost of the drop-in loop (eval_loss_clouds-style autograd Function + torch.optim.Adam) per step, split by
phase, next to the GPU time of the same steps (C2 workload)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from depth_correction_amd.dataset import RoomBoxDataset
from depth_correction_amd.pipeline import build_sequence
from depth_correction_amd.plan import consistency_loss

dev = torch.device('cuda', 0)
ds = RoomBoxDataset(n_pts=200000, n_poses=10, seed_base=1000, dtype=np.float32)
scans_xyz = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
poses = np.stack([p for _, p in ds])
plan, info = build_sequence(scans_xyz, poses, k=10, dtype=torch.float32, device=dev)
w = torch.nn.Parameter(torch.tensor([[1e-3, 2e-3]], dtype=torch.float64, device=dev))
e = torch.tensor([[2.0, 4.0]], dtype=torch.float64, device=dev)
fused = '--fused' in sys.argv
opt = torch.optim.Adam([w], lr=1e-3, **({'fused': True} if fused else {}))
P = info['poses']
acc = dict(zero=0.0, fwd=0.0, div=0.0, bwd=0.0, step=0.0)
n = 300
for it in range(n + 20):
    if it == 20:
        torch.cuda.synchronize(); t_all = time.perf_counter(); acc = {k: 0.0 for k in acc}
    t0 = time.perf_counter(); opt.zero_grad(set_to_none=False)
    t1 = time.perf_counter(); s, cnt = consistency_loss(plan, w, e, P)
    t2 = time.perf_counter(); loss = s / cnt
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter()
    for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[k] += d
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / n * 1e6
if '--graph' in sys.argv:
    # the whole iteration captured once into a hipGraph (torch.cuda.graph) and replayed: no Python, no launches on the host
    w2 = torch.nn.Parameter(torch.tensor([[1e-3, 2e-3]], dtype=torch.float64, device=dev))
    opt2 = torch.optim.Adam([w2], lr=1e-3, capturable=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            opt2.zero_grad(set_to_none=True)
            s, cnt = consistency_loss(plan, w2, e, P)
            (s / cnt).backward()
            opt2.step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    opt2.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        s, cnt = consistency_loss(plan, w2, e, P)
        static_loss = s / cnt
        static_loss.backward()
        opt2.step()
    torch.cuda.synchronize()
    w_before = w2.detach().clone()
    t0 = time.perf_counter()
    for _ in range(n):
        graph.replay()
    torch.cuda.synchronize()
    print(json.dumps({'graph_replay_us_per_step': round((time.perf_counter() - t0) / n * 1e6, 1), 'loss': float(static_loss),
                      'w_moved': float((w2.detach() - w_before).abs().max())}))
print(json.dumps({'fused_adam': fused, 'us_per_step_wall': round(total, 1), 'host_us': {k: round(v / n * 1e6, 1) for k, v in acc.items()}}))
