#!/usr/bin/env python3
"""Headline benchmark: points/sec through min_eigval_loss forward + backward (+ Adam) on MI355X.

Workload (BASELINE.json configs[2], SURVEY 8d "C2"): one sequence of 10 overlapping 200k-point room-box
scans (global cloud N = 2 M), nn_k = 10, ScaledPolynomial model (w = [1e-3, 2e-3], exponents [2, 4]),
min_eigval_loss with normalization, masks from the default eigenvalue-ratio bounds + min 5 valid
neighbours, Adam lr 1e-3.  One "step" = model apply -> pose transform -> global cloud -> neighbourhood
covariance / eigen / loss -> backward -> Adam step on the whole global cloud; the k-NN build is set-up
(amortised once, train.py:172-175) and reported separately.  With --gpus N every rank owns one such
sequence (configs[3] shape: weak scaling, sequences are independent) and the ranks exchange
one RCCL all-reduce of [sum loss, dL/dw] per step.

Prints ONE JSON line (contract in the task statement).  `roofline` describes the dominant kernel with
MEASURED quantities: its duration (HIP events stamped with the dispatch's own start / end inside the timed
region), its HBM traffic and VALU instruction count from the rocprofv3 PMC passes committed under profiles/
(looked up by the exact kernel instantiation this run launched; null when no profile of that kernel exists)
and a live lower bound (`compulsory_bytes`: every array the kernel touches, once).  `bound` is whichever of
the HBM and the VALU-issue fractions is higher.  The SURVEY 8d algorithmic byte count -- which prices every
gather as an HBM access although Morton order + LDS staging serve them on chip -- is reported separately under
`algorithmic`, never as a fraction of the HBM peak.  `cpu_baseline` is the oracle (= CPU restatement of the
reference algorithm) timed on this host's cores on the same C2 workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault('DC_ENABLE_ABLATIONS', '1')          # this tool flips the library's A-B switches (dc_set_option)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
CLOCK_GHZ, N_SIMD = 2.4, 1024     # peak engine clock; 256 CUs x 4 SIMDs; one wave64 VALU instruction per 4 cycles per SIMD


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--workload', default='c2', choices=['c2', 'c4'],
                    help="c2 (default): BASELINE config 2 / 3, the metric's workload; c4: BASELINE config 4 shape -- KITTI-360-like "
                         'sequences, one per rank, joint model + pose optimisation with the point-to-plane ICP loss through train()')
    ap.add_argument('--c4-scans', type=int, default=10, help='--workload c4: scans per sequence')
    ap.add_argument('--c4-sequences', type=int, default=0, help='--workload c4: sequences (default: one per rank)')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--scans', type=int, default=10)
    ap.add_argument('--points', type=int, default=200_000, help='points per scan')
    ap.add_argument('--k', type=int, default=10)
    ap.add_argument('--dtype', default='float32', choices=['float32', 'float64'])
    ap.add_argument('--cpu-scans', type=int, default=10, help='scans in the CPU baseline (10 = the full C2 workload; 0 = skip)')
    ap.add_argument('--cpu-iters', type=int, default=5, help='timed iterations of the fp64 all-cores CPU baseline (BASELINE.md 3: >= 5 after one warm-up)')
    ap.add_argument('--cpu-variants', type=int, default=1, help='also time the fp32 and the 8-thread variants (BASELINE.md 3)')
    ap.add_argument('--no-sort', action='store_true', help='keep the scan-major point order (ablation)')
    ap.add_argument('--point-format', default='auto', choices=['auto', 'q32', 'float'])
    ap.add_argument('--active-only', action='store_true',
                    help='evaluate only masked points as neighbourhood centres (identical loss / gradients, see DESIGN.md)')
    ap.add_argument('--degree-sort', action='store_true', help='order points by in-degree inside 256-point blocks (ablation)')
    ap.add_argument('--autograd', action='store_true', help='drop-in loop: torch autograd + torch.optim.Adam')
    ap.add_argument('--graph', action='store_true',
                    help='with --autograd: capture the whole iteration (loss, backward, Adam) into one hipGraph and replay it')
    ap.add_argument('--timer-every', type=int, default=8,
                    help='HIP-event timing of every N-th launch of the hot kernels inside the timed region (0 = off)')
    ap.add_argument('--bwd-layout', default='runs', choices=['runs', 'slots'], help='backward block-table layout (ablation)')
    ap.add_argument('--fwd-generic', action='store_true', help='run-time slot loop in the forward kernel (ablation)')
    ap.add_argument('--no-block-tables', action='store_true', help='gather from global memory instead of LDS (ablation)')
    ap.add_argument('--no-basis', action='store_true',
                    help='general path: dc_points_fwd every evaluation instead of the basis form x = X0 + (sum w_k c_k) u (ablation)')
    ap.add_argument('--dc-adam', action='store_true', help='with --autograd: depth_correction_amd.optim.Adam (what train() uses) instead of torch.optim.Adam')
    ap.add_argument('--device-warmup-ms', type=float, default=40.0,
                    help='untimed evaluations of the same sequence before the W warm-up steps, for about this long: after an idle or '
                         'host-bound phase (the set-up) the GPU needs ~30 ms of sustained load to reach its clocks (0 = off)')
    ap.add_argument('--no-chain', action='store_true',
                    help='native loop: the ordinary two launches per step (evaluation, reduction + Adam) instead of chained steps '
                         '(one launch per step: each launch also finishes the previous step; dc_sequence_step_chained)')
    ap.add_argument('--two-pass', action='store_true', help='basis form with separate forward and backward kernels (ablation)')
    ap.add_argument('--no-extras', action='store_true', help='skip the C1 / online-correction side measurements')
    ap.add_argument('--multi-sequences', type=int, default=4, help='extras: sequences of the same shape stepped together on one GPU (0 / 1: skip)')
    ap.add_argument('--step-var', type=int, default=None, help='dc_set_option(6, v): form of the one-pass step kernel (A-B measurements)')
    return ap.parse_args()


def algorithmic_bytes(k, active=1.0):
    """SURVEY 8d accounting (fp32 data, int32 indices), bytes per point and launch.  `active` = fraction of the points
    that are neighbourhood centres (1 unless --active-only drops the masked-out centres and their edges)."""
    fwd = 32 + active * (40 + 16 * k)            # raw point inputs + per-centre (outputs, indices, gathers)
    bwd = 72 + active * 28 * k                   # per-point epilogue / saved tensors + per-edge gather and scatter
    return dict(points_fwd=32 + 12, consistency_fwd=fwd - 32 + active * 12, consistency_bwd=bwd, path=fwd + bwd)


def nbytes(*tensors):
    return int(sum(t.numel() * t.element_size() for t in tensors if t is not None))


def compulsory_bytes(plan):
    """Bytes every hot kernel has to move at least once per launch: the arrays it reads and writes, each counted once
    (what the Morton layout + LDS staging reduce the traffic to; rocprofv3's FETCH_SIZE / WRITE_SIZE agree within ~10 %)."""
    ps, ft = plan.ps, plan.fwd_table
    ftl = getattr(plan, 'fwd_table_loss', None) or ft          # what the one-pass step stages from
    built = plan._csr is not None            # the backward structures exist only if an evaluation needed them (lazy)
    bt = plan._bwd_table if built else None
    pt_in = nbytes(ps.vps, ps.dirs, ps.depth, ps.inc, ps.lmask, ps.scan_id)
    fwd_tab = nbytes(ft.blk_ptr, ft.blk_ids, ft.slot_ptr, ft.loc) if ft is not None else nbytes(plan.nbr)
    bwd_tab = nbytes(bt.blk_ptr, bt.blk_ids, bt.slot_ptr, bt.run_ptr, bt.loc) if bt is not None else (nbytes(*plan._csr) if built else 0)
    basis = plan._basis[1] if getattr(plan, '_basis', None) else None
    if basis is not None:
        # basis form: the kernels form the points from the [N, 6 + P] basis rows, no pass over the raw inputs;
        # consistency_step: the one-pass loss + dL/dw kernel (no record written, no backward launch)
        return dict(points_fwd=0, consistency_fwd=nbytes(basis, plan.mask, plan.rec, ft.own_base) + fwd_tab,
                    consistency_bwd=nbytes(basis, plan.rec) + bwd_tab,
                    consistency_step=nbytes(basis, plan.mask, ftl.own_base, ftl.blk_ptr, ftl.blk_ids, ftl.slot_ptr, ftl.loc))
    return dict(points_fwd=pt_in + nbytes(plan.x),
                consistency_fwd=nbytes(plan.x, plan.mask, plan.rec) + fwd_tab,
                consistency_bwd=nbytes(plan.x, plan.rec) + pt_in + bwd_tab)


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box hands
    one GPU's job a share of the host, oversubscribing all visible cores only slows the baseline down)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get('DC_CPU_THREADS')
    return int(env) if env else min(n, 32)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(scans_xyz, poses, k, n_iters, lr, variants):
    """The oracle (reference algorithm as written: materialised [N,K,3,3] products, torch.linalg.eigh, autograd
    backward, torch.optim.Adam) on the host cores, on the C2 workload: fp64 (the reference's default float_type) with
    all cores = the reported value; optionally fp32 and fp64 with 8 threads."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import dc_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    scans = []
    for xyz in scans_xyz:                                   # local_feature_cloud on the CPU (set-up, untimed)
        pts = torch.as_tensor(xyz, dtype=torch.float64)
        depth = pts.norm(dim=-1, keepdim=True)
        dirs = pts / depth
        _, ind = O.knn_ckdtree(pts.numpy(), k)
        f = O.features(pts, torch.as_tensor(ind), dirs)
        scans.append(dict(vps=torch.zeros_like(pts), dirs=dirs, depth=depth, inc=f['inc_angles'],
                          mask=O.local_mask(f['eigvals'], None, [[0, 1, 0.0, 0.25], [1, 2, 0.25, 1.0]])))
    poses = torch.as_tensor(poses, dtype=torch.float64)
    x0 = torch.cat([O.points_from(*O.transform_cloud(s['vps'], s['dirs'], T), s['depth']) for s, T in zip(scans, poses)])
    _, nbr = O.knn_ckdtree(x0.numpy(), k)
    nbr = torch.as_tensor(nbr)
    f0 = O.features(x0, nbr, torch.cat([s['dirs'] for s in scans]))
    mask = O.global_mask(torch.cat([s['mask'] for s in scans]), nbr, f0['eigvals'], min_valid_neighbors=5,
                         eigenvalue_ratio_bounds=[[0, 1, 0.0, 0.25], [1, 2, 0.25, 1.0]])
    del f0
    n = len(x0)

    def run(dtype, threads, iters):
        torch.set_num_threads(threads)
        sc = [{key: (v.to(dtype) if v.dtype.is_floating_point else v) for key, v in s.items()} for s in scans]
        w = torch.nn.Parameter(torch.tensor([[1e-3, 2e-3]], dtype=dtype))
        e = torch.tensor([[2.0, 4.0]], dtype=dtype)
        opt = torch.optim.Adam([w], lr=lr)
        times = []
        for it in range(iters + 1):
            t0 = time.perf_counter()
            opt.zero_grad()
            loss, _ = O.eval_sequence(sc, poses.to(dtype), w, e, nbr, mask, reduction='mean')
            loss.backward()
            opt.step()
            times.append(time.perf_counter() - t0)
        med = float(np.median(times[1:]))
        return dict(points_per_s=n / med, s_per_iter=med, threads=threads, iters=iters, loss=float(loss.detach()))

    main = run(torch.float64, cores, n_iters)
    out = dict(value=main['points_per_s'], unit='points/s', cores=cores, kind='port', cpu=cpu_model(),
               sample='%d scans x %d pts (N=%d), k=%d, fp64, all %d usable cores, 1 warm-up + %d iterations, median; loss %.6g'
                      % (len(scans), len(scans_xyz[0]), n, k, cores, n_iters, main['loss']),
               s_per_iter=main['s_per_iter'])
    if variants:
        out['fp32_all_cores'] = run(torch.float32, cores, 2)
        out['fp64_8_threads'] = run(torch.float64, min(8, cores), 2)
    torch.set_num_threads(cores)
    return out


def launch_ranks(n):
    """Start `n` ranks of this script (one per GPU, RCCL over xGMI) as a child process and wait for them; returns the
    children's exit code.  The parent stays off the GPU for its whole life."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def load_profile_table():
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def profile_key(kernel, n):
    """Key of profiles/traffic.json: the instantiation as rocprofv3 prints it (the library reports a named template constant by
    its name: kStepVar = 7, csrc/dc_consistency.hip) / problem size."""
    return '%s/N%d' % (kernel.replace('kStepVar', '7'), n)


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python3 bench.py --gpus N` on its own: this process has not touched the GPU (importing torch does not), so it
        # may start the N ranks as a CHILD torch.distributed.run (never exec) and hand back their exit code; rank 0 of the
        # children prints the one JSON line on the inherited stdout
        sys.exit(launch_ranks(args.gpus))
    if args.gpus != world:
        raise SystemExit('bench.py --gpus %d was started with WORLD_SIZE=%d: the two must agree' % (args.gpus, world))
    n_visible = torch.cuda.device_count()                 # counting devices does not initialise the GPU
    if local_rank >= n_visible:
        raise SystemExit('bench.py rank %d of %d: needs GPU index %d but this node shows %d GPU(s); --gpus must not exceed '
                         'the GPUs of the node' % (rank, world, local_rank, n_visible))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or os.environ.get('DC_FORCE_DIST') == '1':      # DC_FORCE_DIST: exercise the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        # RCCL prints a version banner to stdout when the communicator comes up; the contract is ONE JSON line there, so file
        # descriptor 1 points at stderr while it does (communicator creation is forced here by a first collective)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', device_id=dev)
            dist.all_reduce(torch.zeros((1,), device=dev))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    if args.workload == 'c4':
        return main_c4(args, world, rank, local_rank, dev, dist)
    from depth_correction_amd.dataset import RoomBoxDataset
    from depth_correction_amd.pipeline import build_sequence
    from depth_correction_amd.plan import consistency_loss
    from depth_correction_amd import ops

    dtype = getattr(torch, args.dtype)
    ds = RoomBoxDataset(n_pts=args.points, n_poses=args.scans, seed_base=1000 + 100 * rank,
                        dtype=np.float32 if dtype == torch.float32 else np.float64)
    scans_xyz, poses = [], []
    for cloud, pose in ds:
        scans_xyz.append(np.stack([cloud[f] for f in 'xyz'], axis=1))
        poses.append(pose)
    poses = np.stack(poses)

    # ---- set-up phase (train.py:94-215): first call in the process (loads the code objects, warms the allocator), then
    # the same build again with the device synchronised between stages = the steady-state cost and its breakdown
    build = lambda **kw: build_sequence(scans_xyz, poses, k=args.k, dtype=dtype, device=dev, spatial_sort=not args.no_sort,
                                        point_format=args.point_format, active_only=args.active_only,
                                        degree_sort=args.degree_sort, block_tables=not args.no_block_tables,
                                        bwd_layout=args.bwd_layout, basis=not args.no_basis, **kw)
    # (apart from it: the device context with torch's first kernel, and dlopen of the library -- what any GPU program pays)
    t0 = time.perf_counter()
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    device_init_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    from depth_correction_amd import _native as _nv0
    _nv0.lib()
    library_load_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    plan, info = build()
    torch.cuda.synchronize()
    setup_first_s = time.perf_counter() - t0
    del plan, info
    t0 = time.perf_counter()
    plan, info = build(stage_times=True)
    torch.cuda.synchronize()
    setup_ms = (time.perf_counter() - t0) * 1e3
    from depth_correction_amd import _native as nv
    if args.fwd_generic:
        nv.check(nv.lib().dc_set_option(1, 1), 'dc_set_option')
    if args.two_pass:
        nv.check(nv.lib().dc_set_option(4, 1), 'dc_set_option')
    if args.step_var is not None:
        nv.check(nv.lib().dc_set_option(6, args.step_var), 'dc_set_option')
    # the k-NN build alone, on the global cloud (reported separately)
    # (one untimed call first: the variant without distances is not the one the set-up ran, and a code object's first launch pages
    # it in -- 11 ms in the round-4 driver line against 1.7 ms for the kernels; then the median of five)
    ops.knn(info['points0'], args.k, want_dist=False)
    torch.cuda.synchronize()
    knn_all = []
    for _ in range(5):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        ops.knn(info['points0'], args.k, want_dist=False)
        ev1.record()
        torch.cuda.synchronize()
        knn_all.append(ev0.elapsed_time(ev1))
    knn_ms = float(np.median(knn_all))

    n_local = plan.n
    from depth_correction_amd.plan import SequenceTrainer, KernelTimer
    w0, e0 = [1e-3, 2e-3], [2.0, 4.0]
    poses_t = info['poses']
    if args.autograd:
        # drop-in style loop: torch autograd Function + torch.optim.Adam (what train.py drives)
        count = torch.tensor([plan.count], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(count)
        total_count = float(count.item())
        w = torch.nn.Parameter(torch.tensor([w0], dtype=torch.float64, device=dev))
        exponent = torch.tensor([e0], dtype=torch.float64, device=dev)
        if args.dc_adam:
            from depth_correction_amd.optim import Adam as DcAdam
            opt = DcAdam([w], lr=1e-3)                      # torch.optim.Adam's update as one launch (capturable as it is)
        else:
            opt = torch.optim.Adam([w], lr=1e-3, capturable=bool(args.graph))
        packed = torch.zeros((1 + w.numel(),), dtype=torch.float64, device=dev)

        def step():
            opt.zero_grad(set_to_none=False)
            s, _ = consistency_loss(plan, w, exponent, poses_t)
            loss = s / total_count
            loss.backward()
            if dist is not None:
                packed[0] = loss.detach()
                packed[1:] = w.grad.reshape(-1)
                dist.all_reduce(packed)
                w.grad.copy_(packed[1:].reshape(w.shape))
                loss = packed[0]
            opt.step()
            return loss.detach()

        if args.graph:
            # the kernels are launched on torch's current stream through the C ABI, so torch.cuda.graph captures them like
            # any torch op: one hipGraph launch per iteration, no Python and no per-kernel launches on the host
            assert dist is None, '--graph is a single-GPU mode'
            eager_step = step
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = eager_step()

            def step():
                graph.replay()
                return static_loss
    else:
        # native loop: one host call per evaluation (dc_sequence_eval) + dc_adam_step; with several ranks the only
        # exchange of the path is one RCCL all-reduce of [sum loss, count, dL/dw] per step
        trainer = SequenceTrainer([plan], w0, e0, [poses_t], lr=1e-3, distributed=dist is not None,
                                  chained=not args.no_chain)
        total_count = trainer.count

        def step():
            acc = trainer.step()
            return acc

    # ---- device warm-up (not optimisation steps: evaluations that leave the weights alone).  From an idle GPU the same step
    # takes 85 us, 77 us after 10 ms and 69 us from 30 ms on (clock ramp); the set-up phase is host-bound, so the
    # timed region would otherwise start on a half-idle device.  The per-window times are reported (config.clock_ramp).
    ramp = None
    if args.device_warmup_ms > 0:
        wv, ev_ = torch.tensor(w0, dtype=torch.float64, device=dev), torch.tensor(e0, dtype=torch.float64, device=dev)
        P12 = plan.poses12(poses_t)
        scratch_out = torch.zeros((2 + 2 * len(w0) + 12 * plan.n_scans,), dtype=torch.float64, device=dev)
        marks = [torch.cuda.Event(enable_timing=True)]
        marks[0].record()
        t_start = time.perf_counter()
        while (time.perf_counter() - t_start) * 1e3 < args.device_warmup_ms and len(marks) < 200:
            for _ in range(25):
                plan.eval_native(wv, ev_, P12, scratch_out)
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
            marks[-1].synchronize() if len(marks) % 8 == 0 else None      # the host must not run far ahead of the clock it watches
        torch.cuda.synchronize()
        ramp = [marks[i].elapsed_time(marks[i + 1]) / 25 * 1e3 for i in range(len(marks) - 1)]
    for _ in range(args.warmup):
        loss = step()
    if not args.autograd:
        trainer.flush()                             # chained steps: the warm-up's last step is finished before the clock starts
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    with KernelTimer(every=args.timer_every) as timer:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        if not args.autograd:
            loss = trainer.flush()                  # chained steps: the last step's sums and Adam update (no-op otherwise)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = timer.read()
        kernel_names = timer.kernels()
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    loss = loss.detach().cpu()
    final_loss = float(loss) if loss.numel() == 1 else float(loss[0] / loss[1])
    rccl = None
    if dist is not None:
        # the path's only exchange, measured by itself right after the timed region (all ranks take part): the packed fp64
        # vector [sum loss, count, dL/dw] every step all-reduces, 100 back-to-back calls between two events
        buf = torch.zeros((2 + len(w0),), dtype=torch.float64, device=dev)
        for _ in range(10):
            dist.all_reduce(buf)
        ar0, ar1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ar0.record()
        for _ in range(100):
            dist.all_reduce(buf)
        ar1.record()
        torch.cuda.synchronize()
        rccl = {'rccl_ranks': dist.get_world_size(), 'backend': dist.get_backend(),
                'allreduce_us_per_step': ar0.elapsed_time(ar1) * 10.0, 'allreduce_bytes': buf.numel() * 8,
                'what': 'one all-reduce (sum) of [sum loss, count, dL/dw] per step; stand-alone time of that call, stream time'}

    # ---- side measurements, after the timed region (the GPU goes idle between their host-synchronised calls; run before
    # the loop they left it at idle clocks for the first timed steps)
    extras = {}
    sustained_kernel_ms = sustained_kernel_names = None
    if world == 1 and not args.autograd and not args.no_extras:
        # the same chained step (a) sustained -- 2 000 steps back to back, what a training run of the reference's default length
        # and longer sees -- and (b) cold -- the first `steps` steps after the device has sat idle for a second (clocks down, no
        # device warm-up: what `--device-warmup-ms 0` times).  Their own trainer: the timed run's state is left alone.
        tr2 = SequenceTrainer([plan], w0, e0, [poses_t], lr=1e-3, chained=not args.no_chain)
        for _ in range(100):
            tr2.step()
        tr2.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            tr2.step()
        tr2.flush()
        torch.cuda.synchronize()
        extras['sustained_ms_per_step'] = (time.perf_counter() - t0) / 2000 * 1e3
        extras['sustained_steps'] = 2000
        # the kernel's own duration over the same 2 000 steps (a second pass: stamping every 8th launch costs a few microseconds of
        # idle device each, so it is kept out of the sustained figure): 250 stamps, what the roofline block is computed from
        with KernelTimer(every=8) as kt2:
            for _ in range(2000):
                tr2.step()
            tr2.flush()
            torch.cuda.synchronize()
            sustained_kernel_ms = kt2.read()
            sustained_kernel_names = kt2.kernels()
        time.sleep(1.0)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr2.step()
        tr2.flush()
        torch.cuda.synchronize()
        extras['cold_ms_per_step'] = (time.perf_counter() - t0) / args.steps * 1e3
        extras['cold_note'] = '%d chained steps + flush started after one second of an idle device, no device warm-up' % args.steps
        del tr2
    if world == 1 and not args.autograd and not args.no_extras and args.dtype == 'float32' and args.scans * args.points <= 4_000_000:
        # the reference's DEFAULT float_type is float64 (config.py:179): the same sequence with float64 clouds (fp64 points and
        # basis rows, consistency_step_basis_kernel<double, ...>), 200 chained steps
        scans64 = [s_.astype(np.float64) for s_ in scans_xyz]
        plan64, info64 = build_sequence(scans64, poses, k=args.k, dtype=torch.float64, device=dev)
        tr64 = SequenceTrainer([plan64], w0, e0, [info64['poses']], lr=1e-3, chained=not args.no_chain)
        for _ in range(1000):          # (~60 ms: the clocks have to come back up after the idle second of the cold measurement above)
            tr64.step()
        tr64.flush()
        torch.cuda.synchronize()
        with KernelTimer(every=8) as kt64:
            t0 = time.perf_counter()
            for _ in range(200):
                tr64.step()
            tr64.flush()
            torch.cuda.synchronize()
            ms64 = (time.perf_counter() - t0) / 200 * 1e3
            k64 = kt64.kernels().get('consistency_fwd')
            kms64 = kt64.read().get('consistency_fwd', (None, 0))[0]
        prof64 = load_profile_table().get(profile_key(k64, plan64.n))
        extras['fp64_cloud_step'] = {'ms_per_step': ms64, 'kernel': k64, 'kernel_ms': kms64, 'points_per_s': plan64.n / (ms64 * 1e-3),
                                     'hbm_bytes': prof64['hbm_bytes'] if prof64 else None,
                                     'hbm_frac': (prof64['hbm_bytes'] / (kms64 * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (prof64 and kms64) else None,
                                     'valu_insts_per_point': prof64['valu_insts_per_point'] if prof64 else None,
                                     'what': 'float64 clouds (the reference default float_type): fp64 points and first sweep, 48-byte staged rows {x fp64 | u, c float32} for the second'}
        del tr64, plan64, info64, scans64
        torch.cuda.empty_cache()
    if world == 1 and not args.autograd and not args.no_extras and args.multi_sequences > 1 and args.scans * args.points <= 4_000_000:
        # (d) S sequences of the same shape on ONE GPU, stepped together as train() does for several train_names (train.py:172-175;
        # eval.py:85-112 pools their sums): S x 118 MB of rows and tables no longer fit the 256 MiB Infinity Cache, so every launch
        # fetches its working set from HBM -- the number the single-sequence line cannot give
        S = args.multi_sequences
        mplans, mposes = [plan], [poses_t]
        for q in range(1, S):
            ds_q = RoomBoxDataset(n_pts=args.points, n_poses=args.scans, seed_base=1000 + 7919 * q,
                                  dtype=np.float32 if dtype == torch.float32 else np.float64)
            xyz_q = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds_q]
            pl_q, info_q = build_sequence(xyz_q, np.stack([p for _, p in ds_q]), k=args.k, dtype=dtype, device=dev)
            mplans.append(pl_q)
            mposes.append(info_q['poses'])
            del ds_q, xyz_q, info_q
        trm = SequenceTrainer(mplans, w0, e0, mposes, lr=1e-3, chained=not args.no_chain)
        for _ in range(300):           # (~57 ms: the set-up of the extra sequences above is host work, the clocks have to come back up)
            trm.step()
        trm.flush()
        torch.cuda.synchronize()
        n_ms = 300
        with KernelTimer(every=8) as ktm:
            t0 = time.perf_counter()
            for _ in range(n_ms):
                trm.step()
            trm.flush()
            torch.cuda.synchronize()
            ms_multi = (time.perf_counter() - t0) / n_ms * 1e3
            km = ktm.read().get('consistency_fwd', (None, 0))
            kname_m = ktm.kernels().get('consistency_fwd')
        prof_m = load_profile_table().get(profile_key(kname_m, mplans[0].n) + '/multi%d' % S)
        comp_m = compulsory_bytes(mplans[0])['consistency_step']
        extras['multi_sequence_step'] = {
            'sequences': S, 'points': int(sum(p_.n for p_ in mplans)), 'working_set_bytes': int(sum(compulsory_bytes(p_)['consistency_step'] for p_ in mplans)),
            'ms_per_step': ms_multi, 'us_per_sequence_step': ms_multi / S * 1e3, 'points_per_s': sum(p_.n for p_ in mplans) / (ms_multi * 1e-3),
            'kernel': kname_m, 'kernel_ms': km[0], 'timed_launches': km[1],
            'launches_per_step': ('%d: one per sequence (a chain over the sequences, dc_sequence_step_linked: every launch finishes the one before it)' % S)
                                 if getattr(trm, 'linked', False) else
                                 ('%d evaluations (the first takes the previous Adam update in its launch) + %d reductions + the sum of the sequences\' sums' % (S, S)),
            'design_bytes_per_launch': comp_m,
            'design_GBps': comp_m / (km[0] * 1e-3) / 1e9 if km[0] else None,
            'traffic_per_launch': prof_m['hbm_bytes'] if prof_m else None,
            'hbm_frac': (prof_m['hbm_bytes'] / (km[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (prof_m and km[0]) else None,
            'traffic_source': prof_m.get('source') if prof_m else 'no committed PMC pass of this configuration',
            'what': '%d sequences of %d x %dk points stepped round robin on one GPU (a loss over several training sequences); the working '
                    'set exceeds the 256 MiB Infinity Cache, unlike the single-sequence step' % (S, args.scans, args.points // 1000)}
        del trm, mplans, mposes
        torch.cuda.empty_cache()
    if world == 1 and not args.no_extras:        # single-process runs only: the other ranks must not wait for rank 0
        # the same step without the loop-invariant hoisting (general path: dc_points_fwd + forward + backward every iteration,
        # what the reference's own loop recomputes) and with separate forward / backward kernels, for comparison
        if not args.autograd and not args.no_basis and not args.two_pass:
            abl = {}
            for name, opt in (('two_pass_basis_form', 4), ('general_path', 3)):
                nv.check(nv.lib().dc_set_option(opt, 1), 'dc_set_option')
                try:
                    tr = SequenceTrainer([plan], w0, e0, [poses_t], lr=1e-3)
                    for _ in range(300):                  # builds the backward tables on first use, then keeps the clocks up
                        tr.step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(50):
                        tr.step()
                    torch.cuda.synchronize()
                    abl[name + '_ms_per_step'] = (time.perf_counter() - t0) / 50 * 1e3
                finally:
                    nv.check(nv.lib().dc_set_option(opt, 0), 'dc_set_option')
            extras['same_step_other_forms'] = abl
            # pose mode (train() with pose corrections: train.py:300-312, eval.py:68-82): loss, dL/dw and dL/d[R|t] of every scan per
            # evaluation -- one launch of consistency_step_pose_kernel + the reduction; and the same through the general path
            wv, ev_ = torch.tensor(w0, dtype=torch.float64, device=dev), torch.tensor(e0, dtype=torch.float64, device=dev)
            pose_out = torch.zeros((2 + 2 * len(w0) + 12 * plan.n_scans,), dtype=torch.float64, device=dev)
            P12 = plan.poses12(poses_t)
            pose = {}
            for name, three in (('one_launch', 0), ('three_kernel_path', 1)):
                nv.check(nv.lib().dc_set_option(7, three), 'dc_set_option')
                try:
                    for _ in range(100):
                        plan.eval_native(wv, ev_, P12, pose_out, want_grad=True, want_pose=True)
                    torch.cuda.synchronize()
                    with KernelTimer(every=8) as ktp:
                        t0 = time.perf_counter()
                        for _ in range(200):
                            plan.eval_native(wv, ev_, P12, pose_out, want_grad=True, want_pose=True)
                        torch.cuda.synchronize()
                        pose[name + '_ms'] = (time.perf_counter() - t0) / 200 * 1e3
                        if not three:
                            pose['kernel'] = ktp.kernels().get('consistency_fwd')
                            pose['kernel_ms'] = ktp.read().get('consistency_fwd', (None, 0))[0]
                finally:
                    nv.check(nv.lib().dc_set_option(7, 0), 'dc_set_option')
            extras['pose_mode_evaluation'] = pose

        # the reference-API loop itself: train() (train.py:220-322) with its default callbacks on this very workload, wall clock
        # per iteration from two runs of different length (set-up cancels); see tools/train_bench.py for the C4 shape
        if args.scans * args.points <= 4_000_000:
            import contextlib
            import io
            import tempfile
            from depth_correction_amd.config import Config as _Cfg
            from depth_correction_amd.train import train as _train
            tcfg = _Cfg(nn_k=args.k, nn_r=None, min_depth=0.0, max_depth=float('inf'), grid_res=0.0, vp_dispersion_bounds=[], lr=1e-3,
                        float_type=args.dtype, device=str(dev), model_kwargs={'w': w0, 'exponent': e0})
            seq = [(c, p) for c, p in ds]

            def train_ms(cfg_):
                # (2 000 iterations apart and the shorter of two runs each: the loop takes 0.05-0.1 ms per iteration, the set-up
                #  in front of it varies by milliseconds from call to call)
                wall = {40: [], 2040: []}
                for n_it in (40, 40, 2040, 40, 2040):
                    c = cfg_.copy()
                    c.n_opt_iters, c.log_dir = n_it, tempfile.mkdtemp()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    with contextlib.redirect_stdout(io.StringIO()):
                        _train(c, train_datasets=[seq], val_datasets=[])
                    torch.cuda.synchronize()
                    wall[n_it].append(time.perf_counter() - t0)
                return (min(wall[2040]) - min(wall[40][1:])) / 2000 * 1e3      # (the first run pays the process's one-time costs)
            extras['train_iteration_ms'] = train_ms(tcfg)
            extras['train_iteration_note'] = ('depth_correction_amd.train.train() itself, default callbacks, cfg.loop_batch = %d: model-only runs '
                                              'go to the chained native step, one launch per iteration; the bookkeeping of a batch runs '
                                              'while the device works on the next' % tcfg.loop_batch)
            # the same with per-pose corrections (scripts/model_poses_learning:71): model weights and nine poses optimised
            from depth_correction_amd.config import PoseCorrection as _PC
            pcfg = tcfg.copy()
            pcfg.pose_correction = _PC.pose
            extras['train_pose_iteration_ms'] = train_ms(pcfg)
            extras['train_pose_iteration_note'] = ('train() with PoseCorrection.pose: per iteration the pose kernel, its reduction and one '
                                                   'finishing launch (dc_pose_train_finish), replayed as a graph')

        # BASELINE config 1 (one 200k-point scan, nn_k = 10, covariance + eig forward only, all DepthCloud features written):
        # 220 launches, each timed by the library's dispatch stamps (the host call takes longer than the kernel, so stream events
        # around a loop of calls would time the host)
        c0 = info['clouds'][0]
        x1, n1 = c0['points'], c0['points'].shape[0]
        for _ in range(20):
            ops.features_fwd(x1, c0['neighbors'], dirs=c0['dirs'])
        with KernelTimer(every=1) as ft:
            for _ in range(220):
                ops.features_fwd(x1, c0['neighbors'], dirs=c0['dirs'])
            torch.cuda.synchronize()
            c1_ms, c1_launches = ft.read()['features_fwd']
            c1_kernel = ft.kernels()['features_fwd']
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        evs[0].record()
        ops.knn(x1, args.k, want_dist=False)
        evs[1].record()
        torch.cuda.synchronize()
        c1 = {'points': n1, 'features_fwd_ms': c1_ms, 'timed_launches': c1_launches, 'kernel': c1_kernel,
              'points_per_s': n1 / (c1_ms * 1e-3), 'algorithmic_GBps': 284 * n1 / (c1_ms * 1e-3) / 1e9,
              'knn_build_ms': evs[0].elapsed_time(evs[1])}
        prof = (load_profile_table().get('_paths') or {}).get('kernels', {}).get('%s @ c1' % c1_kernel.replace('dc::', '')) \
            if n1 == 200_000 and args.k == 10 else None
        comp1 = n1 * (4 * args.k + (12 + 12) + (12 + 36 + 12 + 36 + 12 + 4))           # index rows, centre + direction, six outputs
        traffic = prof['hbm_bytes'] if prof and 'hbm_bytes' in prof else None
        insts = prof.get('valu_insts_per_wave') if prof else None
        waves = (n1 + 63) // 64
        t_s = c1_ms * 1e-3
        hbm_frac = (traffic or comp1) / t_s / 1e9 / HBM_PEAK_GBPS
        valu_frac = None if insts is None else insts * waves / (t_s * N_SIMD * CLOCK_GHZ * 1e9 / 4)
        # the gathers: N K rows of 12 B, each its own cache line when the scan's points are in no spatial order (these rays are
        # drawn at random) -- the kernel is bound by the CUs' vector-memory address pipelines (one line look-up per lane and
        # gather), neither by bytes nor by instructions: tools/ubench/gather_rows.hip measures that floor by itself
        c1['roofline'] = {'bound': 'hbm', 'limiter': 'L1 line look-ups of the random row gathers (neither roofline: see note)',
                          'achieved': (traffic or comp1) / t_s / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': hbm_frac,
                          'traffic': traffic, 'compulsory_bytes': comp1,
                          'traffic_source': (load_profile_table().get('_paths') or {}).get('source') if prof else
                          'no committed PMC profile of this instantiation: achieved uses compulsory_bytes',
                          'valu': {'insts_per_wave': insts, 'frac_of_issue_peak': valu_frac},
                          'gather_lookups_per_launch': n1 * (args.k - 1),
                          'note': 'SURVEY 8d prices C1 at 284 B/pt (algorithmic_GBps above); measured bytes and instructions put '
                                  'the kernel far below both rooflines because its %d x (K - 1) = %d random 12-B gathers from an L2-resident '
                                  '2.4 MB table cost ~7 us by themselves on this chip whatever the cache policy '
                                  '(profiles/r04_ubench_gather_rows.txt); the same kernel on the same points in Morton order: '
                                  'tools/features_bench.py' % (n1, n1 * (args.k - 1))}
        # the same kernel on the same scan with its points in Morton order (a scan in sensor order -- ring by ring -- is spatially
        # coherent like that; this generator draws its rays at random): neighbours share cache lines, the gathers cost a third
        order1 = ops.spatial_order(x1).long()
        xs1, ds1 = x1[order1].contiguous(), c0['dirs'][order1].contiguous()
        _, idx_s1 = ops.knn(xs1, args.k)
        for _ in range(20):
            ops.features_fwd(xs1, idx_s1, dirs=ds1)
        with KernelTimer(every=1) as ft:
            for _ in range(220):
                ops.features_fwd(xs1, idx_s1, dirs=ds1)
            torch.cuda.synchronize()
            c1['features_fwd_ms_points_in_morton_order'] = ft.read()['features_fwd'][0]
        del xs1, ds1, idx_s1, order1
        extras['c1_forward_only'] = c1
        # the online correction node's per-scan work (scripts/depth_correction:31-58): local_feature_cloud (shadow filter,
        # neighbourhoods, features, mask) -> model -> update_points, on an already uploaded 200k-point scan
        from depth_correction_amd.config import Config
        from depth_correction_amd.model import ScaledPolynomial
        from depth_correction_amd.online import correct_cloud
        from depth_correction_amd.scan_io import cloud_on_device
        cfg = Config(nn_k=args.k, nn_r=None, device=str(dev), float_type=args.dtype, shadow_neighborhood_angle=0.017453,
                     shadow_angle_bounds=[float(np.radians(5.0)), float('inf')], log_filters=False)
        model = ScaledPolynomial(w=[1e-3, 2e-3], exponent=[2.0, 4.0], device=dev)
        raw = torch.as_tensor(scans_xyz[0], device=dev)
        lat = []
        for it in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out_cloud = correct_cloud(raw, model, cfg)          # (the uploaded rows: dc_scan_prefilter, then neighbourhoods, features, model)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        extras['online_correction'] = {'points_in': int(raw.shape[0]), 'points_out': len(out_cloud),
                                       'latency_ms': float(np.median(lat[1:])), 'first_call_ms': lat[0],
                                       'what': 'correct_cloud on the uploaded rows: from_points + shadow filter + cloud[mask] (dc_scan_prefilter), k-NN, '
                                               'features, mask, model, points; scan resident on the device; median of 5'}

    if rank == 0:
        timed_region_kernel_ms = {name: v[0] for name, v in kernel_ms.items()}
        stamps_from = 'the timed region (%d steps)' % args.steps
        if sustained_kernel_ms and 'consistency_fwd' in sustained_kernel_ms and sustained_kernel_ms['consistency_fwd'][1] >= 100:
            # (three stamps of a 20-step run are not a measurement; the roofline block uses the sustained run's)
            kernel_ms, kernel_names = sustained_kernel_ms, sustained_kernel_names
            stamps_from = 'the sustained run (2 000 chained steps after the timed region)'
        ms = {name: v[0] for name, v in kernel_ms.items()}
        ab = algorithmic_bytes(args.k, (plan.count / n_local) if args.active_only else 1.0)
        value = n_local * world * args.steps / elapsed
        roofline = None
        if 'consistency_fwd' in ms:
            table = load_profile_table()
            comp = compulsory_bytes(plan)
            # the one-pass kernel (loss + dL/dw) is launched and timed in the forward's place; there is no backward launch then
            one_pass = 'consistency_step' in kernel_names.get('consistency_fwd', '')
            if one_pass:
                comp['consistency_fwd'] = comp['consistency_step']
                ab['consistency_fwd'] = ab['path']
            per_kernel = {}
            for name in ('points_fwd', 'consistency_fwd', 'consistency_bwd'):
                if name not in ms:
                    continue
                prof = table.get(profile_key(kernel_names.get(name, ''), n_local))
                t_s = ms[name] * 1e-3
                traffic = prof['hbm_bytes'] if prof else None
                insts = prof['valu_insts_per_point'] if prof else None
                per_kernel[name] = {
                    'kernel': kernel_names.get(name), 'ms': ms[name], 'traffic': traffic, 'compulsory_bytes': comp[name],
                    'hbm_GBps': (traffic if traffic else comp[name]) / t_s / 1e9,
                    'hbm_frac': (traffic if traffic else comp[name]) / t_s / 1e9 / HBM_PEAK_GBPS,
                    'valu_insts_per_point': insts,
                    # wave-instructions issued / wave-instruction slots of the chip in the kernel's duration
                    'valu_frac': None if insts is None else (insts * n_local / 64) / (t_s * N_SIMD * CLOCK_GHZ * 1e9 / 4),
                    'profile': None if prof is None else prof.get('source'),
                    'algorithmic_GBps': ab[name] * n_local / t_s / 1e9}
            dom = max((n_ for n_ in ('consistency_fwd', 'consistency_bwd') if n_ in ms), key=lambda n_: ms[n_])
            d = per_kernel[dom]
            # `bound` names the roofline achieved / peak / frac are quoted against (bytes over HBM); `limiter` what the counters say
            # actually limits the kernel: the vector ALUs' issue rate (fp64 / int VALU work, not MFMA) or the bytes
            limiter = 'valu' if (d['valu_frac'] or 0.0) > d['hbm_frac'] else 'hbm'
            roofline = {'bound': 'hbm', 'limiter': limiter, 'kernel': 'dc_sequence_step (one-pass loss + dL/dw kernel)' if one_pass else 'dc_' + dom,
                        'instantiation': d['kernel'],
                        'achieved': d['hbm_GBps'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': d['hbm_frac'],
                        'traffic': d['traffic'],
                        'traffic_source': d['profile'] or 'no committed PMC profile of this instantiation: achieved uses compulsory_bytes',
                        'valu': {'insts_per_point': d['valu_insts_per_point'], 'frac_of_issue_peak': d['valu_frac'],
                                 'peak': '%d SIMDs x %.1f GHz / 4 cycles per wave64 instruction' % (N_SIMD, CLOCK_GHZ)},
                        'kernels': per_kernel,
                        'timed_launches': {name: v[1] for name, v in kernel_ms.items()},
                        'timing': 'HIP events stamped with the dispatch start / end of every 8th launch of ' + stamps_from,
                        'gpu_kernel_ms_per_step': sum(ms.values()),
                        'gpu_kernel_ms_per_step_timed_region': sum(timed_region_kernel_ms.values()),
                        # (c) the two byte models side by side
                        'model_bytes_8d': ab['path'] * n_local,
                        'design_bytes': d['compulsory_bytes'],
                        'bytes_note': 'model_bytes_8d = SURVEY 8(d) / BASELINE.md 3 (584 B per point: every gather and scattered gradient an '
                                      'HBM access, saved tensors, a backward pass) x N; over the kernel time it exceeds the HBM peak because '
                                      'this design does not move those bytes: LDS-staged block tables serve the gathers, dL/dw is formed in '
                                      'forward mode (no record, no backward pass), the pose-invariant basis rows are hoisted.  design_bytes = '
                                      'every array the kernel touches, once; frac = measured bytes (traffic) / kernel time / peak',
                        'working_set_fits_mall': bool(d['compulsory_bytes'] < 256 * 2 ** 20),
                        'mall_note': 'the step re-reads the same ~%d MB every launch and the Infinity Cache holds 256 MiB: the counter bytes '
                                     '(fabric requests, MALL hits included) are not HBM traffic and 8 TB/s is not the ceiling that applies; '
                                     'config.multi_sequence_step is the same kernel on a working set that does not fit'
                                     % (d['compulsory_bytes'] // 2 ** 20),
                        'algorithmic': {'bytes_per_point': ab[dom], 'GBps': d['algorithmic_GBps'],
                                        'path_bytes_per_point': ab['path'], 'path_GBps': ab['path'] * value / world / 1e9,
                                        'note': 'SURVEY 8d accounting: every gather / scatter priced as a 12-B HBM access; the '
                                                'Morton layout and LDS-staged block tables serve them on chip, so this figure may '
                                                'exceed the HBM peak and is NOT a roofline fraction'}}
        out = {
            'metric': 'points/sec through min_eigval_loss fwd+bwd (200k pts, k=10); HBM GB/s vs peak',
            'value': value, 'unit': 'points/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'C2: %d overlapping %dk-pt room-box scans per sequence (N=%d per GPU), nn_k=%d, '
                                   'ScaledPolynomial, min_eigval_loss(normalization) fwd+bwd + Adam; one sequence per GPU'
                                   % (args.scans, args.points // 1000, n_local, args.k),
                       'storage': args.dtype + ('+q32 points' if plan.qfmt is not None else ''),
                       'arithmetic': 'fp64 on chip (moments, eigen-solve, loss, accumulators); float32 second sweep (dL/dw terms from exact int32 differences and float32 u, c)',
                       'form': 'basis (x = X0 + (sum_k w_k c_k) u formed inside the kernel; loss and dL/dw in one pass over each centre\'s neighbours; the basis rows are rebuilt only when poses or exponents change)'
                               if getattr(plan, '_basis', None) else 'general (dc_points_fwd every evaluation)',
                       'loop': ('autograd+' + ('optim.Adam (dc_adam_step)' if args.dc_adam else 'torch.optim.Adam') + (' replayed as one hipGraph' if args.graph else '')) if args.autograd else ('native, chained: one launch per step (dc_sequence_step_chained), the last one flushed' if trainer.chained else
                                                     ('native: evaluation (+ the previous Adam update in its launch) -> reduction -> all-reduce'
                                                      if trainer.update_in_next else 'native (dc_sequence_step)')),
                       'masked_points': total_count, 'active_only': bool(args.active_only),
                       'skipped_wavefronts_share': getattr(plan, 'skipped_wavefronts', 0.0),
                       'skipped_note': 'share of the 64-centre wavefronts whose centres are ALL outside the loss mask: they stage their rows '
                                       'and skip the moments, the eigen-solve and the second sweep (every term they would add carries the mask); '
                                       'the plan forms wavefronts of one kind (64 Morton-consecutive points inside, or outside, the mask)', 'spatial_sort': not args.no_sort, 'final_loss': final_loss,
                       'knn_build_ms': knn_ms, 'knn_points_per_s': n_local / (knn_ms * 1e-3),
                       'setup_ms': setup_ms, 'setup_stages_ms': info['setup_ms'], 'setup_first_call_s': setup_first_s, 'device_init_s': device_init_s, 'library_load_s': library_load_s,
                       'setup_note': 'setup_ms: the whole set-up phase (upload, 10 local feature clouds, global k-NN, masks, Morton '
                                     'order, transpose, block tables) rebuilt in a warm process; setup_first_call_s additionally '
                                     'holds the one-time code-object load and allocator warm-up'},
            'roofline': roofline,
        }
        if ramp:
            out['config']['device_warmup'] = {
                'evaluations': 25 * len(ramp), 'ms': sum(ramp) * 25 / 1e3,
                'us_per_evaluation_per_window_of_25': [round(v, 1) for v in ramp],
                'what': 'untimed dc_sequence_eval calls (loss + dL/dw, no optimiser step) before the W warm-up steps: the GPU reaches '
                        'its clocks only after ~30 ms of sustained load; --device-warmup-ms 0 turns this off'}
        out['config'].update(extras)
        if rccl:
            out['config'].update(rccl)
            out['rccl_ranks'] = rccl['rccl_ranks']
            out['allreduce_us_per_step'] = rccl['allreduce_us_per_step']
        if world == 1 and args.cpu_scans > 0:
            torch.cuda.empty_cache()
            out['cpu_baseline'] = cpu_baseline(scans_xyz[:args.cpu_scans], poses[:args.cpu_scans], args.k,
                                               args.cpu_iters, 1e-3, args.cpu_variants)
            out['config']['gpu_over_cpu'] = value / out['cpu_baseline']['value']
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def main_c4(args, world, rank, local_rank, dev, dist):
    """BASELINE config 4 shape (KITTI-360-like sequences of 64 x 2048-ray scans, depth 5-25 m and 0.2 m voxel pre-filters, ball
    neighbourhoods of 0.4 m, joint model + per-pose optimisation with the point-to-plane ICP loss) through the reference-API
    train() (train.py:94-322): sequence q lives on rank q mod N, every rank back-propagates the ICP losses of its sequences and
    ONE packed all-reduce per iteration (distributed.GradReducer: [weighted loss, weight, grads of the shared model weights])
    joins them; pose corrections stay with their owner.  W warm-up + K timed iterations of ONE train() call; the clock starts in
    the callback of iteration W behind a barrier + device synchronisation and stops behind the same pair after train() returns."""
    import contextlib
    import io
    import tempfile
    from depth_correction_amd.config import Config, Loss, PoseCorrection
    from depth_correction_amd.dataset import KittiLikeDataset
    from depth_correction_amd.preproc import filtered_cloud
    from depth_correction_amd.train import TrainCallbacks, train
    n_seq = args.c4_sequences or world
    cfg = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0,
                 max_depth=25.0, vp_dispersion_bounds=[], n_opt_iters=args.warmup + args.steps, lr=1e-3, device=str(dev),
                 log_dir=tempfile.mkdtemp(), model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]})
    # every rank describes all sequences (train() loads only the ones it owns); a sequence's seeds differ by its index
    seqs, n_points = [], 0
    for q in range(n_seq):
        ds = KittiLikeDataset(n_poses=args.c4_scans, seed_base=2000 + 100 * q)
        if q % world == rank or world == 1:
            seq = [(filtered_cloud(cloud, cfg), pose) for cloud, pose in ds]
            n_points += sum(len(c) for c, _ in seq)
        else:
            seq = None                                   # another rank's: train() never touches it here
        seqs.append(seq)
    clock = {}

    class Clock(TrainCallbacks):
        def iteration_started(self, it):
            if it == args.warmup:
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                clock['t0'] = time.perf_counter()

    with contextlib.redirect_stdout(io.StringIO()):
        train(cfg, callbacks=Clock(cfg), train_datasets=seqs, val_datasets=[])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - clock['t0']
    stats = torch.tensor([elapsed, float(n_points)], dtype=torch.float64, device=dev)
    if dist is not None:
        el = stats[:1].clone()
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        pts = stats[1:].clone()
        dist.all_reduce(pts)
        elapsed, n_points = float(el.item()), float(pts.item())
    rccl = None
    if dist is not None:
        buf = torch.zeros((2 + 2,), dtype=torch.float64, device=dev)
        for _ in range(10):
            dist.all_reduce(buf)
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(100):
            dist.all_reduce(buf)
        a1.record()
        torch.cuda.synchronize()
        rccl = {'rccl_ranks': dist.get_world_size(), 'allreduce_us_per_step': a0.elapsed_time(a1) * 10.0}
    if rank == 0:
        out = {'metric': 'points/sec through the point-to-plane ICP loss fwd+bwd + Adam (BASELINE config 4 shape, joint model + pose '
                         'optimisation through train())',
               'value': n_points * args.steps / elapsed, 'unit': 'points/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': 'C4: %d KITTI-360-like sequence(s) of %d scans x 64 x 2048 rays, one per rank; depth 5-25 m + 0.2 m '
                                      'voxel pre-filters (%d points left in all), ball neighbourhoods r = 0.4 m, icp_loss (point to '
                                      'plane), ScaledPolynomial + per-pose corrections, Adam; train() of the reference API'
                                      % (n_seq, args.c4_scans, int(n_points)),
                          'points': 'points of all sequences after the pre-filters (every one is a row of the neighbourhood / feature '
                                    'set-up; the ICP loss itself pairs ~a tenth of them per iteration)',
                          'loop': 'depth_correction_amd.train.train(), per-iteration bookkeeping (several ranks: the plain loop with one '
                                  'all-reduce per iteration)'},
               'roofline': None, 'cpu_baseline': None}
        if rccl:
            out['config'].update(rccl)
            out.update(rccl)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
