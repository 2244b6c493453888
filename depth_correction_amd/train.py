"""Training loop with the reference's structure (train.py:46-327) for ball neighbourhoods: set-up once (local
feature clouds, pose corrections, model, optimizer, global neighbourhoods, masks), then iterate
{train loss, validation loss, checkpoint on joint improvement, backward, optimizer step}.

Multi-GPU (BASELINE configs 3 and 4): when ``torch.distributed`` is initialised with more than one rank
(one process per GPU, ``torchrun``) and ``cfg.distributed`` is not False, the sequences (= datasets) are dealt round
robin over the ranks.  Every rank builds and evaluates only its own sequences -- min-eigenvalue / trace loss or
point-to-plane ICP, with its sequences' pose corrections -- and ONE packed all-reduce per iteration
(``distributed.GradReducer``: [weighted loss, weight, d/dw, d/d(common pose correction)]) joins the ranks: the model
and a ``PoseCorrection.common`` 6-vector are replicated and stepped identically everywhere, per-sequence / per-pose
corrections live on (and are optimised by) the rank that owns the sequence.  Rank 0 writes the checkpoints.

TensorBoard / ROS publishing are out of scope; ``TrainCallbacks`` is the hook for logging.
"""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch

from .config import Config, NeighborhoodType, PoseCorrection
from .dataset import create_dataset
from .distributed import GradReducer, gather_objects, shard_sequences, world_info
from .eval import eval_loss_clouds, initialize_pose_corrections
from .loss import create_loss, icp_correspondences
from .model import load_model
from .preproc import establish_neighborhoods, global_cloud, global_cloud_mask, local_feature_cloud

__all__ = ['TrainCallbacks', 'train', 'release_plans']


class TrainCallbacks(object):
    def __init__(self, cfg: Config = None):
        self.cfg = cfg

    def iteration_started(self, iter):
        pass

    def train_inputs(self, iter, clouds, poses):
        pass

    def val_inputs(self, iter, clouds, poses):
        pass

    def train_loss(self, iter, model, clouds, pose_deltas, poses, masks, loss):
        pass

    def val_loss(self, iter, model, clouds, pose_deltas, poses, masks, loss):
        pass


def _sharding(cfg):
    """(rank, world, sharded): sequences are dealt over the ranks of an initialised process group unless cfg.distributed is False.
    DC_FORCE_DIST=1 makes a one-rank group take the sharded code paths too (collectives included): how they are exercised, and
    timed, on a one-GPU box."""
    if getattr(cfg, 'distributed', None) is False:
        return 0, 1, False
    rank, world = world_info()
    forced = False
    if world == 1 and os.environ.get('DC_FORCE_DIST') == '1':
        import torch.distributed as dist
        forced = dist.is_available() and dist.is_initialized()
    return rank, world, world > 1 or forced


def _only_watches_the_clock(callbacks):
    """True when nothing looks at an iteration's tensors while it runs: the callbacks are the no-op base class, or a subclass that
    overrides ``iteration_started`` only (a host-side hook the native loops call before they launch an iteration)."""
    return isinstance(callbacks, TrainCallbacks) and all(getattr(type(callbacks), name, None) is getattr(TrainCallbacks, name)
                                                         for name in ('train_inputs', 'val_inputs', 'train_loss', 'val_loss'))


def _agree(code, device):
    """The same value on every rank: ``code`` when all ranks hold it, else 0 (one small all-reduce at set-up)."""
    from .distributed import all_reduce_sum
    import torch.distributed as dist
    v = torch.tensor([float(code), -float(code)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
    hi, lo = float(v[0].item()), -float(v[1].item())
    return int(hi) if hi == lo else 0


def _load_sequences(datasets, cfg):
    all_clouds, all_poses = [], []
    for ds in datasets:
        raw, poses = [], []
        for cloud, pose in ds:
            raw.append(cloud)
            poses.append(pose)
        # independent scans that do not fill the chip one at a time: a few streams side by side (pipeline.on_streams)
        if torch.device(cfg.device).type == 'cuda':
            from .pipeline import on_streams
            clouds = on_streams([lambda c=c: local_feature_cloud(c, cfg) for c in raw], cfg.device)
        else:
            clouds = [local_feature_cloud(c, cfg) for c in raw]
        all_clouds.append(clouds)
        all_poses.append(torch.as_tensor(np.stack(poses).astype(cfg.numpy_float_type()), device=cfg.device))
    return all_clouds, all_poses


def _icp_masks(all_clouds, all_poses, ratio):
    """Correspondences of consecutive scans, found once on the GPU (train.py:178-210)."""
    out = []
    for clouds, poses in zip(all_clouds, all_poses):
        seq = []
        for j in range(len(clouds) - 1):
            p1 = clouds[j].transform(poses[j]).to_points()
            p2 = clouds[j + 1].transform(poses[j + 1]).to_points()
            mask1, idx2, _ = icp_correspondences(p1, p2, ratio)
            seq.append((mask1, idx2))
        out.append(seq)
    return out


def _loss_weight(cfg, clouds, loss_clouds):
    """What the (local) mean loss has to be multiplied with so that the weighted losses of disjoint sets of sequences
    add up: the number of pointwise terms behind a min-eigenvalue / trace mean (batch_loss concatenates the pointwise
    losses of all sequences and reduces once, loss.py:205-213), the number of sequences behind an ICP mean
    (loss.py:403)."""
    if cfg.loss == 'icp_loss':
        return float(len(clouds))
    kw = cfg.loss_kwargs
    total = 0.0
    for c in (loss_clouds or []):
        cnt = getattr(c, 'count', None)                       # PlanCloud: masked points of the sequence's plan
        if cnt is None:
            pw = c.loss
            keep = pw.isfinite() if kw.get('only_finite') else (~pw.isnan() if kw.get('skip_nans') else None)
            cnt = pw.numel() if keep is None else int(keep.sum())
        total += float(cnt)
    return total


def _zero_first_pose(pose_deltas):
    for d in pose_deltas:
        if d is not None and d.grad is not None:
            d.grad[0].zero_()                                 # the first pose stays fixed (train.py:309-311)


def _plan_registries():
    from . import eval as _eval, loss as _loss
    return [r for r in (_eval._plans, getattr(_loss, '_icp_plans', None)) if r is not None]


def release_plans(keep=None):
    """Drop the cached per-sequence plans (eval._plans, loss._icp_plans): each holds the local clouds, tables and work buffers
    of a sequence -- hundreds of MB at C2 -- and nothing else releases them in a long-lived process.  ``keep``: per registry
    the set of keys to leave alone (train() passes the keys that existed before it started, so that plans another caller
    built -- an evaluation loop around train(), a train() inside a callback -- survive it)."""
    for i, reg in enumerate(_plan_registries()):
        if keep is None:
            reg.clear()
        else:
            reg.discard_except(keep[i])


def train(cfg: Config, callbacks=None, train_datasets=None, val_datasets=None):
    """Optimise the depth-correction model (and pose corrections); returns the config of the best iteration.  The plans built
    for its sequences are released when it returns (``cfg.keep_plans = True`` keeps them for a caller that goes on
    evaluating the same clouds)."""
    before = [set(reg.keys()) for reg in _plan_registries()]
    try:
        return _train(cfg, callbacks, train_datasets, val_datasets)
    finally:
        if not getattr(cfg, 'keep_plans', False):
            release_plans(keep=before)                        # only what this call added


def _train(cfg: Config, callbacks=None, train_datasets=None, val_datasets=None):
    assert cfg.nn_type == NeighborhoodType.ball
    callbacks = callbacks or TrainCallbacks(cfg)
    rank, world, sharded = _sharding(cfg)
    if sharded and torch.device(cfg.device).type == 'cuda' and torch.device(cfg.device).index is not None:
        # object collectives (the checkpoint gather) and RCCL's own staging use torch's current device: it must be this
        # rank's GPU, whatever the launcher did.  An index-less 'cuda' means "the current device" (the launcher has already
        # called set_device(local_rank)): nothing to change, and set_device would refuse a device without an index
        torch.cuda.set_device(torch.device(cfg.device))
    os.makedirs(cfg.log_dir, exist_ok=True)
    cfg_path = os.path.join(cfg.log_dir, 'train.yaml')
    if rank == 0 and not os.path.exists(cfg_path):
        cfg.to_yaml(cfg_path)
    if train_datasets is None:
        train_datasets = [create_dataset(name, cfg) for name in cfg.train_names]
    if val_datasets is None:
        val_datasets = [create_dataset(name, cfg) for name in cfg.val_names]
    n_train, n_val = len(train_datasets), len(val_datasets)
    if sharded:
        # sequence q lives on rank q mod world: only its owner loads it, searches its neighbourhoods and evaluates it
        train_datasets = [train_datasets[i] for i in shard_sequences(n_train, rank, world)]
        val_datasets = [val_datasets[i] for i in shard_sequences(n_val, rank, world)]
    loss_fun = create_loss(cfg)

    train_clouds, train_poses = _load_sequences(train_datasets, cfg)
    val_clouds, val_poses = _load_sequences(val_datasets, cfg)
    common = None
    if cfg.pose_correction == PoseCorrection.common:
        # ONE 6-vector for every sequence of every rank (eval.py:48-53): replicated, created even by a rank without sequences
        common = torch.zeros((1, 6), dtype=cfg.torch_float_type(), device=cfg.device, requires_grad=True)
        train_pose_deltas = len(train_datasets) * [common]
        val_pose_deltas = len(val_datasets) * [common]
    else:
        train_pose_deltas = initialize_pose_corrections(train_datasets, cfg)
        val_pose_deltas = initialize_pose_corrections(val_datasets, cfg)

    model = load_model(cfg=cfg, eval_mode=False)
    print(model)
    params = []
    model_params = list(model.parameters()) if cfg.optimize_model else []
    if model_params:
        params.append({'params': model_params, 'lr': cfg.lr})
    if cfg.pose_correction == PoseCorrection.common:
        params.append({'params': [common], 'lr': cfg.lr})
    elif cfg.pose_correction != PoseCorrection.none and train_pose_deltas:
        params.append({'params': train_pose_deltas, 'lr': cfg.lr})
    def make_opt(p):
        name, args, kw = cfg.optimizer.split('.')[-1], list(cfg.optimizer_args or []), dict(cfg.optimizer_kwargs or {})
        if name == 'Adam' and not args and set(kw) <= {'lr', 'betas', 'eps', 'weight_decay'} and torch.device(cfg.device).type == 'cuda':
            from .optim import Adam                   # torch.optim.Adam's update, one launch per parameter (optim.py)
            return Adam(p, **kw)
        return getattr(torch.optim, name)(p, *args, **kw)
    optimizer = make_opt(params) if params else None
    val_optimizer = None
    if cfg.pose_correction in (PoseCorrection.sequence, PoseCorrection.pose) and val_datasets:
        val_optimizer = make_opt([{'params': val_pose_deltas, 'lr': cfg.lr}])
    per_sequence = cfg.pose_correction in (PoseCorrection.sequence, PoseCorrection.pose)
    shared = model_params + ([common] if common is not None else [])
    train_reducer = GradReducer(shared, train_pose_deltas if per_sequence else ()) if sharded else None
    val_reducer = GradReducer((), val_pose_deltas if per_sequence else ()) if sharded else None

    # neighbourhoods and masks of the global clouds, established once (train.py:166-215)
    train_global = [global_cloud(clouds=c, poses=p) for c, p in zip(train_clouds, train_poses)]
    val_global = [global_cloud(clouds=c, poses=p) for c, p in zip(val_clouds, val_poses)]
    train_ns = [establish_neighborhoods(cloud=c, cfg=cfg) for c in train_global]
    val_ns = [establish_neighborhoods(cloud=c, cfg=cfg) for c in val_global]
    if cfg.loss == 'icp_loss':
        ratio = cfg.loss_kwargs['icp_inlier_ratio']
        train_masks, val_masks = _icp_masks(train_clouds, train_poses, ratio), _icp_masks(val_clouds, val_poses, ratio)
    else:
        train_masks = [global_cloud_mask(c, c.mask, cfg) for c in train_global]
        val_masks = [global_cloud_mask(c, c.mask, cfg) for c in val_global]
    del train_global, val_global

    def evaluate(clouds, poses, deltas, masks, ns):
        """(local mean loss with its autograd graph | None, its weight, poses_upd, feature clouds) of this rank's sequences."""
        if not clouds:                                              # more ranks than sequences: nothing to evaluate here
            return None, 0.0, [], []
        loss, loss_clouds, poses_upd, feat = eval_loss_clouds(clouds, poses, deltas, masks, ns, model, loss_fun, cfg)
        return loss, (_loss_weight(cfg, clouds, loss_clouds) if sharded else 1.0), poses_upd, feat

    def weighted_of(loss, weight):
        return loss * weight if loss is not None else torch.zeros((), dtype=torch.float64, device=cfg.device)

    batch = int(getattr(cfg, 'loop_batch', 64) or 1)
    if (batch > 1 and _only_watches_the_clock(callbacks) and torch.device(cfg.device).type == 'cuda' and cfg.n_opt_iters > 0):
        # nobody looks at an iteration while it runs: the loop runs without a host synchronisation per iteration -- see _batched_loop.
        # Sharded sequences (one process per GPU): the same loops with the iteration's ONE all-reduce enqueued on the stream between
        # the evaluations and the finishing launches; every rank must take the same loop, so they agree on it first
        use_native = getattr(cfg, 'loop_native', True)
        shard = dict(rank=rank, world=world, n_train=n_train, n_val=n_val) if sharded else None
        native = _native_loop_plan(cfg, model, optimizer, train_clouds, train_poses, train_masks, train_ns, val_clouds, val_poses,
                                   val_masks, val_ns, sharded) if use_native else None
        pose_native = None
        if native is None and use_native:
            pose_native = _native_pose_loop_plan(cfg, model, optimizer, val_optimizer, train_clouds, train_poses, train_masks, train_ns,
                                                 train_pose_deltas, val_clouds, val_poses, val_masks, val_ns, val_pose_deltas)
        code = 1 if native is not None else (2 if pose_native is not None else 0)
        if sharded:
            code = _agree(code, cfg.device)
        if code == 1:
            if not sharded:
                ran, best = _native_loop(cfg, model, optimizer, native[0], native[1], train_poses, val_poses, batch, callbacks)
            else:
                ran, best = _native_shared_loop(cfg, model, optimizer, native[0], native[1], train_poses, val_poses, batch, shard, callbacks)
            if ran:
                return best
            code = 0
        if code == 2:
            return _native_pose_loop(cfg, model, optimizer, val_optimizer, pose_native[0], pose_native[1], train_poses, val_poses,
                                     train_pose_deltas, val_pose_deltas, batch, shard, callbacks)
        if not sharded:
            return _batched_loop(cfg, model, optimizer, val_optimizer, train_pose_deltas, val_pose_deltas, n_val, batch,
                                 lambda: evaluate(train_clouds, train_poses, train_pose_deltas, train_masks, train_ns),
                                 lambda: evaluate(val_clouds, val_poses, val_pose_deltas, val_masks, val_ns), callbacks)

    min_train_loss = min_val_loss = np.inf
    best_cfg = None
    for it in range(cfg.n_opt_iters):
        callbacks.iteration_started(it)
        train_loss, weight, train_poses_upd, train_feat = evaluate(train_clouds, train_poses, train_pose_deltas, train_masks,
                                                                   train_ns)
        if sharded:
            # the local weighted loss is back-propagated first, so that loss and gradients travel in the SAME all-reduce
            if optimizer is not None:
                optimizer.zero_grad()
            weighted = weighted_of(train_loss, weight)
            if weighted.requires_grad:
                weighted.backward()
            train_loss, _ = train_reducer.reduce(weighted, weight, with_grads=True)
        callbacks.train_loss(it, model, train_feat, train_pose_deltas, train_poses_upd, train_masks, train_loss)
        if n_val:
            val_loss, weight, val_poses_upd, val_feat = evaluate(val_clouds, val_poses, val_pose_deltas, val_masks, val_ns)
            if sharded:
                weighted = weighted_of(val_loss, weight)
                if val_optimizer is not None:
                    val_optimizer.zero_grad()
                    if weighted.requires_grad:
                        weighted.backward(inputs=[d for d in val_pose_deltas if d is not None])
                val_loss, _ = val_reducer.reduce(weighted, weight, with_grads=val_optimizer is not None)
            callbacks.val_loss(it, model, val_feat, val_pose_deltas, val_poses_upd, val_masks, val_loss)
        else:
            val_loss = train_loss.detach()

        saved = train_loss.item() < min_train_loss and val_loss.item() < min_val_loss
        if saved:
            min_val_loss = val_loss.item()
            stem = '%s/%03i_%.6g' % (cfg.log_dir, it, min_val_loss)
            deltas = [p.detach().cpu().clone() for p in train_pose_deltas if p is not None]
            poses_out = [p.detach().cpu().clone() for p in train_poses_upd if p is not None]
            if sharded:
                # sequence order: rank r owns sequences r, r + world, ...; every rank takes part in the gather
                by_rank = gather_objects((shard_sequences(n_train, rank, world), deltas, poses_out), device=cfg.device)
                deltas, poses_out = n_train * [None], n_train * [None]
                for idx, ds_, ps_ in by_rank:
                    for k, i in enumerate(idx):
                        deltas[i] = ds_[k] if k < len(ds_) else None
                        poses_out[i] = ps_[k] if k < len(ps_) else None
                deltas, poses_out = [d for d in deltas if d is not None], [p for p in poses_out if p is not None]
            if rank == 0:
                torch.save(model.state_dict(), stem + '_state_dict.pth')
                torch.save(deltas, stem + '_pose_deltas.pth')
                torch.save(poses_out, stem + '_poses_upd.pth')
            best_cfg = cfg.copy()
            best_cfg.model_state_dict = stem + '_state_dict.pth'
            best_cfg.train_pose_deltas = stem + '_pose_deltas.pth'
            if rank == 0:
                best_cfg.to_yaml(os.path.join(cfg.log_dir, 'best.yaml'))
        if rank == 0:
            print('It. %03i: train loss: %.9f, val.: %.9f. Model %s %s.'
                  % (it, train_loss.item(), val_loss.item(), model, 'saved' if saved else 'not saved'))

        if not sharded:
            if optimizer is not None:
                optimizer.zero_grad()
                train_loss.backward()
        if cfg.pose_correction == PoseCorrection.pose:
            _zero_first_pose(train_pose_deltas)
        if optimizer is not None:
            optimizer.step()
        if val_optimizer is not None:
            if not sharded:
                val_optimizer.zero_grad()
                val_loss.backward()
            if cfg.pose_correction == PoseCorrection.pose:
                _zero_first_pose(val_pose_deltas)
            val_optimizer.step()
    return best_cfg


class _Bookkeeper(object):
    """The reference's per-iteration bookkeeping (train.py:219-244: improvement rule, progress line, checkpoint files, best
    config), fed with RECORDED iterations in order.  A batch of records ends with ``end_batch``, which writes the files of the
    batch's last improvement."""

    def __init__(self, cfg, model, shard=None):
        """shard: None, or dict(rank, world, n_train, n_val) when the sequences are dealt over ranks -- every rank replays the same
        records (the losses are all-reduced: identical everywhere), rank 0 prints and writes, and a checkpoint gathers the pose
        corrections / corrected poses of the sequences the other ranks own (one object gather per batch, at its last improvement)."""
        import copy
        self.cfg, self.min_val, self.best, self._last = cfg, np.inf, None, None
        self.shard = shard
        self.lead = shard is None or shard['rank'] == 0
        # formats the progress line from a recorded state, on the host.  .to() and not .cpu(): the models move their plain tensor
        # attributes (fixed exponents) in to() only, and a progress line that reads a device tensor waits for every queued iteration
        self.shadow = copy.deepcopy(model).to('cpu')
        self.shadow_sd = self.shadow.state_dict()

    def record(self, it, tl, vl, sd, deltas, poses):
        """sd: {name: CPU tensor} of the model at iteration ``it``; deltas / poses: lists of CPU tensors."""
        saved = tl < np.inf and vl < self.min_val                 # (the reference never lowers its min_train_loss)
        if saved:
            self.min_val = vl
            self._last = (it, vl, sd, deltas, poses)
        if not self.lead:
            return
        for k, v in sd.items():
            self.shadow_sd[k].copy_(v)
        print('It. %03i: train loss: %.9f, val.: %.9f. Model %s %s.' % (it, tl, vl, self.shadow, 'saved' if saved else 'not saved'))

    def record_fast(self, it, tl, vl, key, value, payload):
        """record() for the loops that log into a device ring: only the tensor ``key`` of the model's state changes from
        iteration to iteration (``value``: a NumPy row), and what a checkpoint needs -- (state dict, corrections, poses) -- is built
        by ``payload()`` only for the iteration whose files are written at the end of the batch."""
        saved = tl < np.inf and vl < self.min_val
        if saved:
            self.min_val = vl
            self._last = (it, vl, payload)
        if not self.lead:
            return
        t = self.shadow_sd[key]
        t.copy_(torch.from_numpy(np.ascontiguousarray(value)).reshape(t.shape))
        print('It. %03i: train loss: %.9f, val.: %.9f. Model %s %s.' % (it, tl, vl, self.shadow, 'saved' if saved else 'not saved'))

    def end_batch(self):
        if self._last is None:
            return
        if len(self._last) == 3:
            it, vl, payload = self._last
            sd, deltas, poses = payload()
        else:
            it, vl, sd, deltas, poses = self._last
        self._last = None
        cfg = self.cfg
        stem = '%s/%03i_%.6g' % (cfg.log_dir, it, vl)
        if self.shard is not None:
            # sequence order: rank r owns sequences r, r + world, ...; every rank takes part in the gather (train.py:232-240 saves
            # the corrections and corrected poses of ALL training sequences)
            sh = self.shard
            by_rank = gather_objects((shard_sequences(sh['n_train'], sh['rank'], sh['world']), [d.clone() for d in deltas],
                                      [p.clone() for p in poses]), device=cfg.device)
            deltas, poses = sh['n_train'] * [None], sh['n_train'] * [None]
            for idx, ds_, ps_ in by_rank:
                for k, i in enumerate(idx):
                    deltas[i] = ds_[k] if k < len(ds_) else None
                    poses[i] = ps_[k] if k < len(ps_) else None
            deltas, poses = [d for d in deltas if d is not None], [p for p in poses if p is not None]
        best = cfg.copy()
        best.model_state_dict = stem + '_state_dict.pth'
        best.train_pose_deltas = stem + '_pose_deltas.pth'
        if self.lead:
            torch.save({k: v.clone() for k, v in sd.items()}, stem + '_state_dict.pth')
            torch.save([d.clone() for d in deltas], stem + '_pose_deltas.pth')
            torch.save([p.clone() for p in poses], stem + '_poses_upd.pth')
            self._write_yaml(best, os.path.join(cfg.log_dir, 'best.yaml'))
        self.best = best

    def _write_yaml(self, best, path):
        """best.to_yaml(path), re-serialising only the entries that changed since the last call: a block-style dump of a mapping is
        the concatenation of the dumps of its entries in key order (checked on the first call), and of ~60 entries two change
        from checkpoint to checkpoint (2.5 ms -> 0.3 ms per checkpoint, which is per batch of the ring-logged loops)."""
        import copy
        import yaml
        d = best.to_dict()
        cache = getattr(self, '_yaml_cache', None)
        if cache is None:
            full = yaml.safe_dump(d)
            parts = {k: yaml.safe_dump({k: v}) for k, v in d.items()}
            if ''.join(parts[k] for k in sorted(d)) != full:
                self._yaml_cache = False                         # not a plain block mapping after all: always dump in full
            else:
                self._yaml_cache = {k: (copy.deepcopy(v), parts[k]) for k, v in d.items()}
            text = full
        elif cache is False or set(cache) != set(d):
            text = yaml.safe_dump(d)
        else:
            for k, v in d.items():
                if cache[k][0] != v:
                    cache[k] = (copy.deepcopy(v), yaml.safe_dump({k: v}))
            text = ''.join(cache[k][1] for k in sorted(d))
        with open(path, 'w') as f:
            f.write(text)


def _batched_loop(cfg, model, optimizer, val_optimizer, train_pose_deltas, val_pose_deltas, n_val, batch, eval_train, eval_val,
                  callbacks=None):
    """The loop of train() (train.py:220-322) without a host synchronisation per iteration.

    The reference reads ``train_loss.item()`` / ``val_loss.item()`` in every iteration (checkpoint decision and the progress
    line) -- a device round trip that costs several times what the GPU needs for the iteration itself.  Here every iteration
    records {train loss, validation loss, the model's state, the pose corrections, the corrected poses} -- what a checkpoint
    of that iteration would hold, a few hundred bytes -- into a device ring of ``batch`` slots; after every ``batch``
    iterations the host synchronises ONCE, replays the reference's bookkeeping over the recorded iterations in order (same
    progress lines, same improvement rule) and writes the checkpoint files of the LAST improvement of the batch (earlier ones
    of the same batch would be superseded immediately; ``cfg.loop_batch = 1`` restores one file set per improvement).  After
    three eager iterations the iteration {losses -> record -> backward -> optimiser steps} is captured once as a hipGraph and
    replayed (the library launches on torch's current stream, optim.Adam counts its steps on the device), so the host issues
    one graph launch per iteration; configurations that cannot be captured keep running eagerly, still without the
    per-iteration round trip.  Losses, weights and pose corrections are those of the plain loop, iteration by iteration."""
    dev = torch.device(cfg.device)
    n_it, R = cfg.n_opt_iters, batch
    sd_items = [(k, v) for k, v in model.state_dict().items()]                   # tensors that alias the live parameters
    deltas = []
    for d in train_pose_deltas:
        if d is not None and all(d is not q for q in deltas):
            deltas.append(d)
    # one fp64 record per iteration: [train loss, val loss | model state | pose corrections | corrected poses], written by
    # three small launches (cat, index_copy, counter) whatever the number of tensors
    ring, layout = [], []                     # ring[0]: [R, record length]; layout: (kind, key / index, shape, dtype, offset, numel)
    counter = torch.zeros((1,), dtype=torch.int64, device=dev)

    def body():
        train_loss, _, train_poses_upd, _ = eval_train()
        if n_val:
            val_loss, _, _, _ = eval_val()
        else:
            val_loss = train_loss.detach()
        poses_now = [p for p in train_poses_upd if p is not None]
        parts = ([train_loss.detach().double().reshape(1), val_loss.detach().double().reshape(1)]
                 + [v.detach().double().reshape(-1) for _, v in sd_items] + [d.detach().double().reshape(-1) for d in deltas]
                 + [p.detach().double().reshape(-1) for p in poses_now])
        if not ring:
            at = 2
            for kind, items in (('sd', [v for _, v in sd_items]), ('delta', deltas), ('pose', poses_now)):
                for i, v in enumerate(items):
                    layout.append((kind, i, tuple(v.shape), v.dtype, at, v.numel()))
                    at += v.numel()
            ring.append(torch.zeros((R, at), dtype=torch.float64, device=dev))
        ring[0].index_copy_(0, counter.remainder(R), torch.cat(parts).unsqueeze(0))
        counter.add_(1)
        if optimizer is not None:
            optimizer.zero_grad()
            train_loss.backward()
        if cfg.pose_correction == PoseCorrection.pose:
            _zero_first_pose(train_pose_deltas)
        if optimizer is not None:
            optimizer.step()
        if val_optimizer is not None:
            val_optimizer.zero_grad()
            val_loss.backward()
            if cfg.pose_correction == PoseCorrection.pose:
                _zero_first_pose(val_pose_deltas)
            val_optimizer.step()

    book = _Bookkeeper(cfg, model)
    state = dict(done=0)

    def drain(upto):
        """Bookkeeping of iterations [done, upto) from the ring; ONE synchronisation."""
        if upto <= state['done']:
            return
        h = ring[0].cpu()                                              # synchronises
        fields = lambda sl, kind: [h[sl, at:at + n].reshape(shape).to(dtype).clone()
                                   for k_, _, shape, dtype, at, n in layout if k_ == kind]
        for it in range(state['done'], upto):
            sl = it % R
            book.record(it, float(h[sl, 0]), float(h[sl, 1]), {k: v for (k, _), v in zip(sd_items, fields(sl, 'sd'))},
                        fields(sl, 'delta'), fields(sl, 'pose'))
        book.end_batch()
        state['done'] = upto

    graph = None
    it = 0
    warm = min(3, n_it)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    started = callbacks.iteration_started if callbacks is not None else (lambda i: None)
    with torch.cuda.stream(side):
        for _ in range(warm):                                # real iterations 0 .. warm - 1, off the default stream as captures ask
            started(it)
            body()
            it += 1
    torch.cuda.current_stream(dev).wait_stream(side)
    if n_it - it >= 4 and getattr(cfg, 'loop_graph', True):
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            graph = g
        except Exception as ex:                               # something in this configuration synchronises: stay eager
            print('train(): the iteration could not be captured (%s: %s); running it eagerly' % (type(ex).__name__, str(ex).split('\n')[0]))
            torch.cuda.synchronize(dev)
            graph = None
    try:
        while it < n_it:
            if it - state['done'] >= R:
                drain(it)
            started(it)
            if graph is not None:
                graph.replay()
            else:
                body()
            it += 1
    except BaseException:
        # an interrupted run (Ctrl-C, an exception in an iteration) keeps what it finished: the ring holds the records of the
        # completed iterations since the last drain, so best.yaml and the checkpoint files of the best of them get written
        try:
            drain(it)
        except Exception:
            pass
        raise
    drain(n_it)
    return book.best


def _native_pose_loop_plan(cfg, model, optimizer, val_optimizer, train_clouds, train_poses, train_masks, train_ns, train_pose_deltas,
                           val_clouds, val_poses, val_masks, val_ns, val_pose_deltas):
    """([train plans], [validation plans]) when the loop WITH per-pose / per-sequence corrections can run on plan.PoseSequenceTrainer
    -- up to sixteen training and validation sequences through the fused min-eigenvalue / trace loss without inlier gating, or
    through the ICP losses; the weights of a polynomial model and the corrections optimised by Adam as train() builds it -- else
    None."""
    from .eval import _plan_for, fused_supported
    from .optim import Adam
    kw = cfg.loss_kwargs
    w = getattr(model, 'w', None)
    icp = cfg.loss == 'icp_loss'
    if icp:
        # icp_loss's own fused path (loss.icp_loss): precomputed correspondences, normals for point to plane, a kernel model
        plane = bool(kw['icp_point_to_plane'])
        ok_loss = (all(m is not None for m in list(train_masks) + list(val_masks))
                   and all(c.dirs.is_cuda and (not plane or c.normals is not None) for seq in list(train_clouds) + list(val_clouds) for c in seq)
                   and getattr(model, 'kernel_kind', None) is not None)
    else:
        ok_loss = (fused_supported(train_clouds, model, cfg) and (not val_clouds or fused_supported(val_clouds, model, cfg))
                   and kw.get('inlier_ratio', 1.0) >= 1.0 and kw.get('inlier_max_loss') is None
                   and not kw.get('only_finite') and not kw.get('skip_nans'))
    if not (cfg.pose_correction in (PoseCorrection.pose, PoseCorrection.sequence) and 1 <= len(train_clouds) <= 16 and len(val_clouds) <= 16
            and isinstance(optimizer, Adam) and (val_optimizer is None or isinstance(val_optimizer, Adam)) and ok_loss
            and isinstance(w, torch.nn.Parameter) and [id(p) for p in model.parameters()] == [id(w)]
            and w.is_cuda and w.dtype == torch.float64 and w.is_contiguous() and 1 <= w.numel() <= 3
            and model.kernel_params()[0] is w and not model.kernel_params()[1].requires_grad
            and all(d is not None for d in list(train_pose_deltas) + list(val_pose_deltas))
            and (bool(val_clouds) == (val_optimizer is not None))):
        return None
    groups = optimizer.param_groups
    want = ([[id(w)]] if cfg.optimize_model else []) + [[id(d) for d in train_pose_deltas]]
    if [[id(p) for p in g['params']] for g in groups] != want:
        return None
    vgroups = val_optimizer.param_groups if val_optimizer is not None else []
    if any(g['weight_decay'] != 0.0 or g['betas'] != groups[0]['betas'] or g['eps'] != groups[0]['eps'] for g in list(groups) + list(vgroups)):
        return None
    if any(m is None for m in list(train_masks) + list(val_masks)):
        return None
    if icp:
        from .loss import _icp_sequence_plan
        plans = [_icp_sequence_plan(c, m, True, plane) for c, m in zip(train_clouds, train_masks)]
        vplans = [_icp_sequence_plan(c, m, True, plane) for c, m in zip(val_clouds, val_masks)]
        return plans, vplans
    plans = [_plan_for(c, p, nn, m, model, cfg) for c, p, nn, m in zip(train_clouds, train_poses, train_ns, train_masks)]
    vplans = [_plan_for(c, p, nn, m, model, cfg) for c, p, nn, m in zip(val_clouds, val_poses, val_ns, val_masks)]
    return plans, vplans


def _native_pose_loop(cfg, model, optimizer, val_optimizer, plans, vplans, train_poses, val_poses, train_pose_deltas, val_pose_deltas,
                      batch, shard=None, callbacks=None):
    """train()'s loop with pose corrections on the library's own launches (plan.PoseSequenceTrainer): per iteration and sequence
    one evaluation (the pose kernel + its reduction) and ONE finishing launch that back-propagates through the pose chain, keeps
    the first pose fixed, takes both Adam updates, forms the next iteration's poses and writes the iteration's record into the
    log's ring -- against ~35 small launches of tensor glue per iteration around the same evaluation otherwise (0.26 -> 0.11 ms
    at C2).  Several sequences: their sums are joined on the device (dc_pose_train_combine: the loss of eval.py:85-112 divides the
    sum of the sequences' sums by the sum of their counts, icp_loss averages their losses) and every finishing launch scales by
    the joint divisor; the first sequence's also steps the weights.  Bookkeeping as in _batched_loop: one synchronisation per
    ``batch`` iterations.  The corrections are optimised in fp64 and written back to the caller's tensors (whatever their dtype)
    when the loop ends.
    Sharded sequences (``shard``; BASELINE config 4: one KITTI-360-like sequence per GPU): the joint sums of the training and the
    validation loss -- [2][loss, divisor, dL/dw], one launch (dc_pose_train_combine2) -- go through the iteration's ONE all-reduce,
    enqueued on the stream between the evaluations and the finishing launches; every rank then takes the identical update of the
    weights, and the corrections stay with the rank that owns the sequence.  No host synchronisation, no autograd."""
    from .distributed import all_reduce_sum
    from .plan import PoseSequenceTrainer
    dev = torch.device(cfg.device)
    n_it, R = cfg.n_opt_iters, batch
    groups = optimizer.param_groups
    w_param = model.w
    w = w_param.detach().reshape(-1)                               # (a view: the updates land in the parameter's storage)
    e = model.kernel_params()[1].detach().reshape(-1).contiguous()
    nt = w.numel()
    g_d = groups[-1]
    zero_first = cfg.pose_correction == PoseCorrection.pose
    icp_kind = model.kernel_kind if cfg.loss == 'icp_loss' else None
    trs = [PoseSequenceTrainer(p_, T, d, zero_first, g_d['lr'], g_d['betas'], g_d['eps'], n_terms=nt, icp_model_kind=icp_kind)
           for p_, T, d in zip(plans, train_poses, train_pose_deltas)]
    tr, plan = trs[0], plans[0]
    vtr = [PoseSequenceTrainer(vp, T, d, zero_first, val_optimizer.param_groups[0]['lr'], g_d['betas'], g_d['eps'], n_terms=nt,
                               icp_model_kind=icp_kind)
           for vp, T, d in zip(vplans, val_poses, val_pose_deltas)]
    # a model that is not optimised still goes into the first sequence's finishing launch, with a zero learning rate: the
    # iteration's record (progress line, *_state_dict.pth) takes its weights from there
    lr_w = groups[0]['lr'] if cfg.optimize_model else 0.0
    w_m, w_v = torch.zeros_like(w), torch.zeros_like(w)
    # several sequences in a loss (or in other ranks' hands): their sums joined on the device -- both losses in one launch, then one
    # all-reduce -- and every finishing launch scaled by the joint divisor; every record carries the joint sums of both losses
    sharded = shard is not None
    has_val = (shard['n_val'] if sharded else len(vtr)) > 0
    totals2 = torch.zeros((2, 2 + nt), dtype=torch.float64, device=dev) if (sharded or len(trs) > 1 or len(vtr) > 1) else None
    totals, vtotals = (totals2[0], totals2[1]) if totals2 is not None else (None, None)
    extra = totals2.reshape(-1) if totals2 is not None else None
    n_extra = 0 if extra is None else extra.numel()
    rings = [torch.zeros((R, t_.record_len + n_extra), dtype=torch.float64, device=dev) for t_ in trs]
    vrings = [torch.zeros((R, v.record_len + n_extra), dtype=torch.float64, device=dev) for v in vtr]
    sd_const = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    w_key = [k for k, v in model.state_dict().items() if v.data_ptr() == w_param.data_ptr()][0]
    d_dtype, T_dtype = train_pose_deltas[0].dtype, train_poses[0].dtype
    all_plans = list(plans) + list(vplans)
    book = _Bookkeeper(cfg, model, shard)

    def fetch():
        """The ring on the host (synchronises: every launched iteration has finished)."""
        bits = 0
        hs, hv = [r_.cpu() for r_ in rings], [v.cpu() for v in vrings]          # synchronises
        for p_ in all_plans:
            bits |= p_.status_bits() if hasattr(p_, 'status_bits') else 0
        return hs, hv, bits

    def bookkeep(fetched, first, upto):
        hs, hv, bits = fetched
        if bits and bits & plan.STATUS_OVERFLOW:
            warnings.warn('train(): points left the extent of the 32-bit fixed-point format (or are NaN) in iterations %d..%d: '
                          'their losses are NaN; build the plan with point_format="f64" for maps that grow this much'
                          % (first, upto - 1))
        HS, HV = [h.numpy() for h in hs], [v.numpy() for v in hv]
        a0 = tr.head + 2 * nt + 12 * plan.n_scans

        def joint_loss(group, rows):
            if tr.icp:                                                # icp_loss: the mean of the sequences' losses (loss.py:403)
                return float(sum(r_[0] for r_ in rows) / len(rows))
            cnt = float(sum(r_[1] for r_ in rows))                    # eval.py:85-112: sum of the sums / sum of the counts
            return float(sum(r_[0] for r_ in rows)) / cnt if cnt > 0 else float('nan')

        def payload_of(rows):
            def build():
                parts = [t_.split_record(torch.from_numpy(r_[:t_.record_len])) for t_, r_ in zip(trs, rows)]
                sd = dict(sd_const)
                sd[w_key] = parts[0][1].reshape(w_param.shape).to(w_param.dtype).clone()
                return sd, [p_[2].to(d_dtype).clone() for p_ in parts], [p_[3].to(T_dtype).clone() for p_ in parts]
            return build

        for it in range(first, upto):
            rows = [H[it % R] for H in HS]
            if n_extra:                                               # the joint sums over all sequences (and ranks) ride in every record
                t2 = rows[0][-n_extra:].reshape(2, 2 + nt)
                tl = float(t2[0, 0] / t2[0, 1]) if t2[0, 1] > 0 else float('nan')
                vl = (float(t2[1, 0] / t2[1, 1]) if t2[1, 1] > 0 else float('nan')) if has_val else tl
            else:
                tl = joint_loss(trs, rows)
                vl = joint_loss(vtr, [V[it % R] for V in HV]) if vtr else tl
            book.record_fast(it, tl, vl, w_key, rows[0][a0:a0 + nt], payload_of([r_.copy() for r_ in rows]))
        book.end_batch()

    def body():
        if totals2 is None and tr.icp:
            # an ICP loss that waits for nobody (one sequence, one rank): evaluation, sums and the finishing step in ONE launch
            # (dc_icp_sequence_step); the validation sequence first -- it reads the weights the training launch is about to step
            done_v = [v.evaluate_finish(w, e, None, None, None, 0.0, vr) for v, vr in zip(vtr, vrings)]
            for v, vr, ok_ in zip(vtr, vrings, done_v):
                if not ok_:
                    v.evaluate(w, e)
                    v.finish(None, None, None, 0.0, vr)
            if not tr.evaluate_finish(w, e, w, w_m, w_v, lr_w, rings[0]):
                tr.evaluate(w, e)
                tr.finish(w, w_m, w_v, lr_w, rings[0])
            return
        for t_ in trs:
            t_.evaluate(w, e)
        for v in vtr:
            v.evaluate(w, e)                                         # validation with the weights of THIS iteration
        if totals2 is not None:
            PoseSequenceTrainer.combine2(trs, vtr, totals2)
            if sharded:
                all_reduce_sum(totals2)                                  # the iteration's ONE collective
        for q, (t_, r_) in enumerate(zip(trs, rings)):               # the first sequence's launch also steps the weights
            first = q == 0
            t_.finish(w if first else None, w_m if first else None, w_v if first else None, lr_w if first else 0.0, r_, totals, extra)
        for v, vr in zip(vtr, vrings):
            v.finish(None, None, None, 0.0, vr, vtotals, extra)

    state = dict(launched=0, graph=None, tried=False)
    G = max(1, min(int(getattr(cfg, 'loop_graph_iters', 8) or 1), R))     # iterations per captured graph

    def run_some(left):
        """Launch the next iteration(s), at most ``left``; returns how many."""
        # (sharded: launched eagerly -- a collective inside a captured graph is not something this path relies on)
        if (state['launched'] >= 3 and not state['tried'] and n_it - state['launched'] >= G + 3 and getattr(cfg, 'loop_graph', True)
                and not sharded):
            # every launch of an iteration takes the same pointers (the record's ring slot follows the device step counter): G
            # iterations are captured once, after three eager ones (pose tables built, allocator warm), and replayed -- one graph
            # launch per G iterations.  (An iteration is one to three short launches: replayed one by one the HOST's ~45 us per
            # graph launch would set the pace of an ICP iteration whose single launch takes half of that.)
            state['tried'] = True
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            try:
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_, stream=side):
                    for _ in range(G):
                        body()
                state['graph'] = g_
                # (a capture launches nothing: the G iterations it recorded have not run)
            except Exception as ex:                                   # not capturable here: keep launching eagerly
                print('train(): the iteration could not be captured as a graph (%s); running eagerly' % (ex,))
            torch.cuda.current_stream(dev).wait_stream(side)
        n = G if (state['graph'] is not None and left >= G) else 1
        if callbacks is not None:
            for q in range(n):
                callbacks.iteration_started(state['launched'] + q)
        if n == G and state['graph'] is not None:
            state['graph'].replay()
        else:
            body()
        state['launched'] += n
        return n

    # batches of R iterations: the records of a batch are fetched (one synchronisation) BEFORE the next batch is launched into the
    # same ring, and replayed through the reference's bookkeeping WHILE the device runs that next batch
    prev, start = None, 0
    try:
        while start < n_it:
            end = min(start + R, n_it)
            fetched = fetch() if prev is not None else None
            it_ = start
            while it_ < end:
                it_ += run_some(end - it_)
            if prev is not None:
                bookkeep(fetched, *prev)
            prev, start = (start, end), end
        if prev is not None:
            bookkeep(fetch(), *prev)
            prev = None
    except BaseException:
        try:                                                          # an interrupted run keeps the batch it had finished
            if prev is not None and state['launched'] == prev[1] and not sharded:
                bookkeep(fetch(), *prev)
        except Exception:
            pass
        raise
    finally:
        with torch.no_grad():                                         # the caller's tensors follow the optimisation
            for d, t_ in zip(list(train_pose_deltas) + list(val_pose_deltas), trs + vtr):
                d.copy_(t_.delta)
        torch.autograd.graph.increment_version(w_param)            # written through its pointer
    return book.best


def _native_loop_plan(cfg, model, optimizer, train_clouds, train_poses, train_masks, train_ns, val_clouds, val_poses, val_masks, val_ns,
                      sharded=False):
    """([train plans], [validation plans]) when the whole loop can run on plan.SequenceTrainer's native steps -- only the weights
    of a polynomial model are optimised, with Adam as train() builds it, over up to sixteen (local) training sequences through the
    fused min-eigenvalue / trace loss without inlier gating -- else None."""
    from .eval import _plan_for, fused_supported
    from .optim import Adam
    kw = cfg.loss_kwargs
    w = getattr(model, 'w', None)
    if not (cfg.pose_correction == PoseCorrection.none and cfg.optimize_model and 1 <= len(train_clouds) <= 16 and len(val_clouds) <= 16
            and isinstance(optimizer, Adam) and len(optimizer.param_groups) == 1
            and fused_supported(train_clouds, model, cfg) and (not val_clouds or fused_supported(val_clouds, model, cfg))
            and kw.get('inlier_ratio', 1.0) >= 1.0 and kw.get('inlier_max_loss') is None
            # (NaN-dropping reductions divide by the evaluation's OWN count, loss.py:125-137; the chained step scales its gradient
            #  by the plan's static count)
            and not kw.get('only_finite') and not kw.get('skip_nans')
            and isinstance(w, torch.nn.Parameter) and [id(p) for p in model.parameters()] == [id(w)]
            and w.is_cuda and w.dtype == torch.float64 and w.is_contiguous() and 1 <= w.numel() <= 3
            and model.kernel_params()[0] is w and not model.kernel_params()[1].requires_grad):
        return None
    g = optimizer.param_groups[0]
    if g['weight_decay'] != 0.0 or any(m is None for m in list(train_masks) + list(val_masks)):
        return None
    plans = [_plan_for(c, p, nn, m, model, cfg) for c, p, nn, m in zip(train_clouds, train_poses, train_ns, train_masks)]
    vplans = [_plan_for(c, p, nn, m, model, cfg) for c, p, nn, m in zip(val_clouds, val_poses, val_ns, val_masks)]
    return plans, vplans


def _native_shared_loop(cfg, model, optimizer, plans, vplans, train_poses, val_poses, batch, shard, callbacks=None):
    """train()'s model-only loop when the loss runs over SEVERAL training sequences and / or the sequences are dealt over ranks
    (BASELINE config 3; train.py:166-175 loops over the sequences, eval.py:85-112 divides the sum of their sums by the sum of their
    counts).  Per iteration, all on the stream and without a host synchronisation:

        evaluation of the first local sequence, whose launch first takes the previous iteration's Adam update from the joint
        gradient (dc_sequence_eval_after_update: it also records the weights it uses) -> evaluations of the other local training
        sequences and of the local validation sequences with those weights -> ONE launch that joins the local sums of both
        losses (dc_pose_train_combine2) -> ONE all-reduce of [training | validation] x {sum, count, dL/dw} over the ranks.

    The joint sums land in the log's ring row; bookkeeping as in _batched_loop (one synchronisation per ``batch`` iterations; rank 0
    prints and writes).  Returns (ran, best config); ran = False, nothing touched, when the plans cannot take this path."""
    from .distributed import all_reduce_sum
    from .plan import SequenceTrainer, combine_sums
    dev = torch.device(cfg.device)
    n_it, R = cfg.n_opt_iters, batch
    sharded = shard is not None
    g = optimizer.param_groups[0]
    w_param = model.w
    e = model.kernel_params()[1].detach()
    tr = SequenceTrainer(plans, w_param.detach(), e, train_poses, lr=g['lr'], betas=g['betas'], eps=g['eps'], distributed=sharded, chained=True)
    assert tr.w.data_ptr() == w_param.data_ptr()
    can = 1 if tr.update_in_next and not any(getattr(p_, 'nan_policy', None) for p_ in list(plans) + list(vplans)) else 0
    if (sharded and _agree(can, cfg.device) != 1) or not can:
        return False, None
    nt = tr.nt
    has_val = (shard['n_val'] if sharded else len(vplans)) > 0
    ring = torch.zeros((R, 2, 2 + nt), dtype=torch.float64, device=dev)     # per iteration: the joint sums of both losses
    ring_w = torch.zeros((R, nt), dtype=torch.float64, device=dev)          # and the weights it used
    vouts = [torch.zeros((2 + 2 * nt + 12 * vp.n_scans,), dtype=torch.float64, device=dev) for vp in vplans]
    vP = [vp.poses12(T) for vp, T in zip(vplans, val_poses)]
    sd_const = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    w_key = [k for k, v in model.state_dict().items() if v.data_ptr() == w_param.data_ptr()][0]
    poses_cpu = [T.detach().cpu().clone() for T in train_poses]
    book = _Bookkeeper(cfg, model, shard)
    all_plans = list(plans) + list(vplans)

    def launch(it):
        slot = it % R
        if callbacks is not None:
            callbacks.iteration_started(it)
        if tr.step(w_used_prev=ring_w[slot], require_chain=True, defer_reduce=True) is None:
            return False
        for vp, P, vo in zip(vplans, vP, vouts):                     # validation with the weights of THIS iteration
            vp.eval_native(tr.w, tr.exponent, P, vo, want_grad=False)
        combine_sums(tr.outs, vouts, ring[slot])
        if sharded:
            all_reduce_sum(ring[slot])                                # the iteration's ONE collective
        tr.use_sums(ring[slot, 0])
        return True

    def fetch():
        h, hw = ring.cpu(), ring_w.cpu()                              # synchronises
        bits = 0
        for p_ in all_plans:
            bits |= p_.status_bits()
        return h.numpy(), hw.numpy(), bits

    def bookkeep(fetched, first, upto):
        H, HW, bits = fetched
        if bits & all_plans[0].STATUS_OVERFLOW:
            warnings.warn('train(): points left the extent of the 32-bit fixed-point format (or are NaN) in iterations %d..%d: '
                          'their losses are NaN; build the plan with point_format="f64" for maps that grow this much'
                          % (first, upto - 1))
        for it in range(first, upto):
            sums = H[it % R]
            tl = float(sums[0, 0] / sums[0, 1]) if sums[0, 1] > 0 else float('nan')
            vl = (float(sums[1, 0] / sums[1, 1]) if sums[1, 1] > 0 else float('nan')) if has_val else tl
            w_row = np.array(HW[it % R], dtype=np.float64, copy=True)

            def payload(w_row=w_row):
                sd = dict(sd_const)
                sd[w_key] = torch.from_numpy(w_row).reshape(w_param.shape).to(w_param.dtype).clone()
                return sd, [], poses_cpu
            book.record_fast(it, tl, vl, w_key, w_row, payload)
        book.end_batch()

    prev, start, launched = None, 0, 0
    try:
        while start < n_it:
            end = min(start + R, n_it)
            fetched = fetch() if prev is not None else None
            for it in range(start, end):
                if not launch(it):
                    if it or sharded:
                        raise RuntimeError('train(): a sequence plan refused the native step after the loop had been chosen')
                    return False, None                             # nothing was launched
                launched = it + 1
            if prev is not None:
                bookkeep(fetched, *prev)
            prev, start = (start, end), end
        if prev is not None:
            bookkeep(fetch(), *prev)
            prev = None
        tr.flush()                                                  # the last iteration's Adam update (train.py:312)
    except BaseException:
        try:                                                        # an interrupted run keeps the batch it had finished
            if prev is not None and launched == prev[1] and not sharded:
                bookkeep(fetch(), *prev)
        except Exception:
            pass
        raise
    finally:
        torch.autograd.graph.increment_version(w_param)            # written through its pointer
    return True, book.best


def _native_loop(cfg, model, optimizer, plans, vplans, train_poses, val_poses, batch, callbacks=None):
    """train()'s loop for the model-only case on the library's own step: ONE launch per (sequence and) iteration.  One training
    sequence: dc_sequence_step_chained_rec -- the launch of iteration t evaluates loss and dL/dw with the weights Adam update t - 1
    left, and its leading blocks first finish iteration t - 1: sums, update, and the record {sums, weights used} of that iteration
    straight into the log's ring slot.  Several training sequences in the loss (train.py:172-175; round 5): a chain over the
    sequences, dc_sequence_step_linked -- every launch finishes the one before it, the first launch of an iteration completes the
    previous iteration.  Plus one evaluation-only launch pair per validation sequence.  The weights ARE ``model.w`` (the trainer
    works on the parameter's storage).  Bookkeeping as in _batched_loop: one synchronisation per ``batch`` iterations.  Returns
    (ran, best config); ran = False, nothing touched, when the plans turn out not to chain."""
    from .plan import SequenceTrainer
    dev = torch.device(cfg.device)
    n_it, R = cfg.n_opt_iters, batch
    g = optimizer.param_groups[0]
    w_param = model.w
    e = model.kernel_params()[1].detach()
    plan = plans[0]
    tr = SequenceTrainer(plans, w_param.detach(), e, train_poses, lr=g['lr'], betas=g['betas'], eps=g['eps'], chained=True)
    assert tr.w.data_ptr() == w_param.data_ptr()
    nt = tr.nt
    ring = torch.zeros((R, 2 + 2 * nt + 12 * plan.n_scans), dtype=torch.float64, device=dev)
    ring_w = torch.zeros((R, nt), dtype=torch.float64, device=dev)
    vrings = [torch.zeros((R, 2 + 2 * nt + 12 * vp.n_scans), dtype=torch.float64, device=dev) for vp in vplans]
    vP = [vp.poses12(T) for vp, T in zip(vplans, val_poses)]
    sd_const = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    w_key = [k for k, v in model.state_dict().items() if v.data_ptr() == w_param.data_ptr()][0]
    poses_cpu = [T.detach().cpu().clone() for T in train_poses]
    book = _Bookkeeper(cfg, model)
    snap = None                      # optimiser state after the last fetched iteration: where a batch can be started again

    def snapshot():
        return tr.w.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.t

    def record(it, sums, w_used, vsums):
        tl = float(sums[0] / sums[1]) if sums[1] > 0 else float('nan')
        if vsums:
            vs, vc = sum(float(v[0]) for v in vsums), sum(float(v[1]) for v in vsums)
            vl = vs / vc if vc > 0 else float('nan')
        else:
            vl = tl
        w_row = np.array(w_used, dtype=np.float64, copy=True)

        def payload():
            sd = dict(sd_const)
            sd[w_key] = torch.from_numpy(w_row).reshape(w_param.shape).to(w_param.dtype).clone()
            return sd, [], poses_cpu
        book.record_fast(it, tl, vl, w_key, w_row, payload)

    def plain_iterations(first):
        """Iterations first .. n_it - 1 with ordinary (two-launch) steps and one synchronisation each: where the loop goes on
        after a chained launch gave up waiting for its weights (status bit 1: its sums are NaN, and so is every weight the
        Adam updates after it produced)."""
        tr.chained = tr.linked = False
        for it in range(first, n_it):
            w_used = tr.w.clone()
            vs = [vp.eval_native(tr.w, tr.exponent, P, vr[0], want_grad=False).cpu() for vp, P, vr in zip(vplans, vP, vrings)]
            sums = tr.step().cpu()
            record(it, sums.numpy(), w_used.cpu().numpy(), [v.numpy() for v in vs])
            book.end_batch()

    def fetch(first, upto):
        """The records of iterations first .. upto - 1 on the host (ONE synchronisation), or None when the chain had to be abandoned
        (the rest of the run has then been done by plain_iterations)."""
        nonlocal snap
        # iteration upto - 1 is still pending in the chain: its weights are the current ones, its sums come with the flush
        ring_w[(upto - 1) % R].copy_(tr.w)
        tr.flush(out=ring[(upto - 1) % R])
        h, hw = ring.cpu(), ring_w.cpu()                            # synchronises
        hv = [v.cpu() for v in vrings]
        bits = 0
        for p_ in plans:
            bits |= p_.status_bits()                                # (free: the copies above have synchronised)
        if bits & plan.STATUS_CHAIN_TIMEOUT:
            warnings.warn('train(): a chained step gave up waiting for its weights (iterations %d..%d); repeating them and '
                          'finishing the run with ordinary steps' % (first, upto - 1))
            for p_ in plans:
                p_.clear_status()
            w0_, m0_, v0_, t0_ = snap
            tr.w.copy_(w0_); tr.exp_avg.copy_(m0_); tr.exp_avg_sq.copy_(v0_); tr.t = t0_
            plain_iterations(first)
            return None
        if bits & plan.STATUS_OVERFLOW:
            warnings.warn('train(): points left the extent of the 32-bit fixed-point format (or are NaN) in iterations %d..%d: '
                          'their losses are NaN; build the plan with point_format="f64" for maps that grow this much'
                          % (first, upto - 1))
        snap = snapshot()                                           # the optimiser after iteration upto - 1: where a batch can restart
        return h.numpy(), hw.numpy(), [v.numpy() for v in hv]

    def bookkeep(fetched, first, upto):
        H, HW, HV = fetched
        for it in range(first, upto):
            sl = it % R
            record(it, H[sl], HW[sl], [v[sl] for v in HV])
        book.end_batch()

    # batches of R iterations: the records of a batch are fetched (one synchronisation) BEFORE the next batch is launched into the
    # same ring, and replayed through the reference's bookkeeping WHILE the device runs that next batch
    snap = snapshot()
    prev, fetched, start, launched = None, None, 0, 0
    try:
        while start < n_it:
            end = min(start + R, n_it)
            if prev is not None:
                fetched = fetch(*prev)
                if fetched is None:
                    prev = None
                    break
            for it in range(start, end):
                slot = (it - 1) % R
                if callbacks is not None:
                    callbacks.iteration_started(it)
                if tr.step(out_prev=ring[slot], w_used_prev=ring_w[slot], require_chain=True) is None:
                    assert it == 0
                    return False, None                             # this plan does not chain; nothing was launched
                for vp, P, vr in zip(vplans, vP, vrings):            # validation with the weights of THIS iteration
                    vp.eval_native(tr.w, tr.exponent, P, vr[it % R], want_grad=False)
                launched = it + 1
            if prev is not None:
                bookkeep(fetched, *prev)
                fetched = None
            prev, start = (start, end), end
        if prev is not None:
            fetched = fetch(*prev)
            if fetched is not None:
                bookkeep(fetched, *prev)
            prev = fetched = None
    except BaseException:
        # an interrupted run keeps the batch it had finished: its records are on the host already, or still whole in the ring
        try:
            if prev is not None:
                if fetched is None and launched == prev[1]:
                    fetched = fetch(*prev)
                if fetched is not None:
                    bookkeep(fetched, *prev)
        except Exception:
            pass
        raise
    finally:
        torch.autograd.graph.increment_version(w_param)            # written through its pointer
    return True, book.best
