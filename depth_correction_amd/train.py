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

import numpy as np
import torch

from .config import Config, NeighborhoodType, PoseCorrection
from .dataset import create_dataset
from .distributed import GradReducer, gather_objects, shard_sequences, world_info
from .eval import eval_loss_clouds, initialize_pose_corrections
from .loss import create_loss, icp_correspondences
from .model import load_model
from .preproc import establish_neighborhoods, global_cloud, global_cloud_mask, local_feature_cloud

__all__ = ['TrainCallbacks', 'train']


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


def train(cfg: Config, callbacks=None, train_datasets=None, val_datasets=None):
    """Optimise the depth-correction model (and pose corrections); returns the config of the best iteration."""
    assert cfg.nn_type == NeighborhoodType.ball
    callbacks = callbacks or TrainCallbacks(cfg)
    rank, world = world_info() if getattr(cfg, 'distributed', None) is not False else (0, 1)
    sharded = world > 1
    if sharded and torch.device(cfg.device).type == 'cuda':
        # object collectives (the checkpoint gather) and RCCL's own staging use torch's current device: it must be this
        # rank's GPU, whatever the launcher did
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
