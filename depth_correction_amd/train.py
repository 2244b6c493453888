"""Training loop with the reference's structure (train.py:46-327) for ball neighbourhoods: set-up once (local
feature clouds, pose corrections, model, optimizer, global neighbourhoods, masks), then iterate
{train loss, validation loss, checkpoint on joint improvement, backward, optimizer step}.

TensorBoard / ROS publishing are out of scope; ``TrainCallbacks`` is the hook for logging.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .config import Config, NeighborhoodType, PoseCorrection
from .dataset import create_dataset
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
        clouds, poses = [], []
        for cloud, pose in ds:
            clouds.append(local_feature_cloud(cloud, cfg))
            poses.append(pose)
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


def train(cfg: Config, callbacks=None, train_datasets=None, val_datasets=None):
    """Optimise the depth-correction model (and pose corrections); returns the config of the best iteration."""
    assert cfg.nn_type == NeighborhoodType.ball
    callbacks = callbacks or TrainCallbacks(cfg)
    os.makedirs(cfg.log_dir, exist_ok=True)
    cfg_path = os.path.join(cfg.log_dir, 'train.yaml')
    if not os.path.exists(cfg_path):
        cfg.to_yaml(cfg_path)
    train_datasets = train_datasets or [create_dataset(name, cfg) for name in cfg.train_names]
    val_datasets = val_datasets or [create_dataset(name, cfg) for name in cfg.val_names]
    loss_fun = create_loss(cfg)

    train_clouds, train_poses = _load_sequences(train_datasets, cfg)
    val_clouds, val_poses = _load_sequences(val_datasets, cfg)
    train_pose_deltas = initialize_pose_corrections(train_datasets, cfg)
    if cfg.pose_correction == PoseCorrection.common:
        val_pose_deltas = len(val_datasets) * [train_pose_deltas[0]]
    else:
        val_pose_deltas = initialize_pose_corrections(val_datasets, cfg)

    model = load_model(cfg=cfg, eval_mode=False)
    print(model)
    params = []
    if cfg.optimize_model and len(list(model.parameters())) > 0:
        params.append({'params': model.parameters(), 'lr': cfg.lr})
    if cfg.pose_correction != PoseCorrection.none:
        params.append({'params': train_pose_deltas, 'lr': cfg.lr})
    make_opt = lambda p: getattr(torch.optim, cfg.optimizer.split('.')[-1])(p, *(cfg.optimizer_args or []),
                                                                           **(cfg.optimizer_kwargs or {}))
    optimizer = make_opt(params)
    val_optimizer = None
    if cfg.pose_correction in (PoseCorrection.sequence, PoseCorrection.pose) and val_datasets:
        val_optimizer = make_opt([{'params': val_pose_deltas, 'lr': cfg.lr}])

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

    min_train_loss = min_val_loss = np.inf
    best_cfg = None
    for it in range(cfg.n_opt_iters):
        callbacks.iteration_started(it)
        train_loss, _, train_poses_upd, train_feat = eval_loss_clouds(train_clouds, train_poses, train_pose_deltas,
                                                                      train_masks, train_ns, model, loss_fun, cfg)
        callbacks.train_loss(it, model, train_feat, train_pose_deltas, train_poses_upd, train_masks, train_loss)
        if val_datasets:
            val_loss, _, val_poses_upd, val_feat = eval_loss_clouds(val_clouds, val_poses, val_pose_deltas, val_masks,
                                                                    val_ns, model, loss_fun, cfg)
            callbacks.val_loss(it, model, val_feat, val_pose_deltas, val_poses_upd, val_masks, val_loss)
        else:
            val_loss = train_loss.detach()

        saved = train_loss.item() < min_train_loss and val_loss.item() < min_val_loss
        if saved:
            min_val_loss = val_loss.item()
            stem = '%s/%03i_%.6g' % (cfg.log_dir, it, min_val_loss)
            torch.save(model.state_dict(), stem + '_state_dict.pth')
            torch.save([p.detach().clone() for p in train_pose_deltas if p is not None], stem + '_pose_deltas.pth')
            torch.save([p.detach().clone() for p in train_poses_upd if p is not None], stem + '_poses_upd.pth')
            best_cfg = cfg.copy()
            best_cfg.model_state_dict = stem + '_state_dict.pth'
            best_cfg.train_pose_deltas = stem + '_pose_deltas.pth'
            best_cfg.to_yaml(os.path.join(cfg.log_dir, 'best.yaml'))
        print('It. %03i: train loss: %.9f, val.: %.9f. Model %s %s.'
              % (it, train_loss.item(), val_loss.item(), model, 'saved' if saved else 'not saved'))

        optimizer.zero_grad()
        train_loss.backward()
        if cfg.pose_correction == PoseCorrection.pose:
            for d in train_pose_deltas:
                d.grad[0].zero_()                      # the first pose stays fixed (train.py:309-311)
        optimizer.step()
        if val_optimizer is not None:
            val_optimizer.zero_grad()
            val_loss.backward()
            if cfg.pose_correction == PoseCorrection.pose:
                for d in val_pose_deltas:
                    d.grad[0].zero_()
            val_optimizer.step()
    return best_cfg
