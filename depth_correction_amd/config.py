"""The slice of the reference's Config (config.py:143-292) that the hot path reads, as a plain attribute bag.

Same attribute names and defaults, ``from_dict`` / ``to_dict`` / ``copy`` / YAML round trip; the ROS-param,
argparse and roslaunch front ends and the experiment bookkeeping are out of scope.  Unlike the
reference's constructor this one does not shell out to ``git``.
"""
from __future__ import annotations

import copy as _copy
import os
import tempfile

import yaml

__all__ = ['Config', 'Loss', 'Model', 'NeighborhoodType', 'PoseCorrection', 'nonempty']


class _Names(type):
    def __iter__(cls):
        return iter(v for k, v in vars(cls).items() if not k.startswith('_') and isinstance(v, str))

    def __contains__(cls, item):
        return item in list(iter(cls))


class NeighborhoodType(metaclass=_Names):
    ball = 'ball'
    plane = 'plane'


class Loss(metaclass=_Names):
    min_eigval_loss = 'min_eigval_loss'
    trace_loss = 'trace_loss'
    icp_loss = 'icp_loss'


class Model(metaclass=_Names):
    Polynomial = 'Polynomial'
    ScaledPolynomial = 'ScaledPolynomial'


class PoseCorrection(metaclass=_Names):
    none = 'none'
    common = 'common'
    sequence = 'sequence'
    pose = 'pose'


def nonempty(iterable):
    return [x for x in iterable if x]


class Config(object):
    def __init__(self, **kwargs):
        self.random_seed = 135
        self.log_dir = os.path.join(tempfile.gettempdir(), 'depth_correction_amd')
        self.enable_ros = False
        # model (config.py:168-180)
        self.model_class = Model.ScaledPolynomial
        self.optimize_model = True
        self.model_args = []
        self.model_kwargs = {}
        self.model_state_dict = ''
        self.float_type = 'float64'
        self.device = 'cuda:0'          # the reference defaults to 'cpu'; this package has no CPU path
        # cloud preprocessing (:182-185)
        self.min_depth = 5.0
        self.max_depth = 25.0
        self.grid_res = 0.2
        # neighbourhoods (:186-194)
        self.nn_type = NeighborhoodType.ball
        self.nn_k = 0
        self.nn_r = 0.25
        self.nn_grid_res = 0.5
        self.min_valid_neighbors = 5
        self.max_neighborhoods = None
        self.nn_scale = None
        # filters (:204-218)
        self.shadow_neighborhood_angle = 0.017453
        self.shadow_angle_bounds = []
        self.dir_dispersion_bounds = []
        self.vp_dispersion_bounds = [0.36, float('inf')]
        self.vp_dispersion_to_depth2_bounds = []
        self.vp_dist_to_depth_bounds = []
        self.eigenvalue_bounds = []
        self.eigenvalue_ratio_bounds = [[0, 1, 0, 0.25], [1, 2, 0.25, 1.]]
        # data (:220-235)
        self.dataset = 'room'
        self.dataset_args = []
        self.dataset_kwargs = {}
        self.train_names = []
        self.val_names = []
        self.test_names = []
        self.data_start = None
        self.data_stop = None
        self.data_step = 1
        # training (:246-266)
        self.loss = Loss.min_eigval_loss
        self.loss_offset = False
        self.loss_kwargs = {'sqrt': False, 'normalization': True, 'inlier_max_loss': None, 'inlier_loss_mult': 1.0,
                            'inlier_ratio': 1.0, 'icp_inlier_ratio': 0.3, 'icp_point_to_plane': True}
        self.n_opt_iters = 100
        self.optimizer = 'Adam'
        self.optimizer_args = []
        self.optimizer_kwargs = {}
        self.lr = 2e-4
        # not in the reference: None = shard the sequences over the ranks whenever torch.distributed is initialised with
        # more than one (train.py of this package), False = every process trains on all sequences
        self.distributed = None
        self.pose_correction = PoseCorrection.none
        self.train_pose_deltas = None
        self.test_pose_deltas = None
        self.log_filters = False
        self.show_results = False
        # this build: use the fused per-sequence kernels whenever the configuration allows it
        self.depth_noise = 0.0           # dataset.noisy_dataset (config.py:242-244)
        self.pose_noise = 0.0
        self.pose_noise_mode = None
        self.fused = True
        # train(): iterations between two host synchronisations when nobody watches single iterations (no-op callbacks, one
        # process); 1 = the reference's per-iteration bookkeeping (train.py _batched_loop); loop_graph: replay the iteration as
        # one hipGraph
        self.loop_batch = 64
        self.loop_graph = True
        self.loop_graph_iters = 8      # iterations per captured graph of the loops with pose corrections (train._native_pose_loop)
        self.loop_native = True        # model-only runs: the library's chained step, one launch per iteration (train._native_loop)
        self.keep_plans = False        # train() releases the per-sequence plans it built when it returns; True keeps them cached
        self.from_dict(kwargs)

    # ---- Configurable subset (configurable.py:44-58,166-179) ----
    def from_dict(self, d):
        for k, v in d.items():
            setattr(self, k, v)
        return self

    def to_dict(self):
        return {k: v for k, v in vars(self).items() if not k.startswith('_')}

    def copy(self):
        return _copy.deepcopy(self)

    def diff(self, other):
        a, b = self.to_dict(), other.to_dict()
        return {k: v for k, v in a.items() if k not in b or b[k] != v}

    def non_default(self):
        return self.diff(type(self)())

    def to_yaml(self, path=None):
        text = yaml.safe_dump(self.to_dict())
        if path is None:
            return text
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, 'w') as f:
            f.write(text)

    def from_yaml(self, path):
        with open(path) as f:
            return self.from_dict(yaml.safe_load(f) or {})

    def data_slice(self):
        return slice(self.data_start, self.data_stop, self.data_step)

    def numpy_float_type(self):
        import numpy as np
        return getattr(np, self.float_type)

    def torch_float_type(self):
        import torch
        return getattr(torch, self.float_type)
