import cProfile, pstats, io, sys, os, tempfile, contextlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np, torch
from depth_correction_amd.config import Config, Loss, PoseCorrection
from depth_correction_amd.dataset import KittiLikeDataset
from depth_correction_amd.preproc import filtered_cloud
from depth_correction_amd.train import TrainCallbacks, train
cfg = Config(loss=Loss.icp_loss, pose_correction=PoseCorrection.pose, nn_k=0, nn_r=0.4, grid_res=0.2, min_depth=5.0,
             max_depth=25.0, vp_dispersion_bounds=[], n_opt_iters=60, lr=1e-3, device='cuda:0',
             log_dir=tempfile.mkdtemp(), model_kwargs={'w': [1e-3, -1e-3], 'exponent': [2.0, 4.0]})
ds = KittiLikeDataset(n_poses=10)
seq = [(filtered_cloud(cloud, cfg), pose) for cloud, pose in ds]
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable(); train(cfg, callbacks=TrainCallbacks(cfg), train_datasets=[seq], val_datasets=[]); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:7000])
