import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, mmpc_loader
from oracle import synth
mm = mmpc_loader.load()
B, N, M = 8192, 30, 8
d = synth.make_batch(B, N=N, M=M, config_id=5, moving=True)
obs = np.zeros((B, 31, 8, 3))
for k in range(31):
    obs[:, k, :, :2] = d["obs"][:, :, :2] + d["obs_vel"] * k * 0.1
    obs[:, k, :, 2] = d["obs"][:, :, 2]
ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M, obs_per_stage=True)
r = ctrl.solve_batch(d["x_init"], d["traj_ref"], d["u_ref"], obs)
print("C5 cold tick: converged", (r["status"] == 0).sum(), "of", B, "iters mean %.2f max %d" % (r["iters"].mean(), r["iters"].max()))
