"""One launch of 1024 identical instances (development aid for counter profiling)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmpc_loader
from oracle import synth
mm = mmpc_loader.load()
N, M = 20, 5
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
same = len(sys.argv) > 2 and sys.argv[2] == "same"
d = synth.make_batch(8192, N=N, M=M)
dev = torch.device("cuda", 0)
ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M)
eng = ctrl._engine
idx = np.zeros(B, int) if same else np.arange(B) % 8192
t = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx])).to(dev)
xi = t(np.clip(d["x_init"], ctrl.xlim[0], ctrl.xlim[1])); tr = t(d["traj_ref"]); ur = t(d["u_ref"]); ob = t(d["obs"])
ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
for _ in range(2):
    out = eng.solve_batch_device(xi, tr, ur, ul, ob)
torch.cuda.synchronize()
print("iters", out["iters"].float().mean().item())
