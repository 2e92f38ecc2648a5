import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, mmpc_loader
from oracle import synth
mm = mmpc_loader.load()
N, M, B = 10, 3, 8192
d = synth.make_batch(B, N=N, M=M, config_id=11)
dev = torch.device("cuda", 0)
ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M)
eng = ctrl._engine
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xi = t(np.clip(d["x_init"], ctrl.xlim[0], ctrl.xlim[1])); tr = t(d["traj_ref"]); ur = t(d["u_ref"]); ob = t(d["obs"])
ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
out = eng.solve_batch_device(xi, tr, ur, ul, ob); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(6):
    e0.record(); eng.solve_batch_device(xi, tr, ur, ul, ob, out=out); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
it = out["iters"].cpu().numpy()
print("N=10 M=3 B=%d lds %d: ms %.3f iters mean %.1f max %d conv %.4f" % (B, eng.lds_bytes, min(ts), it.mean(), it.max(), (out["status"] == 0).float().mean().item()))
