"""GPU probe: kernel time vs batch size and vs iteration-count skew (development aid, not a test)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mmpc_loader
from oracle import synth

mm = mmpc_loader.load()
N, M = 20, 5
d = synth.make_batch(8192, N=N, M=M)
dev = torch.device("cuda", 0)


def run(B, same=None, reps=5, label=""):
    ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M)
    eng = ctrl._engine
    idx = np.arange(B) % 8192 if same is None else np.full(B, same)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx])).to(dev)
    xi = t(np.clip(d["x_init"], ctrl.xlim[0], ctrl.xlim[1])); tr = t(d["traj_ref"]); ur = t(d["u_ref"]); ob = t(d["obs"])
    ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
    out = eng.solve_batch_device(xi, tr, ur, ul, ob)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); eng.solve_batch_device(xi, tr, ur, ul, ob, out=out); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    it = out["iters"].cpu().numpy()
    print("%-28s B=%5d  ms=%8.3f  iters mean %.1f max %d  -> us/iter/wave-slot(1024 slots) %.2f" % (
        label, B, min(ts), it.mean(), it.max(), min(ts) * 1e3 / max(1.0, it.sum() / min(B, 1024))))


if len(sys.argv) > 1 and sys.argv[1] == "quick":
    run(1024, same=0, label="1024 copies of inst 0")
    run(8192, label="8192 distinct")
else:
    run(64, same=0, label="64 copies of inst 0")
    run(256, same=0, label="256 copies of inst 0")
    run(1024, same=0, label="1024 copies of inst 0")
    run(2048, same=0, label="2048 copies of inst 0")
    run(8192, same=0, label="8192 copies of inst 0")
    run(1024, label="first 1024 distinct")
    run(8192, label="8192 distinct")
