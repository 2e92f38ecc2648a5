"""Robustness sweep: other seeds of the C4 generator through the GPU path; converged fraction, iteration statistics, and
agreement with the CPU oracle on the first 256 instances of each seed."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mmpc_loader; mm = mmpc_loader.load()
from oracle import synth, coracle, nlp
B = 8192
ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=20, max_batch=B, n_obstacles=5)
par = nlp.WholeBodyParams()
args = [a for a in sys.argv[1:] if a != "--no-oracle"]
CHECK = "--no-oracle" not in sys.argv
for cid in [int(a) for a in args] or (3, 11, 12, 13, 14, 15):
    d = synth.make_batch(B, config_id=cid)
    ctrl.reset()
    r = ctrl.solve_batch(d["x_init"], d["traj_ref"], d["u_ref"], d["obs"])
    if not CHECK:
        top = np.sort(r["iters"])[-3:]
        print("seed %4d: converged %.5f  iters mean %.2f p99 %.0f top3 %s" % (cid, (r["status"] == 0).mean(), r["iters"].mean(),
              np.percentile(r["iters"], 99), top), flush=True)
        continue
    x = np.clip(d["x_init"][:256], par.xlim[0], par.xlim[1])
    o = coracle.solve_batch(par, x, d["traj_ref"][:256], d["u_ref"][:256], np.zeros((256, 20, 5)), d["obs"][:256], nthreads=8)
    same = np.abs(r["cost"][:256] / o["cost"] - 1) < 1e-6
    print("seed %2d: converged %.5f  iters mean %.2f p99 %.0f max %d | vs oracle (256): same minimum %.3f, max|dX| on those %.2e"
          % (cid, (r["status"] == 0).mean(), r["iters"].mean(), np.percentile(r["iters"], 99), r["iters"].max(),
             same.mean(), np.abs(r["X"][:256][same] - o["X"][same]).max()), flush=True)
