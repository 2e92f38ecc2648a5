"""The generic kernel against the specialised one on whole C4 batches (seeds of the generator): converged fractions, iteration
statistics, and agreement of the solutions (same minimum: relative cost difference < 1e-6; max |dX| on those)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mmpc_loader; mm = mmpc_loader.load()
from oracle import synth
B = 8192
fast = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=20, max_batch=B, n_obstacles=5)
os.environ["MMPC_FORCE_GENERIC"] = "1"
gen = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=20, max_batch=B, n_obstacles=5)
del os.environ["MMPC_FORCE_GENERIC"]
for cid in [int(a) for a in sys.argv[1:]] or range(3, 13):
    d = synth.make_batch(B, config_id=cid)
    fast.reset(); gen.reset()
    a = fast.solve_batch(d["x_init"], d["traj_ref"], d["u_ref"], d["obs"])
    b = gen.solve_batch(d["x_init"], d["traj_ref"], d["u_ref"], d["obs"])
    both = (a["status"] == 0) & (b["status"] == 0)
    same = both & (np.abs(b["cost"] / a["cost"] - 1) < 1e-6)
    print("seed %3d: converged fast %.5f generic %.5f | iters mean %.2f / %.2f max %d / %d | same minimum %.5f, max|dX| on those %.2e" % (
        cid, (a["status"] == 0).mean(), (b["status"] == 0).mean(), a["iters"].mean(), b["iters"].mean(), a["iters"].max(), b["iters"].max(),
        same.mean(), np.abs(a["X"][same] - b["X"][same]).max()), flush=True)
