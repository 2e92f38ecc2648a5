"""Generic kernel on the C4 batch and on the C1 shape (2048 starts, intended rows / as written): ms and solves/s.  MMPC_LIB selects the library."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
sys.argv = [sys.argv[0]]
import bench, mmpc_loader
from oracle import synth
mm = mmpc_loader.load()
dev = torch.device("cuda", 0)
robot = mm.MobileManipulator(0.1)
d = synth.make_batch(8192)
os.environ["MMPC_FORCE_GENERIC"] = "1"
c = mm.MPCWholeBody(robot, [], [], N=20, max_batch=8192, n_obstacles=5)
del os.environ["MMPC_FORCE_GENERIC"]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x, tr, ur, ob = t(np.clip(d["x_init"], c.xlim[0], c.xlim[1])), t(d["traj_ref"]), t(d["u_ref"]), t(d["obs"])
ul = torch.zeros((8192, 20, 5), dtype=torch.float64, device=dev)
c._engine.set_schedule_hint(2)
o = c._engine.solve_batch_device(x, tr, ur, ul, ob)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); o = c._engine.solve_batch_device(x, tr, ur, ul, ob, out=o); e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1)
print("generic on C4 batch: %.1f ms, %.0f solves/s, conv %.4f, mean iters %.2f" % (ms, 8192 / ms * 1e3, float((o["status"] == 0).double().mean()), float(o["iters"].double().mean())))
r = bench.c1_shape_extra(mm, robot, dev)
print("c1 intended %.1f ms (%.0f/s)  as written %.1f ms (%.0f/s) conv %.3f | single solve %.2f ms" % (r["intended_rows"]["ms"], r["intended_rows"]["value"],
      r["rows_as_written"]["ms"], r["rows_as_written"]["value"], r["rows_as_written"]["converged_frac"], r["single_solve_latency_ms"]["median"]))
