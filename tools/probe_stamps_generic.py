"""Per-phase wave-cycle shares of the GENERIC kernel on the demo's shape (NLP as written, two planes): diagnostic build
libmmpc_gstamp.so (-DMMPC_STAMP_GEN), 256 starts = one per CU."""
import sys, os, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["MMPC_LIB"] = os.environ.get("MMPC_STAMP_LIB", os.path.join(ROOT, "mobile-manipulator-mpc_amd", "csrc", "libmmpc_gstamp.so"))
import torch, mmpc_loader
from oracle import synth
mm = mmpc_loader.load()
B, N = 256, 20
x, tr, obs, hs = synth.make_c1_starts(B, N)
oml = [(h[:3], h[3:].reshape(1, 3)) for h in hs]
for name, fc in (("intended rows", False), ("rows as written", None)):
    ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], oml, N=N, max_batch=B, n_obstacles=3, faithful_convex=fc)
    L = mm._capi.lib()
    buf = (ctypes.c_ulonglong * 16)()
    z = np.zeros((B, N, 5))
    ctrl.solve_batch(x, tr, z, obs)
    L.mmpc_debug_read_gstamps(buf)
    r = ctrl.solve_batch(x, tr, z, obs)
    L.mmpc_debug_read_gstamps(buf)
    v = np.array(list(buf), float)[:10]
    names = ["update (move to the accepted point)", "E1 evaluation + convergence / barrier update", "A1 + A2 assembly", "R0 + Riccati pass (+ border roll-out)", "forward roll-out",
             "D1", "D2 row steps", "merit at alpha = 0", "line-search trials", "exit"]
    its = r["iters"].sum()
    print("== %s: %d solves, mean %.1f iterations" % (name, B, r["iters"].mean()))
    for i, n in enumerate(names):
        print("  %-48s %5.1f %%  %9.0f cycles/iter/wave" % (n, 100 * v[i] / v.sum(), v[i] / its))
    print("  total %.0f cycles/iter/wave" % (v.sum() / its))
    v2 = np.array(list(buf), float)[10:16]
    if v2.sum() > 0:
        for i, n in enumerate(["R0 terminal block", "R1 + R2 MFMA chains", "operands of the next stage (stage_entry)", "R3 elimination legs", "P store", "gains + border columns in the pass"]):
            print("     inside the pass: %-42s %9.0f cycles/iter/wave" % (n, v2[i] / its))
        print("     inside the pass: %-42s %9.0f cycles/iter/wave" % ("rest (border roll-out, small solve, retries)", (v[3] - v2.sum()) / its))
