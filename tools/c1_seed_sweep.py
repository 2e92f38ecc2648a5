"""The NLP as written on the demo's shapes over other seeds of the start generator (oracle/synth.py:make_c1_starts; seed 11 is the
one the tests and the bench use): converged fraction and iteration statistics, two and three planes."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mmpc_loader; mm = mmpc_loader.load()
from oracle import synth
B, N = 2048, 20
for npl in (2, 3):
    ctrl = None
    for seed in [int(a) for a in sys.argv[1:]] or range(11, 21):
        x, tr, obs, hs = synth.make_c1_starts(B, N, nplanes=npl, seed=seed)
        if ctrl is None:
            oml = [(h[:3], h[3:].reshape(1, 3)) for h in hs]
            ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], oml, N=N, max_batch=B, n_obstacles=3)
        ctrl.reset()
        r = ctrl.solve_batch(x, tr, np.zeros((B, N, 5)), obs)
        bad = np.nonzero(r["status"] != 0)[0]
        print("%d planes, seed %3d: converged %.5f  iters mean %.2f p99 %.0f max %d%s" % (npl, seed, (r["status"] == 0).mean(), r["iters"].mean(),
              np.percentile(r["iters"], 99), r["iters"].max(), ("  not converged: %s" % [(int(b), int(r["status"][b]), int(r["iters"][b])) for b in bad[:8]]) if len(bad) else ""), flush=True)
