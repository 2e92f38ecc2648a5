"""A/B timing of library variants (tools/build_variant.sh) on the C4 bench batch, one process per variant.

usage: python tools/ab_variants.py tag [tag ...]      (csrc/libmmpc_<tag>.so; the tag `ship` is csrc/libmmpc.so)
Per variant: ms per 8192 distinct instances in plain batch order and by the a-priori key (what bench.py times), ms per 1024 copies
of instance 0 (the per-iteration cost without the drain), iteration statistics and a checksum of X to compare the variants' results.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
import torch, mmpc_loader
from oracle import synth
mm = mmpc_loader.load()
N, M = 20, 5
seed = int(os.environ.get("MMPC_AB_SEED", 3))
d = synth.make_batch(8192, N=N, M=M, seed=seed) if seed != 3 else synth.make_batch(8192, N=N, M=M)
dev = torch.device("cuda", 0)
def run(B, same=None, order=0, reps=6):
    ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M)
    eng = ctrl._engine
    idx = np.arange(B) %% 8192 if same is None else np.full(B, same)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx])).to(dev)
    xi = t(np.clip(d["x_init"], ctrl.xlim[0], ctrl.xlim[1])); tr = t(d["traj_ref"]); ur = t(d["u_ref"]); ob = t(d["obs"])
    ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
    kw = {}
    eng.set_schedule_hint(order)   # 0 batch order, 1 longest-first by the previous launch's iteration counts (exact), 2 a-priori key
    out = eng.solve_batch_device(xi, tr, ur, ul, ob)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); eng.solve_batch_device(xi, tr, ur, ul, ob, out=out, **kw); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    it = out["iters"].cpu().numpy()
    return min(ts), it, float(out["X"].double().sum().item()), float((out["status"] == 0).double().mean().item())
a, ita, _, _ = run(1024, same=0)
b, itb, cs, conv = run(8192, order=0)
c, _, _, _ = run(8192, order=2)
e, _, _, _ = run(8192, order=1)
print("%%s %%.4f %%d %%.4f %%.4f %%.4f %%.3f %%d %%.4f %%.9f" %% (os.environ.get("MMPC_AB_TAG", "?"), a, ita[0], b, c, e, itb.mean(), itb.max(), conv, cs), flush=True)
''' % ROOT

ROUNDS = int(os.environ.get("MMPC_AB_ROUNDS", 3))
tags = sys.argv[1:]
res = {t: [] for t in tags}
for r in range(ROUNDS):          # the variants interleaved, the best of the rounds per figure (run-to-run clock / power state: 1-2 %)
    for tag in tags:
        lib = os.path.join(ROOT, "mobile-manipulator-mpc_amd", "csrc", "libmmpc.so" if tag == "ship" else "libmmpc_%s.so" % tag)
        env = dict(os.environ, MMPC_LIB=lib, MMPC_AB_TAG=tag)
        p = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            print("%s FAILED\n%s" % (tag, p.stderr[-2000:]), flush=True); continue
        res[tag].append(p.stdout.split())
for tag in tags:
    v = res[tag]
    if not v: continue
    m = lambda i: min(float(x[i]) for x in v)
    print("%-8s  1024 x inst 0: %7.3f ms (%s it, %.2f us/it)   8192: batch order %7.3f  key %7.3f  exact %7.3f ms   iters mean %s max %s conv %s  sum X %s" % (
        tag, m(1), v[0][2], m(1) * 1e3 / int(v[0][2]), m(3), m(4), m(5), v[0][6], v[0][7], v[0][8], v[0][9]), flush=True)
