"""Which minimum?  For the second-source fixtures whose engine minimum costs more than the best SLSQP run (tests/test_slsqp_golden.py:
EXPECTED_COSTLIER): cost ratio; on which side of every nearby obstacle the two trajectories pass; whether the engine, started AT the
second source's point (the opt-in warm start: X, U guesses, small mu_0), stays in the cheaper basin; and the first iteration of the
engine's own path from the reference's start (X = tile(x_init), U = U_last) at which its passing sides are those of its final
trajectory for good.  CPU only (numpy / C oracle); writes profiles/r04_which_minimum.txt."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nlp, coracle, ipm_numpy
import test_slsqp_golden as T


def sides(X, obs):
    """+1 / -1: the obstacle is left / right of the path at the closest approach; 0: farther than 0.75 m from its inflated disc"""
    obs = np.asarray(obs).reshape(-1, 3) if np.asarray(obs).ndim == 2 else None
    out = []
    for m in range(0 if obs is None else len(obs)):
        d = np.linalg.norm(X[:, :2] - obs[m, :2], axis=1)
        k = int(np.argmin(d)); k = min(max(k, 0), len(X) - 2)
        if d.min() > obs[m, 2] + 0.4 + 0.75:
            out.append(0); continue
        t = X[k + 1, :2] - X[k, :2]; r = obs[m, :2] - X[k, :2]
        out.append(int(np.sign(t[0] * r[1] - t[1] * r[0])))
    return tuple(out)


lines = []
for name, par, g in T.load_cases():
    if name not in T.EXPECTED_COSTLIER:
        continue
    hs = g["hs"] if len(g["hs"]) else None
    x0 = nlp.clip_x_init(par, g["x_init"])
    obs = g["obs"]
    static = obs.ndim == 2
    o = coracle.solve_batch(par, x0[None], g["traj_ref"][None], g["u_ref"][None], g["u_last"][None], obs[None], hs=hs, max_iter=2000)
    runs = [(float(g["cost"]), g["X"], g["U"])]
    if not int(g["same_min2"]):
        runs.append((float(g["cost2"]), g["X2"], g["U2"]))
    cb, Xb, Ub = min(runs, key=lambda r: r[0])
    ce = float(o["cost"][0])
    stay = []
    for mu0 in (1.0, 1e-2, 1e-4):
        w = coracle.solve_batch(par, x0[None], g["traj_ref"][None], g["u_ref"][None], g["u_last"][None], obs[None], X0=Xb[None], U0=Ub[None],
                                hs=hs, max_iter=2000, mu_init=mu0)
        stay.append("%s(%d it)" % ("stays" if abs(w["cost"][0] / cb - 1) < 1e-5 else "leaves->%.1f" % w["cost"][0], w["iters"][0]))
    part = "-"
    if static and par.kind == "wholebody" and not par.terminal_xy_equality:
        prob = nlp.Problem(par, x0, g["traj_ref"], g["u_ref"], g["u_last"], obs, hs)
        hist = []
        opt = ipm_numpy.Options(); opt.max_iter = 2000
        q = ipm_numpy.solve(prob, U0=g["u_last"], opt=opt, history=hist)
        fin = sides(q["X"], obs)
        sig = [sides(h["X"], obs) for h in hist]
        k = len(sig)
        while k > 0 and sig[k - 1] == fin:
            k -= 1
        part = "it %d of %d (final sides %s, second source %s)" % (k, q["iters"], fin, sides(Xb, obs))
    lines.append("%-8s engine %.1f  best second-source run %.1f  ratio %.2f | warm start at the second source's point, mu0 = 1 / 1e-2 / 1e-4: %s | "
                 "engine's path takes its final sides from %s" % (name, ce, cb, ce / cb, " / ".join(stay), part))
    print(lines[-1], flush=True)
open(os.path.join(ROOT, "profiles", "r04_which_minimum.txt"), "w").write("\n".join(lines) + "\n")
