"""Generates tests/golden/slsqp_solutions.npz: solve() outputs of an INDEPENDENT solver for fixed instances.

TEST INFRASTRUCTURE ONLY (build container; ~5 min).  The reference's own solver (CasADi/IPOPT, requirements.txt:9)
is not installable here and the reference holds no solve() vectors, so the second source is scipy's SLSQP
(oracle/xcheck.py: an active-set SQP - neither IPOPT's algorithm nor the engine's) on the NLP restated in
oracle/nlp.py from controllers/mpc_wholebody_qref.py:142-285 / mpc_base.py:114-189, started the way the reference
starts IPOPT (X = tile(x_init), U = U_last, s = 0; mpc_wholebody_qref.py:301-304) and - since round 3 - a second time from
another start (U = a constant turn, START2), to see how far the second source's own answer depends on the start.  The
instances are FIXED IN ADVANCE (the first ones of each seeded configuration: C3 x32, C5 x16, C2 x8, terminal-xy x8, the demo
scenario), not chosen by outcome; every record keeps SLSQP's own success flag and the certificate
nlp.kkt_certificate_ipopt of its point for BOTH starts, and tests/ state what they require of each.
Usage:  python -m oracle.gen_slsqp_golden [workers]      (build container only; about ten minutes on six cores)
"""
import os
import sys
import time

import numpy as np

from . import nlp, synth, xcheck

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "slsqp_solutions.npz")


def cases():
    out = []
    d = synth.make_batch(8)                                   # C3/C4 shape: whole-body N=20, M=5, cold start
    par = nlp.WholeBodyParams()
    d = synth.make_batch(32)
    for b in range(32):
        out.append(("c3_%d" % b, par, d["x_init"][b], d["traj_ref"][b], d["u_ref"][b], np.zeros((20, 5)), d["obs"][b], None))
    d = synth.make_batch(8, N=15, M=3, kind="base", config_id=2)     # C2 shape: base-only N=15, M=3
    parb = nlp.BaseParams(N=15)
    for b in range(8):
        out.append(("c2_%d" % b, parb, d["x_init"][b], d["traj_ref"][b], d["u_ref"][b], np.zeros((15, 2)), d["obs"][b], None))
    # base-only with the heading term switched on across the +-pi cut (mpc_base.py:146-150,162-166)
    parh = nlp.BaseParams(N=15)
    parh.Q = np.diag([5., 5., 2.0, 0, 0, 1.]); parh.P = np.diag([5., 5., 2.0, 0, 0, 1.])
    x0 = d["x_init"][3].copy(); x0[2] = 3.0
    tr = d["traj_ref"][3].copy(); tr[:, 2] = np.linspace(3.0, 3.6, 16)          # reference heading runs through pi
    out.append(("c2_heading", parh, x0, tr, d["u_ref"][3], np.zeros((15, 2)), d["obs"][3], None))
    d = synth.make_batch(16, N=30, M=8, config_id=5, moving=True)      # C5 shape: N=30, 8 moving obstacles (per-stage centres)
    par5 = nlp.WholeBodyParams(N=30)
    for b in range(16):
        obs = np.zeros((31, 8, 3))
        for k in range(31):
            obs[k, :, :2] = d["obs"][b, :, :2] + d["obs_vel"][b] * k * 0.1
            obs[k, :, 2] = d["obs"][b, :, 2]
        out.append(("c5_%d" % b, par5, d["x_init"][b], d["traj_ref"][b], d["u_ref"][b], np.zeros((30, 5)), obs, None))
    # C1: demo_wholebody_qref.py scenario 2 (:28-33,40-44), first tick, half-space rows in the intended form
    r2 = 1 / np.sqrt(2)
    hs = np.array([[2.5, 2, 0.35 + 0.606 + 0.333, r2, 0, r2], [2.5, 2, 0.35 + 0.606 + 0.333, -r2, 0, r2]])
    obs1 = np.array([[2.5, 3.0, 0.6], [2.5, 1.0, 0.6], [4.4, 5, 0.1]])
    par1 = nlp.WholeBodyParams()
    x_start = np.zeros(9)
    traj = np.linspace(x_start, np.array([5, 5, -np.pi, 0, 0, 0, 0, 0, 0.0]), 51)[:21]
    out.append(("c1_demo", par1, x_start, traj, np.zeros((20, 5)), np.zeros((20, 5)), obs1, hs))
    xs2 = np.array([1.9, 2.0, 0.0, 0.3, 0, 0, 0.3, -1.2, 1.6])               # under the "tent" of the two planes
    traj2 = np.linspace(xs2, np.array([3.2, 2.0, 0, 0, 0, 0, 0.3, -1.2, 1.6]), 51)[:21]
    out.append(("c1_tent", par1, xs2, traj2, np.zeros((20, 5)), np.zeros((20, 5)), obs1, hs))
    # terminal-xy equality of the 'approach' phase (interface_wholebody_qref.py:166-167)
    d = synth.make_batch(8)
    part = nlp.WholeBodyParams(); part.terminal_xy_equality = True
    for k in range(8):
        b = (5 + k) % 8
        xi = nlp.clip_x_init(part, d["x_init"][b])
        tr = d["traj_ref"][b].copy(); tr[:, :2] = xi[None, :2] + 0.5 * (tr[:, :2] - xi[None, :2])
        out.append(("txy_%d" % k, part, d["x_init"][b], tr, d["u_ref"][b], np.zeros((20, 5)), d["obs"][b], None))
    return out


# second start of the second source: a constant gentle turn (the survey's probe start), zero arm rates
START2 = (1.0, -1.5, 0.0, 0.0, 0.0)


def run_slsqp(prob, par, U0):
    """SLSQP from U0; when it gives up in its line search (the S s^2 term, S = 1e5, makes the QP subproblems ill-conditioned) a
    second run on the same NLP after the change of variables s = sigma / sqrt(S); the run with the smaller certificate is
    kept, whatever minimiser it ends in."""
    q = xcheck.solve_slsqp(prob, U0=U0, maxiter=1000)
    c = nlp.kkt_certificate_ipopt(prob, q["X"], q["U"], q["s"])
    form = 0
    if c["E0"] > 1e-4:
        q2 = xcheck.solve_slsqp(prob, U0=U0, maxiter=1000, slack_scale=np.sqrt(float(np.ravel(par.S)[0])))
        c2 = nlp.kkt_certificate_ipopt(prob, q2["X"], q2["U"], q2["s"])
        if c2["E0"] < c["E0"]:
            q, c, form = q2, c2, 1
    return q, c, form


def solve_case(case):
    name, par, x_init, traj, uref, ulast, obs, hs = case
    t0 = time.time()
    prob = nlp.Problem(par, nlp.clip_x_init(par, x_init), traj, uref, ulast, obs, hs)
    q, c, form = run_slsqp(prob, par, ulast)
    U2 = np.tile(np.array(START2[:par.nu]), (par.N, 1))
    q2, c2, form2 = run_slsqp(prob, par, U2)
    rec = {}
    rec[name + "/kind"] = np.array(0 if par.kind == "wholebody" else 1)
    rec[name + "/N"] = np.array(par.N)
    rec[name + "/Q"] = par.Q; rec[name + "/P"] = par.P
    rec[name + "/terminal_xy"] = np.array(int(par.terminal_xy_equality))
    rec[name + "/x_init"] = np.asarray(x_init, float); rec[name + "/traj_ref"] = traj; rec[name + "/u_ref"] = uref
    rec[name + "/u_last"] = ulast; rec[name + "/obs"] = obs
    rec[name + "/hs"] = hs if hs is not None else np.zeros((0, 6))
    # (the second start's trajectory is only kept when it ended somewhere else; otherwise its distance to the first one's)
    same = abs(q["cost"] - q2["cost"]) <= 1e-6 * abs(q["cost"]) and np.abs(q["X"] - q2["X"]).max() <= 1e-3
    rec[name + "/same_min2"] = np.array(int(same))
    if same:
        rec[name + "/dX2"] = np.array(np.abs(q["X"] - q2["X"]).max()); rec[name + "/dU2"] = np.array(np.abs(q["U"] - q2["U"]).max())
    for tag, qq, cc, ff in (("", q, c, form), ("2", q2, c2, form2)):
        if not (tag == "2" and same):
            rec[name + "/X" + tag] = qq["X"]; rec[name + "/U" + tag] = qq["U"]; rec[name + "/s" + tag] = qq["s"]
        rec[name + "/cost" + tag] = np.array(qq["cost"]); rec[name + "/slsqp_success" + tag] = np.array(int(qq["success"]))
        rec[name + "/slsqp_iters" + tag] = np.array(qq["iters"]); rec[name + "/cert_E0" + tag] = np.array(cc["E0"])
        rec[name + "/slack_scaled" + tag] = np.array(ff)
    line = "%-10s start 1: iters %4d ok %d cost %.9g cert %.2e%s | start 2: iters %4d ok %d cost %.9g cert %.2e%s  (%.0f s)" % (
        name, q["iters"], q["success"], q["cost"], c["E0"], " [scaled]" if form else "", q2["iters"], q2["success"], q2["cost"], c2["E0"],
        " [scaled]" if form2 else "", time.time() - t0)
    return name, rec, line


def main():
    import multiprocessing as mp
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cs = cases()
    rec = {}
    names = []
    with mp.Pool(workers) as pool:
        for name, r, line in pool.imap(solve_case, cs):
            print(line); sys.stdout.flush()
            names.append(name); rec.update(r)
    rec["names"] = np.array(names)
    np.savez_compressed(OUT, **rec)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
