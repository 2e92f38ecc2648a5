"""Opt-in warm start of include/mmpc.h (mmpc_set_warm_start): initial U separate from the U_last parameter, initial X, initial
barrier parameter.  The NLP is unchanged, so the minimiser is the one the reference's protocol reaches - in far fewer
iterations when the guess is the previous optimum shifted by one stage and its roll-out."""
import numpy as np
import pytest

from oracle import synth, coracle, nlp, ipm_numpy
from tests import emu_helper


def _tick1(B, N=20, M=5, cid=3):
    """tick 0 solved cold by the C oracle; returns the tick-1 problem data with the shifted guess and its roll-out"""
    par = nlp.WholeBodyParams(N=N)
    d = synth.make_batch(B, N=N, M=M, config_id=cid)
    x0 = np.clip(d["x_init"], par.xlim[0], par.xlim[1])
    r0 = coracle.solve_batch(par, x0, d["traj_ref"], d["u_ref"], np.zeros((B, N, 5)), d["obs"], nthreads=8)
    assert (r0["status"] == 0).all()
    U = r0["U"]
    x1 = np.clip(np.array([nlp.wholebody_f(x0[b], U[b, 0], par.dt) for b in range(B)]), par.xlim[0], par.xlim[1])
    tr1 = np.concatenate([d["traj_ref"][:, 1:], d["traj_ref"][:, -1:]], axis=1)
    Ug = np.concatenate([U[:, 1:], U[:, -1:]], axis=1)
    Xg = np.zeros((B, N + 1, 9)); Xg[:, 0] = x1
    for k in range(N):
        Xg[:, k + 1] = np.array([nlp.wholebody_f(Xg[b, k], Ug[b, k], par.dt) for b in range(B)])
    return par, d, x1, tr1, U, Ug, Xg


def test_c_oracle_and_emulator_agree_and_need_fewer_iterations():
    B = 96   # (the share of instances that end in the reference protocol's minimum: 0.97-0.99 at 96 and 256, 21 of 24 on the first 24)
    par, d, x1, tr1, U, Ug, Xg = _tick1(B)
    ref = coracle.solve_batch(par, x1, tr1, d["u_ref"], U, d["obs"], nthreads=8)                       # reference protocol
    o = coracle.solve_batch(par, x1, tr1, d["u_ref"], U, d["obs"], X0=Xg, U0=Ug, nthreads=8, mu_init=0.1)
    assert (ref["status"] == 0).all() and (o["status"] == 0).all()
    same = np.abs(o["cost"] / ref["cost"] - 1) < 1e-6
    assert same.mean() >= 0.9 and np.abs(o["X"][same] - ref["X"][same]).max() < 1e-5
    assert o["iters"].mean() < 0.7 * ref["iters"].mean()
    e = emu_helper.solve_batch(par, x1, tr1, d["u_ref"], U, d["obs"], x_guess=Xg, u_guess=Ug, fast=True, mu_init=0.1)
    assert (e["status"] == 0).all() and (e["iters"] == o["iters"]).mean() > 0.9
    assert np.abs(e["X"] - o["X"]).max() < 1e-6 and np.abs(e["U"] - o["U"]).max() < 1e-5   # (weakly determined inputs, as in the GPU test below)
    # numpy restatement, one instance
    prob = nlp.Problem(par, x1[0], tr1[0], d["u_ref"][0], U[0], d["obs"][0])
    opt = ipm_numpy.Options(); opt.mu_init = 0.1
    q = ipm_numpy.solve(prob, U0=Ug[0], X0=Xg[0], opt=opt)
    assert q["status"] == 0 and np.abs(q["X"] - o["X"][0]).max() < 1e-6


@pytest.mark.gpu
def test_gpu_warm_start_option(mm):
    import torch
    B = 256
    par, d, x1, tr1, U, Ug, Xg = _tick1(B)
    o = coracle.solve_batch(par, x1, tr1, d["u_ref"], U, d["obs"], X0=Xg, U0=Ug, nthreads=8, mu_init=0.1)
    ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=20, max_batch=B, n_obstacles=5)
    eng = ctrl._engine
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ug = t(Ug)
    eng.set_warm_start(ug, 0.1)
    r = eng.solve_batch_device(t(x1), t(tr1), t(d["u_ref"]), t(U), t(d["obs"]), x_guess=t(Xg))
    assert bool((r["status"] == 0).all())
    assert (r["iters"].cpu().numpy() == o["iters"]).mean() > 0.9
    assert np.abs(r["X"].cpu().numpy() - o["X"]).max() < 1e-6 and np.abs(r["U"].cpu().numpy() - o["U"]).max() < 1e-5   # (weakly determined inputs: both ends are KKT <= 1e-8 points)
    # back to the reference's protocol: same handle, same inputs, cold barrier parameter, U starts at U_last
    eng.set_warm_start(None, 1.0)
    ref = eng.solve_batch_device(t(x1), t(tr1), t(d["u_ref"]), t(U), t(d["obs"]))
    oref = coracle.solve_batch(par, x1, tr1, d["u_ref"], U, d["obs"], nthreads=8)
    assert np.abs(ref["X"].cpu().numpy() - oref["X"]).max() < 1e-6
    assert float(r["iters"].double().mean()) < 0.7 * float(ref["iters"].double().mean())
