"""CPU: the oracle's solver (C port) against (i) the numpy reference implementation of the same
algorithm, (ii) the KKT certificate of the reference NLP, (iii) an independent SLSQP solve."""
import numpy as np
import pytest

from oracle import nlp, coracle, ipm_numpy, synth, xcheck


def _opts():
    o = ipm_numpy.Options()
    o.mu_init, o.curv_self = 1.0, False
    return o


@pytest.mark.parametrize("kind,N,M,cid", [("wholebody", 20, 5, 3), ("base", 15, 3, 2)])
def test_c_port_equals_numpy_algorithm(kind, N, M, cid):
    B = 3
    d = synth.make_batch(B, N=N, M=M, kind=kind, config_id=cid)
    par = nlp.WholeBodyParams(N=N) if kind == "wholebody" else nlp.BaseParams(N=N)
    ul = np.zeros((B, N, par.nu))
    r = coracle.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], ul, d["obs"])
    for b in range(B):
        prob = nlp.Problem(par, nlp.clip_x_init(par, d["x_init"][b]), d["traj_ref"][b], d["u_ref"][b], ul[b], d["obs"][b])
        q = ipm_numpy.solve(prob, opt=_opts())
        assert q["status"] == 0 and r["status"][b] == 0
        assert q["iters"] == r["iters"][b]
        assert np.abs(q["X"] - r["X"][b]).max() < 1e-9 and np.abs(q["U"] - r["U"][b]).max() < 1e-9


def test_kkt_certificate_of_reference_nlp():
    """Every returned point must be a KKT point of the NLP of mpc_wholebody_qref.py:142-285
    (stationarity via NNLS multipliers on the active set, feasibility to 1e-9)."""
    B = 12
    d = synth.make_batch(B)
    par = nlp.WholeBodyParams()
    ul = np.zeros((B, 20, 5))
    r = coracle.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], ul, d["obs"], nthreads=4)
    assert (r["status"] == 0).all()
    for b in range(B):
        prob = nlp.Problem(par, nlp.clip_x_init(par, d["x_init"][b]), d["traj_ref"][b], d["u_ref"][b], ul[b], d["obs"][b])
        c = nlp.kkt_certificate(prob, r["X"][b], r["U"][b], r["s"][b])
        assert c["eq_violation"] < 1e-9 and c["ineq_violation"] < 1e-9
        assert c["stationarity_rel"] < 1e-6, c
        assert abs(c["cost"] - r["cost"][b]) < 1e-9 * max(1, abs(c["cost"]))


def test_independent_slsqp_agrees():
    """Stated cross-solver tolerance: |dX|,|dU| <= 1e-4 (SLSQP's own accuracy), cost 1e-6 rel."""
    d = synth.make_batch(1)
    par = nlp.WholeBodyParams()
    ul = np.zeros((1, 20, 5))
    r = coracle.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], ul, d["obs"])
    prob = nlp.Problem(par, nlp.clip_x_init(par, d["x_init"][0]), d["traj_ref"][0], d["u_ref"][0], ul[0], d["obs"][0])
    q = xcheck.solve_slsqp(prob)
    assert abs(q["cost"] - r["cost"][0]) < 1e-6 * abs(r["cost"][0])
    assert np.abs(q["X"] - r["X"][0]).max() < 1e-4 and np.abs(q["U"] - r["U"][0]).max() < 1e-4


def test_warm_start_protocol():
    """Second tick: U init = U_last = previous optimum unshifted (mpc_wholebody_qref.py:303,310,330),
    X init = tile(x_init) (:302); must converge and respect |u - u_last| <= 0.5 on the arm inputs (:205)."""
    B = 8
    d = synth.make_batch(B)
    par = nlp.WholeBodyParams()
    r = coracle.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], np.zeros((B, 20, 5)), d["obs"])
    x1 = np.array([coracle.f("wholebody", 0.1, np.clip(d["x_init"][b], par.xlim[0], par.xlim[1]), r["U"][b, 0]) for b in range(B)])
    r2 = coracle.solve_batch(par, x1, d["traj_ref"], d["u_ref"], r["U"], d["obs"])
    assert (r2["status"] == 0).all()
    assert (np.abs(r2["U"][:, :, 2:] - r["U"][:, :, 2:]) <= 0.5 + 1e-7).all()
