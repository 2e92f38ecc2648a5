"""CPU: the HIP solver core (mobile-manipulator-mpc_amd/csrc/mmpc_core.h) compiled for the host with
-DMMPC_EMU (every phase = a loop over the 64 lanes) against the independent C oracle; run again
with the lanes of every phase in reverse order (exposes intra-phase races), and under ASAN."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import nlp, coracle, synth, ipm_numpy

TOL = 1e-6
import emu_helper


def _cmp(par, d, obs, ul, x_guess=None, X0=None, fast=False):
    r = coracle.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], ul, obs, X0=X0, nthreads=4)
    e = emu_helper.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], ul, obs, x_guess=x_guess, fast=fast)
    e2 = emu_helper.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], ul, obs, x_guess=x_guess, reverse=True, fast=fast)
    assert (r["status"] == 0).all() and (e["status"] == 0).all()
    assert (np.abs(r["iters"] - e["iters"]) <= 2).mean() > 0.8
    assert np.abs(r["X"] - e["X"]).max() < 1e-6 and np.abs(r["U"] - e["U"]).max() < 1e-6
    assert np.abs(r["s"] - e["s"]).max() < 1e-8
    assert np.array_equal(e["X"], e2["X"]) and np.array_equal(e["U"], e2["U"]) and np.array_equal(e["iters"], e2["iters"])
    return r, e


@pytest.mark.parametrize("fast", [False, True])
def test_emu_wholebody_c3(fast):
    """generic kernel (mmpc_core.h) and specialised kernel (mmpc_fast.h) on config C3"""
    B = 24
    d = synth.make_batch(B)
    _cmp(nlp.WholeBodyParams(), d, d["obs"], np.zeros((B, 20, 5)), fast=fast)


@pytest.mark.parametrize("fast", [False, True])
def test_emu_base_c2_and_warm_start(fast):
    B = 16
    d = synth.make_batch(B, N=15, M=3, kind="base", config_id=2)
    par = nlp.BaseParams(N=15)
    r, _ = _cmp(par, d, d["obs"], np.zeros((B, 15, 2)), fast=fast)
    d2 = dict(d)
    d2["x_init"] = np.array([coracle.f("base", 0.1, d["x_init"][b], r["U"][b, 0]) for b in range(B)])
    _cmp(par, d2, d["obs"], r["U"], x_guess=r["X"], X0=r["X"], fast=fast)     # mpc_base.py:200-201


@pytest.mark.parametrize("fast", [False, True])
def test_emu_wholebody_warm_tick_and_demo_obstacle_count(fast):
    """second tick (U init = U_last = previous optimum) and the demo's M=3 (demo_wholebody_qref.py:40-44)"""
    B = 8
    par = nlp.WholeBodyParams()
    d = synth.make_batch(B)
    r = coracle.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], np.zeros((B, 20, 5)), d["obs"])
    d2 = dict(d)
    d2["x_init"] = np.array([coracle.f("wholebody", 0.1, np.clip(d["x_init"][b], par.xlim[0], par.xlim[1]), r["U"][b, 0]) for b in range(B)])
    _cmp(par, d2, d["obs"], r["U"], fast=fast)
    d3 = synth.make_batch(B, N=20, M=3, config_id=9)
    _cmp(par, d3, d3["obs"], np.zeros((B, 20, 5)), fast=fast)


@pytest.mark.parametrize("fast", [False, True])
def test_emu_wholebody_c5_moving_obstacles(fast):
    B = 6
    d = synth.make_batch(B, N=30, M=8, config_id=5, moving=True)
    obs = np.zeros((B, 31, 8, 3))
    for k in range(31):
        obs[:, k, :, :2] = d["obs"][:, :, :2] + d["obs_vel"] * k * 0.1
        obs[:, k, :, 2] = d["obs"][:, :, 2]
    _cmp(nlp.WholeBodyParams(N=30), d, obs, np.zeros((B, 30, 5)), fast=fast)


def test_emu_terminal_xy_equality():
    """X[N,:2] == X_ref[N,:2] (interface_wholebody_qref.py:166-167), generic kernel vs oracle; the terminal point is
    placed at half the synthetic range so that it is reachable within the horizon."""
    B = 6
    d = synth.make_batch(B)
    par = nlp.WholeBodyParams()
    par.terminal_xy_equality = True
    x0 = np.clip(d["x_init"], par.xlim[0], par.xlim[1])
    d["traj_ref"][:, :, :2] = x0[:, None, :2] + 0.5 * (d["traj_ref"][:, :, :2] - x0[:, None, :2])
    r, e = _cmp(par, d, d["obs"], np.zeros((B, 20, 5)))
    assert np.abs(e["X"][:, 20, :2] - d["traj_ref"][:, 20, :2]).max() < 1e-9
    for b in range(2):
        prob = nlp.Problem(par, x0[b], d["traj_ref"][b], d["u_ref"][b], np.zeros((20, 5)), d["obs"][b])
        c = nlp.kkt_certificate(prob, e["X"][b], e["U"][b], e["s"][b])
        assert c["eq_violation"] < 1e-9 and c["stationarity_rel"] < 1e-6


def _c1_inputs():
    """demo_wholebody_qref.py:27-44 (scenario 2): x_start = 0, target (5,5,-pi), 3 circles, 2 half-spaces; plus a
    start next to the block with the arm raised so that the half-space rows are active."""
    r2 = 1 / np.sqrt(2)
    hs = np.array([[2.5, 2, 0.35 + 0.606 + 0.333, r2, 0, r2], [2.5, 2, 0.35 + 0.606 + 0.333, -r2, 0, r2]])
    obs = np.array([[2.5, 3.0, 0.6], [2.5, 1.0, 0.6], [5 - 0.6, 5, 0.1]])
    x0 = np.zeros(9)
    x1 = np.array([1.9, 2.0, 0.0, 0.3, 0, 0, 0.3, -1.2, 1.6])
    g0 = np.linspace(x0, np.array([5, 5, -np.pi, 0, 0, 0, 0, 0, 0.0]), 51)[:21]
    g1 = np.linspace(x1, np.array([3.2, 2.0, 0, 0, 0, 0, 0.3, -1.2, 1.6]), 51)[:21]
    return dict(x_init=np.stack([x0, x1]), traj_ref=np.stack([g0, g1]), u_ref=np.zeros((2, 20, 5)),
                obs=np.stack([obs, obs])), hs


def test_emu_halfspace_obstacles_c1():
    """config C1 geometry with the intended half-space rows (a-9; quirk Q8 not reproduced, see nlp.halfspace_row)"""
    d, hs = _c1_inputs()
    par = nlp.WholeBodyParams()
    r = coracle.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], np.zeros((2, 20, 5)), d["obs"], hs=hs)
    e = emu_helper.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], np.zeros((2, 20, 5)), d["obs"], hs=hs)
    assert (r["status"] == 0).all() and (e["status"] == 0).all() and (r["iters"] == e["iters"]).all()
    assert np.abs(r["X"] - e["X"]).max() < 1e-9 and np.abs(r["U"] - e["U"]).max() < 1e-9
    assert r["s"][1].max() > 1e-3          # the half-space rows bite in the second instance
    for b in range(2):
        prob = nlp.Problem(par, d["x_init"][b], d["traj_ref"][b], d["u_ref"][b], np.zeros((20, 5)), d["obs"][b], hs)
        c = nlp.kkt_certificate(prob, e["X"][b], e["U"][b], e["s"][b])
        assert c["eq_violation"] < 1e-9 and c["ineq_violation"] < 1e-9 and c["stationarity_rel"] < 2e-5
    # rows evaluated literally (world points from the FK, max over planes) agree with the closed form
    rng = np.random.default_rng(5)
    for _ in range(20):
        x = rng.uniform(-2, 2, 9)
        for i in range(6):
            assert abs(nlp.halfspace_row(x, i, hs, order=0) - nlp.halfspace_row_direct(x, i, hs)) < 1e-14


def test_emu_hard_instances_converge():
    """instances of the 8192 batch that need the filter-reset heuristic / many iterations"""
    d = synth.make_batch(8192)
    idx = np.array([1363, 1644, 3269])
    dd = {k: v[idx] for k, v in d.items()}
    _cmp(nlp.WholeBodyParams(), dd, dd["obs"], np.zeros((3, 20, 5)), fast=True)


def test_emu_edge_sizes():
    """M=0 (no obstacles), N=1, and N=63 (largest supported horizon: 64 stages = one per lane)."""
    for N, M in ((1, 0), (3, 1), (63, 2)):
        B = 2
        d = synth.make_batch(B, N=N, M=max(M, 1), config_id=7)
        obs = d["obs"][:, :M]
        _cmp(nlp.WholeBodyParams(N=N), d, obs, np.zeros((B, N, 5)))


def test_emu_under_asan():
    code = ("import sys; sys.path[:0]=[%r,%r]\n"
            "import numpy as np, emu_helper\n"
            "from oracle import nlp, synth\n"
            "d=synth.make_batch(3)\n"
            "e=emu_helper.solve_batch(nlp.WholeBodyParams(),d['x_init'],d['traj_ref'],d['u_ref'],np.zeros((3,20,5)),d['obs'],asan=True)\n"
            "assert (e['status']==0).all()\n"
            "e=emu_helper.solve_batch(nlp.WholeBodyParams(),d['x_init'],d['traj_ref'],d['u_ref'],np.zeros((3,20,5)),d['obs'],asan=True,fast=True)\n"
            "assert (e['status']==0).all()\n"
            "d=synth.make_batch(2,N=15,M=3,kind='base')\n"
            "e=emu_helper.solve_batch(nlp.BaseParams(N=15),d['x_init'],d['traj_ref'],d['u_ref'],np.zeros((2,15,2)),d['obs'],asan=True)\n"
            "assert (e['status']==0).all()\n"
            # the second-order correction's passes (its scratch is an exact-size heap block here, global memory on the device): two of the
            # round-3 tail cases on both kernels, N = 30 with the gains in their global block
            "import os; z=np.load(os.path.join(%r,'golden','c5_tail_cases.npz')); p30=nlp.WholeBodyParams(N=30)\n"
            "for f in (False, True):\n"
            "    e=emu_helper.solve_batch(p30,z['x'][:2],z['loc'][:2],np.zeros((2,30,5)),z['ul'][:2],z['obs'][:2],asan=True,fast=f,max_iter=2000)\n"
            "    assert (e['status']==0).all() and e['iters'].max()<=60\n"
            "print('ASAN-OK')\n") % (emu_helper._HERE + "/..", emu_helper._HERE, emu_helper._HERE)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ASAN-OK" in p.stdout, p.stderr[-3000:]


def _pose_batch(B, N, seed=5):
    rng = np.random.default_rng(seed)
    x0 = np.zeros((B, 9)); ref = np.zeros((B, N + 1, 4)); obs = np.zeros((B, 2, 3))
    for b in range(B):
        x0[b] = [0, 0, rng.uniform(-1, 1), rng.uniform(0, 0.5), 0, 0, rng.uniform(-0.5, 0.5), rng.uniform(-2, -0.3), rng.uniform(0.3, 2.5)]
        E0 = nlp.endpoint_pose(x0[b])
        tgt = E0 + np.array([rng.uniform(0.5, 2), rng.uniform(-1, 1), rng.uniform(-0.2, 0.2), rng.uniform(-0.5, 0.5)])
        ref[b] = np.linspace(E0, tgt, N + 1)
        obs[b] = [[E0[0] + 0.5 * (tgt[0] - E0[0]), E0[1] + 0.5 * (tgt[1] - E0[1]) + 0.3, 0.3], [3, 3, 0.2]]
    return x0, ref, obs


def test_emu_pose_reference_controller():
    """controllers/mpc_wholebody.py (endpoint-pose reference, KIND 2 of the generic kernel) vs the C and numpy oracles,
    first solve and a warm-started second one (X and U initial guesses = previous optimum, :134-139)."""
    N, B = 10, 6
    par = nlp.pose_ref_params(N=N)
    x0, ref, obs = _pose_batch(B, N)
    ul = np.zeros((B, N, 5)); ur = np.zeros((B, N, 5))
    o = coracle.solve_batch(par, x0, ref, ur, ul, obs)
    e = emu_helper.solve_batch(par, x0, ref, ur, ul, obs)
    er = emu_helper.solve_batch(par, x0, ref, ur, ul, obs, reverse=True)
    assert (o["status"] == 0).all() and (e["status"] == 0).all()
    assert np.abs(e["X"] - o["X"]).max() < TOL and np.abs(e["U"] - o["U"]).max() < TOL
    assert np.array_equal(e["X"], er["X"])
    prob = nlp.Problem(par, x0[0], ref[0], ur[0], ul[0], obs[0])
    q = ipm_numpy.solve(prob)
    assert np.abs(q["X"] - e["X"][0]).max() < TOL
    k = nlp.kkt_certificate(prob, e["X"][0], e["U"][0], e["s"][0])
    assert k["stationarity_rel"] < 1e-6 and k["eq_violation"] < 1e-9 and k["ineq_violation"] < 1e-9
    # second tick: plant step, shifted reference, warm start of X and U
    x1 = np.array([nlp.f_dyn("wholebody", x0[b], e["U"][b, 0], par.dt) for b in range(B)])
    ref1 = np.concatenate([ref[:, 1:], ref[:, -1:]], axis=1)
    o2 = coracle.solve_batch(par, x1, ref1, ur, e["U"], obs, X0=e["X"])
    e2 = emu_helper.solve_batch(par, x1, ref1, ur, e["U"], obs, x_guess=e["X"])
    assert (e2["status"] == 0).all() and np.abs(e2["X"] - o2["X"]).max() < TOL


@pytest.mark.parametrize("fast", [False, True])
def test_emu_custom_weights_and_limits(fast):
    """diagonal weights other than the defaults and tightened input / rate / velocity limits (rows that bind)"""
    B = 8
    d = synth.make_batch(B, config_id=21)
    rng = np.random.default_rng(21)
    par = nlp.WholeBodyParams()
    par.ulim = np.array([[-1.5, -2.0, -0.8, -0.8, -0.8], [1.5, 2.0, 0.8, 0.8, 0.8]])
    par.dulim = np.array([[-1.0, -np.inf, -0.3, -0.3, -0.3], [1.0, np.inf, 0.3, 0.3, 0.3]])
    par.xlim = par.xlim.copy(); par.xlim[:, 3:5] = [[-1.2, -1.2], [1.2, 1.2]]
    q = rng.uniform(0.5, 30, 9) * (rng.uniform(size=9) > 0.3)
    par.Q, par.P = np.diag(q), np.diag(q * rng.uniform(1, 3, 9))
    par.R, par.W, par.S = np.diag(rng.uniform(0.05, 0.5, 5)), np.diag(rng.uniform(0.05, 0.5, 5)), 3e4
    d["x_init"] = np.clip(d["x_init"], par.xlim[0], par.xlim[1])
    r, e = _cmp(par, d, d["obs"], np.zeros((B, 20, 5)), fast=fast)
    assert (np.abs(e["U"][:, :, 0]) > 1.0 - 1e-6).any()      # the merged box min(ulim, u_last + dulim) = 1.0 binds


@pytest.mark.parametrize("fast", [False, True])
def test_second_order_correction_on_the_round3_tail(fast):
    """Five C5 solves (N = 30, 8 moving obstacles, warm start per mpc_wholebody_qref.py:301-310) recorded on the GPU in round 3 -
    the two that stopped at the reference's iteration cap and three other stragglers of that run (tests/golden/c5_tail_cases.npz:
    inputs, and the iteration counts of the round-3 solver on the C oracle: 2966, 2132, 32, 33, 56): near-degenerate minima beside
    an active circle row, where a step the linearised row allows violates it to second order.  With the second-order correction (and the inertia correction) they take 27-46 iterations on the C oracle, and the
    host builds of both kernels - the specialised one runs its corrected passes in the second copy of its loop, with the
    gains in the global block of the long horizons - follow it iteration for iteration."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "c5_tail_cases.npz"))
    par = nlp.WholeBodyParams(N=30)
    assert (z["round3_iters"][:2] > 2000).all()
    o = coracle.solve_batch(par, z["x"], z["loc"], np.zeros_like(z["ul"]), z["ul"], z["obs"], max_iter=2000)
    assert (o["status"] == 0).all() and o["iters"].max() <= 60, o["iters"]
    e = emu_helper.solve_batch(par, z["x"], z["loc"], np.zeros_like(z["ul"]), z["ul"], z["obs"], fast=fast, max_iter=2000)
    assert (e["status"] == 0).all() and np.array_equal(e["iters"], o["iters"]), (e["iters"], o["iters"])
    assert np.abs(e["X"] - o["X"]).max() < TOL and np.abs(e["U"] - o["U"]).max() < 1e-5
