"""GPU (MI355X): evidence for solve() that does not depend on the build's own algorithm.

The reference's solver (CasADi 3.6.4 -> IPOPT, requirements.txt:9) cannot run in this image and the reference holds
no solve() vectors, so the HIP outputs are checked against
  (i)  IPOPT's own termination test of the reference NLP (controllers/mpc_wholebody_qref.py:142-285, tol 1e-8 at :283),
       evaluated on the host with multipliers recovered by bounded least squares (oracle.nlp.kkt_certificate_ipopt) -
       on >= 512 instances of the bench batch including the ones that needed the most iterations, and on the base-only,
       moving-obstacle, terminal-equality and half-space shapes;
  (ii) committed solutions of an independent active-set SQP (tests/golden/slsqp_solutions.npz): 1e-4 on X, 5e-4 on U,
       1e-6 relative on the cost for the 13 fixtures where SLSQP ends at the same minimiser;
  (iii) the CPU oracle on ALL 8192 instances of the bench batch, with the tolerance that is actually met (2e-5; 1e-6 for
       all but one instance - see the test).
Everything goes through the C ABI (mmpc_amd._capi)."""
import numpy as np
import pytest

from oracle import nlp, coracle, synth
from cert_pool import certify
from test_slsqp_golden import load_cases, check_fixture, check_summary, cert_ok, CERT_TOL

pytestmark = pytest.mark.gpu

# (CERT_TOL = 1.5e-8 and cert_ok: tests/test_slsqp_golden.py - IPOPT declares convergence at E0 <= 1e-8 with ITS multipliers; the
#  certificate's multipliers are fitted independently, so its E0 is an upper bound a small factor above the engine's own figure)


def _wb(mm, N, M, B, **kw):
    return mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=N, max_batch=B, n_obstacles=M, **kw)


def _gpu_batch(mm, d, N, M, **kw):
    import torch
    B = d["x_init"].shape[0]
    par = nlp.WholeBodyParams(N=N)
    ctrl = _wb(mm, N, M, B, **kw)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xi = np.clip(d["x_init"], par.xlim[0], par.xlim[1])
    ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
    r = ctrl._engine.solve_batch_device(t(xi), t(d["traj_ref"]), t(d["u_ref"]), ul, t(d["obs"]))
    torch.cuda.synchronize()
    return par, xi, {k: v.cpu().numpy() for k, v in r.items()}


def test_bench_batch_passes_ipopt_termination_test_incl_tail(mm):
    """The 8192-instance batch bench.py times: the 32 instances with the highest iteration counts + 480 drawn at random."""
    B, N, M = 8192, 20, 5
    d = synth.make_batch(B)
    par, xi, r = _gpu_batch(mm, d, N, M)
    assert (r["status"] == 0).all()
    order = np.argsort(-r["iters"], kind="stable")
    sel = np.concatenate([order[:32], np.random.default_rng(7).choice(order[32:], 480, replace=False)])
    ul = np.zeros((N, 5))
    items = [(nlp.Problem(par, xi[b], d["traj_ref"][b], d["u_ref"][b], ul, d["obs"][b]), r["X"][b], r["U"][b], r["s"][b]) for b in sel]
    cs = certify(items)
    E0 = np.array([c["E0"] for c in cs])
    worst = int(np.argmax(E0))
    assert all(cert_ok(c) for c in cs), (int(sel[worst]), int(r["iters"][sel[worst]]), cs[worst])
    assert max(c["eq_violation"] for c in cs) <= 1e-8 and max(c["ineq_violation"] for c in cs) <= 1e-8
    cost = np.array([c["cost"] for c in cs])
    assert np.abs(cost / r["cost"][sel] - 1).max() < 1e-10       # the kernel reports the reference's cost (:317)
    print("certified %d instances: E0 max %.2e median %.2e; tail iterations %s" % (len(sel), E0.max(), np.median(E0), r["iters"][order[:8]].tolist()))


def test_full_bench_batch_against_cpu_oracle(mm):
    """All 8192 instances, GPU vs the scalar C oracle (same algorithm; different libm, summation order, and the Riccati
    factorisation runs on MFMA tiles).  Both stop at the first iterate whose scaled KKT error is <= 1e-8.  That test pins
    the minimiser only as far as the problem's curvature allows: the yaw-acceleration input carries R = 0.1 and no rate
    term (mpc_wholebody_qref.py:14-15), and the oracle ITSELF moves by 2.9e-6 (instance 6981) to 2.2e-5 (instance 2) in U
    when its tolerance is tightened from 1e-8 to 1e-11 (cost unchanged to 1e-10).  Measured on this batch: 8181 of 8192
    instances take the same number of iterations and agree to ~1e-11; ONE instance (6981: 33 iterations on the CPU, one
    more or less on the GPU) differs by 1.2e-6 in X / 9.8e-6 in U, one more by > 1e-7.  The assertion states that:
    <= 2e-5 everywhere, <= 1e-6 for all but at most 4 instances, equal cost (1e-9), and every instance above 1e-6 must
    pass the certificate of the reference NLP on its GPU output."""
    B, N, M = 8192, 20, 5
    d = synth.make_batch(B)
    par, xi, r = _gpu_batch(mm, d, N, M)
    o = coracle.solve_batch(par, xi, d["traj_ref"], d["u_ref"], np.zeros((B, N, 5)), d["obs"], nthreads=16, max_iter=2000)
    assert (r["status"] == 0).all() and (o["status"] == 0).all()
    dX = np.abs(r["X"] - o["X"]).reshape(B, -1).max(1); dU = np.abs(r["U"] - o["U"]).reshape(B, -1).max(1)
    dev = np.maximum(dX, dU)
    print("full batch: max |dX| %.3e max |dU| %.3e; > 1e-6: %s; > 1e-7: %d; equal iteration counts %.4f; max cost rel %.2e"
          % (dX.max(), dU.max(), np.nonzero(dev > 1e-6)[0].tolist(), int((dev > 1e-7).sum()), (r["iters"] == o["iters"]).mean(),
             np.abs(r["cost"] / o["cost"] - 1).max()))
    assert (r["iters"] == o["iters"]).mean() > 0.99
    assert np.abs(r["cost"] / o["cost"] - 1).max() < 1e-9          # nobody ends in another local minimum
    assert dev.max() < 2e-5
    loose = np.nonzero(dev > 1e-6)[0]
    assert len(loose) <= 4
    ul = np.zeros((N, 5))
    cs = certify([(nlp.Problem(par, xi[b], d["traj_ref"][b], d["u_ref"][b], ul, d["obs"][b]), r["X"][b], r["U"][b], r["s"][b]) for b in loose])
    for b, c in zip(loose, cs):
        assert cert_ok(c), (int(b), c)


def test_c2_base_batch_1024(mm):
    """BASELINE config C2 at its stated size: base-only, N=15, M=3, B=1024; parity with the oracle on all, certificate on 64."""
    B, N, M = 1024, 15, 3
    d = synth.make_batch(B, N=N, M=M, kind="base", config_id=2)
    par = nlp.BaseParams(N=N)
    ctrl = mm.MPCBase(mm.Base(0.1), [], N=N, max_batch=B, n_obstacles=M)
    r = ctrl.solve_batch(d["x_init"], d["traj_ref"], d["u_ref"], d["obs"])
    o = coracle.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], np.zeros((B, N, 2)), d["obs"], nthreads=16, max_iter=2000)
    assert (r["status"] == 0).all() and (o["status"] == 0).all()
    assert np.abs(r["cost"] / o["cost"] - 1).max() < 1e-8
    assert np.abs(r["X"] - o["X"]).max() < 5e-6 and np.abs(r["U"] - o["U"]).max() < 5e-6
    sel = np.concatenate([np.argsort(-r["iters"], kind="stable")[:8], np.arange(56)])
    ul = np.zeros((N, 2))
    cs = certify([(nlp.Problem(par, d["x_init"][b], d["traj_ref"][b], d["u_ref"][b], ul, d["obs"][b]), r["X"][b], r["U"][b], r["s"][b]) for b in sel])
    assert all(cert_ok(c) for c in cs), max(c["E0"] for c in cs)


def test_base_heading_term_across_the_pi_cut(mm):
    """MPCBase with Q[2,2] = P[2,2] != 0 and a heading reference that runs through +-pi: the angleDiff error
    (mpc_base.py:146-150,162-166) is exercised with a non-zero weight - specialised kernel and generic kernel."""
    B, N, M = 64, 15, 3
    d = synth.make_batch(B, N=N, M=M, kind="base", config_id=2)
    rng = np.random.default_rng(12)
    par = nlp.BaseParams(N=N)
    par.Q = np.diag([5., 5., 2.0, 0, 0, 1.]); par.P = np.diag([5., 5., 2.0, 0, 0, 1.])
    psi0 = rng.uniform(2.6, 3.6, B)                                   # both sides of pi
    d["x_init"][:, 2] = np.where(rng.uniform(size=B) < 0.5, psi0, psi0 - 2 * np.pi)   # same heading, other branch
    d["traj_ref"][:, :, 2] = psi0[:, None] + np.linspace(0, 1, N + 1)[None, :] * rng.uniform(-1.0, 1.0, B)[:, None]
    o = coracle.solve_batch(par, d["x_init"], d["traj_ref"], d["u_ref"], np.zeros((B, N, 2)), d["obs"], nthreads=16, max_iter=2000)
    assert (o["status"] == 0).all()
    for generic in (False, True):
        ctrl = mm.MPCBase(mm.Base(0.1), [], N=N, Q=par.Q, P=par.P, max_batch=B, n_obstacles=M)
        if generic:      # a dense (symmetric, still PSD) state weight is routed to the generic kernel
            Qd = par.Q.copy(); Qd[0, 1] = Qd[1, 0] = 0.5
            ctrl.setWeight(Q=Qd, P=Qd)
            pard = nlp.BaseParams(N=N); pard.Q, pard.P = Qd, Qd
            ref = coracle.solve_batch(pard, d["x_init"], d["traj_ref"], d["u_ref"], np.zeros((B, N, 2)), d["obs"], nthreads=16, max_iter=2000)
        else:
            pard, ref = par, o
        r = ctrl.solve_batch(d["x_init"], d["traj_ref"], d["u_ref"], d["obs"])
        assert (r["status"] == 0).all(), generic
        assert np.abs(r["X"] - ref["X"]).max() < 5e-6 and np.abs(r["U"] - ref["U"]).max() < 5e-6, generic
        # the wrap is live: some stage errors differ from the plain difference by 2 pi
        plain = r["X"][:, :, 2] - d["traj_ref"][:, :, 2]
        wrapped = np.vectorize(nlp.angle_diff)(r["X"][:, :, 2], d["traj_ref"][:, :, 2])
        assert (np.abs(plain - wrapped) > 6.0).any()
        ul = np.zeros((N, 2))
        cs = certify([(nlp.Problem(pard, d["x_init"][b], d["traj_ref"][b], d["u_ref"][b], ul, d["obs"][b]), r["X"][b], r["U"][b], r["s"][b]) for b in range(16)])
        assert all(cert_ok(c) for c in cs), (generic, max(c["E0"] for c in cs))


def _c5_batch(B):
    d = synth.make_batch(B, N=30, M=8, config_id=5, moving=True)
    obs = np.zeros((B, 31, 8, 3))
    for k in range(31):
        obs[:, k, :, :2] = d["obs"][:, :, :2] + d["obs_vel"] * k * 0.1
        obs[:, k, :, 2] = d["obs"][:, :, 2]
    d["obs"] = obs
    return d


def test_c5_shape_every_instance_certified(mm):
    """N=30, 8 moving obstacles (per-stage centres), cold tick.  Every instance must converge on the GPU and pass the
    certificate; instances whose minimiser differs from the oracle's (two-sided passages: implementations that differ in
    the last bit can settle on either side) are listed, must be rare and must not cost more than the oracle's by > 5 %."""
    B, N, M = 256, 30, 8
    d = _c5_batch(B)
    par, xi, r = _gpu_batch(mm, d, N, M, obs_per_stage=True)
    assert (r["status"] == 0).all(), np.nonzero(r["status"])[0].tolist()
    o = coracle.solve_batch(par, xi, d["traj_ref"], d["u_ref"], np.zeros((B, N, 5)), d["obs"], nthreads=16, max_iter=2000)
    assert (o["status"] == 0).all()
    same = np.abs(r["cost"] / o["cost"] - 1) < 1e-6
    print("C5 cold tick: other minimum than the oracle at", np.nonzero(~same)[0].tolist(), "of", B)
    assert same.mean() >= 0.97
    assert np.abs(r["X"][same] - o["X"][same]).max() < 1e-5 and np.abs(r["U"][same] - o["U"][same]).max() < 1e-5
    assert (r["cost"][~same] <= 1.05 * o["cost"][~same]).all()
    sel = np.unique(np.concatenate([np.nonzero(~same)[0], np.argsort(-r["iters"], kind="stable")[:16], np.arange(32)]))
    ul = np.zeros((N, 5))
    cs = certify([(nlp.Problem(par, xi[b], d["traj_ref"][b], d["u_ref"][b], ul, d["obs"][b]), r["X"][b], r["U"][b], r["s"][b]) for b in sel])
    E0 = np.array([c["E0"] for c in cs])
    assert all(cert_ok(c) for c in cs), (int(sel[int(np.argmax(E0))]), E0.max())


def test_c5_full_size_properties(mm):
    """C5 at its stated batch: B=8192, N=30, 8 moving obstacles, 3 warm-started receding-horizon ticks through the batched
    closed-loop driver: every solve converges to scaled KKT <= 1e-8, the returned trajectories satisfy the dynamics and the
    boxes, and the tick is deterministic."""
    import torch
    B, N, M = 8192, 30, 8
    d = synth.make_batch(B, N=N, M=M, config_id=5, moving=True)
    par = nlp.WholeBodyParams(N=N)
    ctrl = _wb(mm, N, M, B, obs_per_stage=True)
    eng = ctrl._engine
    dev = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x = t(np.clip(d["x_init"], par.xlim[0], par.xlim[1]))
    glob = t(d["traj_ref"])
    step = (glob[:, N] - glob[:, 0]) / N
    glob = glob[:, :1] + step[:, None, :] * torch.arange(51, **f64)[None, :, None]
    obs0, vel = t(d["obs"]), t(d["obs_vel"])
    uref = torch.zeros((B, N, 5), **f64); ul = torch.zeros((B, N, 5), **f64)
    xlo, xhi = t(par.xlim[0]), t(par.xlim[1])
    karr = torch.arange(N + 1, **f64)
    for tick in range(3):
        start = torch.argmin(torch.linalg.norm(x[:, None, :2] - glob[:, :, :2], dim=2), dim=1)
        idx = torch.clamp(start[:, None] + torch.arange(N + 1, device=dev)[None, :], max=50)
        loc = torch.gather(glob, 1, idx[:, :, None].expand(B, N + 1, 9)).contiguous()
        obs = obs0[:, None, :, :].repeat(1, N + 1, 1, 1)
        obs[..., :2] += vel[:, None, :, :] * ((tick + karr) * 0.1)[None, :, None, None]
        obs = obs.contiguous()
        xc = torch.minimum(torch.maximum(x, xlo), xhi)
        r = eng.solve_batch_device(xc, loc, uref, ul, obs)
        torch.cuda.synchronize()
        assert (r["status"] == 0).all(), (tick, int((r["status"] != 0).sum()))
        assert float(r["err"].max()) <= 1e-8
        X, U = r["X"], r["U"]
        c, sn = torch.cos(X[:, :-1, 2]), torch.sin(X[:, :-1, 2])
        Xn = X[:, :-1].clone()
        Xn[:, :, 0] += 0.1 * X[:, :-1, 3]; Xn[:, :, 1] += 0.1 * X[:, :-1, 4]; Xn[:, :, 2] += 0.1 * X[:, :-1, 5]
        Xn[:, :, 3] += 0.1 * (U[:, :, 0] * c - X[:, :-1, 4] * X[:, :-1, 5])
        Xn[:, :, 4] += 0.1 * (U[:, :, 0] * sn + X[:, :-1, 3] * X[:, :-1, 5])
        Xn[:, :, 5] += 0.1 * U[:, :, 1]
        Xn[:, :, 6:] += 0.1 * U[:, :, 2:]
        assert float((Xn - X[:, 1:]).abs().max()) < 1e-8
        assert bool((U <= t(par.ulim[1]) + 1e-7).all()) and bool((U >= t(par.ulim[0]) - 1e-7).all())
        assert bool((X[:, 1:] <= xhi + 1e-7).all()) and bool((X[:, 1:] >= xlo - 1e-7).all())
        assert bool(((U[:, :, 2:] - ul[:, :, 2:]).abs() <= 0.5 + 1e-7).all())        # rate bound vs U_last (:205)
        r2 = eng.solve_batch_device(xc, loc, uref, ul, obs)
        torch.cuda.synchronize()
        assert torch.equal(r["X"], r2["X"]) and torch.equal(r["U"], r2["U"]) and torch.equal(r["iters"], r2["iters"])
        ul = r["U"].clone()
        u0 = U[:, 0]
        cx, sx = torch.cos(xc[:, 2]), torch.sin(xc[:, 2])
        x = torch.stack([xc[:, 0] + 0.1 * xc[:, 3], xc[:, 1] + 0.1 * xc[:, 4], xc[:, 2] + 0.1 * xc[:, 5],
                         xc[:, 3] + 0.1 * (u0[:, 0] * cx - xc[:, 4] * xc[:, 5]), xc[:, 4] + 0.1 * (u0[:, 0] * sx + xc[:, 3] * xc[:, 5]),
                         xc[:, 5] + 0.1 * u0[:, 1], xc[:, 6] + 0.1 * u0[:, 2], xc[:, 7] + 0.1 * u0[:, 3], xc[:, 8] + 0.1 * u0[:, 4]], dim=1)


_RECORD = {}


@pytest.mark.parametrize("name,par,g", load_cases(), ids=[n for n, _, _ in load_cases()])
def test_gpu_against_the_second_source(mm, name, par, g):
    """HIP path vs the committed SLSQP solutions (two SLSQP starts per fixture), rules of tests/test_slsqp_golden.py: every
    output passes the certificate; one whose cost equals that of an SLSQP run is that minimiser to |dX| <= 1e-4, |dU| <= 5e-4;
    the others are recorded as another local minimum, with whether they cost more than the second source's best run."""
    N = par.N
    obs = g["obs"]
    per_stage = obs.ndim == 3
    M = obs.shape[-2]
    hs = g["hs"] if len(g["hs"]) else None
    if par.kind == "base":
        ctrl = mm.MPCBase(mm.Base(0.1), [], N=N, Q=par.Q, P=par.P, max_batch=1, n_obstacles=M)
    else:
        oml = [(h[:3], h[3:].reshape(1, 3)) for h in hs] if hs is not None else []
        ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], oml, N=N, Q=par.Q, P=par.P, max_batch=1, n_obstacles=M,
                               obs_per_stage=per_stage, faithful_convex=False if hs is not None and len(hs) >= 2 else None)
        if par.terminal_xy_equality:
            ctrl.opti.subject_to(ctrl.X[N, :2] == ctrl.X_ref[N, :2])
    assert np.abs(g["u_last"]).max() == 0.0          # cold start in every fixture
    r = ctrl.solve_batch(g["x_init"][None], g["traj_ref"][None], g["u_ref"][None], obs[None])
    assert r["status"][0] == 0
    check_fixture(name, par, g, r["X"][0], r["U"][0], r["s"][0], float(r["cost"][0]), _RECORD)


def test_gpu_second_source_summary():
    """How many of the 67 fixtures the HIP path solves to the minimiser of one of the two SLSQP runs, how many end in a
    costlier local minimum than the second source's best run (printed with -s; bounds in tests/test_slsqp_golden.py)."""
    check_summary(_RECORD, len(load_cases()))


def test_failed_instances_keep_their_warm_start(mm):
    """mmpc_solve_batch replaces u_latest / x_guess only for converged instances (the reference assigns them after a
    successful solve, mpc_wholebody_qref.py:329-330), and a larger batch after a smaller one starts its new rows cold."""
    B, N, M = 32, 15, 3
    d = synth.make_batch(B, N=N, M=M, kind="base", config_id=2)
    ctrl = mm.MPCBase(mm.Base(0.1), [], N=N, max_batch=B, n_obstacles=M, max_iter=8)    # about half converge in 8 iterations
    r = ctrl.solve_batch(d["x_init"], d["traj_ref"], d["u_ref"], d["obs"])
    conv = r["status"] == 0
    assert conv.any() and (r["status"][~conv] == 1).all() and (~conv).sum() >= 4
    ul = ctrl._engine.get_u_latest(B)
    assert np.abs(ul[~conv]).max() == 0.0 and np.array_equal(ul[conv], r["U"][conv])
    full = mm.MPCBase(mm.Base(0.1), [], N=N, max_batch=B, n_obstacles=M)
    r8 = full.solve_batch(d["x_init"][:8], d["traj_ref"][:8], d["u_ref"][:8], d["obs"][:8])
    assert (r8["status"] == 0).all()
    rB = full.solve_batch(d["x_init"], d["traj_ref"], d["u_ref"], d["obs"])                 # rows 8.. have no X guess yet
    par = nlp.BaseParams(N=N)
    o = coracle.solve_batch(par, d["x_init"][8:], d["traj_ref"][8:], d["u_ref"][8:], np.zeros((B - 8, N, 2)), d["obs"][8:], nthreads=8)
    assert (rB["status"] == 0).all()
    assert np.abs(rB["X"][8:] - o["X"]).max() < 5e-6
    o8 = coracle.solve_batch(par, d["x_init"][:8], d["traj_ref"][:8], d["u_ref"][:8], r8["U"], d["obs"][:8], X0=r8["X"], nthreads=8)
    assert np.abs(rB["X"][:8] - o8["X"]).max() < 5e-6


def test_launches_on_two_streams_are_ordered(mm):
    """One handle, alternating HIP streams: each launch waits for the handle's previous one (event), so the schedule hint
    it reads is complete and the outputs equal the one-stream outputs bit for bit."""
    import torch
    B, N, M = 2048, 20, 5
    d = synth.make_batch(B)
    par = nlp.WholeBodyParams()
    ctrl = _wb(mm, N, M, B)
    eng = ctrl._engine
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xi, tr, ur, ob = t(np.clip(d["x_init"], par.xlim[0], par.xlim[1])), t(d["traj_ref"]), t(d["u_ref"]), t(d["obs"])
    ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
    ref = eng.solve_batch_device(xi, tr, ur, ul, ob)
    torch.cuda.synchronize()
    X0 = ref["X"].clone()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    outs = [None, None]
    for i in range(6):
        with torch.cuda.stream(streams[i & 1]):
            outs[i & 1] = eng.solve_batch_device(xi, tr, ur, ul, ob, out=outs[i & 1])
    torch.cuda.synchronize()
    assert torch.equal(outs[0]["X"], X0) and torch.equal(outs[1]["X"], X0)
    for mode in (0, 2):       # batch order; a-priori difficulty key of the batch's own data
        eng.set_schedule_hint(mode)
        r = eng.solve_batch_device(xi, tr, ur, ul, ob)
        torch.cuda.synchronize()
        assert torch.equal(r["X"], X0) and torch.equal(r["iters"], ref["iters"])


def test_iteration_budget_and_continuation(mm):
    """mmpc_set_iteration_budget / mmpc_resume_batch_device: a launch with a budget of 24 iterations leaves the instances
    that need more suspended (status 3, iterate so far in X/U/s); the continuation finishes exactly those, and every
    output - iteration counts included - is bitwise the output of the uninterrupted solve.  Also through the host-pointer
    entry point, which resumes by itself."""
    import torch
    B, N, M = 4096, 20, 5
    d = synth.make_batch(B)
    par = nlp.WholeBodyParams()
    ctrl = _wb(mm, N, M, B)
    eng = ctrl._engine
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xi, tr, ur, ob = t(np.clip(d["x_init"], par.xlim[0], par.xlim[1])), t(d["traj_ref"]), t(d["u_ref"]), t(d["obs"])
    ul = torch.zeros((B, N, 5), dtype=torch.float64, device=dev)
    ref = eng.solve_batch_device(xi, tr, ur, ul, ob)
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in ref.items()}
    long = ref["iters"] > 24
    assert 0 < int(long.sum()) < B // 4
    eng.set_iteration_budget(24)
    out = eng.solve_batch_device(xi, tr, ur, ul, ob)
    torch.cuda.synchronize()
    assert eng.suspended_count() == int(long.sum())
    assert bool((out["status"][long] == 3).all()) and bool((out["status"][~long] == 0).all())
    assert bool((out["iters"][long] == 24).all())
    assert torch.equal(out["X"][~long], ref["X"][~long]) and torch.equal(out["iters"][~long], ref["iters"][~long])
    eng.resume_batch_device(xi, tr, ur, ul, ob, out=out)
    torch.cuda.synchronize()
    for k in ("X", "U", "s", "status", "iters", "cost", "err"):
        assert torch.equal(out[k], ref[k]), k
    # host-pointer entry point: budgeted main launch + continuation inside the call
    r = ctrl.solve_batch(d["x_init"], d["traj_ref"], d["u_ref"], d["obs"])
    assert (r["status"] == 0).all() and np.array_equal(r["X"], ref["X"].cpu().numpy())
    eng.set_iteration_budget(0)
    with pytest.raises(RuntimeError):
        mm.MPCWholeBody(mm.MobileManipulator(0.1), [], [], N=12, max_batch=4, n_obstacles=2)._engine.set_iteration_budget(8)   # generic kernel


def test_device_pointer_calls_reject_oversized_and_stale(mm):
    """include/mmpc.h: B > max_batch is MMPC_E_ARG for the device-pointer calls too (the handle's launch order, difficulty keys,
    save areas and suspended list are sized for max_batch: with a budget set an oversized launch would write past them), and a
    continuation is only valid directly after the budgeted launch it continues (same B)."""
    import torch
    B, N, M = 512, 20, 5
    d = synth.make_batch(B + 1)
    par = nlp.WholeBodyParams()
    ctrl = _wb(mm, N, M, B)
    eng = ctrl._engine
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xi, tr, ur, ob = t(np.clip(d["x_init"], par.xlim[0], par.xlim[1])), t(d["traj_ref"]), t(d["u_ref"]), t(d["obs"])
    ul = torch.zeros((B + 1, N, 5), dtype=torch.float64, device=dev)
    for budget in (0, 8):
        eng.set_iteration_budget(budget)
        with pytest.raises(RuntimeError, match="max_batch"):
            eng.solve_batch_device(xi, tr, ur, ul, ob)
    # budget 8 is still set: a launch of B instances, then its continuation
    out = eng.solve_batch_device(xi[:B], tr[:B], ur[:B], ul[:B], ob[:B])
    torch.cuda.synchronize()
    n_susp = eng.suspended_count()
    assert 0 < n_susp <= B
    with pytest.raises(RuntimeError, match="resume"):      # another batch size than the launch it would continue
        eng.resume_batch_device(xi[:B // 2], tr[:B // 2], ur[:B // 2], ul[:B // 2], ob[:B // 2], out={k: v[:B // 2] for k, v in out.items()})
    out = eng.solve_batch_device(xi[:B], tr[:B], ur[:B], ul[:B], ob[:B])
    eng.resume_batch_device(xi[:B], tr[:B], ur[:B], ul[:B], ob[:B], out=out)
    torch.cuda.synchronize()
    assert bool((out["status"] == 0).all())
    with pytest.raises(RuntimeError, match="resume"):      # a second continuation of the same launch
        eng.resume_batch_device(xi[:B], tr[:B], ur[:B], ul[:B], ob[:B], out=out)
    out = eng.solve_batch_device(xi[:B], tr[:B], ur[:B], ul[:B], ob[:B])
    eng.set_iteration_budget(0)
    with pytest.raises(RuntimeError, match="resume"):      # the budget changed in between
        eng.resume_batch_device(xi[:B], tr[:B], ur[:B], ul[:B], ob[:B], out=out)


@pytest.mark.parametrize("nplanes", [2, 3])
def test_as_written_halfspace_rows_batch(mm, nplanes):
    """64 starts around the demo's 'tent', half-space rows AS WRITTEN (quirk Q8, mmpc_config.as_written; L = 2: the if_else
    branch of obsAvoidConvex, L = 3: mmax): GPU (generic kernel) vs the C oracle, and the certificate of the as-written NLP on
    every converged GPU output.  In about a third of these starts the intended optimum violates an as-written row."""
    r2 = 1 / np.sqrt(2)
    if nplanes == 2:
        hs = np.array([[2.5, 2, 0.35 + 0.606 + 0.333, r2, 0, r2], [2.5, 2, 0.35 + 0.606 + 0.333, -r2, 0, r2]])
    else:
        hs = np.array([[2.5, 2, 1.3, r2, 0, r2], [2.5, 2, 1.3, -r2, 0, r2], [2.5, 2, 1.5, 0, 0, 1.0]])
    B, N = 64, 20
    rng = np.random.default_rng(11)
    x = np.zeros((B, 9)); tr = np.zeros((B, N + 1, 9))
    for b in range(B):
        x0 = np.array([rng.uniform(1.4, 2.6), rng.uniform(1.6, 2.4), rng.uniform(-0.4, 0.4), rng.uniform(0, 0.8), 0, 0,
                       rng.uniform(-0.3, 0.6), rng.uniform(-1.6, -0.6), rng.uniform(0.8, 2.2)])
        x0[4] = x0[3] * np.sin(x0[2]); x0[3] = x0[3] * np.cos(x0[2])
        tg = x0.copy(); tg[0] += rng.uniform(0.8, 1.8); tg[1] += rng.uniform(-0.3, 0.3); tg[3:6] = 0
        x[b] = x0; tr[b] = np.linspace(x0, tg, 51)[:N + 1]
    obs = np.broadcast_to(np.array([[2.5, 3.4, 0.3], [2.5, 0.6, 0.3], [6, 6, 0.1]]), (B, 3, 3)).copy()
    z = np.zeros((B, N, 5))
    par = nlp.WholeBodyParams()
    oml = [(h[:3], h[3:].reshape(1, 3)) for h in hs]
    ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], oml, N=N, max_batch=B, n_obstacles=3)      # the NLP as written
    r = ctrl.solve_batch(x, tr, z, obs)
    o = coracle.solve_batch(par, x, tr, z, z, obs, hs=hs, max_iter=2000, nthreads=16, as_written=True)
    conv = (r["status"] == 0) & (o["status"] == 0)
    assert conv.sum() >= (B if nplanes == 2 else B - 4), (np.nonzero(r["status"])[0].tolist(), np.nonzero(o["status"])[0].tolist())
    assert (r["q8_margin"][r["status"] == 0] <= ctrl.Q8_TOL).all()
    same = conv & (np.abs(r["cost"] / o["cost"] - 1) < 1e-6)
    assert same.sum() >= conv.sum() - 3
    assert np.abs(r["X"][same] - o["X"][same]).max() < 1e-5 and np.abs(r["U"][same] - o["U"][same]).max() < 1e-5
    ok = np.nonzero(r["status"] == 0)[0]
    cs = certify([(nlp.Problem(par, x[b], tr[b], z[b], z[b], obs[b], hs, as_written=True), r["X"][b], r["U"][b], r["s"][b]) for b in ok])
    E0 = np.array([c["E0"] for c in cs])
    assert all(cert_ok(c) for c in cs), (int(ok[int(np.argmax(E0))]), E0.max())
    # the as-written rows matter here: solved with the intended rows only, many of these optima violate one
    loose = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], oml, N=N, max_batch=B, n_obstacles=3, faithful_convex=False)
    ri = loose.solve_batch(x, tr, z, obs)
    viol = mm.controllers._q8.as_written_extra_rows(ri["X"], ri["s"], hs).reshape(B, -1).max(1)
    assert (viol > 1e-6).sum() >= 8


def test_as_written_converges_on_2048_starts(mm):
    """bench.py's C1-shape batch (2048 starts around the demo's two planes, oracle/synth.py:make_c1_starts) under the NLP AS
    WRITTEN on the GPU: 100 % converged (round 2: 43 of them ran into the iteration cap - s_{N-1} reaching back to x_{N-2} was
    eliminated with a diagonal block only; it is a border variable of the stage-wise system now), same answers as the C oracle,
    and the certificate of the as-written NLP on the former failures and on the slowest instances."""
    x, tr, obs, hs = synth.make_c1_starts()
    B, N = x.shape[0], 20
    par = nlp.WholeBodyParams()
    z = np.zeros((B, N, 5))
    oml = [(h[:3], h[3:].reshape(1, 3)) for h in hs]
    ctrl = mm.MPCWholeBody(mm.MobileManipulator(0.1), [], oml, N=N, max_batch=B, n_obstacles=3)      # the NLP as written
    r = ctrl.solve_batch(x, tr, z, obs)
    assert (r["status"] == 0).all(), np.nonzero(r["status"])[0].tolist()
    assert (r["q8_margin"] <= ctrl.Q8_TOL).all()
    o = coracle.solve_batch(par, x, tr, z, z, obs, hs=hs, max_iter=2000, nthreads=16, as_written=True)
    assert (o["status"] == 0).all()
    same = np.abs(r["cost"] / o["cost"] - 1) < 1e-6
    assert same.sum() >= B - 20, int(same.sum())           # (a kink of a max can send the two builds to different branches)
    # (U is the least determined block - the arm inputs carry no R weight -: a 1e-8 KKT test pins it to ~1e-4 on the flattest of
    #  the 2048 instances; measured: X 2.0e-6, U 1.0e-4 on one instance, 2.0e-5 on the next)
    dX = np.abs(r["X"] - o["X"]).reshape(B, -1).max(1); dU = np.abs(r["U"] - o["U"]).reshape(B, -1).max(1)
    assert dX[same].max() < 1e-5 and dU[same].max() < 5e-4 and (dU[same] > 5e-5).sum() <= 4, (dX[same].max(), dU[same].max(), int((dU[same] > 5e-5).sum()))
    loose = np.nonzero(same & (dU > 5e-5))[0]
    sel = np.unique(np.concatenate([[66, 144, 293, 1948], loose, np.argsort(-r["iters"], kind="stable")[:12]]))
    cs = certify([(nlp.Problem(par, x[b], tr[b], z[b], z[b], obs[b], hs, as_written=True), r["X"][b], r["U"][b], r["s"][b]) for b in sel])
    assert all(cert_ok(c) for c in cs), max(c["E0"] for c in cs)
    print("as written, 2048 starts: mean %.1f iterations, max %d; %d of 2048 in the C oracle's minimum" % (r["iters"].mean(), r["iters"].max(), int(same.sum())))


def test_non_finite_inputs_fail_per_instance(mm):
    """NaN / inf in the data of SOME instances: those end with status 2 (MMPC_STATUS_NUMERIC - opti.solve() raises)
    after a handful of iterations, every other instance of the batch gets the result it gets without them.  (The
    specialised kernels use v_max/v_min_f64, which drop a NaN operand: the failure is detected on the sums of the
    evaluation, see the status 2 test of the main loop in csrc/mmpc_fast.h.)"""
    import os
    for N, M, kind in ((20, 5, "wb"), (30, 8, "wb"), (15, 3, "base"), (20, 5, "generic")):
        B = 64
        if kind in ("wb", "generic"):
            d = synth.make_batch(B, N=N, M=M)
            par = nlp.WholeBodyParams(N=N)
            if kind == "generic":
                os.environ["MMPC_FORCE_GENERIC"] = "1"     # read when the handle is created
            try:
                ctrl = _wb(mm, N, M, B)
            finally:
                os.environ.pop("MMPC_FORCE_GENERIC", None)
            xi = np.clip(d["x_init"], par.xlim[0], par.xlim[1])
        else:
            d = synth.make_batch(B, N=N, M=M, kind="base", config_id=2)
            ctrl = mm.MPCBase(mm.Base(0.1), [], N=N, max_batch=B, n_obstacles=M)
            xi = d["x_init"].copy()
        clean = ctrl.solve_batch(xi, d["traj_ref"], d["u_ref"], d["obs"])
        assert (clean["status"] == 0).all()
        ctrl.reset()
        x2, tr2, ob2, ur2 = xi.copy(), d["traj_ref"].copy(), d["obs"].copy(), d["u_ref"].copy()
        x2[3, 0] = np.nan; tr2[9, N // 2, 1] = np.nan; ob2[17, 0, 2] = np.inf; ur2[30, 2, 0] = np.nan; tr2[41, 0, 0] = -np.inf
        bad = np.array([3, 9, 17, 30, 41])
        r = ctrl.solve_batch(x2, tr2, ur2, ob2)
        good = np.setdiff1d(np.arange(B), bad)
        assert (r["status"][bad] == 2).all(), (kind, N, r["status"][bad], r["iters"][bad])
        assert (r["iters"][bad] <= 3).all()
        assert (r["status"][good] == 0).all() and np.array_equal(r["X"][good], clean["X"][good]) and np.array_equal(r["U"][good], clean["U"][good])


@pytest.mark.gpu
def test_c5_seed6_fleet_converges_on_every_tick(mm):
    """The C5 fleet of generator seed 6 (8192 robots, N = 30, 8 moving obstacles, warm start per mpc_wholebody_qref.py:301-310), ten
    ticks in lock step: in round 3 two of its 81 920 solves stopped at the reference's iteration cap (ticks 3 and 6: a
    near-degenerate minimum next to an active circle row, where Newton steps that the linearised row allows violate it to second
    order - the Maratos effect; they needed 2132 and 2966 iterations) and the closed-loop driver raised on them as the reference
    would on an IPOPT failure.  With the second-order correction every solve converges."""
    import torch
    B, N, M, T = 8192, 30, 8, 10
    d = synth.make_batch(B, N=N, M=M, config_id=6, moving=True)
    par = nlp.WholeBodyParams(N=N)
    dev = torch.device("cuda", 0)
    glob = torch.from_numpy(d["traj_ref"]).to(dev)
    step = (glob[:, N] - glob[:, 0]) / N
    glob = glob[:, :1] + step[:, None, :] * torch.arange(51, dtype=torch.float64, device=dev)[None, :, None]
    fleet = mm.DeviceFleet(mm, np.clip(d["x_init"], par.xlim[0], par.xlim[1]), glob, d["obs"], d["obs_vel"], N=N)
    r = fleet.run_lockstep(T)
    torch.cuda.synchronize()
    it = r["iters"].cpu().numpy()
    assert bool(r["all_converged"]), it.max(0)
    assert it.max() <= 1000 and 30 <= it.mean() <= 38, (it.max(0), it.mean())


@pytest.mark.gpu
def test_long_horizon_kernel_against_its_host_build(mm):
    """The N = 30 kernel keeps its feedback gains in global memory and feeds the roll-out through a four-slot ring in LDS (a
    cross-lane hand-over the host build of the kernel does not have): same iteration counts and trajectories as the host build on
    the first instances of the C5 generator, obstacle table per stage."""
    import torch
    import emu_helper
    B, N, M = 48, 30, 8
    d = synth.make_batch(B, N=N, M=M, config_id=5, moving=True)
    par = nlp.WholeBodyParams(N=N)
    x = np.clip(d["x_init"], par.xlim[0], par.xlim[1])
    obs = d["obs"][:, None, :, :].repeat(N + 1, axis=1).copy()
    obs[..., :2] += d["obs_vel"][:, None, :, :] * (np.arange(N + 1.0) * 0.1)[None, :, None, None]
    ul = np.zeros((B, N, 5))
    e = emu_helper.solve_batch(par, x, d["traj_ref"], d["u_ref"], ul, obs, fast=True, max_iter=2000)
    ctrl = _wb(mm, N, M, B, obs_per_stage=True)
    r = ctrl.solve_batch(x, d["traj_ref"], d["u_ref"], obs)
    assert (r["status"] == 0).all() and (e["status"] == 0).all()
    assert (r["iters"] == e["iters"]).mean() >= 0.9, (r["iters"], e["iters"])
    same = np.abs(r["cost"] / e["cost"] - 1) < 1e-6
    assert same.mean() >= 0.95 and np.abs(r["X"][same] - e["X"][same]).max() < 1e-6
