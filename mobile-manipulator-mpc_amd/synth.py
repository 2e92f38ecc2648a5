"""Seeded synthetic (x_init, traj_ref, u_ref, obstacles) generator - SURVEY §8(d).

Numpy only (no GPU, no library): the workload generator of bench.py, the tools and the tests (which reach it as
`oracle.synth`, a re-export).  Modelled on demo_wholebody_qref.py:28-44 (scenario
constants) and interface_wholebody_qref.py:247-266 (globalPlan2D: linspace from the
start state to the target over t_move/dt + 1 points) / :353-396 (first N+1 rows as
the local reference).  The reference has no synthetic generator; this one is the
build's, shared by tests, bench.py and the CPU baseline so that all of them see
identical inputs.
"""
import numpy as np

PI = np.pi


def make_batch(B, N=20, M=5, kind="wholebody", seed=20240114, config_id=3, moving=False, dt=0.1,
               t_move=5.0):
    rng = np.random.default_rng(seed + config_id)
    nx = 9 if kind == "wholebody" else 6
    nu = 5 if kind == "wholebody" else 2
    x_init = np.zeros((B, nx))
    traj = np.zeros((B, N + 1, nx))
    u_ref = np.zeros((B, N, nu))
    obs = np.zeros((B, M, 3))
    vel = np.zeros((B, M, 2))
    n_glob = int(round(t_move / dt)) + 1
    for b in range(B):
        p0 = rng.uniform(-1, 1, 2)
        psi0 = rng.uniform(-PI, PI)
        V = rng.uniform(0, 1)
        dpsi = rng.uniform(-0.5, 0.5)
        x0 = np.zeros(nx)
        x0[:6] = [p0[0], p0[1], psi0, V * np.cos(psi0), V * np.sin(psi0), dpsi]
        if nx == 9:
            x0[6] = rng.uniform(-PI / 4, PI / 4)
            x0[7] = rng.uniform(-3 * PI / 4, -PI / 8)
            x0[8] = rng.uniform(PI / 8, PI)
        rng_goal = rng.uniform(4, 8)
        bearing = rng.uniform(-PI, PI)
        goal = p0 + rng_goal * np.array([np.cos(bearing), np.sin(bearing)])
        psi_goal = rng.uniform(-PI, PI)
        xt = np.zeros(nx)
        xt[:3] = [goal[0], goal[1], psi_goal]
        if nx == 9:
            xt[6:] = x0[6:]
        glob = np.linspace(x0, xt, n_glob)
        if N + 1 <= n_glob:
            traj[b] = glob[:N + 1]
        else:
            traj[b, :n_glob] = glob
            traj[b, n_glob:] = glob[-1]
        x_init[b] = x0
        d = (goal - p0) / rng_goal
        nrm = np.array([-d[1], d[0]])
        for m in range(M):
            while True:
                tpar = rng.uniform(0.15, 0.9)
                lat = rng.uniform(0.3, 1.5) * (1 if rng.uniform() < 0.5 else -1)
                r = rng.uniform(0.1, 0.6)
                c = p0 + tpar * (goal - p0) + lat * nrm
                if np.linalg.norm(c - p0) >= r + 0.4 + 0.2:
                    break
            obs[b, m] = [c[0], c[1], r]
            vel[b, m] = rng.uniform(-0.5, 0.5, 2)
    out = dict(x_init=x_init, traj_ref=traj, u_ref=u_ref, obs=obs)
    if moving:
        out["obs_vel"] = vel
    return out


def make_c1_starts(B=2048, N=20, nplanes=2, seed=11):
    """Starts around the two (three) half-space obstacles of demo_wholebody_qref.py:21-33 ('the tent'): the shape of BASELINE
    config C1 as a batch (bench.py: c1_shape_generic_kernel; tests: the NLP as written must converge on all of them)."""
    r2 = 1 / np.sqrt(2)
    if nplanes == 2:
        hs = np.array([[2.5, 2, 0.35 + 0.606 + 0.333, r2, 0, r2], [2.5, 2, 0.35 + 0.606 + 0.333, -r2, 0, r2]])
    else:
        hs = np.array([[2.5, 2, 1.3, r2, 0, r2], [2.5, 2, 1.3, -r2, 0, r2], [2.5, 2, 1.5, 0, 0, 1.0]])
    rng = np.random.default_rng(seed)
    x = np.zeros((B, 9)); tr = np.zeros((B, N + 1, 9))
    for b in range(B):
        x0 = np.array([rng.uniform(1.4, 2.6), rng.uniform(1.6, 2.4), rng.uniform(-0.4, 0.4), rng.uniform(0, 0.8), 0, 0,
                       rng.uniform(-0.3, 0.6), rng.uniform(-1.6, -0.6), rng.uniform(0.8, 2.2)])
        x0[4] = x0[3] * np.sin(x0[2]); x0[3] = x0[3] * np.cos(x0[2])
        tg = x0.copy(); tg[0] += rng.uniform(0.8, 1.8); tg[1] += rng.uniform(-0.3, 0.3); tg[3:6] = 0
        x[b] = x0; tr[b] = np.linspace(x0, tg, 51)[:N + 1]
    obs = np.broadcast_to(np.array([[2.5, 3.4, 0.3], [2.5, 0.6, 0.3], [6, 6, 0.1]]), (B, 3, 3)).copy()
    return x, tr, obs, hs
