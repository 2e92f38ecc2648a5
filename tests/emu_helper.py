"""ctypes driver for the host lane-emulation build of the HIP solver core (tests/emu/mmpc_emu.cpp).
TEST ONLY: builds with g++ -DMMPC_EMU; never used by the product package."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "emu", "mmpc_emu.cpp")
_CORE = os.path.join(_HERE, "..", "mobile-manipulator-mpc_amd", "csrc", "mmpc_core.h")
_FAST = os.path.join(_HERE, "..", "mobile-manipulator-mpc_amd", "csrc", "mmpc_fast.h")
_TILE = os.path.join(_HERE, "..", "mobile-manipulator-mpc_amd", "csrc", "mmpc_tile.h")
_IK = os.path.join(_HERE, "..", "mobile-manipulator-mpc_amd", "csrc", "mmpc_ik.h")


class MmpcParams(C.Structure):
    _fields_ = [("N", C.c_int), ("M", C.c_int), ("obs_per_stage", C.c_int), ("max_iter", C.c_int),
                ("use_xguess", C.c_int), ("terminal_xy_eq", C.c_int),
                ("dt", C.c_double), ("tol", C.c_double), ("mu_init", C.c_double), ("S", C.c_double),
                ("Q2", C.c_double * 81), ("P2", C.c_double * 81), ("RW2", C.c_double * 25),
                ("R2", C.c_double * 25), ("W2", C.c_double * 25),
                ("ulim", (C.c_double * 5) * 2), ("xlim", (C.c_double * 9) * 2), ("dulim", (C.c_double * 5) * 2),
                ("L", C.c_int), ("hs", (C.c_double * 6) * 8), ("u_guess", C.c_void_p), ("as_written", C.c_int)]


def build(asan=False, defs=()):
    """defs: extra -D switches of the kernel source (A/B switches such as MMPC_PADMAP=0): a library of its own per set"""
    tag = ("_" + "_".join(d.replace("=", "") for d in defs)) if defs else ""
    out = os.path.join(_HERE, "emu", "_build", ("libmmpc_emu_asan%s.so" if asan else "libmmpc_emu%s.so") % tag)
    csrc = os.path.dirname(_FAST)
    newest = max(os.path.getmtime(f) for f in [_SRC, _CORE, _FAST, _TILE, _IK] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith('.inc')])
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if asan else ["-O2"]
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", *flags, *["-D" + d for d in defs], "-o", out, _SRC])
    return out


def make_params(par, M, obs_per_stage=False, use_xguess=False, tol=1e-8, mu_init=1.0, max_iter=200, hs=None, as_written=False):
    """MmpcParams as the C ABI would build it from mmpc_config + weights (terminal_xy_eq from par)."""
    p = MmpcParams()
    nx, nu = par.nx, par.nu
    p.N, p.M, p.obs_per_stage, p.max_iter = par.N, M, int(obs_per_stage), max_iter
    p.use_xguess, p.terminal_xy_eq = int(use_xguess), int(bool(getattr(par, 'terminal_xy_equality', False)))
    p.dt, p.tol, p.mu_init, p.S = par.dt, tol, mu_init, float(np.ravel(par.S)[0])
    Q2 = par.Q + par.Q.T; P2 = par.P + par.P.T; R2 = par.R + par.R.T; W2 = par.W + par.W.T
    for i, v in enumerate(Q2.ravel()): p.Q2[i] = v
    for i, v in enumerate(P2.ravel()): p.P2[i] = v
    for i, v in enumerate(R2.ravel()): p.R2[i] = v
    for i, v in enumerate(W2.ravel()): p.W2[i] = v
    for i, v in enumerate((R2 + W2).ravel()): p.RW2[i] = v
    for r in range(2):
        for j in range(nu):
            p.ulim[r][j] = par.ulim[r, j]; p.dulim[r][j] = par.dulim[r, j]
        for j in range(nx):
            p.xlim[r][j] = par.xlim[r, j]
    p.L = 0
    p.as_written = int(bool(as_written))
    if hs is not None and len(hs):
        hs = np.asarray(hs, float).reshape(-1, 6)
        p.L = hs.shape[0]
        for j in range(p.L):
            for a in range(6):
                p.hs[j][a] = hs[j, a]
    return p


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def solve_batch(par, x_init, traj_ref, u_ref, u_last, obs, x_guess=None, reverse=False, asan=False, fast=False, u_guess=None, defs=(), **kw):
    lib = C.CDLL(build(asan, defs))
    assert lib.mmpc_emu_params_size() == C.sizeof(MmpcParams)
    x_init = np.ascontiguousarray(x_init, float); traj_ref = np.ascontiguousarray(traj_ref, float)
    u_ref = np.ascontiguousarray(u_ref, float); u_last = np.ascontiguousarray(u_last, float)
    obs = np.ascontiguousarray(obs, float)
    if x_guess is not None:
        x_guess = np.ascontiguousarray(x_guess, float)
    B = x_init.shape[0]
    N, nx, nu = par.N, par.nx, par.nu
    M = obs.shape[-2]
    prm = make_params(par, M, obs.ndim == 4, x_guess is not None, **kw)
    if u_guess is not None:
        u_guess = np.ascontiguousarray(u_guess, float)
        prm.u_guess = u_guess.ctypes.data
    X = np.zeros((B, N + 1, nx)); U = np.zeros((B, N, nu)); s = np.zeros((B, N + 1))
    status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); cost = np.zeros(B); err = np.zeros(B)
    fn = lib.mmpc_emu_solve_fast if fast else lib.mmpc_emu_solve
    kind = (2 if getattr(par, "pose_ref", False) else 0) if par.kind == "wholebody" else 1
    rc = fn(kind, C.byref(prm), B, _p(x_init), _p(traj_ref), _p(u_ref),
                       _p(u_last), _p(x_guess), _p(obs), _p(X), _p(U), _p(s),
                       status.ctypes.data_as(C.POINTER(C.c_int)), iters.ctypes.data_as(C.POINTER(C.c_int)),
                       _p(cost), _p(err), int(reverse))
    if rc != 0:
        raise RuntimeError('no fast instantiation for this configuration')
    return dict(X=X, U=U, s=s, status=status, iters=iters, cost=cost, err=err)


def ik_batch(q0, target_xz, asan=False):
    """mmpc_ik_solve (the function the GPU kernel runs per lane) on the host."""
    lib = C.CDLL(build(asan))
    q0 = np.ascontiguousarray(np.atleast_2d(q0), float); t = np.ascontiguousarray(np.atleast_2d(target_xz), float)
    B = q0.shape[0]
    q = np.zeros((B, 3)); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
    lib.mmpc_emu_ik(B, _p(q0), _p(t), _p(q), st.ctypes.data_as(C.POINTER(C.c_int)), it.ctypes.data_as(C.POINTER(C.c_int)))
    return dict(q=q, status=st, iters=it)


def solve_fast_budgeted(par, x_init, traj_ref, u_ref, u_last, obs, budget, max_rounds=200, **kw):
    """The specialised kernel (host build) with an iteration budget per launch: first launch, then continuation launches
    (each again with the budget) until nobody is suspended.  Returns the outputs and the number of launches."""
    lib = C.CDLL(build(False))
    x_init = np.ascontiguousarray(x_init, float); traj_ref = np.ascontiguousarray(traj_ref, float)
    u_ref = np.ascontiguousarray(u_ref, float); u_last = np.ascontiguousarray(u_last, float); obs = np.ascontiguousarray(obs, float)
    B = x_init.shape[0]
    N, nx, nu = par.N, par.nx, par.nu
    M = obs.shape[-2]
    prm = make_params(par, M, obs.ndim == 4, False, **kw)
    kind = 0 if par.kind == "wholebody" else 1
    sd = lib.mmpc_emu_fast_state_doubles(kind, N, M)
    assert sd > 0
    state = np.full((B, sd), np.nan)
    X = np.zeros((B, N + 1, nx)); U = np.zeros((B, N, nu)); s = np.zeros((B, N + 1))
    status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); cost = np.zeros(B); err = np.zeros(B)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    launches = 0
    for rnd in range(max_rounds):
        rc = lib.mmpc_emu_solve_fast_budget(kind, C.byref(prm), B, _p(x_init), _p(traj_ref), _p(u_ref), _p(u_last), None, _p(obs),
                                            _p(X), _p(U), _p(s), ip(status), ip(iters), _p(cost), _p(err), int(budget), _p(state),
                                            int(rnd > 0))
        assert rc == 0
        launches += 1
        if not (status == 3).any():
            break
    return dict(X=X, U=U, s=s, status=status, iters=iters, cost=cost, err=err), launches
