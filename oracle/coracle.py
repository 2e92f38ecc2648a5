"""ctypes binding of oracle/mmpc_oracle.c (TEST INFRASTRUCTURE ONLY - see that file's header)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libmmpc_oracle.so")


class OracleCfg(C.Structure):
    _fields_ = [("kind", C.c_int), ("N", C.c_int), ("M", C.c_int), ("obs_per_stage", C.c_int),
                ("terminal_xy_eq", C.c_int), ("dt", C.c_double),
                ("Q", C.c_double * 81), ("P", C.c_double * 81), ("R", C.c_double * 25), ("W", C.c_double * 25),
                ("S", C.c_double),
                ("ulim", (C.c_double * 5) * 2), ("xlim", (C.c_double * 9) * 2), ("dulim", (C.c_double * 5) * 2),
                ("tol", C.c_double), ("mu_init", C.c_double), ("max_iter", C.c_int),
                ("L", C.c_int), ("hs", (C.c_double * 6) * 8), ("pose_ref", C.c_int), ("as_written", C.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "mmpc_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.mmpc_oracle_cfg_size() == C.sizeof(OracleCfg)
    return _lib


def make_cfg(par, M, obs_per_stage=False, tol=1e-8, mu_init=1.0, max_iter=200, hs=None, as_written=False):
    c = OracleCfg()
    c.kind = 0 if par.kind == "wholebody" else 1
    c.N, c.M, c.obs_per_stage = par.N, M, int(obs_per_stage)
    c.terminal_xy_eq = int(par.terminal_xy_equality)
    c.pose_ref = int(getattr(par, "pose_ref", False))
    c.as_written = int(bool(as_written))
    c.dt = par.dt
    nx, nu = par.nx, par.nu

    def put(dst, a):
        a = np.ascontiguousarray(a, dtype=float).ravel()
        for i, v in enumerate(a):
            dst[i] = v
    put(c.Q, par.Q); put(c.P, par.P); put(c.R, par.R); put(c.W, par.W)
    c.S = float(np.ravel(par.S)[0])
    for r in range(2):
        for j in range(nu):
            c.ulim[r][j] = par.ulim[r, j]
            c.dulim[r][j] = par.dulim[r, j]
        for j in range(nx):
            c.xlim[r][j] = par.xlim[r, j]
    c.tol, c.mu_init, c.max_iter = tol, mu_init, max_iter
    c.L = 0
    if hs is not None and len(hs):
        hs = np.asarray(hs, float).reshape(-1, 6)
        c.L = hs.shape[0]
        for j in range(c.L):
            for a in range(6):
                c.hs[j][a] = hs[j, a]
    return c


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def solve_batch(par, x_init, traj_ref, u_ref, u_last, obs, X0=None, nthreads=1, U0=None, **kw):
    """x_init (B,nx) [already clipped], traj_ref (B,N+1,nx), u_ref/u_last (B,N,nu), obs (B,M,3)|(B,N+1,M,3)."""
    x_init = np.ascontiguousarray(x_init, float); traj_ref = np.ascontiguousarray(traj_ref, float)
    u_ref = np.ascontiguousarray(u_ref, float); u_last = np.ascontiguousarray(u_last, float)
    obs = np.ascontiguousarray(obs, float)
    B = x_init.shape[0]
    N, nx, nu = par.N, par.nx, par.nu
    per_stage = obs.ndim == 4
    M = obs.shape[-2]
    cfg = make_cfg(par, M, per_stage, **kw)
    X = np.zeros((B, N + 1, nx)); U = np.zeros((B, N, nu)); s = np.zeros((B, N + 1))
    status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); cost = np.zeros(B); err = np.zeros(B)
    if X0 is not None:
        X0 = np.ascontiguousarray(X0, float)
    if U0 is not None:
        U0 = np.ascontiguousarray(U0, float)
    lib().mmpc_oracle_solve_batch_guess(C.byref(cfg), B, _p(x_init), _p(traj_ref), _p(u_ref), _p(u_last),
                                  _p(X0) if X0 is not None else None, _p(U0) if U0 is not None else None, _p(obs), _p(X), _p(U), _p(s),
                                  status.ctypes.data_as(C.POINTER(C.c_int)), iters.ctypes.data_as(C.POINTER(C.c_int)),
                                  _p(cost), _p(err), int(nthreads))
    return dict(X=X, U=U, s=s, status=status, iters=iters, cost=cost, err=err)


def fk(x):
    e = np.zeros(4); j2 = np.zeros(3); j3 = np.zeros(3)
    x = np.ascontiguousarray(x, float)
    lib().mmpc_oracle_fk(_p(x), _p(e), _p(j2), _p(j3))
    return e, j2, j3


def f(kind, dt, x, u):
    nx = 9 if kind == "wholebody" else 6
    out = np.zeros(nx)
    x = np.ascontiguousarray(x, float); u = np.ascontiguousarray(u, float)
    lib().mmpc_oracle_f(0 if kind == "wholebody" else 1, C.c_double(dt), _p(x), _p(u), _p(out))
    return out
