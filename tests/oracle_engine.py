"""TEST ONLY: a stand-in for `_capi.Engine` that solves with the CPU oracle (oracle/mmpc_oracle.c), so that the host
logic around the engine (controller classes, task state machine) can be exercised without a GPU.  The product never
imports this; the GPU tests run the same scenarios through the real engine."""
import numpy as np

from oracle import coracle, nlp
from tests import emu_helper


class OracleEngine:
    def __init__(self, kind, N, M, dt, ulim, xlim, dulim, max_batch=1, device=0, obs_per_stage=False, tol=1e-8,
                 mu_init=1.0, max_iter=200, halfspaces=None, as_written=False):
        self.kind, self.N, self.M = kind, N, M
        self.par = nlp.WholeBodyParams(N=N, dt=dt) if kind == 0 else nlp.BaseParams(N=N, dt=dt)
        self.par.ulim, self.par.xlim, self.par.dulim = np.asarray(ulim, float), np.asarray(xlim, float), np.asarray(dulim, float)
        self.nx, self.nu = self.par.nx, self.par.nu
        self.hs = halfspaces if halfspaces is not None and len(halfspaces) else None
        self.tol, self.max_iter = tol, max_iter
        self.as_written = bool(as_written) and self.hs is not None and len(self.hs) >= 2
        self.u_latest = None
        self.x_guess = None

    def set_weights(self, Q=None, R=None, P=None, S=None, W=None):
        p = self.par
        if Q is not None: p.Q = np.asarray(Q, float)
        if R is not None: p.R = np.asarray(R, float)
        if P is not None: p.P = np.asarray(P, float)
        if S is not None: p.S = float(np.ravel(S)[0])
        if W is not None and self.kind == 0: p.W = np.asarray(W, float)

    def set_terminal_xy_equality(self, on):
        self.par.terminal_xy_equality = bool(on)

    def reset(self):
        self.u_latest = None
        self.x_guess = None

    def solve_batch(self, x_init, traj_ref, u_ref, obs):
        B = np.asarray(x_init).shape[0]
        ul = np.zeros((B, self.N, self.nu)) if self.u_latest is None else self.u_latest[:B]
        X0 = self.x_guess[:B] if (self.kind == 1 and self.x_guess is not None) else None
        r = coracle.solve_batch(self.par, x_init, traj_ref, u_ref, ul, obs, X0=X0, hs=self.hs, tol=self.tol,
                                max_iter=self.max_iter, as_written=self.as_written)
        self.u_latest = r["U"].copy()
        self.x_guess = r["X"].copy()
        r["u0"] = r["U"][:, 0, :].copy()
        return r


def patch(monkeypatch, mm):
    """routes the package's engine and IK entry points to the CPU oracle / host build for the duration of a test"""
    monkeypatch.setattr(mm._capi, "Engine", OracleEngine)
    monkeypatch.setattr(mm._capi, "ik_batch", lambda q0, t, device=0: emu_helper.ik_batch(q0, t))
