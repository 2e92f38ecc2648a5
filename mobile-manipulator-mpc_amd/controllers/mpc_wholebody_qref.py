import numpy as np

from .. import _capi
from ._facade import OptiFacade, Sym
from . import _q8

PI, INF = np.pi, np.inf


class MPCWholeBody:
    """Drop-in counterpart of controllers/mpc_wholebody_qref.py:6-331 backed by the HIP engine.

    Same constructor arguments, attributes and methods the closed-loop driver uses
    (interface_wholebody_qref.py:110,114,134,143,166-167,175,194,212): ``solve``, ``reset``,
    ``setWeight``, ``angleDiff``, ``N``, ``f_dynamics``, ``robot_model``, ``obstacle_list``,
    ``opti`` / ``X`` / ``X_ref`` (facade for the terminal-xy equality).  Extension:
    ``solve_batch`` and ``max_batch`` / ``device`` keyword arguments.
    """

    def __init__(self, robot, obstacle_list, obstacle_manipulation_list, N=10,
                 Q=5 * np.diag([5, 5, 0, 0, 0, 1, 1, 1, 1]),
                 P=5 * np.diag([5, 5, 0, 0, 0, 1, 1, 1, 1]),
                 R=np.diag([0.1, 0.1, 0.0, 0.0, 0.0]),
                 S=np.diag([1e5]),
                 W=np.diag([0, 0, 1e-1, 1e-1, 1e-1]),
                 ulim=np.array([[-2, -PI, -1, -1, -1], [2, PI, 1, 1, 1]]),
                 xlim=np.array([[-100, -100, -INF, -2, -2, -PI, -PI / 2, -PI, 0],
                                [100, 100, INF, 2, 2, PI, PI / 2, 0, 3 * PI / 2]]),
                 dulim=np.array([[-INF, -INF, -0.5, -0.5, -0.5], [INF, INF, 0.5, 0.5, 0.5]]),
                 max_batch=1, device=0, obs_per_stage=False, n_obstacles=None, tol=1e-8, max_iter=2000,
                 faithful_convex=None):
        self.N = N
        self.Q_value, self.R_value, self.P_value, self.S_value, self.W_value = Q, R, P, S, W
        self.dt = robot.dt
        self.dulim, self.ulim, self.xlim = np.asarray(dulim, float), np.asarray(ulim, float), np.asarray(xlim, float)
        self.f_dynamics = robot.f_kinematics
        self.robot_model = robot
        self.base_radius = robot.base.base_radius()
        self.obstacle_list = obstacle_list
        self.obstacle_manipulation_list = obstacle_manipulation_list
        self.endpoint_self_collision_radius = 0.05   # mpc_wholebody_qref.py:43
        self.obstacle_expand_dist = 0.03             # :44
        # half-space ("manipulation") obstacles, mpc_wholebody_qref.py:57-89.  For L >= 2 planes the reference AS WRITTEN emits L
        # rows per (stage, arm sample point): the last is the intended -max_j n_j.((p_j - 0.03 n_j) - P_i) <= s_k, the other L-1
        # read the previous stage's `constr` entries (see _q8.py) and tie x_k to x_{k-1}.
        # Default (faithful_convex None / True): the NLP as written - the generic kernel carries the coupled rows
        # (mmpc_config.as_written); after every solve the extra rows are re-evaluated on the host as a post-condition.
        # faithful_convex=False: the intended rows only.  L = 1 has no extra rows.
        hs = [np.concatenate([np.asarray(pt, float).reshape(3), np.asarray(nrm, float).reshape(3)])
              for pt, nrm in obstacle_manipulation_list]
        self._hs = np.array(hs, float).reshape(-1, 6)
        self._q8_check = len(hs) >= 2 and faithful_convex is not False
        self.q8_margin = None      # largest value of an as-written extra row at the last solve (<= 0: all satisfied)
        if abs(self.base_radius - 0.4) > 0 or abs(self.endpoint_self_collision_radius - 0.05) > 0:
            raise ValueError("the kernels bake base_radius 0.4 / self-collision radius 0.05 (base.py:15, :43)")
        self._M = len(obstacle_list) if n_obstacles is None else int(n_obstacles)
        self._engine = _capi.Engine(_capi.KIND_WHOLEBODY, N, self._M, self.dt, self.ulim, self.xlim, self.dulim,
                                    max_batch=max_batch, device=device, obs_per_stage=obs_per_stage, tol=tol,
                                    max_iter=max_iter, halfspaces=hs, as_written=self._q8_check)
        self.max_batch = max_batch
        self.X, self.X_ref = Sym("X"), Sym("X_ref")
        self.reset()

    # ---- reference surface -------------------------------------------------------------
    def angleDiff(self, a, b):
        """mpc_wholebody_qref.py:92-117 on floats."""
        a = np.fmod(a + PI, 2 * PI) - PI
        b = np.fmod(b + PI, 2 * PI) - PI
        d = a - b
        if a * b >= 0:
            return d
        if a > b:
            return d if d <= PI else d - 2 * PI
        return d if d > -PI else d + 2 * PI

    def setWeight(self, Q=None, R=None, P=None, S=None, W=None):
        """mpc_wholebody_qref.py:119-139."""
        if Q is not None: self.Q_value = Q
        if R is not None: self.R_value = R
        if P is not None: self.P_value = P
        if S is not None: self.S_value = S
        if W is not None: self.W_value = W
        self._engine.set_weights(self.Q_value, self.R_value, self.P_value, self.S_value, self.W_value)

    def reset(self):
        """mpc_wholebody_qref.py:142-285: a fresh NLP - warm start cleared, extra constraints dropped."""
        self.opti = OptiFacade(self)
        self.x_guess = None
        self.u_latest = None
        self._engine.set_terminal_xy_equality(False)
        self._engine.reset()
        self.setWeight()

    def _set_terminal_xy_equality(self, on):
        self._engine.set_terminal_xy_equality(on)

    def _obs_array(self, B):
        o = np.array([[ob.x, ob.y, ob.radius] for ob in self.obstacle_list], float).reshape(-1, 3)
        return np.broadcast_to(o, (B,) + o.shape).copy()

    def solve(self, x_init, traj_ref, u_ref):
        """mpc_wholebody_qref.py:287-331: returns U*[0] (5,); keeps u_latest / x_guess."""
        x_init[6:] = np.maximum(np.minimum(x_init[6:], self.xlim[1, 6:]), self.xlim[0, 6:]).squeeze()  # :290 (in place)
        x_init = np.maximum(np.minimum(x_init, self.xlim[1]), self.xlim[0]).squeeze()                   # :291
        assert x_init[7] <= 0 and x_init[8] >= 0                                                        # :292
        r = self._engine.solve_batch(np.asarray(x_init, float)[None], np.asarray(traj_ref, float)[None],
                                     np.asarray(u_ref, float)[None], self._obs_array(1))
        if r["status"][0] != 0:
            # the reference dies here too (RuntimeError from opti.solve(), then UnboundLocalError, :314-329)
            raise RuntimeError("MPC solve failed: status %d after %d iterations" % (r["status"][0], r["iters"][0]))
        if self._q8_check:
            self._apply_q8_check(r)
            if r["status"][0] != 0:
                raise RuntimeError("post-condition failed: an as-written half-space row (quirk Q8) is violated by %.3g at a "
                                   "solution the engine reported as converged" % self.q8_margin)
        print("cost: ", r["cost"][0])                                                                   # :317
        self.x_guess = r["X"][0]
        self.u_latest = r["U"][0]
        return self.u_latest[0, :]

    # ---- batched extension -------------------------------------------------------------
    def solve_batch(self, x_init, traj_ref, u_ref, obs=None):
        """B independent instances of solve(); obs (B,M,3) [or (B,N+1,M,3)] overrides obstacle_list.
        Returns dict(u0,X,U,s,status,iters,cost); the warm start of instance b is kept for the next call."""
        x_init = np.array(x_init, float)
        x_init = np.maximum(np.minimum(x_init, self.xlim[1]), self.xlim[0])
        B = x_init.shape[0]
        if obs is None:
            obs = self._obs_array(B)
        r = self._engine.solve_batch(x_init, traj_ref, u_ref, obs)
        if self._q8_check:
            self._apply_q8_check(r)
        return r

    Q8_TOL = 1e-7      # post-condition on the extra rows (the solver stops at a scaled KKT error of 1e-8)

    def _apply_q8_check(self, r):
        """Marks converged instances whose solution violates an as-written extra row with _capi.STATUS_Q8_REFUSED."""
        rows = _q8.as_written_extra_rows(r["X"], r["s"], self._hs)                 # (B, N, 6, L-1)
        worst = rows.reshape(rows.shape[0], -1).max(axis=1)
        self.q8_margin = float(worst.max())
        r["q8_margin"] = worst
        r["status"] = np.where((r["status"] == 0) & (worst > self.Q8_TOL), _capi.STATUS_Q8_REFUSED, r["status"]).astype(np.int32)
