import numpy as np

from .. import _capi

PI, INF = np.pi, np.inf


class MPCBase:
    """Drop-in counterpart of controllers/mpc_base.py:6-226 (base-only diff-drive MPC) on the HIP engine."""

    def __init__(self, robot, obstacle_list, N=10,
                 Q=np.diag([5., 5., 0.0, 0, 0, 1.]), P=np.diag([5., 5., 0.0, 0, 0, 1.]), R=np.diag([1., 1.]),
                 M=np.diag([1e5]),
                 ulim=np.array([[-2, -PI], [2, PI]]),
                 xlim=np.array([[-100, -100, -2, -2, -PI], [100, 100, 2, 2, PI]]),
                 max_batch=1, device=0, obs_per_stage=False, n_obstacles=None, tol=1e-8, max_iter=2000):
        self.Q_value, self.R_value, self.P_value, self.M_value = Q, R, P, M
        self.dt = robot.dt
        self.N = N
        self.ulim, self.xlim = np.asarray(ulim, float), np.asarray(xlim, float)
        self.f_dynamics = robot.f_kinematics
        self.base_radius = robot.base_radius
        self.obstacle_list = obstacle_list
        # the reference's xlim has 5 columns (x, y, dx, dy, dpsi); psi is unbounded (mpc_base.py:16,155-156)
        xl = np.array([[self.xlim[0, 0], self.xlim[0, 1], -INF, self.xlim[0, 2], self.xlim[0, 3], self.xlim[0, 4]],
                       [self.xlim[1, 0], self.xlim[1, 1], INF, self.xlim[1, 2], self.xlim[1, 3], self.xlim[1, 4]]])
        self._M = len(obstacle_list) if n_obstacles is None else int(n_obstacles)
        self._engine = _capi.Engine(_capi.KIND_BASE, N, self._M, self.dt, self.ulim, xl,
                                    np.array([[-INF, -INF], [INF, INF]]), max_batch=max_batch, device=device,
                                    obs_per_stage=obs_per_stage, tol=tol, max_iter=max_iter)
        self.max_batch = max_batch
        self.reset()

    def angleDiff(self, a, b):
        """mpc_base.py:56-94 on floats."""
        a = np.fmod(a + PI, 2 * PI) - PI
        b = np.fmod(b + PI, 2 * PI) - PI
        d = a - b
        if a * b >= 0:
            return d
        if a > b:
            return d if d <= PI else d - 2 * PI
        return d if d > -PI else d + 2 * PI

    def setWeight(self, Q=None, R=None, P=None, M=None):
        """mpc_base.py:96-112."""
        if Q is not None: self.Q_value = Q
        if R is not None: self.R_value = R
        if P is not None: self.P_value = P
        if M is not None: self.M_value = M
        self._engine.set_weights(self.Q_value, self.R_value, self.P_value, self.M_value, np.zeros((2, 2)))

    def reset(self):
        """mpc_base.py:114-189."""
        self.X_guess = None
        self.U_guess = None
        self._engine.reset()
        self.setWeight()

    def _obs_array(self, B):
        o = np.array([[ob.x, ob.y, ob.radius] for ob in self.obstacle_list], float).reshape(-1, 3)
        return np.broadcast_to(o, (B,) + o.shape).copy()

    def solve(self, x_init, traj_ref, u_ref):
        """mpc_base.py:191-226 (x_init is NOT clipped in this class, quirk Q13)."""
        r = self._engine.solve_batch(np.asarray(x_init, float).reshape(1, 6), np.asarray(traj_ref, float)[None],
                                     np.asarray(u_ref, float)[None], self._obs_array(1))
        if r["status"][0] != 0:
            raise RuntimeError("MPC solve failed: status %d after %d iterations" % (r["status"][0], r["iters"][0]))
        self.X_guess = r["X"][0]
        self.U_guess = r["U"][0]
        return self.U_guess[0, :]

    def solve_batch(self, x_init, traj_ref, u_ref, obs=None):
        x_init = np.array(x_init, float)
        if obs is None:
            obs = self._obs_array(x_init.shape[0])
        return self._engine.solve_batch(x_init, traj_ref, u_ref, obs)
