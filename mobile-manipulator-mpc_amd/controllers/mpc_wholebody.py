import numpy as np

from .. import _capi

PI, INF = np.pi, np.inf


class MPCWholeBody:
    """Counterpart of controllers/mpc_wholebody.py:6-173 (the pose-reference whole-body controller) backed by the HIP
    engine, kind MMPC_KIND_WHOLEBODY_POSE: the state cost tracks the endpoint pose forward_tranformation(x)[0] =
    (x, y, z, psi) against X_ref (N+1,4) with 4x4 Q, P (:11-12,79-80,104-106); input / input-change / state boxes and the
    circle-obstacle rows as in the joint-reference controller; no self-collision or half-space rows (:100-101 TODO).
    `solve(x_init, traj_ref, u_ref)` keeps `x_guess` and `u_latest`, both used as the next initial guess (:134-139).

    Not reproduced: the state-box rows at stage 0 (:86) act on x_0 = x_init, a fixed quantity after the clip at :129-130 -
    they do not change the feasible set and are dropped (IPOPT relaxes such bounds; the solution is the same).
    """

    def __init__(self, robot, obstacle_list, N=10,
                 Q=5 * np.diag([1, 1, 1, 1]), P=50 * np.diag([1, 1, 1, 1]),
                 R=np.diag([0.1, 0.1, 0.0, 0.0, 0.0]), S=np.diag([1e5]), W=np.diag([0, 0, 1e-1, 1e-1, 1e-1]),
                 ulim=np.array([[-2, -PI, -1, -1, -1], [2, PI, 1, 1, 1]]),
                 xlim=np.array([[-100, -100, -INF, -2, -2, -PI, -PI / 2, -PI * 3 / 4, 0],
                                [100, 100, INF, 2, 2, PI, PI / 2, 0, PI]]),
                 dulim=np.array([[-INF, -INF, -0.5, -0.5, -0.5], [INF, INF, 0.5, 0.5, 0.5]]),
                 max_batch=1, device=0, n_obstacles=None, tol=1e-8, max_iter=2000):
        self.N = N
        self.Q, self.R, self.P, self.S, self.W = Q, R, P, S, W
        self.dt = robot.dt
        self.dulim, self.ulim, self.xlim = np.asarray(dulim, float), np.asarray(ulim, float), np.asarray(xlim, float)
        self.f_dynamics = robot.f_kinematics
        self.robot_model = robot
        self.base_radius = robot.base.base_radius()
        self.obstacle_list = obstacle_list
        if abs(self.base_radius - 0.4) > 0:
            raise ValueError("the kernels bake base_radius 0.4 (base.py:15)")
        self._M = len(obstacle_list) if n_obstacles is None else int(n_obstacles)
        self._engine = _capi.Engine(_capi.KIND_WHOLEBODY_POSE, N, self._M, self.dt, self.ulim, self.xlim, self.dulim,
                                    max_batch=max_batch, device=device, tol=tol, max_iter=max_iter)
        self.max_batch = max_batch
        self.reset()

    def reset(self):
        """mpc_wholebody.py:50-127: a fresh NLP - warm start cleared; the weights are the constructor's (they are
        constants of the graph upstream, :26-30)."""
        self.x_guess = None
        self.u_latest = None
        self._engine.reset()
        self._engine.set_weights(self.Q, self.R, self.P, self.S, self.W)

    def _obs_array(self, B):
        o = np.array([[ob.x, ob.y, ob.radius] for ob in self.obstacle_list], float).reshape(-1, 3)
        return np.broadcast_to(o, (B,) + o.shape).copy()

    def solve(self, x_init, traj_ref, u_ref):
        """mpc_wholebody.py:126-173: traj_ref (N+1,4) endpoint poses; returns U*[0] (5,)."""
        x_init[6:] = np.maximum(np.minimum(x_init[6:], self.xlim[1, 6:]), self.xlim[0, 6:]).squeeze()   # :129 (in place)
        x_init = np.maximum(np.minimum(x_init, self.xlim[1]), self.xlim[0]).squeeze()                    # :130
        assert x_init[7] <= 0 and x_init[8] >= 0                                                         # :131
        r = self._engine.solve_batch(np.asarray(x_init, float)[None], np.asarray(traj_ref, float)[None],
                                     np.asarray(u_ref, float)[None], self._obs_array(1))
        if r["status"][0] != 0:
            raise RuntimeError("MPC solve failed: status %d after %d iterations" % (r["status"][0], r["iters"][0]))
        self.x_guess = r["X"][0]
        self.u_latest = r["U"][0]
        return self.u_latest[0, :]

    def solve_batch(self, x_init, traj_ref, u_ref, obs=None):
        """B independent instances; traj_ref (B,N+1,4).  Returns dict(u0,X,U,s,status,iters,cost)."""
        x_init = np.array(x_init, float)
        x_init = np.maximum(np.minimum(x_init, self.xlim[1]), self.xlim[0])
        if obs is None:
            obs = self._obs_array(x_init.shape[0])
        return self._engine.solve_batch(x_init, traj_ref, u_ref, obs)
