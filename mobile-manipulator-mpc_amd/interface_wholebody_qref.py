"""Batched counterpart of the receding-horizon driver's planning / plant-step logic
(interface_wholebody_qref.py:100-143, 247-266, 353-410), without the simulator, the plots and the
task state machine.  B independent robots advance in lock-step; every tick is one `solve_batch`.
"""
import numpy as np


def global_plan_2d(x_start, x_target, t_move, dt, nu=5):
    """Interface.globalPlan2D (:247-266): straight line in state space, zero input reference.
    x_start, x_target: (B, nx) -> traj_ref (B, T+1, nx), u_ref (B, T, nu), T = int(t_move/dt)."""
    x_start = np.atleast_2d(np.asarray(x_start, float))
    x_target = np.atleast_2d(np.asarray(x_target, float))
    T = int(t_move / dt)
    traj = np.ascontiguousarray(np.transpose(np.linspace(x_start, x_target, T + 1), (1, 0, 2)))   # same call as :264
    return traj, np.zeros((x_start.shape[0], T, nu))


def calc_local_ref_traj(current_state, traj_ref, u_ref, N, distance_index=(0, 1)):
    """Interface.calcLocalRefTraj (:353-396): window of N+1 reference rows starting at the global-reference
    point nearest to the current state (first minimum, strict '<'), tail padded with the last row."""
    current_state = np.atleast_2d(np.asarray(current_state, float))
    di = np.asarray(distance_index)
    B, Tp1, nx = traj_ref.shape
    dist = np.linalg.norm(current_state[:, None, di] - traj_ref[:, :, di], axis=2)
    start = np.argmin(dist, axis=1)                      # first occurrence of the minimum, as the '<' loop
    idx = np.minimum(start[:, None] + np.arange(N + 1)[None, :], Tp1 - 1)
    uidx = np.minimum(start[:, None] + np.arange(N)[None, :], u_ref.shape[1] - 1)
    b = np.arange(B)[:, None]
    return traj_ref[b, idx], u_ref[b, uidx]


def calc_local_ref_pose(current_state, traj_ref, u_ref, N, angle_diff):
    """Interface.calcLocalRefPose (:398-410): hold the final reference, heading made continuous with the
    current heading (psi_ref := psi + angleDiff(psi_ref, psi))."""
    current_state = np.atleast_2d(np.asarray(current_state, float))
    B = current_state.shape[0]
    loc = np.repeat(traj_ref[:, -1:, :], N + 1, axis=1).copy()
    lu = np.repeat(u_ref[:, -1:, :], N, axis=1).copy()
    for b in range(B):
        psi = current_state[b, 2]
        loc[b, :, 2] = psi + angle_diff(traj_ref[b, -1, 2], psi)
    return loc, lu


class BatchedRecedingHorizon:
    """B closed loops in lock-step, physical_sim=False branch of Interface.timerCallback (:100-143):
    local reference -> controller.solve -> x <- f(x, u0)."""

    def __init__(self, controller, x_start, x_target, t_move=5.0, obs=None, obs_vel=None):
        self.c = controller
        self.dt = controller.dt
        self.N = controller.N
        self.x = np.array(x_start, float)
        self.traj_ref, self.u_ref = global_plan_2d(self.x, x_target, t_move, self.dt, nu=self.x.shape[1] - 4)
        self.obs = None if obs is None else np.array(obs, float)
        self.obs_vel = None if obs_vel is None else np.array(obs_vel, float)
        self.tick = 0
        self.x_log, self.u_log, self.iters_log = [self.x.copy()], [], []

    def obstacles_now(self):
        """static (B,M,3) or, with velocities, per-stage centres c + v (tick + k) dt  (B,N+1,M,3) - the build's
        definition of the moving-obstacle scenario (README.md:57,85-88 names the branch, no code in the snapshot)."""
        if self.obs_vel is None:
            return self.obs
        k = (self.tick + np.arange(self.N + 1))[None, :, None, None] * self.dt
        o = np.repeat(self.obs[:, None, :, :], self.N + 1, axis=1).copy()
        o[..., :2] += self.obs_vel[:, None, :, :] * k
        return o

    def step(self):
        loc, lu = calc_local_ref_traj(self.x, self.traj_ref, self.u_ref, self.N)
        r = self.c.solve_batch(self.x, loc, lu, self.obstacles_now())
        if (r["status"] != 0).any():
            raise RuntimeError("MPC solve failed for %d instances" % int((r["status"] != 0).sum()))
        u0 = r["u0"]
        xc = np.clip(self.x, self.c.xlim[0], self.c.xlim[1]) if self.x.shape[1] == 9 else self.x
        self.x = np.array([self.c.f_dynamics(xc[b], u0[b]) for b in range(self.x.shape[0])])   # :143
        self.tick += 1
        self.x_log.append(self.x.copy()); self.u_log.append(u0.copy()); self.iters_log.append(r["iters"].copy())
        return r

    def run(self, ticks):
        for _ in range(ticks):
            self.step()
        return np.array(self.x_log), np.array(self.u_log)
