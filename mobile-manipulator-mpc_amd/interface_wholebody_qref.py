"""Host-side counterpart of the receding-horizon driver (interface_wholebody_qref.py) around the MPC engine:

* `Interface` - the reference's closed loop for ONE robot with its task state machine (move -> approach -> rotate ->
  move finish -> manipulate -> manipulate finish, :146-228), global plans (:247-295), local references (:353-410) and the
  no-simulator plant step (:143).  Same constructor and method names, so `demo_wholebody_qref.py` runs against it with
  `physical_sim=False`; the pybullet branch (`physical_sim=True`, :66-80, :412-) and the plots are out of scope.
* `BatchedRecedingHorizon` - B independent robots in lock-step through the 'move' phase, every tick one `solve_batch`
  (the C5 workload).
"""
import copy

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


class Interface:
    """Closed loop of interface_wholebody_qref.py:13-228 without the simulator.  `controller` is an `MPCWholeBody`
    (this package's or any object with the same surface).  `run(max_steps)` adds a tick cap (the reference loops until
    'manipulate finish')."""

    def __init__(self, dt, t_move, t_manipulate, x_start, global_pose_target, controller, physical_sim=False):
        if physical_sim:
            raise NotImplementedError("the pybullet simulation branch (simulation/albert_robot.py) is outside this package")
        self.dt = dt
        self.desired_t_move = t_move
        self.desired_t_manipulate = t_manipulate
        self.global_pose_target = np.asarray(global_pose_target, float)      # x y z psi of the endpoint (:21)
        self.local_pose_target = None
        self.working_radius = 0.6                                            # :23
        g = self.global_pose_target
        x_start = np.asarray(x_start, float)
        # base goal: working_radius behind the endpoint target along its heading, arm as at the start (:24-33)
        self.x_target = np.array([g[0] - self.working_radius * np.cos(g[3]), g[1] - self.working_radius * np.sin(g[3]), g[3],
                                  0, 0, 0, x_start[6], x_start[7], x_start[8]])
        self.x_start = x_start
        self.controller = controller
        self.physical_sim = False
        self.manipulator_pose_log, self.endpoint_relative_pos_log, self.x_log, self.u_log, self.flag_log = [], [], [], [], []
        self.mpc_step_counter = 0
        self.is_active = False
        self.traj_ref = None
        self.u_ref = None
        self.task_flag = 'move'
        self.current_state = None
        self.verbose = False

    # ---- main loop (:82-143) ------------------------------------------------------------------------------------
    def run(self, max_steps=None):
        self.current_state = np.array(self.x_start, float)
        self.task_flag = 'move'
        self.is_active = True
        self.mpc_step_counter = 0
        while self.is_active and (max_steps is None or self.mpc_step_counter < max_steps):
            self.timerCallback()
        return self.task_flag

    def timerCallback(self):
        self.mpc_step_counter += 1
        if self.verbose:
            print(self.mpc_step_counter, self.task_flag, self.current_state)
        self.x_log.append(copy.deepcopy(self.current_state))
        self.flag_log.append(self.task_flag)
        self.current_joints_pose = np.hstack([np.asarray(p, float).reshape(-1)
                                              for p in self.controller.robot_model.forward_tranformation(self.current_state)])
        self.manipulator_pose_log.append(copy.deepcopy(self.current_joints_pose))
        self.endpoint_relative_pos_log.append(np.asarray(
            self.controller.robot_model.manipulator.forward_tranformation(self.current_state[-3:])[0]).squeeze())
        self.is_active = self.stateMachineUpdate()
        if not self.is_active:
            return
        self.command = self.controller.solve(self.current_state, self.local_traj_ref, self.local_u_ref)       # :134
        self.u_log.append(copy.deepcopy(np.asarray(self.command)))
        self.current_state = np.asarray(self.controller.f_dynamics(self.current_state, self.command)).squeeze()  # :143

    # ---- task state machine (:146-228) ---------------------------------------------------------------------------
    def stateMachineUpdate(self):
        robot_status = True
        c, x = self.controller, self.current_state
        if self.task_flag == 'move' and self.traj_ref is None:
            self.globalPlan2D()
        if self.task_flag in ('move', 'approach'):
            goal = self.traj_ref[-1]
            if self.task_flag == 'move' and abs(x[0] - goal[0]) <= 2 and abs(x[1] - goal[1]) <= 2:
                self.task_flag = 'approach'
                N = c.N
                c.opti.subject_to(c.X[N, :2] == c.X_ref[N, :2])                      # hard terminal position (:166-167)
            if np.linalg.norm(x[0:2] - goal[0:2]) <= 0.2:
                self.task_flag = 'rotate'
                c.setWeight(P=np.diag([5, 5, 5, 0, 0, 1, 1, 1, 1]), Q=np.diag([5, 5, 5, 0, 0, 1, 1, 1, 1]))   # :175-178
            elif self.task_flag == 'move':
                self.calcLocalRefTraj([0, 1])
            else:
                self.calcLocalRefPose()
        if self.task_flag == 'rotate':
            goal = self.traj_ref[-1]
            if abs(c.angleDiff(x[2], goal[2])) <= 0.5 * np.pi / 180 and np.linalg.norm(x[0:2] - goal[0:2]) <= 0.01:
                self.task_flag = 'move finish'
            else:
                self.calcLocalRefPose()
        if self.task_flag == 'move finish':
            self.task_flag = 'manipulate'
            g = self.global_pose_target
            # endpoint target in the arm base frame: range + the 0.007 base-link offset, height above joint 1 (:205-209)
            self.local_pose_target = np.array([np.hypot(g[0] - x[0], g[1] - x[1]) + 0.007, 0.0, g[2] - (0.606 + 0.333)])
            self.globalPlanManipulator()
            c.setWeight(P=np.diag([500, 500, 500, 0, 0, 1, 1, 1, 1]), Q=np.diag([500, 500, 500, 0, 0, 1, 1, 1, 1]))   # :211-215
        if self.task_flag == 'manipulate':
            if np.linalg.norm(self.current_joints_pose[:3] - self.global_pose_target[:3]) <= 0.01:
                self.task_flag = 'manipulate finish'
                robot_status = False
            else:
                self.calcLocalRefTraj([6, 7, 8])
        return robot_status

    # ---- global plans (:247-295) ---------------------------------------------------------------------------------
    def globalPlan2D(self):
        tr, ur = global_plan_2d(self.x_start, self.x_target, self.desired_t_move, self.dt)
        self.traj_ref, self.u_ref = tr[0], ur[0]

    def globalPlanManipulator(self):
        q_goal = self.controller.robot_model.manipulator.inverse_transformation(self.current_state[-3:], self.local_pose_target)
        x_target = np.hstack((self.current_state[:6], q_goal))
        T = int(self.desired_t_manipulate / self.dt)
        self.traj_ref = np.linspace(self.current_state, x_target, T + 1)
        self.u_ref = np.zeros((T, 5))

    # ---- local references (:353-410) -----------------------------------------------------------------------------
    def calcLocalRefTraj(self, distance_index, different_space=False):
        if different_space:
            raise NotImplementedError("cartesian-space references belong to controllers/mpc_wholebody.py (pose reference)")
        lt, lu = calc_local_ref_traj(self.current_state, self.traj_ref[None], self.u_ref[None], self.controller.N, distance_index)
        self.local_traj_ref, self.local_u_ref = lt[0], lu[0]
        assert self.local_traj_ref.shape[0] == self.controller.N + 1 and self.local_u_ref.shape[0] == self.controller.N

    def calcLocalRefPose(self):
        lt, lu = calc_local_ref_pose(self.current_state, self.traj_ref[None], self.u_ref[None], self.controller.N,
                                     self.controller.angleDiff)
        self.local_traj_ref, self.local_u_ref = lt[0], lu[0]
