import numpy as np

A2, A3, A5, A6, A7 = 0.316, 0.0825, 0.384, 0.088, 0.107   # manipulator_3DoF.py:18-22


class ManipulatorPanda3DoF:
    """Numeric host mirror of robot_models/manipulator_3DoF.py (FK + joint integrator)."""

    def __init__(self, dt):
        self.dt = dt

    def forward_tranformation(self, q):
        """Planar 3-DoF Panda FK (manipulator_3DoF.py:10-77): returns (endpoint, joint2, joint3),
        each (x, 0, z) in the arm base frame.  The reference's expanded sin/cos products are the
        three rotating segments below (angles q1, q1-q2, q1-q2-q3)."""
        q = np.asarray(q, float).reshape(-1)
        a, b = q[0] - q[1], q[0] - q[1] - q[2]
        x2 = A2 * np.sin(q[0]) + A3 * np.cos(q[0])
        z2 = A2 * np.cos(q[0]) - A3 * np.sin(q[0])
        x3 = x2 - A3 * np.cos(a) + A5 * np.sin(a)
        z3 = z2 + A3 * np.sin(a) + A5 * np.cos(a)
        xe = x3 + A6 * np.cos(b) - A7 * np.sin(b)
        ze = z3 - A6 * np.sin(b) - A7 * np.cos(b)
        return np.array([xe, 0.0, ze]), np.array([x2, 0.0, z2]), np.array([x3, 0.0, z3])

    def f_kinematics(self, q, q_dot):
        """q + q_dot*dt (manipulator_3DoF.py:189-191; the reference mutates q in place, this does not)."""
        return np.asarray(q, float) + np.asarray(q_dot, float) * self.dt

    def inverse_transformation(self, q_initial_guess, x_target, device=0):
        """Joint angles that bring the endpoint to x_target = (x, 0, z) in the arm base frame, started at
        q_initial_guess (manipulator_3DoF.py:79-133; same objective :111 and joint box :123).  Raises ValueError when the
        solver does not reach a stationary point, as the reference does on an IPOPT failure (:124-125).  For a reachable
        target the solution set is a curve; the point returned is the one this solver reaches from the guess, which need
        not be IPOPT's (inverse kinematics)."""
        x_target = np.asarray(x_target, float).squeeze()
        q0 = np.asarray(q_initial_guess, float).squeeze()
        if x_target.shape[0] != 3:
            raise ValueError("Wrong target ")                       # :131 (4- and 6-component targets are `pass` upstream)
        assert x_target[1] == 0.0, "y should always be 0"          # :100
        r = self.inverse_transformation_batch(q0[None, :], x_target[None, [0, 2]], device=device)
        if r["status"][0] != 0:
            raise ValueError(f"No solution in joint space found for given target {x_target} in cartesian space")
        return r["q"][0]

    @staticmethod
    def inverse_transformation_batch(q_initial_guess, target_xz, device=0):
        """B independent IK problems in one launch: q0 (B,3), target_xz (B,2) -> dict(q, status, iters)."""
        from .. import _capi
        return _capi.ik_batch(q_initial_guess, target_xz, device=device)
