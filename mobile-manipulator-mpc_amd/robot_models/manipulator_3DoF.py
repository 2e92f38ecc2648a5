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

    def inverse_transformation(self, q_initial_guess, x_target):
        raise NotImplementedError("batched IK (manipulator_3DoF.py:79-133) is on the 'next' list (SURVEY 8f-3)")
