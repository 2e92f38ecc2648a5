import numpy as np

from .base import Base
from .manipulator_3DoF import ManipulatorPanda3DoF


class MobileManipulator:
    """Numeric host mirror of robot_models/mobile_manipulator.py:9-75."""

    def __init__(self, dt):
        self.dt = dt
        self.base = Base(dt)
        self.manipulator = ManipulatorPanda3DoF(dt)
        self.baselink2joint1_x = -0.007           # mobile_manipulator.py:14 (kept: quirk Q11)
        self.baselink2joint1_z = 0.606 + 0.333    # mobile_manipulator.py:15

    def forward_tranformation(self, state):
        """(pose_endpoint[4], pos_joint_2[3], pos_joint_3[3]) in world coordinates (:17-55)."""
        state = np.asarray(state, float).reshape(-1)
        x, q = state[:6], state[6:]
        e, j2, j3 = self.manipulator.forward_tranformation(q)
        c, s = np.cos(x[2]), np.sin(x[2])

        def lift(p):
            return [x[0] + (p[0] + self.baselink2joint1_x) * c, x[1] + (p[0] + self.baselink2joint1_x) * s,
                    p[2] + self.baselink2joint1_z]
        return np.array(lift(e) + [x[2]]), np.array(lift(j2)), np.array(lift(j3))

    def f_kinematics(self, x, u):
        """9-state step = base (x[:6],u[:2]) ++ arm (x[6:],u[2:])  (:57-75)."""
        x = np.asarray(x, float).reshape(-1)
        u = np.asarray(u, float).reshape(-1)
        return np.concatenate([self.base.f_kinematics(x[:6], u[:2]), self.manipulator.f_kinematics(x[6:], u[2:])])
