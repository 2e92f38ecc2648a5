import numpy as np


class Base:
    """Numeric host mirror of robot_models/base.py:6-31 (the symbolic copy lives in the HIP kernel)."""

    def __init__(self, dt):
        self.dt = dt
        self.base_length = 2 * (0.7 / 2 + 0.157)   # base.py:9
        self.base_width = 0.52                      # base.py:10

    def base_radius(self):
        return 0.4                                  # base.py:15

    def f_kinematics(self, x, u, limited_yaw=False):
        """Explicit-Euler diff-drive step with world-frame velocity states (base.py:17-31)."""
        x = np.asarray(x, float).reshape(-1)
        u = np.asarray(u, float).reshape(-1)
        dt = self.dt
        x_next = np.array([
            x[0] + dt * x[3],
            x[1] + dt * x[4],
            x[2] + dt * x[5],
            x[3] + dt * (u[0] * np.cos(x[2]) - x[4] * x[5]),
            x[4] + dt * (u[0] * np.sin(x[2]) + x[3] * x[5]),
            x[5] + dt * u[1],
        ])
        if limited_yaw:
            x_next[2] = np.fmod(x_next[2] + np.pi, 2 * np.pi) - np.pi
        return x_next
