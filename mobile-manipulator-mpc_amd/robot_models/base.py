"""Numeric host mirror of the differential-drive base model (reference: robot_models/base.py:6-31).  The copy that the
solver differentiates lives in the HIP kernels (mmpc_core.h: dynamics defects and the [A B] coefficient vector)."""
import numpy as np

_WHEEL_TRACK_HALF = 0.7 / 2 + 0.157      # base.py:9 (half length of the footprint)


class Base:
    base_width = 0.52                    # base.py:10
    base_length = 2 * _WHEEL_TRACK_HALF

    def __init__(self, dt):
        self.dt = float(dt)

    @staticmethod
    def base_radius():
        """radius of the disc the ground obstacles are inflated by (base.py:15)"""
        return 0.4

    def f_kinematics(self, x, u, limited_yaw=False):
        """One explicit-Euler step of (x, y, psi, dx, dy, dpsi) under (dV, dw): the pose integrates the world-frame
        velocities, the velocities turn with the yaw rate and accelerate along the heading (base.py:17-31).
        Accepts (6,) / (2,) or batches (..., 6) / (..., 2)."""
        x = np.asarray(x, float)
        u = np.asarray(u, float)
        flat = x.ndim == 1 or x.shape[-1] != 6
        if flat:
            x = x.reshape(-1)[None, :]
            u = u.reshape(-1)[None, :]
        pose, vel = x[..., :3], x[..., 3:6]
        heading = np.stack([np.cos(x[..., 2]), np.sin(x[..., 2])], axis=-1)
        spin = np.stack([-vel[..., 1], vel[..., 0]], axis=-1) * vel[..., 2:3]          # (-dy, dx) * dpsi
        nxt = np.concatenate([pose + self.dt * vel,
                              vel[..., :2] + self.dt * (u[..., 0:1] * heading + spin),
                              vel[..., 2:3] + self.dt * u[..., 1:2]], axis=-1)
        if limited_yaw:                                                                 # base.py:28-29 (never enabled upstream)
            nxt[..., 2] = np.fmod(nxt[..., 2] + np.pi, 2 * np.pi) - np.pi
        return nxt[0] if flat else nxt
