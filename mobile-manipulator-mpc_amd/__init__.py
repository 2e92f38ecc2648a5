"""MI355X-native batched MPC solve engine - host-side mirror of the reference's controller API.

Drop-in counterpart of HsinyuG/mobile-manipulator-mpc's hot path only:
``controllers.mpc_wholebody_qref.MPCWholeBody`` / ``controllers.mpc_base.MPCBase`` with the
reference's ``reset()`` / ``solve(x_init, traj_ref, u_ref)`` / ``setWeight`` surface, plus
``solve_batch`` over thousands of independent instances.  All numerical work happens in the
HIP library ``csrc/libmmpc.so`` (C ABI: include/mmpc.h); there is no CPU fallback - importing
the controllers without the built library, or without a GPU, raises.

The directory name is not a Python identifier; import it through ``mmpc_loader.load()`` at the
repository root, which registers it as ``mmpc_amd``.
"""
from .robot_models.obstacles import Obstacles
from .robot_models.base import Base
from .robot_models.manipulator_3DoF import ManipulatorPanda3DoF
from .robot_models.mobile_manipulator import MobileManipulator
from .controllers.mpc_wholebody_qref import MPCWholeBody
from .controllers.mpc_base import MPCBase
from .controllers.mpc_wholebody import MPCWholeBody as MPCWholeBodyPoseRef
from . import _capi
from . import synth
from . import interface_wholebody_qref
from .interface_wholebody_qref import BatchedRecedingHorizon, Interface
from .fleet import DeviceFleet
from .build import build_extension

__all__ = ["Obstacles", "Base", "ManipulatorPanda3DoF", "MobileManipulator", "MPCWholeBody", "MPCBase", "MPCWholeBodyPoseRef",
           "build_extension", "_capi", "BatchedRecedingHorizon", "Interface", "interface_wholebody_qref", "DeviceFleet", "synth"]
