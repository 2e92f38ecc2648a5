"""Re-export of the package's workload generator (mobile-manipulator-mpc_amd/synth.py, numpy only) under the name the tests use.
Loaded by file path: the package directory is not a Python identifier and importing the package itself needs torch."""
import importlib.util
import os

_p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mobile-manipulator-mpc_amd", "synth.py")
_spec = importlib.util.spec_from_file_location("mmpc_amd_synth", _p)
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
make_batch = _m.make_batch
make_c1_starts = _m.make_c1_starts
PI = _m.PI
