"""Imports the package directory `mobile-manipulator-mpc_amd/` (not a valid identifier) as `mmpc_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, "mobile-manipulator-mpc_amd")


def load():
    if "mmpc_amd" in sys.modules:
        return sys.modules["mmpc_amd"]
    spec = importlib.util.spec_from_file_location("mmpc_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["mmpc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
