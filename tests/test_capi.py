"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mmpc.h declares;
without a GPU the product path fails loudly (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols(mm):
    lib_path = mm.build_extension()
    assert os.path.exists(lib_path)
    hdr = open(os.path.join(ROOT, "include", "mmpc.h")).read()
    declared = set(re.findall(r"\b(mmpc_[a-z0-9_]+)\s*\(", hdr)) - {"mmpc_handle_s"}
    assert len(declared) >= 12
    L = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(L, name), name
    assert set(mm._capi.EXPORTS) == declared
    assert b"gfx950" in mm._capi.lib().mmpc_version()


def test_code_object_is_gfx950(mm):
    data = open(mm.build_extension(), "rb").read()
    assert b"gfx950" in data and b"mmpc_solve_kernel" in data


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(mm):
    robot = mm.MobileManipulator(0.1)
    with pytest.raises(RuntimeError, match="mmpc_create failed"):
        mm.MPCWholeBody(robot, [], [], N=20)


def test_missing_library_fails_loudly(mm, monkeypatch):
    monkeypatch.setattr(mm._capi, "_lib", None)
    monkeypatch.setattr(mm._capi, "LIB_PATH", "/nonexistent/libmmpc.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mm._capi.lib()


def test_host_mirrors_of_robot_models(mm):
    import numpy as np
    from oracle import nlp
    r = mm.MobileManipulator(0.1)
    rng = np.random.default_rng(0)
    for _ in range(20):
        x, u = rng.uniform(-2, 2, 9), rng.uniform(-1, 1, 5)
        assert np.abs(r.f_kinematics(x, u) - nlp.wholebody_f(x, u, 0.1)).max() < 1e-15
        for a, b in zip(r.forward_tranformation(x), nlp.wholebody_fk(x)):
            assert np.abs(a - b).max() < 1e-15
