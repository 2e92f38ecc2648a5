"""Builds csrc/libmmpc.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libmmpc.so")
SOURCES = ["mmpc_hip.hip"]
HEADERS = ["mmpc_core.h", "mmpc_tile.h", "mmpc_fast.h", "mmpc_fast_iter.inc", "mmpc_fast_a1r.inc", "mmpc_fast_a1s.inc", "mmpc_fast_d2.inc", "mmpc_ik.h", os.path.join("..", "..", "include", "mmpc.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_extension(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libmmpc.so (in-tree, so that it travels with the repo)."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    # -amdgpu-mfma-vgpr-form: the Riccati tiles stay in VGPRs (the pivots are read back by VALU instructions after every
    # rank-one MFMA; from AGPRs that is a v_accvgpr_read per word): +1.1 % on C4 and C5
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC", "-shared", "-o", LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB
