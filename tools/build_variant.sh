#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <tag> [hipcc flags...] -> csrc/libmmpc_<tag>.so with only the C4 kernel
# (0, 20, 5) and the plane-free static generic shape, so that a build takes a fraction of the full one.  MMPC_FULL=1: every kernel.
tag=$1; shift
cd "$(dirname "$0")/../mobile-manipulator-mpc_amd/csrc"
LISTS=(-D'MMPC_FAST_LIST(X)=X(0, 20, 5, 1)' -D'MMPC_STATIC_LIST(X)=X(0, 20, 5, 0, 0, 0)')
[ -n "$MMPC_FULL" ] && LISTS=()
[ -n "$MMPC_C5" ] && LISTS=(-D'MMPC_FAST_LIST(X)=X(0, 30, 8, 1)' -D'MMPC_STATIC_LIST(X)=X(0, 20, 5, 0, 0, 0)')
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fPIC -shared "${LISTS[@]}" "$@" -o libmmpc_$tag.so mmpc_hip.hip 2>/dev/null && echo "built libmmpc_$tag.so"
