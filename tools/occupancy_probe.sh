#!/bin/bash
# builds the (0, 10, 5) instantiation of the specialised kernel for one and for two waves per SIMD (see tools/occupancy_probe.py)
# and, on the GPU box (argument "run"), times both
cd "$(dirname "$0")/../mobile-manipulator-mpc_amd/csrc"
if [ "$1" != run ]; then
  for w in 1 2; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fPIC -shared "-DMMPC_FAST_LIST(X)=X(0, 10, 5, $w)" -o libmmpc_n10w$w.so mmpc_hip.hip &
  done
  wait
else
  cd ../..
  for w in 1 2; do MMPC_LIB=$PWD/mobile-manipulator-mpc_amd/csrc/libmmpc_n10w$w.so timeout -k 10 200 python tools/occupancy_probe.py || exit 1; done
fi
