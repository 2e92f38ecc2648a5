#!/bin/bash
# launch-order probe (tools/probe_order.py) for each libmmpc_<tag>.so given; MMPC_SEEDS: C4 seeds (default 3)
for tag in "$@"; do
  lib=mobile-manipulator-mpc_amd/csrc/libmmpc_$tag.so
  [ "$tag" = base ] && lib=mobile-manipulator-mpc_amd/csrc/libmmpc.so
  for s in ${MMPC_SEEDS:-3}; do MMPC_LIB=$PWD/$lib timeout -k 10 200 python tools/probe_order.py $s 2>&1 | grep -v amdgpu.ids; done
done
