#!/bin/bash
for tag in "$@"; do
  echo "== $tag"
  MMPC_STAMP_LIB=$PWD/mobile-manipulator-mpc_amd/csrc/libmmpc_$tag.so MMPC_PROBE_N=30 MMPC_PROBE_M=8 MMPC_PROBE_DISTINCT=1 timeout -k 10 200 python3 tools/probe_stamps.py 2>&1 | grep -v amdgpu.ids | tail -1
done
