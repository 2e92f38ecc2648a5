#!/bin/bash
# instruction / cycle counters of two builds of the library on the same launch (tools/pmc_collect.sh for each MMPC_LIB)
for tag in "$@"; do
  lib=mobile-manipulator-mpc_amd/csrc/libmmpc_$tag.so
  [ "$tag" = base ] && lib=mobile-manipulator-mpc_amd/csrc/libmmpc.so
  export MMPC_LIB=$PWD/$lib
  bash tools/pmc_collect.sh gpurun_out/pmc_$tag > /dev/null 2>&1
  grep iters gpurun_out/pmc_$tag/p1.log | tail -1
  python3 - gpurun_out/pmc_$tag/summary.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["per_wave_per_iteration"]
print(sys.argv[2], {k: round(v, 1) for k, v in d.items()})
PY
done
