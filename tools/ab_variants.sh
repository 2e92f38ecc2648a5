#!/bin/bash
# A/B of alternative builds of the library (MMPC_LIB): bench line without the CPU leg for each libmmpc_<tag>.so given
mkdir -p gpurun_out
for tag in "$@"; do
  lib=mobile-manipulator-mpc_amd/csrc/libmmpc_$tag.so
  [ "$tag" = base ] && lib=mobile-manipulator-mpc_amd/csrc/libmmpc.so
  MMPC_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/ab_$tag.err; continue; }
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print("%-8s value %.0f  ms/step %.3f  median kernel %.3f  conv %.4f" % (sys.argv[1], d['value'], d['ms_per_step'], d['median_kernel_ms'], d['solver']['converged_frac']))
PY
done
