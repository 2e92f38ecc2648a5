#!/bin/bash
# A/B of alternative builds of the library (MMPC_LIB): headline only (no extras, no CPU leg) for each libmmpc_<tag>.so given;
# tags that end in "stamp" run the phase-stamp probe instead
mkdir -p gpurun_out
for tag in "$@"; do
  lib=mobile-manipulator-mpc_amd/csrc/libmmpc_$tag.so
  [ "$tag" = base ] && lib=mobile-manipulator-mpc_amd/csrc/libmmpc.so
  case $tag in
    *stamp) echo "== $tag"; MMPC_STAMP_LIB=$PWD/$lib MMPC_PROBE_DISTINCT=1 timeout -k 10 200 python tools/probe_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stamps_$tag.txt ;;
    *) MMPC_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-extras > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/ab_$tag.err; continue; }
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s value %.0f  ms/step %.3f  median kernel %.3f  conv %.4f mean iters %.2f max %d" % (sys.argv[1], d['value'], d['ms_per_step'], d['median_kernel_ms'], d['solver']['converged_frac'], d['solver']['mean_iters'], d['solver']['max_iters']))
PY
    ;;
  esac
done
