#!/bin/bash
# A/B of alternative builds on the C5 bench (reference warm-start protocol figure only)
mkdir -p gpurun_out
for tag in "$@"; do
  lib=mobile-manipulator-mpc_amd/csrc/libmmpc_$tag.so
  [ "$tag" = base ] && lib=mobile-manipulator-mpc_amd/csrc/libmmpc.so
  MMPC_LIB=$PWD/$lib timeout -k 10 300 python bench.py --config c5 --steps 2 --warmup 2 --no-cpu > gpurun_out/abc5_$tag.json 2> gpurun_out/abc5_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/abc5_$tag.err; continue; }
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/abc5_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print("%-8s C5 value %.0f  ms/step %.1f  kernel ms/tick %s" % (sys.argv[1], d['value'], d['ms_per_step'], [round(x, 1) for x in d['roofline']['kernel_ms_per_tick']]))
PY
done
