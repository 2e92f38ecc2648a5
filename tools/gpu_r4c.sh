#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r4c_gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r4c_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err || { tail -5 gpurun_out/r4c_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4c_bench.json').read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f frac %.4f" % (d['value'], d['ms_per_step'], d['roofline']['frac']))
print(d['extras_status']); print(d['extras_summary'])
PY
