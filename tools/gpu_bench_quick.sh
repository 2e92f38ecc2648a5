#!/bin/bash
# bench line only (no CPU leg): python bench.py --no-cpu prints the timed-loop figures
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail -5 gpurun_out/bench_quick.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f median kernel %.3f | %s | frac %.4f" % (d['value'], d['ms_per_step'], d['median_kernel_ms'], d['solver'], d['roofline']['frac']))
PY
