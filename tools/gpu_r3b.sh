#!/bin/bash
# round 3: GPU parity tests + default bench line (a-priori order), logs under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/gpu_tests.log 2>&1; rc=$?
grep -E "second-source|as written, 2048|certified|passed|failed|Error" gpurun_out/gpu_tests.log | tail -12
[ $rc -eq 0 ] || { tail -30 gpurun_out/gpu_tests.log; echo "gpu tests rc=$rc"; exit $rc; }
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_cycle.json 2> gpurun_out/bench_cycle.err || { tail -5 gpurun_out/bench_cycle.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_cycle.json').read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f median kernel %.3f | hinted %.3f ms | cpu %.0f dX %.2e | %s" % (
    d['value'], d['ms_per_step'], d['median_kernel_ms'], d['schedule_hint']['hinted_ms'],
    d['cpu_baseline']['value'], d['cpu_baseline']['max_abs_dX_vs_gpu'], d['solver']))
print("roofline frac %.4f" % d['roofline']['frac'])
print("c1:", json.dumps(d.get('c1_shape_generic_kernel')))
PY
