#!/bin/bash
# one GPU round: parity tests, bench line (run through gpurun from the repo root).  Steps are chained with && so that
# nothing is started on the GPU after a step that was killed or timed out.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -v -s > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || { echo "gpu tests rc=$rc"; exit $rc; }
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_cycle.json 2> gpurun_out/bench_cycle.err || { tail -5 gpurun_out/bench_cycle.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_cycle.json').read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f median kernel %.3f | hinted %.3f ms | two streams %.0f | cpu %.0f dX %.2e (>1e-6: %d) | %s" % (
    d['value'], d['ms_per_step'], d['median_kernel_ms'], d['schedule_hint']['hinted_ms'], d['two_streams']['value'],
    d['cpu_baseline']['value'], d['cpu_baseline']['max_abs_dX_vs_gpu'], d['cpu_baseline']['n_dX_above_1e-6'], d['solver']))
print("roofline frac %.4f" % d['roofline']['frac'])
PY
