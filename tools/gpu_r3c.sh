#!/bin/bash
# bench line incl. the new extras + the two-rank rehearsals
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_cycle.json 2> gpurun_out/bench_cycle.err || { tail -20 gpurun_out/bench_cycle.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_cycle.json').read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f median kernel %.3f | hinted %.3f ms | cpu %.0f dX %.2e" % (
    d['value'], d['ms_per_step'], d['median_kernel_ms'], d['schedule_hint']['hinted_ms'], d['cpu_baseline']['value'], d['cpu_baseline']['max_abs_dX_vs_gpu']))
print("roofline frac %.4f" % d['roofline']['frac'])
print("pipelined:", json.dumps({k: v for k, v in d.get('pipelined_budget', {}).items() if k != 'note'}))
print("c5:", json.dumps(d.get('c5'))[:1500])
PY
timeout -k 10 900 python -m pytest tests/test_bench_ranks.py -m gpu -x -q > gpurun_out/bench_ranks.log 2>&1; rc=$?
tail -15 gpurun_out/bench_ranks.log
exit $rc
