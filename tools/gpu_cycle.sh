#!/bin/bash
# one GPU round: parity tests, bench line, phase stamps (run through gpurun from the repo root)
mkdir -p gpurun_out; timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_cycle.json 2> gpurun_out/bench_cycle.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_cycle.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['cpu_baseline']['max_abs_dX_vs_gpu'], d['solver'])"
if [ -f mobile-manipulator-mpc_amd/csrc/libmmpc_stamp.so ]; then timeout -k 10 300 python tools/probe_stamps.py > gpurun_out/stamps_cycle.txt 2>&1; cat gpurun_out/stamps_cycle.txt; fi
