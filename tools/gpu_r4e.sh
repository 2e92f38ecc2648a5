#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r4e_gpu_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r4e_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for s in 3 4 5 6 7 8 9 10 11 12; do python tools/probe_order.py $s 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r4e_seeds.txt; cat gpurun_out/r4e_seeds.txt
timeout -k 10 600 python tools/c5_seed_sweep.py > gpurun_out/r4e_c5_seed_sweep.txt 2>&1 || { tail -5 gpurun_out/r4e_c5_seed_sweep.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r4e_c5_seed_sweep.txt
