#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r4b_gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r4b_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/probe_base.py > gpurun_out/r4b_base_c2.txt 2>&1; tail -3 gpurun_out/r4b_base_c2.txt
timeout -k 10 500 python tools/c5_seed_sweep.py 5 6 7 > gpurun_out/r4b_c5_seed_sweep.txt 2>&1 || { tail -5 gpurun_out/r4b_c5_seed_sweep.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r4b_c5_seed_sweep.txt
