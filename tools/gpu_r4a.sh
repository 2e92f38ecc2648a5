#!/bin/bash
# round 4, first GPU call: parity tests, headline, C4 / C5 seed sweeps of the new iteration (inertia correction, second-order correction)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_closed_loop_demo_state_machine > gpurun_out/r4a_gpu_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r4a_gpu_tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-extras > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err || { tail -5 gpurun_out/r4a_bench.err; exit 1; }
cat gpurun_out/r4a_bench.json | cut -c1-1500
timeout -k 10 300 python tools/seed_sweep.py --no-oracle 3 4 5 6 7 8 9 10 11 12 > gpurun_out/r4a_seed_sweep.txt 2>&1 || { tail -5 gpurun_out/r4a_seed_sweep.txt; exit 1; }
cat gpurun_out/r4a_seed_sweep.txt
timeout -k 10 500 python tools/c5_seed_sweep.py > gpurun_out/r4a_c5_seed_sweep.txt 2>&1 || { tail -5 gpurun_out/r4a_c5_seed_sweep.txt; exit 1; }
cat gpurun_out/r4a_c5_seed_sweep.txt
