#!/bin/bash
# round 4, final profile set, part A (run through gpurun from the repo root): GPU tests, C4 kernel trace, HBM traffic counters,
# instruction-mix / MFMA counters, phase stamps (coarse: no stamps inside the stage loop; fine: three per stage), base kind.
# Each step is chained: nothing is started on the GPU after a step that failed.
set -e
tag=r04
o=gpurun_out/$tag
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "[0] GPU tests"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $o/gpu_tests.log 2>&1 || { tail -20 $o/gpu_tests.log; exit 1; }
tail -2 $o/gpu_tests.log
echo "[1] C4 kernel trace"
rocprofv3 --kernel-trace --stats -d $o/kt -o run --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras > $o/kt.log 2>&1
cp $(find $o/kt -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
echo "[2] C4 FETCH_SIZE / WRITE_SIZE"
rocprofv3 --pmc FETCH_SIZE -d $o/pf -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $o/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/pw -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $o/pw.log 2>&1
python3 tools/pmc_traffic.py $o "mmpc_fast_kernel" $o/pmc_traffic.json $((8192 * 5784)) "bench.py --steps 3 --warmup 1 --no-cpu --no-extras (B=8192, N=20, M=5)"
echo "[3] instruction-mix / MFMA counters (1024 identical instances)"
bash tools/pmc_collect.sh $o/pmc > /dev/null
cp $o/pmc/summary.json $o/pmc_mfma.json
echo "[4] phase stamps"
{ echo "== coarse build (-DMMPC_STAMP -DMMPC_STAMP_COARSE: no stamps inside the stage loop; the whole backward pass incl. R0 and the gains is booked under 'gain back-substitution')";
  MMPC_PROBE_DISTINCT=1 python3 tools/probe_stamps.py 2>&1 | grep -v amdgpu.ids;
  echo "== fine build (-DMMPC_STAMP: three stamps per stage, ~5 k cycles per iteration of their own)";
  MMPC_STAMP_LIB=$PWD/mobile-manipulator-mpc_amd/csrc/libmmpc_stampf.so MMPC_PROBE_DISTINCT=1 python3 tools/probe_stamps.py 2>&1 | grep -v amdgpu.ids; } > $o/phase_stamps.txt
python3 tools/probe_base.py 2>&1 | grep -v amdgpu.ids > $o/base_c2.txt || true
cp $o/pmc_traffic.json profiles/${tag}_pmc_traffic.json; cp $o/pmc_mfma.json profiles/${tag}_pmc_mfma.json
tail -3 $o/phase_stamps.txt; cat $o/base_c2.txt
