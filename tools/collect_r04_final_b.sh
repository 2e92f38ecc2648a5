#!/bin/bash
# round 4, final profile set, part B: C5 kernel trace / traffic / stamps, seed sweeps, bench lines (needs profiles/r04_pmc_*.json of part A)
set -e
tag=r04
o=gpurun_out/$tag
mkdir -p $o/c5
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "[5] C5 kernel trace"
rocprofv3 --kernel-trace --stats -d $o/c5/kt -o run --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras > $o/c5/kt.log 2>&1
cp $(find $o/c5/kt -name "*kernel_stats.csv" | head -1) $o/c5_kernel_stats.csv
echo "[6] C5 FETCH_SIZE / WRITE_SIZE"
rocprofv3 --pmc FETCH_SIZE -d $o/c5/pf -o run --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras > $o/c5/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/c5/pw -o run --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras > $o/c5/pw.log 2>&1
python3 tools/pmc_traffic.py $o/c5 "mmpc_fast_kernel" $o/c5_pmc_traffic.json $((8192 * 14336)) "bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras (B=8192, N=30, M=8 moving obstacles, per tick)"
cp $o/c5_pmc_traffic.json profiles/${tag}_c5_pmc_traffic.json
echo "[7] C5 phase stamps (coarse build)"
MMPC_STAMP_LIB=$PWD/mobile-manipulator-mpc_amd/csrc/libmmpc_stampc5.so MMPC_PROBE_N=30 MMPC_PROBE_M=8 MMPC_PROBE_DISTINCT=1 python3 tools/probe_stamps.py 2>&1 | grep -v amdgpu.ids > $o/c5_phase_stamps.txt || true
echo "[8] seeds"
for s in 3 4 5 6 7 8 9 10 11 12; do python3 tools/probe_order.py $s 2>&1 | grep -v amdgpu.ids; done > $o/seeds_3_12.txt; cat $o/seeds_3_12.txt
timeout -k 10 400 python3 tools/c5_seed_sweep.py 2>&1 | grep -v amdgpu.ids > $o/c5_seed_sweep.txt; cat $o/c5_seed_sweep.txt
timeout -k 10 400 python3 tools/seed_sweep.py --no-oracle $(seq 3 80) 2>&1 | grep -v amdgpu.ids > $o/seed_sweep.txt; tail -3 $o/seed_sweep.txt
echo "[9] bench lines"
python3 bench.py --steps 20 --warmup 3 > $o/bench.json 2> $o/bench.err
python3 bench.py --config c5 --steps 2 --warmup 2 > $o/bench_c5.json 2> $o/bench_c5.err
tail -c 900 $o/bench.json; echo; tail -c 500 $o/bench_c5.json
