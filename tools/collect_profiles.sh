#!/bin/bash
# Regenerates the per-round profile artifacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# kernel-trace stats and PMC passes are separate rocprofv3 runs (no --pmc together with trace domains).  Each step is
# chained with && : nothing is started on the GPU after a step that failed.
set -e
tag=${1:-r02}
o=gpurun_out/$tag
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "[1/8] C4 kernel trace"
rocprofv3 --kernel-trace --stats -d $o/kt -o run --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $o/kt.log 2>&1
cp $(find $o/kt -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
echo "[2/8] C4 FETCH_SIZE / WRITE_SIZE"
rocprofv3 --pmc FETCH_SIZE -d $o/pf -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $o/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/pw -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $o/pw.log 2>&1
python3 tools/pmc_traffic.py $o "mmpc_fast_kernel" $o/pmc_traffic.json $((8192 * 5784)) "bench.py --steps 3 --warmup 1 --no-cpu (B=8192, N=20, M=5)"
echo "[3/8] instruction-mix / MFMA counters (1024 identical instances)"
bash tools/pmc_collect.sh $o/pmc > /dev/null
cp $o/pmc/summary.json $o/pmc_mfma.json
echo "[4/8] C5 kernel trace"
mkdir -p $o/c5
rocprofv3 --kernel-trace --stats -d $o/c5/kt -o run --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras > $o/c5/kt.log 2>&1
cp $(find $o/c5/kt -name "*kernel_stats.csv" | head -1) $o/c5_kernel_stats.csv
echo "[5/8] C5 FETCH_SIZE / WRITE_SIZE"
rocprofv3 --pmc FETCH_SIZE -d $o/c5/pf -o run --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras > $o/c5/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/c5/pw -o run --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras > $o/c5/pw.log 2>&1
python3 tools/pmc_traffic.py $o/c5 "mmpc_fast_kernel" $o/c5_pmc_traffic.json $((8192 * 14336)) "bench.py --config c5 --steps 1 --warmup 2 --no-cpu (B=8192, N=30, M=8 moving obstacles, per tick)"
echo "[6/8] phase stamps"
if [ -f mobile-manipulator-mpc_amd/csrc/libmmpc_stamp.so ]; then python3 tools/probe_stamps.py > $o/phase_stamps.txt 2>&1; fi
python3 tools/probe_base.py > $o/base_c2.txt 2>&1 || true
# the bench line reads the PMC summaries of the same build from profiles/<tag>_*.json
cp $o/pmc_traffic.json profiles/${tag}_pmc_traffic.json; cp $o/pmc_mfma.json profiles/${tag}_pmc_mfma.json
cp $o/c5_pmc_traffic.json profiles/${tag}_c5_pmc_traffic.json
echo "[7/8] bench lines"
python3 bench.py --steps 20 --warmup 3 > $o/bench.json 2> $o/bench.err
echo "[8/8] C5 bench line"
python3 bench.py --config c5 --steps 2 --warmup 2 > $o/bench_c5.json 2> $o/bench_c5.err
tail -c 600 $o/bench.json
