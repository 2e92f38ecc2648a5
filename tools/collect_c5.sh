#!/bin/bash
# C5 part of tools/collect_profiles.sh (kernel trace, PMC traffic, bench line)
set -e
tag=${1:-r02}
o=gpurun_out/$tag
mkdir -p $o/c5
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf $o/c5/kt $o/c5/pf $o/c5/pw
rocprofv3 --kernel-trace --stats -d $o/c5/kt -o run --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras > $o/c5/kt.log 2>&1
cp $(find $o/c5/kt -name "*kernel_stats.csv" | head -1) $o/c5_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $o/c5/pf -o run --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras > $o/c5/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/c5/pw -o run --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 2 --no-cpu --no-extras > $o/c5/pw.log 2>&1
python3 tools/pmc_traffic.py $o/c5 "mmpc_fast_kernel" $o/c5_pmc_traffic.json $((8192 * 14336)) "bench.py --config c5 --steps 1 --warmup 2 --no-cpu (B=8192, N=30, M=8 moving obstacles, per tick)"
cp $o/c5_pmc_traffic.json profiles/${tag}_c5_pmc_traffic.json
python3 bench.py --config c5 --steps 2 --warmup 2 > $o/bench_c5.json 2> $o/bench_c5.err
tail -c 400 $o/bench_c5.json
