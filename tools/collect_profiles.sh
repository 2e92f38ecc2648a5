#!/bin/bash
# Regenerates the per-round profile artifacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r01_f
# kernel-trace stats and PMC passes are separate rocprofv3 runs (no --pmc together with trace domains).
set -e
tag=${1:-r01_f}
o=gpurun_out/$tag
mkdir -p $o
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $o/kt -o run --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $o/kt.log 2>&1
cp $(find $o/kt -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
# HBM traffic: FETCH_SIZE and WRITE_SIZE in their own passes
rocprofv3 --pmc FETCH_SIZE -d $o/pf -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $o/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/pw -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $o/pw.log 2>&1
python3 - "$o" <<'PY'
import sys, glob, csv, json, os
o = sys.argv[1]
def per_launch(d, name):
    v = []
    for f in glob.glob(os.path.join(o, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "mmpc_fast_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                v.append(float(r["Counter_Value"]))
    return v
f, w = per_launch("pf", "FETCH_SIZE"), per_launch("pw", "WRITE_SIZE")
fm, wm = sorted(f)[len(f) // 2], sorted(w)[len(w) // 2]
json.dump({"kernel": "mmpc_fast_kernel<0,20,5>", "workload": "bench.py --steps 3 --warmup 1 --no-cpu (B=8192, N=20, M=5)",
           "collection": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (no trace domains), per launch",
           "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
           "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports half of a coalesced stream's bytes -> doubled "
                         "(the loads here are 8 B/lane, for which the guide gives no calibration: upper estimate); WRITE_SIZE as is",
           "traffic_bytes_per_launch": int((2 * fm + wm) * 1024), "algorithmic_bytes_per_launch": 8192 * 5784},
          open(os.path.join(o, "pmc_traffic.json"), "w"), indent=1)
PY
bash tools/pmc_collect.sh $o/pmc > /dev/null
cp $o/pmc/summary.json $o/pmc_mfma.json
if [ -f mobile-manipulator-mpc_amd/csrc/libmmpc_stamp.so ]; then python3 tools/probe_stamps.py > $o/phase_stamps.txt 2>&1; fi
python3 tools/probe_base.py > $o/base_c2.txt 2>&1 || true
# the bench line reads the PMC summaries of the same build from profiles/<tag>_*.json
cp $o/pmc_traffic.json profiles/${tag}_pmc_traffic.json; cp $o/pmc_mfma.json profiles/${tag}_pmc_mfma.json
python3 bench.py --steps 20 --warmup 3 > $o/bench.json 2> $o/bench.err
python3 bench.py --config c5 --steps 2 --warmup 1 > $o/bench_c5.json 2> $o/bench_c5.err
tail -1 $o/bench.json
