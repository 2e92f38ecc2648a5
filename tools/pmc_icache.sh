#!/bin/bash
# instruction-cache counters of builds of the library on the bench batch (one rocprofv3 --pmc pass each)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for tag in "$@"; do
  lib=mobile-manipulator-mpc_amd/csrc/libmmpc_$tag.so
  [ "$tag" = base ] && lib=mobile-manipulator-mpc_amd/csrc/libmmpc.so
  export MMPC_LIB=$PWD/$lib
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS -d gpurun_out/pmci_$tag -o run --output-format csv -- python3 tools/probe_one.py 8192 > gpurun_out/pmci_$tag.log 2>&1
  python3 - gpurun_out/pmci_$tag $tag <<'PY'
import sys, glob, csv, collections, os
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "mmpc_fast_kernel" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print(sys.argv[2], {k: acc[k] / n[k] for k in acc}, "miss rate %.4f" % (acc.get("SQC_ICACHE_MISSES", 0) / max(acc.get("SQC_ICACHE_REQ", 1), 1)))
PY
done
