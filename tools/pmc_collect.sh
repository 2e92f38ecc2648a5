#!/bin/bash
# PMC passes over one launch of 1024 identical instances (tools/probe_one.py); each pass is its own rocprofv3 run
# with --pmc only.  Usage (on the GPU box, repo root):  bash tools/pmc_collect.sh gpurun_out/pmc
set -e
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L 2>/dev/null | grep -o -i "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > "$out/mfma_counters.txt" || true
pass() { n=$1; shift; rocprofv3 --pmc "$@" -d "$out/p$n" -o run --output-format csv -- python3 tools/probe_one.py 1024 same > "$out/p$n.log" 2>&1; }
pass 1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64
pass 2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass 3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pass 4 SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES
python3 tools/pmc_summary.py "$out" > "$out/summary.json"
cat "$out/summary.json"
