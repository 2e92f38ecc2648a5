#!/bin/bash
# registers / spills / scratch / LDS of every kernel of the product library, from the code object's metadata
# usage: tools/kernel_meta.sh [extra hipcc flags]
cd "$(dirname "$0")/../mobile-manipulator-mpc_amd/csrc"
OUT=${MMPC_META_OUT:-/tmp/mmpc_meta.co}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only --no-gpu-bundle-output -c -o $OUT "$@" mmpc_hip.hip || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT | python3 -c '
import re,sys
t=sys.stdin.read()
for blk in re.split(r"\n\s*- \.agpr_count:", t)[1:]:
    g=lambda k: (re.search(r"\."+k+r":\s*(\S+)", blk) or [None,"?"])[1]
    print("%-78s vgpr %3s agpr %3s vspill %3s sspill %3s scratch %5s lds %6s" % (g("name")[:78], g("vgpr_count"), blk.split()[0], g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
