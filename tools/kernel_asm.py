#!/usr/bin/env python3
"""Extracts one kernel's ISA from the device assembly of mmpc_hip.hip (hipcc -S --cuda-device-only) into its own file:
   python tools/kernel_asm.py /tmp/mmpc.s _Z16mmpc_fast_kernelILi0ELi20ELi5ELi1ELb0ELi0E /tmp/kf.s"""
import sys
src, prefix, out = sys.argv[1:4]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
k = lines[start:end + 1]
open(out, 'w').write('\n'.join(k))
print(len(k), "lines;", sum('v_mfma' in l for l in k), "mfma;", sum('scratch_' in l for l in k), "scratch;",
      sum('v_accvgpr' in l for l in k), "accvgpr moves;", sum('s_waitcnt' in l for l in k), "waits")
