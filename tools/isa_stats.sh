#!/bin/bash
# per-phase static instruction / LDS / scratch / AGPR-move / s_waitcnt counts of the (0,20,5) fast kernel (by source line ranges)
cd "$(dirname "$0")/../mobile-manipulator-mpc_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -gline-tables-only -S -o /tmp/mmpc3.s --cuda-device-only mmpc_hip.hip 2>/dev/null
python3 ../../tools/kernel_asm.py /tmp/mmpc3.s _Z16mmpc_fast_kernelILi0ELi20ELi5ELi1ELb0ELi0E /tmp/kf3.s
python3 - <<'PY'
import re,collections,bisect
files={}
for line in open('/tmp/mmpc3.s'):
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',line)
    if m: files[m.group(1)]=(m.group(3) or m.group(2))
last=None
C=collections.defaultdict(lambda: collections.Counter())
for line in open('/tmp/kf3.s'):
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',line)
    if m:
        f=files.get(m.group(1),'').split('/')[-1]
        if f=='mmpc_fast.h': last=int(m.group(2))
        continue
    if not re.match(r'\s+(v_|s_|ds_|global_|scratch_|buffer_)',line): continue
    C[last]['instr']+=1
    if 'scratch_' in line: C[last]['scratch']+=1
    if 'v_accvgpr' in line: C[last]['acc']+=1
    if 's_waitcnt' in line: C[last]['wait']+=1
    if re.match(r'\s+ds_',line): C[last]['lds']+=1
    if 'v_mfma' in line: C[last]['mfma']+=1
marks=[]
for i,l in enumerate(open('mmpc_fast.h'),1):
    m=re.match(r'\s*// (?:-{4,}|={4,})\s*(.{0,34})',l)
    if m and i>300: marks.append((i,m.group(1).strip()))
marks.sort()
def ph(l):
    if l is None: return 'pre'
    j=bisect.bisect_right([m[0] for m in marks],l)-1
    return marks[j][1] if j>=0 else 'setup'
A=collections.defaultdict(lambda: collections.Counter())
for l,c in C.items():
    A[ph(l)].update(c)
for k,v in A.items(): print("%-36s instr %5d lds %4d wait %4d acc %4d scratch %3d mfma %2d"%(k,v['instr'],v['lds'],v['wait'],v['acc'],v['scratch'],v['mfma']))
PY
