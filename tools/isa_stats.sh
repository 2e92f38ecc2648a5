#!/bin/bash
# per-phase static instruction / scratch / AGPR-move / s_waitcnt counts of the (0,20,5) fast kernel
cd "$(dirname "$0")/../mobile-manipulator-mpc_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -S -o /tmp/mmpc3.s --cuda-device-only mmpc_hip.hip 2>/dev/null
awk '/^_Z16mmpc_fast_kernelILi0ELi20ELi5E/,/s_endpgm/' /tmp/mmpc3.s > /tmp/kf3.s
python3 - <<'PY'
import re,collections,bisect
files={}
for line in open('/tmp/mmpc3.s'):
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',line)
    if m: files[m.group(1)]=(m.group(3) or m.group(2))
last=None
sc=collections.Counter(); acc=collections.Counter(); tot=collections.Counter(); wt=collections.Counter()
for line in open('/tmp/kf3.s'):
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',line)
    if m:
        f=files.get(m.group(1),'').split('/')[-1]
        if f=='mmpc_fast.h': last=int(m.group(2))
        continue
    if 'scratch_' in line: sc[last]+=1
    if 'v_accvgpr' in line: acc[last]+=1
    if 's_waitcnt' in line: wt[last]+=1
    if re.match(r'\s+(v_|s_|ds_|global_|scratch_|buffer_)',line): tot[last]+=1
marks=[]
for i,l in enumerate(open('mmpc_fast.h'),1):
    for tag in ["load","slack / multiplier init","E1 (stage","E1 (pair","A1 (stage","A1 (pair","R0: full","R1: T_ext","R2: [F","R3/R4","R5: P_k","forward roll-out","D1: mult","D2: row","merit of a trial","filter line search","---- update","results"]:
        if ("// "+"-"*10 in l or "// ====" in l or "// ----" in l or "// R" in l) and tag in l: marks.append((i,tag))
marks.sort()
def ph(l):
    if l is None: return 'pre'
    j=bisect.bisect_right([m[0] for m in marks],l)-1
    return marks[j][1] if j>=0 else 'setup'
A=collections.defaultdict(lambda:[0,0,0,0])
for l,c in tot.items(): A[ph(l)][0]+=c
for l,c in sc.items(): A[ph(l)][1]+=c
for l,c in acc.items(): A[ph(l)][2]+=c
for l,c in wt.items(): A[ph(l)][3]+=c
for k,v in A.items(): print("%-26s instr %5d scratch %4d accvgpr %4d waitcnt %4d"%(k,*v))
PY
