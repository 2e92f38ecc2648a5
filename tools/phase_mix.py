"""Static instruction mix per phase of the REGULAR copy of the specialised kernel's iteration loop.
usage: python tools/phase_mix.py full.s kernel.s      (full.s: hipcc -S -gline-tables-only output; kernel.s: tools/kernel_asm.py extract)
The Riccati rows (R1R2 / handover / R3 legs / Pstore) are per trip of the stage loop (two stages)."""
import re, sys, collections
full, kern = sys.argv[1], sys.argv[2]
files = {}
for line in open(full):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m: files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
PH = [(9, 158, 'E1 stage'), (159, 235, 'E1 pair'), (236, 300, 'filter'), (301, 361, 'conv'), (362, 421, 'A1 stage'), (422, 455, 'A1 pair'),
      (456, 468, 'R0'), (469, 485, 'R1R2'), (486, 492, 'handover'), (493, 543, 'R3 legs'), (544, 556, 'Pstore'), (557, 629, 'gains'),
      (630, 761, 'forward'), (762, 797, 'D1'), (798, 836, 'D2'), (837, 845, 'trial')]
# phase boundaries by marker comments of mmpc_fast_iter.inc (so the table follows the file)
marks = []
src = open(sys.argv[3] if len(sys.argv) > 3 else 'mobile-manipulator-mpc_amd/csrc/mmpc_fast_iter.inc').read().split('\n')
def find(s, after=0):
    for i, l in enumerate(src):
        if i >= after and s in l: return i + 1
    raise KeyError(s)
b = [('E1 stage', find('MMPC_TS(0)')), ('E1 pair', find('MMPC_TS(1)')), ('filter', find('if (in_ls) {')), ('conv', find('#if MMPC_ITER_SOC\n') if False else find("if (soc_st == 1) soc_rows(true")),
     ('A1 stage', find('MMPC_TS(2)')), ('A1 pair', find('MMPC_TS(3)')), ('R0', find('MMPC_TS(4)')), ('R1R2', find('MMPC_TSF(5)')), ('handover', find('MMPC_TSF(6)')),
     ('R3 legs', find('// R3: the inputs are eliminated')), ('Pstore', find('MMPC_TSF(7)')), ('gains', find('if (ric_bad) failed = 1;')), ('forward', find('MMPC_TS(8)')),
     ('D1', find('MMPC_TS(9)')), ('D2', find('MMPC_TS(10)')), ('trial', find('MMPC_TS(11)')), ('end', len(src) + 1)]
def phase(f, l, cur):
    if f == 'mmpc_fast_iter.inc':
        for (n, lo), (_, hi) in zip(b[:-1], b[1:]):
            if lo <= l < hi: return n
    if f in ('mmpc_fast_a1s.inc', 'mmpc_fast_a1r.inc'): return 'A1 stage'
    if f == 'mmpc_fast_d2.inc': return 'D2'
    return cur
lines = open(kern).read().split('\n')
# the regular copy: from the first to the second occurrence of the E1 stage-lane block
starts = []
for i, line in enumerate(lines):
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m and files.get(m.group(1)) == 'mmpc_fast_iter.inc' and int(m.group(2)) in range(b[0][1] + 40, b[0][1] + 50) and (not starts or i - starts[-1] > 3000):
        starts.append(i)
lo, hi = starts[0] - 50, (starts[1] - 50 if len(starts) > 1 else len(lines))
cur = '?'; apply_ = False
C = collections.OrderedDict()
for i in range(lo, hi):
    line = lines[i]
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        f = files.get(m.group(1), ''); l = int(m.group(2))
        cur = phase(f, l, cur); continue
    if not re.match(r'\s+(v_|s_|ds_|global_|scratch_|buffer_)', line): continue
    c = C.setdefault(cur, collections.Counter()); c['instr'] += 1
    for key, pat in (('acc', 'v_accvgpr'), ('wait', 's_waitcnt'), ('nop', 's_nop'), ('rcp', 'v_rcp|v_rsq'), ('rdl', 'v_readlane'), ('dpp', '_dpp'),
                     ('mov', r'v_mov_b'), ('cnd', 'v_cndmask'), ('imul', 'v_mul_lo|v_mul_hi')):
        if re.search(pat, line): c[key] += 1
    if re.match(r'\s+ds_', line): c['lds'] += 1
    if re.match(r'\s+s_', line): c['salu'] += 1
tot = 0
for k, v in C.items():
    print("%-10s %5d  " % (k, v['instr']) + " ".join("%s %d" % (q, v[q]) for q in ('acc', 'salu', 'lds', 'wait', 'nop', 'rcp', 'rdl', 'dpp', 'mov', 'cnd', 'imul') if v[q])); tot += v['instr']
print("static total of the regular copy:", tot)
