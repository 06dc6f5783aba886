"""Innermost-loop census of a disassembled kernel (llvm-objdump -d --no-show-raw-insn output restricted
to one function): per loop, instruction mix and scratch (spill) traffic.  usage: loop_scan.py k.s"""
import re
import sys

ins = []
for l in open(sys.argv[1]):
    m = re.match(r'\s+(\S.*?)\s+// ([0-9A-Fa-f]+):', l)
    if m:
        ins.append((int(m.group(2), 16), m.group(1)))
addr2i = {a: i for i, (a, t) in enumerate(ins)}
loops = []
for i, (a, t) in enumerate(ins):
    m = re.match(r's_c?branch\S*\s+(\d+)', t)
    if m:
        off = int(m.group(1))
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + 4 * off
        if tgt <= a and tgt in addr2i:
            loops.append((addr2i[tgt], i))
loops.sort(key=lambda x: (x[1] - x[0]))
owner = [None] * len(ins)
for li, (s, e) in enumerate(loops):
    for k in range(s, e + 1):
        if owner[k] is None:
            owner[k] = li
from collections import Counter
stats = {}
for k, (a, t) in enumerate(ins):
    li = owner[k]
    c = stats.setdefault(li, Counter())
    op = t.split()[0]
    c['n'] += 1
    if op.startswith('scratch_store'): c['sst'] += 1
    elif op.startswith('scratch_load'): c['sld'] += 1
    elif op.startswith('global_load'): c['gld'] += 1
    elif op.startswith('global_store'): c['gst'] += 1
    elif op.startswith('ds_'): c['ds'] += 1
    elif op.startswith('v_pk_'): c['vpk'] += 1; c['valu'] += 1
    elif op.startswith('v_'): c['valu'] += 1
    elif op == 's_barrier': c['bar'] += 1
    if 'dpp' in t or 'permlane' in t: c['dpp'] += 1
print("loop  [start,end] own-insns valu pk dpp ds gld gst sld sst bar")
for li in sorted(stats, key=lambda x: (-1 if x is None else loops[x][0])):
    c = stats[li]
    if li is None:
        print("outside loops: n=%d sld=%d sst=%d" % (c['n'], c['sld'], c['sst']))
        continue
    s, e = loops[li]
    nest = sum(1 for (s2, e2) in loops if s2 <= s and e2 >= e) - 1
    print("%4d %s[%5d,%5d] n=%4d valu=%4d pk=%3d dpp=%3d ds=%3d gld=%3d gst=%3d sld=%3d sst=%3d bar=%d" %
          (li, "  " * nest, s, e, c['n'], c['valu'], c['vpk'], c['dpp'], c['ds'], c['gld'], c['gst'], c['sld'], c['sst'], c['bar']))
