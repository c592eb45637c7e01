#!/usr/bin/env python3
"""Per-basic-block instruction counts of one kernel in a hipcc -S listing (experiment helper).
usage: isa_blocks.py LISTING.s SYMBOL_PREFIX [min_instrs]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 6
SLOW = ('v_mad_i64', 'v_mad_u64', 'v_mul_lo_u32', 'v_mul_hi')
bb = 'entry'; counts = {bb: dict(valu=0, salu=0, vmem=0, lds=0, slow=0, mov=0, br=[], all=0)}; order = [bb]
for i in range(start + 1, len(lines)):
    l = lines[i].strip()
    if l.startswith('.Lfunc_end'): break
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        bb = m.group(1); counts[bb] = dict(valu=0, salu=0, vmem=0, lds=0, slow=0, mov=0, br=[], all=0); order.append(bb); continue
    if not l or l.startswith(';') or l.startswith('.'): continue
    op = l.split()[0]; c = counts[bb]; c['all'] += 1
    if op.startswith('v_'):
        c['valu'] += 1
        if op.startswith(SLOW): c['slow'] += 1
        if op.startswith('v_mov_b32_e32'): c['mov'] += 1
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): c['br'].append(l.split()[-1])
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('global_') or op.startswith('buffer_'): c['vmem'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
tot = dict(valu=0, salu=0, vmem=0, slow=0)
for b in order:
    c = counts[b]
    for k in tot: tot[k] += c[k]
    if c['all'] >= mn: print(b, c['all'], 'valu', c['valu'], 'slow', c['slow'], 'mov', c['mov'], 'salu', c['salu'], 'vmem', c['vmem'], 'lds', c['lds'], c['br'])
print('total', tot)
