#!/usr/bin/env python
"""Compact view of a kernel's instruction stream from hipcc -S output: python tools/isa_stream.py file.s mangled_name [--loop]
M = MFMA, v = VALU, e = transcendental, d = LDS read, w = LDS write, g = global/LDS-DMA load, s = SALU, W = s_waitcnt, B = s_barrier, n = s_nop, a = accvgpr move, | = label, b = branch"""
import re, sys
s = open(sys.argv[1]).read(); name = sys.argv[2]
m = re.search(r'^' + re.escape(name) + r':', s, re.M); body = s[m.start():]; body = body[:body.index('.Lfunc_end')]
out = []; counts = {}
for l in body.split('\n')[1:]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if re.match(r'\.LBB\d+_\d+:', t): out.append('\n|' + t[:-1] + ' ')
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): c = 'M'
    elif op.startswith('v_accvgpr'): c = 'a'
    elif op in ('v_exp_f32', 'v_rcp_f32', 'v_log_f32', 'v_rsq_f32', 'v_sqrt_f32'): c = 'e'
    elif op.startswith('v_'): c = 'v'
    elif op.startswith('ds_read') or op.startswith('ds_load'): c = 'd'
    elif op.startswith('ds_'): c = 'w'
    elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('scratch_'): c = 'S' if op.startswith('scratch_') else 'g'
    elif op == 's_waitcnt': c = 'W'
    elif op == 's_barrier': c = 'B'
    elif op == 's_nop': c = 'n'
    elif op.startswith('s_cbranch') or op == 's_branch': c = 'b'
    elif op.startswith('s_'): c = 's'
    else: c = '?'
    out.append(c); counts[c] = counts.get(c, 0) + 1
print(''.join(out)); print(counts)
