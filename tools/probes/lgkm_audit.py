#!/usr/bin/env python
"""Audit of a kernel's LDS-load waits in hipcc's -S output: walks the listing in program order (fall-through approximation across labels), keeps the FIFO of outstanding
ds_read destinations, retires them at s_waitcnt lgkmcnt(N) (LDS returns in order), and reports every instruction that reads or writes a VGPR an outstanding load still owns.
    python tools/probes/lgkm_audit.py file.s mangled_kernel_name"""
import re, sys
s = open(sys.argv[1]).read(); name = sys.argv[2]
body = s[s.index(name + ':'):]; body = body[:body.index('.Lfunc_end')]
def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out
fifo = []          # (line_no, text, dest regs)
n_bad = 0
for ln, l in enumerate(body.split('\n')):
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') or t.endswith(':'): continue
    op = t.split()[0]; args = t[len(op):]
    if op == 's_waitcnt':
        m = re.search(r'lgkmcnt\((\d+)\)', t)
        if m:
            n = int(m.group(1))
            # scalar loads (s_load) also count in lgkmcnt and return out of order: any wait with outstanding s_loads is only safe at 0; ignored here (none in the loops)
            while len(fifo) > n: fifo.pop(0)
        continue
    parts = [a.strip() for a in args.split(',')]
    used = regs(args)
    owned = set().union(*[f[2] for f in fifo]) if fifo else set()
    clash = used & owned
    if clash:
        n_bad += 1
        owner = [f for f in fifo if f[2] & clash][0]
        print(f'line {ln}: `{t}` touches v{sorted(clash)} still owned by line {owner[0]} `{owner[1]}` ({len(fifo)} loads outstanding)')
    if op.startswith('ds_read') or op.startswith('ds_load'):
        fifo.append((ln, t, regs(parts[0])))
print('violations:', n_bad)
