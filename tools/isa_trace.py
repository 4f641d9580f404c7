#!/usr/bin/env python3
"""Compact schedule view of one kernel in a hipcc -save-temps .s file: per basic block counts and, for the
block with the most MFMAs, one character per instruction (M mfma, e exp, r ds_read, G global/DMA, w waitcnt,
B barrier, . other VALU, s SALU). Usage: isa_trace.py file.s first_line last_line"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')[int(sys.argv[2]) - 1:int(sys.argv[3])]
hdr = [l for l in lines if re.search(r'; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize)', l)]
lines = [l.strip() for l in lines if l.strip() and not l.strip().startswith(';')]
blocks, cur, name = [], [], 'entry'
for l in lines:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name, cur = l, []
    else:
        cur.append(l)
blocks.append((name, cur))
def cls(op):
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('v_exp'): return 'e'
    if op.startswith('ds_read'): return 'r'
    if op.startswith('ds_'): return 'd'
    if op.startswith(('global', 'buffer', 'flat')): return 'G'
    if op.startswith('s_waitcnt'): return 'w'
    if op.startswith('s_barrier'): return 'B'
    if op.startswith('s_nop'): return 'n'
    if op.startswith('v_'): return '.'
    if op.startswith('s_'): return 's'
    return '?'
for name, b in blocks:
    t = ''.join(cls(l.split()[0]) for l in b)
    if len(b) > 30:
        print(name, len(b), {c: t.count(c) for c in 'Mer.Gws'})
big = max(blocks, key=lambda nb: sum(1 for l in nb[1] if l.startswith('v_mfma')))
t = ''.join(cls(l.split()[0]) for l in big[1])
print(big[0])
for i in range(0, len(t), 120):
    print(t[i:i + 120])
