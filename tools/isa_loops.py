#!/usr/bin/env python
"""Loops of one kernel in a device assembly listing (hipcc -S --cuda-device-only): for every backward branch the body's
instruction mix -- MFMAs, vector-memory instructions, SGPR spill traffic (v_readlane / v_writelane), scratch accesses.
Usage: python tools/isa_loops.py <listing.s> <mangled kernel name prefix> [min MFMAs per loop]"""
import collections
import re
import sys

text = open(sys.argv[1]).read().splitlines()
prefix = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
start = next(i for i, l in enumerate(text) if l.startswith(prefix) and l.rstrip().split(':')[0].endswith(l.split(':')[0]) and ':' in l)
end = next(i for i in range(start, len(text)) if text[i].startswith('.Lfunc_end'))
lines = text[start:end]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
total = collections.Counter()
for l in lines:
    t = l.split()
    if t and (t[0].startswith('v_readlane') or t[0].startswith('v_writelane')):
        total['readlane/writelane in the whole kernel'] += 1
print('kernel %s: %d lines, %s' % (lines[0].split(':')[0], len(lines), dict(total)))
seen = set()
for i, l in enumerate(lines):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
        continue
    a = labels[m.group(1)]
    body = [x.split()[0] for x in lines[a:i + 1] if x.strip() and not x.strip().startswith((';', '.'))]
    c = collections.Counter()
    for op in body:
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op.startswith(('buffer_load', 'global_load', 'buffer_store', 'global_store')):
            c['vmem'] += 1
        elif op.startswith(('v_readlane', 'v_writelane')):
            c['readlane/writelane'] += 1
        elif op.startswith('scratch_'):
            c['scratch'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif op.startswith('s_barrier'):
            c['barrier'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
    if c['mfma'] >= min_mfma and (a, i) not in seen:
        seen.add((a, i))
        print('  loop %s (lines %d-%d): %s' % (m.group(1), a, i, dict(c)))
