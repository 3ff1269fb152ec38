#!/usr/bin/env python
"""Compile the library with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs / AGPRs / SGPR + VGPR spills / scratch / occupancy / LDS.  Usage: python tools/kernel_resources.py [regex] [-o out.so] [--reuse]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else None
out = sys.argv[sys.argv.index('-o') + 1] if '-o' in sys.argv else '/tmp/aae_resources_build.so'
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
       os.path.join(ROOT, 'augmentedautoencoder_amd', 'csrc', 'aae_hip.hip'), '-o', out, '-Rpass-analysis=kernel-resource-usage']
log = '/tmp/aae_resources.log'
if '--reuse' in sys.argv and os.path.exists(log):          # summarise the remarks of the previous build again
    err = open(log).read()
else:
    err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE).stderr.decode()
    open(log, 'w').write(err)
cur, rows = None, []
for line in err.splitlines():
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        if 'error' in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows).encode(), stdout=subprocess.PIPE).stdout.decode().splitlines()
for r, n in zip(rows, names):
    n = n.replace('void ', '').replace('aae::', '')
    n = n[:n.find('(')] if '(' in n else n
    if pat and not pat.search(n):
        continue
    print('%-72s vgpr %3s agpr %3s sgpr-spill %3s vgpr-spill %3s scratch %4s occ %s lds %s' % (
        n[:72], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('SGPRs Spill', '?'), r.get('VGPRs Spill', '?'),
        r.get('ScratchSize [bytes/lane]', '?'), r.get('Occupancy [waves/SIMD]', '?'), r.get('LDS Size [bytes/block]', '?')))
