#!/usr/bin/env python
"""rocprofv3 kernel_stats CSVs -> one compact table (kernel, calls, average microseconds) per file."""
import csv
import sys

for f in sys.argv[1:]:
    print('== ' + f.split('/')[-1])
    total = 0.0
    for r in csv.DictReader(open(f)):
        n = r['Name'].replace('void aae::', '').replace('aae::', '')
        if n.startswith('__amd') or 'at::native' in n:
            continue
        n = n[:n.find('(')] if '(' in n else n
        us = float(r['AverageNs']) / 1e3
        total += us
        print('  %-64s %5s %8.2f us  (min %s)' % (n[:64], r['Calls'], us, r.get('MinNs', '?')))
    print('  sum of averages %.2f us' % total)
