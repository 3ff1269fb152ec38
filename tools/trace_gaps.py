#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV -> per kernel symbol: dispatches, duration (median / min) and the gap to the previous dispatch's end
(median): tells a device-bound launch train (gap ~ the 1-2 us dispatch floor) from a host-bound one.  python tools/trace_gaps.py <dir> [substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np

root, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
for path in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    per = defaultdict(lambda: {'dur': [], 'gap': [], 'period': []})
    prev_end, prev_start = None, None
    for r in rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        k = r['Kernel_Name'].replace('void ', '').replace('aae::', '')[:70]
        if sub in k:
            per[k]['dur'].append(e - s)
            if prev_end is not None:
                per[k]['gap'].append(s - prev_end)
                per[k]['period'].append(s - prev_start)
        prev_end, prev_start = e, s
    for k, v in per.items():
        d, g, p = np.asarray(v['dur']), np.asarray(v['gap'] or [0]), np.asarray(v['period'] or [0])
        print('%-70s n=%6d dur med %.2f min %.2f us | gap med %.2f us | period med %.2f us' % (k, len(d), np.median(d) / 1e3, d.min() / 1e3, np.median(g) / 1e3, np.median(p) / 1e3))
