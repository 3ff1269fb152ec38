#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size) duration statistics.  The three conv
layers of the encoder share one kernel symbol (conv_igemm_f32_kernel<...>), so the stock --stats
table averages conv2/conv3/conv4 together; the launch grid tells them apart
(conv2 = 1048576 threads, conv3 = 524288, conv4 = 131072 at B = 256).
Usage: python tools/trace_by_grid.py <..._kernel_trace.csv> > profiles/rNN/kernel_trace_by_grid.csv"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        grid = r.get('Grid_Size') or '%sx%sx%s' % (r.get('Grid_Size_X'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
        rows[(r['Kernel_Name'], grid)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
w = csv.writer(sys.stdout)
w.writerow(['Name', 'Grid', 'Calls', 'AverageNs', 'MinNs', 'MaxNs'])
for (name, grid), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, grid, len(d), '%.1f' % (sum(d) / len(d)), min(d), max(d)])
