#!/usr/bin/env python
"""Condense tools/ubench/wino_layer_time.hip (-DAAE_WINO_STAMPS) output: per (variant, layer, mapping) one line with the cycles per block
by segment and the unit timeline of one K-loop stage for the older (waves 0-3) and younger (waves 4-7) wave of a SIMD."""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        if not line.startswith('{'):
            continue
        r = json.loads(line)
        c = r.get('cycles_per_block')
        head = 'var %-3s %s S=%d  %.4f ms  frac %.3f' % (r.get('var', 0), r['layer'], r['xcd_cols'], r['ms'], r['mfma_frac_of_157'])
        if not c:
            print(head)
            continue
        s = c['sum']
        w = c['stage2_by_wave']
        old = [sum(w[i][k] for i in range(4)) / 4 for k in range(6)]
        yng = [sum(w[i][k] for i in range(4, 8)) / 4 for k in range(6)]
        print('%s | span %d: prologues %d kloops %d (mfma %d, eff %.3f) out-transforms %d store %d | stage cycles %s | waves0-3 units %s wait %d | waves4-7 units %s wait %d' % (
            head, c['span'], s['prologues'], s['k_loops'], s['mfma_cycles'], s['mfma_cycles'] / s['k_loops'], s['output_transforms'], c['final_store'],
            c['phase0_stage_cycles'][:4], [int(x) for x in old[:4]], old[4], [int(x) for x in yng[:4]], yng[4]))
