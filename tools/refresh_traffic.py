#!/usr/bin/env python
"""profiles/traffic.json <- the round's own PMC passes.  Usage:
    python tools/refresh_traffic.py <pmc_summary.json of the default bench> [<pmc_summary.json of tools/prof_mix.py>] --tag r12
The fp32 entries (conv2 / conv3 / conv4 at B = 256) come from the kernels `conv_igemm_f32_kernel<false, true, false, L, true, *>`
(L = 1, 2, 3) of the first summary, the `scan` entry (B = 1 stream scan) from the second when given.  bench.py copies the dominant
kernel's entry into roofline.traffic and names this file's `source` as where it came from."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = 'r12'
argv = list(sys.argv[1:])
if '--tag' in argv:
    i = argv.index('--tag')
    tag = argv[i + 1]
    del argv[i:i + 2]
args = argv
path = os.path.join(ROOT, 'profiles', 'traffic.json')
t = json.load(open(path))
main = json.load(open(args[0]))
changed = {}
for name, layer in (('conv2', 1), ('conv3', 2), ('conv4', 3)):
    rows = [v for k, v in main.items() if k.startswith('conv_igemm_f32_kernel<false, true, false, %d, true' % layer) and 'hbm_side_bytes' in v]
    if rows:
        best = max(rows, key=lambda v: v.get('dispatches_seen', 0))
        t['f32'][name] = int(round(best['hbm_side_bytes'], -5))
        changed[name] = t['f32'][name]
    # matrix-pipe occupancy of the same kernels: SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles per SIMD (1024 SIMDs), GRBM_GUI_ACTIVE the
    # active cycles summed over the 8 XCDs -> busy fraction = busy / (GUI_ACTIVE / 8 * 1024); delivered clock = GUI_ACTIVE / 8 / duration
    busy = [v for k, v in main.items() if k.startswith('conv_igemm_f32_kernel<false, true, false, %d, true' % layer)
            and 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v.get('GRBM_GUI_ACTIVE', 0) > 0]
    if busy:
        b = max(busy, key=lambda v: v.get('dispatches_seen', 0))
        t.setdefault('mfma', {})[name] = {'mfma_busy_frac': round(b['SQ_VALU_MFMA_BUSY_CYCLES'] / (b['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0), 4),
                                          'delivered_GHz': round(b['GRBM_GUI_ACTIVE'] / 8.0 / b['avg_ns_under_pmc'], 3)}
# the Winograd layer kernels (the default conv2 ... conv4 from three quarters of a round of blocks on): keys 'convN/wino'; the grid tells the
# layers of the default net apart (geometry 0: conv2 = 4096 blocks, conv3 = 2048 at B = 256; geometry 1: conv4)
for name, kern, grid in (('conv2/wino', 'conv_wino_layer_kernel<0, false>', 4096 * 512), ('conv3/wino', 'conv_wino_layer_kernel<0, false>', 2048 * 512),
                         ('conv4/wino', 'conv_wino_layer_kernel<1, false>', None)):
    rows = [v for k, v in main.items() if k.startswith(kern) and (grid is None or k.endswith('grid=%d' % grid))]
    hb = [v for v in rows if 'hbm_side_bytes' in v]
    if hb:
        t['f32'][name] = int(round(max(hb, key=lambda v: v.get('dispatches_seen', 0))['hbm_side_bytes'], -5))
        changed[name] = t['f32'][name]
    busy = [v for v in rows if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v.get('GRBM_GUI_ACTIVE', 0) > 0]
    if busy:
        b = max(busy, key=lambda v: v.get('dispatches_seen', 0))
        t.setdefault('mfma', {})[name] = {'mfma_busy_frac': round(b['SQ_VALU_MFMA_BUSY_CYCLES'] / (b['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0), 4),
                                          'delivered_GHz': round(b['GRBM_GUI_ACTIVE'] / 8.0 / b['avg_ns_under_pmc'], 3)}
# the opt-in split-precision mode is power / clock limited: what clock did the chip deliver under its dominant kernel?
for name, kern in (('conv2', 'conv_igemm_x3h_wide_kernel<1, 1>'), ('conv3', 'conv_igemm_x3h_wide_kernel<1, 2>'), ('conv4', 'conv_igemm_x3h_dma_kernel<1, 3, 2>')):
    rows = [v for k, v in main.items() if k.startswith(kern) and 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v.get('GRBM_GUI_ACTIVE', 0) > 0]
    if rows:
        b = max(rows, key=lambda v: v.get('dispatches_seen', 0))
        t.setdefault('mfma_x3h', {})[name] = {'mfma_busy_frac': round(b['SQ_VALU_MFMA_BUSY_CYCLES'] / (b['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0), 4),
                                              'delivered_GHz': round(b['GRBM_GUI_ACTIVE'] / 8.0 / b['avg_ns_under_pmc'], 3)}
t['mfma_source'] = 'profiles/%s/pmc_summary.json: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); delivered_GHz = GRBM_GUI_ACTIVE / 8 / kernel duration under the PMC pass' % tag
t['f32_source'] = 'profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the default bench command in round %s: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch)' % (tag, tag)
if len(args) > 1:
    mix = json.load(open(args[1]))
    rows = [(k, v) for k, v in mix.items() if k.startswith('scan_stream') and '<1,' in k and 'hbm_side_bytes' in v]
    if rows:
        k, v = max(rows, key=lambda kv: kv[1].get('dispatches_seen', 0))
        t['scan']['hbm_side_bytes_warm'] = int(round(v['hbm_side_bytes'], -4))
        t['scan']['kernel_us_under_pmc_warm'] = round(v['avg_ns_under_pmc'] / 1e3, 2)
        t['scan']['source'] = 'profiles/%s_small/pmc_summary.json, kernel %s (rocprofv3 PMC passes of tools/prof_mix.py in round %s; not measured in the bench run)' % (tag, k, tag)
        changed['scan'] = t['scan']['hbm_side_bytes_warm']
json.dump(t, open(path, 'w'), indent=1)
print('traffic.json updated:', changed)
