#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one directory per pass, as
tools/gpu_run.sh, steps pmc_bench / prof:..., writes them) + the derived figures DESIGN.md quotes:
  hbm_bytes  = 2 * FETCH_SIZE_KB * 1024 ... see below (gfx950 correction of MI355X_MICROARCH.md:
               FETCH_SIZE counts 64-B units reported in KB of 32 B -> x2; WRITE_SIZE as is)
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD-cycles per CU-cycle basis) relative to GRBM_GUI_ACTIVE
Usage: python tools/pmc_summary.py gpurun_out/prof_r02 [--json profiles/r02/pmc_summary.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace('void ', '').replace('aae::', '')
    cut = name.find('(')
    return name[:cut] if cut > 0 else name


def main():
    root = sys.argv[1]
    out_json = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
    acc = defaultdict(lambda: defaultdict(list))        # kernel -> counter -> [per-dispatch values]
    dur = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, '*pmc[0-9]*', '**', '*counter_collection.csv'), recursive=True)):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row['Kernel_Name'])
                if not any(t in k for t in ('conv_', 'scan_', 'splitk', 'argmax', 'l2norm', 'upconv', 'topk', 'dense_gemv', 'detect_', 'wino_')):
                    continue
                k = '%s grid=%s' % (k, row['Grid_Size'])          # the conv layers share one kernel; the grid tells them apart
                acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
                dur[k].append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
    summary = {}
    for k in sorted(acc):
        c = {name: sum(v) / len(v) for name, v in acc[k].items()}
        c['dispatches_seen'] = max(len(v) for v in acc[k].values())
        c['avg_ns_under_pmc'] = sum(dur[k]) / len(dur[k])
        if 'FETCH_SIZE' in c or 'WRITE_SIZE' in c:
            # FETCH_SIZE / WRITE_SIZE are reported in KB; gfx950: wide (64-B) fetches are counted as one
            # 32-B unit -> fetch bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM/rocprofv3 section)
            c['hbm_side_bytes'] = 2.0 * c.get('FETCH_SIZE', 0.0) * 1024.0 + c.get('WRITE_SIZE', 0.0) * 1024.0
        if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c and c['TCC_HIT_sum'] + c['TCC_MISS_sum'] > 0:
            c['l2_hit_rate'] = c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum'])
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c and c['GRBM_GUI_ACTIVE'] > 0:
            # busy cycles are summed over the 256 CUs x 4 SIMDs... the counter is per-SE aggregated; report the raw ratio
            c['mfma_busy_per_gui_cycle'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['GRBM_GUI_ACTIVE']
        if 'SQ_WAVE_CYCLES' in c and 'SQ_WAIT_INST_ANY' in c and c['SQ_WAVE_CYCLES'] > 0:
            c['wait_inst_any_frac'] = c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']
        summary[k] = c
    for k, c in summary.items():
        keys = ['avg_ns_under_pmc', 'hbm_side_bytes', 'l2_hit_rate', 'mfma_busy_per_gui_cycle', 'wait_inst_any_frac', 'GRBM_GUI_ACTIVE',
                'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE']
        print(k[:70].ljust(70), ' '.join('%s=%.4g' % (n, c[n]) for n in keys if n in c))
    if out_json:
        os.makedirs(os.path.dirname(out_json), exist_ok=True)
        with open(out_json, 'w') as f:
            json.dump(summary, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
