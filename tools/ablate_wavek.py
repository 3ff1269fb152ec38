#!/usr/bin/env python
"""Where does a wave-split-K layer spend its time at B = 1?  Per-kernel HIP-event times with parts of the kernel
switched off (encoder option wavek_ablate: 1 no A loads, 2 no B loads, 4 no MFMAs, 8 no cross-block hand-off)."""
import _experiments  # noqa: F401  (the kernel variants compared here live in the experiments build: libaae_hip_experiments.so)
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig

enc = EncoderEngine(EncoderConfig(), synth.make_weights(seed=2024))
for B in (1, 4):
    x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
    for name, flags in (('full', 0), ('no A', 1), ('no B', 2), ('no A, no B', 3), ('no MFMA', 4), ('no hand-off', 8),
                        ('MFMA only', 3 + 8), ('loads only', 4 + 8), ('nothing', 15)):
        enc.set_option('wavek_ablate', flags)
        acc = {}
        for _ in range(12):
            _, recs = enc.encode_timed(x)
            for i, (l, ms, _) in enumerate(recs):
                acc.setdefault(i, []).append(ms * 1e3)
        print(json.dumps({'what': 'wavek_ablate', 'B': B, 'variant': name, 'kernels_us': [round(sorted(v)[len(v) // 2], 2) for _, v in sorted(acc.items())]}), flush=True)
enc.set_option('wavek_ablate', 0)

# ---- in-kernel timeline: shader-clock stamps of wave 0 of every block at 8 phase boundaries ----
import ctypes
import numpy as np
enc.set_option('wavek_timeline', 1)
PH = ['index math', 'K loop (loads+MFMA)', 'cross-wave LDS sum', 'partial stores issued', 'ticket', 'sum of partials (last block)', 'epilogue']
for B in (1, 4):
    x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
    for _ in range(5):
        enc.encode(x)
    buf = np.zeros((3, 512, 8), dtype=np.int64)
    enc.lib.aae_encoder_debug_timeline(enc.handle, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    _, recs = enc.encode_timed(x)
    for li in range(3):
        nblk = int((buf[li, :, 0] != 0).sum())
        t = buf[li, :nblk].astype(np.float64)
        if nblk == 0:
            continue
        t0 = t[:, 0].min()
        start = t[:, 0] - t0
        end = np.where(t[:, 7] > 0, t[:, 7], t[:, 5]) - t0          # non-last blocks of a split layer leave after the ticket
        row = {'what': 'wavek_timeline', 'B': B, 'layer': 'conv%d' % (li + 2), 'label': recs[li + 1][0].split(' ')[0], 'blocks': nblk,
               'ticks_per_us_assumed': 2400, 'block_start_spread_us': round(float(start.max()) / 2400, 2),
               'kernel_span_us': round(float(end.max()) / 2400, 2)}
        # phases present in every block: 0->1, 1->2, 2->3 ; split layers: 3->4, 4->5 ; last blocks: 5->6, 6->7
        def seg(a, b, mask=None):
            d = (t[:, b] - t[:, a])
            ok = (t[:, b] > 0) & (t[:, a] > 0)
            if mask is not None:
                ok &= mask
            d = d[ok]
            return None if d.size == 0 else [round(float(d.mean()) / 2400, 2), round(float(d.max()) / 2400, 2)]
        row['phases_us_mean_max'] = {'index math': seg(0, 1), 'K loop': seg(1, 2), 'LDS sum': seg(2, 3), 'partial stores': seg(3, 4),
                                     'ticket': seg(4, 5), 'sum partials (last block)': seg(5, 6), 'epilogue': seg(6, 7) if t[:, 4].max() > 0 else seg(3, 7)}
        print(json.dumps(row), flush=True)
