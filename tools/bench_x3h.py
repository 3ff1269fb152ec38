#!/usr/bin/env python
"""f32x3h (opt-in split precision) at B = 256: per-kernel HIP-event times and encode throughput for the 128 x 128 LDS-DMA igemm
and the 256 x 256 / 64 x 128-wave-tile kernel (option x3h_wide256), with a bit-identity check between the two.  JSON lines."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig

cfg = EncoderConfig()
enc = EncoderEngine(cfg, synth.make_weights(seed=2024), max_batch=256)
x = torch.from_numpy(synth.make_crops(256, seed=1234)).cuda()
enc.set_option('precision', 1)
ref = None
for rep in range(2):
    for wide in (0, 1):
        enc.set_option('x3h_wide256', wide)
        for _ in range(3):
            enc.encode(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            z = enc.encode(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        acc = {}
        for _ in range(5):
            _, recs = enc.encode_timed(x)
            for l, t, f in recs:
                acc.setdefault(l.split(' ')[0], []).append((t, f))
        ref = z.clone() if ref is None else ref
        print(json.dumps({'what': 'x3h', 'x3h_wide256': wide, 'rep': rep, 'encode_ms': round(ms, 4), 'crops_per_s': round(256 / ms * 1e3, 1),
                          'identical_to_first': bool(torch.equal(z, ref)),
                          'kernels': [(k, round(sum(t for t, _ in v) / len(v), 4), round(v[0][1] / (sum(t for t, _ in v) / len(v)) / 1e9, 1)) for k, v in acc.items()]}), flush=True)
