#!/usr/bin/env python
"""Per-kernel HIP-event times of the encoder at mid batch sizes (median of 10), with the layer's MFMA floor beside it.
Usage: python tools/mid_batch_split.py [B,B,...] [opt=value,...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig

batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else '5,6,8,10,12,16,24,32,48,64').split(',')]
cfg = EncoderConfig()
enc = EncoderEngine(cfg, synth.make_weights(seed=2024), max_batch=max(batches))
for kv in (sys.argv[2].split(',') if len(sys.argv) > 2 else []):
    k, v = kv.split('=')
    enc.set_option(k, int(v))
cb = CodebookEngine(synth.make_codebook(92232, 128, seed=7))


def timeit(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B in batches:
    x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
    acc, order = {}, []
    for _ in range(10):
        _, recs = enc.encode_timed(x)
        for i, (label, ms, flops) in enumerate(recs):
            key = (i, label.split(' ')[0])
            if key not in acc:
                acc[key] = ([], flops)
                order.append(key)
            acc[key][0].append(ms)
    rows = []
    for k in order:
        t = sorted(acc[k][0])[5] * 1e3
        floor = acc[k][1] / 157.3e12 * 1e6
        rows.append('%s %.1f us (mfma floor %.1f = %.2f)' % (k[1].replace('conv_wavek_f32_', 'wk').replace('conv_igemm_f32_', 'ig'), t, floor, floor / t if t else 0))
    print(json.dumps({'B': B, 'encode_us': round(timeit(lambda: enc.encode(x), 50), 1), 'encode+nn_us': round(timeit(lambda: enc.encode_nn(cb, x, 1), 50), 1),
                      'mfma_floor_us': round(B * 27.2, 1), 'kernels': rows}), flush=True)
