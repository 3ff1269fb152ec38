#!/usr/bin/env python
"""A/B of the slabs in flight (2 | 3) of the grouped query's 64 x 32 layers (conv4 of few-detection frames: a cold weight stream of
26 MB per object against 5 us of MFMA work per object -- HBM-bound, so bytes in flight matter).  Whole frame timed, A B A B on one box."""
import _experiments  # noqa: F401  (the three-slab form lives in the experiments build)
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine, MultiObjectQuery
from augmentedautoencoder_amd.weights import EncoderConfig
from bench_multi import time_us

cfg = EncoderConfig()
dev = torch.device('cuda', 0)
objs = [(EncoderEngine(cfg, synth.make_weights(seed=2024 + i), device=dev, max_batch=64),
         CodebookEngine(synth.make_codebook(92232, 128, seed=7 + i), device=dev)) for i in range(16)]
xs = [torch.from_numpy(synth.make_crops(4, seed=500 + i)).to(dev) for i in range(16)]
for n_obj, d in ((8, 1), (4, 1), (16, 1), (8, 2)):
    xcat = torch.cat([xi[:d] for xi in xs[:n_obj]]).contiguous()
    out = {'objects': n_obj, 'detections_per_object': d, 'depth2_us': [], 'depth3_conv4_us': [], 'depth3_conv3_conv4_us': [], 'spread_64x32_us': []}
    ref = None
    for rnd in range(4):
        for key, v in (('depth2_us', 0), ('depth3_conv4_us', 3 << 8), ('depth3_conv3_conv4_us', (3 << 8) | (3 << 4)), ('spread_64x32_us', 0x10000)):
            for e, _ in objs:
                e.set_option('multi_force_depth', v)
            mq = MultiObjectQuery([(e, c, d) for e, c in objs[:n_obj]], device=dev)
            out[key].append(round(time_us(lambda: mq(xcat), 40), 1))
            z, idx, sc = [t.clone() for t in mq(xcat)]
            if ref is None:
                ref = (z, idx, sc)
            out['bit_identical'] = bool(out.get('bit_identical', True) and torch.equal(z, ref[0]) and torch.equal(idx, ref[1]))
    print(json.dumps(out), flush=True)
