#!/usr/bin/env python
"""A/B of the grouped query's block -> (object, tile) mapping for 8 | 16 equal-sized objects: every object's blocks on ONE XCD
(option multi_xcd_affine = 1) against every object spread over all eight (0).  A B A B on one box; one JSON line per frame shape."""
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
for n_obj, d in ((8, 1), (8, 1), (8, 4), (16, 1)):
    xcat = torch.cat([xi[:d] for xi in xs[:n_obj]]).contiguous()
    out = {'objects': n_obj, 'detections_per_object': d, 'one_xcd_per_object_us': [], 'spread_over_xcds_us': []}
    ref = None
    for rnd in range(6):
        for key, v in (('one_xcd_per_object_us', 1), ('spread_over_xcds_us', 0)):
            for e, _ in objs:
                e.set_option('multi_xcd_affine', v)
            mq = MultiObjectQuery([(e, c, d) for e, c in objs[:n_obj]], device=dev)
            out[key].append(round(time_us(lambda: mq(xcat), 40), 1))
            z, idx, sc = [t.clone() for t in mq(xcat)]
            if ref is None:
                ref = (z, idx, sc)
            out['bit_identical'] = bool(out.get('bit_identical', True) and torch.equal(z, ref[0]) and torch.equal(idx, ref[1]) and torch.equal(sc, ref[2]))
    print(json.dumps(out), flush=True)
