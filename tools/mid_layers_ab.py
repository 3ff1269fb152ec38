#!/usr/bin/env python
"""Frames between the per-detection and the chip-filling regime: a few classes with 5 ... 16 boxes each.  aae_encode_nn_multi with mid-batch groups (round 6: per-LAYER choice --
the layers the group fills run as one Winograd launch across the objects, the others per object) against multi_mid_group = 0 (one call per class), wall time per frame.
    python tools/mid_layers_ab.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from augmentedautoencoder_amd import synth                                    # noqa: E402
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine, MultiObjectQuery   # noqa: E402
from augmentedautoencoder_amd.weights import EncoderConfig                    # noqa: E402

objs = [(EncoderEngine(EncoderConfig(), synth.make_weights(seed=100 + o), max_batch=64), CodebookEngine(synth.make_codebook(92232, 128, seed=200 + o))) for o in range(8)]


def t(counts, grouped):
    for e, _ in objs:
        e.set_option('multi_mid_group', int(grouped))
    x = torch.from_numpy(synth.make_crops(sum(counts), seed=1)).cuda()
    mq = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs[:len(counts)], counts)])
    for _ in range(5):
        out = mq(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        out = mq(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 40 * 1e6, mq.launches, out[0].clone()


for rnd in range(2):
    for name, counts in (('4x6', [6] * 4), ('4x8', [8] * 4), ('4x12', [12] * 4), ('2x16', [16] * 2), ('8x6', [6] * 8), ('3x{5,9,14}', [5, 9, 14]), ('8x16', [16] * 8)):
        a, la, za = t(counts, 1)
        b, lb, zb = t(counts, 0)
        err = float((za - zb).abs().max() / zb.abs().max())
        print(json.dumps({'frame': name, 'round': rnd, 'grouped_us': round(a, 1), 'grouped_launches': la, 'one_call_per_class_us': round(b, 1), 'ratio': round(a / b, 3), 'max_rel_latent_difference': err}), flush=True)
