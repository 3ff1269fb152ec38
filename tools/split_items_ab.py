#!/usr/bin/env python
"""aae_encode_nn_multi: classes with a few boxes beyond four answered as items of <= 4 boxes inside the frame's per-detection group (encoder option multi_split_items, default 1)
against the same frame with the option off (mid-batch group / the class's own call).  The "split by the caller" rows list every class as items of <= 4 whatever its size: the
measurement behind the rule (5 ... 8 boxes; up to 12 for the frame's only class beyond four).   python tools/split_items_ab.py"""
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


def run(items, x, lib_split):
    for e, _ in objs:
        e.set_option('multi_split_items', int(lib_split))
    mq = MultiObjectQuery(items)
    for _ in range(5):
        out = mq(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        out = mq(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 40 * 1e6, mq.launches, out[0].clone(), out[1].clone()


def by_caller(counts):
    items = []
    for (e, c), n in zip(objs, counts):
        while n > 0:
            k = min(4, n)
            items.append((e, c, k))
            n -= k
    return items


frames = (('4x5', [5] * 4), ('4x6', [6] * 4), ('4x8', [8] * 4), ('2x8', [8, 8]), ('8x5', [5] * 8), ('8x6', [6] * 8), ('6x8', [8] * 6), ('8x{9,1,1,1,1,1,1,1}', [9, 1, 1, 1, 1, 1, 1, 1]),
          ('3x{5,9,14}', [5, 9, 14]), ('2x10', [10, 10]), ('4x10', [10] * 4), ('4x12', [12] * 4), ('2x12', [12, 12]), ('3x16', [16] * 3), ('8x10', [10] * 8), ('8x16', [16] * 8))
for name, counts in frames:
    x = torch.from_numpy(synth.make_crops(sum(counts), seed=1)).cuda()
    whole = [(e, c, n) for (e, c), n in zip(objs, counts)]
    off, l_off, z_off, i_off = run(whole, x, 0)
    on, l_on, z_on, i_on = run(whole, x, 1)
    rec = {'frame': name, 'option_off_us': round(off, 1), 'option_off_launches': l_off, 'option_on_us': round(on, 1), 'option_on_launches': l_on, 'on_over_off': round(on / off, 3),
           'max_rel_latent_difference': float((z_on - z_off).abs().max() / z_off.abs().max()), 'indices_equal': bool(torch.equal(i_on, i_off))}
    sp = by_caller(counts)
    c, l_c, _, _ = run(sp, x, 0)
    rec.update({'split_by_the_caller_us': round(c, 1), 'split_by_the_caller_items': len(sp)})
    print(json.dumps(rec), flush=True)
