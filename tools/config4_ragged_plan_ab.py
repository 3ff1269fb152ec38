#!/usr/bin/env python
"""A/B of the launch plan of the incomplete-block images in a mid-batch group (aae_multi_impl.h: plan_mid_ragged -> plan_wavek_group) on the config-4 frame
(256 crops over 8 objects, buckets {34, 26, 27, 32, 31, 32, 33, 41}: conv4 hands 12 images of 6 objects to one grouped wave-split-K launch).

    python tools/config4_ragged_plan_ab.py [--reps 30] [--rounds 3]

Variants: encoder options multi_force_shape / multi_force_g for conv4 (nibble / byte 2): wave tile 32 x 32 | 64 x 32 | 64 x 64 and the K cut; plus multi_mid_ragged = 0
(every image in the Winograd launch).  Alternating, one JSON line per (round, variant), a summary at the end."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from augmentedautoencoder_amd import synth                                    # noqa: E402
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine, MultiObjectQuery   # noqa: E402
from augmentedautoencoder_amd.weights import EncoderConfig                    # noqa: E402


def main():
    opts = dict(a[2:].split('=') for a in sys.argv[1:] if a.startswith('--') and '=' in a)
    reps, rounds = int(opts.get('reps', 30)), int(opts.get('rounds', 3))
    counts = [34, 26, 27, 32, 31, 32, 33, 41]
    objs = [(EncoderEngine(EncoderConfig(), synth.make_weights(seed=100 + o), max_batch=64), CodebookEngine(synth.make_codebook(92232, 128, seed=200 + o))) for o in range(8)]
    x = torch.from_numpy(synth.make_crops(sum(counts), seed=4321)).cuda()
    variants = [('default', {}), ('ragged_in_winograd_launch', {'multi_mid_ragged': 0})]
    for shape, name in ((1, '32x32'), (2, '64x32'), (3, '64x64')):
        for g in (0, 2, 3, 4, 6, 8):
            variants.append(('%s_g%s' % (name, g or 'auto'), {'multi_force_shape': shape << 8, 'multi_force_g': g << 16}))
    base = {'multi_mid_ragged': 1, 'multi_force_shape': 0, 'multi_force_g': 0}
    samples = {n: [] for n, _ in variants}
    for r in range(rounds):
        for name, o in (variants if r % 2 == 0 else variants[::-1]):
            for e, _ in objs:
                for k, v in dict(base, **o).items():
                    e.set_option(k, v)
            try:
                mq = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs, counts)])
                for _ in range(5):
                    mq(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    mq(x)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / reps * 1e3
                launches = mq.launches
            except Exception as err:                                           # (a forced plan the product build has no kernel for)
                print(json.dumps({'what': 'config4_ragged_plan_ab', 'variant': name, 'error': str(err)[:160]}), flush=True)
                continue
            samples[name].append(ms)
            print(json.dumps({'what': 'config4_ragged_plan_ab', 'round': r, 'variant': name, 'ms_per_frame': round(ms, 4), 'launches': launches}), flush=True)
    for name, _ in variants:
        if samples[name]:
            v = np.asarray(samples[name])
            print(json.dumps({'what': 'config4_ragged_plan_ab_summary', 'variant': name, 'ms_median': round(float(np.median(v)), 4), 'ms_min': round(float(v.min()), 4),
                              'crops_per_s': round(256 / float(np.median(v)) * 1e3, 1)}), flush=True)


if __name__ == '__main__':
    main()
