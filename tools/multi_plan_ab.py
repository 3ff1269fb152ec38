#!/usr/bin/env python
"""A/B of the group plan's choices (plan_wavek_group): wave tile and K split of one conv layer forced, the other layers on the plan's own
choice; whole grouped frame timed.  One JSON line per (objects, detections, layer, shape, g)."""
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


def main():
    cfg = EncoderConfig()
    dev = torch.device('cuda', 0)
    n_max = 16
    objs = [(EncoderEngine(cfg, synth.make_weights(seed=2024 + i), device=dev, max_batch=64),
             CodebookEngine(synth.make_codebook(92232, 128, seed=7 + i), device=dev)) for i in range(n_max)]
    xs = [torch.from_numpy(synth.make_crops(4, seed=500 + i)).to(dev) for i in range(n_max)]
    cases = [(8, 1), (8, 2), (8, 4), (16, 1), (4, 1), (4, 4), (2, 1)]
    for n_obj, d in cases:
        xcat = torch.cat([xi[:d] for xi in xs[:n_obj]]).contiguous()

        def run(shape, g):
            for e, _ in objs:
                e.set_option('multi_force_shape', shape)
                e.set_option('multi_force_g', g)
            mq = MultiObjectQuery([(e, c, d) for e, c in objs[:n_obj]], device=dev)
            return time_us(lambda: mq(xcat), 30)
        base = run(0, 0)
        print(json.dumps({'objects': n_obj, 'detections': d, 'plan': 'auto', 'us': round(base, 1)}), flush=True)
        for layer in (1, 2, 3):
            for shape in (1, 2, 3):
                for g in ((0, 1, 2, 4, 8, 16) if layer == 3 else (0, 1, 2, 4)):
                    t = run(shape << (4 * (layer - 1)), g << (8 * (layer - 1)))
                    print(json.dumps({'objects': n_obj, 'detections': d, 'layer': 'conv%d' % (layer + 1), 'shape': ['', '32x32', '64x32', '64x64'][shape], 'g': g or 'auto',
                                      'us': round(t, 1), 'vs_auto': round(t - base, 1)}), flush=True)


if __name__ == '__main__':
    main()
