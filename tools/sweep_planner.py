#!/usr/bin/env python
"""Per-layer kernel times of both implicit-GEMM families over a batch grid, each wave-tile shape of the wave-split-K kernel
forced in turn: the measurements the planner's cost model (plan_layer in csrc/aae_hip_impl.h) is fitted to and checked
against.  One JSON object per (batch, candidate) on stdout.  Usage: python tools/sweep_planner.py [B,B,...] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig

batches = [int(b) for b in sys.argv[1].split(',')] if len(sys.argv) > 1 else [5, 6, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64, 96, 128]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
enc = EncoderEngine(EncoderConfig(), synth.make_weights(seed=2024), max_batch=max(batches))
CANDS = {
    'auto': {},
    'wavek_32x32': {'planner_cost_model': 0, 'wavek': 1, 'wavek_balance': 0, 'wavek_max_tiles': 8192, 'wavek_tiny_max_tiles': 1 << 20, 'wavek_narrow_max_tiles': 1 << 20},
    'wavek_64x32': {'planner_cost_model': 0, 'wavek': 1, 'wavek_balance': 0, 'wavek_max_tiles': 8192, 'wavek_tiny_max_tiles': 0, 'wavek_narrow_max_tiles': 1 << 20},
    'wavek_64x64': {'planner_cost_model': 0, 'wavek': 1, 'wavek_balance': 0, 'wavek_max_tiles': 8192, 'wavek_tiny_max_tiles': 0, 'wavek_narrow_max_tiles': 0},
    'igemm': {'planner_cost_model': 0, 'wavek': 0, 'wavek_dense': 1},
}
DEFAULTS = {'planner_cost_model': 1, 'wavek': 1, 'wavek_balance': 1, 'wavek_max_tiles': 512, 'wavek_tiny_max_tiles': 64, 'wavek_narrow_max_tiles': 128, 'wavek_dense': 1}


def kernel_split(x):
    acc, order = {}, []
    for _ in range(reps):
        _, recs = enc.encode_timed(x)
        for i, (label, ms, _) in enumerate(recs):
            key = (i, label.split(' ')[0])
            if key not in acc:
                acc[key] = []
                order.append(key)
            acc[key].append(ms)
    return [(k[1], round(1e3 * sorted(acc[k])[len(acc[k]) // 2], 2)) for k in order]


def whole(x, n=30):
    for _ in range(3):
        enc.encode(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        enc.encode(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in batches:
    x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
    for name, opts in CANDS.items():
        for k, v in DEFAULTS.items():
            try:
                enc.set_option(k, v)
            except ValueError:
                pass
        for k, v in opts.items():
            try:
                enc.set_option(k, v)
            except ValueError:
                pass
        print(json.dumps({'what': 'planner_sweep', 'B': B, 'candidate': name, 'encode_us': round(whole(x), 1), 'kernels_us': kernel_split(x)}), flush=True)
