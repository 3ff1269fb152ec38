#!/usr/bin/env python
"""Eager latency of the fused per-detection query (aae_encode_nn) at B = 1 ... 4 (+ 8, 16) under named sets of encoder options,
alternating the variants (A B A B) so that box drift cancels.  Usage: python tools/latency_variants.py "base= tiny8=wavek_tiny_waves=8"
One JSON line per (B, variant)."""
import _experiments  # noqa: F401  (the kernel variants compared here live in the experiments build: libaae_hip_experiments.so)
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig

variants = []
for vv in (sys.argv[1] if len(sys.argv) > 1 else 'base=').split():
    name, _, opts = vv.partition('=')
    variants.append((name, [kv.split('=') for kv in opts.split(',') if kv]))
batches = [int(b) for b in (sys.argv[2].split(',') if len(sys.argv) > 2 else '1,2,3,4,8,16'.split(','))]
cfg = EncoderConfig()
w = synth.make_weights(seed=2024)
encs, cbs = {}, {}
E = synth.make_codebook(92232, 128, seed=7)
for name, opts in variants:
    e = EncoderEngine(cfg, w, max_batch=max(batches))
    c = CodebookEngine(E)
    for k, v in opts:
        if k == 'scan_mode':                  # (a codebook option: AAE_SCAN_* of include/aae_hip.h)
            c.set_scan_mode(int(v))
        else:
            e.set_option(k, int(v))
    encs[name], cbs[name] = e, c


def timeit(fn, reps, warm=10):
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
    acc = {name: [] for name, _ in variants}
    for rnd in range(4):
        for name, _ in variants:
            e, cb = encs[name], cbs[name]
            acc[name].append(timeit(lambda: e.encode_nn(cb, x, 1), 200))
    for name, _ in variants:
        _, recs = encs[name].encode_timed(x)
        print(json.dumps({'B': B, 'variant': name, 'encode+nn_us': [round(t, 2) for t in acc[name]], 'min_us': round(min(acc[name]), 2),
                          'kernels': [l.split(' ')[0] for l, _, _ in recs]}), flush=True)
