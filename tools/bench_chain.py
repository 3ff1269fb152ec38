#!/usr/bin/env python
"""Per-detection latency of the fused query (aae_encode_nn), B = 1 ... 4: conv1 + ONE persistent launch (detect_chain.h)
against the six stand-alone launches, eager and as one HIP-graph replay; optional encoder options name=value,...
One JSON object per line.  Usage: python tools/bench_chain.py [reps] [opt=value,...]"""
import _experiments  # noqa: F401  (the kernel variants compared here live in the experiments build: libaae_hip_experiments.so)
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CapturedNearestNeighbour, CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig


def time_us(fn, reps, warm=10):
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


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
enc = EncoderEngine(EncoderConfig(), synth.make_weights(seed=2024), max_batch=4)
for kv in (sys.argv[2].split(',') if len(sys.argv) > 2 else []):
    name, value = kv.split('=')
    enc.set_option(name, int(value))
cb = CodebookEngine(synth.make_codebook(92232, 128, seed=7))
x = torch.from_numpy(synth.make_crops(4, seed=3)).cuda()
for B in (1, 2, 3, 4):
    xb = x[:B].contiguous()
    row = {'what': 'fused encode+nn', 'B': B}
    for chain in (1, 0):
        enc.set_option('detect_chain', chain)
        eager = time_us(lambda: enc.encode_nn(cb, xb, 1), reps)
        enc_only = time_us(lambda: enc.encode(xb), reps)
        cap = CapturedNearestNeighbour(enc, cb, B, force_graph=True)
        graph = time_us(lambda: cap.graph.replay(), reps)
        del cap
        row['chain' if chain else 'six_launches'] = {'eager_us': round(eager, 2), 'graph_replay_us': round(graph, 2), 'encoder_only_us': round(enc_only, 2)}
    enc.set_option('detect_chain', 0)
    row['speedup'] = round(row['six_launches']['eager_us'] / row['chain']['eager_us'], 3)
    print(json.dumps(row), flush=True)
