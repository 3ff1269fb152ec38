#!/usr/bin/env python
"""Small-batch (per-detection) latency on the MI355X box: encode + nearest neighbour for B = 1 ... 16, per-kernel
HIP-event times, the wave-split-K path against the 128 x 128 split-K path, kernel variants, eager vs one HIP-graph
replay.  One JSON object per line on stdout.

    python tools/bench_small.py [latency] [variants] [thresholds] [prof]
"""
import _experiments  # noqa: F401  (the kernel variants compared here live in the experiments build: libaae_hip_experiments.so)
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from augmentedautoencoder_amd import _lib
from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CapturedNearestNeighbour, CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig


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
    return e0.elapsed_time(e1) / reps


def kernel_split(enc, x, reps=10):
    acc, order = {}, []
    for _ in range(reps):
        _, recs = enc.encode_timed(x)
        for i, (label, ms, _) in enumerate(recs):
            key = (i, label.split(' ')[0])
            if key not in acc:
                acc[key] = []
                order.append(key)
            acc[key].append(ms)
    return [(k[1], round(1e3 * sorted(acc[k])[len(acc[k]) // 2], 2)) for k in order]     # median, microseconds


def set_small(enc, on):
    for name in ('wavek', 'gemv_ticket', 'wavek_dense'):
        enc.set_option(name, 1 if on else 0)


def main():
    what = sys.argv[1:] or ['latency', 'variants', 'thresholds']
    cfg = EncoderConfig()
    enc = EncoderEngine(cfg, synth.make_weights(seed=2024), max_batch=1024)
    cb = CodebookEngine(synth.make_codebook(92232, 128, seed=7))
    if 'latency' in what:
        for B in (1, 2, 3, 4, 6, 8, 12, 16, 32):
            x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
            row = {'what': 'latency', 'B': B}
            for tag, on in (('new', True), ('old', False)):
                set_small(enc, on)
                cb.set_scan_mode(_lib.AAE_SCAN_AUTO if on else (_lib.AAE_SCAN_STREAM_2L if B <= 4 else _lib.AAE_SCAN_AUTO))
                z = enc.encode(x)
                row[tag] = {'encode_us': round(1e3 * timeit(lambda: enc.encode(x), 100), 2),
                            'nn_us': round(1e3 * timeit(lambda: cb.nn(z, 1, 1), 200), 2),
                            'encode+nn_us': round(1e3 * timeit((lambda: enc.encode_nn(cb, x, 1)) if on else (lambda: cb.nn(enc.encode(x), 1, 1)), 100), 2),
                            'kernels_us': kernel_split(enc, x)}
                if B <= 4:
                    cap = CapturedNearestNeighbour(enc, cb, B, force_graph=True)
                    row[tag]['graph_replay_us'] = round(1e3 * timeit(lambda: cap.graph.replay(), 200), 2)
                    del cap
            set_small(enc, True)
            cb.set_scan_mode(_lib.AAE_SCAN_AUTO)
            row['speedup'] = round(row['old']['encode+nn_us'] / row['new']['encode+nn_us'], 3)
            print(json.dumps(row), flush=True)
    if 'variants' in what:
        for B in (1, 2, 4):
            x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
            for waves, depth, narrow in ((4, 3, 16), (4, 2, 16), (8, 2, 16), (4, 3, 0), (8, 2, 0), (4, 3, 64)):
                enc.set_option('wavek_waves', waves)
                enc.set_option('wavek_depth', depth)
                enc.set_option('wavek_narrow_max_tiles', narrow)
                print(json.dumps({'what': 'variant', 'B': B, 'waves': waves, 'depth': depth, 'narrow_max_tiles': narrow,
                                  'encode_us': round(1e3 * timeit(lambda: enc.encode(x), 100), 2),
                                  'kernels_us': kernel_split(enc, x)}), flush=True)
            enc.set_option('wavek_waves', 4)
            enc.set_option('wavek_depth', 2)
            enc.set_option('wavek_narrow_max_tiles', 128)
            enc.set_option('wavek_tiny_max_tiles', 64)
    if 'thresholds' in what:
        # where does the wave-split-K kernel stop paying?  per layer, B = 4 ... 64, both kernel families
        for B in (4, 6, 8, 12, 16, 24, 32, 48, 64, 128, 256):
            x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
            row = {'what': 'threshold', 'B': B}
            for tag, tiles in (('wavek_512', 512), ('wavek_off', 0)):
                enc.set_option('wavek_max_tiles', tiles)
                enc.set_option('wavek_dense', 1 if tiles else 0)
                row[tag] = kernel_split(enc, x, reps=6)
            enc.set_option('wavek_max_tiles', 256)
            enc.set_option('wavek_dense', 1)
            print(json.dumps(row), flush=True)
    if 'prof' in what:
        # a few launches of the B = 1 chain for rocprofv3 --kernel-trace --stats
        for B in (1, 4):
            x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
            for _ in range(30):
                cb.nn(enc.encode(x), 1, 1)
        torch.cuda.synchronize()
    enc.close()
    cb.close()


if __name__ == '__main__':
    main()
