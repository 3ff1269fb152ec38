#!/usr/bin/env python
"""Soak run for the in-launch ticketed reductions of the per-detection path (conv_wavek partial tiles, dense GEMV chunk rows,
scan block partials): thousands of fused queries whose inputs change every time, through one long-lived pair of workspaces,
about half of them under a second stream that saturates HBM; every answer is compared bit for bit with a second engine that
runs the same query on its own workspaces.  A lost arrival would hang (run under `timeout`), a stale or early read of a partial
shows up as a mismatch.  Not part of the test suite (~1 GPU-minute):  python tools/soak_tickets.py [queries]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig


def main():
    queries = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    weights = synth.make_weights(seed=2024)
    E = synth.make_codebook(92232, 128, seed=7, planted_duplicates=16)
    ex, ey = EncoderEngine(EncoderConfig(), weights), EncoderEngine(EncoderConfig(), weights)
    cx, cy = CodebookEngine(E), CodebookEngine(E)
    side = torch.cuda.Stream()
    big_a = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
    big_b = torch.zeros(1 << 30, dtype=torch.uint8, device='cuda')
    rng = np.random.default_rng(1)
    pool = torch.from_numpy(synth.make_crops(256, seed=77)).cuda()
    bad, checks, t0 = 0, 0, time.time()
    per_B = {1: 0, 2: 0, 3: 0, 4: 0, 6: 0, 9: 0, 12: 0}       # (3, 6, 9, 12: layers whose last tiles are cut in K -- tickets indexed from the first cut tile)
    for it in range(queries):
        B = int(rng.choice([1, 1, 1, 2, 3, 4, 6, 9, 12]))
        sel = torch.from_numpy(rng.choice(256, B, replace=False)).cuda()
        x = pool[sel] ^ int(rng.integers(0, 256))                    # new pixels every query
        if it % 40 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    big_a.copy_(big_b)
        zx, ix, sx = ex.encode_nn(cx, x, 1)
        per_B[B] += 1
        if it % 4 == 0 or it == queries - 1:                         # comparing every query would serialise the queue
            zy, iy, sy = ey.encode_nn(cy, x, 1)
            checks += 1
            bad += int(not (torch.equal(zx, zy) and torch.equal(ix, iy) and torch.equal(sx, sy)))
    torch.cuda.synchronize()
    print(json.dumps({'what': 'soak_tickets', 'queries': queries, 'per_batch_size': per_B, 'checked': checks, 'mismatching_checks': bad,
                      'seconds': round(time.time() - t0, 1)}), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
