#!/usr/bin/env python
"""Whole-call time of the stand-alone B = 1..4 codebook query (back-to-back Python calls, HIP events) for the package found
first on sys.path -- run it from two source trees in turn to A/B library versions on one box.  Usage: python scan_ab.py <tag>"""
import json
import sys

import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine

tag = sys.argv[1] if len(sys.argv) > 1 else 'x'
cb = CodebookEngine(synth.make_codebook(92232, 128, seed=7))
if len(sys.argv) > 2:
    cb.set_scan_mode(int(sys.argv[2]))
z = torch.randn(4, 128, device='cuda')


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


out = {'tag': tag}
for b in (1, 2, 4):
    zb = z[:b].contiguous()
    out['B%d' % b] = [round(time_us(lambda: cb.nn(zb, 1, 1), 400), 2) for _ in range(5)]
print(json.dumps(out))
if hasattr(cb, 'nn_timed'):
    out2 = {'tag': tag + ' queued from C (aae_codebook_nn_timed, 200 reps)'}
    for b in (1, 2, 4):
        zb = z[:b].contiguous()
        cb.nn_timed(zb, 1, 1, reps=20)
        out2['B%d' % b] = [round(cb.nn_timed(zb, 1, 1, reps=200)[2] * 1e3, 2) for _ in range(5)]
    print(json.dumps(out2))
