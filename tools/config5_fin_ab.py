#!/usr/bin/env python
"""Config 5 (368928 x 128 bf16) and the default fp32 codebook at B = 32 ... 256: the stand-alone arg-max query with the block partials merged by an
argmax_reduce launch (AAE_SCAN_AUTO) against the in-launch finish by the last row block to arrive (AAE_SCAN_AUTO_FIN), alternating on one box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import _lib, synth
from augmentedautoencoder_amd.engine import CodebookEngine


def time_us(fn, reps, warm=20):
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


for name, rows, dtype in (('config5_bf16_4x', 368928, 'bf16'), ('default_f32', 92232, 'f32')):
    cb = CodebookEngine(synth.make_codebook(rows, 128, seed=7), dtype=dtype)
    for B in (32, 64, 128, 256):
        z = torch.randn(B, 128, device='cuda') * 3.0
        out = {'what': 'argmax_reduce_launch_vs_in_launch_finish', 'codebook': name, 'B': B, 'reduce_launch_us': [], 'in_launch_finish_us': []}
        answers = {}
        for rnd in range(3):
            for key, mode in (('reduce_launch_us', _lib.AAE_SCAN_AUTO), ('in_launch_finish_us', _lib.AAE_SCAN_AUTO_FIN)):
                cb.set_scan_mode(mode)
                out[key].append(round(time_us(lambda: cb.nn(z, 1, 1), 300), 2))
                idx, sc = cb.nn(z, 1, 1)
                answers[key] = (idx.cpu().numpy().copy(), sc.cpu().numpy().copy())
        out['identical_answers'] = bool((answers['reduce_launch_us'][0] == answers['in_launch_finish_us'][0]).all() and
                                        (answers['reduce_launch_us'][1] == answers['in_launch_finish_us'][1]).all())
        print(json.dumps(out), flush=True)
    cb.close()
