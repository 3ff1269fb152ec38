#!/usr/bin/env python
"""Config 5 (368928 x 128 bf16 rows) and the default fp32 codebook: whole-call time of the stand-alone top-1 query for B > 4 with
the queries normalised inside the scan (AAE_SCAN_AUTO) against the launch in front (AAE_SCAN_AUTO_PACKED), alternating A B A B on
one box.  Answers are compared as well.  One JSON line per (codebook, B)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import _lib, synth
from augmentedautoencoder_amd.engine import CodebookEngine


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


for name, rows, dtype in (('config5_bf16_4x', 368928, 'bf16'), ('default_f32', 92232, 'f32')):
    cb = CodebookEngine(synth.make_codebook(rows, 128, seed=7), dtype=dtype)
    for B in (8, 32, 128, 256):
        z = torch.randn(B, 128, device='cuda') * 3.0
        out = {'codebook': name, 'B': B, 'fused_us': [], 'packed_us': []}
        answers = {}
        for rnd in range(3):
            for key, mode in (('fused_us', _lib.AAE_SCAN_AUTO), ('packed_us', _lib.AAE_SCAN_AUTO_PACKED)):
                cb.set_scan_mode(mode)
                out[key].append(round(time_us(lambda: cb.nn(z, 1, 1), 300), 2))
                idx, sc = cb.nn(z, 1, 1)
                answers[key] = (idx.cpu().numpy().copy(), sc.cpu().numpy().copy())
        out['identical_answers'] = bool((answers['fused_us'][0] == answers['packed_us'][0]).all() and
                                        (answers['fused_us'][1] == answers['packed_us'][1]).all())
        print(json.dumps(out), flush=True)
    cb.close()
