#!/usr/bin/env python
"""Top-1 query of 5 ... 64 latent codes: the query-resident scan with the rows of a tile over FOUR waves per query group (AAE_SCAN_AUTO)
against two (AAE_SCAN_AUTO_RH2, rounds 2-3), A B A B on one box, answers compared.  One JSON line per (codebook, B)."""
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


for name, rows, dtype in (('default_f32', 92232, 'f32'), ('config5_bf16_4x', 368928, 'bf16')):
    cb = CodebookEngine(synth.make_codebook(rows, 128, seed=7), dtype=dtype)
    for B in (5, 8, 16, 32, 48, 64):
        z = torch.randn(B, 128, device='cuda') * 3.0
        out = {'codebook': name, 'B': B, 'four_waves_us': [], 'two_waves_us': []}
        answers = {}
        for rnd in range(3):
            for key, mode in (('four_waves_us', _lib.AAE_SCAN_AUTO), ('two_waves_us', _lib.AAE_SCAN_AUTO_RH2)):
                cb.set_scan_mode(mode)
                out[key].append(round(time_us(lambda: cb.nn(z, 1, 1), 300), 2))
                idx, sc = cb.nn(z, 1, 1)
                answers[key] = (idx.cpu().numpy().copy(), sc.cpu().numpy().copy())
        out['identical_answers'] = bool((answers['four_waves_us'][0] == answers['two_waves_us'][0]).all() and
                                        (answers['four_waves_us'][1] == answers['two_waves_us'][1]).all())
        print(json.dumps(out), flush=True)
    cb.close()
