#!/usr/bin/env python
"""Top-1 query of 5 ... 32 latent codes: the query-resident scan answering inside its own launch (AAE_SCAN_AUTO_FIN: the last row block
to arrive merges the block partials) against the arg-max reduce launch behind it (AAE_SCAN_AUTO), A B A B on one box, answers compared;
then the fused encoder + query call the same way.  One JSON line per case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import _lib, synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
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


modes = (('in_launch_us', _lib.AAE_SCAN_AUTO_FIN), ('reduce_launch_us', _lib.AAE_SCAN_AUTO))
for name, rows, dtype in (('default_f32', 92232, 'f32'), ('config5_bf16_4x', 368928, 'bf16')):
    cb = CodebookEngine(synth.make_codebook(rows, 128, seed=7, planted_duplicates=16), dtype=dtype)
    for B in (5, 8, 16, 32):
        z = torch.randn(B, 128, device='cuda') * 3.0
        out = {'codebook': name, 'B': B, 'in_launch_us': [], 'reduce_launch_us': []}
        answers = {}
        for rnd in range(3):
            for key, mode in modes:
                cb.set_scan_mode(mode)
                out[key].append(round(time_us(lambda: cb.nn(z, 1, 1), 300), 2))
                idx, sc = cb.nn(z, 1, 1)
                answers[key] = (idx.cpu().numpy().copy(), sc.cpu().numpy().copy())
        out['identical_answers'] = bool((answers['in_launch_us'][0] == answers['reduce_launch_us'][0]).all() and
                                        (answers['in_launch_us'][1] == answers['reduce_launch_us'][1]).all())
        print(json.dumps(out), flush=True)
    if dtype == 'f32':
        enc = EncoderEngine(EncoderConfig(), synth.make_weights(seed=2024), max_batch=32)
        for B in (8, 12, 16):
            x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
            out = {'fused_query_B': B, 'in_launch_us': [], 'reduce_launch_us': []}
            answers = {}
            for rnd in range(3):
                for key, mode in modes:
                    cb.set_scan_mode(mode)
                    out[key].append(round(time_us(lambda: enc.encode_nn(cb, x, 1), 200), 2))
                    _, idx, sc = enc.encode_nn(cb, x, 1)
                    answers[key] = (idx.cpu().numpy().copy(), sc.cpu().numpy().copy())
            out['identical_answers'] = bool((answers['in_launch_us'][0] == answers['reduce_launch_us'][0]).all() and
                                            (answers['in_launch_us'][1] == answers['reduce_launch_us'][1]).all())
            print(json.dumps(out), flush=True)
        enc.close()
    cb.close()
