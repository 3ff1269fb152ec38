#!/usr/bin/env python
"""A/B of the fill rule for Winograd launches inside per-detection groups (aae_multi_impl.h: plan_multi, option multi_group_winograd): frames of 4 ... 16 classes x 1 ... 4
boxes under winograd_min_fill_pct = 56 (the default) / 50 / 40 / 30, wall time per frame from Python (50 frames).  Round 6 (profiles/r15/group_winograd_fill_ab.jsonl): equal
on 8 x 1 ... 4 x 4; at 16 x 1 a threshold below 56 is 45 % SLOWER (conv3 / conv4 take half-empty rounds of Winograd blocks) -- the single-object rule stands for groups too.
    python tools/group_winograd_fill_ab.py"""
import sys, os, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine, MultiObjectQuery
from augmentedautoencoder_amd.weights import EncoderConfig
objs = [(EncoderEngine(EncoderConfig(), synth.make_weights(seed=100 + o), max_batch=64), CodebookEngine(synth.make_codebook(92232, 128, seed=200 + o))) for o in range(16)]
def t(counts, pct, n_obj):
    for e, _ in objs: e.set_option('winograd_min_fill_pct', pct)
    x = torch.from_numpy(synth.make_crops(sum(counts), seed=1)).cuda()
    mq = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs[:n_obj], counts)])
    for _ in range(5): mq(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): mq(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 50 * 1e6
for rnd in range(2):
    for name, counts in (('8x1', [1]*8), ('8x2', [2]*8), ('4x4', [4]*4), ('16x1', [1]*16), ('4x2',[2]*4), ('mixed7', [1,1,2,1,1,1])):
        print(json.dumps({'frame': name, 'round': rnd, **{'us_fill_pct_%d' % p: round(t(counts, p, len(counts)), 1) for p in (56, 50, 40, 30)}}), flush=True)
