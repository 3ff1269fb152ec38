#!/usr/bin/env python
"""Where the time goes inside the persistent per-detection launch (detect_chain.h): option chain_timeline makes thread 0 of
every block stamp the 100 MHz wall clock at each phase edge; this prints, per edge, when the first / median / last block
passed it (microseconds after the first block started) over a few queries.  Usage: python tools/chain_timeline.py [B] [opt=value,...]"""
import _experiments  # noqa: F401  (the kernel variants compared here live in the experiments build: libaae_hip_experiments.so)
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from augmentedautoencoder_amd import _lib, synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
enc = EncoderEngine(EncoderConfig(), synth.make_weights(seed=2024), max_batch=4)
for kv in (sys.argv[2].split(',') if len(sys.argv) > 2 else []):
    name, value = kv.split('=')
    enc.set_option(name, int(value))
inner = int(os.environ.get('CHAIN_INNER_LAYER', '0'))          # 2, 3, 4: also the stamps inside that conv layer's phase
enc.set_option('detect_chain', 1)
enc.set_option('chain_timeline', inner if inner else 1)
cb = CodebookEngine(synth.make_codebook(92232, 128, seed=7))
x = torch.from_numpy(synth.make_crops(B, seed=3)).cuda()
names = ['start', 'conv2 work', 'arrive', 'release', 'conv3 work', 'arrive', 'release', 'conv4 work', 'arrive', 'release', 'dense work',
         'arrive', 'release', 'rows scanned', 'end']
G, S = 256, 40
buf = (ctypes.c_longlong * (3 * 512 * 8))()
rows, inner_rows = [], []
inner_names = ['start', 'index arithmetic done', 'K loop done', 'cross-wave sum done', 'partial stores visible', 'ticket taken', 'partials summed', 'end']
for rep in range(30):
    enc.encode_nn(cb, x, 1)
    torch.cuda.synchronize()
    _lib.check(enc.lib, enc.lib.aae_encoder_debug_timeline(enc.handle, buf), 'timeline')
    if rep < 5:
        continue
    t = np.frombuffer(buf, dtype=np.int64)[:G * S].reshape(G, S)[:, :len(names)].astype(np.float64) / 100.0     # us
    t -= t[:, 0].min()
    if inner:
        raw = np.frombuffer(buf, dtype=np.int64)[G * S:G * S + 256 * 8].reshape(256, 8).astype(np.float64) / 100.0
        base = np.frombuffer(buf, dtype=np.int64)[:G * S].reshape(G, S)[:, 0].min() / 100.0
        ok = raw[:, 7] > 0
        u = raw - base
        inner_rows.append(np.stack([np.nanmin(np.where(u > 0, u, np.nan), axis=0), np.nanmedian(np.where(u > 0, u, np.nan), axis=0),
                                    np.nanmax(np.where(u > 0, u, np.nan), axis=0)], axis=0))
    rows.append(np.stack([t.min(axis=0), np.median(t, axis=0), t.max(axis=0)], axis=0))
m = np.mean(rows, axis=0)
prev = 0.0
for i, n in enumerate(names):
    print(json.dumps({'edge': '%2d %s' % (i, n), 'first_us': round(float(m[0, i]), 2), 'median_us': round(float(m[1, i]), 2),
                      'last_us': round(float(m[2, i]), 2), 'last_minus_previous_last_us': round(float(m[2, i] - prev), 2)}))
    prev = float(m[2, i])
if inner_rows:
    mi = np.nanmean(inner_rows, axis=0)
    for i, n in enumerate(inner_names):
        print(json.dumps({'inside conv layer %d' % inner: n, 'first_us': round(float(mi[0, i]), 2), 'median_us': round(float(mi[1, i]), 2),
                          'last_us': round(float(mi[2, i]), 2)}))
