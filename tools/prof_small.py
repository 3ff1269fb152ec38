#!/usr/bin/env python
"""encode + nn at a small batch size, for `rocprofv3 --kernel-trace --stats` (per-kernel GPU durations without
the event-timing launch gaps).  Usage: python tools/prof_small.py [B] [reps] [new|old|noprep] [opt=value,...]"""
import _experiments  # noqa: F401  (the kernel variants compared here live in the experiments build: libaae_hip_experiments.so)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import _lib, synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else 'new'
enc = EncoderEngine(EncoderConfig(), synth.make_weights(seed=2024), max_batch=max(B, 1))
cb = CodebookEngine(synth.make_codebook(92232, 128, seed=7))
if mode == 'old':          # the 128 x 128 split-K igemm + separate reduce launches
    for name in ('wavek', 'gemv_ticket', 'wavek_dense'):
        enc.set_option(name, 0)
    if B <= 4:
        cb.set_scan_mode(_lib.AAE_SCAN_STREAM_2L)
if mode == 'noprep':       # every ticketed launch installs its own nonce (the arrivals queue up behind the install)
    enc.set_option('ticket_prep', 0)
for kv in (sys.argv[4].split(',') if len(sys.argv) > 4 else []):       # extra encoder options: name=value,name=value
    name, value = kv.split('=')
    if name == 'scan_mode':                 # (a codebook option: AAE_SCAN_* of include/aae_hip.h)
        cb.set_scan_mode(int(value))
    else:
        enc.set_option(name, int(value))
x = torch.from_numpy(synth.make_crops(B, seed=3)).cuda()
for _ in range(reps):
    if mode == 'old':
        cb.nn(enc.encode(x), 1, 1)
    else:
        enc.encode_nn(cb, x, 1)          # the fused per-detection call (aae_encode_nn)
torch.cuda.synchronize()
