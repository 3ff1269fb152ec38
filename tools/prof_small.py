#!/usr/bin/env python
"""encode + nn at a small batch size, for `rocprofv3 --kernel-trace --stats` (per-kernel GPU durations without
the event-timing launch gaps).  Usage: python tools/prof_small.py [B] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
enc = EncoderEngine(EncoderConfig(), synth.make_weights(seed=2024), max_batch=max(B, 1))
cb = CodebookEngine(synth.make_codebook(92232, 128, seed=7))
x = torch.from_numpy(synth.make_crops(B, seed=3)).cuda()
for _ in range(reps):
    cb.nn(enc.encode(x), 1, 1)
torch.cuda.synchronize()
