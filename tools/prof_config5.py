#!/usr/bin/env python
"""BASELINE config 5 alone (368928 x 128 bf16 codebook, B = 256: arg-max and top-5) for rocprofv3 passes.
Usage: python tools/prof_config5.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cb5 = CodebookEngine(synth.make_codebook(368928, 128, seed=11, planted_duplicates=0), dtype='bf16')
z5 = torch.randn(256, 128, device='cuda')
for _ in range(reps):
    cb5.nn(z5, 1, 1)
for _ in range(max(reps // 2, 5)):
    cb5.nn(z5, 5, 1)
torch.cuda.synchronize()
