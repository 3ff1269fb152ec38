#!/usr/bin/env python
"""The HBM-bound and per-detection launches in ONE process, for rocprofv3 passes (tools/gpu_run.sh prof:small:python,tools/prof_mix.py):
  * fused encode+nn (aae_encode_nn) at B = 1 and B = 4 -- the reference's per-detection operating point
    (m3_interface/ae_pose_estimator.py:143-170);
  * the stand-alone codebook query at B = 1, warm (one codebook) and cold (8 copies visited in turn, 378 MB);
  * BASELINE config 5: 368928 x 128 bf16 codebook, B = 256 arg-max / top-5 and B = 1.
Usage: python tools/prof_mix.py [reps] [opt=value,...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
enc = EncoderEngine(EncoderConfig(), synth.make_weights(seed=2024), max_batch=4)
for kv in (sys.argv[2].split(',') if len(sys.argv) > 2 else []):
    name, value = kv.split('=')
    enc.set_option(name, int(value))
E = synth.make_codebook(92232, 128, seed=7)
cb = CodebookEngine(E)
x = torch.from_numpy(synth.make_crops(4, seed=3)).cuda()
for b in (1, 4):
    xb = x[:b].contiguous()
    for _ in range(reps):
        enc.encode_nn(cb, xb, 1)
torch.cuda.synchronize()
z = enc.encode(x)
z1 = z[:1].contiguous()
for _ in range(reps):
    cb.nn(z1, 1, 1)
torch.cuda.synchronize()
# cold: 8 copies visited in turn (378 MB > the 256 MB Infinity Cache).  128 rows fewer than the warm codebook, so that the
# launches have a grid of their own (720 instead of 721 blocks) and the per-kernel summaries keep warm and cold apart.
copies = [CodebookEngine(E[:-128]) for _ in range(8)]
for i in range(max(reps, 64)):
    copies[i % 8].nn(z1, 1, 1)
torch.cuda.synchronize()
for c in copies:
    c.close()
E5 = synth.make_codebook(368928, 128, seed=11, planted_duplicates=0)
cb5 = CodebookEngine(E5, dtype='bf16')
z5 = torch.randn(256, 128, device='cuda')
for _ in range(max(reps // 4, 10)):
    cb5.nn(z5, 1, 1)
for _ in range(max(reps // 8, 5)):
    cb5.nn(z5, 5, 1)
for _ in range(max(reps // 2, 10)):
    cb5.nn(z5[:1], 1, 1)
torch.cuda.synchronize()
