#!/usr/bin/env python
"""Soak of the mid-batch grouped query (aae_encode_nn_multi, config-4 frame: 256 crops over 8 objects, 9 grouped launches): 400 frames under side-stream memory
traffic must reproduce the first frame bit for bit; six rotations of the bucket sizes over the objects (other layouts, other incomplete blocks) against the
per-object calls (<= 1e-5 of the latent scale).  Round 6: 0 mismatching frames.   python tools/soak_mid_batch.py"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine, MultiObjectQuery
from augmentedautoencoder_amd.weights import EncoderConfig
counts = [34, 26, 27, 32, 31, 32, 33, 41]
objs = [(EncoderEngine(EncoderConfig(), synth.make_weights(seed=100 + o), max_batch=64), CodebookEngine(synth.make_codebook(92232, 128, seed=200 + o))) for o in range(8)]
x = torch.from_numpy(synth.make_crops(sum(counts), seed=4321)).cuda()
mq = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs, counts)])
z0, i0, s0 = [t.clone() for t in mq(x)]
side = torch.cuda.Stream()
a = torch.empty(512 << 20, dtype=torch.uint8, device='cuda'); b = torch.zeros(512 << 20, dtype=torch.uint8, device='cuda')
bad = 0
t0 = time.time()
for it in range(400):
    if it % 4 == 0:
        with torch.cuda.stream(side):
            for _ in range(10): a.copy_(b)
    z, i, s = mq(x)
    if not (torch.equal(z, z0) and torch.equal(i, i0) and torch.equal(s, s0)): bad += 1
torch.cuda.synchronize()
print('soak: 400 frames of config 4 (9 grouped launches each) under side-stream memory traffic, mismatching frames:', bad, 'launches', mq.launches, '%.1f s' % (time.time() - t0))
# alternating layouts in one workspace: counts rotate
for r in range(6):
    cs = counts[r:] + counts[:r]
    mq2 = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs, cs)])
    z, i, s = mq2(x)
    at = 0
    for (e, c), n in zip(objs, cs):
        wz, wi, wsc = e.encode_nn(c, x[at:at + n], 1)
        assert torch.equal(i[at:at + n].cpu(), wi[:, 0].cpu()) or float((z[at:at+n]-wz).abs().max()/wz.abs().max()) < 1e-5
        assert float((z[at:at+n]-wz).abs().max()/wz.abs().max()) < 1e-5, (r, n)
        at += n
print('rotated layouts OK')
# 4 classes x 6 boxes: with the defaults answered as items of <= 4 boxes inside the per-detection group (multi_split_items); a second pass with that option off takes the
# mid-batch group with its per-layer choice (conv2 / conv3 grouped, conv4 per object)
mq3 = MultiObjectQuery([(e, c, 6) for e, c in objs[:4]])
x3 = x[:24]
z0, i0, s0 = [t.clone() for t in mq3(x3)]
bad = 0
for it in range(300):
    if it % 4 == 0:
        with torch.cuda.stream(side):
            for _ in range(10): a.copy_(b)
    z, i, s = mq3(x3)
    if not (torch.equal(z, z0) and torch.equal(i, i0) and torch.equal(s, s0)): bad += 1
torch.cuda.synchronize()
print('soak: 300 frames of 4 classes x 6 boxes (split into items of four), mismatching frames:', bad, 'launches', mq3.launches)
for e, _ in objs: e.set_option('multi_split_items', 0)
mq4 = MultiObjectQuery([(e, c, 6) for e, c in objs[:4]])
z0, i0, s0 = [t.clone() for t in mq4(x3)]
bad = 0
for it in range(300):
    z, i, s = mq4(x3)
    if not (torch.equal(z, z0) and torch.equal(i, i0) and torch.equal(s, s0)): bad += 1
torch.cuda.synchronize()
print('soak: 300 frames of 4 classes x 6 boxes (mid-batch group, per-layer choice), mismatching frames:', bad, 'launches', mq4.launches)

