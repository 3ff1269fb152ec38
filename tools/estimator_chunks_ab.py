#!/usr/bin/env python
"""AePoseEstimator.process on a 1080p frame, D detections of two classes: the chunk plan of _chunk_plan (large chunks first, each at most three times what follows it,
the last one geometry_chunk) against equal chunks of geometry_chunk (the form of the first half of round 4), A B A B on one box."""
import configparser
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from augmentedautoencoder_amd import session as S, synth
from augmentedautoencoder_amd.codebook import Codebook
from augmentedautoencoder_amd.dataset import Dataset
from augmentedautoencoder_amd.encoder import Encoder
from augmentedautoencoder_amd.pose_estimator import AePoseEstimator, BoundingBox

targs = configparser.ConfigParser()
targs.read_string("[Dataset]\nH: 128\nW: 128\nC: 3\nRADIUS: 700\nPAD_FACTOR: 1.2\nK: [1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]\n"
                  "[Embedding]\nEMBED_BB: True\nMIN_N_VIEWS: 2562\nNUM_CYCLO: 36\n")
S.reset_default_graph()
books = {}
for k, name in enumerate(['obj_a', 'obj_b']):
    ds = Dataset('', h=128, w=128, c=3, min_n_views=2562, radius=700, num_cyclo=36)
    with S.variable_scope(name):
        e = Encoder(S.Placeholder((128, 128, 3)), 128, [128, 256, 512, 512], 5, [2, 2, 2, 2], False)
        c = Codebook(e, ds, True)
    e.load_weights(synth.make_weights(seed=50 + k))
    c.assign_embedding(synth.make_codebook(92232, 128, seed=60 + k))
    r = np.random.default_rng(70 + k)
    c.assign_obj_bbs(np.stack([r.integers(250, 350, 92232), r.integers(180, 260, 92232), r.integers(80, 200, 92232), r.integers(80, 200, 92232)], 1))
    books[name] = c
est = AePoseEstimator(codebooks=books, train_args={'obj_a': targs, 'obj_b': targs})
rng = np.random.default_rng(0)
img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
camK = np.array([[1075.65, 0, 960.0], [0, 1073.9, 540.0], [0, 0, 1]])
graded = AePoseEstimator._chunk_plan


def equal(self, counts):
    c = self.geometry_chunk
    return [([n] if n else []) if n <= 2 * c else [min(c, n - a) for a in range(0, n, c)] for n in counts]


for D in (48, 64, 128, 256):
    dets = []
    for i in range(D):
        x, y, w, h = rng.uniform(0, 1500), rng.uniform(0, 800), rng.uniform(60, 400), rng.uniform(60, 270)
        dets.append(BoundingBox(xmin=x / 1920, xmax=(x + w) / 1920, ymin=y / 1080, ymax=(y + h) / 1080, classes={'obj_a' if i % 3 else 'obj_b': 1.0}))
    out = {'detections': D, 'graded_ms': [], 'equal_ms': []}
    poses = {}
    for rnd in range(3):
        for key, fn in (('graded_ms', graded), ('equal_ms', equal)):
            AePoseEstimator._chunk_plan = fn
            for _ in range(3):
                got = est.process(dets, img, camK)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                est.process(dets, img, camK)
            torch.cuda.synchronize()
            out[key].append(round((time.perf_counter() - t0) / 10 * 1e3, 3))
            poses[key] = np.stack([g.trafo for g in got])
    out['identical_poses'] = bool(np.array_equal(poses['graded_ms'], poses['equal_ms']))
    print(json.dumps(out), flush=True)
