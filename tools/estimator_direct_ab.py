#!/usr/bin/env python
"""AePoseEstimator.process with few detections: the crop kernel reading the staged rectangle in place (pinned host memory) against
the copy to the device in front of it, A B A B on one box.  One JSON line per detection count."""
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
ds = Dataset('', h=128, w=128, c=3, min_n_views=2562, radius=700, num_cyclo=36)
with S.variable_scope('obj_a'):
    e = Encoder(S.Placeholder((128, 128, 3)), 128, [128, 256, 512, 512], 5, [2, 2, 2, 2], False)
    c = Codebook(e, ds, True)
e.load_weights(synth.make_weights(seed=50))
c.assign_embedding(synth.make_codebook(92232, 128, seed=60))
r = np.random.default_rng(70)
c.assign_obj_bbs(np.stack([r.integers(250, 350, 92232), r.integers(180, 260, 92232), r.integers(80, 200, 92232), r.integers(80, 200, 92232)], 1))
est = AePoseEstimator(codebooks={'obj_a': c}, train_args={'obj_a': targs})
rng = np.random.default_rng(0)
img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
camK = np.array([[1075.65, 0, 960.0], [0, 1073.9, 540.0], [0, 0, 1]])
for D, span in ((1, 0), (1, 1), (2, 200), (4, 300), (4, 800), (8, 600), (16, 1000), (64, 1000)):
    dets = []
    for i in range(D):
        x, y = 700 + rng.uniform(0, span), 400 + rng.uniform(0, span * 0.5)
        w, h = (rng.uniform(60, 150), rng.uniform(60, 150)) if span != 1 else (330.0, 260.0)
        dets.append(BoundingBox(xmin=x / 1920, xmax=(x + w) / 1920, ymin=y / 1080, ymax=(y + h) / 1080, classes={'obj_a': 1.0}))
    est.process(dets, img, camK)
    stage = list(est._stages.values())[0]
    out = {'detections': D, 'staged_bytes': int(stage.img_host.numel()) if D == 0 else None, 'in_place_us': [], 'copied_us': []}
    poses = {}
    for rnd in range(3):
        for key, limit in (('in_place_us', 1 << 30), ('copied_us', 0)):
            stage.direct_rows = limit
            for _ in range(20):
                got = est.process(dets, img, camK)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 200 if D <= 8 else 20
            for _ in range(reps):
                est.process(dets, img, camK)
            torch.cuda.synchronize()
            out[key].append(round((time.perf_counter() - t0) / reps * 1e6, 1))
            poses[key] = np.stack([g.trafo for g in got])
    out['identical_poses'] = bool(np.array_equal(poses['in_place_us'], poses['copied_us']))
    print(json.dumps(out), flush=True)
