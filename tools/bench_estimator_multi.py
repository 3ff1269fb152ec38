#!/usr/bin/env python
"""AePoseEstimator.process on a 1080p frame whose detections spread over EIGHT object classes (one AAE per class, as
m3_interface/ae_pose_estimator.py:61-78 keeps them): one C call per frame with one launch per layer across the classes
(multi_call, the default) against one aae_detect_nn call per class.  One JSON line per class mix."""
import configparser
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from augmentedautoencoder_amd import session as S
from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.codebook import Codebook
from augmentedautoencoder_amd.dataset import Dataset
from augmentedautoencoder_amd.encoder import Encoder
from augmentedautoencoder_amd.pose_estimator import AePoseEstimator, BoundingBox


def main():
    n_cls = 8
    targs = configparser.ConfigParser()
    targs.read_string("[Dataset]\nH: 128\nW: 128\nC: 3\nRADIUS: 700\nPAD_FACTOR: 1.2\nK: [1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]\n"
                      "[Embedding]\nEMBED_BB: True\nMIN_N_VIEWS: 2562\nNUM_CYCLO: 36\n")
    S.reset_default_graph()
    books, names = {}, ['obj_%02d' % k for k in range(n_cls)]
    for k, name in enumerate(names):
        ds = Dataset('', h=128, w=128, c=3, min_n_views=2562, radius=700, num_cyclo=36)
        with S.variable_scope(name):
            e = Encoder(S.Placeholder((128, 128, 3)), 128, [128, 256, 512, 512], 5, [2, 2, 2, 2], False)
            c = Codebook(e, ds, True)
        e.load_weights(synth.make_weights(seed=50 + k))
        c.assign_embedding(synth.make_codebook(92232, 128, seed=60 + k))
        r = np.random.default_rng(70 + k)
        c.assign_obj_bbs(np.stack([r.integers(250, 350, 92232), r.integers(180, 260, 92232), r.integers(80, 200, 92232), r.integers(80, 200, 92232)], 1))
        books[name] = c
    est = AePoseEstimator(codebooks=books, train_args={n: targs for n in names})
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    camK = np.array([[1075.65, 0, 960.0], [0, 1073.9, 540.0], [0, 0, 1]])
    mixes = {'1 class x 1': [1], '8 classes x 1': [1] * 8, '8 classes x {1,1,2,4,1,3,1,2}': [1, 1, 2, 4, 1, 3, 1, 2], '8 classes x 4': [4] * 8,
             '4 classes x 1': [1] * 4, '2 classes x 1': [1] * 2, '8 classes x {9,1,1,1,1,1,1,1}': [9, 1, 1, 1, 1, 1, 1, 1]}
    for label, counts in mixes.items():
        dets = []
        for k, n in enumerate(counts):
            for _ in range(n):
                x, y, w, h = rng.uniform(0, 1500), rng.uniform(0, 800), rng.uniform(60, 400), rng.uniform(60, 270)
                dets.append(BoundingBox(xmin=x / 1920, xmax=(x + w) / 1920, ymin=y / 1080, ymax=(y + h) / 1080, classes={names[k]: 1.0}))
        D = len(dets)
        row = {'what': 'estimator_multi', 'mix': label, 'detections': D, 'image': '1080x1920'}
        ref_out = None
        for mode in (1, 0, 1, 0):
            est.multi_call = bool(mode)
            reps = max(20, 400 // D)
            for _ in range(max(3, reps // 10)):
                out = est.process(dets, img, camK)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = est.process(dets, img, camK)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            key = 'one_call_per_frame_ms' if mode else 'one_call_per_class_ms'
            row[key] = round(min(row.get(key, 1e9), dt * 1e3), 4)
            poses = np.stack([p.trafo for p in out])
            if ref_out is None:
                ref_out = poses
            else:
                row['max_pose_difference_between_modes'] = float(max(row.get('max_pose_difference_between_modes', 0.0), np.abs(poses - ref_out).max()))
        row['speedup'] = round(row['one_call_per_class_ms'] / row['one_call_per_frame_ms'], 3)
        print(json.dumps(row), flush=True)
    est.close()


if __name__ == '__main__':
    main()
