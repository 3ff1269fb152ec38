#!/usr/bin/env python
"""Where one AePoseEstimator.process call with D detections spends its wall time on the host: before the C call, inside it (launches),
waiting for the GPU, after the wait (geometry + results).  perf_counter stamps around EncoderEngine.detect_nn and Event.synchronize.
Usage: python tools/estimator_timeline.py [D] [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from augmentedautoencoder_amd.engine import EncoderEngine
from prof_estimator import build, detections

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
est = build()
rng = np.random.default_rng(0)
img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
camK = np.array([[1075.65, 0, 960.0], [0, 1073.9, 540.0], [0, 0, 1]])
dets = detections(rng, D)
stamps = []
orig_detect, orig_sync = EncoderEngine.detect_nn, torch.cuda.Event.synchronize


def detect(self, *a, **k):
    stamps.append(('call', time.perf_counter()))
    r = orig_detect(self, *a, **k)
    stamps.append(('called', time.perf_counter()))
    return r


def sync(self):
    stamps.append(('wait', time.perf_counter()))
    r = orig_sync(self)
    stamps.append(('woke', time.perf_counter()))
    return r


EncoderEngine.detect_nn, torch.cuda.Event.synchronize = detect, sync
for _ in range(20):
    est.process(dets, img, camK)
acc = {'before_first_call': 0.0, 'in_calls': 0.0, 'between_calls_and_waits': 0.0, 'waiting': 0.0, 'after_last_wake': 0.0, 'total': 0.0}
for _ in range(reps):
    del stamps[:]
    t0 = time.perf_counter()
    est.process(dets, img, camK)
    t1 = time.perf_counter()
    calls = [t for k, t in stamps if k == 'call']
    called = [t for k, t in stamps if k == 'called']
    waits = [t for k, t in stamps if k == 'wait']
    woke = [t for k, t in stamps if k == 'woke']
    acc['before_first_call'] += calls[0] - t0
    acc['in_calls'] += sum(b - a for a, b in zip(calls, called))
    acc['waiting'] += sum(b - a for a, b in zip(waits, woke))
    acc['after_last_wake'] += t1 - woke[-1]
    acc['total'] += t1 - t0
acc['between_calls_and_waits'] = acc['total'] - acc['before_first_call'] - acc['in_calls'] - acc['waiting'] - acc['after_last_wake']
print(json.dumps({'detections': D, 'reps': reps, 'us_per_call': {k: round(v / reps * 1e6, 1) for k, v in acc.items()}}))
