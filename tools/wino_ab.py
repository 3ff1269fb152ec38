#!/usr/bin/env python
"""A/B on the GPU: the conv layers as polyphase Winograd (encoder option "winograd") against the direct fp32 kernels -- per-layer
device time (encode_timed: HIP events around every launch group), whole forward, and the error of both against the float64 oracle
on a few crops.  python tools/wino_ab.py [B ...]"""
import json
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from augmentedautoencoder_amd.engine import EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig
from oracle import reference_cpu as ref
from oracle import synth


def main():
    batches = [int(a) for a in sys.argv[1:]] or [256]
    cfg = EncoderConfig()
    w = synth.make_weights(seed=2024)
    direct = EncoderEngine(cfg, w, max_batch=max(batches))
    direct.set_option('winograd', 0)
    wino = EncoderEngine(cfg, w, max_batch=max(batches))
    wino.set_option('winograd', int(os.environ.get('WINO_MODE', '1')))
    if os.environ.get('WINO_MIN'):          # (default: the product rule -- a layer from three quarters of a round of blocks on, never below B = 8)
        wino.set_option('winograd_min_batch', int(os.environ['WINO_MIN']))
    if os.environ.get('WINO_MIN_BLOCKS'):
        wino.set_option('winograd_min_blocks', int(os.environ['WINO_MIN_BLOCKS']))
    for B in batches:
        x = torch.from_numpy(synth.make_crops(B, seed=7)).cuda()
        out = {'what': 'winograd_vs_direct', 'B': B}
        for name, enc in (('direct', direct), ('winograd', wino)):
            for _ in range(5):
                enc.encode(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                z = enc.encode(x)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            _, recs = enc.encode_timed(x)
            layers = {}
            for label, lms, fl in recs:
                key = label.split(':')[0]
                layers[key] = layers.get(key, 0.0) + lms
            out[name] = {'forward_ms': round(ms, 4), 'crops_per_s': round(B / ms * 1e3, 1), 'layer_ms': {k: round(v, 4) for k, v in layers.items()},
                         'labels': [r[0] for r in recs][:3]}
            out[name + '_z'] = z
        zd, zw = out.pop('direct_z'), out.pop('winograd_z')
        n = min(B, 4)
        z64 = ref.encoder_forward_np(ref.input_to_float(x[:n].cpu().numpy()), w, cfg.strides, cfg.batch_norm)
        sc = np.abs(z64).max()
        out['z_rel_err_vs_fp64'] = {'direct': float(np.abs(zd[:n].cpu().numpy() - z64).max() / sc), 'winograd': float(np.abs(zw[:n].cpu().numpy() - z64).max() / sc)}
        out['z_rel_diff_winograd_vs_direct'] = float((zw - zd).abs().max().item() / sc)
        cos = torch.nn.functional.cosine_similarity(zw, zd, dim=1)
        out['min_cosine_winograd_vs_direct'] = float(cos.min().item())
        out['speedup_forward'] = round(out['direct']['forward_ms'] / out['winograd']['forward_ms'], 3)
        print(json.dumps(out))


if __name__ == '__main__':
    main()
