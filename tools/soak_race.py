#!/usr/bin/env python
"""Soak run for the kernels whose correctness depends on DMA / load landing order (LDS-DMA igemm, weights-to-registers
variant with counted vmcnt + bare barrier, f32x3h DMA): thousands of launches, with and without a second stream saturating
HBM, every result compared bit for bit with the register-staged kernels.  Not part of the test suite (takes ~1 GPU-minute)."""
import _experiments  # noqa: F401  (the kernel variants compared here live in the experiments build: libaae_hip_experiments.so)
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from augmentedautoencoder_amd.engine import EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig
from oracle import synth


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    enc = EncoderEngine(EncoderConfig(), synth.make_weights(seed=2024))
    side = torch.cuda.Stream()
    big_a = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
    big_b = torch.zeros(1 << 30, dtype=torch.uint8, device='cuda')
    report = []
    for B in (256, 37):
        x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
        for precision in (0, 1):
            enc.set_option('precision', precision)
            enc.set_option('igemm_dma', 0); enc.set_option('x3h_dma', 0); enc.set_option('igemm_breg', 0)
            want = enc.encode(x).clone()
            enc.set_option('igemm_dma', 1); enc.set_option('x3h_dma', 1); enc.set_option('igemm_breg', 1)
            bad = 0
            t0 = time.time()
            for it in range(iters):
                if it % 50 == 0:                              # keep HBM busy about half of the time
                    with torch.cuda.stream(side):
                        for _ in range(4):
                            big_a.copy_(big_b)
                z = enc.encode(x)
                if it % 10 == 0 or it == iters - 1:           # comparing every launch would serialise the queue
                    bad += int(not torch.equal(z, want))
            torch.cuda.synchronize()
            report.append({'B': B, 'precision': precision, 'launches': iters, 'mismatching_checks': bad, 'seconds': round(time.time() - t0, 1)})
            print(json.dumps(report[-1]), flush=True)
    sys.exit(1 if any(r['mismatching_checks'] for r in report) else 0)


if __name__ == '__main__':
    main()
