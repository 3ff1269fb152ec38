#!/usr/bin/env python
"""Random frames through aae_encode_nn_multi on the GPU against one aae_encode_nn call per class: 2 ... 8 classes with 1 ... 40 boxes each (every mix of per-detection
groups, mid-batch groups with their per-layer choice / handed-over images / shared scans, and lone classes).  Latents within 5e-6 of the per-class calls, indices equal wherever
the per-class top-2 cosine gap exceeds 2e-6.   python tools/gpu_fuzz_multi.py [frames] [seed]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from augmentedautoencoder_amd import synth                                    # noqa: E402
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine, MultiObjectQuery   # noqa: E402
from augmentedautoencoder_amd.weights import EncoderConfig                    # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    books = [synth.make_codebook(92232, 128, seed=300 + o) for o in range(8)]
    objs = [(EncoderEngine(EncoderConfig(), synth.make_weights(seed=100 + o), max_batch=64), CodebookEngine(books[o])) for o in range(8)]
    worst, flips = 0.0, 0
    for f in range(frames):
        k = int(rng.integers(2, 9))
        style = int(rng.integers(0, 4))
        if style == 0:
            counts = [int(rng.integers(1, 5)) for _ in range(k)]
        elif style == 1:
            counts = [int(rng.integers(5, 41)) for _ in range(k)]
        elif style == 2:
            counts = [int(rng.integers(1, 41)) for _ in range(k)]
        else:
            counts = [int(rng.integers(5, 13)) for _ in range(k)]
        order = rng.permutation(8)[:k]
        x = torch.from_numpy(synth.make_crops(sum(counts), seed=1000 + f)).cuda()
        items = [(objs[o][0], objs[o][1], n) for o, n in zip(order, counts)]
        mq = MultiObjectQuery(items)
        z, idx, score = mq(x)
        torch.cuda.synchronize()
        at, rel, bad = 0, 0.0, 0
        for (e, c, n) in items:
            wz, wi, ws = e.encode_nn(c, x[at:at + n], 1)
            rel = max(rel, float((z[at:at + n] - wz).abs().max() / wz.abs().max()))
            diff = (idx[at:at + n].cpu() != wi[:, 0].cpu()).nonzero().flatten().tolist()
            if diff:
                o = [oo for oo in range(8) if objs[oo][0] is e][0]
                zn = wz / wz.norm(dim=1, keepdim=True)
                cs = (zn[diff].double().cpu().numpy() @ books[o].astype(np.float64).T)
                top = np.sort(cs, axis=1)[:, -2:]
                bad += int(((top[:, 1] - top[:, 0]) > 2e-6).sum())
            at += n
        worst = max(worst, rel)
        flips += bad
        print(json.dumps({'frame': f, 'counts': counts, 'launches': mq.launches, 'max_rel_latent_difference': rel, 'index_differences_beyond_the_gap': bad}), flush=True)
        assert rel < 5e-6 and bad == 0, (counts, rel, bad)
    print(json.dumps({'frames': frames, 'worst_rel_latent_difference': worst, 'index_differences_beyond_the_gap': flips}))


if __name__ == '__main__':
    main()
