#!/usr/bin/env python
"""Diagnostic of the top-k pruning inside the query-resident scan (config 5 shapes): timing with / without the shared
bound, and the bound words left in the workspace after a call.  Usage: python tools/diag_prune.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from augmentedautoencoder_amd import _lib, synth
from augmentedautoencoder_amd.engine import CodebookEngine

N = 368928
cb = CodebookEngine(synth.make_codebook(N, 128, seed=11, planted_duplicates=0), dtype='bf16')


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / reps


for B in (32, 256):
    z = torch.randn(B, 128, device='cuda')
    out = {'what': 'diag_prune', 'B': B}
    for name, mode in (('pruned', _lib.AAE_SCAN_AUTO), ('unpruned', _lib.AAE_SCAN_AUTO_NO_PRUNE)):
        cb.set_scan_mode(mode)
        out[name + '_top5_us'] = round(timed(lambda: cb.nn(z, 5, 1)), 2)
    cb.set_scan_mode(_lib.AAE_SCAN_AUTO)
    i5, s5 = cb.nn(z, 5, 1)
    torch.cuda.synchronize()
    nbytes = cb.workspace_bytes(B, 5)
    buf, ptr = cb.ws.get(nbytes)
    off = ptr - buf.data_ptr()
    Bpad = 64 * ((B + 63) // 64)
    REPLICAS = 8                                              # kPruneReplicas: [replica][query][16 words], the region ends the workspace
    region = ((REPLICAS * Bpad * 16 * 4 + 255) // 256) * 256
    allw = buf[off + nbytes - region: off + nbytes][:REPLICAS * Bpad * 64].view(torch.int32).view(REPLICAS, Bpad, 16).cpu().numpy()
    out['replicas_identical'] = bool(all(np.array_equal(allw[0], allw[r]) for r in range(1, REPLICAS)))
    words = allw[0]
    keys = words.astype(np.int64)
    bits = np.where(keys >= 0, keys, keys ^ 0x7fffffff).astype(np.int64) & 0xffffffff
    vals = bits.astype(np.uint32).view(np.float32).reshape(Bpad, 16)
    kth = np.sort(vals, axis=1)[:, -5]
    out['words_query0'] = [float(v) for v in vals[0]]
    out['bound_vs_true_5th_query0'] = [float(kth[0]), float(s5[0, 4])]
    out['empty_words'] = int((words == -2**31).sum())
    out['bound_below_true_5th_everywhere'] = bool(np.all(kth[:B] <= s5[:, 4].cpu().numpy() + 0))
    print(json.dumps(out))
