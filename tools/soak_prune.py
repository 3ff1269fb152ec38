#!/usr/bin/env python
"""Soak run for the pruned top-k of the query-resident scan (codebook_scan_resident.h): the bound the blocks publish to each
other is read with no ordering at all, so which candidates a block drops depends on timing -- the ANSWER must not.  Thousands
of top-k queries (new latents, batch size and k every time; fp32 and bf16 codebooks; structured latents near codebook rows so
that near-ties at the k-th place are common; about a third under a second stream that saturates HBM), each compared bit for bit
-- indices and scores -- with the same query on a second engine that takes every candidate (AAE_SCAN_AUTO_NO_PRUNE), and every
20th also with the similarity-matrix path.  Not part of the test suite (~1-2 GPU-minutes):  python tools/soak_prune.py [queries]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from augmentedautoencoder_amd import _lib, synth
from augmentedautoencoder_amd.engine import CodebookEngine


def main():
    queries = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    books = []
    for dtype, N in (('f32', 92232), ('bf16', 368928)):
        E = synth.make_codebook(N, 128, seed=7, planted_duplicates=16)
        a, b, m = CodebookEngine(E, dtype=dtype), CodebookEngine(E, dtype=dtype), CodebookEngine(E, dtype=dtype)
        b.set_scan_mode(_lib.AAE_SCAN_AUTO_NO_PRUNE)
        m.set_scan_mode(_lib.AAE_SCAN_MFMA)
        books.append((dtype, torch.from_numpy(E[:4096]).cuda(), a, b, m))
    side = torch.cuda.Stream()
    big_a = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
    big_b = torch.zeros(1 << 30, dtype=torch.uint8, device='cuda')
    rng = np.random.default_rng(3)
    gen = torch.Generator(device='cuda').manual_seed(5)
    bad, bad_matrix, matrix_checks, t0 = 0, 0, 0, time.time()
    per = {}
    for it in range(queries):
        dtype, rows, a, b, m = books[it % 2]
        B = int(rng.choice([5, 8, 17, 32, 33, 64, 100, 128, 129, 200, 256]))
        k = int(rng.choice([2, 3, 4, 5, 8]))
        z = torch.randn(B, 128, device='cuda', generator=gen)
        if it % 3 == 0:                                   # latents close to codebook rows: the top of the list is crowded
            sel = torch.randint(0, rows.shape[0], (B,), device='cuda', generator=gen)
            z = rows[sel] * 3.0 + 0.05 * z
        if it % 3 == 1:
            with torch.cuda.stream(side):
                for _ in range(2):
                    big_a.copy_(big_b)
        ia, sa = a.nn(z, k, 1)
        ib, sb = b.nn(z, k, 1)
        ok = torch.equal(ia, ib) and torch.equal(sa, sb)
        bad += int(not ok)
        if it % 20 == 0:
            im, sm = m.nn(z, k, 1)
            matrix_checks += 1
            bad_matrix += int(not (torch.equal(ia, im) and torch.equal(sa, sm)))
        per[(dtype, k)] = per.get((dtype, k), 0) + 1
    torch.cuda.synchronize()
    print(json.dumps({'what': 'soak_prune', 'queries': queries, 'per_dtype_and_k': {'%s k=%d' % kk: v for kk, v in sorted(per.items())},
                      'pruned_vs_unpruned_mismatches': bad, 'similarity_matrix_path_checks': matrix_checks,
                      'pruned_vs_matrix_path_mismatches': bad_matrix, 'seconds': round(time.time() - t0, 1)}), flush=True)
    sys.exit(1 if bad or bad_matrix else 0)


if __name__ == '__main__':
    main()
