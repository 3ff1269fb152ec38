#!/usr/bin/env python
"""Which buffer's PLACEMENT moves the B <= 4 stream scan by up to 20 %?  (tools/scan_b1_ab.py on three byte-identical copies of the library
gave three different, individually stable kernel periods: it is per-handle state, not code.)  One library, and
  (a) several codebook handles (own 47 MB hipMalloc each) sharing ONE workspace;
  (b) one handle, the workspace at byte offsets 0, 256, ..., inside one buffer;
  (c) one handle, one workspace, the codebook re-created after freeing / allocating padding of various sizes.
python tools/scan_placement_probe.py [lib.so]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan_b1_ab import load                     # noqa: E402
from augmentedautoencoder_amd import synth      # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else 'augmentedautoencoder_amd/libaae_hip.so'
    lib = load(path)
    dev = torch.device('cuda', 0)
    E = synth.make_codebook(92232, 128, seed=7, planted_duplicates=8)
    z = torch.randn(4, 128, device=dev)
    idx = torch.empty(4, dtype=torch.int64, device=dev)
    score = torch.empty(4, dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def create():
        h = ctypes.c_void_p()
        rc = lib.aae_codebook_create(E.ctypes.data, 92232, 128, 1, 0, ctypes.byref(h))
        assert rc == 0, lib.aae_last_error()
        return h

    def timed(h, ptr, nbytes, B, n=2000):
        ms = ctypes.c_float()
        rc = lib.aae_codebook_nn_timed(h, z.data_ptr(), B, 1, 1, idx.data_ptr(), score.data_ptr(), ptr, nbytes, stream, n, ctypes.byref(ms))
        assert rc == 0, lib.aae_last_error()
        return round(ms.value * 1e3, 3)
    h0 = create()
    nbytes = max(int(lib.aae_codebook_workspace_bytes(h0, b, 1)) for b in (1, 2, 4))
    big = torch.zeros(nbytes + (1 << 20), dtype=torch.uint8, device=dev)
    base = big.data_ptr() + (-big.data_ptr()) % 4096
    for _ in range(3):
        timed(h0, base, nbytes, 1, 500)
    # (a) handles
    handles = [h0] + [create() for _ in range(4)]
    for rnd in range(2):
        for k, h in enumerate(handles):
            print(json.dumps({'what': 'scan_placement', 'vary': 'codebook handle (own 47 MB allocation), one workspace', 'handle': k, 'round': rnd,
                              'B1_us': timed(h, base, nbytes, 1), 'B2_us': timed(h, base, nbytes, 2), 'B4_us': timed(h, base, nbytes, 4)}), flush=True)
    # (b) workspace offset
    for rnd in range(2):
        for off in (0, 256, 512, 1024, 2048, 4096, 8192, 65536, 65536 + 256, 262144):
            print(json.dumps({'what': 'scan_placement', 'vary': 'workspace offset, handle 0', 'workspace_offset': off, 'round': rnd,
                              'B1_us': timed(h0, base + off, nbytes, 1), 'B2_us': timed(h0, base + off, nbytes, 2), 'B4_us': timed(h0, base + off, nbytes, 4)}), flush=True)
    for h in handles:
        lib.aae_codebook_destroy(h)


if __name__ == '__main__':
    main()
