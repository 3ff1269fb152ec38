#!/usr/bin/env python
"""A/B of the single-codebook B = 1 stream scan across BUILDS of the library, on ONE box, alternating (the kernel
period drifted 10.5 -> 12.7 -> 14.3 us over the closing runs of rounds 4 / 5 without a scan code change between the last two).

    python tools/scan_b1_ab.py tag=path/to/lib.so [tag=path ...] [--rounds 6] [--reps 2000]

Every library is loaded into THIS process (ctypes; torch owns the device memory), gets its own codebook handle over the same 92232 x 128
fp32 rows and the same query, and is timed with ITS OWN aae_codebook_nn_timed (queries queued back to back from C between two HIP events:
the kernel period, no per-call host cost) at B = 1, 2, 4 -- round-robin over the libraries, `rounds` times, so that clock / thermal drift
of the box hits all of them alike.  One JSON line per (round, library); a summary line per library at the end (median / min / max).
--handles=H: H codebook handles per library, created interleaved across the libraries (WHERE a handle's 47 MB land moves the period by 10-20 %
-- three byte-identical copies of one library gave 12.2 / 13.0 / 13.2 us, each stable: profiles/r15/scan_identical_library_copies.jsonl -- so
builds are compared by their median over several placements)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from augmentedautoencoder_amd import synth      # noqa: E402


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.aae_codebook_create.restype = i32
    lib.aae_codebook_create.argtypes = [vp, i32, i32, i32, i32, ctypes.POINTER(vp)]
    lib.aae_codebook_workspace_bytes.restype = sz
    lib.aae_codebook_workspace_bytes.argtypes = [vp, i32, i32]
    lib.aae_codebook_nn_timed.restype = i32
    lib.aae_codebook_nn_timed.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, sz, vp, i32, ctypes.POINTER(ctypes.c_float)]
    lib.aae_codebook_destroy.restype = None
    lib.aae_codebook_destroy.argtypes = [vp]
    lib.aae_last_error.restype = ctypes.c_char_p
    return lib


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    opts = dict(a[2:].split('=') for a in sys.argv[1:] if a.startswith('--') and '=' in a)
    rounds, reps = int(opts.get('rounds', 6)), int(opts.get('reps', 2000))
    dev = torch.device('cuda', 0)
    E = synth.make_codebook(92232, 128, seed=7, planted_duplicates=8)
    z = torch.randn(4, 128, device=dev)
    idx = torch.empty(4, dtype=torch.int64, device=dev)
    score = torch.empty(4, dtype=torch.float32, device=dev)
    builds = []
    nh = int(opts.get('handles', 1))
    libs = [(a.split('=', 1)[0], a.split('=', 1)[1], load(a.split('=', 1)[1])) for a in args]
    for k in range(nh):
        for tag, path, lib in libs:
            h = ctypes.c_void_p()
            rc = lib.aae_codebook_create(E.ctypes.data, 92232, 128, 1, 0, ctypes.byref(h))    # (dtype 1 = AAE_DTYPE_F32, host source)
            if rc:
                raise RuntimeError('%s: aae_codebook_create rc=%d %s' % (tag, rc, lib.aae_last_error()))
            nbytes = max(int(lib.aae_codebook_workspace_bytes(h, b, 1)) for b in (1, 2, 4))
            ws = torch.zeros(nbytes + 256, dtype=torch.uint8, device=dev)
            builds.append((tag if nh == 1 else '%s#%d' % (tag, k), path, lib, h, ws, nbytes))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    samples = {tag: {1: [], 2: [], 4: []} for tag, *_ in builds}

    def timed(lib, h, ws, nbytes, B, n):
        ms = ctypes.c_float()
        ptr = ws.data_ptr() + (-ws.data_ptr()) % 256
        rc = lib.aae_codebook_nn_timed(h, z.data_ptr(), B, 1, 1, idx.data_ptr(), score.data_ptr(), ptr, nbytes, stream, n, ctypes.byref(ms))
        if rc:
            raise RuntimeError('aae_codebook_nn_timed rc=%d %s' % (rc, lib.aae_last_error()))
        return ms.value * 1e3
    for tag, path, lib, h, ws, nbytes in builds:          # warm up: clocks, code objects
        for B in (1, 2, 4):
            timed(lib, h, ws, nbytes, B, 500)
    for r in range(rounds):
        order = builds if r % 2 == 0 else builds[::-1]
        for tag, path, lib, h, ws, nbytes in order:
            rec = {'what': 'scan_b1_ab', 'round': r, 'build': tag}
            for B in (1, 2, 4):
                us = timed(lib, h, ws, nbytes, B, reps)
                samples[tag][B].append(us)
                rec['B%d_kernel_period_us' % B] = round(us, 3)
            print(json.dumps(rec), flush=True)
    for tag, path, *_ in builds:
        out = {'what': 'scan_b1_ab_summary', 'build': tag, 'library': path, 'rounds': rounds, 'reps_per_sample': reps}
        for B in (1, 2, 4):
            v = np.asarray(samples[tag][B])
            out['B%d_kernel_period_us' % B] = {'median': round(float(np.median(v)), 3), 'min': round(float(v.min()), 3), 'max': round(float(v.max()), 3)}
        out['B1_frac_of_8TBps'] = round(92232 * 128 * 4 / (float(np.median(samples[tag][1])) * 1e-6) / 8e12, 3)
        print(json.dumps(out), flush=True)
    if nh > 1:
        for tag, path, lib in libs:
            out = {'what': 'scan_b1_ab_by_build', 'build': tag, 'library': path, 'handles': nh}
            for B in (1, 2, 4):
                med = np.asarray([np.median(samples['%s#%d' % (tag, k)][B]) for k in range(nh)])
                out['B%d_kernel_period_us_over_handles' % B] = {'median': round(float(np.median(med)), 3), 'min': round(float(med.min()), 3), 'max': round(float(med.max()), 3)}
            print(json.dumps(out), flush=True)
    for tag, path, lib, h, ws, nbytes in builds:
        lib.aae_codebook_destroy(h)


if __name__ == '__main__':
    main()
