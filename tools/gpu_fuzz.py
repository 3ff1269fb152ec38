#!/usr/bin/env python
"""Random [Network] shapes / batch sizes / kernel-variant switches on the real GPU against the fp64 torch oracle
(encoder layer outputs, latent, codebook nn incl. upright and top-k).  Not part of the test suite; run through
gpurun:  python tools/gpu_fuzz.py [n_cases] [seed]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig
from oracle import reference_cpu as ref
from oracle import synth


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    torch.set_num_threads(32)
    worst = {'layer': 0.0, 'z': 0.0, 'cos': 0.0}
    for case in range(n_cases):
        n_layers = int(rng.integers(2, 5))
        H = int(rng.choice([32, 48, 64, 96, 128]))
        W = int(rng.choice([32, 48, 64, 96, 128]))
        C = int(rng.choice([1, 3]))
        filters = [int(rng.choice([32, 64, 96, 128, 160, 256])) for _ in range(n_layers)]
        strides = [int(rng.choice([1, 2, 2])) for _ in range(n_layers)]
        latent = int(rng.choice([32, 64, 128]))
        bn = bool(rng.integers(0, 2))
        cfg = EncoderConfig((H, W, C), filters, strides, 5, latent, bn)
        if cfg.flatten_size % 32 or cfg.flatten_size > 262144:
            continue
        B = int(rng.choice([1, 2, 3, 4, 5, 7, 9, 12, 16, 33, 64, 100, 129, 256]))
        if B * max(a * b * c for a, b, _, _, _, c in [(s[3], s[4], 0, 0, 0, s[5]) for s in cfg.layer_shapes()]) > 6e7:
            B = min(B, 8)
        w = synth.make_weights(seed=case, shape=cfg.shape, num_filter=filters, strides=strides, latent=latent, batch_norm=bn)
        x = synth.make_crops(B, seed=1000 + case, shape=cfg.shape)
        enc = EncoderEngine(cfg, w, max_batch=max(B, 1))
        opts = {k: int(rng.integers(0, 2)) for k in ('igemm_dma', 'igemm_breg', 'igemm_breg_wide', 'dense_gemv', 'x3h_dma',
                                                     'wavek', 'wavek_dense', 'gemv_ticket', 'ticket_prep', 'compact_workspace')}
        # the small-batch igemm family: wave count / prefetch depth / tile shape switch / how far up in batch size it is used
        opts['wavek_waves'] = int(rng.choice([4, 8]))
        opts['wavek_depth'] = int(rng.choice([2, 3]))
        opts['wavek_narrow_max_tiles'] = int(rng.choice([0, 16, 64]))
        opts['wavek_max_tiles'] = int(rng.choice([64, 256, 512]))
        opts['wavek_tiny_max_tiles'] = int(rng.choice([0, 64, 512]))            # 32 x 32 wave tiles
        opts['first_group_split_max_tiles'] = int(rng.choice([0, 128, 4096]))   # conv1: one block per 32-pixel group
        opts['x3h_wide256'] = int(rng.integers(0, 2))                            # f32x3h: 256 x 256 tiles ...
        opts['x3h_wide256_min_blocks'] = int(rng.choice([1, 256]))               # ... also on grids that do not fill the chip
        opts['planner_cost_model'] = int(rng.integers(0, 2))                     # per-layer kernel choice by estimated time (5 <= B < 256)
        opts['detect_chain'] = int(rng.integers(0, 2))                           # B <= 4: conv2 ... scan as one persistent launch where it applies
        opts['wavek_tiny_waves'] = int(rng.choice([4, 8]))                       # 32 x 32 wave tiles on eight waves ...
        opts['wavek_pingpong'] = int(rng.integers(0, 2))                         # ... and the barrier-paced schedule of the 8-wave blocks
        # round 4, second half: tiles beyond the last full round cut in K (by the estimate, or forced on any un-split layer), K splits
        # sized for 1 ... 3 blocks per CU, the dense GEMV up to 8 rows, B = 3 under the estimate
        opts['wavek_tail_split'] = int(rng.integers(0, 2))
        opts['wavek_force_tail_tiles'] = int(rng.choice([0, 0, 1, 5, 40, 200]))
        opts['wavek_force_tail_g'] = int(rng.choice([2, 3, 5, 8]))
        opts['wavek_g_boost'] = int(rng.choice([1, 2, 3]))
        opts['dense_gemv_max_batch'] = int(rng.choice([4, 8]))
        opts['planner_cost_batch3'] = int(rng.integers(0, 2))
        for k, v in list(opts.items()):
            try:
                enc.set_option(k, v)
            except ValueError:                             # a kernel variant of the experiments build (AAE_EXPERIMENTS=1): the product library keeps its default
                opts[k] = 'default'
        precision = int(rng.integers(0, 2)) if cfg.shape[2] in (1, 3) and all(f % 32 == 0 for f in filters) else 0
        try:
            enc.set_option('precision', precision)
        except ValueError:
            precision = 0
        z, recs = enc.encode_timed(x)
        z64, acts = ref.encoder_forward_torch(ref.input_to_float(x), w, strides, bn, 'float64', return_activations=True)
        tol = 2e-5
        for i, a in enumerate(acts):
            if opts['compact_workspace'] and i + 2 < len(acts):
                continue                                   # overwritten by layer i + 2 (two alternating activation buffers)
            g = enc.activation(i).cpu().numpy()
            err = float(np.abs(g - a).max() / max(np.abs(a).max(), 1e-9))
            worst['layer'] = max(worst['layer'], err)
            assert err < tol, (case, 'layer', i, err, cfg.shape, filters, strides, B, opts, precision, [l for l, _, _ in recs])
        errz = float(np.abs(z.cpu().numpy() - z64).max() / np.abs(z64).max())
        worst['z'] = max(worst['z'], errz)
        assert errz < tol, (case, 'z', errz, cfg.shape, filters, strides, B, opts, precision)
        if latent == 128:
            N = int(rng.choice([300, 4097, 92232]))
            dtype = str(rng.choice(['f32', 'bf16']))
            E = synth.make_codebook(N, 128, seed=case, planted_duplicates=min(8, N // 72))
            cb = CodebookEngine(E, dtype=dtype)
            # AAE_SCAN_AUTO, or one of its A/B forms: the walking stream scan (B <= 4), packed-query planes instead of the in-scan
            # normalisation, two instead of four waves per query group (B <= 32)
            try:
                cb.set_scan_mode(int(rng.choice([0, 0, 6, 7, 8])))
            except ValueError:                             # (the walking stream scan: experiments build)
                cb.set_scan_mode(0)
            cs = cb.similarity(z).cpu().numpy()
            from augmentedautoencoder_amd.weights import bf16_bits_to_f32, to_bf16_bits
            Eo = bf16_bits_to_f32(to_bf16_bits(E)) if dtype == 'bf16' else E
            cs64 = ref.cos_similarity(z.cpu().numpy(), Eo)
            errc = float(np.abs(cs - cs64).max())
            worst['cos'] = max(worst['cos'], errc)
            assert errc < 1e-5, (case, 'cos', errc, N, dtype, B)
            idx, sc = cb.nn(z, 1, 1)
            assert np.array_equal(idx[:, 0].cpu().numpy(), np.argmax(cs, axis=1)), (case, 'argmax', N, dtype, B)
            # the fused call (encoder + top-1 scan, tickets prepared by the first kernel where it can) gives the same bits
            stride = int(rng.choice([1, 36]))
            zf, i_f, s_f = enc.encode_nn(cb, x, stride)
            i_s, s_s = cb.nn(z, 1, stride)
            assert torch.equal(zf, z) and torch.equal(i_f, i_s) and torch.equal(s_f, s_s), (case, 'fused', stride, N, dtype, B, opts)
            up, _ = cb.nn(z, 1, 36)
            assert np.array_equal(up[:, 0].cpu().numpy(), ref.nearest_indices_reference(cs, 1, upright=True, num_cyclo=36)), (case, 'upright')
            k = int(rng.integers(2, 9))
            ik, _ = cb.nn(z, k, 1)
            assert np.array_equal(ik.cpu().numpy(), ref.topk_canonical(cs, k)), (case, 'topk', k, N, dtype, B)
            cb.close()
        enc.close()
        print(json.dumps({'case': case, 'shape': cfg.shape, 'filters': filters, 'strides': strides, 'latent': latent, 'bn': bn, 'B': B,
                          'precision': precision, 'opts': opts, 'kernels': [l.split(':')[1].split(' ')[0] for l, _, _ in recs]}), flush=True)
    print(json.dumps({'cases': n_cases, 'worst_rel_layer': worst['layer'], 'worst_rel_z': worst['z'], 'worst_abs_cos': worst['cos']}))


if __name__ == '__main__':
    main()
