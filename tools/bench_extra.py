#!/usr/bin/env python
"""Secondary measurements on the MI355X box (not the headline bench): small-batch latency,
scan-kernel bandwidth per mode, encoder-only throughput (config 3), CPU thread sweep.
Writes one JSON object per line to stdout."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from augmentedautoencoder_amd import _lib
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
from augmentedautoencoder_amd.weights import EncoderConfig
from oracle import synth


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    what = sys.argv[1:] or ['latency', 'scan', 'embed']
    cfg = EncoderConfig()
    weights = synth.make_weights(seed=2024)
    E = synth.make_codebook(92232, 128, seed=7)
    enc = EncoderEngine(cfg, weights, max_batch=1024)
    cb = CodebookEngine(E)
    nbytes = E.size * 4
    if 'latency' in what:
        for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
            x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
            ms_enc = timeit(lambda: enc.encode(x), 20 if B >= 64 else 50)
            z = enc.encode(x)
            ms_nn = timeit(lambda: cb.nn(z, 1, 1), 50)
            ms_all = timeit(lambda: cb.nn(enc.encode(x), 1, 1), 20 if B >= 64 else 50)
            _, recs = enc.encode_timed(x)
            print(json.dumps({'what': 'latency', 'B': B, 'encode_ms': round(ms_enc, 4), 'nn_ms': round(ms_nn, 4),
                              'encode+nn_ms': round(ms_all, 4), 'crops_per_s': round(B / ms_all * 1e3, 1),
                              'kernels': [(l.split(':')[0] + ':' + l.split(':')[1].split()[0], round(ms, 4)) for l, ms, _ in recs]}))
    if 'scan' in what:
        for B in (1, 2, 4, 8, 32, 48, 64, 128, 256, 512):
            z = torch.randn(B, 128, device='cuda')
            for mode, name in ((_lib.AAE_SCAN_STREAM, 'stream'), (_lib.AAE_SCAN_GEMV, 'gemv'), (_lib.AAE_SCAN_MFMA, 'mfma'), (_lib.AAE_SCAN_AUTO, 'auto')):
                if mode in (_lib.AAE_SCAN_STREAM, _lib.AAE_SCAN_GEMV) and B > 4:
                    continue
                try:
                    cb.set_scan_mode(mode)
                except ValueError:                      # (the round-1 shuffle scan: experiments build, AAE_EXPERIMENTS=1)
                    continue
                ms = timeit(lambda: cb.nn(z, 1, 1), 100)
                print(json.dumps({'what': 'scan', 'B': B, 'mode': name, 'ms': round(ms, 4),
                                  'algorithmic_GBps': round(nbytes / ms / 1e6, 1), 'frac_of_8TBps': round(nbytes / ms / 1e6 / 8000, 3)}))
            cb.set_scan_mode(_lib.AAE_SCAN_AUTO)
    if 'stagger' in what:
        x = torch.from_numpy(synth.make_crops(256, seed=1)).cuda()
        for rep in range(2):
            for st in (0, 1, 2, 3, 4, 6):
                enc.set_option('igemm_stagger', st)
                ms = timeit(lambda: enc.encode(x), 10)
                print(json.dumps({'what': 'stagger', 'kcycles': st, 'rep': rep, 'encode_ms': round(ms, 4),
                                  'encoder_tflops': round(cfg.flops_per_crop() * 256 / ms / 1e9, 2)}))
        enc.set_option('igemm_stagger', 0)
    if 'scanprof' in what:
        # few launches of each small-batch scan variant, for a rocprofv3 --kernel-trace --stats wrapper
        for B in (1, 4):
            z = torch.randn(B, 128, device='cuda')
            for mode in (_lib.AAE_SCAN_STREAM, _lib.AAE_SCAN_GEMV, _lib.AAE_SCAN_MFMA):
                cb.set_scan_mode(mode)
                for _ in range(20):
                    cb.nn(z, 1, 1)
        torch.cuda.synchronize()
        cb.set_scan_mode(_lib.AAE_SCAN_AUTO)
    if 'x3h' in what:
        x = torch.from_numpy(synth.make_crops(256, seed=1)).cuda()
        for prec, dma in ((0, 0), (1, 0), (1, 1), (1, 0), (1, 1)):
            enc.set_option('precision', prec)
            enc.set_option('x3h_dma', dma)
            ms = timeit(lambda: enc.encode(x), 10)
            _, recs = enc.encode_timed(x)
            print(json.dumps({'what': 'x3h', 'precision': prec, 'dma': dma, 'B': 256, 'encode_ms': round(ms, 4),
                              'crops_per_s': round(256 / ms * 1e3, 1), 'encoder_tflops_equiv': round(cfg.flops_per_crop() * 256 / ms / 1e9, 2),
                              'kernels': [(l.split(':')[0] + ':' + l.split(':')[1].split()[0], round(t, 4), round(f / t / 1e9, 1) if t > 0 else 0) for l, t, f in recs]}))
        enc.set_option('precision', 0)
    if 'dma' in what:
        x = torch.from_numpy(synth.make_crops(256, seed=1)).cuda()
        for dma in (0, 1, 0, 1):
            enc.set_option('igemm_dma', dma)
            ms = timeit(lambda: enc.encode(x), 10)
            _, recs = enc.encode_timed(x)
            print(json.dumps({'what': 'dma', 'igemm_dma': dma, 'B': 256, 'encode_ms': round(ms, 4),
                              'crops_per_s': round(256 / ms * 1e3, 1), 'encoder_tflops': round(cfg.flops_per_crop() * 256 / ms / 1e9, 2),
                              'kernels': [(l.split(':')[0] + ':' + l.split(':')[1].split()[0], round(t, 4), round(f / t / 1e9, 1) if t > 0 else 0) for l, t, f in recs]}))
        enc.set_option('igemm_dma', 0)
    if 'conv1' in what:
        x = torch.from_numpy(synth.make_crops(256, seed=1)).cuda()
        for tpb, tb in ((8, 1024), (16, 512), (32, 256), (11, 768), (12, 768), (6, 1536), (22, 384), (4, 2048), (2, 4096), (8, 1024), (16, 512)):
            enc.set_option('first_max_tiles_per_block', tpb)
            enc.set_option('first_target_blocks', tb)
            ts = []
            for _ in range(5):
                _, recs = enc.encode_timed(x)
                ts.append(recs[0][1])
            print(json.dumps({'what': 'conv1', 'max_tiles_per_block': tpb, 'target_blocks': tb, 'conv1_ms': [round(t, 4) for t in ts]}))
        enc.set_option('first_max_tiles_per_block', 16)
        enc.set_option('first_target_blocks', 512)
    if 'decoder' in what:
        # next row N4: Decoder.x for batches of latent codes (default shapes), kernel split + torch-CPU reference beside it
        from augmentedautoencoder_amd.engine import DecoderEngine
        from augmentedautoencoder_amd.weights import DecoderConfig
        from oracle import decoder_cpu as dref
        dcfg = DecoderConfig()
        wd = dref.make_decoder_weights(seed=4242)
        dec = DecoderEngine(dcfg, wd)
        for B in (1, 16, 256):
            z = torch.randn(B, 128, device='cuda') * 0.5
            ms = timeit(lambda: dec.decode(z), 10 if B >= 64 else 30)
            _, recs = dec.decode_timed(z)
            print(json.dumps({'what': 'decoder', 'B': B, 'ms': round(ms, 4), 'images_per_s': round(B / ms * 1e3, 1),
                              'nominal_tflops': round(dcfg.flops_per_image() * B / ms / 1e9, 2),
                              'kernels': [(l.split(' ')[0], round(t, 4), round(f / t / 1e9, 1) if t > 0 else 0) for l, t, f in recs]}))
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        zc = np.random.default_rng(0).standard_normal((8, 128)).astype(np.float32)
        dref.decoder_forward_torch(zc[:2], wd, dcfg.shape, dcfg.num_filters, dcfg.strides)
        t0 = time.perf_counter()
        dref.decoder_forward_torch(zc, wd, dcfg.shape, dcfg.num_filters, dcfg.strides)
        dt = time.perf_counter() - t0
        print(json.dumps({'what': 'decoder_cpu', 'threads': torch.get_num_threads(), 'batch': 8, 'images_per_s': round(8 / dt, 2)}))
        dec.close()
    if 'graph' in what:
        # small-batch latency: eager launches vs one HIP-graph replay (CapturedNearestNeighbour)
        from augmentedautoencoder_amd.engine import CapturedNearestNeighbour
        for B in (1, 2, 4, 8, 16):
            x = torch.from_numpy(synth.make_crops(B, seed=B)).cuda()
            ms_eager = timeit(lambda: cb.nn(enc.encode(x), 1, 1), 100, warm=5)
            cap = CapturedNearestNeighbour(enc, cb, B)
            i0, s0 = cb.nn(enc.encode(x), 1, 1)
            i1, s1 = cap(x)
            same = bool(torch.equal(i0, i1) and torch.equal(s0, s1))
            ms_graph = timeit(lambda: cap(x), 100, warm=5)
            _, recs = enc.encode_timed(x)
            print(json.dumps({'what': 'graph', 'B': B, 'eager_ms': round(ms_eager, 4), 'graph_ms': round(ms_graph, 4),
                              'identical': same, 'encoder_kernel_ms_sum': round(sum(t for _, t, _ in recs), 4),
                              'crops_per_s_graph': round(B / ms_graph * 1e3, 1)}))
    if 'stream' in what:
        # PCIe-inclusive throughput: host uint8 batches, H2D overlapped with compute (StreamingNearestNeighbour)
        from augmentedautoencoder_amd.engine import StreamingNearestNeighbour
        host = [synth.make_crops(256, seed=100 + i) for i in range(4)]
        nb = 40
        def gen():
            for i in range(nb):
                yield host[i % 4]
        sp = StreamingNearestNeighbour(enc, cb, 256)
        for _ in sp.run(host[:2]):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = [r for r in sp.run(gen())]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        xdev = [torch.from_numpy(h).cuda() for h in host]
        ok = all(np.array_equal(got[i][0], cb.nn(enc.encode(xdev[i % 4]), 1, 1)[0].cpu().numpy()) for i in range(4))
        # serial form: copy, then compute, per batch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nb):
            x = torch.from_numpy(host[i % 4]).cuda()
            idx, sc = cb.nn(enc.encode(x), 1, 1)
            idx.cpu()
        torch.cuda.synchronize()
        dt_serial = time.perf_counter() - t0
        print(json.dumps({'what': 'stream', 'batches': nb, 'batch': 256, 'overlapped_crops_per_s': round(nb * 256 / dt, 1),
                          'overlapped_ms_per_batch': round(dt / nb * 1e3, 4), 'serial_crops_per_s': round(nb * 256 / dt_serial, 1),
                          'serial_ms_per_batch': round(dt_serial / nb * 1e3, 4), 'identical_to_resident': bool(ok)}))
    if 'wide' in what:
        # f32x3h conv layers: 128x128 tiles (4 waves, 2 blocks/CU) vs 256x128 tiles (8 waves, 1 block/CU)
        x = torch.from_numpy(synth.make_crops(256, seed=1)).cuda()
        enc.set_option('precision', 1)
        z0 = None
        for wide in (0, 256, 0, 256):
            enc.set_option('x3h_wide_min_blocks', wide)
            ms = timeit(lambda: enc.encode(x), 10)
            z, recs = enc.encode_timed(x)
            z0 = z if z0 is None else z0
            print(json.dumps({'what': 'wide', 'x3h_wide_min_blocks': wide, 'encode_ms': round(ms, 4), 'crops_per_s': round(256 / ms * 1e3, 1),
                              'identical': bool(torch.equal(z, z0)),
                              'kernels': [(l.split(' ')[0], round(t, 4), round(f / t / 1e9, 1)) for l, t, f in recs[:4]]}))
        enc.set_option('x3h_wide_min_blocks', 0)
        enc.set_option('precision', 0)
    if 'breg' in what:
        # fp32 igemm: weights through LDS (0) vs straight from global memory into the MFMA B fragments (1)
        x = torch.from_numpy(synth.make_crops(256, seed=1)).cuda()
        z0 = None
        for v in (0, 1, 0, 1):
            enc.set_option('igemm_breg', v)
            ms = timeit(lambda: enc.encode(x), 10)
            z, recs = enc.encode_timed(x)
            z0 = z if z0 is None else z0
            print(json.dumps({'what': 'breg', 'igemm_breg': v, 'encode_ms': round(ms, 4), 'crops_per_s': round(256 / ms * 1e3, 1),
                              'identical': bool(torch.equal(z, z0)),
                              'kernels': [(l.split(' ')[0], round(t, 4), round(f / t / 1e9, 1)) for l, t, f in recs[:4]]}))
        enc.set_option('igemm_breg', 0)
    if 'estimator' in what:
        # next row N1: AePoseEstimator.process on a 1080p frame -- all detections of a class as one batch vs the
        # reference's flow (one crop + one B=1 query per detection)
        import configparser
        from augmentedautoencoder_amd import session as S
        from augmentedautoencoder_amd.codebook import Codebook
        from augmentedautoencoder_amd.dataset import Dataset
        from augmentedautoencoder_amd.encoder import Encoder
        from augmentedautoencoder_amd.pose_estimator import AePoseEstimator, BoundingBox
        targs = configparser.ConfigParser()
        targs.read_string("[Dataset]\nH: 128\nW: 128\nC: 3\nRADIUS: 700\nPAD_FACTOR: 1.2\nK: [1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]\n"
                          "[Embedding]\nEMBED_BB: True\nMIN_N_VIEWS: 2562\nNUM_CYCLO: 36\n")
        S.reset_default_graph()
        books = {}
        for k, name in enumerate(['obj_a', 'obj_b']):
            ds = Dataset('', h=128, w=128, c=3, min_n_views=2562, radius=700, num_cyclo=36)
            with S.variable_scope(name):
                e = Encoder(S.Placeholder((128, 128, 3)), 128, [128, 256, 512, 512], 5, [2, 2, 2, 2], False)
                c = Codebook(e, ds, True)
            e.load_weights(synth.make_weights(seed=50 + k))
            c.assign_embedding(synth.make_codebook(92232, 128, seed=60 + k))
            r = np.random.default_rng(70 + k)
            c.assign_obj_bbs(np.stack([r.integers(250, 350, 92232), r.integers(180, 260, 92232), r.integers(80, 200, 92232), r.integers(80, 200, 92232)], 1))
            books[name] = c
        est = AePoseEstimator(codebooks=books, train_args={'obj_a': targs, 'obj_b': targs})
        rng = np.random.default_rng(0)
        img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
        camK = np.array([[1075.65, 0, 960.0], [0, 1073.9, 540.0], [0, 0, 1]])
        for D in (1, 4, 16, 64):
            dets = []
            for i in range(D):
                x, y, w, h = rng.uniform(0, 1500), rng.uniform(0, 800), rng.uniform(60, 400), rng.uniform(60, 270)
                dets.append(BoundingBox(xmin=x / 1920, xmax=(x + w) / 1920, ymin=y / 1080, ymax=(y + h) / 1080, classes={'obj_a' if i % 3 else 'obj_b': 1.0}))
            reps = max(10, 400 // D)                    # (10 calls of a 0.2 ms frame measured the first calls after a pause, not the loop)
            for _ in range(max(3, reps // 10)):
                est.process(dets, img, camK)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                est.process(dets, img, camK)
            torch.cuda.synchronize()
            dt_batched = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                for d in dets:
                    est.process([d], img, camK)
            torch.cuda.synchronize()
            dt_loop = (time.perf_counter() - t0) / reps
            print(json.dumps({'what': 'estimator', 'detections': D, 'image': '1080x1920', 'batched_ms': round(dt_batched * 1e3, 3),
                              'per_detection_loop_ms': round(dt_loop * 1e3, 3), 'detections_per_s_batched': round(D / dt_batched, 1)}))
    if 'n256' in what:
        x = torch.from_numpy(synth.make_crops(256, seed=1)).cuda()
        z0 = None
        for v in (0, 1, 0, 1):
            enc.set_option('igemm_breg_wide', v)
            ms = timeit(lambda: enc.encode(x), 10)
            z, recs = enc.encode_timed(x)
            z0 = z if z0 is None else z0
            print(json.dumps({'what': 'n256', 'igemm_breg_wide': v, 'encode_ms': round(ms, 4), 'crops_per_s': round(256 / ms * 1e3, 1),
                              'identical': bool(torch.equal(z, z0)),
                              'kernels': [(l.split(' ')[0], round(t, 4), round(f / t / 1e9, 1)) for l, t, f in recs[:4]]}))
        enc.set_option('igemm_breg_wide', 0)
    if 'config5' in what:
        # 368928 x 128 bf16 codebook (94.4 MB), batched queries, arg-max and top-5
        E5 = synth.make_codebook(368928, 128, seed=11)
        cb5 = CodebookEngine(E5, dtype='bf16')
        for B in (1, 32, 256):
            z = torch.randn(B, 128, device='cuda')
            ms1 = timeit(lambda: cb5.nn(z, 1, 1), 30)
            ms5 = timeit(lambda: cb5.nn(z, 5, 1), 10)
            print(json.dumps({'what': 'config5', 'B': B, 'argmax_ms': round(ms1, 4), 'top5_ms': round(ms5, 4),
                              'argmax_algorithmic_GBps': round(368928 * 128 * 2 / ms1 / 1e6, 1),
                              'argmax_tflops_fp32_equiv': round(2.0 * B * 368928 * 128 / ms1 / 1e9, 2)}))
        cb5.close()
    if 'crops' in what:
        from augmentedautoencoder_amd.engine import crop_resize
        from augmentedautoencoder_amd.pose_estimator import AePoseEstimator
        rng = np.random.default_rng(0)
        img = torch.from_numpy(rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)).cuda()
        for D in (1, 16, 64, 256):
            boxes = [[float(rng.uniform(0, 1500)), float(rng.uniform(0, 800)), float(rng.uniform(60, 400)), float(rng.uniform(60, 280))] for _ in range(D)]
            rows = AePoseEstimator.box_rows(boxes, 1.2)
            ms = timeit(lambda: crop_resize(img, rows, (128, 128)), 30)
            print(json.dumps({'what': 'crops', 'detections': D, 'ms': round(ms, 4), 'crops_per_s': round(D / ms * 1e3, 1),
                              'out_GBps': round(D * 49152 / ms / 1e6, 2)}))
    if 'embed' in what:
        # config 3: encoder-only over 92232 views in batches of 64 (reference BATCH_SIZE) / 256 / 1024
        for bs in (64, 256, 1024):
            x = torch.from_numpy(synth.make_crops(bs, seed=3)).cuda()
            n_batches = -(-92232 // bs)
            ms = timeit(lambda: enc.encode(x), 10)
            print(json.dumps({'what': 'embed', 'batch': bs, 'ms_per_batch': round(ms, 4), 'crops_per_s': round(bs / ms * 1e3, 1),
                              'seconds_for_92232_views': round(n_batches * ms * 1e-3, 3)}))
    if 'cpu' in what:
        from oracle import reference_cpu as ref
        crops = synth.make_crops(64, seed=1)
        xf = ref.input_to_float(crops)
        for nt in (8, 16, 32, 64, 128, 256):
            if nt > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(nt)
            ref.encoder_forward_torch(xf[:8], weights, [2, 2, 2, 2], False, 'float32')
            t0 = time.perf_counter()
            reps = 2
            for _ in range(reps):
                ref.encoder_forward_torch(xf, weights, [2, 2, 2, 2], False, 'float32')
            dt = (time.perf_counter() - t0) / reps
            print(json.dumps({'what': 'cpu', 'threads': nt, 'batch': 64, 'crops_per_s': round(64 / dt, 1)}))


if __name__ == '__main__':
    main()
