#!/usr/bin/env python
"""A frame with detections of several object classes (one AAE per class: m3_interface/ae_pose_estimator.py:61-78):
per-object calls against the grouped query (one launch per layer across the objects).  One JSON object per line.

    python tools/bench_multi.py                 # objects x detections sweep, sequential vs grouped, cold by construction
    python tools/bench_multi.py --profile 1     # a loop of 8 x 1 frames, grouped then sequential: for rocprofv3 --kernel-trace --stats
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from augmentedautoencoder_amd import synth
from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine, MultiObjectQuery
from augmentedautoencoder_amd.weights import EncoderConfig

PEAK_F32_TFLOPS, PEAK_HBM_GBPS = 157.3, 8000.0


def time_us(fn, reps, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    profile = int(sys.argv[sys.argv.index('--profile') + 1]) if '--profile' in sys.argv else 0
    n_max = 8 if profile else 16
    cfg = EncoderConfig()
    dev = torch.device('cuda', 0)
    objs = [(EncoderEngine(cfg, synth.make_weights(seed=2024 + i), device=dev, max_batch=64),
             CodebookEngine(synth.make_codebook(92232, 128, seed=7 + i), device=dev)) for i in range(n_max)]
    xs = [torch.from_numpy(synth.make_crops(4, seed=500 + i)).to(dev) for i in range(n_max)]
    cb_bytes = 92232 * 128 * 4

    def frames(n_obj, d):
        sel = objs[:n_obj]

        def seq():
            for (e, c), xi in zip(sel, xs):
                e.encode_nn(c, xi[:d], 1)
        mq = MultiObjectQuery([(e, c, d) for e, c in sel], device=dev)
        xcat = torch.cat([xi[:d] for xi in xs[:n_obj]]).contiguous()
        return seq, (lambda: mq(xcat)), mq

    if profile:
        seq, grp, mq = frames(8, profile)
        if '--per-object-plans' in sys.argv:
            for e, _ in objs:
                e.set_option('multi_group_plan', 0)
        for _ in range(40):
            grp()
        torch.cuda.synchronize()
        if '--sequential' in sys.argv:
            for _ in range(40):
                seq()
            torch.cuda.synchronize()
        return
    for n_obj in (1, 2, 4, 8, 16):
        for d in (1, 2, 4):
            seq, grp, mq = frames(n_obj, d)
            t_seq = time_us(seq, 40)
            t_grp = time_us(grp, 40)
            # ... with every object on its own launch plan (option multi_group_plan = 0): bit-identical to the per-object calls, checked on the timed inputs
            for e, _ in objs:
                e.set_option('multi_group_plan', 0)
            seq0, grp0, mq0 = frames(n_obj, d)
            t_grp0 = time_us(grp0, 40)
            want = [objs[k][0].encode_nn(objs[k][1], xs[k][:d], 1) for k in range(n_obj)]
            z, idx, score = grp0()
            same = all(torch.equal(z[k * d:(k + 1) * d], w[0]) and torch.equal(idx[k * d:(k + 1) * d], w[1][:, 0]) and torch.equal(score[k * d:(k + 1) * d], w[2][:, 0])
                       for k, w in enumerate(want))
            for e, _ in objs:
                e.set_option('multi_group_plan', 1)
            zg, idxg, _ = grp()
            same_idx = bool(torch.equal(idxg, idx))
            print(json.dumps({'what': 'frame', 'objects': n_obj, 'detections_per_object': d, 'sequential_us': round(t_seq, 1), 'grouped_us': round(t_grp, 1),
                              'grouped_per_object_plans_us': round(t_grp0, 1),
                              'grouped_over_sequential': round(t_grp / t_seq, 3), 'us_per_detection_grouped': round(t_grp / (n_obj * d), 2),
                              'launches_grouped': mq.launches, 'launches_sequential': 6 * n_obj, 'per_object_plans_bit_identical': bool(same),
                              'group_plan_same_indices': same_idx, 'group_plan_z_max_rel_diff': float((zg - z).abs().max() / z.abs().max()),
                              'mfma_floor_us': round(n_obj * d * cfg.flops_per_crop() / (PEAK_F32_TFLOPS * 1e6), 1),
                              'hbm_floor_us': round(n_obj * (cfg.param_bytes() + cb_bytes) / (PEAK_HBM_GBPS * 1e3), 1),
                              'resident_MB': round(n_obj * (cfg.param_bytes() + cb_bytes) / 1e6, 1)}), flush=True)
    # a frame whose classes have DIFFERENT detection counts (the usual case): per-object plans need one group per count
    for counts in ([1, 1, 2, 4, 1, 3, 1, 2], [2, 1, 1, 3], [4, 4, 1, 1, 1, 1, 2, 2, 1, 1, 3, 1]):
        sel = objs[:len(counts)]
        xcat = torch.cat([xs[k][:n] for k, n in enumerate(counts)]).contiguous()

        def seq():
            for (e, c), xi, n in zip(sel, xs, counts):
                e.encode_nn(c, xi[:n], 1)
        mq = MultiObjectQuery([(e, c, n) for (e, c), n in zip(sel, counts)], device=dev)
        t_seq, t_grp = time_us(seq, 40), time_us(lambda: mq(xcat), 40)
        launches = mq.launches
        for e, _ in objs:
            e.set_option('multi_group_plan', 0)
        mq0 = MultiObjectQuery([(e, c, n) for (e, c), n in zip(sel, counts)], device=dev)
        t_grp0 = time_us(lambda: mq0(xcat), 40)
        launches0 = mq0.launches
        for e, _ in objs:
            e.set_option('multi_group_plan', 1)
        print(json.dumps({'what': 'mixed_frame', 'detections_per_object': counts, 'detections': sum(counts), 'sequential_us': round(t_seq, 1), 'grouped_us': round(t_grp, 1),
                          'grouped_per_object_plans_us': round(t_grp0, 1), 'grouped_over_sequential': round(t_grp / t_seq, 3), 'launches_grouped': launches,
                          'launches_grouped_per_object_plans': launches0, 'launches_sequential': 6 * len(counts),
                          'mfma_floor_us': round(sum(counts) * cfg.flops_per_crop() / (PEAK_F32_TFLOPS * 1e6), 1)}), flush=True)
    # the codebook stage alone
    for n_obj in (1, 2, 4, 8, 16):
        for d in (1, 4):
            z = torch.randn(n_obj * d, 128, device=dev)
            mq = MultiObjectQuery([(None, c, d) for _, c in objs[:n_obj]], device=dev)
            t_grp = time_us(lambda: mq.nn(z), 100, warm=10)

            def seq():
                for k, (_, c) in enumerate(objs[:n_obj]):
                    c.nn(z[k * d:(k + 1) * d], 1, 1)
            t_seq = time_us(seq, 100, warm=10)
            print(json.dumps({'what': 'scan', 'codebooks': n_obj, 'queries_per_codebook': d, 'grouped_us': round(t_grp, 2), 'sequential_us': round(t_seq, 2),
                              'MB': round(n_obj * cb_bytes / 1e6, 1), 'grouped_frac_of_hbm_peak': round(n_obj * cb_bytes / t_grp / 1e3 / PEAK_HBM_GBPS, 3),
                              'sequential_frac_of_hbm_peak': round(n_obj * cb_bytes / t_seq / 1e3 / PEAK_HBM_GBPS, 3)}), flush=True)


if __name__ == '__main__':
    main()
