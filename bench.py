#!/usr/bin/env python
"""Headline benchmark: crops/sec (encode + codebook-NN), 128x128x3 crops against a
92232x128 fp32 codebook (BASELINE.json metric), one object per GPU.

    python bench.py --gpus N --steps K --warmup W

Launch forms.  N = 1: this process is the only rank.  N > 1: either start it under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment), or start it plainly -- `python bench.py --gpus N ...` with no
WORLD_SIZE in the environment re-executes itself under torch.distributed.run on a free local port, one rank per visible
GPU, and passes rank 0's JSON line through.  Fewer than N visible GPUs: ONE JSON line with an `error` field, exit code 2,
no traceback.  `--dry-run-dist` takes the same launch path on CPU (gloo) up to process-group creation and one all_gather.

A step = one pass of the hot path over one batch of 256 device-resident uint8
crops per GPU: 4-conv encoder -> dense -> l2-normalise -> codebook scan ->
arg-max (+ for N > 1 the RCCL all_gather of the (index, score) pairs, the only
collective on this path).  Rank 0 prints ONE JSON line.

`value` is measured in exact fp32 (fp32 MFMA, bit-equal to an fma chain).  The same
line carries a `split_precision` object: the identical step in the opt-in f32x3h
mode (fp32 in/out, every product as 3 fp16 MFMAs on (hi, lo) operand pairs, fp32
accumulate; same parity tolerances) -- reported beside
the headline, never as it.  `--precision f32x3h` makes that mode the measured one.

After the headline (N = 1 only, `--no-extras` skips them) the same line carries the other BASELINE configs and the
per-detection regime, each measured in this run: `latency` (B = 1, 2, 4 fused encode+nn, eager and as one HIP-graph
replay), `scan` (the codebook query alone: whole call warm, cold = rotating over 8 codebook copies that exceed the
256 MB Infinity Cache, fraction of the 8 TB/s HBM peak), `config3` (encode 92232 views: ae_embed), `config5`
(368928 x 128 bf16 codebook, B = 256, arg-max and top-5), `pcie_inclusive` (host uint8 batches, H2D overlapped)
and `decoder`.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 256
N_ROWS = 92232
PEAK_F32_TFLOPS = 157.3        # MI355X fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_X3H_TFLOPS = 2500.0 / 3   # f32x3h: dense fp16 MFMA peak / 3 MFMAs per fp32-equivalent product
PEAK_HBM_GBPS = 8000.0         # HBM3E spec


def cpu_baseline(weights, E, crops, min_seconds=10.0, max_iters=60):
    """The oracle's fp32 torch-CPU restatement of the same step on a bounded sample
    (checker code timed as the CPU reference point; never on the product path).  Three legs, as BASELINE.md section 2
    lays them out: B = 64 end to end (the value), its encode-only / NN-only split, and B = 1 -- the reference's real
    operating point, one session.run per detection (m3_interface/ae_pose_estimator.py:143-170)."""
    import numpy as np
    import torch
    from oracle import reference_cpu as ref
    # thread count: the best point of a sweep on the MI355X box's host (2 x EPYC 9575F, 256
    # hardware threads): 8 -> 151, 16 -> 189, 32 -> 223, 64 -> 126, 128 -> 58, 256 -> 14 crops/s
    # (tools/bench_extra.py cpu); oneDNN oversubscribes badly beyond 32 threads at this size.
    nthreads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(nthreads)
    sample = crops[:64]

    def encode(x):
        return ref.encoder_forward_torch(ref.input_to_float(x), weights, [2, 2, 2, 2], False, 'float32')

    def nn(z):
        cs = ref.cos_similarity(z, E, np.float32)
        return ref.nearest_indices_reference(cs, 1)

    def run(fn, seconds, cap):
        done, t0 = 0, time.perf_counter()
        while True:
            fn()
            done += 1
            el = time.perf_counter() - t0
            if el >= seconds or done >= cap:
                return done, el
    encode(sample[:8])                                             # warm-up
    done, el = run(lambda: nn(encode(sample)), min_seconds, max_iters)
    z64 = encode(sample)
    d_e, t_e = run(lambda: encode(sample), 1.5, 20)
    d_n, t_n = run(lambda: nn(z64), 1.0, 50)
    one = sample[:1]
    encode(one)
    # B = 1 with the NN leg in torch too: one runtime, one thread pool -- what a single-runtime reference (TensorFlow) does per detection.
    # (NumPy's BLAS pool behind torch's OpenMP pool costs 25+ ms of hand-over per call at B = 1: kept as ..._numpy_nn for the record)
    E_t = torch.from_numpy(np.ascontiguousarray(E, dtype=np.float32))

    def nn_torch(z):
        zt = torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32))
        zn = zt / zt.norm(dim=1, keepdim=True).clamp_min(1e-12)
        return torch.argmax(zn @ E_t.T, dim=1)
    nn_torch(z64[:1])
    d_1, t_1 = run(lambda: nn_torch(encode(one)), 2.0, 400)
    d_1e, t_1e = run(lambda: encode(one), 1.0, 400)
    d_1n, t_1n = run(lambda: nn_torch(z64[:1]), 0.7, 400)
    d_1x, t_1x = run(lambda: nn(encode(one)), 1.0, 100)
    model = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            model = next((l.split(':', 1)[1].strip() for l in f if l.startswith('model name')), 'unknown')
    except OSError:
        pass
    return {'value': round(done * len(sample) / el, 2), 'unit': 'crops/s', 'cores': nthreads, 'kind': 'port',
            'sample': '%d x %d crops, fp32 torch-CPU (oneDNN) encoder + numpy fp32 codebook matmul/argmax, %.1f s'
                      % (done, len(sample), el),
            'B64': {'encode+nn_crops_per_s': round(done * 64 / el, 2), 'encode_only_crops_per_s': round(d_e * 64 / t_e, 2),
                    'nn_only_crops_per_s': round(d_n * 64 / t_n, 1)},
            'B1': {'encode+nn_crops_per_s': round(d_1 / t_1, 2), 'encode+nn_ms': round(t_1 / d_1 * 1e3, 2),
                   'encode_only_ms': round(t_1e / d_1e * 1e3, 2), 'nn_only_ms': round(t_1n / d_1n * 1e3, 2),
                   'encode+nn_ms_numpy_nn': round(t_1x / d_1x * 1e3, 2),
                   'note': "the reference's operating point: one session.run + arg-max per detection, encoder AND codebook matmul in one runtime (torch-CPU, one "
                           "thread pool).  ..._numpy_nn: the same with the matmul in NumPy -- alternating between torch's OpenMP pool and NumPy's BLAS pool every call "
                           "costs more than the sum of the parts (an artefact of mixing runtimes, not of the reference)"},
            'host': {'cpu_model': model, 'logical_cpus': os.cpu_count(),
                     'note': 'threads = best point of a sweep (8..256) on this host; more threads are slower'}}


METRIC = 'crops/sec (encode+codebook-NN), 128x128x3 vs 92232x128 codebook, 1/8 GPU'


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher: become the launcher.  One rank per GPU under
    torch.distributed.run on 127.0.0.1 and a free port; the children inherit stdout, so rank 0's JSON line is this
    command's JSON line.  Returns the exit code."""
    import socket
    import subprocess
    if not args.dry_run_dist:
        import torch
        visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if visible < args.gpus:
            print(json.dumps({'metric': METRIC, 'value': None, 'unit': 'crops/s', 'n_gpus': args.gpus, 'steps': args.steps,
                              'warmup': args.warmup, 'error': '--gpus %d but %d GPU(s) visible to this process' % (args.gpus, visible),
                              'visible_gpus': visible}))
            return 2
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '8')
    # a free port is found by bind + close; between the close and torch.distributed.run's own bind another process (parallel CI)
    # may take it: a launch that dies of "address already in use" is repeated on another port
    rc = 1
    for attempt in range(4):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
        p = subprocess.run(cmd, env=env, stderr=subprocess.PIPE)
        err = p.stderr.decode('utf-8', 'replace')
        rc = p.returncode
        if rc != 0 and attempt < 3 and ('address already in use' in err.lower() or 'eaddrinuse' in err.lower()):
            continue
        sys.stderr.write(err)
        break
    return rc


def dry_run_dist(args):
    """The N-rank launch path up to process-group creation, on CPU: gloo group, one all_gather of the rank ids, rank 0
    prints one JSON line.  (tests/test_bench_launch.py; the GPU path differs only in backend and device.)"""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    mine = torch.tensor([rank], dtype=torch.int64)
    every = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(every, mine)
    dist.barrier()
    if rank == 0:
        print(json.dumps({'dry_run_dist': True, 'backend': 'gloo', 'n_gpus': args.gpus, 'world_size': dist.get_world_size(),
                          'ranks_seen': [int(t.item()) for t in every], 'launched_by': os.environ.get('AAE_BENCH_LAUNCHER', 'external')}))
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', choices=['f32', 'f32x3h'], default='f32',
                    help='arithmetic of the measured step: f32 (default, exact fp32 MFMA) or f32x3h (split precision)')
    ap.add_argument('--no-split-precision', action='store_true', help='skip the extra f32x3h measurement')
    ap.add_argument('--profile-steps', type=int, default=5, help='instrumented per-kernel timing passes after the timed region')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the RCCL process group and take the multi-GPU code path (pair packing, all_gather, config4) even at world size 1')
    ap.add_argument('--full-extras', action='store_true', help='(kept for old command lines: config3 at batch 64 is part of the default run since round 6)')
    ap.add_argument('--no-config3-b64', action='store_true',
                    help='skip config3 at the reference batch size 64 (profiling runs: those launches share the kernel symbols of the headline batch '
                         'and blur the per-symbol averages of rocprofv3 --stats)')
    ap.add_argument('--enc-opt', action='append', default=[], metavar='NAME=INT',
                    help='set an encoder option before measuring (kernel-variant A/B under a profiler); recorded in config')
    ap.add_argument('--no-extras', action='store_true', help='skip the secondary measurements (latency, scan, config3, config5, pcie, decoder)')
    ap.add_argument('--no-config4', action='store_true', help='skip the mixed-batch config4 measurement (eight objects, 256 crops)')
    ap.add_argument('--config4', action='store_true', help='measure config4 even under --no-extras')
    ap.add_argument('--dry-run-dist', action='store_true',
                    help='take the N-rank launch path on CPU (gloo): process group + one all_gather, no GPU work (launch-path test)')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        os.environ['AAE_BENCH_LAUNCHER'] = 'self'
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.dry_run_dist:
        if 'WORLD_SIZE' not in os.environ:                  # N = 1 without a launcher: a one-rank group in this process
            os.environ.update({'RANK': '0', 'WORLD_SIZE': '1', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29534'})
        sys.exit(dry_run_dist(args))

    import torch
    import torch.distributed as dist
    from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine, pack_pairs, unpack_pairs
    from augmentedautoencoder_amd.weights import EncoderConfig
    from augmentedautoencoder_amd import synth            # seeded synthetic inputs (no oracle code on the measured path)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if rank == 0:
            print(json.dumps({'metric': METRIC, 'value': None, 'unit': 'crops/s', 'n_gpus': args.gpus,
                              'error': '--gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world)}))
        sys.exit(2)
    if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
        if rank == 0:
            print(json.dumps({'metric': METRIC, 'value': None, 'unit': 'crops/s', 'n_gpus': args.gpus,
                              'error': 'rank %d has no GPU (local rank %d, %d visible)' % (rank, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0)}))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist        # one process per GPU over RCCL; --force-dist runs that path on a single GPU
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)

    B = args.batch
    cfg = EncoderConfig()
    weights = synth.make_weights(seed=2024 + rank)            # object `rank`: its own encoder ...
    E = synth.make_codebook(N_ROWS, 128, seed=7 + rank)       # ... and its own codebook
    crops = synth.make_crops(B, seed=1234 + rank)
    enc = EncoderEngine(cfg, weights, device=dev, max_batch=B)
    cb = CodebookEngine(E, device=dev)
    x = torch.from_numpy(crops).to(dev)                       # inputs resident in HBM before the timed region
    packed = torch.empty((B, 2), dtype=torch.int64, device=dev)
    gathered = torch.empty((world * B, 2), dtype=torch.int64, device=dev) if use_dist else None

    def step():
        z = enc.encode(x)
        idx, score = cb.nn(z, 1, 1)
        if use_dist:
            pack_pairs(idx, score, None, packed)                  # one launch: (index, score bits) pairs, the gather payload
            dist.all_gather_into_tensor(gathered, packed)
        return idx, score

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(precision, winograd=1):
        """W warm-up steps, then exactly K timed steps between barrier+synchronize fences, max over
        ranks; then per-kernel durations from HIP events on the launch stream."""
        enc.set_option('precision', 1 if precision == 'f32x3h' else 0)
        enc.set_option('winograd', winograd)
        for kv in args.enc_opt:
            enc.set_option(kv.split('=')[0], int(kv.split('=')[1]))
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        rank_ms = None
        if use_dist:
            mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            every = torch.empty((world,), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(every, mine)
            rank_ms = [round(float(v) / args.steps * 1e3, 4) for v in every.tolist()]
            elapsed = float(every.max().item())               # the slowest rank's clock
        # split precision: the range flags of all forwards of the loop are read here, once, outside the timed region (no
        # host round trip per step); a batch that left the fp16 pair range would have been recomputed in exact fp32
        x3h_redone = enc.settle()
        per, order = {}, []
        for _ in range(max(args.profile_steps, 1)):
            _, recs = enc.encode_timed(x)
            for label, ms, flops in recs:
                if label not in per:
                    per[label] = [0.0, flops, 0]
                    order.append(label)
                per[label][0] += ms
                per[label][2] += 1
        kernels = []
        for label in order:
            ms = per[label][0] / per[label][2]
            kernels.append({'kernel': label, 'ms': round(ms, 4),
                            'tflops': round(per[label][1] / (ms * 1e-3) / 1e12, 2) if ms > 0 else None})
        dom = max(kernels, key=lambda k: k['ms'])
        dom_flops = per[dom['kernel']][1]
        achieved = dom_flops / (dom['ms'] * 1e-3) / 1e12
        # a polyphase-Winograd launch (csrc/kernels/conv_winograd_f32.h): its record counts the multiply-adds it EXECUTES (`points` products per
        # 2 x 2 outputs and channel pair), so that achieved / peak stays a statement about the kernel; the direct form of the same phase
        # multiplies 4 x taps per tile -- its flops / the same time is the "TF-equivalent" figure
        wino = re.search(r'conv_wino_f32 (layer|phase \d\d) \((\d+) taps as (\d+) products', dom['kernel'])
        traffic_key = dom['kernel'].split(':')[0] + ('/wino' if wino else '')
        peak = PEAK_F32_TFLOPS if precision == 'f32' else PEAK_X3H_TFLOPS
        traffic, traffic_src, busy = None, None, {}
        try:      # HBM-side bytes per launch from the committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE)
            with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
                tj = json.load(f)
            traffic = tj.get(precision, {}).get(traffic_key)
            traffic_src = tj.get(precision + '_source')
            busy = dict(tj.get('mfma' if precision == 'f32' else 'mfma_x3h', {}).get(traffic_key, {}), source=tj.get('mfma_source'))
        except Exception:
            traffic = None
        return {
            'value': round(world * B * args.steps / elapsed, 1),
            'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'roofline': {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': round(achieved, 2), 'peak': round(peak, 1),
                         'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
                         'traffic_source': ('profiles/traffic.json <- ' + (traffic_src or 'rocprofv3 PMC passes of an earlier run of this command: 2*FETCH_SIZE + WRITE_SIZE') +
                                            '; a committed measurement, not taken in this run (PMC collection needs the profiler)') if traffic is not None else None,
                         'flops_per_launch': dom_flops, 'avg_ms': dom['ms'],
                         'arithmetic': ('polyphase Winograd F(2x2): flops_per_launch = the multiply-adds the launch executes (%s products per 2x2 outputs where the '
                                        'direct form multiplies %d)' % (wino.group(3), 4 * int(wino.group(2)))) if wino else 'direct',
                         'direct_form_flops_per_launch': dom_flops * 4 * int(wino.group(2)) / int(wino.group(3)) if wino else None,
                         'tf_equivalent_of_the_direct_form': round(achieved * 4 * int(wino.group(2)) / int(wino.group(3)), 2) if wino else None,
                         'mfma_busy_frac': busy.get('mfma_busy_frac'), 'delivered_GHz_under_pmc': busy.get('delivered_GHz'),
                         'mfma_busy_source': busy.get('source'),
                         # (the split-precision mode is clock-limited: its fraction of the peak AT THE DELIVERED CLOCK, nominal 2.4 GHz)
                         'frac_of_peak_at_delivered_clock': round(achieved / (peak * busy['delivered_GHz'] / 2.4), 4) if busy.get('delivered_GHz') else None},
            'encoder_tflops': round(cfg.flops_per_crop() * B / (sum(k['ms'] for k in kernels) * 1e-3) / 1e12, 2),
            'kernels': kernels,
            'rank_ms_per_step': rank_ms,
            'x3h_batches_recomputed_in_fp32': x3h_redone if precision == 'f32x3h' else None,
        }

    main_res = measure(args.precision)
    # the same step on the direct fp32 kernels (option winograd = 0: what rounds 1-4 reported as the headline)
    direct_res = measure('f32', winograd=0) if args.precision == 'f32' and not args.no_split_precision else None
    split_res = None
    if args.precision == 'f32' and not args.no_split_precision:
        split_res = measure('f32x3h')
        enc.set_option('precision', 0)
    enc.set_option('winograd', 1)

    def time_us(fn, reps, warm=5):
        """average microseconds per call of fn over `reps` back-to-back calls (HIP events on the launch stream)"""
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

    extras = {}
    N_OBJ = 8                                          # eight objects (seeds 2024 + i / 7 + i)
    objs = {rank: (enc, cb)}

    def obj(i):
        """object i: its own encoder weights and its own 92232-row codebook (object `rank` is the headline's)"""
        if i not in objs:
            objs[i] = (EncoderEngine(cfg, synth.make_weights(seed=2024 + i), device=dev, max_batch=B),
                       CodebookEngine(synth.make_codebook(N_ROWS, 128, seed=7 + i), device=dev))
        return objs[i]

    if not args.no_config4 and (not args.no_extras or args.config4) and args.precision == 'f32':
        # ---- BASELINE config 4 as SURVEY 8d defines it: ONE mixed batch of 256 crops with class labels integers(0, 8): EIGHT objects at
        # every N, object o on rank o mod N (8 / N objects per GPU); every rank runs encode + scan on its buckets (the host routes by
        # class id, as the reference's per-box loop does, m3_interface/ae_pose_estimator.py:143-170) and -- N > 1 -- one RCCL all_gather
        # of the padded (index, score) pairs re-assembles the batch.  Strong scaling of a fixed 256-crop batch over 1 / 2 / 4 / 8 GPUs:
        # the N = 1 line is the single-GPU multi-object number the curve starts from.
        import numpy as np
        from augmentedautoencoder_amd.dist import ShardedPoseEngine, owner_of
        labels = np.random.default_rng(0).integers(0, N_OBJ, BATCH)
        all_crops = torch.from_numpy(synth.make_crops(BATCH, seed=4321)).to(dev)     # the same mixed batch on every rank
        mine = [o for o in range(N_OBJ) if owner_of(o, world) == rank]
        my_bucket = {o: all_crops[torch.from_numpy(np.flatnonzero(labels == o)).to(dev)].contiguous() for o in mine
                     if (labels == o).any()}                                         # host-side routing, outside the timed region
        for o in mine:
            obj(o)

        def local_infer(o, c):
            e, c_b = objs[o]
            return e.encode_nn(c_b, c, 1)[1:]

        # this rank's buckets in ONE C call (aae_encode_nn_multi): one Winograd launch per conv layer across its objects where the group fills
        # the chip (round 6; before: one six-launch chain per object, conv3 / conv4 off the Winograd form at ~32 crops per bucket)
        from augmentedautoencoder_amd.engine import MultiObjectQuery
        order = [o for o in mine if o in my_bucket]
        mq4 = MultiObjectQuery([(objs[o][0], objs[o][1], int(my_bucket[o].shape[0])) for o in order], device=dev) if len(order) >= 2 else None
        x4 = torch.cat([my_bucket[o] for o in order]).contiguous() if mq4 is not None else None     # (host-side routing: outside the timed region, like the buckets)
        starts = np.concatenate([[0], np.cumsum([int(my_bucket[o].shape[0]) for o in order])]) if mq4 is not None else None

        def local_infer_many(_buckets):
            _, idx_all, score_all = mq4(x4)
            return idx_all, score_all, order              # (one array for the rank: one pack launch)
        spe = ShardedPoseEngine(local_infer, world_size=None if use_dist else 1, rank=None if use_dist else 0, device=dev,
                                pack_pairs=pack_pairs, unpack_pairs=unpack_pairs, local_infer_many=local_infer_many if mq4 is not None else None)
        for _ in range(max(args.warmup, 3)):
            spe.infer(my_bucket, labels)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            idx4, _ = spe.infer(my_bucket, labels)
        torch.cuda.synchronize()
        t4 = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
            t4 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t4, op=dist.ReduceOp.MAX)
            t4 = float(t4.item())
        extras['config4'] = {'workload': 'one mixed batch of %d crops over %d objects (own weights + own 92232-row codebook each), object o on rank o mod %d '
                                         '(%d per GPU), routed by class id%s' % (BATCH, N_OBJ, world, len(mine),
                                                                               ', RCCL all_gather of (idx, score) pairs padded to %d rows per rank' % BATCH if use_dist else ', single GPU: no collective'),
                             'value': round(BATCH * args.steps / t4, 1), 'unit': 'crops/s', 'ms_per_batch': round(t4 / args.steps * 1e3, 4),
                             'objects': N_OBJ, 'objects_per_gpu': len(mine),
                             'bucket_sizes': np.bincount(labels, minlength=N_OBJ).tolist(), 'scaling': 'strong (global batch and object count fixed)',
                             'answers_complete': bool((idx4 >= 0).all().item()),
                             'launches_per_step_besides_encode_nn': 'one pack_pairs per rank, %sunpack_pairs (buffers owned by the cached plan)' % ('all_gather, ' if use_dist else ''),
                             'query': ('aae_encode_nn_multi: one C call for the rank\'s %d buckets, %d grouped launches (one launch per layer across the objects: conv1, the Winograd conv layers, the dense layer, + one for the incomplete four-image blocks of conv4; '
                                       'the codebook scans as one launch per row-part count + one reduce launch)' % (len(order), mq4.launches or 0)) if mq4 is not None else 'aae_encode_nn per object'}
    if not use_dist and not args.no_extras and args.precision == 'f32':
        from augmentedautoencoder_amd.engine import CapturedNearestNeighbour, DecoderEngine, StreamingNearestNeighbour
        from augmentedautoencoder_amd.weights import DecoderConfig
        import numpy as np
        cb_bytes = N_ROWS * 128 * 4
        # ---- per-detection latency: the reference calls session.run once per detected box (ae_pose_estimator.py:143-170)
        lat = {}
        for b in (1, 2, 4):
            xb = x[:b].contiguous()
            eager = time_us(lambda: enc.encode_nn(cb, xb, 1), 200)
            cap = CapturedNearestNeighbour(enc, cb, b, force_graph=True)     # (the object itself makes the eager call at B <= 4: it is faster)
            graph = time_us(lambda: cap.graph.replay(), 200)
            _, recs = enc.encode_timed(xb)
            lat['B%d' % b] = {'encode+nn_us': round(eager, 2), 'graph_replay_us': round(graph, 2), 'crops_per_s': round(b / eager * 1e6, 1),
                              'launches': len(recs) if any(l.startswith('chain:') for l, _, _ in recs) else len(recs) + 1}
            cap.close()
            del cap
        lat['note'] = ('fused aae_encode_nn: conv1, three wave-split-K convolutions, dense GEMV, stream scan -- each split reduction finished '
                       'inside its own launch; floor of this chain = 27 us of fp32 MFMA work (4.28 GFLOP at 157 TF) + 107 MB of weights/codebook.  '
                       'The same query as conv1 + ONE persistent launch (option detect_chain, grid barriers between the phases) is built, '
                       'bit-identical and measured slower: profiles/r11_small/chain_vs_six_launches_latency.jsonl')
        # cold: the same query rotating over 4 objects (4 x 107 MB of weights + codebook > the 256 MB Infinity Cache), so that every
        # call finds its weights and codebook in HBM -- a frame's detections belong to different classes (ae_pose_estimator.py:61-78)
        rot = [obj(i) for i in range(4)]
        for b in (1, 4):
            xb = x[:b].contiguous()
            k = [0]

            def cold_query():
                e, c_b = rot[k[0] % len(rot)]
                e.encode_nn(c_b, xb, 1)
                k[0] += 1
            lat['B%d_cold_us' % b] = round(time_us(cold_query, 200, warm=8), 2)
        extras['latency'] = lat
        # ---- a frame with detections of EIGHT classes (the reference keeps one AAE per class in one process and a frame's boxes spread
        # over them: m3_config_tless.cfg:10-39, ae_pose_estimator.py:61-78,143-170): 8 objects x d detections, visited in turn, so
        # 8 x 107 MB = 856 MB of weights + codebooks cannot sit in the Infinity Cache.  sequential = one aae_encode_nn per object;
        # grouped = ONE launch per layer across the objects (aae_encode_nn_multi); answers equal up to fp32 summation order (group plan).
        all8 = [obj(i) for i in range(N_OBJ)]
        xs = [torch.from_numpy(synth.make_crops(4, seed=500 + i)).to(dev) for i in range(N_OBJ)]
        from augmentedautoencoder_amd import engine as _engine
        MultiQuery = getattr(_engine, 'MultiObjectQuery', None)
        lm = {'objects': N_OBJ, 'weights_plus_codebooks_MB': round(N_OBJ * (cfg.param_bytes() + cb_bytes) / 1e6, 1),
              'note': 'sequential = one six-launch aae_encode_nn chain per object; grouped = aae_encode_nn_multi: one launch per layer across the objects (problem table in the '
                      'kernel arguments, tickets per (object, tile), one scan launch over all codebooks), the launch plan chosen for the group (option multi_group_plan; 0 = '
                      'per-object plans, bit-identical to the sequential calls); a conv layer whose blocks fill the chip over all objects of the frame runs as ONE polyphase-Winograd launch '
                      '(option multi_group_winograd: conv2 from 9 detections per frame, conv3 from 18 -- mfma_floor_us is the DIRECT form\'s floor, which such frames can undercut).  '
                      'Cold by construction: 853 MB of weights + codebooks against a 256 MB Infinity Cache'}
        for d in (1, 4):
            def frame_seq():
                for (e, c_b), xi in zip(all8, xs):
                    e.encode_nn(c_b, xi[:d], 1)
            seq = time_us(frame_seq, 40, warm=4)
            row = {'sequential_us': round(seq, 1), 'sequential_us_per_detection': round(seq / (N_OBJ * d), 2)}
            if MultiQuery is not None:
                mq = MultiQuery([(e, c_b, d) for e, c_b in all8], device=dev)
                xcat = torch.cat([xi[:d] for xi in xs]).contiguous()
                grp = time_us(lambda: mq(xcat), 40, warm=4)
                row.update({'grouped_us': round(grp, 1), 'grouped_us_per_detection': round(grp / (N_OBJ * d), 2),
                            'grouped_over_sequential': round(grp / seq, 3), 'launches_grouped': mq.launches, 'launches_sequential': 6 * N_OBJ,
                            'mfma_floor_us': round(N_OBJ * d * cfg.flops_per_crop() / (PEAK_F32_TFLOPS * 1e6), 1),
                            'hbm_floor_us': round(N_OBJ * (cfg.param_bytes() + cb_bytes) / (PEAK_HBM_GBPS * 1e3), 1)})
            lm['8x%d' % d] = row
        if MultiQuery is not None:
            # the usual frame: classes with DIFFERENT detection counts -- still six launches (the group's GEMV / scan are instantiated for the largest count)
            counts = [1, 1, 2, 4, 1, 3, 1, 2]

            def mixed_seq():
                for (e, c_b), xi, n in zip(all8, xs, counts):
                    e.encode_nn(c_b, xi[:n], 1)
            mqm = MultiQuery([(e, c_b, n) for (e, c_b), n in zip(all8, counts)], device=dev)
            xmix = torch.cat([xi[:n] for xi, n in zip(xs, counts)]).contiguous()
            seq_m = time_us(mixed_seq, 40, warm=4)
            grp_m = time_us(lambda: mqm(xmix), 40, warm=4)
            lm['8x{1,1,2,4,1,3,1,2}'] = {'detections': sum(counts), 'sequential_us': round(seq_m, 1), 'grouped_us': round(grp_m, 1), 'grouped_over_sequential': round(grp_m / seq_m, 3),
                                         'launches_grouped': mqm.launches, 'launches_sequential': 6 * N_OBJ,
                                         'mfma_floor_us': round(sum(counts) * cfg.flops_per_crop() / (PEAK_F32_TFLOPS * 1e6), 1)}
            # a bin-picking frame: four classes with six boxes each -- between the per-detection chain and a chip-filling batch.  The mid-batch group takes the layers the frame
            # fills (conv2, conv3) as one Winograd launch across the classes, the others per class (round 6)
            x6 = [torch.from_numpy(synth.make_crops(6, seed=600 + i)).to(dev) for i in range(4)]

            def six_seq():
                for (e, c_b), xi in zip(all8[:4], x6):
                    e.encode_nn(c_b, xi, 1)
            mq6 = MultiQuery([(e, c_b, 6) for e, c_b in all8[:4]], device=dev)
            xcat6 = torch.cat(x6).contiguous()
            seq_6 = time_us(six_seq, 40, warm=4)
            grp_6 = time_us(lambda: mq6(xcat6), 40, warm=4)
            lm['4x6'] = {'detections': 24, 'sequential_us': round(seq_6, 1), 'grouped_us': round(grp_6, 1), 'grouped_over_sequential': round(grp_6 / seq_6, 3),
                         'launches_grouped': mq6.launches, 'mfma_floor_us': round(24 * cfg.flops_per_crop() / (PEAK_F32_TFLOPS * 1e6), 1)}
            # the codebook stage alone over the eight codebooks (378 MB > the 256 MB Infinity Cache: every call streams from HBM):
            # ONE launch (aae_codebook_nn_multi) against eight aae_codebook_nn calls
            z8 = torch.randn(N_OBJ, 128, device=dev)
            mqs = MultiQuery([(None, c_b, 1) for _, c_b in all8], device=dev)
            one = time_us(lambda: mqs.nn(z8), 100, warm=10)

            def scans_seq():
                for k8, (_, c_b) in enumerate(all8):
                    c_b.nn(z8[k8:k8 + 1], 1, 1)
            seq8 = time_us(scans_seq, 100, warm=10)
            lm['scan_8_codebooks'] = {'grouped_us': round(one, 2), 'sequential_us': round(seq8, 2), 'bytes': N_OBJ * cb_bytes,
                                      'grouped_GBps': round(N_OBJ * cb_bytes / one / 1e3, 1), 'grouped_frac_of_hbm_peak': round(N_OBJ * cb_bytes / one / 1e3 / PEAK_HBM_GBPS, 3),
                                      'sequential_frac_of_hbm_peak': round(N_OBJ * cb_bytes / seq8 / 1e3 / PEAK_HBM_GBPS, 3), 'launches_grouped': mqs.launches}
        extras['latency_multi'] = lm
        # ---- the codebook query alone
        z = enc.encode(x)
        z1 = z[:1].contiguous()
        warm1 = time_us(lambda: cb.nn(z1, 1, 1), 300)
        warm256 = time_us(lambda: cb.nn(z, 1, 1), 100)
        copies = [cb] + [CodebookEngine(E, device=dev) for _ in range(7)]          # 8 x 47.2 MB = 378 MB > 256 MB Infinity Cache
        k = [0]

        def cold_call():
            copies[k[0] % len(copies)].nn(z1, 1, 1)
            k[0] += 1
        cold1 = time_us(cold_call, 320, warm=16)
        for c in copies[1:]:
            c.close()
        def single_call_us(fn, reps=200):
            """median over `reps` calls, each between its own pair of HIP events on an otherwise idle stream: one launch
            incl. its dispatch, without the overlap back-to-back calls get"""
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            return ts[len(ts) // 2]
        solo1 = single_call_us(lambda: cb.nn(z1, 1, 1))
        # device time per single-launch query, measured HERE: 100 launches queued from C between two HIP events (kernel +
        # dependent-launch gap; the Python loops above pay more host time per call than this kernel runs)
        cb.nn_timed(z1, 1, 1, reps=20)
        ks = sorted(cb.nn_timed(z1, 1, 1, reps=100)[2] * 1e3 for _ in range(9))
        kernel_med, kernel_min = ks[len(ks) // 2], ks[0]
        k256 = sorted(cb.nn_timed(z, 1, 1, reps=50)[2] * 1e3 for _ in range(5))[2]
        scan_traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
                scan_traffic = json.load(f).get('scan', None)
        except Exception:
            scan_traffic = None
        extras['scan'] = {
            'single_call_between_events_us': round(solo1, 2),
            'kernel_period_us': round(kernel_med, 2), 'kernel_period_us_min': round(kernel_min, 2),
            'kernel_period_GBps': round(cb_bytes / kernel_med / 1e3, 1), 'kernel_period_frac': round(cb_bytes / kernel_med / 1e3 / PEAK_HBM_GBPS, 3),
            'B256_kernel_period_us': round(k256, 2),
            'kernel_period_source': 'this run: 100 queries queued back to back from C between two HIP events on the launch stream (aae_codebook_nn_timed): '
                                    'kernel + dependent-launch gap per query, median / min of 9 such runs; rocprofv3 kernel durations: profiles/',
            'traffic': (scan_traffic or {}).get('hbm_side_bytes_warm'), 'traffic_cold': (scan_traffic or {}).get('hbm_side_bytes_cold'),
            'traffic_source': (scan_traffic or {}).get('source', 'profiles/traffic.json (rocprofv3 PMC passes of tools/prof_mix.py: not measured in this run)'),

            'codebook_bytes': cb_bytes, 'peak_GBps': PEAK_HBM_GBPS,
            'B1_whole_call_warm_us': round(warm1, 2), 'B1_warm_GBps': round(cb_bytes / warm1 / 1e3, 1), 'B1_warm_frac': round(cb_bytes / warm1 / 1e3 / PEAK_HBM_GBPS, 3),
            'B1_whole_call_cold_us': round(cold1, 2), 'B1_cold_GBps': round(cb_bytes / cold1 / 1e3, 1), 'B1_cold_frac': round(cb_bytes / cold1 / 1e3 / PEAK_HBM_GBPS, 3),
            'B256_whole_call_us': round(warm256, 2),
            'note': 'whole aae_codebook_nn call = ONE launch at B <= 4 (normalise + stream + arg-max hand-off inside the scan kernel); whole_call = '
                    'back-to-back calls on one stream (consecutive launches overlap: a call can cost less than the duration of its own kernel); '
                    'single_call_between_events = one call between its own HIP events on an idle stream (median; includes the host launch latency); '
                    'kernel_period = queries queued from C between two events (no per-call host cost); traffic = the committed rocprofv3 PMC figures.  warm: the 47 MB codebook stays in the '
                    '256 MB Infinity Cache between calls (algorithmic bytes, not HBM bytes); cold: 8 codebook copies visited in turn, so every call streams from HBM.  '
                    'B=256 is MFMA-bound (crossover B~39).  Kernel-only durations under rocprofv3: profiles/'}
        # ---- BASELINE config 5: 4x codebook in bf16, B = 256, arg-max and top-5
        N5 = 368928
        E5 = synth.make_codebook(N5, 128, seed=11, planted_duplicates=0)
        cb5 = CodebookEngine(E5, device=dev, dtype='bf16')
        z5 = torch.randn(256, 128, device=dev)
        t_arg = time_us(lambda: cb5.nn(z5, 1, 1), 300, warm=30)      # (15 ms timed: 50 calls = 2.5 ms sat inside the clock ramp after the idle gap in front)
        t_top5 = time_us(lambda: cb5.nn(z5, 5, 1), 150, warm=15)
        t_b1 = time_us(lambda: cb5.nn(z5[:1], 1, 1), 300, warm=30)
        flops5 = 2.0 * 256 * N5 * 128
        extras['config5'] = {'rows': N5, 'dtype': 'bf16 codebook, queries as 2 bf16 terms (cosine within 3.8e-6 worst case); B=1: fp32 queries', 'codebook_bytes': N5 * 128 * 2,
                             'B256_argmax_us': round(t_arg, 2), 'B256_top5_us': round(t_top5, 2), 'B1_argmax_us': round(t_b1, 2),
                             'B256_argmax_algorithmic_GBps': round(N5 * 256 / t_arg / 1e3, 1),
                             'B256_argmax_TFLOPs_nominal': round(flops5 / t_arg / 1e6, 1),
                             'B256_argmax_frac_of_bf16_mfma_peak_nominal': round(flops5 / t_arg / 1e6 / 2500.0, 3),
                             'B256_argmax_frac_of_bf16_mfma_peak_issued': round(2 * flops5 / t_arg / 1e6 / 2500.0, 3),
                             'B1_argmax_GBps': round(N5 * 256 / t_b1 / 1e3, 1), 'B1_frac_of_HBM_peak': round(N5 * 256 / t_b1 / 1e3 / PEAK_HBM_GBPS, 3),
                             'note': 'frac ..._nominal: 2*B*N*128 FLOP against 2.5 PFLOP/s dense bf16; ..._issued counts the 2 MFMAs per product of the query split.  top-5: sorted lists inside the scan, '
                                     'candidates below a bound the blocks publish to each other are dropped (same answers as the unpruned lists and the similarity-matrix path: tests + tools/soak_prune.py)'}
        cb5.close()
        del E5
        # ---- BASELINE config 3: ae_embed -- encode 92232 views (codebook.py:190-219), encoder only, inputs resident
        c3 = {}
        for bs in ((256,) if args.no_config3_b64 else (64, 256)):
            bs = min(bs, B)
            xb = x[:bs].contiguous()
            nb = -(-N_ROWS // bs)
            enc.encode(xb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nb):
                enc.encode(xb)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c3['batch%d' % bs] = {'seconds': round(dt, 3), 'crops_per_s': round(nb * bs / dt, 1), 'batches': nb}
        c3['note'] = ('92232 synthetic views, fp32, the float64 normalise on the host is not timed; batch64 = the reference batch size '
                      '(train_template.cfg:61), batch256 = the headline batch')
        extras['config3'] = c3
        # ---- PCIe-inclusive: host uint8 batches, H2D of batch i+1 overlapped with compute of batch i
        host = [synth.make_crops(B, seed=100 + i) for i in range(4)]
        sp = StreamingNearestNeighbour(enc, cb, B)
        for _ in sp.run(host[:2]):
            pass
        torch.cuda.synchronize()
        nbat = 24
        t0 = time.perf_counter()
        for _ in sp.run(host[i % 4] for i in range(nbat)):
            pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        extras['pcie_inclusive'] = {'crops_per_s': round(nbat * B / dt, 1), 'ms_per_batch': round(dt / nbat * 1e3, 3), 'batches': nbat,
                                    'note': 'host (pageable) uint8 batches of %d, indices copied back; never the headline value' % B}
        # ---- decoder (eval_plots reconstructions)
        dcfg = DecoderConfig()
        dec = DecoderEngine(dcfg, synth.make_decoder_weights_for(dcfg), device=dev, max_batch=256)
        zd = torch.randn(256, 128, device=dev) * 0.5
        t_dec = time_us(lambda: dec.decode(zd), 10, warm=2)
        extras['decoder'] = {'B256_ms': round(t_dec / 1e3, 3), 'images_per_s': round(256 / t_dec * 1e6, 1)}
        dec.close()

    if rank == 0:
        x3h_label = 'f32 in/out, 3xfp16-split MFMA with fp32 accumulate (f32x3h)'
        out = {
            'metric': METRIC,
            'value': main_res['value'],
            'unit': 'crops/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': main_res['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'f32' else x3h_label,
            'data': 'synthetic',
            'config': {'workload': 'configs[1]: single object per GPU, batch=%d uint8 128x128x3 crops, HIP 4-conv encoder -> 128-d + '
                                   'cosine-NN vs %dx128 fp32 codebook, top-1' % (B, N_ROWS),
                       'batch_per_gpu': B, 'codebook_rows': N_ROWS, 'latent': 128,
                       'parallelism': 'objects sharded 1 per GPU; all_gather of (idx, score) only' if use_dist else 'single GPU'},
            'roofline': main_res['roofline'],
            'encoder_tflops': main_res['encoder_tflops'],
            'encoder_tflops_note': 'algorithmic (direct-form) flops of the encoder / kernel time: with the polyphase-Winograd conv layers a TF-EQUIVALENT that may exceed the matrix peak',
            'kernels': main_res['kernels'],
        }
        if use_dist:
            out['rccl_world_size'] = dist.get_world_size()
            out['launched_by'] = os.environ.get('AAE_BENCH_LAUNCHER', 'external launcher (torch.distributed.run)')
            if world > 1:
                out['multi_gpu_note'] = ('N > 1: the line carries the weak-scaling headline and config4 only; cpu_baseline, latency, scan, config3, '
                                         'config5, pcie_inclusive and decoder are single-GPU measurements and are reported by the N = 1 run')
            out['rank_ms_per_step'] = {'min': min(main_res['rank_ms_per_step']), 'max': max(main_res['rank_ms_per_step']),
                                       'per_rank': main_res['rank_ms_per_step']}
        if args.enc_opt:
            out['config']['encoder_options'] = args.enc_opt
        out.update(extras)
        if direct_res is not None:
            out['direct_fp32'] = {'mode': 'encoder option winograd = 0: every conv layer as a direct implicit GEMM (25 multiplies per tap set instead of 12.25)',
                                  'value': direct_res['value'], 'unit': 'crops/s', 'ms_per_step': direct_res['ms_per_step'], 'roofline': direct_res['roofline'],
                                  'encoder_tflops': direct_res['encoder_tflops'], 'kernels': direct_res['kernels']}
        if split_res is not None:
            out['split_precision'] = {'mode': x3h_label, 'value': split_res['value'], 'unit': 'crops/s',
                                      'ms_per_step': split_res['ms_per_step'], 'roofline': split_res['roofline'],
                                      'encoder_tflops_fp32_equivalent': split_res['encoder_tflops'], 'kernels': split_res['kernels'],
                                      'batches_recomputed_in_fp32': split_res['x3h_batches_recomputed_in_fp32'],
                                      'range_check': 'per-forward device flags, read once after the timed loop (no host sync per step)',
                                      'note': 'opt-in mode, same parity tolerances (cosine 1e-5, tie-aware index equality); not the headline'}
        if not use_dist and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(weights, E, crops)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
