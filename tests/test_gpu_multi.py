"""The grouped multi-object query on the MI355X: one launch per layer across objects (aae_encode_nn_multi,
aae_codebook_nn_multi, aae_detect_nn_multi) against (a) the per-object calls -- bit for bit -- and (b) the fp64 oracle OF EACH
OBJECT (cosine within 1e-5, tie-aware index).  The reference's layout: one AAE per class in one process, a frame's boxes
spread over the classes (m3_interface/ae_pose_estimator.py:61-78,143-170; cfg_m3vision/m3_config_tless.cfg:10-39)."""
import numpy as np
import pytest

from oracle import reference_cpu as ref
from oracle import synth
import parity_report as report
from test_gpu_parity import COS_TOL, STRIDES, _check_indices

pytestmark = pytest.mark.gpu

N_OBJ = 8


@pytest.fixture(scope='module')
def eight_objects():
    import torch
    from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine
    from augmentedautoencoder_amd.weights import EncoderConfig
    weights = [synth.make_weights(seed=2024 + o) for o in range(N_OBJ)]
    books = [synth.make_codebook(92232, 128, seed=7 + o, planted_duplicates=8) for o in range(N_OBJ)]
    objs = [(EncoderEngine(EncoderConfig(), weights[o], max_batch=64), CodebookEngine(books[o])) for o in range(N_OBJ)]
    yield weights, books, objs, torch.device('cuda', 0)
    for e, c in objs:
        e.close()
        c.close()


def _group_plan(objs, on):
    """option multi_group_plan of every object: 1 (default) = one launch plan per group, 0 = every object its own plan"""
    for e, _ in objs:
        e.set_option('multi_group_plan', int(on))


def _per_object(objs, counts, x):
    import torch
    zs, idxs, scores, at = [], [], [], 0
    for (e, c), n in zip(objs, counts):
        z, i, s = e.encode_nn(c, x[at:at + n], 1)
        zs.append(z.clone()), idxs.append(i[:, 0].clone()), scores.append(s[:, 0].clone())
        at += n
    return torch.cat(zs), torch.cat(idxs), torch.cat(scores)


def test_eight_objects_one_launch_per_layer_against_per_object_calls_and_the_fp64_oracle(eight_objects):
    """8 objects x {1, 1, 2, 4, 1, 3, 1, 2} detections: four groups (n = 1, 2, 3, 4) of six launches each instead of 48 launches"""
    import torch
    from augmentedautoencoder_amd.engine import MultiObjectQuery
    weights, books, objs, dev = eight_objects
    counts = [1, 1, 2, 4, 1, 3, 1, 2]
    rows = sum(counts)
    crops_host = synth.make_crops(rows, seed=8765)
    x = torch.from_numpy(crops_host).to(dev)
    z0, i0, s0 = _per_object(objs, counts, x)
    x1_host = synth.make_crops(N_OBJ, seed=99)
    x1 = torch.from_numpy(x1_host).to(dev)
    w0, wi, wsc = _per_object(objs, [1] * N_OBJ, x1)

    def against_oracle(z, idx, score, cnts, host, tag):
        at = 0
        for o, n in enumerate(cnts):
            z64 = ref.encoder_forward_torch(ref.input_to_float(host[at:at + n]), weights[o], STRIDES, False, 'float64')
            cs64 = ref.cos_similarity(z64, books[o])
            assert np.abs(z[at:at + n].cpu().numpy() - z64).max() / np.abs(z64).max() < 2e-5, (tag, o)
            assert np.abs(score[at:at + n].cpu().numpy() - cs64.max(axis=1)).max() <= COS_TOL, (tag, o)
            _check_indices(idx[at:at + n].cpu().numpy(), cs64, where='grouped multi-object query (%s), object %d (%d detections)' % (tag, o, n))
            at += n
    try:
        # ---- every object on its own launch plan: bit-identical to the per-object calls
        _group_plan(objs, 0)
        mq = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs, counts)])
        z1, i1, s1 = mq(x)
        torch.cuda.synchronize()
        assert mq.launches == 4 * 6
        assert torch.equal(z1, z0) and torch.equal(i1, i0) and torch.equal(s1, s0)
        against_oracle(z1, i1, s1, counts, crops_host, 'per-object plans')
        mq1 = MultiObjectQuery([(e, c, 1) for e, c in objs])            # one detection per object: ONE group, six launches for the whole frame
        z2, i2, s2 = mq1(x1)
        assert mq1.launches == 6
        assert torch.equal(z2, w0) and torch.equal(i2, wi) and torch.equal(s2, wsc)
        xf = torch.from_numpy(ref.input_to_float(crops_host).astype(np.float32)).to(dev)       # float input takes the same path
        zf, idf, sf = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs, counts)])(xf)
        assert torch.equal(zf, z0) and torch.equal(idf, i0) and torch.equal(sf, s0)
    finally:
        _group_plan(objs, 1)
    # ---- the default: one launch plan per group (larger wave tiles, K split for the group's tile count): summation order differs
    mq = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs, counts)])
    z3, i3, s3 = mq(x)
    assert mq.launches == 6                                    # objects with 1 ... 4 detections share ONE group: six launches for the frame's 15 detections
    assert float((z3 - z0).abs().max() / z0.abs().max()) < 1e-5
    against_oracle(z3, i3, s3, counts, crops_host, 'group plans')
    mq1 = MultiObjectQuery([(e, c, 1) for e, c in objs])
    z4, i4, s4 = mq1(x1)
    assert mq1.launches == 6
    against_oracle(z4, i4, s4, [1] * N_OBJ, x1_host, 'group plan, 8 x 1')
    z5, i5, s5 = [t.clone() for t in mq1(x1)]
    assert torch.equal(z5, z4) and torch.equal(i5, i4) and torch.equal(s5, s4)               # deterministic


def test_fifty_frames_alternating_two_class_mixes_in_one_workspace(eight_objects):
    """the same workspace serves frames whose class mixes differ: every object's slice (activations, partial sums, ticket
    words) lands where ANOTHER object's data of the previous frame lies.  Stale ticket words or partials would show as a
    differing answer; 50 frames, compared with the per-object answers bit for bit."""
    import torch
    from augmentedautoencoder_amd.engine import MultiObjectQuery
    _, _, objs, dev = eight_objects
    mix_a = [(0, 1), (1, 1), (2, 2), (3, 1), (5, 4), (6, 1)]
    mix_b = [(7, 2), (4, 1), (2, 1), (0, 3), (1, 1), (3, 1), (6, 2), (5, 1)]
    qa = MultiObjectQuery([(objs[o][0], objs[o][1], n) for o, n in mix_a])
    qb = MultiObjectQuery([(objs[o][0], objs[o][1], n) for o, n in mix_b])
    qb.ws = qa.ws                                              # ONE scratch buffer for both layouts
    xa = torch.from_numpy(synth.make_crops(qa.rows, seed=31)).to(dev)
    xb = torch.from_numpy(synth.make_crops(qb.rows, seed=32)).to(dev)
    want_a = _per_object([objs[o] for o, _ in mix_a], [n for _, n in mix_a], xa)
    want_b = _per_object([objs[o] for o, _ in mix_b], [n for _, n in mix_b], xb)
    qb(xb)                                                     # (sizes the shared buffer for the larger layout first)
    bad = {0: 0, 1: 0}
    try:
        for plan in (0, 1):
            _group_plan(objs, plan)
            if plan == 1:                                      # group plans: the reference answers are the layout's own first frame (summation order differs from the per-object calls)
                want_a = [t.clone() for t in qa(xa)]
                want_b = [t.clone() for t in qb(xb)]
            for frame in range(50):
                q, x, want = (qa, xa, want_a) if frame % 2 == 0 else (qb, xb, want_b)
                z, i, s = q(x)
                bad[plan] += int(not (torch.equal(z, want[0]) and torch.equal(i, want[1]) and torch.equal(s, want[2])))
    finally:
        _group_plan(objs, 1)
    torch.cuda.synchronize()
    report.record('multi', report.current_test(), frames=100, frames_differing_with_per_object_plans=bad[0], frames_differing_with_group_plans=bad[1])
    assert bad == {0: 0, 1: 0}


def test_codebook_stage_alone_streams_eight_codebooks_in_one_launch(eight_objects):
    import torch
    from augmentedautoencoder_amd.engine import MultiObjectQuery
    _, books, objs, dev = eight_objects
    counts = [1, 4, 1, 1, 4, 1, 1, 1]
    rows = sum(counts)
    rng = np.random.default_rng(12)
    z = torch.from_numpy(rng.standard_normal((rows, 128)).astype(np.float32)).to(dev)
    mq = MultiObjectQuery([(None, c, n) for (_, c), n in zip(objs, counts)])
    idx, score = mq.nn(z)
    assert mq.launches == 2                                    # n = 1: six codebooks in one launch; n = 4: two
    at = 0
    for (_, c), n, E in zip(objs, counts, books):
        wi, ws = c.nn(z[at:at + n], 1, 1)
        assert torch.equal(idx[at:at + n], wi[:, 0]) and torch.equal(score[at:at + n], ws[:, 0])
        cs64 = ref.cos_similarity(z[at:at + n].cpu().numpy().astype(np.float64), E)
        _check_indices(idx[at:at + n].cpu().numpy(), cs64, where='grouped scan, %d queries' % n)
        at += n


def test_frame_in_one_call_with_upright_and_large_classes(eight_objects):
    """aae_detect_nn_multi through engine.detect_nn_multi: crops + every class in one C call; a class with 9 detections and a
    class in split precision take the per-object path inside the call, an upright class is grouped on its compacted copy"""
    import torch
    from augmentedautoencoder_amd.engine import _Workspace, crop_resize, detect_nn_multi
    _, _, objs, dev = eight_objects
    rng = np.random.default_rng(3)
    img = torch.from_numpy(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)).to(dev)
    counts = [2, 9, 1, 3]
    strides = [1, 1, 36, 1]
    total = sum(counts)
    boxes = np.stack([rng.integers(0, 400, total), rng.integers(0, 300, total), rng.integers(40, 200, total), rng.integers(40, 160, total)], axis=1)
    rows_np = np.concatenate([boxes, (np.maximum(boxes[:, 2], boxes[:, 3]) * 1.2).astype(np.int64)[:, None]], axis=1).astype(np.int32)
    rows = torch.from_numpy(rows_np).to(dev)
    items = [(objs[o][0], objs[o][1], n, st) for o, (n, st) in enumerate(zip(counts, strides))]
    crops = torch.empty((total, 128, 128, 3), dtype=torch.uint8, device=dev)
    z = torch.empty((total, 128), dtype=torch.float32, device=dev)
    idx = torch.empty((total,), dtype=torch.int64).pin_memory()          # answers straight into pinned host memory
    score = torch.empty((total,), dtype=torch.float32, device=dev)
    ws = _Workspace(dev)
    objs[3][0].set_option('precision', 2)                                # (B = 3 stays exact fp32 under 'where it is faster'; the item stays grouped)
    try:
        launches = detect_nn_multi(items, img, rows, crops, z, idx, score, ws)
        torch.cuda.synchronize()
        assert launches == 6                                             # n = 2, n = 1 (upright), n = 3 in ONE group; the 9-detection class: per-object path
        assert torch.equal(crops, crop_resize(img, rows_np, (128, 128)))
        z_grp, idx_grp = z.clone(), idx.clone()
        _group_plan(objs, 0)                                             # every object its own plan: one group per detection count, bit-identical answers
        launches = detect_nn_multi(items, img, rows, crops, z, idx, score, ws)
        torch.cuda.synchronize()
        assert launches == 3 * 6
    finally:
        objs[3][0].set_option('precision', 0)
        _group_plan(objs, 1)
    at = 0
    for (e, c, n, st) in items:
        wz, wi, wsc = e.encode_nn(c, crops[at:at + n], st)
        assert torch.equal(z[at:at + n], wz) and torch.equal(idx[at:at + n], wi[:, 0].cpu()) and torch.equal(score[at:at + n], wsc[:, 0])
        assert torch.equal(idx_grp[at:at + n], wi[:, 0].cpu()) and float((z_grp[at:at + n] - wz).abs().max() / wz.abs().max()) < 1e-5
        if st > 1:
            assert (idx[at:at + n] % st == 0).all()
        at += n


def test_objects_that_share_one_workspace_alternate_on_the_ticketed_path(eight_objects):
    """engine.share_workspaces (what AePoseEstimator does by default): eight objects scratch in ONE encoder workspace and ONE
    codebook workspace.  Per-detection batches (B = 1 ... 4: every split reduction finished through ticket words at the front of
    that workspace) of DIFFERENT objects follow each other, so each call finds the previous object's tickets, partial sums and
    activations where its own will go; every answer must equal the one the object gives on its own workspace, bit for bit.
    A second stream is refused while the shared buffer still has work queued."""
    import torch
    from augmentedautoencoder_amd import engine as E
    _, _, objs, dev = eight_objects
    rng = np.random.default_rng(77)
    pool = torch.from_numpy(synth.make_crops(32, seed=606)).to(dev)
    plan = [(int(rng.integers(0, N_OBJ)), int(rng.integers(1, 5)), int(rng.integers(0, 28))) for _ in range(120)]
    want = []
    for o, b, at in plan:                                      # every object on its own buffers
        z, i, s = objs[o][0].encode_nn(objs[o][1], pool[at:at + b], 1)
        want.append((z.clone(), i.clone(), s.clone()))
    rebound = E.share_workspaces([e for e, _ in objs]) + E.share_workspaces([c for _, c in objs])
    try:
        assert len(rebound) == 2 * (N_OBJ - 1) and objs[0][0].ws is objs[5][0].ws and objs[0][1].ws is objs[5][1].ws
        bad = 0
        for (o, b, at), (wz, wi, wsc) in zip(plan, want):
            z, i, s = objs[o][0].encode_nn(objs[o][1], pool[at:at + b], 1)
            bad += int(not (torch.equal(z, wz) and torch.equal(i, wi) and torch.equal(s, wsc)))
        assert bad == 0
        report.record('multi', report.current_test(), calls=len(plan), calls_differing_on_the_shared_workspace=bad)
        side = torch.cuda.Stream()
        big_a = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        big_b = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)
        for _ in range(40):
            big_a.copy_(big_b)                                 # keeps the first stream busy for milliseconds ...
        with torch.cuda.stream(side):
            with pytest.raises(RuntimeError, match='shared by several engines'):
                objs[1][0].encode_nn(objs[1][1], pool[:1], 1)  # ... so the shared buffer refuses the second one
        torch.cuda.synchronize()
        with torch.cuda.stream(side):                          # drained: the buffer moves to the new stream
            objs[1][0].encode_nn(objs[1][1], pool[:1], 1)
        torch.cuda.synchronize()
    finally:
        E.unshare_workspaces(rebound)
    assert objs[0][0].ws is not objs[5][0].ws


def test_mid_batch_group_eight_buckets_of_a_256_crop_batch(eight_objects):
    """SURVEY section 8d config 4 on one GPU: 256 crops over 8 objects ({34, 26, 27, 32, 31, 32, 33, 41}): ONE launch per conv layer
    across the objects (conv1 in its whole-tile form, conv2 ... conv4 as Winograd: conv4's four-image blocks fill the chip where one bucket alone fills a quarter; the dense layer), the scans in shared launches.
    Against (a) each object's own call with every eligible layer forced to Winograd -- bit for bit except the images conv4 hands to the wave-split-K kernel --, (b) each object's default call
    (conv3 / conv4 on the direct kernels at these sizes: fp32 rounding of the two forms), (c) each object's fp64 oracle."""
    import torch
    from augmentedautoencoder_amd.engine import MultiObjectQuery
    weights, books, objs, dev = eight_objects
    counts = [34, 26, 27, 32, 31, 32, 33, 41]
    rows = sum(counts)
    crops_host = synth.make_crops(rows, seed=4321)
    x = torch.from_numpy(crops_host).to(dev)
    zd, idd, sd = _per_object(objs, counts, x)                           # the default per-object calls
    mq = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs, counts)])
    z1, i1, s1 = mq(x)
    torch.cuda.synchronize()
    # conv1, conv2, conv3, conv4, dense: one launch each for all eight objects + conv4's incomplete four-image blocks (the last n mod 4 images of six objects: 67 blocks x 8
    # column blocks would open a third round of blocks) as one grouped wave-split-K launch; the eight codebook scans as one launch per row-part count (buckets of up to
    # 32 crops: four row parts per block, larger ones two) + one arg-max reduce launch
    assert mq.launches == 9
    z1, i1, s1 = z1.clone(), i1.clone(), s1.clone()
    for e, _ in objs:
        e.set_option('winograd_min_blocks', 1)
    try:
        zf, idf, sf = _per_object(objs, counts, x)                       # every layer as the object's OWN Winograd launch
    finally:
        for e, _ in objs:
            e.set_option('winograd_min_blocks', 0)
    scale = float(zd.abs().max())
    at = 0
    for n in counts:                                                     # bit for bit where the same kernel computed, fp32 rounding of the two forms on the handed-over images
        full = n // 4 * 4
        assert torch.equal(z1[at:at + full], zf[at:at + full]) and torch.equal(i1[at:at + full], idf[at:at + full]) and torch.equal(s1[at:at + full], sf[at:at + full])
        if n > full:
            assert float((z1[at + full:at + n] - zf[at + full:at + n]).abs().max()) / scale < 5e-6
        at += n
    for e, _ in objs:
        e.set_option('multi_mid_ragged', 0)
    try:
        mqr = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs, counts)])
        zr, ir, sr = mqr(x)
        torch.cuda.synchronize()
        assert mqr.launches == 8 and torch.equal(zr, zf) and torch.equal(ir, idf) and torch.equal(sr, sf)       # every image in the Winograd launches: the objects' own bits
    finally:
        for e, _ in objs:
            e.set_option('multi_mid_ragged', 1)
    assert float((z1 - zd).abs().max()) / scale < 5e-6
    at = 0
    for o, n in enumerate(counts):
        sub = slice(at, at + min(n, 6))                                  # (six crops per object against the fp64 oracle: seconds, not minutes)
        z64 = ref.encoder_forward_torch(ref.input_to_float(crops_host[sub]), weights[o], STRIDES, False, 'float64')
        cs64 = ref.cos_similarity(z64, books[o])
        assert np.abs(z1[sub].cpu().numpy() - z64).max() / np.abs(z64).max() < 2e-5, o
        assert np.abs(s1[sub].cpu().numpy() - cs64.max(axis=1)).max() <= COS_TOL, o
        _check_indices(i1[sub].cpu().numpy(), cs64, where='mid-batch group, object %d (%d crops)' % (o, n))
        at += n
    # option multi_mid_group = 0: the buckets one after the other, bit-identical to the default per-object calls
    for e, _ in objs:
        e.set_option('multi_mid_group', 0)
    try:
        mq0 = MultiObjectQuery([(e, c, n) for (e, c), n in zip(objs, counts)])
        z0, i0, s0 = mq0(x)
        torch.cuda.synchronize()
        assert mq0.launches == 0 and torch.equal(z0, zd) and torch.equal(i0, idd) and torch.equal(s0, sd)
    finally:
        for e, _ in objs:
            e.set_option('multi_mid_group', 1)


def test_winograd_weights_are_built_only_for_objects_of_a_group_that_forms():
    """ADVICE r5 / VERDICT r5 weak 8: the Winograd-domain copies of conv2 ... conv4 cost 83.5 MB per object.  aae_multi_workspace_bytes builds them for the objects of a
    group with a layer in the Winograd form -- two classes with four boxes each (an estimator frame: conv2 = 128 blocks, half a round) never get there and must not pay."""
    import torch
    from augmentedautoencoder_amd.engine import CodebookEngine, EncoderEngine, MultiObjectQuery
    from augmentedautoencoder_amd.weights import EncoderConfig
    dev = torch.device('cuda', 0)
    book = synth.make_codebook(36 * 64, 128, seed=3)
    objs = [(EncoderEngine(EncoderConfig(), synth.make_weights(seed=900 + o), max_batch=64), CodebookEngine(book)) for o in range(2)]
    x = torch.from_numpy(synth.make_crops(64, seed=5)).to(dev)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    mq = MultiObjectQuery([(e, c, 4) for e, c in objs])
    mq(x[:8])
    torch.cuda.synchronize()
    assert mq.launches == 6                                               # the per-detection group on the wave-split-K kernel
    small = free0 - torch.cuda.mem_get_info()[0]
    assert small < 120 << 20, small                                       # (workspace slices only; 2 x 83.5 MB of weights would show)
    mq2 = MultiObjectQuery([(e, c, 6) for e, c in objs])                  # 2 x 6 boxes: conv2 = 192 blocks pass the rule -> a mid-batch group with conv2 as one Winograd launch
    z2, i2, s2 = mq2(x[:12])
    torch.cuda.synchronize()
    assert mq2.launches >= 2 and free0 - torch.cuda.mem_get_info()[0] > 140 << 20      # ... and now the weights are there
    at = 0
    for e, c in objs:
        wz, wi, wsc = e.encode_nn(c, x[at:at + 6], 1)
        assert float((z2[at:at + 6] - wz).abs().max() / wz.abs().max()) < 5e-6 and torch.equal(i2[at:at + 6].cpu(), wi[:, 0].cpu())
        at += 6
    for e, c in objs:
        e.close()
        c.close()
