"""The grouped multi-object query (aae_encode_nn_multi & co., csrc/aae_multi_impl.h, kernels/multi_launch.h) on the CPU fiber
emulator: one launch per layer across objects must give, bit for bit, what one aae_encode_nn call per object gives -- in
every block order (every split reduction is finished by 'the last block to arrive'), with a workspace that carries
whatever another frame's objects left in it, and for items the grouped kernels do not cover (answered by the per-object
path inside the same call)."""
import numpy as np
import pytest

import emu_backend as eb
from augmentedautoencoder_amd import _lib
from augmentedautoencoder_amd.weights import EncoderConfig
from oracle import reference_cpu as ref
from oracle import synth


PER_OBJECT_PLANS = {'multi_group_plan': 0}     # every object on its own launch plan: the grouped call is bit-identical to aae_encode_nn


def _object(cfg, seed, N, opts=None, dtype='f32'):
    w = synth.make_weights(seed=seed, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=cfg.latent_space_size,
                           batch_norm=cfg.batch_norm, kernel_size=cfg.kernel_size)
    enc = eb.EmuEncoder(w, cfg)
    for k, v in (opts or {}).items():
        enc.set_option(k, v)
    cb = eb.EmuCodebook(synth.make_codebook(N, cfg.latent_space_size, seed=seed + 100, planted_duplicates=5), dtype=dtype)
    return enc, cb, w


def _per_object(items, x):
    zs, idxs, scores, at = [], [], [], 0
    for enc, cb, n, stride in items:
        z, i, s = eb.encode_nn(enc, cb, x[at:at + n], stride)
        zs.append(z), idxs.append(i[:, 0]), scores.append(s[:, 0])
        at += n
    return np.concatenate(zs), np.concatenate(idxs), np.concatenate(scores)


def _close(objs):
    for enc, cb, _ in objs:
        enc.close()
        cb.close()


# (detections per object, filters of the four conv layers, planner thresholds) that make plan_wavek choose each wave-tile shape the
# grouped kernels are instantiated for on a 16 x 16 input (tests/test_emu_kernels.py: the persistent launch's cases)
_SHAPE_CASES = {
    '000': (1, [32, 64, 64, 64], {}),
    '100': (2, [32, 256, 64, 64], {'wavek_tiny_max_tiles': 2, 'wavek_narrow_max_tiles': 128}),
    '010': (3, [32, 64, 256, 64], {'wavek_tiny_max_tiles': 2, 'wavek_narrow_max_tiles': 128}),
    '210': (4, [32, 384, 256, 64], {'wavek_tiny_max_tiles': 2, 'wavek_narrow_max_tiles': 4}),
}


# (every wave-tile shape sequence; all three block orders on the cheapest one -- the emulator runs one fiber per GPU thread)
@pytest.mark.parametrize('shapes,order', [('000', 0), ('000', 1), ('000', 2), ('100', 2), ('010', 1), ('210', 2)])
def test_one_launch_per_layer_across_objects_equals_the_per_object_calls(shapes, order):
    n, filters, opts = _SHAPE_CASES[shapes]
    cfg = EncoderConfig((16, 16, 3), filters, [2, 2, 2, 1], 5, 128, True)
    eb.set_block_order(order)
    try:
        objs = [_object(cfg, 300 + 7 * o, 36 * (9 + 2 * o) + 3 * o, dict(opts, **PER_OBJECT_PLANS)) for o in range(3)]       # three objects, codebooks of different sizes
        items = [(e, c, n, 1) for e, c, _ in objs]
        x = synth.make_crops(3 * n, seed=55, shape=cfg.shape)
        z0, i0, s0 = _per_object(items, x)
        z1, i1, s1, launches = eb.encode_nn_multi(items, x)
        assert launches == 6                                             # conv1, three wave-split-K layers, dense GEMV, scan -- for all three objects
        assert np.array_equal(z1, z0) and np.array_equal(i1, i0) and np.array_equal(s1, s0)
        for k, (e, c, w) in enumerate(objs):                             # ... and the answers are right, not just equal
            z64 = ref.encoder_forward_np(ref.input_to_float(x[k * n:(k + 1) * n]), w, cfg.strides, cfg.batch_norm)
            assert np.abs(z1[k * n:(k + 1) * n] - z64).max() / np.abs(z64).max() < 5e-6
            cs = c.similarity(z1[k * n:(k + 1) * n])
            assert np.array_equal(i1[k * n:(k + 1) * n], np.argmax(cs, axis=1))
        _close(objs)
    finally:
        eb.set_block_order(0)


@pytest.mark.parametrize('shapes,members,orders', [('000', 4, (0, 1, 2, 0)), ('100', 3, (0, 2)), ('210', 2, (0, 2))])
def test_group_plan_same_bits_in_every_block_order_and_right_against_the_oracle(shapes, members, orders):
    """the default: ONE launch plan per group, chosen for the group's total tile count (plan_wavek_group) -- larger wave tiles,
    other K splits than the per-object plans.  The answers then differ from aae_encode_nn's by fp32 summation order only;
    they must not depend on the order in which blocks arrive (tickets), nor on what the workspace held before."""
    n, filters, opts = _SHAPE_CASES[shapes]
    cfg = EncoderConfig((16, 16, 3), filters, [2, 2, 2, 1], 5, 128, True)
    objs = [_object(cfg, 400 + 7 * o, 36 * (9 + o) + o, dict(opts, wavek_target_blocks=8)) for o in range(members)]     # (8 "compute units": the group's tiles span several rounds)
    items = [(e, c, n, 1) for e, c, _ in objs]
    x = synth.make_crops(members * n, seed=56, shape=cfg.shape)
    z0, i0, s0 = _per_object(items, x)
    ws = eb.MultiWorkspace()
    got = []
    for order in orders:
        eb.set_block_order(order)
        try:
            got.append(eb.encode_nn_multi(items, x, ws))
        finally:
            eb.set_block_order(0)
    for z, i, s, launches in got[1:]:
        assert launches == 6
        assert np.array_equal(z, got[0][0]) and np.array_equal(i, got[0][1]) and np.array_equal(s, got[0][2])
    z1, i1, s1, _ = got[0]
    assert np.abs(z1 - z0).max() / np.abs(z0).max() < 2e-6               # the per-object plan's answers up to summation order
    for k, (e, c, w) in enumerate(objs):
        z64 = ref.encoder_forward_np(ref.input_to_float(x[k * n:(k + 1) * n]), w, cfg.strides, cfg.batch_norm)
        assert np.abs(z1[k * n:(k + 1) * n] - z64).max() / np.abs(z64).max() < 5e-6
        cs = c.similarity(z1[k * n:(k + 1) * n])
        assert np.array_equal(i1[k * n:(k + 1) * n], np.argmax(cs, axis=1))
    # objects with DIFFERENT detection counts share the group's launches too (the GEMV / scan instantiated for the largest count)
    if n > 1:
        mixed = [(e, c, max(1, n - k), 1) for k, (e, c, _) in enumerate(objs)]
        rows = sum(it[2] for it in mixed)
        xm = synth.make_crops(rows, seed=57, shape=cfg.shape)
        zm, im, sm, launches = eb.encode_nn_multi(mixed, xm, ws)
        assert launches == 6
        zp, ip, sp = _per_object(mixed, xm)
        assert np.abs(zm - zp).max() / np.abs(zp).max() < 2e-6
        at = 0
        for (e, c, k, _) in mixed:
            cs = c.similarity(zm[at:at + k])
            assert np.array_equal(im[at:at + k], np.argmax(cs, axis=1)) and np.array_equal(sm[at:at + k], cs.max(axis=1))
            at += k
    # a single-member group keeps the per-object plan: bit-identical to aae_encode_nn
    z2, i2, s2, _ = eb.encode_nn_multi(items[:1], x[:n], ws)
    assert np.array_equal(z2, z0[:n]) and np.array_equal(i2, i0[:n]) and np.array_equal(s2, s0[:n])
    _close(objs)


@pytest.mark.parametrize('order', [0, 2])
def test_mixed_frame_groups_by_detection_count_and_falls_back_per_item(order):
    """detections {1, 2, 1, 6, 4, 1, 3, 2} over five objects: the n = 1 items form one group, n = 2 another, n = 3 and n = 4 one
    each; n = 6, the bf16 codebook and the un-prepared upright query go through the per-object path inside the same call; the
    upright query on a prepared copy is grouped.  An object may appear twice (two slices of the workspace)."""
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    eb.set_block_order(order)
    try:
        objs = [_object(cfg, 500 + 11 * o, 36 * (8 + o) + o, PER_OBJECT_PLANS) for o in range(4)]
        objs.append(_object(cfg, 590, 36 * 9, PER_OBJECT_PLANS, dtype='bf16'))
        objs[1][1].prepare_upright(36)
        e = [o[0] for o in objs]
        c = [o[1] for o in objs]
        items = [(e[0], c[0], 1, 1), (e[1], c[1], 2, 36), (e[2], c[2], 1, 1), (e[3], c[3], 6, 1), (e[0], c[0], 4, 1),
                 (e[4], c[4], 1, 1), (e[2], c[2], 3, 1), (e[3], c[3], 2, 1), (e[2], c[2], 1, 36), (e[1], c[1], 1, 1)]
        rows = sum(it[2] for it in items)
        x = synth.make_crops(rows, seed=77, shape=cfg.shape)
        z0, i0, s0 = _per_object(items, x)
        ws = eb.MultiWorkspace()
        z1, i1, s1, launches = eb.encode_nn_multi(items, x, ws)
        # groups: n = 1 (items 0, 2, 9), n = 2 stride 36 / stride 1 (items 1, 7: one group, the stride is per item), n = 4, n = 3
        assert launches == 4 * 4, launches
        assert np.array_equal(z1, z0) and np.array_equal(i1, i0) and np.array_equal(s1, s0)
        assert (i1[1:3] % 36 == 0).all()                                 # the upright answers are multiples of the stride
        # the same workspace again, now with the items in another order: every slice lands on other objects' leftovers
        perm = [4, 0, 8, 1, 9, 3, 2, 7, 5, 6]
        starts = np.cumsum([0] + [it[2] for it in items])
        items2 = [items[p] for p in perm]
        x2 = np.concatenate([x[starts[p]:starts[p + 1]] for p in perm])
        z2, i2, s2, _ = eb.encode_nn_multi(items2, x2, ws)
        assert np.array_equal(z2, np.concatenate([z0[starts[p]:starts[p + 1]] for p in perm]))
        assert np.array_equal(i2, np.concatenate([i0[starts[p]:starts[p + 1]] for p in perm]))
        assert np.array_equal(s2, np.concatenate([s0[starts[p]:starts[p + 1]] for p in perm]))
        _close(objs)
    finally:
        eb.set_block_order(0)


def test_alternating_class_mixes_in_one_workspace_never_meet_a_stale_ticket():
    """frames alternate between two class mixes in ONE workspace: a slice's ticket words are then wherever the previous frame
    kept activations, partial sums or other objects' tickets -- nonce-tagged words and the preparation by conv1's extra blocks
    must make that invisible (also with the preparation switched off: the install path)."""
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128, True)
    objs = [_object(cfg, 700 + 13 * o, 36 * 7 + 5 * o, PER_OBJECT_PLANS) for o in range(4)]
    e = [o[0] for o in objs]
    c = [o[1] for o in objs]
    mix_a = [(e[0], c[0], 1, 1), (e[1], c[1], 1, 1), (e[2], c[2], 2, 1), (e[3], c[3], 1, 1)]
    mix_b = [(e[3], c[3], 2, 1), (e[0], c[0], 4, 1), (e[2], c[2], 1, 1)]
    xa = synth.make_crops(5, seed=91, shape=cfg.shape)
    xb = synth.make_crops(7, seed=92, shape=cfg.shape)
    want_a, want_b = _per_object(mix_a, xa), _per_object(mix_b, xb)
    ws = eb.MultiWorkspace()
    for order in (0, 2, 1):
        eb.set_block_order(order)
        try:
            for prep in ((1, 0) if order != 1 else (1,)):
                for enc in e:
                    enc.set_option('ticket_prep', prep)
                for frame in range(3):
                    items, x, want = (mix_a, xa, want_a) if frame % 2 == 0 else (mix_b, xb, want_b)
                    z, i, s, _ = eb.encode_nn_multi(items, x, ws)
                    assert np.array_equal(z, want[0]) and np.array_equal(i, want[1]) and np.array_equal(s, want[2]), (order, prep, frame)
        finally:
            eb.set_block_order(0)
    _close(objs)


@pytest.mark.parametrize('order', [0, 2])
def test_several_codebooks_in_one_scan_launch(order):
    """aae_codebook_nn_multi: the codebook stage alone -- the grouped items' codebooks streamed by ONE launch, each through its
    own ticket; equal to one aae_codebook_nn per item."""
    eb.set_block_order(order)
    try:
        J = 128
        cbs = [eb.EmuCodebook(synth.make_codebook(36 * (6 + 3 * k) + k, J, seed=40 + k, planted_duplicates=4)) for k in range(5)]
        cbs[3].prepare_upright(36)
        ns = [1, 3, 1, 3, 7]
        strides = [1, 1, 1, 36, 1]
        items = [(None, cb, n, st) for cb, n, st in zip(cbs, ns, strides)]
        z = np.random.default_rng(5).standard_normal((sum(ns), J)).astype(np.float32)
        want_i, want_s, at = [], [], 0
        for cb, n, st in zip(cbs, ns, strides):
            i, s = cb.nn(z[at:at + n], 1, st)
            want_i.append(i[:, 0]), want_s.append(s[:, 0])
            at += n
        idx, score, launches = eb.codebook_nn_multi(items, z)
        assert launches == 2                                             # n = 1 (two codebooks) and n = 3 (two codebooks, one of them the upright copy); n = 7: per-object path
        assert np.array_equal(idx, np.concatenate(want_i)) and np.array_equal(score, np.concatenate(want_s))
        for cb in cbs:
            cb.close()
    finally:
        eb.set_block_order(0)


def test_eight_equal_objects_one_xcd_per_object_mapping_is_only_a_permutation_of_blocks():
    """8 (16, ...) equal-sized objects: the conv launches put all blocks of an object on one XCD (ConvWaveKMultiArgs::xcd_affine) -- another
    assignment of physical blocks to (object, tile), the same work: bit-identical answers with the mapping on and off"""
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    objs = [_object(cfg, 1200 + o, 36 * 5 + o) for o in range(8)]
    items = [(e, c, 1, 1) for e, c, _ in objs]
    x = synth.make_crops(8, seed=4, shape=cfg.shape)
    got = {}
    for affine in (1, 0):
        for e, _, _ in objs:
            e.set_option('multi_xcd_affine', affine)
        eb.set_block_order(2 if affine else 0)
        try:
            got[affine] = eb.encode_nn_multi(items, x)
        finally:
            eb.set_block_order(0)
    assert got[1][3] == got[0][3] == 4
    assert all(np.array_equal(a, b) for a, b in zip(got[1][:3], got[0][:3]))
    zp, ip, sp = _per_object(items, x)
    assert np.abs(got[1][0] - zp).max() / np.abs(zp).max() < 2e-6 and np.array_equal(got[1][1], ip)
    _close(objs)


def test_more_objects_than_one_launch_holds():
    """kMultiMax = 16 objects per launch: 19 items of one shape take two launches per layer"""
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    objs = [_object(cfg, 900 + o, 36 * 5 + o, PER_OBJECT_PLANS) for o in range(3)]
    items = [(objs[k % 3][0], objs[k % 3][1], 1, 1) for k in range(19)]
    x = synth.make_crops(19, seed=3, shape=cfg.shape)
    z0, i0, s0 = _per_object(items, x)
    z1, i1, s1, launches = eb.encode_nn_multi(items, x)
    assert launches == 2 * 4
    assert np.array_equal(z1, z0) and np.array_equal(i1, i0) and np.array_equal(s1, s0)
    _close(objs)


def test_frame_in_one_call_crops_included():
    """aae_detect_nn_multi = aae_crop_resize_u8 over all boxes + aae_encode_nn_multi"""
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    objs = [_object(cfg, 950 + o, 36 * 6 + o, PER_OBJECT_PLANS) for o in range(2)]
    img = np.random.default_rng(8).integers(0, 256, (60, 80, 3), dtype=np.uint8)
    boxes = np.array([[5, 4, 30, 20, 36], [40, 10, 25, 40, 48], [-6, 30, 40, 28, 44]], dtype=np.int32)
    items = [(objs[0][0], objs[0][1], 2, 1), (objs[1][0], objs[1][1], 1, 1)]
    crops, z, idx, score = eb.detect_nn_multi(items, img, boxes)
    assert np.array_equal(crops, eb.crop_resize(img, boxes, (16, 16)))
    z0, i0, s0 = _per_object(items, crops)
    assert np.array_equal(z, z0) and np.array_equal(idx, i0) and np.array_equal(score, s0)
    _close(objs)


def test_argument_errors_are_reported_not_crashed():
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    cfg64 = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 64)
    a, b = _object(cfg, 11, 200), _object(cfg64, 12, 200)
    x = synth.make_crops(2, seed=1, shape=cfg.shape)
    with pytest.raises(ValueError, match='latent size'):
        eb.encode_nn_multi([(a[0], a[1], 1, 1), (b[0], b[1], 1, 1)], x)
    with pytest.raises(ValueError, match='pairs a'):
        eb.encode_nn_multi([(a[0], b[1], 2, 1)], x)
    L = eb.lib()
    arr = eb._multi_items([(a[0], a[1], 2, 1)])
    assert L.aae_multi_workspace_bytes(arr, 0, 0) == 0
    n = L.aae_multi_workspace_bytes(arr, 1, 0)
    buf = eb._aligned(n)
    z = np.zeros((2, 128), np.float32)
    idx = np.zeros(2, np.int64)
    sc = np.zeros(2, np.float32)
    assert L.aae_encode_nn_multi(arr, 1, x.ctypes.data, _lib.AAE_DTYPE_U8, z.ctypes.data, idx.ctypes.data, sc.ctypes.data, buf.ctypes.data, n - 256, None) == -4
    assert L.aae_encode_nn_multi(arr, 1, x.ctypes.data, 7, z.ctypes.data, idx.ctypes.data, sc.ctypes.data, buf.ctypes.data, n, None) == -1
    arr[0].n = 0
    assert L.aae_encode_nn_multi(arr, 1, x.ctypes.data, _lib.AAE_DTYPE_U8, z.ctypes.data, idx.ctypes.data, sc.ctypes.data, buf.ctypes.data, n, None) == -1
    _close([a, b])


_WINO = {'winograd_min_batch': 1, 'winograd_min_blocks': 1, 'first_group_split_max_tiles': 0}      # (every eligible layer as Winograd, alone and in a group: the emulated launches are tiny against the 256 compute units the fill rule counts with; conv1 in its whole-tile form, as at the batch sizes that group on the GPU)


@pytest.mark.parametrize('order', [0, 2])
def test_per_detection_group_runs_conv_layers_as_one_winograd_launch_where_the_group_fills_the_chip(order):
    """three objects with 1, 2 and 4 detections (a frame of a detector): with the block-count rule forced low the group's conv2 (16 x 16-pixel regions) and conv3 (four 8 x 8 images per
    block: every object one ragged block) run as ONE Winograd launch each across the objects (multi_group_winograd) between the grouped conv1 and the GEMV / scan launches.  The conv
    layers are the objects' own Winograd launches bit for bit; latents against the per-object calls within the group plan's summation order, and right against the oracle."""
    cfg = EncoderConfig((64, 64, 3), [32, 64, 64], [2, 2, 2], 5, 128, True)
    counts = [1, 2, 4]
    opts = {'winograd_min_batch': 1, 'winograd_min_blocks': 1}
    eb.set_block_order(order)
    try:
        objs = [_object(cfg, 950 + 7 * o, 36 * (8 + o) + o, opts) for o in range(3)]
        items = [(e, c, n, 1) for (e, c, _), n in zip(objs, counts)]
        x = synth.make_crops(sum(counts), seed=63, shape=cfg.shape)
        z0, i0, s0 = _per_object(items, x)                               # (every object's own call: Winograd conv layers too under these options)
        z1, i1, s1, launches = eb.encode_nn_multi(items, x)
        assert launches == 5                                             # conv1, conv2 (Winograd), conv3 (Winograd), GEMV, scan
        assert np.abs(z1 - z0).max() / np.abs(z0).max() < 2e-6 and np.array_equal(i1, i0)
        at = 0
        for (e, c, w), n in zip(objs, counts):
            z64 = ref.encoder_forward_np(ref.input_to_float(x[at:at + n]), w, cfg.strides, cfg.batch_norm)
            assert np.abs(z1[at:at + n] - z64).max() / np.abs(z64).max() < 5e-6
            assert np.array_equal(i1[at:at + n], np.argmax(c.similarity(z1[at:at + n]), axis=1))
            at += n
        for e, _, _ in objs:
            e.set_option('multi_group_winograd', 0)                      # the wave-split-K kernel for every conv layer again
        z2, i2, s2, launches = eb.encode_nn_multi(items, x)
        assert launches == 5 and np.abs(z2 - z0).max() / np.abs(z0).max() < 5e-6 and np.array_equal(i2, i0)
        assert not np.array_equal(z2, z1)                                # (another kernel computed the conv layers)
        _close(objs)
    finally:
        eb.set_block_order(0)


@pytest.mark.parametrize('order', [0, 2])          # (ascending and scrambled block order; the descending order ran green through round 6 and was dropped for CPU suite time)
def test_mid_batch_group_one_winograd_launch_per_conv_layer_across_objects(order):
    """objects with 5 or more detections each: ONE launch per layer across the objects -- conv1, the dense layer where the members' plans agree, the scans, ONE Winograd launch per conv layer across
    the objects (conv_wino_layer_multi_kernel) -- both block geometries (16 x 16-pixel regions of one image; four 8 x 8 images per block,
    ragged groups), three objects with different detection counts; bit for bit the per-object calls, and right against the oracle."""
    cfg = EncoderConfig((64, 64, 3), [32, 64, 64], [2, 2, 2], 5, 128, True)
    counts = [5, 7, 6]
    eb.set_block_order(order)
    try:
        objs = [_object(cfg, 500 + 7 * o, 36 * (9 + 2 * o) + 3 * o, _WINO) for o in range(3)]
        items = [(e, c, n, 1) for (e, c, _), n in zip(objs, counts)]
        x = synth.make_crops(sum(counts), seed=58, shape=cfg.shape)
        z0, i0, s0 = _per_object(items, x)
        z1, i1, s1, launches = eb.encode_nn_multi(items, x)
        assert launches == 5                                             # (the grouped launches: conv1 in its whole-tile form, conv2, conv3, the three scans as one launch + one reduce launch; the dense GEMV per object)
        assert np.array_equal(z1, z0) and np.array_equal(i1, i0) and np.array_equal(s1, s0)
        at = 0
        for (e, c, w), n in zip(objs, counts):
            z64 = ref.encoder_forward_np(ref.input_to_float(x[at:at + n]), w, cfg.strides, cfg.batch_norm)
            assert np.abs(z1[at:at + n] - z64).max() / np.abs(z64).max() < 5e-6
            cs = c.similarity(z1[at:at + n])
            assert np.array_equal(i1[at:at + n], np.argmax(cs, axis=1))
            at += n
        # a frame that mixes a per-detection group (n <= 4), a mid-batch group and a loner that joins neither (its own options)
        lone = _object(cfg, 600, 36 * 8, dict(_WINO, multi_mid_group=0))
        small = [_object(cfg, 610 + o, 36 * 7 + o) for o in range(2)]
        mixed = [(small[0][0], small[0][1], 2, 1)] + items[:2] + [(lone[0], lone[1], 5, 1), (small[1][0], small[1][1], 1, 1)]
        xm = synth.make_crops(sum(it[2] for it in mixed), seed=59, shape=cfg.shape)
        zp, ip, sp = _per_object(mixed, xm)
        zm, im, sm, _ = eb.encode_nn_multi(mixed, xm)
        assert np.array_equal(im, ip)
        at = 0
        for k, (e, c, n, _) in enumerate(mixed):
            same = np.array_equal(zm[at:at + n], zp[at:at + n]) and np.array_equal(sm[at:at + n], sp[at:at + n])
            assert same or k in (0, 4), k                                # (the two per-detection items share a group plan: summation order)
            assert np.abs(zm[at:at + n] - zp[at:at + n]).max() / np.abs(zp).max() < 2e-6 and np.abs(sm[at:at + n] - sp[at:at + n]).max() < 1e-6
            at += n
        _close(objs + [lone] + small)
    finally:
        eb.set_block_order(0)


@pytest.mark.parametrize('order', [2])
def test_mid_batch_group_hands_incomplete_four_image_blocks_to_one_wave_split_k_launch(order):
    """conv3 of this net packs four 8 x 8 images per block: counts {5, 7, 6} = 6 blocks, 3 of them incomplete.  On a "chip" of 4 compute units the
    incomplete ones open a second round: the objects' last 1 / 3 / 2 images go to ONE grouped wave-split-K launch (plan_mid_ragged) -- the same
    answers up to fp32 summation order on those images, bit for bit on the others; option multi_mid_ragged = 0 keeps every image in the Winograd launch."""
    cfg = EncoderConfig((64, 64, 3), [32, 64, 64], [2, 2, 2], 5, 128, True)
    counts = [5, 7, 6]
    opts = dict(_WINO, wavek_target_blocks=4, winograd_xcd_cols=0)
    eb.set_block_order(order)
    try:
        objs = [_object(cfg, 800 + 7 * o, 36 * (9 + 2 * o) + 3 * o, opts) for o in range(3)]
        items = [(e, c, n, 1) for (e, c, _), n in zip(objs, counts)]
        x = synth.make_crops(sum(counts), seed=61, shape=cfg.shape)
        z0, i0, s0 = _per_object(items, x)
        z1, i1, s1, launches = eb.encode_nn_multi(items, x)
        assert launches == 6                                             # conv1, conv2, conv3's complete blocks, conv3's last images, scans, reduce
        at = 0
        for (e, c, w), n in zip(objs, counts):
            full = n // 4 * 4
            assert np.array_equal(z1[at:at + full], z0[at:at + full])
            assert not np.array_equal(z1[at + full:at + n], z0[at + full:at + n])          # (another kernel computed them ...)
            assert np.abs(z1[at:at + n] - z0[at:at + n]).max() / np.abs(z0).max() < 2e-6   # (... to the same numbers)
            z64 = ref.encoder_forward_np(ref.input_to_float(x[at:at + n]), w, cfg.strides, cfg.batch_norm)
            assert np.abs(z1[at:at + n] - z64).max() / np.abs(z64).max() < 5e-6
            assert np.array_equal(i1[at:at + n], np.argmax(c.similarity(z1[at:at + n]), axis=1))
            at += n
        for e, _, _ in objs:
            e.set_option('multi_mid_ragged', 0)
        z2, i2, s2, launches = eb.encode_nn_multi(items, x)
        assert launches == 5 and np.array_equal(z2, z0) and np.array_equal(i2, i0) and np.array_equal(s2, s0)
        for e, _, _ in objs:
            e.set_option('multi_mid_scan', 0)
        z3, i3, s3, launches = eb.encode_nn_multi(items, x)                # (the scans per object again: the same answers from two launches fewer in the count)
        assert launches == 3 and np.array_equal(z3, z0) and np.array_equal(i3, i0) and np.array_equal(s3, s0)
        _close(objs)
    finally:
        eb.set_block_order(0)


def test_mid_batch_fill_rule_counts_the_complete_blocks_when_the_incomplete_ones_leave():
    """Two objects with 25 + 24 crops on a "chip" of 12 compute units, conv2 with four 8 x 8 images per block: 13 blocks would occupy two rounds at 54 % (below the 56 % of
    the fill rule: no group); handing the one incomplete block's image to the wave-split-K launch leaves 12 blocks = one full round: the group forms."""
    cfg = EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128)
    counts = [25, 24]
    opts = {'winograd_min_batch': 1, 'first_group_split_max_tiles': 0, 'wavek_target_blocks': 12, 'winograd_xcd_cols': 0}
    objs = [_object(cfg, 900 + o, 36 * 8 + o, opts) for o in range(2)]
    items = [(e, c, n, 1) for (e, c, _), n in zip(objs, counts)]
    x = synth.make_crops(sum(counts), seed=62, shape=cfg.shape)
    z0, i0, s0 = _per_object(items, x)
    z1, i1, s1, launches = eb.encode_nn_multi(items, x)
    assert launches == 6                                                 # conv1, conv2's twelve complete blocks, the 25th crop of object 0, the dense layer, scans, reduce
    assert np.abs(z1 - z0).max() / np.abs(z0).max() < 5e-6
    at = 0
    for (e, c, w), n in zip(objs, counts):
        z64 = ref.encoder_forward_np(ref.input_to_float(x[at:at + n]), w, cfg.strides, cfg.batch_norm)
        assert np.abs(z1[at:at + n] - z64).max() / np.abs(z64).max() < 5e-6
        assert np.array_equal(i1[at:at + n], np.argmax(c.similarity(z1[at:at + n]), axis=1))
        at += n
    for e, _, _ in objs:
        e.set_option('multi_mid_ragged', 0)
    z2, i2, s2, launches = eb.encode_nn_multi(items, x)
    assert launches == 0 and np.array_equal(z2, z0) and np.array_equal(i2, i0)          # (13 blocks in two rounds: below the rule, one object after the other)
    _close(objs)


def test_mid_batch_group_takes_the_layers_it_fills_and_leaves_the_others_to_the_objects():
    """two classes with six boxes each on a "chip" of 8 compute units: conv2 (one 16 x 16-pixel region per box: 12 blocks, two rounds at 75 %) runs as ONE Winograd launch across
    the objects, conv3 (four 8 x 8 images per block: 4 blocks, half a round) does not pass the rule and runs per object on the kernel the object's own forward takes -- here
    bit for bit the per-object calls (their conv2 passes the rule alone too)."""
    cfg = EncoderConfig((64, 64, 3), [32, 64, 64], [2, 2, 2], 5, 128, True)
    opts = {'winograd_min_batch': 1, 'first_group_split_max_tiles': 0, 'wavek_target_blocks': 8, 'winograd_xcd_cols': 0}
    objs = [_object(cfg, 970 + o, 36 * 8 + o, opts) for o in range(2)]
    items = [(e, c, 6, 1) for e, c, _ in objs]
    x = synth.make_crops(12, seed=64, shape=cfg.shape)
    z0, i0, s0 = _per_object(items, x)
    z1, i1, s1, launches = eb.encode_nn_multi(items, x)
    assert launches == 4                                                 # conv1, conv2, the scans, the reduce (conv3 and the dense GEMV per object)
    assert np.array_equal(z1, z0) and np.array_equal(i1, i0) and np.array_equal(s1, s0)
    for k, (e, c, w) in enumerate(objs):
        z64 = ref.encoder_forward_np(ref.input_to_float(x[6 * k:6 * k + 6]), w, cfg.strides, cfg.batch_norm)
        assert np.abs(z1[6 * k:6 * k + 6] - z64).max() / np.abs(z64).max() < 5e-6
    _close(objs)


def test_a_class_with_a_few_boxes_beyond_four_joins_the_per_detection_group_as_items_of_four():
    """a frame with 6 boxes of one class and 2 of another: the library answers the first class as two items (4 + 2 boxes, the same handles) inside the frame's per-detection
    group -- six launches for the frame; answers within the group plan's summation order of the per-class calls and right against the oracle.  multi_split_items = 0: the
    6-box class takes its own call."""
    cfg = EncoderConfig((16, 16, 3), [32, 64, 64, 32], [2, 2, 2, 1], 5, 128, True)
    objs = [_object(cfg, 980 + o, 36 * 9 + o) for o in range(2)]
    counts = [6, 2]
    items = [(e, c, n, 1) for (e, c, _), n in zip(objs, counts)]
    x = synth.make_crops(8, seed=65, shape=cfg.shape)
    z0, i0, s0 = _per_object(items, x)
    z1, i1, s1, launches = eb.encode_nn_multi(items, x)
    assert launches == 6                                                 # conv1, three conv layers, GEMV, scan: one group of three items
    assert np.abs(z1 - z0).max() / np.abs(z0).max() < 5e-6 and np.array_equal(i1, i0) and np.abs(s1 - s0).max() < 1e-6
    at = 0
    for (e, c, w), n in zip(objs, counts):
        z64 = ref.encoder_forward_np(ref.input_to_float(x[at:at + n]), w, cfg.strides, cfg.batch_norm)
        assert np.abs(z1[at:at + n] - z64).max() / np.abs(z64).max() < 5e-6
        assert np.array_equal(i1[at:at + n], np.argmax(c.similarity(z1[at:at + n]), axis=1))
        at += n
    for e, _, _ in objs:
        e.set_option('multi_split_items', 0)
    z2, i2, s2, launches = eb.encode_nn_multi(items, x)
    assert np.array_equal(z2[:6], z0[:6]) and np.array_equal(i2, i0)     # the 6-box class: its own call, bit for bit
    _close(objs)


def test_mid_batch_group_needs_two_members_that_fill_the_chip():
    cfg = EncoderConfig((64, 64, 3), [32, 64, 64], [2, 2, 2], 5, 128, True)
    objs = [_object(cfg, 700 + o, 36 * 8, {'winograd_min_batch': 1, 'multi_split_items': 0}) for o in range(2)]            # default fill rule: two tiny launches do not fill 256 compute units
    items = [(e, c, 5, 1) for e, c, _ in objs]
    x = synth.make_crops(10, seed=60, shape=cfg.shape)
    z0, i0, s0 = _per_object(items, x)
    z1, i1, s1, launches = eb.encode_nn_multi(items, x)
    assert launches == 0 and np.array_equal(z1, z0) and np.array_equal(i1, i0)
    _close(objs)
