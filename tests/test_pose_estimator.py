"""N1: batched multi-object estimator (augmentedautoencoder_amd/pose_estimator.py) -- crop
extraction kernel against the oracle's restatement of extract_square_patch + cv2 bilinear
resize (bit-exact, integer work), and process() against the oracle pipeline run detection by
detection the way the reference does (m3_interface/ae_pose_estimator.py:143-222).
CPU tests run the kernels on the emulator; the gpu-marked ones on the MI355X."""
import configparser
import os

import numpy as np
import pytest

from augmentedautoencoder_amd import session as S
from augmentedautoencoder_amd.codebook import Codebook, _parse_K
from augmentedautoencoder_amd.dataset import Dataset
from augmentedautoencoder_amd.encoder import Encoder
from augmentedautoencoder_amd.pose_estimator import AePoseEstimator, BoundingBox, PoseEstimate
from oracle import reference_cpu as ref
from oracle import synth

TRAIN_CFG = """
[Dataset]
H: {h}
W: {w}
C: 3
RADIUS: 700
PAD_FACTOR: 1.2
K: [1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]
[Embedding]
EMBED_BB: True
MIN_N_VIEWS: 12
NUM_CYCLO: 6
"""


def _scene(seed=0, H=240, W=320):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 3 + yy) % 256, (yy * 5 + xx // 2) % 256, (xx + yy * 2) % 256], -1).astype(np.float64)
    img += rng.normal(0, 20, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


BOXES = [[40.7, 30.2, 90.9, 60.1], [0.0, 0.0, 50.5, 120.0], [200.3, 100.9, 119.6, 139.0],   # touches right/bottom edge
         [150.0, 80.0, 20.0, 33.0],                                                          # small box: up-scaling
         [10.2, 200.8, 300.0, 39.1]]                                                         # wide box


def test_cv_resize_restatement_properties():
    img = _scene(1, 90, 70)
    assert np.array_equal(ref.cv_resize_linear_u8(img, (70, 90)), img)                     # identity
    assert np.unique(ref.cv_resize_linear_u8(np.full((200, 200, 3), 77, np.uint8), (128, 128))).tolist() == [77]
    half = ref.cv_resize_linear_u8(img[:64, :64], (32, 32))                                # exact 2x decimation = 2x2 box mean (+rounding)
    box = img[:64, :64].astype(np.int64).reshape(32, 2, 32, 2, 3).sum(axis=(1, 3))
    assert np.abs(half.astype(np.int64) - (box + 2) // 4).max() <= 1


@pytest.mark.parametrize('src_hw,dst_wh', [((90, 70), (128, 128)), ((300, 411), (128, 128)), ((37, 53), (64, 48)), ((200, 200), (17, 23))])
def test_cv_resize_restatement_against_an_independent_float_bilinear(src_hw, dst_wh):
    """cv2 is not installed, so the fixed-point INTER_LINEAR restatement cannot be pinned to OpenCV itself.  Its sampling
    convention can be pinned to an independent implementation: torch's bilinear interpolation with align_corners=False
    samples at the same half-pixel centres with border clamping (no antialiasing), in float arithmetic.  OpenCV's 11-bit
    coefficients and its two rounding stages move a grey level by at most one against that -- a wrong convention
    (align_corners, floor instead of half-pixel, an off-by-one tap) shows up as tens of grey levels on this noise image."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, src_hw + (3,), dtype=np.uint8)
    got = ref.cv_resize_linear_u8(img, dst_wh).astype(np.int64)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].to(torch.float64)
    want = F.interpolate(t, size=(dst_wh[1], dst_wh[0]), mode='bilinear', align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1.0 + 1e-9, np.abs(got - want).max()
    assert np.abs(got - np.rint(want)).mean() < 0.2          # and almost always the same grey level


@pytest.mark.parametrize('out_hw', [(128, 128), (16, 24)])
def test_crop_kernel_bit_exact_on_emulator(out_hw):
    import emu_backend as eb
    img = _scene(2)
    rows = AePoseEstimator.box_rows(BOXES, 1.2)
    got = eb.crop_resize(img, rows, out_hw)
    for i, bb in enumerate(BOXES):
        want = ref.extract_square_patch_black_borders(img, bb, 1.2, resize=(out_hw[1], out_hw[0]))
        assert np.array_equal(got[i], want), 'box %d' % i


def _tiny_estimator(engine_factory, crop_fn=None):
    """Two object classes with a 16x16 encoder; engines injected by the caller."""
    S.reset_default_graph()
    targs = configparser.ConfigParser()
    targs.read_string(TRAIN_CFG.format(h=16, w=16))
    codebooks, train_args, weights, embeds = {}, {}, {}, {}
    for k, name in enumerate(['obj_a', 'obj_b']):
        ds = Dataset('', h=16, w=16, c=3, min_n_views=12, radius=700, num_cyclo=6)
        with S.variable_scope(name):
            enc = Encoder(S.Placeholder((16, 16, 3)), 128, [32, 64], 5, [2, 2], False)
            cb = Codebook(enc, ds, True)
        w = synth.make_weights(seed=20 + k, shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2], latent=128)
        enc.load_weights(w)
        E = synth.make_codebook(ds.embedding_size, 128, seed=30 + k, planted_duplicates=2, num_cyclo=6)
        cb.assign_embedding(E)
        rng = np.random.default_rng(40 + k)
        bbs = np.stack([rng.integers(250, 350, 72), rng.integers(180, 260, 72), rng.integers(80, 200, 72), rng.integers(80, 200, 72)], 1)
        cb.assign_obj_bbs(bbs)
        engine_factory(enc, cb, w, E)
        codebooks[name], train_args[name], weights[name], embeds[name] = cb, targs, w, (E, bbs, ds)
    est = AePoseEstimator(codebooks=codebooks, train_args=train_args, upright=False)
    if crop_fn is not None:
        est.extract_square_patches = crop_fn
    return est, weights, embeds


def _detections(W=320, H=240):
    dets = []
    classes = ['obj_a', 'obj_b', 'obj_a', 'unknown', 'obj_b', 'obj_a']
    boxes = BOXES + [[-5.0, 10.0, 30.0, 30.0]]                       # last one invalid (negative) -> skipped
    for c, (x, y, w, h) in zip(classes, boxes):
        dets.append(BoundingBox(xmin=x / W, xmax=(x + w) / W, ymin=y / H, ymax=(y + h) / H, classes={c: 0.9, 'zzz': 0.1}))
    return dets


def _oracle_process(img, dets, weights, embeds, camK, mm):
    """The reference's per-detection loop, on the oracle."""
    H, W = img.shape[:2]
    out = []
    K_train = np.array(_parse_K('[1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]')).reshape(3, 3)
    for box in dets:
        clas = max(box.classes, key=box.classes.get)
        if clas not in weights:
            continue
        bb = [box.xmin * W, box.ymin * H, (box.xmax - box.xmin) * W, (box.ymax - box.ymin) * H]
        if np.any(np.array(bb) < 0):
            continue
        crop = ref.extract_square_patch_black_borders(img, bb, 1.2, resize=(16, 16))
        E, bbs, ds = embeds[clas]
        z = ref.encoder_forward_np(ref.input_to_float(crop), weights[clas], [2, 2])
        idx = ref.nearest_indices_reference(ref.cos_similarity(z, E), 1)
        R, t = ref.auto_pose6d_geometry(idx, ds.viewsphere_for_embedding, bbs.astype(np.int32), bb, camK, K_train, 700.0)
        Hm = np.eye(4)
        Hm[:3, :3] = R.squeeze()
        Hm[:3, 3] = t.squeeze() if mm else t.squeeze() / 1000.
        out.append((clas, Hm))
    return out


def test_process_batched_equals_per_detection_oracle_cpu():
    import emu_backend as eb
    from emu_engines import EmuCodebookEngine, EmuEncoderEngine
    import torch

    def inject(enc, cb, w, E):
        enc._engine = EmuEncoderEngine(enc.config, w)
        cb._engine = EmuCodebookEngine(E)

    def crop_fn(scene_img, boxes_xywh, pad_factor, resize=(128, 128)):
        img = scene_img.numpy() if torch.is_tensor(scene_img) else scene_img
        return torch.from_numpy(eb.crop_resize(img, AePoseEstimator.box_rows(boxes_xywh, pad_factor), (resize[1], resize[0])))

    est, weights, embeds = _tiny_estimator(inject, crop_fn)
    img = _scene(3)
    camK = np.array([[572.4, 0, 160.0], [0, 573.6, 120.0], [0, 0, 1]])
    dets = _detections()
    for mm in (False, True):
        got = est.process(dets, img, camK, mm=mm)
        want = _oracle_process(img, dets, weights, embeds, camK, mm)
        assert len(got) == len(want) == 4 and all(isinstance(g, PoseEstimate) for g in got)
        for g, (clas, Hm) in zip(got, want):
            assert g.name == clas and g.trafo.shape == (4, 4)
            assert np.allclose(g.trafo, Hm, atol=1e-9)
    assert est.process([], img, camK) == []
    assert est.query_process_requirements() == ['color_img', 'camK', 'bboxes']
    with pytest.raises(NotImplementedError):
        AePoseEstimator(codebooks=est.all_codebooks, train_args=est.all_train_args, topk=2)


def test_process_uploads_only_the_union_of_the_boxes_with_identical_results():
    """process() sends the union rectangle of the boxes (shifted boxes) instead of the frame when that is < 60 % of it:
    the black-border crop reads nothing outside its box, so every pose must be EXACTLY the one of the full-frame path --
    boxes in a corner, one hanging over the right / bottom image border, one of a single pixel column."""
    import emu_backend as eb
    from emu_engines import EmuCodebookEngine, EmuEncoderEngine
    import torch
    seen = []

    def inject(enc, cb, w, E):
        enc._engine = EmuEncoderEngine(enc.config, w)
        cb._engine = EmuCodebookEngine(E)

    def crop_fn(scene_img, boxes_xywh, pad_factor, resize=(128, 128)):
        img = scene_img.numpy() if torch.is_tensor(scene_img) else scene_img
        seen.append(img.shape[:2])
        return torch.from_numpy(eb.crop_resize(img, AePoseEstimator.box_rows(boxes_xywh, pad_factor), (resize[1], resize[0])))

    est, weights, embeds = _tiny_estimator(inject, crop_fn)
    img = _scene(4)
    H, W = img.shape[:2]
    camK = np.array([[572.4, 0, 160.0], [0, 573.6, 120.0], [0, 0, 1]])
    names = sorted(weights)

    def box(x, y, w, h, c):
        return BoundingBox(xmin=x / W, xmax=(x + w) / W, ymin=y / H, ymax=(y + h) / H, classes={c: 1.0})
    dets = [box(W - 70.3, H - 60.8, 50.2, 40.9, names[0]), box(W - 40.5, H - 30.2, 60.0, 50.0, names[-1]),      # second one crosses both borders
            box(W - 90.0, H - 80.0, 1.0, 33.0, names[0]), box(W - 65.7, H - 75.1, 22.6, 21.4, names[-1])]
    est.upload_union_only = False
    full = est.process(dets, img, camK, mm=True)
    shapes_full = list(seen)
    del seen[:]
    est.upload_union_only = True
    part = est.process(dets, img, camK, mm=True)
    assert all(s == (H, W) for s in shapes_full) and all(s[0] < H and s[1] < W for s in seen)     # the union path really ran
    assert len(full) == len(part) == 4
    for a, b in zip(full, part):
        assert a.name == b.name and np.array_equal(a.trafo, b.trafo)


@pytest.mark.gpu
def test_crop_kernel_bit_exact_on_gpu():
    from augmentedautoencoder_amd.engine import crop_resize
    img = _scene(5, 480, 640)
    rng = np.random.default_rng(1)
    boxes = [[float(rng.uniform(0, 500)), float(rng.uniform(0, 350)), float(rng.uniform(8, 300)), float(rng.uniform(8, 300))] for _ in range(24)]
    boxes += [[600.0, 440.0, 100.0, 100.0], [0.0, 0.0, 640.0, 480.0], [100.0, 100.0, 3.0, 2.0]]
    rows = AePoseEstimator.box_rows(boxes, 1.2)
    got = crop_resize(img, rows, (128, 128)).cpu().numpy()
    for i, bb in enumerate(boxes):
        want = ref.extract_square_patch_black_borders(img, bb, 1.2, resize=(128, 128))
        assert np.array_equal(got[i], want), 'box %d %s' % (i, bb)


@pytest.mark.gpu
def test_process_on_gpu_full_size_two_objects():
    """Default 128x128 network, two objects, 12 detections in one 480x640 image."""
    S.reset_default_graph()
    targs = configparser.ConfigParser()
    targs.read_string(TRAIN_CFG.format(h=128, w=128).replace('MIN_N_VIEWS: 12', 'MIN_N_VIEWS: 162').replace('NUM_CYCLO: 6', 'NUM_CYCLO: 36'))
    codebooks, train_args, info = {}, {}, {}
    for k, name in enumerate(['obj_a', 'obj_b']):
        ds = Dataset('', h=128, w=128, c=3, min_n_views=162, radius=700, num_cyclo=36)
        with S.variable_scope(name):
            enc = Encoder(S.Placeholder((128, 128, 3)), 128, synth.DEFAULT_NUM_FILTER, 5, [2, 2, 2, 2], False)
            cb = Codebook(enc, ds, True)
        w = synth.make_weights(seed=50 + k)
        enc.load_weights(w)
        E = synth.make_codebook(ds.embedding_size, 128, seed=60 + k, planted_duplicates=8)
        cb.assign_embedding(E)
        rng = np.random.default_rng(70 + k)
        bbs = np.stack([rng.integers(250, 350, len(E)), rng.integers(180, 260, len(E)), rng.integers(80, 200, len(E)), rng.integers(80, 200, len(E))], 1)
        cb.assign_obj_bbs(bbs)
        codebooks[name], train_args[name], info[name] = cb, targs, (w, E, bbs, ds)
    est = AePoseEstimator(codebooks=codebooks, train_args=train_args)
    # the objects of one estimator scratch in the same device memory (one encoder workspace instead of one per object)
    assert codebooks['obj_a']._encoder.engine.ws is codebooks['obj_b']._encoder.engine.ws and codebooks['obj_a'].engine.ws is codebooks['obj_b'].engine.ws
    img = _scene(9, 480, 640)
    rng = np.random.default_rng(2)
    dets, raw = [], []
    for i in range(12):
        x, y, w, h = rng.uniform(0, 400), rng.uniform(0, 300), rng.uniform(40, 200), rng.uniform(40, 170)
        c = 'obj_a' if i % 3 else 'obj_b'
        dets.append(BoundingBox(xmin=x / 640, xmax=(x + w) / 640, ymin=y / 480, ymax=(y + h) / 480, classes={c: 1.0}))
        raw.append((c, [x / 640 * 640, y / 480 * 480, (x + w) / 640 * 640 - x / 640 * 640, (y + h) / 480 * 480 - y / 480 * 480]))
    camK = np.array([[1075.65, 0, 320.0], [0, 1073.9, 240.0], [0, 0, 1]])
    got = est.process(dets, img, camK, mm=True)
    assert len(got) == 12
    # classes cut into chunks (one aae_detect_nn call + event each, geometry of chunk k under the kernels of chunk k + 1) and
    # the whole frame instead of the boxes' union rectangle: the same poses, bit for bit
    est.geometry_chunk = 2
    chunked = est.process(dets, img, camK, mm=True)
    est.geometry_chunk, est.upload_union_only = 16, False
    whole = est.process(dets, img, camK, mm=True)
    est.upload_union_only = True
    for a, b, c in zip(got, chunked, whole):
        assert a.name == b.name == c.name and np.array_equal(a.trafo, b.trafo) and np.array_equal(a.trafo, c.trafo)
    # one C call per frame (aae_detect_nn_multi, the default) against one aae_detect_nn call per class: the same poses
    assert est.multi_call
    est.multi_call = False
    per_class = est.process(dets, img, camK, mm=True)
    est.multi_call = True
    assert all(a.name == b.name and np.array_equal(a.trafo, b.trafo) for a, b in zip(got, per_class))
    # ... and with the completion events instead of watching the pinned index buffer
    est.poll_results = False
    with_events = est.process(dets, img, camK, mm=True)
    est.poll_results = True
    assert all(np.array_equal(a.trafo, b.trafo) for a, b in zip(got, with_events))
    est.geometry_chunk = 0                               # (used to hang the chunk planner: ADVICE r4)
    assert all(np.array_equal(a.trafo, b.trafo) for a, b in zip(got, est.process(dets, img, camK, mm=True)))
    est.geometry_chunk = 16
    # frames with few boxes are not copied to the device at all: the crop kernel reads the pinned staging buffer in place
    for stage in getattr(est, '_stages', {}).values():
        stage.direct_rows = 64
    in_place = est.process(dets, img, camK, mm=True)
    few = est.process(dets[:5], img, camK, mm=True)
    for stage in getattr(est, '_stages', {}).values():
        stage.direct_rows = 0
    few_copied = est.process(dets[:5], img, camK, mm=True)
    for stage in getattr(est, '_stages', {}).values():
        stage.direct_rows = 8
    assert all(np.array_equal(a.trafo, b.trafo) for a, b in zip(got, in_place)) and len(few) == 5
    assert all(a.name == b.name and np.array_equal(a.trafo, b.trafo) for a, b in zip(few, few_copied))
    K_train = np.array(_parse_K('[1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]')).reshape(3, 3)
    for g, box in zip(got, dets):
        c = max(box.classes, key=box.classes.get)
        w, E, bbs, ds = info[c]
        bb = [box.xmin * 640, box.ymin * 480, (box.xmax - box.xmin) * 640, (box.ymax - box.ymin) * 480]
        crop = ref.extract_square_patch_black_borders(img, bb, 1.2, resize=(128, 128))
        z64 = ref.encoder_forward_torch(ref.input_to_float(crop), w, [2, 2, 2, 2], False, 'float64')
        cs64 = ref.cos_similarity(z64, E)
        idx = ref.nearest_indices_reference(cs64, 1)
        srt = np.sort(cs64[0])
        R, t = ref.auto_pose6d_geometry(idx, ds.viewsphere_for_embedding, bbs.astype(np.int32), bb, camK, K_train, 700.0)
        assert g.name == c
        if srt[-1] - srt[-2] >= 2e-6:                       # away from near-ties (SURVEY 8c's gap) the whole pose must agree
            assert np.allclose(g.trafo[:3, :3], R.squeeze(), atol=1e-9) and np.allclose(g.trafo[:3, 3], t.squeeze(), atol=1e-6)


# ---- golden fixtures recorded from the reference's own estimator code (tests/golden/make_pose_estimator_golden.py:
# ae_pose_estimator.py loaded with TF / cv2 / m3vision stubbed, cv2.resize returning its input, auto_pose6d a logger) ----
PG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pose_estimator_ref.npz'))


def _golden_pixel_boxes():
    H, W = PG['img'].shape[:2]
    rel = PG['boxes_rel']
    return [[xmin * W, ymin * H, (xmax - xmin) * W, (ymax - ymin) * H] for xmin, xmax, ymin, ymax in rel]


def test_square_patch_geometry_matches_reference_extract_square_patch():
    """The black-bordered square patch the reference hands to cv2.resize (before interpolation) == the
    oracle's crop == the HIP crop kernel (emulated) asked for an output of the patch's own size, where its
    bilinear stage is an identity."""
    import emu_backend as eb
    img = PG['img']
    assert np.array_equal(PG['m_call_boxes'], np.array([b for b, c in zip(_golden_pixel_boxes(), PG['box_classes'])
                                                        if c in ('obj_a', 'obj_b') and min(b) >= 0]))
    for i in range(int(PG['n_patches'])):
        want = PG['patch_%d' % i]
        size = want.shape[0]
        bb, pad = PG['m_call_boxes'][i], float(PG['pad_factors'][i])
        assert AePoseEstimator.box_rows([bb], pad)[0, 4] == size
        assert np.array_equal(ref.extract_square_patch_black_borders(img, bb, pad, resize=(size, size)), want), i
        got = eb.crop_resize(img, AePoseEstimator.box_rows([bb], pad), (size, size))
        assert np.array_equal(got[0], want), i
        assert list(PG['patch_%d_dsize_interp' % i][:2]) == ([128, 128] if PG['m_call_classes'][i] == 'obj_a' else [64, 64])
        assert PG['patch_%d_dsize_interp' % i][2] == 1                       # cv2.INTER_LINEAR


class _ScriptedCodebook(object):
    """Stands where a Codebook stands in AePoseEstimator: answers with the (R, t) the reference run was fed."""

    class _E(object):
        class engine(object):
            import torch as _t
            device = _t.device('cpu')
    _encoder = _E()

    def __init__(self, Rs, ts):
        self.Rs, self.ts, self.n, self.seen = Rs, ts, 0, []

    def nearest_rotation(self, session, crops, top_n=1, upright=False, return_idcs=False):
        assert top_n == 1 and return_idcs
        self.seen.append((len(crops), upright))
        out = np.arange(self.n, self.n + len(crops))
        self.n += len(crops)
        return out

    def pose_from_indices(self, idcs, predicted_bb, K_test, train_args, depth_pred=None):
        i = int(idcs[0])
        return self.Rs[i][None].copy(), self.ts[i][None].copy()


def test_process_bookkeeping_matches_reference_process():
    """Class filter, relative -> pixel boxes, invalid-box skip, detection order, 4x4 assembly, mm vs m,
    camPose composition: AePoseEstimator.process against the trafos the reference's process() produced."""
    import torch
    img, camK = PG['img'], PG['camK']
    dets = [BoundingBox(xmin=r[0], xmax=r[1], ymin=r[2], ymax=r[3], classes={c: 0.8, 'other': 0.15})
            for r, c in zip(PG['boxes_rel'], PG['box_classes'])]
    targs = configparser.ConfigParser()
    targs.read_string(TRAIN_CFG.format(h=16, w=16))
    for tag, mm, use_pose in (('m', False, False), ('mm', True, False), ('campose', False, True)):
        cls = PG['%s_call_classes' % tag]
        books = {n: _ScriptedCodebook(PG['%s_returned_R' % tag][cls == n], PG['%s_returned_t' % tag][cls == n]) for n in ('obj_a', 'obj_b')}
        est = AePoseEstimator(codebooks=books, train_args={'obj_a': targs, 'obj_b': targs}, upright=(tag == 'mm'), camPose=use_pose)
        crops_seen = []

        def crop_fn(scene_img, boxes_xywh, pad_factor, resize=(128, 128)):
            crops_seen.extend(boxes_xywh)
            return torch.zeros((len(boxes_xywh), resize[1], resize[0], 3), dtype=torch.uint8)
        est.extract_square_patches = crop_fn
        got = est.process(dets, img, camK, camPose=PG['camPose'] if use_pose else None, mm=mm)
        assert [g.name for g in got] == [str(n) for n in PG['%s_names' % tag]]
        assert np.array_equal(np.stack([g.trafo for g in got]), PG['%s_trafos' % tag])
        assert sorted(map(tuple, crops_seen)) == sorted(map(tuple, PG['%s_call_boxes' % tag]))     # same boxes reach the crop stage
        assert all(up == bool(PG['%s_call_upright' % tag][0]) for b in books.values() for _, up in b.seen)


def test_camera_matrix_expression_is_parsed_not_evaluated():
    """[Dataset] K (cfg/train_template.cfg:11 writes 720/2): numbers, + - * / and lists only."""
    assert _parse_K('[1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]') == [1075.65, 0, 360.0, 0, 1073.9, 270.0, 0, 0, 1]
    assert _parse_K('(-1, +2.5*2, 3-1)') == [-1, 5.0, 2]
    for bad in ("().__class__.__base__.__subclasses__()", "__import__('os').system('true')", '[1, 2**3]', '[1] * 3', 'K', '[True]'):
        with pytest.raises((ValueError, SyntaxError)):
            _parse_K(bad)


def test_image_format_comes_from_the_test_config(tmp_path, monkeypatch):
    """ae_pose_estimator.py:41-43: color_format / color_data_type / depth_data_type of the [auto_pose] section are what
    query_image_format() hands to the caller; process() casts a non-uint8 image as the reference's uint8 canvas does."""
    assert AePoseEstimator._numpy_type('np.uint8') is np.uint8 and AePoseEstimator._numpy_type('numpy.float32') is np.float32
    with pytest.raises(ValueError):
        AePoseEstimator._numpy_type('os.system')
    cfg = tmp_path / 'm3.cfg'
    cfg.write_text('[auto_pose]\ncolor_format: rgb\ncolor_data_type: np.uint8\ndepth_data_type: np.float64\ncamPose: False\n'
                   'upright: False\ntopk: 1\nclass_2_encoder: {}\n')
    monkeypatch.setenv('AE_WORKSPACE_PATH', str(tmp_path))
    est = AePoseEstimator(str(cfg))
    assert est.query_image_format() == {'color_format': 'rgb', 'color_data_type': np.uint8, 'depth_data_type': np.float64}
    plain = AePoseEstimator(codebooks={'a': object()}, train_args={'a': _args_for_format_test()})
    assert plain.query_image_format()['color_format'] == 'bgr'


def _args_for_format_test():
    args = configparser.ConfigParser()
    args.read_string('[Dataset]\nPAD_FACTOR: 1.2\nW: 128\nH: 128\n')
    return args


def test_detection_chunks_cover_every_count_and_end_in_a_small_chunk():
    """process() sends the detections of a frame to the GPU in chunks whose float64 geometry runs under the kernels of the
    chunks behind them: every detection exactly once, chunks at most three times what follows them (+ a remainder), never
    across classes, the very last one geometry_chunk detections (or a whole small class)."""
    from augmentedautoencoder_amd.pose_estimator import AePoseEstimator

    class Shell(object):
        geometry_chunk = 16
        _chunk_plan = AePoseEstimator._chunk_plan

    sh = Shell()
    rng = np.random.default_rng(3)
    cases = [[n] for n in list(range(0, 200)) + [255, 256, 257, 1000, 4097]] + [list(rng.integers(0, 120, rng.integers(2, 6))) for _ in range(300)]
    for counts in cases:
        plan = sh._chunk_plan(counts)
        assert len(plan) == len(counts)
        flat = []
        for n, sizes in zip(counts, plan):
            assert sum(sizes) == n and all(s > 0 for s in sizes), (counts, plan)
            flat += sizes
        if not flat:
            continue
        assert flat[-1] <= 32, (counts, plan)
        behind = 0
        for s in flat[::-1]:
            assert s <= (32 if behind < 16 else 3 * behind + 16), (counts, plan)
            behind += s
    for bad in (0, -3):                                   # a non-positive chunk size means "one detection at the end", never an endless loop
        sh.geometry_chunk = bad
        assert [sum(p) for p in sh._chunk_plan([5, 0, 17])] == [5, 0, 17]
    sh.geometry_chunk = 16
    assert AePoseEstimator._chunk_sizes(sh, 64) == [48, 16] and AePoseEstimator._chunk_sizes(sh, 256) == [192, 48, 16]
    assert sh._chunk_plan([43, 21]) == [[43], [21]] and sh._chunk_plan([21, 43]) == [[21], [27, 16]] and sh._chunk_plan([300, 2]) == [[230, 54, 16], [2]]


def test_workspace_sharing_is_explicit_and_reversible():
    """engine.share_workspaces rebinds the scratch buffer of every engine but the first of a kind to ONE buffer, remembers
    each engine's own, and unshare_workspaces puts them back (the estimator must not silently keep the
    caller's engines tied together)"""
    from augmentedautoencoder_amd import engine as E

    class WS(object):
        shared = False

    class Eng(object):
        def __init__(self):
            self.ws, self.device = WS(), 'cuda:0'
    a, b, c = Eng(), Eng(), Eng()
    own = [a.ws, b.ws, c.ws]
    rebound = E.share_workspaces([a, None, b, c])
    assert rebound == [b, c] and a.ws is b.ws is c.ws is own[0] and own[0].shared
    assert E.share_workspaces([a, b]) == []                # idempotent
    E.unshare_workspaces(rebound)
    assert [a.ws, b.ws, c.ws] == own and not hasattr(b, '_own_ws')
