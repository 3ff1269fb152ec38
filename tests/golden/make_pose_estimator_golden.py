"""Golden fixtures for the caller side (N1) from the REFERENCE's own estimator code:

    /root/reference/auto_pose/m3_interface/ae_pose_estimator.py
        AePoseEstimator.extract_square_patch (:106-131, black_borders=True branch)
        AePoseEstimator.process              (:133-232: class filter, relative -> pixel boxes, invalid-box
                                              skip, per-detection auto_pose6d, 4x4 assembly, mm / m, camPose)

The module is loaded from where it lies with tensorflow / cv2 / m3vision / auto_pose.ae replaced by
stand-ins.  cv2.resize is a stand-in that RETURNS ITS INPUT (OpenCV is not installed), so what is
recorded for the crop is the square black-bordered patch *before* interpolation -- the geometry of the
cut, which is what kernels/crop_resize_u8.h must reproduce (its bilinear stage is an identity when the
output size equals the patch size).  Codebook.auto_pose6d is a stand-in that logs its arguments and
returns a deterministic (R, t) per call, so process()'s bookkeeping is recorded end to end.

Run in the build container only:  python tests/golden/make_pose_estimator_golden.py
    ->  tests/golden/pose_estimator_ref.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/auto_pose/m3_interface/ae_pose_estimator.py'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class PoseEstimate(object):
    def __init__(self, name='', trafo=None):
        self.name, self.trafo = name, trafo


class Box(object):
    def __init__(self, xmin, xmax, ymin, ymax, classes):
        self.xmin, self.xmax, self.ymin, self.ymax, self.classes = xmin, xmax, ymin, ymax, classes


RESIZE_LOG = []


def fake_resize(img, dsize, interpolation=None):
    RESIZE_LOG.append((img.copy(), tuple(dsize), interpolation))
    return img


def load_reference_estimator():
    tf = _stub('tensorflow')
    compat = _stub('tensorflow.compat')
    v1 = _stub('tensorflow.compat.v1', disable_eager_execution=lambda: None)
    tf.compat, compat.v1 = compat, v1
    _stub('cv2', INTER_NEAREST=0, INTER_LINEAR=1, resize=fake_resize)
    ap = _stub('auto_pose')
    ae = _stub('auto_pose.ae', factory=types.ModuleType('factory'), utils=types.ModuleType('utils'))
    ap.ae = ae
    _stub('m3vision')
    _stub('m3vision.interfaces')
    _stub('m3vision.interfaces.pose_estimator', PoseEstInterface=object, PoseEstimate=PoseEstimate, Roi3D=object)
    spec = importlib.util.spec_from_file_location('ref_ae_pose_estimator', REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.AePoseEstimator


class FakeCodebook(object):
    """auto_pose6d stand-in: logs (crop, box, K, topk, upright), returns the next (R, t) of its table."""

    def __init__(self, name, Rs, ts, log):
        self.name, self.Rs, self.ts, self.log, self.calls = name, Rs, ts, log, 0

    def auto_pose6d(self, sess, det_img, box_xywh, camK, topk, train_args, upright=False, depth_pred=None):
        i = self.calls
        self.calls += 1
        self.log.append((self.name, det_img.copy(), np.array(box_xywh, dtype=np.float64), int(topk), bool(upright), train_args))
        return self.Rs[i][None].copy(), self.ts[i][None].copy()


def random_rotations(rng, n):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], 1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], 1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1)], 1)


def main():
    Est = load_reference_estimator()
    rng = np.random.default_rng(31415)
    H, W = 120, 160
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    camK = np.array([[572.4, 0, 80.0], [0, 573.6, 60.0], [0, 0, 1]])
    camPose = np.eye(4)
    camPose[:3, :3] = random_rotations(rng, 1)[0]
    camPose[:3, 3] = [0.1, -0.2, 1.5]
    # pixel boxes (x, y, w, h): inside the image (the reference's slice assignment raises otherwise), odd sizes,
    # one touching the border, fractional coordinates (astype(int32) truncation)
    px = [(20.0, 10.0, 40.0, 30.0), (100.3, 50.7, 31.9, 45.2), (0.0, 0.0, 25.0, 25.0), (60.0, 70.0, 50.0, 20.0),
          (130.0, 90.0, 30.0, 30.0), (5.0, 40.0, 17.0, 63.0)]
    classes = ['obj_a', 'obj_b', 'obj_a', 'not_configured', 'obj_b', 'obj_a']
    boxes = [Box(x / W, (x + w) / W, y / H, (y + h) / H, {c: 0.8, 'other': 0.15}) for (x, y, w, h), c in zip(px, classes)]
    boxes.append(Box(-0.02, 0.2, 0.1, 0.3, {'obj_a': 0.9}))                  # negative -> 'invalid bb', skipped

    out = {'img': img, 'camK': camK, 'camPose': camPose, 'boxes_rel': np.array([[b.xmin, b.xmax, b.ymin, b.ymax] for b in boxes]),
           'box_classes': np.array(classes + ['obj_a'])}
    pads = {'obj_a': 1.2, 'obj_b': 1.0}
    for tag, mm, use_pose in (('m', False, False), ('mm', True, False), ('campose', False, True)):
        log = []
        est = object.__new__(Est)
        est.class_2_encoder = {'obj_a': 'exp_group/obj_a', 'obj_b': 'exp_group/obj_b'}
        est.pad_factors, est.patch_sizes = dict(pads), {'obj_a': (128, 128), 'obj_b': (64, 64)}
        est.all_train_args = {'obj_a': 'args_a', 'obj_b': 'args_b'}
        est.all_codebooks = {n: FakeCodebook(n, random_rotations(np.random.default_rng(7 + k), 8),
                                             np.random.default_rng(17 + k).uniform(-300, 900, (8, 3)), log)
                             for k, n in enumerate(['obj_a', 'obj_b'])}
        est.sess, est._topk, est._upright, est._camPose = 'sess', 1, tag == 'mm', use_pose
        del RESIZE_LOG[:]
        res = est.process(boxes, img, camK, camPose=camPose if use_pose else None, mm=mm)
        out['%s_names' % tag] = np.array([r.name for r in res])
        out['%s_trafos' % tag] = np.stack([r.trafo for r in res])
        out['%s_call_classes' % tag] = np.array([l[0] for l in log])
        out['%s_call_boxes' % tag] = np.stack([l[2] for l in log])
        out['%s_call_upright' % tag] = np.array([l[4] for l in log])
        out['%s_call_train_args' % tag] = np.array([l[5] for l in log])
        out['%s_returned_R' % tag] = np.stack([est.all_codebooks[l[0]].Rs[i] for l, i in zip(log, _call_index(log))])
        out['%s_returned_t' % tag] = np.stack([est.all_codebooks[l[0]].ts[i] for l, i in zip(log, _call_index(log))])
        if tag == 'm':
            for i, (l, (patch, dsize, interp)) in enumerate(zip(log, RESIZE_LOG)):
                assert np.array_equal(l[1], patch)
                out['patch_%d' % i] = patch                                  # the square patch the reference hands to cv2.resize
                out['patch_%d_dsize_interp' % i] = np.array([dsize[0], dsize[1], interp])
            out['n_patches'] = np.int64(len(log))
            out['pad_factors'] = np.array([pads[l[0]] for l in log])
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, 'pose_estimator_ref.npz'), **out)
    for k, v in out.items():
        print(k, getattr(v, 'shape', ()), getattr(v, 'dtype', type(v)))


def _call_index(log):
    seen, idx = {}, []
    for l in log:
        idx.append(seen.get(l[0], 0))
        seen[l[0]] = idx[-1] + 1
    return idx


if __name__ == '__main__':
    main()
