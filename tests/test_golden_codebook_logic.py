"""Golden fixtures recorded from the REFERENCE's own host code (tests/golden/make_codebook_logic_golden.py
ran auto_pose/ae/{codebook,dataset,utils}.py with TensorFlow / cv2 stubbed out and `session.run` returning a
similarity matrix we provide): input normalisation, arg-max / upright / top-n index selection, squeeze
behaviour, row -> rotation table, auto_pose6d geometry, batch iteration, workspace paths.

Checked here: (1) the oracle restatements reproduce the reference outputs exactly -- this is what pins the
oracle the GPU parity tests compare against; (2) the product's host logic (Codebook / Dataset / utils of
augmentedautoencoder_amd) reproduces them with the scan replaced by a double that answers from the recorded
similarity matrix."""
import configparser
import hashlib
import os

import numpy as np
import pytest
import torch

from augmentedautoencoder_amd import session as S, utils as u, viewsphere as vs
from augmentedautoencoder_amd.codebook import Codebook, _parse_K
from augmentedautoencoder_amd.dataset import Dataset
from augmentedautoencoder_amd.encoder import Encoder
from oracle import reference_cpu as ref

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'codebook_logic_ref.npz'))


def test_batch_iteration_and_workspace_paths_match_reference_utils():
    for key in [k for k in G.files if k.startswith('batches_')]:
        _, N, bs = key.split('_')
        want = G[key]
        assert np.array_equal(np.array(list(u.batch_iteration_indices(int(N), int(bs)))), want)
        assert np.array_equal(np.array(list(ref.batch_iteration_indices(int(N), int(bs)))), want)
    got = [u.get_dataset_path('/ws'), u.get_checkpoint_dir('/ws/experiments/g/e'), u.get_log_dir('/ws', 'e', 'g'),
           u.get_log_dir('/ws', 'e'), u.get_train_config_exp_file_path('/ws/experiments/g/e', 'e'),
           u.get_checkpoint_basefilename('/ws/experiments/g/e')]
    assert got == [str(p) for p in G['paths']]


def test_row_to_rotation_table_is_bit_exact_against_reference_dataset():
    small = vs.viewsphere_for_embedding(42, 700.0, 36)
    assert small.dtype == np.float64 and np.array_equal(small, G['viewsphere_42x36']) and len(small) == int(G['embedding_size_42x36'])
    full = vs.viewsphere_for_embedding(2562, 700.0, 36)
    assert len(full) == int(G['embedding_size_full']) == 92232
    assert np.array_equal(full[G['viewsphere_full_rows']], G['viewsphere_full_samples'])
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(G['viewsphere_full_sha256'])   # all 92232 x 3 x 3 doubles
    # linspace(0, 2pi, 36) includes both ends: rows 36k and 36k+35 are the same rotation up to sin(2pi) = -2.4e-16
    assert np.abs(full[36 * 5] - full[36 * 5 + 35]).max() < 1e-15


def test_uint8_normalisation_fed_to_the_graph_equals_the_lookup_table():
    fed = G['fed_after_u8']                                       # reference: x/255. in float64; the placeholder is float32
    assert fed.dtype == np.float64
    lut = ref.u8_lut_f32()
    assert np.array_equal(fed.astype(np.float32), lut[G['crops_u8']])
    assert np.array_equal(ref.input_to_float(G['crops_u8']).astype(np.float32), fed.astype(np.float32))
    assert bool(G['fed_float_is_unchanged']) and list(G['fed_single_shape']) == [1, 8, 8, 3]


def test_oracle_index_selection_matches_reference_numpy_code():
    cs = G['cs']
    assert np.array_equal(ref.nearest_indices_reference(cs, 1), G['idcs_top1'])
    assert G['idcs_top1'][1] == 400                               # the planted tie: first index
    assert np.array_equal(ref.nearest_indices_reference(cs, 1, upright=True, num_cyclo=36), G['idcs_upright'])
    assert G['idcs_upright'][3] % 36 == 0 and G['idcs_top1'][3] == 36 * 7 + 3
    for k in (2, 4, 8):
        want = G['idcs_top%d' % k]
        assert np.array_equal(ref.nearest_indices_reference(cs[:1], k), want)
        assert np.array_equal(ref.topk_canonical(cs[:1], k)[0], want)    # canonical order == reference order when scores are distinct


def _train_args():
    a = configparser.ConfigParser()
    a.read_string(str(G['train_cfg']))
    return a


def test_oracle_auto_pose6d_geometry_matches_reference():
    args = _train_args()
    K_train = np.array(_parse_K(args.get('Dataset', 'K'))).reshape(3, 3)
    Rs_all = G['viewsphere_42x36']
    for ci in range(4):
        top_n, depth, upright = G['pose%d_topn_depth_upright' % ci]
        row = int(G['pose%d_crop_row' % ci])
        idcs = ref.nearest_indices_reference(G['cs'][row:row + 1], int(top_n), upright=bool(upright), num_cyclo=36)
        R, t = ref.auto_pose6d_geometry(idcs, Rs_all, G['bbs'], G['pose%d_bb' % ci], G['K_test'], K_train, 700.0,
                                        depth_pred=None if depth < 0 else float(depth))
        assert np.array_equal(R, G['pose%d_R' % ci]) and np.array_equal(t, G['pose%d_t' % ci])


class _RowEncoder(object):
    """encode(x) -> the row of the recorded crop batch each input is (the double for the encoder)."""
    device = torch.device('cpu')

    def encode(self, x):
        x = np.asarray(x)
        if x.dtype != np.uint8:
            x = np.rint(x * 255.).astype(np.uint8)
        rows = [int(np.flatnonzero((G['crops_u8'] == c).all(axis=(1, 2, 3)))[0]) for c in x]
        return torch.tensor(rows, dtype=torch.float32)[:, None]

    def encode_nn(self, codebook_engine, x, col_stride=1):
        """the fused per-detection call of the product engine: encode + top-1 scan"""
        z = self.encode(x)
        idx, score = codebook_engine.nn(z, 1, col_stride)
        return z, idx, score

    def encode_checked(self, x):
        return self.encode(x)

    def settle(self):
        return 0


class _RecordedScan(object):
    """nn(z, topk, stride): the codebook scan answered from the recorded similarity matrix."""
    device = torch.device('cpu')

    def nn(self, z, topk=1, col_stride=1):
        cs = G['cs'][z[:, 0].long().numpy()]
        if topk == 1:
            idx = ref.nearest_indices_reference(cs, 1, upright=col_stride > 1, num_cyclo=max(col_stride, 1))[:, None]
        else:
            idx = ref.topk_canonical(cs, topk)
        return torch.from_numpy(idx.astype(np.int64)), torch.from_numpy(np.take_along_axis(cs, idx, axis=1))


@pytest.fixture()
def recorded_codebook():
    S.reset_default_graph()
    ds = Dataset('', h='8', w='8', c='3', min_n_views='42', radius='700', num_cyclo='36')
    enc = Encoder(S.Placeholder(ds.shape), 128, [32, 64], 5, [2, 2], False)
    cb = Codebook(enc, ds, True)
    enc._engine = _RowEncoder()
    cb._engine = _RecordedScan()
    cb.assign_obj_bbs(G['bbs'])
    yield cb
    S.reset_default_graph()


def test_product_codebook_api_reproduces_reference_outputs(recorded_codebook):
    cb, crops = recorded_codebook, G['crops_u8']
    assert np.array_equal(cb.nearest_rotation(None, crops, return_idcs=True), G['idcs_top1'])
    assert np.array_equal(cb.nearest_rotation(None, crops, upright=True, return_idcs=True), G['idcs_upright'])
    Rb = cb.nearest_rotation(None, crops)
    assert Rb.dtype == np.float64 and np.array_equal(Rb, G['R_batch'])
    R1 = cb.nearest_rotation(None, crops[0])                      # HWC in -> [3,3] out (squeeze)
    assert R1.shape == (3, 3) and np.array_equal(R1, G['R_single'])
    assert np.array_equal(cb.nearest_rotation(None, crops[0].astype(np.float32) / 255.), G['R_single'])
    for k in (2, 4, 8):
        assert np.array_equal(cb.nearest_rotation(None, crops[0], top_n=k, return_idcs=True), G['idcs_top%d' % k])
        assert np.array_equal(cb.nearest_rotation(None, crops[0], top_n=k), G['R_top%d' % k])
    assert np.array_equal(cb.nearest_rotation_batch(None, crops), G['R_batch_fn'])
    args = _train_args()
    for ci in range(4):
        top_n, depth, upright = G['pose%d_topn_depth_upright' % ci]
        cb.embed_obj_bbs_values = None
        R, t = cb.auto_pose6d(None, crops[int(G['pose%d_crop_row' % ci])], G['pose%d_bb' % ci], G['K_test'], int(top_n), args,
                              depth_pred=None if depth < 0 else float(depth), upright=bool(upright))
        assert R.shape == G['pose%d_R' % ci].shape and np.array_equal(R, G['pose%d_R' % ci])
        assert np.array_equal(t, G['pose%d_t' % ci])
        if int(top_n) == 1:
            # the batched form the estimator uses for all detections of an object (array arithmetic where NumPy gives an
            # array element the bits of a scalar, per-detection calls elsewhere): the reference's recorded numbers, bit for bit
            idx = cb.nearest_rotation(None, crops[int(G['pose%d_crop_row' % ci])], upright=bool(upright), return_idcs=True)
            Rb, tb = cb.poses_from_indices(np.atleast_1d(idx), [G['pose%d_bb' % ci]], G['K_test'], args,
                                           depth_preds=None if depth < 0 else [float(depth)])
            assert np.array_equal(Rb.squeeze(), G['pose%d_R' % ci].squeeze()) and np.array_equal(tb.squeeze(), G['pose%d_t' % ci].squeeze())


def test_batched_pose_geometry_equals_the_per_detection_calls(recorded_codebook):
    """Codebook.poses_from_indices against pose_from_indices (itself pinned to the reference above) on a few hundred random
    boxes / rows, with and without a depth prediction; AePoseEstimator.box_rows against the reference's per-box expression."""
    from augmentedautoencoder_amd.codebook import _batch_geometry_matches_scalar
    from augmentedautoencoder_amd.pose_estimator import AePoseEstimator
    cb, args = recorded_codebook, _train_args()
    cb.embed_obj_bbs_values = None
    assert _batch_geometry_matches_scalar()                   # (else the batched form is not in use on this NumPy build)
    rng = np.random.default_rng(17)
    N = cb._dataset.embedding_size
    n = 300
    idcs = rng.integers(0, N, n)
    bbs = [[float(rng.uniform(0, 600)), float(rng.uniform(0, 400)), float(rng.uniform(5, 300)), float(rng.uniform(5, 300))] for _ in range(n)]
    for depths in (None, [float(d) for d in rng.uniform(200, 2000, n)]):
        Rb, tb = cb.poses_from_indices(idcs, bbs, G['K_test'], args, depth_preds=depths)
        for i in range(n):
            R, t = cb.pose_from_indices([idcs[i]], bbs[i], G['K_test'], args, depth_pred=None if depths is None else depths[i])
            assert np.array_equal(Rb[i], R[0]) and np.array_equal(tb[i], t[0]), i
        # the row-independent half computed ahead of time (the estimator does it while the GPU is busy): the same bits, in the
        # array form and in the per-detection form (fewer than four detections)
        Rp, tp = cb.poses_from_indices(idcs, bbs, G['K_test'], args, depth_preds=depths, prepared=cb.poses_prepare(bbs, G['K_test'], args, depths))
        assert np.array_equal(Rp, Rb) and np.array_equal(tp, tb)
        d3 = None if depths is None else depths[:3]
        R3, t3 = cb.poses_from_indices(idcs[:3], bbs[:3], G['K_test'], args, depth_preds=d3, prepared=cb.poses_prepare(bbs[:3], G['K_test'], args, d3))
        assert np.array_equal(R3, Rb[:3]) and np.array_equal(t3, tb[:3])
    boxes = [[rng.uniform(-5, 2000), rng.uniform(-5, 1100), rng.uniform(0.2, 900), rng.uniform(0.2, 700)] for _ in range(200)]
    for pad in (1.2, 1.0, 1.5):
        want = np.array([list(np.array(bb).astype(np.int32)) + [int(np.maximum(np.array(bb).astype(np.int32)[3], np.array(bb).astype(np.int32)[2]) * pad)]
                         for bb in boxes], dtype=np.int32)
        assert np.array_equal(AePoseEstimator.box_rows(boxes, pad), want)
    assert AePoseEstimator.box_rows([], 1.2).shape == (0, 5)
