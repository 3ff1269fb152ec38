"""Golden fixtures for the HOST logic of the hot path, produced by running the REFERENCE's own code:

    /root/reference/auto_pose/ae/codebook.py   Codebook.nearest_rotation (:55-75: dtype handling, argmax /
                                               upright stride / top-n argpartition+sort, squeeze),
                                               Codebook.auto_pose6d (:79-129: translation + rotation correction),
                                               Codebook.nearest_rotation_batch (:131-133)
    /root/reference/auto_pose/ae/dataset.py    Dataset.viewsphere_for_embedding (:39-58), embedding_size (:375-377)
    /root/reference/auto_pose/ae/utils.py      batch_iteration_indices (:20-26), path helpers

The arithmetic that lives in TensorFlow cannot run here (not installed); everything around it can.
The modules are loaded from where they lie with their third-party imports (tensorflow, cv2,
progressbar -- none used by the functions above) replaced by empty stand-in modules, the objects are
created without their TF-graph-building constructors (object.__new__), and `session.run` is a
stand-in that returns a cosine-similarity matrix / bbox table WE provide -- i.e. exactly the
boundary the HIP kernels replace.  What is recorded is therefore the reference's behaviour from
"similarity matrix" to "indices, rotations, translations".

Run in the build container only (the reference tree does not exist on the GPU box):
    python tests/golden/make_codebook_logic_golden.py     ->  tests/golden/codebook_logic_ref.npz
"""
import configparser
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/auto_pose'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference_modules():
    tf = _stub('tensorflow')
    compat = _stub('tensorflow.compat')
    v1 = _stub('tensorflow.compat.v1', disable_eager_execution=lambda: None)
    tf.compat, compat.v1 = compat, v1
    _stub('cv2', INTER_NEAREST=0, INTER_LINEAR=1, INTER_CUBIC=2, INTER_AREA=3)     # only default-argument constants are touched
    _stub('progressbar')
    for pkg, path in (('auto_pose', REF), ('auto_pose.ae', REF + '/ae'), ('auto_pose.ae.pysixd_stuff', REF + '/ae/pysixd_stuff')):
        p = types.ModuleType(pkg)
        p.__path__ = [path]                       # a package shell: the real __init__ (which imports TF graphs) is NOT run
        sys.modules[pkg] = p
    mods = {}
    for name, rel in (('auto_pose.ae.utils', 'ae/utils.py'),
                      ('auto_pose.ae.pysixd_stuff.transform', 'ae/pysixd_stuff/transform.py'),
                      ('auto_pose.ae.pysixd_stuff.view_sampler', 'ae/pysixd_stuff/view_sampler.py'),
                      ('auto_pose.ae.dataset', 'ae/dataset.py'),
                      ('auto_pose.ae.codebook', 'ae/codebook.py')):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        mods[name.rsplit('.', 1)[1]] = m
    return mods


class FakeEncoder(object):
    x = 'encoder.x placeholder'
    latent_space_size = 128


class FakeSession(object):
    """session.run stand-in: hands back what the TF graph would have produced."""

    def __init__(self, codebook, cs, bbs, crops_u8):
        self.codebook, self.cs, self.bbs, self.crops_u8 = codebook, cs, bbs, crops_u8
        self.fed = []

    def rows(self):
        """which recorded crop each fed image is -> its row of the similarity matrix"""
        x = self.fed[-1]                          # (nearest_rotation_batch feeds its argument as is, codebook.py:132)
        if x.dtype != np.uint8:
            x = np.rint(np.asarray(x, dtype=np.float64) * 255.).astype(np.uint8)
        return np.array([int(np.flatnonzero((self.crops_u8 == c).all(axis=(1, 2, 3)))[0]) for c in x])

    def run(self, fetch, feed_dict=None):
        if feed_dict:
            self.fed.append(np.asarray(list(feed_dict.values())[0]))
        if fetch is self.codebook.cos_similarity:
            return self.cs[self.rows()]
        if fetch is self.codebook.nearest_neighbor_idx:
            return np.argmax(self.cs[self.rows()], axis=1)                      # tf.argmax: first index on ties
        if fetch is self.codebook.embed_obj_bbs_var:
            return self.bbs
        raise KeyError(fetch)


TRAIN_CFG = """
[Dataset]
K: [1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]
RADIUS: 700
"""


def main():
    mods = load_reference_modules()
    Dataset, Codebook, u = mods['dataset'].Dataset, mods['codebook'].Codebook, mods['utils']
    out = {}

    # ---- utils ---------------------------------------------------------------------------------
    for N, bs in ((92232, 64), (10, 3), (64, 64), (1, 7)):
        out['batches_%d_%d' % (N, bs)] = np.array(list(u.batch_iteration_indices(N, bs)), dtype=np.int64)
    paths = [u.get_dataset_path('/ws'), u.get_checkpoint_dir('/ws/experiments/g/e'), u.get_log_dir('/ws', 'e', 'g'),
             u.get_log_dir('/ws', 'e'), u.get_train_config_exp_file_path('/ws/experiments/g/e', 'e'),
             u.get_checkpoint_basefilename('/ws/experiments/g/e')]
    out['paths'] = np.array(paths)

    # ---- Dataset.viewsphere_for_embedding --------------------------------------------------------
    ds = object.__new__(Dataset)
    ds._kw = {'num_cyclo': '36', 'min_n_views': '42', 'radius': '700'}
    Rs_small = ds.viewsphere_for_embedding                       # 42 views x 36 in-plane = 1512 rotations
    out['viewsphere_42x36'] = Rs_small
    out['embedding_size_42x36'] = np.int64(ds.embedding_size)
    full = object.__new__(Dataset)
    full._kw = {'num_cyclo': '36', 'min_n_views': '2562', 'radius': '700'}
    Rs_full = full.viewsphere_for_embedding                      # the real 92232 x 3 x 3 table (6.6 MB): keep a digest + samples
    rows = np.array([0, 1, 35, 36, 71, 1000, 36 * 777 + 5, 46116, 92231 - 36, 92231])
    out['viewsphere_full_rows'] = rows
    out['viewsphere_full_samples'] = Rs_full[rows]
    out['viewsphere_full_sha256'] = np.array(hashlib.sha256(np.ascontiguousarray(Rs_full).tobytes()).hexdigest())
    out['embedding_size_full'] = np.int64(full.embedding_size)

    # ---- Codebook.nearest_rotation / auto_pose6d on a provided similarity matrix ---------------------
    rng = np.random.default_rng(2718)
    N, B = 1512, 6
    cs = rng.uniform(-1, 1, (B, N)).astype(np.float32)
    cs[1, 400] = cs[1, 900] = 0.999                              # exact tie: np.argmax keeps the first
    cs[2, 36 * 5] = 0.9995                                       # a multiple of num_cyclo wins outright
    cs[3, 36 * 7 + 3] = 0.9999                                   # best overall is NOT upright; upright picks its own best
    bbs = np.stack([rng.integers(200, 400, N), rng.integers(150, 300, N), rng.integers(60, 200, N), rng.integers(60, 200, N)], 1).astype(np.int32)
    cb = object.__new__(Codebook)
    cb._encoder, cb._dataset, cb.embed_bb = FakeEncoder(), ds, True
    cb.cos_similarity, cb.nearest_neighbor_idx, cb.embed_obj_bbs_var = 'cos_similarity', 'nearest_neighbor_idx', 'embed_obj_bbs_var'
    cb.embed_obj_bbs_values = None
    crops_u8 = rng.integers(0, 256, (B, 8, 8, 3), dtype=np.uint8)
    sess = FakeSession(cb, cs, bbs, crops_u8)
    out['cs'], out['bbs'] = cs, bbs

    out['idcs_top1'] = cb.nearest_rotation(sess, crops_u8, return_idcs=True)
    out['fed_after_u8'] = sess.fed[-1]                            # what the reference feeds for a uint8 batch: x/255. (float64)
    out['crops_u8'] = crops_u8
    out['idcs_upright'] = cb.nearest_rotation(sess, crops_u8, upright=True, return_idcs=True)
    out['R_batch'] = cb.nearest_rotation(sess, crops_u8)          # [B,3,3]
    out['R_single'] = cb.nearest_rotation(sess, crops_u8[0])      # HWC input -> expand_dims -> squeeze -> [3,3]
    out['fed_single_shape'] = np.array(sess.fed[-1].shape)
    crop_f = crops_u8[0].astype(np.float32) / 255.
    cb.nearest_rotation(sess, crop_f)
    out['fed_float_is_unchanged'] = np.array(np.array_equal(sess.fed[-1][0], crop_f) and sess.fed[-1].dtype == np.float32)
    for k in (2, 4, 8):
        out['idcs_top%d' % k] = cb.nearest_rotation(sess, crops_u8[0], top_n=k, return_idcs=True)
        out['R_top%d' % k] = cb.nearest_rotation(sess, crops_u8[0], top_n=k)
    out['R_batch_fn'] = cb.nearest_rotation_batch(sess, crops_u8)

    args = configparser.ConfigParser()
    args.read_string(TRAIN_CFG)
    K_test = np.array([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]])
    cases = [((250., 180., 96., 120.), 1, None, False), ((10., 20., 300., 260.), 1, None, True), ((400., 100., 64., 64.), 3, None, False),
             ((320., 240., 128., 100.), 1, 812.5, False)]
    for ci, (bb, top_n, depth, upright) in enumerate(cases):
        cb.embed_obj_bbs_values = None
        Rs_est, ts_est = cb.auto_pose6d(sess, crops_u8[ci % B], np.array(bb), K_test, top_n, args, depth_pred=depth, upright=upright)
        out['pose%d_bb' % ci] = np.array(bb)
        out['pose%d_topn_depth_upright' % ci] = np.array([top_n, -1.0 if depth is None else depth, float(upright)])
        out['pose%d_crop_row' % ci] = np.int64(ci % B)
        out['pose%d_R' % ci], out['pose%d_t' % ci] = Rs_est, ts_est
    out['K_test'] = K_test
    out['train_cfg'] = np.array(TRAIN_CFG)

    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, 'codebook_logic_ref.npz'), **out)
    for k, v in out.items():
        print(k, getattr(v, 'shape', ()), getattr(v, 'dtype', type(v)))


if __name__ == '__main__':
    main()
