"""Host logic of the reference-shaped API on CPU: cfg parsing, checkpoint round trip,
factory/ae_embed plumbing, and Codebook semantics (shapes, squeeze, top_n, upright,
auto_pose6d geometry, update_embedding) with the emulated kernels injected as engines.
Mirrors how the reference's callers use the API (auto_pose/test/aae_image.py:41-62,
auto_pose/m3_interface/ae_pose_estimator.py:61-78,157-170)."""
import configparser
import os

import numpy as np
import pytest

from augmentedautoencoder_amd import ae_embed, ae_factory as factory, session as S, utils as u, weights as W
from augmentedautoencoder_amd.codebook import Codebook, _parse_K
from augmentedautoencoder_amd.dataset import Dataset, SyntheticViewSource
from augmentedautoencoder_amd.encoder import Encoder
from emu_engines import EmuCodebookEngine, EmuEncoderEngine
from oracle import reference_cpu as ref
from oracle import synth

CFG = """
[Paths]
MODEL_PATH: /nonexistent/model.ply
BACKGROUND_IMAGES_GLOB: /nonexistent/*.jpg
[Dataset]
MODEL: reconst
H: 16
W: 16
C: 3
RADIUS: 700
RENDER_DIMS: (720, 540)
K: [1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]
[Embedding]
EMBED_BB: True
MIN_N_VIEWS: 12
NUM_CYCLO: 6
[Network]
BATCH_NORMALIZATION: False
LATENT_SPACE_SIZE: 128
NUM_FILTER: [32, 64]
STRIDES: [2, 2]
KERNEL_SIZE_ENCODER: 5
[Training]
BATCH_SIZE: 16
"""


def _args():
    a = configparser.ConfigParser()
    a.read_string(CFG)
    return a


@pytest.fixture()
def tiny():
    S.reset_default_graph()
    args = _args()
    with S.variable_scope('tiny'):
        ds = factory.build_dataset('', args)
        enc = factory.build_encoder(S.Placeholder(ds.shape), args)
        cb = factory.build_codebook(enc, ds, args)
    w = synth.make_weights(seed=8, shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2], latent=128)
    enc.load_weights(w)
    enc._engine = EmuEncoderEngine(enc.config, w)
    E = synth.make_codebook(ds.embedding_size, 128, seed=9, planted_duplicates=3, num_cyclo=6)
    cb.assign_embedding(E)
    cb._engine = EmuCodebookEngine(E)
    return args, ds, enc, cb, w, E


def test_cfg_parsing_matches_reference_keys():
    cfg = W.EncoderConfig.from_cfg(_args())
    assert cfg.shape == (16, 16, 3) and cfg.num_filter == [32, 64] and cfg.strides == [2, 2]
    assert cfg.kernel_size == 5 and cfg.latent_space_size == 128 and cfg.batch_norm is False
    assert cfg.flatten_size == 4 * 4 * 64
    d = W.EncoderConfig()                                    # defaults = cfg/train_template.cfg
    assert d.flatten_size == 32768 and d.flops_per_crop() == 2 * 2140667904
    assert _parse_K('[1075.65, 0, 720/2, 0, 1073.90, 540/2, 0, 0, 1]')[2] == 360.0
    with pytest.raises(Exception):
        _parse_K('__import__("os").system("true")')


def test_weight_validation():
    cfg = W.EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    w = synth.make_weights(seed=1, shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2], latent=128)
    arrays = W.ordered_weight_arrays(w, cfg)
    assert [a.shape for a in arrays] == [(5, 5, 3, 32), (32,), (5, 5, 32, 64), (64,), (1024, 128), (128,)]
    bad = dict(w)
    bad['conv2d_1/kernel'] = bad['conv2d_1/kernel'][:, :, :16]
    with pytest.raises(ValueError, match='conv2d_1/kernel'):
        W.ordered_weight_arrays(bad, cfg)
    del bad['dense/bias']
    with pytest.raises(ValueError):
        W.ordered_weight_arrays({k: v for k, v in w.items() if k != 'dense/bias'}, cfg)


def test_nearest_rotation_shapes_and_values(tiny):
    args, ds, enc, cb, w, E = tiny
    crops = synth.make_crops(3, seed=2, shape=(16, 16, 3))
    z64 = ref.encoder_forward_np(ref.input_to_float(crops), w, [2, 2])
    cs64 = ref.cos_similarity(z64, E)
    idcs = cb.nearest_rotation(None, crops, return_idcs=True)
    assert idcs.dtype == np.int64 and np.array_equal(idcs, np.argmax(cs64, axis=1))
    R = cb.nearest_rotation(None, crops)
    assert R.shape == (3, 3, 3) and np.array_equal(R, ds.viewsphere_for_embedding[idcs])
    assert cb.nearest_rotation(None, crops[0]).shape == (3, 3)                      # HWC input, squeezed output
    assert np.array_equal(cb.nearest_rotation(None, crops[0] / 255., return_idcs=True), idcs[:1])   # float path
    up = cb.nearest_rotation(None, crops, upright=True, return_idcs=True)
    assert np.array_equal(up, ref.nearest_indices_reference(cs64.astype(np.float32), 1, upright=True, num_cyclo=6))
    top = cb.nearest_rotation(None, crops[1], top_n=4, return_idcs=True)
    assert np.array_equal(top, ref.topk_canonical(cs64[1:2], 4)[0])
    assert cb.nearest_rotation(None, crops[1], top_n=4).shape == (4, 3, 3)
    assert cb.nearest_rotation_batch(None, crops).shape == (3, 3, 3)
    sess = S.Session()
    assert np.abs(sess.run(cb.cos_similarity, {enc.x: crops}) - cs64).max() < 1e-5
    assert np.array_equal(sess.run(cb.embedding_normalized), E)
    assert cb.test_embedding(sess, crops[0]).shape == (128,)
    assert np.abs(cb.test_embedding(sess, crops, normalized=False) - z64).max() < 1e-5


def test_empty_batch_is_an_empty_answer(tiny):
    """TF + NumPy semantics of the reference: session.run on a [0,H,W,C] feed gives a [0,N] similarity, np.argmax
    over it an empty int64 vector and Rs[idcs].squeeze() a [0,3,3] array -- no exception."""
    args, ds, enc, cb, w, E = tiny
    none = np.zeros((0, 16, 16, 3), dtype=np.uint8)
    idcs = cb.nearest_rotation(None, none, return_idcs=True)
    assert idcs.dtype == np.int64 and idcs.shape == (0,)
    assert cb.nearest_rotation(None, none).shape == (0, 3, 3)
    assert cb.nearest_rotation(None, none, upright=True, return_idcs=True).shape == (0,)
    assert cb.nearest_rotation_batch(None, none).shape == (0, 3, 3)
    sess = S.Session()
    assert sess.run(cb.cos_similarity, {enc.x: none}).shape == (0, len(E))
    assert cb.test_embedding(sess, none).shape == (0, 128)
    with pytest.raises(ValueError):
        cb.nearest_rotation(None, none, top_n=3)               # the reference's squeeze()-based top-n needs exactly one crop


def test_auto_pose6d_matches_reference_geometry(tiny):
    args, ds, enc, cb, w, E = tiny
    rng = np.random.default_rng(4)
    bbs = np.stack([rng.integers(200, 300, ds.embedding_size), rng.integers(150, 250, ds.embedding_size),
                    rng.integers(60, 200, ds.embedding_size), rng.integers(60, 200, ds.embedding_size)], 1)
    cb.assign_obj_bbs(bbs)
    crop = synth.make_crops(1, seed=6, shape=(16, 16, 3))[0]
    K_test = np.array([[572.4, 0, 325.3], [0, 573.6, 242.0], [0, 0, 1]])
    box = [310, 180, 90, 120]
    for top_n in (1, 3):
        Rs, ts = cb.auto_pose6d(None, crop, box, K_test, top_n, args)
        idcs = np.atleast_1d(cb.nearest_rotation(None, crop, top_n=top_n, return_idcs=True))
        K_train = np.array(_parse_K(args.get('Dataset', 'K'))).reshape(3, 3)
        Rw, tw = ref.auto_pose6d_geometry(idcs, ds.viewsphere_for_embedding, bbs.astype(np.int32), box, K_test, K_train, 700.0)
        assert Rs.shape == (top_n, 3, 3) and ts.shape == (top_n, 3)
        assert np.allclose(Rs, Rw, atol=1e-12) and np.allclose(ts, tw, atol=1e-9)
    Rd, td = cb.auto_pose6d(None, crop, box, K_test, 1, args, depth_pred=655.0)
    assert td[0, 2] == 655.0


def test_update_embedding_and_checkpoint_round_trip(tiny, tmp_path):
    args, ds, enc, cb, w, E = tiny
    ds.set_view_source(SyntheticViewSource(ds.shape, seed=3))
    cb.update_embedding(None, 16)                           # 72 rows: 4 batches of 16 + one of 8
    newE = cb.embedding_value()
    batch, bbs = ds.render_embedding_image_batch(0, ds.embedding_size)
    want = ref.normalize_codebook(ref.encoder_forward_np(ref.input_to_float(batch), w, [2, 2]))
    assert np.abs(newE - want).max() < 1e-5
    assert np.array_equal(cb.embed_obj_bbs_value(), np.asarray(bbs).astype(np.int32))
    ckpt_dir = str(tmp_path / 'checkpoints')
    os.makedirs(ckpt_dir)
    saver = S.Saver(scope='tiny')
    saver.save(None, os.path.join(ckpt_dir, 'chkpt'), global_step=30000)
    saver.save(None, os.path.join(ckpt_dir, 'chkpt'), global_step=30001)
    st = S.get_checkpoint_state(ckpt_dir)
    assert st.model_checkpoint_path.endswith('chkpt-30001.npz') and len(st.all_model_checkpoint_paths) == 2
    # a fresh graph restores encoder weights + codebook + bbs from the checkpoint
    S.reset_default_graph()
    with S.variable_scope('tiny'):
        ds2 = factory.build_dataset('', args)
        enc2 = factory.build_encoder(S.Placeholder(ds2.shape), args)
        cb2 = factory.build_codebook(enc2, ds2, args)
    factory.restore_checkpoint(None, S.Saver(scope='tiny'), ckpt_dir, at_step=30000)
    assert np.array_equal(cb2.embedding_value(), newE) and np.array_equal(cb2.embed_obj_bbs_value(), cb.embed_obj_bbs_value())
    for k in w:
        assert np.array_equal(enc2.weights[k], w[k])
    with pytest.raises(FileNotFoundError):
        factory.restore_checkpoint(None, None, str(tmp_path / 'nothing'))


def test_build_codebook_from_name_and_errors(tmp_path, monkeypatch):
    S.reset_default_graph()
    monkeypatch.delenv('AE_WORKSPACE_PATH', raising=False)
    with pytest.raises(RuntimeError, match='AE_WORKSPACE_PATH'):
        factory.build_codebook_from_name('obj', 'grp')
    ws = tmp_path / 'ws'
    log_dir = u.get_log_dir(str(ws), 'obj', 'grp')
    os.makedirs(log_dir)
    monkeypatch.setenv('AE_WORKSPACE_PATH', str(ws))
    with pytest.raises(FileNotFoundError):
        factory.build_codebook_from_name('obj', 'grp')
    with open(u.get_train_config_exp_file_path(log_dir, 'obj'), 'w') as f:
        f.write(CFG)
    cb, ds = factory.build_codebook_from_name('obj', 'grp', return_dataset=True)
    assert isinstance(cb, Codebook) and isinstance(ds, Dataset) and ds.embedding_size == 72
    assert cb._encoder.latent_space_size == 128 and cb._dataset is ds
    with pytest.raises(RuntimeError, match='no weights'):
        cb._encoder.engine
    with pytest.raises(NotImplementedError):
        factory.build_queue(ds, None)
    with pytest.raises(NotImplementedError):
        ds.render_embedding_image_batch(0, 4)


def test_ae_embed_cli_argument_errors(tmp_path, monkeypatch):
    monkeypatch.delenv('AE_WORKSPACE_PATH', raising=False)
    with pytest.raises(SystemExit):
        ae_embed.main(['grp/obj'])
    monkeypatch.setenv('AE_WORKSPACE_PATH', str(tmp_path))
    with pytest.raises(SystemExit, match='config'):
        ae_embed.main(['grp/obj'])


DEC_CFG = CFG.replace('KERNEL_SIZE_ENCODER: 5', 'KERNEL_SIZE_ENCODER: 5\nKERNEL_SIZE_DECODER: 5\nLOSS: L2\nBOOTSTRAP_RATIO: 4\nAUXILIARY_MASK: False\nVARIATIONAL: 0')


def test_decoder_api_reconstruction_calls_of_eval_plots(tmp_path, monkeypatch):
    """sess.run(decoder.x, {encoder.x: x}) and sess.run(decoder.x, {decoder._latent_code: z})
    (auto_pose/eval/eval_plots.py:33,59,78) through build_codebook_from_name(return_decoder=True)
    (auto_pose/eval/ae_eval.py:76) and a TF-style checkpoint holding encoder + decoder variables."""
    from augmentedautoencoder_amd import tf_checkpoint as T
    from augmentedautoencoder_amd.decoder import Decoder
    from emu_engines import EmuDecoderEngine
    from oracle import decoder_cpu as dref
    S.reset_default_graph()
    ws = tmp_path / 'ws'
    log_dir = u.get_log_dir(str(ws), 'obj', 'grp')
    os.makedirs(log_dir)
    monkeypatch.setenv('AE_WORKSPACE_PATH', str(ws))
    with open(u.get_train_config_exp_file_path(log_dir, 'obj'), 'w') as f:
        f.write(DEC_CFG)
    cb, ds, dec = factory.build_codebook_from_name('obj', 'grp', return_dataset=True, return_decoder=True)
    assert isinstance(dec, Decoder) and dec.config.num_filters == [64, 32] and dec.reconstruction_target.shape == (16, 16, 3)
    with pytest.raises(RuntimeError, match='no weights'):
        dec.engine
    w_enc = synth.make_weights(seed=3, shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2], latent=128)
    w_dec = dref.make_decoder_weights(seed=4, out_shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2], latent=128)
    assert set(w_dec) == {'dense_1/kernel', 'dense_1/bias', 'conv2d_2/kernel', 'conv2d_2/bias', 'conv2d_3/kernel', 'conv2d_3/bias'}
    ckpt_dir = u.get_checkpoint_dir(log_dir)
    os.makedirs(ckpt_dir)
    blob = {'obj/' + k: v for k, v in list(w_enc.items()) + list(w_dec.items())}
    T.write_bundle(os.path.join(ckpt_dir, 'chkpt-30000'), blob)
    with open(os.path.join(ckpt_dir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "chkpt-30000"\nall_model_checkpoint_paths: "chkpt-30000"\n')
    factory.restore_checkpoint(None, S.Saver(scope='obj'), ckpt_dir)
    enc = cb._encoder
    enc._engine = EmuEncoderEngine(enc.config, enc.weights)
    dec._engine = EmuDecoderEngine(dec.config, dec.weights)
    sess = S.Session()
    x = synth.make_crops(2, seed=8, shape=(16, 16, 3))
    reconst = sess.run(dec.x, feed_dict={enc.x: x / 255.})
    z64 = ref.encoder_forward_np(ref.input_to_float(x), w_enc, [2, 2])
    want = dref.decoder_forward_np(z64, w_dec, (16, 16, 3), [64, 32], [2, 2])
    assert reconst.shape == (2, 16, 16, 3) and reconst.dtype == np.float32 and np.abs(reconst - want).max() < 5e-6
    code = np.random.default_rng(2).standard_normal((1, 128)).astype(np.float32)
    reconst2 = sess.run(dec.x, feed_dict={dec._latent_code: code})
    assert np.abs(reconst2 - dref.decoder_forward_np(code, w_dec, (16, 16, 3), [64, 32], [2, 2])).max() < 5e-6
    # the native container keeps the decoder variables too
    S.Saver(scope='obj').save(None, os.path.join(str(tmp_path), 'native'), global_step=1)
    weights, _, _ = W.load_npz(os.path.join(str(tmp_path), 'native-1.npz'))
    assert 'dense_1/kernel' in weights and 'conv2d/kernel' in weights
    S.reset_default_graph()


def test_module_registry_is_weak_and_close_frees_the_engines():
    """The registry behind Saver(scope=...) must not keep dropped objects (and their device memory) alive; close()
    releases the engines at once and leaves the registry (m3_interface/ae_pose_estimator.py:61-78 holds N objects in
    one process for its whole life -- a service that swaps objects needs the memory back)."""
    import gc
    S.reset_default_graph()
    closed = []

    class _Eng(object):
        def close(self):
            closed.append(self)

    def build(scope):
        with S.variable_scope(scope):
            ds = Dataset('', h=16, w=16, c=3, min_n_views=12, radius=700, num_cyclo=6)
            enc = Encoder(S.Placeholder((16, 16, 3)), 128, [32, 64], 5, [2, 2], False)
            return enc, Codebook(enc, ds, True)

    enc_a, cb_a = build('a')
    enc_b, cb_b = build('b')
    assert [m[0] for m in S.graph_members()] == ['a', 'a', 'b', 'b'] and len(S.graph_members('b')) == 2
    enc_b._engine, cb_b._engine = _Eng(), _Eng()
    with cb_b:                                           # context-manager form of close()
        pass
    assert len(closed) == 1 and cb_b._engine is None and [m[0] for m in S.graph_members()] == ['a', 'a', 'b']
    cb_b.close(close_encoder=True)
    assert len(closed) == 2 and S.graph_members('b') == []
    del enc_a, cb_a, enc_b, cb_b
    gc.collect()
    assert S.graph_members() == []                       # nothing was kept alive by the registry
    S.reset_default_graph()
