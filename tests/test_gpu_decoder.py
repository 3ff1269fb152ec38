"""Decoder.x on the MI355X through the C ABI (aae_decoder_*), default network shapes
(train_template.cfg: 128x128x3, NUM_FILTER [128,256,512,512], STRIDES [2,2,2,2], k=5) against the
fp64 restatement of auto_pose/ae/decoder.py:36-84 (oracle/decoder_cpu.py).
Tolerance: reconstruction is a sigmoid image in [0,1]; |x_gpu - x_fp64| <= 2e-6 absolute, hidden
activations within 2e-5 of their maximum (fp32 accumulation over K up to 12800)."""
import numpy as np
import pytest

from augmentedautoencoder_amd.weights import DecoderConfig
from oracle import decoder_cpu as dref

pytestmark = pytest.mark.gpu


def _engine(cfg, w):
    from augmentedautoencoder_amd.engine import DecoderEngine
    return DecoderEngine(cfg, w)


def test_default_decoder_matches_fp64_oracle_and_is_deterministic():
    cfg = DecoderConfig()
    w = dref.make_decoder_weights(seed=4242)
    dec = _engine(cfg, w)
    z = np.random.default_rng(1).standard_normal((2, 128)).astype(np.float32) * 0.5
    x = dec.decode(z[:1]).cpu().numpy()
    x64, acts = dref.decoder_forward_np(z[:1], w, cfg.shape, cfg.num_filters, cfg.strides, return_activations=True)
    for i, a in enumerate(acts):
        g = dec.activation(i).cpu().numpy()
        assert np.abs(g - a).max() / np.abs(a).max() < 2e-5, 'stage %d' % i
    assert x.shape == (1, 128, 128, 3) and np.abs(x - x64).max() <= 2e-6
    _, recs = dec.decode_timed(z)
    labels = [l for l, _, _ in recs]
    assert sum('upconv2x_igemm_f32_dma' in l for l in labels) == 3 and 'upconv2x_narrow' in labels[-1]
    # batched call == per-item calls, run to run identical
    xb = dec.decode(z).cpu().numpy()
    assert np.array_equal(xb[:1], x) and np.array_equal(dec.decode(z).cpu().numpy(), xb)
    big = np.random.default_rng(2).standard_normal((300, 128)).astype(np.float32)      # > max_batch: chunked
    xl = dec.decode(big).cpu().numpy()
    assert xl.shape == (300, 128, 128, 3) and np.isfinite(xl).all()
    # (a 1-code call splits the dense layer's K loop, a 256-code chunk does not: same value up to fp32 summation order)
    assert np.abs(dec.decode(big[7:8]).cpu().numpy() - xl[7:8]).max() <= 1e-6 and np.abs(dec.decode(big[299:]).cpu().numpy() - xl[299:]).max() <= 1e-6
    dec.close()


def test_decoder_with_batch_norm_and_fallback_shapes():
    for kw in (dict(shape=(32, 48, 1), num_filter=[32, 64, 64], strides=[2, 2, 2], batch_norm=True),
               dict(shape=(24, 24, 2), num_filter=[8, 24], strides=[3, 1], latent_space_size=20),
               dict(shape=(32, 32, 3), num_filter=[32, 64], strides=[2, 2], kernel_size=3)):
        w = dref.make_decoder_weights(seed=11, out_shape=kw['shape'], num_filter=kw['num_filter'], strides=kw['strides'],
                                      kernel_size=kw.get('kernel_size', 5), latent=kw.get('latent_space_size', 128),
                                      batch_norm=kw.get('batch_norm', False))
        cfg = DecoderConfig(**kw)
        dec = _engine(cfg, w)
        z = np.random.default_rng(3).standard_normal((5, cfg.latent_space_size)).astype(np.float32)
        x = dec.decode(z).cpu().numpy()
        x64 = dref.decoder_forward_np(z, w, cfg.shape, cfg.num_filters, cfg.strides, cfg.batch_norm)
        assert np.abs(x - x64).max() <= 2e-6, kw
        dec.close()


def test_autoencoder_round_trip_encoder_into_decoder_stays_on_device():
    """eval_plots.plot_reconstruction_test: sess.run(decoder.x, {encoder.x: x})."""
    from augmentedautoencoder_amd import session as S
    from augmentedautoencoder_amd.decoder import Decoder
    from augmentedautoencoder_amd.encoder import Encoder
    from oracle import reference_cpu as ref, synth
    S.reset_default_graph()
    enc = Encoder(S.Placeholder((128, 128, 3)), 128, [128, 256, 512, 512], 5, [2, 2, 2, 2], False)
    dec = Decoder(S.Placeholder((128, 128, 3)), enc.z, [512, 512, 256, 128], 5, [2, 2, 2, 2], encoder=enc)
    w_enc, w_dec = synth.make_weights(seed=2024), dref.make_decoder_weights(seed=4242)
    enc.load_weights(w_enc)
    dec.load_weights(w_dec)
    x = synth.make_crops(3, seed=5)
    reconst = S.Session().run(dec.x, feed_dict={enc.x: x})
    z64 = ref.encoder_forward_torch(ref.input_to_float(x), w_enc, [2, 2, 2, 2], False, 'float64')
    want = dref.decoder_forward_np(z64[:1], w_dec, (128, 128, 3), [512, 512, 256, 128], [2, 2, 2, 2])
    assert reconst.shape == (3, 128, 128, 3) and np.abs(reconst[:1] - want).max() <= 5e-6
    S.reset_default_graph()
