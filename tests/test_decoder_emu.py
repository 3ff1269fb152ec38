"""Decoder.x (next row N4) on the CPU fiber emulator vs the fp64 restatement of
auto_pose/ae/decoder.py:36-84 (oracle/decoder_cpu.py): phase-folded 2x upsample + conv on the
matrix cores (SCATTER igemm), the narrow sigmoid output layer, BN, and the shape-agnostic fallback."""
import numpy as np
import pytest

import emu_backend as eb
from augmentedautoencoder_amd.weights import DecoderConfig
from oracle import decoder_cpu as dref


def _run(cfg_kw, B, seed):
    w = dref.make_decoder_weights(seed=seed, out_shape=cfg_kw['shape'], num_filter=cfg_kw['num_filter'], strides=cfg_kw['strides'],
                                  kernel_size=cfg_kw.get('kernel_size', 5), latent=cfg_kw.get('latent_space_size', 128),
                                  batch_norm=cfg_kw.get('batch_norm', False))
    cfg = DecoderConfig(**cfg_kw)
    z = np.random.default_rng(seed + 1).standard_normal((B, cfg.latent_space_size)).astype(np.float32)
    dec = eb.EmuDecoder(w, cfg)
    x = dec.forward(z)
    x64, acts = dref.decoder_forward_np(z, w, cfg.shape, cfg.num_filters, cfg.strides, cfg.batch_norm, return_activations=True)
    for i, a in enumerate(acts):
        err = np.abs(dec.activation(i) - a).max() / max(np.abs(a).max(), 1e-9)
        assert err < 5e-6, 'stage %d rel err %.2e (%s)' % (i, err, dec.labels())
    assert x.shape == x64.shape and np.abs(x - x64).max() < 2e-6
    assert x.min() >= 0.0 and x.max() <= 1.0
    labels = dec.labels()
    dec.close()
    return labels


def test_phase_folded_upconv_on_matrix_cores_and_narrow_output_layer():
    labels = _run(dict(shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2]), 2, 3)
    assert labels[0].startswith('dense:conv_igemm_f32') and 'upconv2x_narrow' in labels[-1]
    assert any(l.startswith('up1:upconv2x_igemm') for l in labels)


def test_three_stage_decoder_with_batch_norm_and_grayscale_output():
    labels = _run(dict(shape=(16, 24, 1), num_filter=[32, 32, 64], strides=[2, 2, 2], batch_norm=True), 1, 5)
    assert sum('upconv2x_igemm' in l for l in labels) == 2 and 'upconv2x_narrow' in labels[-1]


def test_kernel_size_3_phases_have_different_padding():
    _run(dict(shape=(8, 8, 3), num_filter=[32, 32], strides=[2, 2], kernel_size=3), 2, 7)


def test_fallback_for_odd_channel_counts_stride_1_and_3x_upsampling():
    labels = _run(dict(shape=(12, 12, 2), num_filter=[8, 24], strides=[3, 1], latent_space_size=20), 2, 9)
    assert all('upconv_direct' in l or 'generic' in l for l in labels), labels
    labels = _run(dict(shape=(8, 8, 3), num_filter=[32, 32, 64], strides=[2, 1, 2]), 1, 11)      # stride 1: plain igemm stage
    assert any(l.startswith('up2:conv_igemm_f32') for l in labels), labels


def test_decoder_argument_errors():
    cfg = DecoderConfig((16, 16, 3), [32, 64], [2, 2])
    w = dref.make_decoder_weights(out_shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2])
    bad = dict(w)
    bad.pop('conv2d_3/bias')
    with pytest.raises(ValueError, match='missing decoder weight'):
        eb.EmuDecoder(bad, cfg)
    bad = dict(w)
    bad['dense_1/kernel'] = bad['dense_1/kernel'][:, :-1]
    with pytest.raises(ValueError, match='shape'):
        eb.EmuDecoder(bad, cfg)
    with pytest.raises(ValueError, match='even kernel'):
        eb.EmuDecoder(dref.make_decoder_weights(out_shape=(16, 16, 3), num_filter=[32, 64], strides=[2, 2], kernel_size=4),
                      DecoderConfig((16, 16, 3), [32, 64], [2, 2], kernel_size=4))
