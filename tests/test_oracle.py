"""The oracle against itself: three independent implementations (numpy im2col fp64,
torch conv2d, plain-C direct loops) must agree, and the host-side NumPy semantics
the reference relies on (arg-max ties, top-n, upright, float64 normalisation,
batch iteration) are pinned.  The reference ships no golden vectors for this path
("parity unpinned", see oracle/reference_cpu.py)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import reference_cpu as ref
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def coracle():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    return ctypes.CDLL(os.path.join(ROOT, 'oracle', 'libaae_oracle.so'))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@given(st.integers(1, 40), st.integers(1, 7), st.integers(1, 4))
@settings(max_examples=200, deadline=None)
def test_same_padding_formula(size, k, s):
    out, before, after = ref.same_pad(size, k, s)
    assert out == -(-size // s)
    assert before >= 0 and after >= before and after - before <= 1      # extra pixel goes at the end
    assert (out - 1) * s + k <= size + before + after                    # last window fits
    if before + after > 0:
        assert (out - 1) * s + k == size + before + after               # ... exactly


def test_default_network_padding_is_1_before_2_after():
    assert ref.same_pad(128, 5, 2) == (64, 1, 2)
    assert ref.same_pad(8, 5, 2) == (4, 1, 2)


@pytest.mark.parametrize('shape,filters,strides,k,bn', [
    ((12, 10, 3), [8, 16], [2, 2], 5, False),
    ((9, 13, 1), [6], [1], 3, True),
    ((16, 16, 3), [32, 64], [2, 2], 5, True),
    ((7, 7, 2), [4, 4, 4], [2, 1, 2], 5, False),
])
def test_numpy_torch_and_c_oracles_agree(coracle, shape, filters, strides, k, bn):
    w = synth.make_weights(seed=3, shape=shape, num_filter=filters, strides=strides, kernel_size=k, latent=10, batch_norm=bn)
    x = ref.input_to_float(synth.make_crops(3, seed=4, shape=shape))
    z_np, acts_np = ref.encoder_forward_np(x, w, strides, bn, np.float64, return_activations=True)
    z_t, acts_t = ref.encoder_forward_torch(x, w, strides, bn, 'float64', return_activations=True)
    assert np.abs(z_np - z_t).max() < 1e-12
    for a, b in zip(acts_np, acts_t):
        assert np.abs(a - b).max() < 1e-12
    z32 = ref.encoder_forward_torch(x, w, strides, bn, 'float32')
    assert np.abs(z32 - z_np).max() / np.abs(z_np).max() < 1e-5
    # plain C, layer by layer, fp64
    convs, bns = ref.layer_names(len(strides), bn)
    h = np.ascontiguousarray(x.astype(np.float32).astype(np.float64))
    for i, s in enumerate(strides):
        kern = np.ascontiguousarray(w[convs[i] + '/kernel'].astype(np.float64))
        bias = np.ascontiguousarray(w[convs[i] + '/bias'].astype(np.float64))
        B, H, W, C = h.shape
        Ho, Wo = ref.same_pad(H, k, s)[0], ref.same_pad(W, k, s)[0]
        out = np.zeros((B, Ho, Wo, kern.shape[3]))
        sc = sh = None
        if bn:
            g, be, mu, var = (w[bns[i] + '/' + n].astype(np.float64) for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'))
            inv = 1.0 / np.sqrt(var + 1e-3) * g
            sc, sh = np.ascontiguousarray(inv), np.ascontiguousarray(be - mu * inv)
        coracle.aae_oracle_conv2d_f64(_p(h), B, H, W, C, _p(kern), k, k, kern.shape[3], _p(bias), s, 1,
                                      _p(sc) if bn else None, _p(sh) if bn else None, _p(out))
        assert np.abs(out - acts_np[i]).max() < 1e-12, 'layer %d' % i
        h = out
    flat = np.ascontiguousarray(h.reshape(h.shape[0], -1))
    z_c = np.zeros((flat.shape[0], 10))
    dk = np.ascontiguousarray(w['dense/kernel'].astype(np.float64))
    db = np.ascontiguousarray(w['dense/bias'].astype(np.float64))
    coracle.aae_oracle_dense_f64(_p(flat), flat.shape[0], flat.shape[1], _p(dk), 10, _p(db), _p(z_c))
    assert np.abs(z_c - z_np).max() < 1e-12


def test_c_oracle_cos_argmax_matches_numpy(coracle):
    E = synth.make_codebook(36 * 9, 16, seed=1, planted_duplicates=5)
    z = np.random.default_rng(2).standard_normal((6, 16))
    z[0] = E[36 * 2 + 35] * 2.5                                    # exact tie with row 72
    q = np.ascontiguousarray(ref.l2_normalize(z))
    qc = np.zeros_like(q)
    coracle.aae_oracle_l2_normalize_f64(_p(np.ascontiguousarray(z)), 6, 16, _p(qc))
    assert np.abs(q - qc).max() < 1e-15
    for stride in (1, 36):
        cs = np.zeros((6, E.shape[0]))
        idx = np.zeros(6, dtype=np.int64)
        best = np.zeros(6)
        coracle.aae_oracle_cos_argmax_f64(_p(qc), 6, _p(E), E.shape[0], 16, stride, _p(cs), _p(idx), _p(best))
        cs_np = ref.cos_similarity(z, E)
        assert np.abs(cs - cs_np).max() < 1e-14
        assert np.array_equal(idx, ref.nearest_indices_reference(cs, 1, upright=(stride > 1), num_cyclo=36))


def test_input_conversion_table():
    lut = ref.u8_lut_f32()
    x = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1)
    assert np.array_equal(ref.input_to_float(x).astype(np.float32).ravel(), lut)
    assert lut[0] == 0.0 and lut[255] == 1.0
    naive = np.arange(256, dtype=np.float32) * np.float32(1.0 / 255.0)
    assert np.any(naive != lut)          # why the kernels use a table, not a multiply
    assert ref.input_to_float(np.zeros((4, 4, 3), np.uint8)).shape == (1, 4, 4, 3)


def test_l2_normalize_epsilon_and_zero_vector():
    z = np.zeros((2, 8))
    z[1, 0] = 1e-7                                                   # sum sq 1e-14 < eps 1e-12
    q = ref.l2_normalize(z)
    assert np.all(q[0] == 0)
    assert abs(q[1, 0] - 1e-7 / 1e-6) < 1e-12                        # divided by sqrt(eps), not by the norm


def test_argmax_first_index_on_ties_and_upright_and_topn():
    cs = np.zeros((2, 72), dtype=np.float32)
    cs[0, [5, 40]] = 0.9
    cs[1, [36, 37]] = 0.5
    cs[1, 37] = 0.7
    assert ref.nearest_indices_reference(cs, 1).tolist() == [5, 37]
    assert ref.nearest_indices_reference(cs, 1, upright=True, num_cyclo=36).tolist() == [0, 36]
    row = np.array([[0.1, 0.9, 0.3, 0.8, 0.5]], dtype=np.float32)
    assert ref.nearest_indices_reference(row, 3).tolist() == [1, 3, 4]
    assert ref.topk_canonical(row, 3).tolist() == [[1, 3, 4]]
    tie = np.array([[0.2, 0.9, 0.9, 0.1]], dtype=np.float32)
    assert ref.topk_canonical(tie, 2).tolist() == [[1, 2]]


def test_batch_iteration_and_codebook_normalisation():
    idx = list(ref.batch_iteration_indices(92232, 64))
    assert len(idx) == 1442 and idx[0] == (0, 64) and idx[-1] == (92224, 92232)
    assert list(ref.batch_iteration_indices(128, 64)) == [(0, 64), (64, 128)]
    z = np.random.default_rng(0).standard_normal((10, 7))
    E = ref.normalize_codebook(z)
    assert E.dtype == np.float32
    assert np.abs(np.linalg.norm(E.astype(np.float64), axis=1) - 1).max() < 1e-7
