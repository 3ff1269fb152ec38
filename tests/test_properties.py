"""Randomised property tests (hypothesis, bounded example counts): checkpoint container round trips,
crop kernel == oracle on arbitrary boxes, top-k / arg-max selection == canonical order under heavy ties."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import emu_backend as eb
from augmentedautoencoder_amd import tf_checkpoint as T
from augmentedautoencoder_amd.pose_estimator import AePoseEstimator
from oracle import reference_cpu as ref

# derandomize: the same examples on every run (no flaky surprises in CI); widen locally with --hypothesis-seed
FEW = settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


@FEW
@given(st.binary(min_size=0, max_size=5000), st.integers(0, 2 ** 32 - 1))
def test_crc32c_is_incremental_and_mask_round_trips(data, seed):
    cut = seed % (len(data) + 1)
    assert T.crc32c(data[cut:], T.crc32c(data[:cut])) == T.crc32c(data)
    assert T.unmask_crc(T.mask_crc(seed)) == seed


@FEW
@given(st.dictionaries(st.text(alphabet='abcdefghij/_0123456789', min_size=1, max_size=40),
                       st.tuples(st.sampled_from(['float32', 'int32', 'int64', 'float64', 'uint8']),
                                 st.lists(st.integers(0, 7), min_size=0, max_size=4)), min_size=1, max_size=12),
       st.integers(64, 2048))
def test_bundle_round_trip_for_arbitrary_names_shapes_and_block_sizes(tmp_path_factory, spec, block_size):
    rng = np.random.default_rng(len(spec) + block_size)
    tensors = {name: (rng.standard_normal(shape) * 100).astype(dt) for name, (dt, shape) in spec.items()}
    d = str(tmp_path_factory.mktemp('bundle'))
    prefix = os.path.join(d, 'ck')
    T.write_bundle(prefix, tensors)
    # re-write the index with a small block size so that keys spill over several prefix-compressed blocks
    pairs = T.read_table(prefix + '.index')
    T.write_table(prefix + '.index', pairs, block_size=block_size)
    r = T.BundleReader(prefix)
    assert r.names() == sorted(tensors)
    for name, a in tensors.items():
        got = r.tensor(name)
        assert got.dtype == a.dtype and got.shape == a.shape and np.array_equal(got, a)


@FEW
@given(st.integers(0, 10 ** 6), st.integers(1, 6), st.sampled_from([(16, 16), (24, 20), (7, 33)]))
def test_crop_kernel_equals_oracle_on_random_boxes(seed, n_boxes, out_hw):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(20, 90)), int(rng.integers(20, 120))
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    boxes = []
    for _ in range(n_boxes):                                      # boxes may stick out of the image on every side
        w, h = rng.uniform(2, W * 1.2), rng.uniform(2, H * 1.2)
        boxes.append([rng.uniform(-0.3 * W, W), rng.uniform(-0.3 * H, H), w, h])
    pad = float(rng.choice([1.0, 1.2, 1.6]))
    got = eb.crop_resize(img, AePoseEstimator.box_rows(boxes, pad), out_hw)
    for i, bb in enumerate(boxes):
        want = ref.extract_square_patch_black_borders(img, bb, pad, resize=(out_hw[1], out_hw[0]))
        assert np.array_equal(got[i], want), (bb, pad)


@FEW
@given(st.integers(0, 10 ** 6), st.integers(130, 2300), st.integers(1, 9), st.integers(1, 3))
def test_nearest_and_topk_selection_under_heavy_ties(seed, N, k, B):
    """Quantised codebook rows produce many exactly equal scores: indices must follow the canonical order
    (score descending, lower index first) of the kernel's own similarity matrix."""
    rng = np.random.default_rng(seed)
    J = 128
    base = rng.integers(-2, 3, (max(N // 7, 2), J)).astype(np.float32)
    E = base[rng.integers(0, len(base), N)]                        # many duplicate rows
    E[np.abs(E).sum(axis=1) == 0, 0] = 1.0
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    z = E[rng.integers(0, N, B)] * rng.uniform(0.5, 4.0, (B, 1)).astype(np.float32)
    cb = eb.EmuCodebook(E.astype(np.float32))
    cs = cb.similarity(z)
    idx, score = cb.nn(z)
    assert np.array_equal(idx[:, 0], np.argmax(cs, axis=1)) and np.array_equal(score[:, 0], cs.max(axis=1))
    k = min(k, N)
    ik, sk = cb.nn(z, topk=k)
    assert np.array_equal(ik, ref.topk_canonical(cs, k)) and np.array_equal(sk, np.take_along_axis(cs, ik, axis=1))
    cb.close()


@settings(max_examples=4, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(st.integers(0, 10 ** 6))
def test_random_small_encoders_on_the_emulated_kernels(seed):
    """Random [Network] shapes through the emulated HIP kernels (first-layer MFMA kernel, implicit GEMM in its
    register-staged / LDS-DMA / weights-to-registers forms, split-K, BN, f32x3h) vs the fp64 oracle."""
    from augmentedautoencoder_amd.weights import EncoderConfig
    from oracle import synth
    rng = np.random.default_rng(seed)
    H, W = int(rng.choice([8, 12, 16, 20])), int(rng.choice([8, 12, 16, 24]))
    n_layers = int(rng.integers(2, 4))
    filters = [int(rng.choice([32, 64, 96])) for _ in range(n_layers)]
    strides = [int(rng.choice([1, 2])) for _ in range(n_layers)]
    cfg = EncoderConfig((H, W, int(rng.choice([1, 3]))), filters, strides, 5, int(rng.choice([32, 128])), bool(rng.integers(0, 2)))
    if cfg.flatten_size % 32:
        return
    B = int(rng.integers(1, 5))
    w = synth.make_weights(seed=seed % 1000, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides,
                           latent=cfg.latent_space_size, batch_norm=cfg.batch_norm)
    x = synth.make_crops(B, seed=seed % 997, shape=cfg.shape)
    z64, acts = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, cfg.batch_norm, return_activations=True)
    enc = eb.EmuEncoder(w, cfg)
    dma, breg, nosplit, x3h = (int(v) for v in rng.integers(0, 2, 4))
    enc.set_option('igemm_dma', dma)
    enc.set_option('igemm_breg', breg)
    if nosplit:
        enc.set_option('splitk_min_base_blocks', 0)
    z = enc.forward(x)
    for i, a in enumerate(acts):
        err = np.abs(enc.activation(i) - a).max() / max(np.abs(a).max(), 1e-9)
        assert err < 5e-6, (cfg.shape, filters, strides, B, dma, breg, nosplit, i, err, enc.labels())
    assert np.abs(z - z64).max() / np.abs(z64).max() < 5e-6
    if x3h:
        enc.set_option('precision', 1)
        z3 = enc.forward(x)
        assert np.abs(z3 - z64).max() / np.abs(z64).max() < 5e-6, (cfg.shape, filters, strides, B, enc.labels())
    enc.close()
