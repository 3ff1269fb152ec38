"""The HIP kernel sources + launch planning, executed on the CPU fiber emulator
(tests/emu/) against the fp64 oracle: tile index math, MFMA fragment maps, SAME
padding, masking of partial tiles, split-K, BN epilogue, scan/argmax/top-k.  Sizes are
tiny (the emulator runs one fiber per GPU thread); full sizes run under -m gpu."""
import numpy as np
import pytest

import emu_backend as eb
from augmentedautoencoder_amd import _lib
from augmentedautoencoder_amd.weights import EncoderConfig
from oracle import reference_cpu as ref
from oracle import synth


def _run(cfg, B, seed, f32_in=False, nosplit=False, dma=0, breg=0, wavek=0, options=None, chain=0):
    w = synth.make_weights(seed=seed, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides,
                           latent=cfg.latent_space_size, batch_norm=cfg.batch_norm, kernel_size=cfg.kernel_size)
    x = synth.make_crops(B, seed=seed + 1, shape=cfg.shape)
    xin = ref.input_to_float(x).astype(np.float32) if f32_in else x
    enc = eb.split_k_small_batches(eb.EmuEncoder(w, cfg))
    if nosplit:
        enc.set_option('splitk_min_base_blocks', 0)
        enc.set_option('dense_gemv', 0)         # keep the dense layer on the (un-split) MFMA tile too
    enc.set_option('igemm_dma', dma)           # operand slabs by LDS-DMA instead of register staging
    enc.set_option('igemm_breg', breg)         # weights straight from global memory into the MFMA B fragments
    enc.set_option('wavek', wavek)             # 0: the 128 x 128 split-K igemm + reduce launch also for small batches
    enc.set_option('wavek_dense', wavek)
    enc.set_option('gemv_ticket', wavek)
    enc.set_option('detect_chain', chain)      # 0: B <= 4 as stand-alone launches (the kernels most tests here are about)
    enc.set_option('planner_cost_model', 0)    # kernel family / tile shape by the thresholds the tests steer (the cost model has its own test)
    for k, v in (options or {}).items():
        enc.set_option(k, v)
    z = enc.forward(xin)
    z64, acts = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, cfg.batch_norm, return_activations=True)
    for i, a in enumerate(acts):
        err = np.abs(enc.activation(i) - a).max() / max(np.abs(a).max(), 1e-9)
        assert err < 5e-6, 'layer %d rel err %.2e (%s)' % (i, err, enc.labels())
    assert np.abs(z - z64).max() / np.abs(z64).max() < 5e-6
    labels = enc.labels()
    enc.close()
    return labels


@pytest.mark.parametrize('dma', [0, 1])
def test_first_layer_mfma_igemm_splitk_and_dense(dma):
    labels = _run(EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128), 3, 1, dma=dma)
    assert 'conv_first_f32' in labels[0] and 'splitk' in labels[1] and labels[-1].endswith('splitk_reduce')
    assert ('f32_dma' in labels[1]) == bool(dma)


@pytest.mark.parametrize('dma,breg', [(0, 0), (1, 0), (1, 1)])
def test_unsplit_igemm_epilogue_with_batchnorm(dma, breg):
    labels = _run(EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128, True), 2, 41, nosplit=True, dma=dma, breg=breg)
    assert all('splitk' not in l for l in labels)
    assert any('dma_breg' in l for l in labels) == bool(breg)


def test_weights_to_registers_variant_partial_tiles_and_several_slabs():
    # conv2: M = 5*8*8 = 320 rows (two full 128-row tiles + a partial one), K = 25 taps x 32 channels = 25 slabs
    _run(EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128), 5, 61, nosplit=True, dma=1, breg=1)


def test_partial_tiles_row_straddling_and_odd_channel_counts():
    # conv1 output 20x12 = 240 px: one full 128-px tile + a partial one, rows straddle tiles;
    # Cout 48 (not a multiple of 32/128); layer 2 falls back to the generic kernel (Cin 48)
    labels = _run(EncoderConfig((40, 24, 3), [48, 32], [2, 2], 5, 20, True), 2, 11, f32_in=True)
    assert 'conv_direct_generic' in labels[1]


@pytest.mark.parametrize('dma', [0, 1])
def test_grayscale_stride1_and_two_channel_blocks(dma):
    _run(EncoderConfig((12, 12, 1), [160, 32], [1, 2], 5, 128), 1, 21, dma=dma)


def test_generic_fallback_everything():
    labels = _run(EncoderConfig((10, 14, 3), [24, 8], [2, 1], 3, 12, True), 2, 31)
    assert all('generic' in l for l in labels)


@pytest.mark.parametrize('B,mode', [(1, _lib.AAE_SCAN_GEMV), (4, _lib.AAE_SCAN_GEMV), (3, _lib.AAE_SCAN_MFMA),
                                    (1, _lib.AAE_SCAN_STREAM), (2, _lib.AAE_SCAN_STREAM), (3, _lib.AAE_SCAN_AUTO),
                                    (1, _lib.AAE_SCAN_STREAM_WALK), (4, _lib.AAE_SCAN_STREAM_WALK),
                                    (33, _lib.AAE_SCAN_AUTO), (70, _lib.AAE_SCAN_AUTO)])
def test_codebook_scan_kernels(B, mode):
    N, J = 36 * 11 + 5, 128                                   # 401 rows: 3 full 128-row blocks + a partial one
    E = synth.make_codebook(N, J, seed=7, planted_duplicates=11)
    assert np.array_equal(E[36 * 3], E[36 * 3 + 35])
    cb = eb.EmuCodebook(E)
    cb.set_mode(mode)
    rows = np.random.default_rng(B).integers(0, N, B)
    z = synth.make_queries_near_rows(E, rows, noise=0.3, seed=B)
    z[0] = E[36 * 3 + 35] * 3.0                               # exact tie between rows 108 and 143
    idx, score = cb.nn(z)
    cs = cb.similarity(z)
    cs64 = ref.cos_similarity(z, E)
    assert np.abs(cs - cs64).max() < 2e-6
    assert np.array_equal(idx[:, 0], np.argmax(cs, axis=1))
    assert idx[0, 0] == 108                                   # lower-index twin wins
    assert np.abs(score[:, 0] - cs64.max(axis=1)).max() < 2e-6
    up, _ = cb.nn(z, col_stride=36)
    assert np.array_equal(up[:, 0], ref.nearest_indices_reference(cs, 1, upright=True, num_cyclo=36))
    if B <= 4:
        ik, sk = cb.nn(z, topk=5)
        assert np.array_equal(ik, ref.topk_canonical(cs, 5))
        assert np.all(np.diff(sk, axis=1) <= 0)
    cb.close()


@pytest.mark.parametrize('B', [1, 2, 3, 4])
def test_walking_stream_scan_equals_the_one_batch_per_wave_form_bitwise(B):
    """B <= 4: scan_stream_walk_kernel (a block per compute unit -- three on the emulator -- walks the codebook, two 32-row
    batches in flight per wave, running best in scalar registers) against scan_stream_kernel (one batch per wave): indices,
    scores and similarity rows bit for bit, with a ragged last batch, exact ties inside a batch, across the two batches a wave
    has in flight and across waves, a zero latent (every score 0 -> row 0), and the upright stride."""
    N, J = 36 * 60 + 7, 128                                   # 2167 rows = 68 batches: five or six per wave
    E = synth.make_codebook(N, J, seed=17, planted_duplicates=0)
    E[5] = E[36 * 40]                                         # twin rows far apart (different waves), close together (same batch) ...
    E[36 * 40 + 3] = E[36 * 40 + 1]
    E[36 * 20 + 384] = E[36 * 20]                             # ... and 12 batches apart: the two rings of one wave (12 waves on the emulator)
    rows = [36 * 40, 36 * 40 + 3, 36 * 20 + 384, 100][:B]
    z = np.stack([E[r] * (1.5 + i) for i, r in enumerate(rows)]).astype(np.float32)
    if B == 4:
        z[3] = 0.0
    want = [5, 36 * 40 + 1, 36 * 20, 0][:B]
    got = {}
    for mode in (_lib.AAE_SCAN_STREAM_WALK, _lib.AAE_SCAN_AUTO):
        cb = eb.EmuCodebook(E)
        cb.set_mode(mode)
        idx, score = cb.nn(z)
        up, ups = cb.nn(z, col_stride=36)
        got[mode] = (idx.copy(), score.copy(), cb.similarity(z).copy(), up.copy(), ups.copy())
        cb.close()
    a, b = got[_lib.AAE_SCAN_STREAM_WALK], got[_lib.AAE_SCAN_AUTO]
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    assert a[0][:, 0].tolist() == want
    cs64 = ref.cos_similarity(z, E)
    assert np.abs(a[2] - cs64).max() < 2e-6
    assert np.array_equal(a[3][:, 0], ref.nearest_indices_reference(a[2], 1, upright=True, num_cyclo=36))


def test_small_latent_and_l2_normalize():
    E = synth.make_codebook(36 * 4, 32, seed=3, planted_duplicates=2)
    cb = eb.EmuCodebook(E)
    z = np.random.default_rng(1).standard_normal((5, 32)).astype(np.float32)
    z[4] = 0.0                                                 # zero latent: eps path, all scores 0 -> index 0
    for mode in (_lib.AAE_SCAN_MFMA,):
        cb.set_mode(mode)
        idx, score = cb.nn(z)
        cs64 = ref.cos_similarity(z, E)
        assert np.array_equal(idx[:4, 0], np.argmax(cs64[:4], axis=1))
        assert idx[4, 0] == 0 and score[4, 0] == 0.0
    assert np.abs(eb.l2_normalize(z) - ref.l2_normalize(z)).max() < 1e-7
    cb.close()


@pytest.mark.parametrize('dma', [0, 1])
@pytest.mark.parametrize('cfg,B,nosplit', [
    (EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128), 3, False),
    (EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128, True), 2, True),
    (EncoderConfig((12, 12, 1), [160, 32], [1, 2], 5, 128), 1, False),
    (EncoderConfig((32, 32, 3), [64, 32], [2, 2], 5, 128), 1, False),      # conv1: two full 128-pixel tiles -> lane-pair packed plane stores
])
def test_split_precision_f32x3h_path(cfg, B, nosplit, dma):
    """f32x3h (3 fp16 MFMAs per product on (hi, lo) operand pairs, fp32 accumulate): same
    fp32-roundoff error class as the exact fp32 path, measured against the fp64 oracle."""
    w = synth.make_weights(seed=5, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides,
                           latent=cfg.latent_space_size, batch_norm=cfg.batch_norm)
    x = synth.make_crops(B, seed=6, shape=cfg.shape)
    enc = eb.split_k_small_batches(eb.EmuEncoder(w, cfg))
    enc.set_option('precision', 1)
    enc.set_option('x3h_dma', dma)       # operand slabs by LDS-DMA instead of register staging
    if nosplit:
        enc.set_option('splitk_min_base_blocks', 0)
    z = enc.forward(x)
    assert any(('x3h_dma' if dma else 'x3h') in l for l in enc.labels())
    z64, acts = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, cfg.batch_norm, return_activations=True)
    for i, a in enumerate(acts):
        assert np.abs(enc.activation(i) - a).max() / np.abs(a).max() < 5e-6, 'layer %d' % i
    assert np.abs(z - z64).max() / np.abs(z64).max() < 5e-6
    enc.close()
    gen = eb.EmuEncoder(synth.make_weights(seed=1, shape=(10, 14, 3), num_filter=[24, 8], strides=[2, 1], kernel_size=3, latent=12),
                        EncoderConfig((10, 14, 3), [24, 8], [2, 1], 3, 12))
    with pytest.raises(ValueError, match='f32x3h'):
        gen.set_option('precision', 1)                     # needs the matrix-core kernels on every layer
    gen.close()


@pytest.mark.parametrize('B,mode', [(1, _lib.AAE_SCAN_AUTO), (2, _lib.AAE_SCAN_AUTO), (4, _lib.AAE_SCAN_AUTO), (1, _lib.AAE_SCAN_MFMA),
                                    (5, _lib.AAE_SCAN_AUTO), (70, _lib.AAE_SCAN_AUTO)])
def test_bf16_codebook_scan(B, mode):
    """BASELINE config 5 in miniature: bf16 codebook rows, queries as two bf16 terms on the bf16 matrix cores (worst case
    3.8e-6 for unit vectors, codebook_scan_bf16.h; the B <= 4 streaming kernel keeps fp32 queries); parity against the fp64
    oracle evaluated on the bf16-rounded codebook."""
    from augmentedautoencoder_amd.weights import bf16_bits_to_f32, to_bf16_bits
    N, J = 36 * 11 + 5, 128
    E = synth.make_codebook(N, J, seed=7, planted_duplicates=11)
    Eb = bf16_bits_to_f32(to_bf16_bits(E))
    assert np.abs(Eb - E).max() < 2.0 ** -8 and np.array_equal(to_bf16_bits(Eb), to_bf16_bits(E))
    cb = eb.EmuCodebook(E, dtype='bf16')
    cb.set_mode(mode)                                         # B <= 4: streaming kernel unless the MFMA kernel is forced
    rows = np.random.default_rng(B).integers(0, N, B)
    z = synth.make_queries_near_rows(E, rows, noise=0.3, seed=B)
    z[0] = Eb[36 * 3 + 35] * 3.0                              # exact tie between rows 108 and 143
    idx, score = cb.nn(z)
    cs = cb.similarity(z)
    cs64 = ref.cos_similarity(z, Eb)
    assert np.abs(cs - cs64).max() < 2e-6
    assert np.array_equal(idx[:, 0], np.argmax(cs, axis=1)) and idx[0, 0] == 108
    up, _ = cb.nn(z, col_stride=36)
    assert np.array_equal(up[:, 0], ref.nearest_indices_reference(cs, 1, upright=True, num_cyclo=36))
    if B <= 5:
        ik, sk = cb.nn(z, topk=5)
        assert np.array_equal(ik, ref.topk_canonical(cs, 5)) and np.all(np.diff(sk, axis=1) <= 0)
    cb.close()
    with pytest.raises(ValueError, match='J == 128'):
        eb.EmuCodebook(synth.make_codebook(72, 32, seed=1), dtype='bf16')


def test_topk_two_level_selection_across_chunks():
    """top-k over several 2048-entry chunks (tail chunk of 77 rows, k larger than the tail, ties that
    straddle a chunk boundary): canonical order = score descending, lower index first."""
    N, J = 2 * 2048 + 77, 128
    E = synth.make_codebook(N, J, seed=3, planted_duplicates=0)
    E[2047] = E[2048] = E[5]                                   # three identical rows, two of them across the chunk edge
    E[N - 1] = E[4100]                                         # a tie inside the tail chunk
    cb = eb.EmuCodebook(E)
    z = np.stack([E[5] * 2.0, E[4100] * 0.5 + 0.01 * E[7], np.random.default_rng(0).standard_normal(J).astype(np.float32)])
    cs = cb.similarity(z)
    for k in (1, 7, 100):
        ik, sk = cb.nn(z, topk=k)
        assert np.array_equal(ik, ref.topk_canonical(cs, k)), k
        assert np.array_equal(sk, np.take_along_axis(cs, ik, axis=1))
    ik, _ = cb.nn(z, topk=7)
    assert list(ik[0, :3]) == [5, 2047, 2048]
    cb.close()


def test_f32x3h_wide_tile_variant_is_bit_identical():
    """conv_igemm_x3h_dma_kernel<.., WM=4>: 256 x 128 tiles / 8 waves (host option x3h_wide_min_blocks) must
    reproduce the 128 x 128 kernel bit for bit, including a partial second M tile."""
    cfg = EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128)
    w = synth.make_weights(seed=5, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128)
    x = synth.make_crops(5, seed=6, shape=cfg.shape)                     # conv2: M = 320 = one full + one partial 256-row tile
    enc = eb.split_k_small_batches(eb.EmuEncoder(w, cfg))
    enc.set_option('precision', 1)
    enc.set_option('splitk_min_base_blocks', 0)
    z0, a0 = enc.forward(x), None
    a0 = enc.activation(1)
    enc.set_option('x3h_wide_min_blocks', 1)
    z1 = enc.forward(x)
    assert any('x3h_dma256' in l for l in enc.labels())
    assert np.array_equal(z0, z1) and np.array_equal(a0, enc.activation(1))
    enc.close()


def test_wide_block_tile_of_the_weights_to_registers_variant_is_bit_identical():
    """conv_igemm_f32_kernel<..., BREG, NW=4>: 128 x 256 block tiles (each wave 64 x 128) for layers whose padded
    Cout is a multiple of 256 -- same accumulation order per output, so bit-identical to the 128 x 128 tiles."""
    for shape, B, filters, bn in (((16, 16, 3), 3, [32, 256], False), ((24, 16, 3), 5, [64, 512], True)):
        cfg = EncoderConfig(shape, filters, [2, 2], 5, 128, bn)
        w = synth.make_weights(seed=5, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128, batch_norm=bn)
        x = synth.make_crops(B, seed=6, shape=cfg.shape)
        enc = eb.split_k_small_batches(eb.EmuEncoder(w, cfg))
        enc.set_option('splitk_min_base_blocks', 0)
        enc.set_option('igemm_breg_wide', 0)
        z0, a0 = enc.forward(x), None
        a0 = enc.activation(1)
        enc.set_option('igemm_breg_wide', 1)
        enc.set_option('igemm_breg_wide_min_blocks', 1)
        z1 = enc.forward(x)
        assert any('breg_n256' in l for l in enc.labels())
        assert np.array_equal(z0, z1) and np.array_equal(a0, enc.activation(1))
        z64 = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, bn)
        assert np.abs(z1 - z64).max() / np.abs(z64).max() < 5e-6
        enc.close()


@pytest.mark.parametrize('B', [1, 2, 3, 4, 5, 7, 8, 9])
def test_dense_layer_as_weight_streaming_gemv_for_tiny_batches(B):
    """B <= 8: dense_gemv_f32_kernel (+ the fixed-order chunk reduction) instead of a padded MFMA tile (B = 5 ... 8 on the
    8-row form of the block); B = 9 stays on the MFMA path, and so does B = 5 with dense_gemv_max_batch = 4.  Both against
    the fp64 oracle, and against each other within fp32 summation-order noise."""
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    w = synth.make_weights(seed=12, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128)
    x = synth.make_crops(B, seed=13, shape=cfg.shape)
    enc = eb.split_k_small_batches(eb.EmuEncoder(w, cfg))
    z = enc.forward(x)
    assert any('dense_gemv' in l for l in enc.labels()) == (B <= 8)
    z64 = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, False)
    assert np.abs(z - z64).max() / np.abs(z64).max() < 5e-6
    if B > 4:
        enc.set_option('dense_gemv_max_batch', 4)
        z4 = enc.forward(x)
        assert not any('dense_gemv' in l for l in enc.labels())
        assert np.abs(z - z4).max() / np.abs(z64).max() < 2e-6
        enc.set_option('dense_gemv_max_batch', 8)
    enc.set_option('dense_gemv', 0)
    z_mfma = enc.forward(x)
    assert not any('dense_gemv' in l for l in enc.labels())
    assert np.abs(z - z_mfma).max() / np.abs(z64).max() < 2e-6
    enc.close()


@pytest.mark.parametrize('B', [2, 3, 4, 6, 8])
def test_dense_gemv_two_level_finish_over_chunk_groups(B):
    """More than 32 chunks and B >= 2: the GEMV's chunk rows are added by a two-level tree that follows the two-level ticket
    (the last arriver of each of 16 chunk groups adds its group, the last group finisher adds the group rows + bias + BN).
    64 chunks here (8 x 8 x 128 features); every block order must give the same bits, and those must agree with the oracle
    and with the separate reduce launch within summation-order noise."""
    cfg = EncoderConfig((32, 32, 3), [32, 128], [2, 2], 5, 128, batch_norm=True)
    w = synth.make_weights(seed=21, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128, batch_norm=True)
    x = synth.make_crops(B, seed=22, shape=cfg.shape)
    z64 = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, True)
    outs = []
    try:
        for order in (0, 1, 2):
            eb.set_block_order(order)
            enc = eb.EmuEncoder(w, cfg)
            enc.set_option('detect_chain', 0)
            z = enc.forward(x)
            assert enc.labels()[-1].startswith('dense:dense_gemv_f32_ticket chunks=64'), enc.labels()
            outs.append(z.copy())
            enc.close()
    finally:
        eb.set_block_order(0)
    assert np.abs(outs[0] - z64).max() / np.abs(z64).max() < 5e-6
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    enc = eb.EmuEncoder(w, cfg)
    enc.set_option('gemv_ticket', 0)
    z2 = enc.forward(x)
    assert enc.labels()[-1] == 'dense:splitk_reduce', enc.labels()
    assert np.abs(outs[0] - z2).max() / np.abs(z64).max() < 2e-6
    enc.close()


@pytest.mark.parametrize('shape,filters,B', [
    ((40, 24, 3), [48, 32], 2),      # W*C = 72: dword staging, ragged last tile (240 px), Cout cut by the 32-channel wave tile
    ((24, 20, 1), [32, 32], 3),      # one input channel, W*C = 20, pl*C = 1 -> three lead floats
    ((16, 10, 3), [64, 32], 2),      # W*C = 30 is not a multiple of 4: element-wise staging only
    ((64, 32, 3), [32, 32], 1),      # 512 px: four full tiles walked by one block with its weights in registers
])
def test_first_layer_dword_and_elementwise_staging_agree_bitwise(shape, filters, B):
    """conv1 stages uint8 rows either as aligned dwords (W*C % 4 == 0) or element by element; float input
    always takes the element path.  All three must give the same bits -- staging only moves data."""
    cfg = EncoderConfig(shape, filters, [2, 2], 5, 16)
    w = synth.make_weights(seed=5, shape=cfg.shape, num_filter=filters, strides=cfg.strides, latent=16)
    x = synth.make_crops(B, seed=6, shape=cfg.shape)
    acts = []
    for vec4, f32_in in ((1, False), (0, False), (1, True)):
        enc = eb.split_k_small_batches(eb.EmuEncoder(w, cfg))
        enc.set_option('first_vec4', vec4)
        enc.set_option('first_max_tiles_per_block', 3)
        enc.forward(ref.input_to_float(x).astype(np.float32) if f32_in else x)
        assert 'conv_first_f32' in enc.labels()[0]
        acts.append(enc.activation(0).copy())
        enc.close()
    assert np.array_equal(acts[0], acts[1]) and np.array_equal(acts[0], acts[2])
    # ... and the per-detection form (one block per 32-pixel group instead of four groups per block; uint8 dword / element
    # staging and float input) runs the same MFMA steps in the same order: identical bits again
    for vec4, f32_in in ((1, False), (0, False), (1, True)):
        enc = eb.EmuEncoder(w, cfg)
        enc.set_option('first_vec4', vec4)
        enc.set_option('first_group_split_max_tiles', 1 << 20)
        enc.forward(ref.input_to_float(x).astype(np.float32) if f32_in else x)
        assert 'conv_first_f32_g4' in enc.labels()[0]
        assert np.array_equal(enc.activation(0), acts[0]), (vec4, f32_in)
        enc.close()
    _, want = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, False, return_activations=True)
    assert np.abs(acts[0] - want[0]).max() / np.abs(want[0]).max() < 5e-6


@pytest.mark.parametrize('precision,bn', [(0, False), (0, True), (1, False)])
def test_few_split_reduce_kernel_matches_the_grouped_one_bitwise(precision, bn):
    """Mid batches split K into <= 8 parts over a large output tile; those sums take the barrier-free float4
    reduce kernel (fp32 output, BN epilogue, and the two-plane f32x3h output).  Same bits as the grouped
    kernel, and both within fp32 roundoff of the oracle."""
    cfg = EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128, bn)
    w = synth.make_weights(seed=8, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128, batch_norm=bn)
    x = synth.make_crops(4, seed=9, shape=cfg.shape)
    outs = []
    for small in (1, 0):
        enc = eb.split_k_small_batches(eb.EmuEncoder(w, cfg))
        enc.set_option('precision', precision)
        enc.set_option('splitk_target_blocks', 8)          # conv2: 2 base blocks -> 4 splits over 256 x 64 outputs
        enc.set_option('reduce_small', small)
        z = enc.forward(x)
        assert any('splitk4' in l for l in enc.labels()), enc.labels()
        outs.append((enc.activation(1).copy(), z.copy()))
        enc.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    z64, acts = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, bn, return_activations=True)
    assert np.abs(outs[0][0] - acts[1]).max() / np.abs(acts[1]).max() < 5e-6
    assert np.abs(outs[0][1] - z64).max() / np.abs(z64).max() < 5e-6


@pytest.mark.parametrize('dtype,N,B', [('f32', 1000, 40), ('f32', 700, 130), ('f32', 520, 5), ('bf16', 1100, 70), ('bf16', 640, 129), ('bf16', 400, 9),
                                       ('bf16', 1100, 200), ('f32', 1000, 256), ('f32', 2300, 32), ('f32', 130, 17), ('bf16', 900, 31), ('f32', 1500, 20)])
def test_query_resident_scan_equals_the_tile_resident_kernels_bitwise(dtype, N, B):
    """Batches (B > 4, top-1, stride 1) keep the queries in registers and stream the codebook through LDS
    (codebook_scan_resident.h).  Same per-accumulator MFMA order as the tile-resident kernels -> the same bits:
    indices AND scores, ragged last tile, partially filled query groups, planted duplicates (first index wins)."""
    E = synth.make_codebook(N, 128, seed=3, planted_duplicates=6)
    rng = np.random.default_rng(4)
    z = rng.standard_normal((B, 128)).astype(np.float32) * rng.uniform(0.1, 20, (B, 1)).astype(np.float32)
    z[:3] = E[[35, 71, N - 1]] * 2.0
    cb = eb.EmuCodebook(E, dtype=dtype)
    idx_r, sc_r = cb.nn(z)                                  # AUTO -> query-resident kernel
    cb.set_mode(_lib.AAE_SCAN_MFMA)                         # forces the tile-resident MFMA kernel
    idx_t, sc_t = cb.nn(z)
    assert np.array_equal(idx_r, idx_t) and np.array_equal(sc_r, sc_t)
    # AUTO normalises the raw codes in the scan's prologue (norm added up in l2norm_pack_kernel's order); AUTO_PACKED runs
    # that kernel in front and reads its packed planes: the same fragments, the same bits
    cb.set_mode(_lib.AAE_SCAN_AUTO_PACKED)
    idx_p, sc_p = cb.nn(z)
    assert np.array_equal(idx_r, idx_p) and np.array_equal(sc_r, sc_p)
    # at most 32 queries: four waves per query group share the rows of a tile (fp32: 128-row tiles in two LDS images); AUTO_RH2 = two
    cb.set_mode(_lib.AAE_SCAN_AUTO_RH2)
    idx_2, sc_2 = cb.nn(z)
    assert np.array_equal(idx_r, idx_2) and np.array_equal(sc_r, sc_2)
    # opt-in: at most 32 queries answered inside the scan launch by the last row block to arrive (any arrival order)
    try:
        for order in (0, 1, 2):
            eb.set_block_order(order)
            cb.set_mode(_lib.AAE_SCAN_AUTO_FIN)
            idx_f, sc_f = cb.nn(z)
            assert np.array_equal(idx_r, idx_f) and np.array_equal(sc_r, sc_f), order
            idx_u, _ = cb.nn(z, 1, 36)                     # (upright without a compacted copy: the masked tile-resident kernels)
            cb.set_mode(_lib.AAE_SCAN_AUTO)
            assert np.array_equal(idx_u, cb.nn(z, 1, 36)[0])
    finally:
        eb.set_block_order(0)
        cb.set_mode(_lib.AAE_SCAN_AUTO)
    cs = cb.similarity(z)
    assert np.array_equal(idx_r[:, 0], np.argmax(cs, axis=1)) and np.array_equal(sc_r[:, 0], cs.max(axis=1))
    cb.close()


@pytest.mark.parametrize('dtype,N,B', [('f32', 460, 130), ('f32', 520, 5), ('bf16', 700, 70), ('bf16', 300, 9), ('bf16', 1100, 140), ('f32', 1300, 33)])
def test_topk_inside_the_query_resident_scan_equals_the_similarity_matrix_path(dtype, N, B):
    """top-k for 2 <= k <= 8 and B > 4 keeps K sorted (score, row) pairs per lane inside scan_resident_kernel<.., K> and merges
    the per-block lists (topk_merge_kernel) -- no [B][N] similarity matrix.  Canonical order (score descending, lower row
    first on ties) = what the matrix path (AAE_SCAN_MFMA: similarity + two-level selection) returns, bit for bit, scores
    included; planted duplicate rows (exact ties inside and across lanes / blocks), queries equal to rows, ragged last tile,
    partially filled query groups, every list size that is instantiated (K = 2, 4, 5, 8)."""
    E = synth.make_codebook(N, 128, seed=3, planted_duplicates=min(8, N // 72))
    E[N // 2 + 1] = E[N // 2] = E[7]                       # a three-way tie across row halves / blocks
    rng = np.random.default_rng(4)
    z = rng.standard_normal((B, 128)).astype(np.float32) * rng.uniform(0.1, 20, (B, 1)).astype(np.float32)
    z[:4] = E[[35, 71, N - 1, 7]] * 2.0
    cb = eb.EmuCodebook(E, dtype=dtype)
    cs = cb.similarity(z)
    for k in (2, 3, 5, 8) if B < 100 else (5, 8):
        cb.set_mode(_lib.AAE_SCAN_AUTO)
        ik, sk = cb.nn(z, topk=k)
        assert np.array_equal(ik, ref.topk_canonical(cs, k)), k
        assert np.array_equal(sk, np.take_along_axis(cs, ik, axis=1)), k
        cb.set_mode(_lib.AAE_SCAN_MFMA)
        im, sm = cb.nn(z, topk=k)
        assert np.array_equal(ik, im) and np.array_equal(sk, sm), k
        # the lists drop candidates below the bound the blocks publish to each other (a lower bound of the final k-th best,
        # ties kept): the answers must not depend on it, nor on the order in which the blocks run and publish
        cb.set_mode(_lib.AAE_SCAN_AUTO_NO_PRUNE)
        iu, su = cb.nn(z, topk=k)
        assert np.array_equal(ik, iu) and np.array_equal(sk, su), k
        for order in (1, 2):
            eb.set_block_order(order)
            cb.set_mode(_lib.AAE_SCAN_AUTO)
            io, so = cb.nn(z, topk=k)
            eb.set_block_order(0)
            assert np.array_equal(ik, io) and np.array_equal(sk, so), (k, order)
    cb.close()


@pytest.mark.parametrize('dtype,N,B', [('f32', 36 * 21 + 5, 3), ('f32', 36 * 40, 40), ('bf16', 36 * 30 + 17, 2), ('bf16', 36 * 35, 70)])
def test_upright_search_on_the_compacted_copy_equals_the_masked_scan(dtype, N, B):
    """col_stride = 36: after aae_codebook_prepare_upright the scan runs over the every-36th-row copy (N/36 rows) and the
    row id is scaled back; before it, the full scan masks 35/36 of its candidates.  Same rows, same arithmetic, same
    tie rule -> same bits; aae_codebook_update refreshes the copy."""
    E = synth.make_codebook(N, 128, seed=13, planted_duplicates=4)
    E[72] = E[0]                                            # duplicate among the upright rows: the lower one must win
    rng = np.random.default_rng(14)
    z = rng.standard_normal((B, 128)).astype(np.float32)
    z[0] = E[72] * 3.0
    cb = eb.EmuCodebook(E, dtype=dtype)
    idx_m, sc_m = cb.nn(z, 1, 36)                           # masked full scan
    cb.prepare_upright(36)
    idx_c, sc_c = cb.nn(z, 1, 36)                           # compacted copy
    assert np.array_equal(idx_m, idx_c) and np.array_equal(sc_m, sc_c)
    assert idx_c[0, 0] == 0 and np.all(idx_c % 36 == 0)
    cs = cb.similarity(z)
    assert np.array_equal(idx_c[:, 0], ref.nearest_indices_reference(cs, 1, upright=True, num_cyclo=36))
    idx_p, _ = cb.nn(z, 1, 1)                               # the plain search is untouched
    assert np.array_equal(idx_p[:, 0], np.argmax(cs, axis=1))
    cb.close()


# ---- small-batch path: wave-split-K igemm + in-launch ticketed reductions (conv_wavek_f32.h, dense_gemv ticket) ----
@pytest.mark.parametrize('order', [0, 1, 2])
@pytest.mark.parametrize('waves,depth', [(4, 3), (4, 2), (8, 2)])
def test_wave_split_k_igemm_narrow_tiles_and_ticketed_gemv(waves, depth, order):
    """B = 3 of a 2-layer net: conv2 has M = 48 rows (one partial 64-row tile), 25 slabs over 3 blocks x `waves`
    waves (some waves get 2 slabs, some 3: ragged ring drain), 64 x 32 wave tiles; the dense layer is the GEMV
    whose chunk sums are finished by the last block to arrive.  Any block order must give the same bits."""
    eb.set_block_order(order)
    try:
        cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
        labels = _run(cfg, 3, 1, wavek=1, options={'wavek_waves': waves, 'wavek_depth': depth, 'wavek_tiny_max_tiles': 0})
        assert 'conv_wavek_f32_64x32_w%d_d%d_g%d ' % (waves, depth, 3 if waves == 4 else 1) in labels[1], labels
        assert labels[-1].startswith('dense:dense_gemv_f32_ticket') and len(labels) == 3, labels
    finally:
        eb.set_block_order(0)


def test_wave_split_k_results_do_not_depend_on_block_arrival_order():
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128, True)
    w = synth.make_weights(seed=3, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128, batch_norm=True)
    x = synth.make_crops(2, seed=4, shape=cfg.shape)
    outs = []
    for order in (0, 1, 2):
        eb.set_block_order(order)
        try:
            enc = eb.EmuEncoder(w, cfg)
            enc.set_option('detect_chain', 0)              # the stand-alone ticketed launches
            z = enc.forward(x)
            outs.append((z.copy(), enc.activation(1)))
            assert any('wavek' in l for l in enc.labels()) and enc.labels()[-1].startswith('dense:dense_gemv_f32_ticket')
            enc.close()
        finally:
            eb.set_block_order(0)
    for z, a in outs[1:]:
        assert np.array_equal(z, outs[0][0]) and np.array_equal(a, outs[0][1])


@pytest.mark.parametrize('narrow', [0, 16])
def test_wave_split_k_igemm_wide_tiles_several_m_tiles_and_dense(narrow):
    """conv2: M = 5*8*8 = 320 rows = 5 M tiles of 64 (no partial), N = 64, 25 slabs; narrow = 0 forces 64 x 64 wave
    tiles (5 tiles, K split over 5 blocks x 4 waves -> waves with a single slab), 16 keeps the 64 x 32 form.
    B = 5 with the GEMV held to B <= 4: the dense layer (M = 5, K = 4096 = 128 slabs) also runs on the wave-split-K kernel."""
    cfg = EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128, True)
    labels = _run(cfg, 5, 61, wavek=1, options={'wavek_narrow_max_tiles': narrow, 'wavek_tiny_max_tiles': 0, 'dense_gemv_max_batch': 4})
    assert ('conv_wavek_f32_64x64' if narrow == 0 else 'conv_wavek_f32_64x32') in labels[1], labels
    assert labels[2].startswith('dense:conv_wavek_f32_64x') and len(labels) == 3, labels


def test_wave_split_k_without_cross_block_split_and_with_cout_padding():
    """96 output channels (CoutPad 128: the second 64-column tile is half padding) and a 3 x 3 kernel: 9 slabs are
    too few to split across blocks (gsplits == 1) -- the epilogue runs straight from the LDS sum, no tickets."""
    cfg = EncoderConfig((32, 32, 3), [32, 96], [2, 1], 3, 64)
    labels = _run(cfg, 2, 71, wavek=1, options={'wavek_narrow_max_tiles': 0, 'wavek_tiny_max_tiles': 0})
    assert 'conv_wavek_f32_64x64' in labels[1] and '_g1 ' in labels[1], labels


@pytest.mark.parametrize('order', [0, 2])
@pytest.mark.parametrize('B,stride', [(1, 1), (3, 1), (2, 36), (6, 1)])
def test_fused_encode_nn_prepares_every_ticket_and_equals_the_two_calls(B, stride, order):
    """aae_encode_nn: conv1's block 0 installs the nonces of the later ticketed launches (wave-split-K tiles, GEMV
    chunks, the scan's block partials); the answers must be those of aae_encoder_forward + aae_codebook_nn bit for bit,
    with prepared tickets, unprepared ones (ticket_prep = 0) and the two-launch scan."""
    eb.set_block_order(order)
    try:
        cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
        w = synth.make_weights(seed=21, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128)
        x = synth.make_crops(B, seed=22, shape=cfg.shape)
        N = 36 * 11 + 5
        E = synth.make_codebook(N, 128, seed=7, planted_duplicates=11)
        enc, cb = eb.EmuEncoder(w, cfg), eb.EmuCodebook(E)
        if stride > 1:
            cb.prepare_upright(stride)
        enc.set_option('detect_chain', 0)
        z0 = enc.forward(x)
        cb.set_mode(_lib.AAE_SCAN_STREAM_2L)
        i0, s0 = cb.nn(z0, 1, stride)
        cb.set_mode(_lib.AAE_SCAN_AUTO)
        for chain in (0, 1):                          # six stand-alone launches / conv1 + the persistent launch (B <= 4)
            enc.set_option('detect_chain', chain)
            for prep in (1, 0):
                enc.set_option('ticket_prep', prep)
                z1, i1, s1 = eb.encode_nn(enc, cb, x, stride)
                assert np.array_equal(z1, z0) and np.array_equal(i1, i0) and np.array_equal(s1, s0), (chain, prep)
        cb.set_mode(_lib.AAE_SCAN_STREAM)          # single-launch scan without preparation: the install path
        i2, s2 = cb.nn(z0, 1, stride)
        assert np.array_equal(i2, i0) and np.array_equal(s2, s0)
        cb.set_mode(_lib.AAE_SCAN_AUTO_FIN)        # (B > 4: the query-resident scan answering inside its launch, stand-alone and fused)
        i3, s3 = cb.nn(z0, 1, stride)
        z4, i4, s4 = eb.encode_nn(enc, cb, x, stride)
        assert np.array_equal(i3, i0) and np.array_equal(s3, s0) and np.array_equal(z4, z0) and np.array_equal(i4, i0) and np.array_equal(s4, s0)
        cb.set_mode(_lib.AAE_SCAN_AUTO)
        cs = cb.similarity(z0)
        want = ref.nearest_indices_reference(cs, 1, upright=True, num_cyclo=36) if stride > 1 else np.argmax(cs, axis=1)
        assert np.array_equal(i0[:, 0], want)
        enc.close()
        cb.close()
    finally:
        eb.set_block_order(0)


# (batch, filters of the four conv layers, planner thresholds) that make plan_wavek choose each instantiated shape sequence of the
# persistent launch on a 16 x 16 input: a layer of <= 64 rows has CoutPad / 64 tiles of 64 x 64, so the filter counts steer the shapes
_CHAIN_CASES = {
    '000': (1, [32, 64, 64, 64], {}),
    '100': (2, [32, 256, 64, 64], {'wavek_tiny_max_tiles': 2, 'wavek_narrow_max_tiles': 128}),
    '010': (3, [32, 64, 256, 64], {'wavek_tiny_max_tiles': 2, 'wavek_narrow_max_tiles': 128}),
    '210': (4, [32, 384, 256, 64], {'wavek_tiny_max_tiles': 2, 'wavek_narrow_max_tiles': 4}),
}


@pytest.mark.parametrize('order', [0, 1, 2])
@pytest.mark.parametrize('shapes', ['000', '100', '010', '210'])
def test_persistent_detect_chain_equals_the_stand_alone_launches(shapes, order):
    """detect_chain.h: conv2 ... conv4, dense and the scan as phases of ONE resident launch (grid barriers between them, operands
    of the next phase requested while a barrier closes).  Same work items, same summation orders: every layer output, the latents,
    the index and the score must be the bits of the six stand-alone launches -- for each instantiated (batch class, wave-tile
    shape sequence), grids of 1, 2 and 3 resident blocks (work items walked grid-stride; the K splits of one tile on different
    blocks or on the same one), any interleaving of the blocks, workspaces full of garbage, with batch-norm."""
    B, filters, opts = _CHAIN_CASES[shapes]
    if shapes == '210' and order == 1:
        pytest.skip('the largest case runs in ascending and scrambled block order only (CPU suite time)')
    eb.set_block_order(order)
    try:
        cfg = EncoderConfig((16, 16, 3), filters, [2, 2, 2, 1], 5, 128, True)
        w = synth.make_weights(seed=31, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128, batch_norm=True)
        E = synth.make_codebook(36 * 9 + 7, 128, seed=8, planted_duplicates=9)
        enc, cb = eb.EmuEncoder(w, cfg), eb.EmuCodebook(E)
        enc.set_option('wavek_balance', 0)
        enc.set_option('planner_cost_batch3', 0)     # (the shape sequences below are steered through the thresholds)
        for k, v in opts.items():
            enc.set_option(k, v)
        x = synth.make_crops(B, seed=40 + B, shape=cfg.shape)
        enc.set_option('detect_chain', 0)
        z0, i0, s0 = eb.encode_nn(enc, cb, x, 1)
        acts0 = [enc.activation(i) for i in range(4)]
        names = {'0': '32x32', '1': '64x32', '2': '64x64'}
        assert all(('conv_wavek_f32_%s_' % names[c]) in l for c, l in zip(shapes, enc.labels()[1:4])), enc.labels()
        enc.set_option('detect_chain', 1)
        for blocks in ((3, 2, 1) if order == 0 else (3,)):
            enc.set_option('detect_chain_blocks', blocks)
            z1, i1, s1 = eb.encode_nn(enc, cb, x, 1)
            labels = enc.labels()
            assert len(labels) == 2 and labels[1] == 'chain:detect_chain_f32 B=%d blocks=%d shapes=%s phases=conv2..conv4+dense+scan' % (B, blocks, shapes), labels
            assert np.array_equal(z1, z0) and np.array_equal(i1, i0) and np.array_equal(s1, s0), (B, blocks)
            for i in range(4):
                assert np.array_equal(enc.activation(i), acts0[i]), (B, blocks, i)
        zf = enc.forward(x)                                      # encoder alone: the same launch without its scan phase
        assert np.array_equal(zf, z0) and enc.labels()[1].endswith('+dense'), enc.labels()
        enc.close()
        cb.close()
    finally:
        eb.set_block_order(0)


def test_detect_chain_falls_back_where_it_does_not_apply():
    """B > 4, other depths than four conv layers, shape sequences without a kernel, compact workspaces, an un-prepared upright
    stride (masked scan) and split precision keep the stand-alone launches."""
    B, filters, opts = _CHAIN_CASES['000']
    cfg = EncoderConfig((16, 16, 3), filters, [2, 2, 2, 1], 5, 128)
    w = synth.make_weights(seed=21, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128)
    E = synth.make_codebook(36 * 5, 128, seed=7, planted_duplicates=3)
    enc, cb = eb.EmuEncoder(w, cfg), eb.EmuCodebook(E)
    x = synth.make_crops(5, seed=2, shape=cfg.shape)
    eb.encode_nn(enc, cb, x[:1], 1)
    assert not any('chain' in l for l in enc.labels())           # opt-in: off unless asked for
    enc.set_option('detect_chain', 1)
    eb.encode_nn(enc, cb, x[:1], 1)
    assert enc.labels()[1].endswith('+dense+scan'), enc.labels()
    eb.encode_nn(enc, cb, x, 1)
    assert not any('chain' in l for l in enc.labels())           # B = 5
    z, i, s = eb.encode_nn(enc, cb, x[:1], 36)               # no compacted copy for stride 36: chain without the scan phase, masked scan launch
    assert enc.labels()[1].endswith('+dense'), enc.labels()
    cs = cb.similarity(z)
    assert np.array_equal(i[:, 0], ref.nearest_indices_reference(cs, 1, upright=True, num_cyclo=36))
    eb.encode_nn(enc, cb, x[:2], 1)                          # B = 2 with shapes 000: no kernel for that pair
    assert not any('chain' in l for l in enc.labels())
    enc.set_option('compact_workspace', 1)
    z2 = enc.forward(x[:1])
    assert not any('chain' in l for l in enc.labels()) and np.array_equal(z2, z)
    enc.set_option('compact_workspace', 0)
    enc.set_option('precision', 1)
    enc.forward(x[:1])
    assert not any('chain' in l for l in enc.labels())
    enc.close()
    cb.close()
    cfg3 = EncoderConfig((16, 16, 3), [32, 64, 64], [2, 2, 1], 5, 128)      # three conv layers
    enc3 = eb.EmuEncoder(synth.make_weights(seed=5, shape=cfg3.shape, num_filter=cfg3.num_filter, strides=cfg3.strides, latent=128), cfg3)
    enc3.set_option('detect_chain', 1)
    enc3.forward(synth.make_crops(1, seed=3, shape=cfg3.shape))
    assert not any('chain' in l for l in enc3.labels())
    enc3.close()


def test_planner_by_cost_picks_a_family_and_a_tile_shape_per_layer():
    """B >= 5: plan_wavek estimates the time of the 128-row igemm and of the three wave-split-K tile shapes for every conv layer and
    takes the cheapest (fitted on MI355X, tools/sweep_planner.py); whatever it picks must compute the layer: every activation and
    the latents against the oracle, for batch sizes on both sides of the families' block-count steps, and identically with the
    threshold planner's kernels where both pick the same."""
    cfg = EncoderConfig((32, 32, 3), [32, 64, 64], [2, 2, 2], 5, 128)
    seen = set()
    for B in (5, 9, 16):
        # ("wavek_target_blocks" is the model's CU count: 8 instead of 256 puts this small network's layers on both sides of the round steps)
        labels = _run(cfg, B, 90 + B, wavek=1, options={'planner_cost_model': 1, 'wavek_target_blocks': 8})
        assert all(('conv_wavek_f32_' in l) or ('conv_igemm_f32' in l) or ('splitk_reduce' in l) for l in labels[1:-1]), labels
        seen.update(l.split(':')[1].split(' ')[0].split('_w4')[0].split('_splitk')[0] for l in labels[1:-1])
    assert len(seen) >= 2, seen                                  # more than one kernel form was chosen over these batches
    # B <= 4 stays with the hand-tuned per-detection plan whatever the option says
    a = _run(cfg, 3, 95, wavek=1, options={'planner_cost_model': 1})
    b = _run(cfg, 3, 95, wavek=1, options={'planner_cost_model': 0})
    assert a == b


def test_compact_workspace_alternates_two_activation_buffers():
    """Option compact_workspace: layer i writes buffer i % 2 -- same latents, smaller workspace, earlier layers not inspectable."""
    cfg = EncoderConfig((32, 32, 3), [32, 64, 32], [2, 2, 1], 5, 64)
    w = synth.make_weights(seed=9, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=64)
    x = synth.make_crops(3, seed=10, shape=cfg.shape)
    enc = eb.EmuEncoder(w, cfg)
    n_full = enc.L.aae_encoder_workspace_bytes(enc.h, 3)
    z0 = enc.forward(x)
    last = enc.activation(2)
    enc.set_option('compact_workspace', 1)
    assert enc.L.aae_encoder_workspace_bytes(enc.h, 3) < n_full
    z1 = enc.forward(x)
    assert np.array_equal(z0, z1) and np.array_equal(enc.activation(2), last)
    with pytest.raises(ValueError, match='overwritten'):
        enc.activation(0)
    enc.close()


@pytest.mark.parametrize('B', [1, 3])
def test_wave_split_k_igemm_32x32_wave_tiles(B):
    """Option wavek_tiny_max_tiles: 32 x 32 wave tiles (one accumulator per wave) -- four times the tiles, so K is split across
    fewer blocks.  conv2 here: M = B*16 rows (a partial 32-row tile at B = 3: 48 rows), Cout 96 -> CoutPad 128 = four column tiles."""
    cfg = EncoderConfig((16, 16, 3), [32, 96], [2, 2], 5, 128, True)
    labels = _run(cfg, B, 81, wavek=1, options={'wavek_tiny_max_tiles': 64})
    assert 'conv_wavek_f32_32x32_w4_d2' in labels[1], labels


@pytest.mark.parametrize('order', [0, 2])
@pytest.mark.parametrize('B', [1, 3])
def test_wave_split_k_igemm_32x32_wave_tiles_on_eight_waves(B, order):
    """Option wavek_tiny_waves = 8: the 32 x 32 wave tiles with two waves per SIMD (K cut eight ways inside a block; the 256
    float4 pieces of a tile are finished by the first four waves).  With and without a cross-block K split, any block order."""
    cfg = EncoderConfig((16, 16, 3), [32, 96], [2, 2], 5, 128, True)
    eb.set_block_order(order)
    try:
        labels = _run(cfg, B, 81, wavek=1, options={'wavek_tiny_max_tiles': 64, 'wavek_tiny_waves': 8})
        assert 'conv_wavek_f32_32x32_w8_d2' in labels[1], labels
        labels = _run(cfg, B, 81, wavek=1, options={'wavek_tiny_max_tiles': 64, 'wavek_tiny_waves': 8, 'wavek_target_blocks': 2})
        assert 'conv_wavek_f32_32x32_w8_d2_g1 ' in labels[1], labels
    finally:
        eb.set_block_order(0)


@pytest.mark.parametrize('narrow', [0])
def test_spread_load_schedule_is_bit_identical_to_the_burst(narrow):
    """Option wavek_spread (default on): 64 x 64 wave tiles (four accumulators) issue the operand loads of the next slab one per
    q-step between the MFMAs of the current one instead of as a burst in front of them.  Same MFMA order per accumulator:
    identical bits, with ragged K ranges and a cross-block split."""
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    w = synth.make_weights(seed=5, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128)
    x = synth.make_crops(5, seed=6, shape=cfg.shape)
    outs = []
    for spread in (1, 0):
        enc = eb.EmuEncoder(w, cfg)
        for k, v in (('detect_chain', 0), ('planner_cost_model', 0), ('wavek_tiny_max_tiles', 0), ('wavek_narrow_max_tiles', narrow), ('wavek_spread', spread)):
            enc.set_option(k, v)
        z = enc.forward(x)
        assert ('conv_wavek_f32_64x32_w4_d2' if narrow else 'conv_wavek_f32_64x64_w4_d2') in enc.labels()[1], enc.labels()
        outs.append((z.copy(), enc.activation(1).copy()))
        enc.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    z64 = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, False)
    assert np.abs(outs[0][0] - z64).max() / np.abs(z64).max() < 5e-6


@pytest.mark.parametrize('B', [1, 3])
def test_spread_schedule_of_the_32x32_tiles_with_two_accumulator_chains(B):
    """wavek_spread bit 1: the 32 x 32 wave tile takes its odd q-steps into a second accumulator (consecutive MFMAs independent,
    the next slab's loads between them); the two chains are added behind the K loop -- another fixed summation order, so the
    check is the oracle's tolerance, layer by layer, with a partial M tile and a cross-block split."""
    cfg = EncoderConfig((16, 16, 3), [32, 96], [2, 2], 5, 128, True)
    labels = _run(cfg, B, 81, wavek=1, options={'wavek_tiny_max_tiles': 64, 'wavek_spread': 3})
    assert 'conv_wavek_f32_32x32_w4_d2' in labels[1], labels
    labels = _run(cfg, B, 81, wavek=1, options={'wavek_tiny_max_tiles': 64, 'wavek_spread': 3, 'wavek_target_blocks': 2})
    assert 'conv_wavek_f32_32x32_w4_d2_g1 ' in labels[1], labels


@pytest.mark.parametrize('tiny', [64])
def test_eight_wave_pingpong_schedule_is_bit_identical_to_the_free_running_loop(tiny):
    """8-wave blocks: the two waves of a SIMD alternate load issue and MFMAs with a block barrier between the half-steps
    (option wavek_pingpong; measured slower on MI355X, default off).  Same per-wave fma chains as the free-running loop: identical bits, for ragged K
    ranges (25 slabs over 3 blocks x 8 waves: some waves get one slab, some two); 32 x 32 wave tiles (the only form that keeps the schedule)."""
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    w = synth.make_weights(seed=5, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128)
    x = synth.make_crops(3, seed=6, shape=cfg.shape)
    outs = []
    for pp in (1, 0):
        enc = eb.EmuEncoder(w, cfg)
        for k, v in (('detect_chain', 0), ('wavek_waves', 8), ('wavek_tiny_waves', 8), ('wavek_tiny_max_tiles', tiny), ('wavek_pingpong', pp)):
            enc.set_option(k, v)
        z = enc.forward(x)
        assert '_w8_d2' in enc.labels()[1], enc.labels()
        outs.append((z.copy(), enc.activation(1).copy()))
        enc.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    z64 = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, False)
    assert np.abs(outs[0][0] - z64).max() / np.abs(z64).max() < 5e-6


def test_f32x3h_range_flag_on_the_emulated_kernels():
    """aae_encoder_x3h_saturated: every kernel that writes fp16 (hi, lo) activation pairs raises the sticky flag when a value
    leaves the range the pair carries (|x * 2^4| < 65504); asking returns it and clears it; exact-fp32 mode never raises it."""
    import ctypes
    cfg = EncoderConfig((16, 16, 3), [32, 64], [2, 2], 5, 128)
    w = synth.make_weights(seed=5, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128)
    x = synth.make_crops(2, seed=6, shape=cfg.shape)
    _, acts = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, False, return_activations=True)

    def flag(enc):
        f = ctypes.c_int(7)
        assert enc.L.aae_encoder_x3h_saturated(enc.h, ctypes.byref(f), None) == 0
        return f.value

    for layer, name in ((0, 'conv2d'), (1, 'conv2d_1')):            # conv1's plane epilogue, then the x3h igemm's (via its split-K reduce)
        for frac, want in ((0.9, 0), (1.3, 1)):
            f = frac * 4094.0 / float(np.abs(acts[layer]).max())
            ws = dict(w)
            ws[name + '/kernel'], ws[name + '/bias'] = w[name + '/kernel'] * np.float32(f), w[name + '/bias'] * np.float32(f)
            enc = eb.EmuEncoder(ws, cfg)
            enc.forward(x)
            assert flag(enc) == 0                                    # exact fp32: no pairs, no flag
            enc.set_option('precision', 1)
            z = enc.forward(x)
            assert flag(enc) == want and flag(enc) == 0, (layer, frac)
            # the per-forward form: every f32x3h forward owns a slot; one poll reads (and clears) several of them
            slots = []
            for _ in range(3):
                enc.forward(x)
                slots.append(enc.L.aae_encoder_x3h_last_slot())
            assert len(set(slots)) == 3 and min(slots) >= 0
            arr, out = (ctypes.c_int * 3)(*slots), (ctypes.c_int * 3)()
            assert enc.L.aae_encoder_x3h_poll(enc.h, arr, 3, out, None) == 0 and list(out) == [want] * 3
            assert enc.L.aae_encoder_x3h_poll(enc.h, arr, 3, out, None) == 0 and list(out) == [0, 0, 0]
            enc.set_option('precision', 0)
            enc.forward(x)
            assert enc.L.aae_encoder_x3h_last_slot() == -1
            if not want:
                z64 = ref.encoder_forward_np(ref.input_to_float(x), ws, cfg.strides, False)
                assert np.abs(z - z64).max() / np.abs(z64).max() < 5e-6
            enc.close()


def test_precision_2_uses_split_precision_only_for_batches_that_fill_the_chip():
    """Encoder option precision = 2: f32x3h where it is faster.  A batch whose first implicit-GEMM layer has fewer than
    x3h_min_tiles 64 x 64 output tiles runs the exact fp32 path (bit-identical to precision 0, layer outputs are fp32), a
    larger one runs f32x3h (bit-identical to precision 1, layer outputs are (hi, lo) pairs); the C ABI tells which."""
    cfg = EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128)
    w = synth.make_weights(seed=61, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128)
    enc = eb.EmuEncoder(w, cfg)
    enc.set_option('x3h_min_tiles', 8)                       # conv2: 64 output pixels per crop x 128 padded channels = two 64 x 64 tiles per crop
    want = {}
    for B in (2, 6):
        x = synth.make_crops(B, seed=62 + B, shape=cfg.shape)
        for prec in (0, 1):
            enc.set_option('precision', prec)
            want[B, prec] = (enc.forward(x), enc.activation(0), enc.activation(1))
        enc.set_option('precision', 2)
        split = enc.L.aae_encoder_split_precision_for_batch(enc.h, B)
        assert split == (1 if B >= 4 else 0)
        got = (enc.forward(x), enc.activation(0), enc.activation(1))
        for g, wv in zip(got, want[B, split]):
            assert np.array_equal(g, wv), (B, split)
    assert not np.array_equal(want[6, 0][0], want[6, 1][0])  # (the two modes do differ in the last bits)
    enc.close()


@pytest.mark.parametrize('shape,B,filters,bn', [((32, 32, 3), 5, [32, 256], False), ((24, 16, 3), 7, [64, 512], True)])
def test_f32x3h_256x256_tile_kernel_is_bit_identical(shape, B, filters, bn):
    """conv_igemm_x3h_wide_kernel: 256 x 256 block tiles, 8 waves of 64 x 128, B fragments refreshed in place, A fragments
    double-buffered, one barrier per slab -- same k-step and product order per accumulator as the 128 x 128 kernel, so the same
    bits; M = 320 / 168 rows (a full and a partial 256-row tile / one partial tile), Cout 256 and 512 (one and two N tiles),
    the dense layer consuming the (hi, lo) pairs the wide kernel wrote."""
    cfg = EncoderConfig(shape, filters, [2] * len(filters), 5, 128, bn)
    w = synth.make_weights(seed=15, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128, batch_norm=bn)
    x = synth.make_crops(B, seed=16, shape=cfg.shape)
    enc = eb.split_k_small_batches(eb.EmuEncoder(w, cfg))
    enc.set_option('precision', 1)
    enc.set_option('splitk_min_base_blocks', 0)
    enc.set_option('x3h_wide256', 0)
    z0 = enc.forward(x)
    a0 = [enc.activation(i) for i in range(1, len(filters))]
    assert not any('wide256' in l for l in enc.labels())
    enc.set_option('x3h_wide256', 1)
    enc.set_option('x3h_wide256_min_blocks', 1)
    z1 = enc.forward(x)
    assert any('x3h_wide256' in l for l in enc.labels()), enc.labels()
    for i, a in enumerate(a0):
        assert np.array_equal(a, enc.activation(i + 1)), 'layer %d' % (i + 1)
    assert np.array_equal(z0, z1)
    z64 = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, bn)
    assert np.abs(z1 - z64).max() / np.abs(z64).max() < 2e-5       # the latent tolerance of the GPU parity tests
    enc.close()


@pytest.mark.parametrize('order', [0, 1, 2])
@pytest.mark.parametrize('shape_opts,tail,g', [({'wavek_narrow_max_tiles': 0, 'wavek_tiny_max_tiles': 0}, 3, 2),      # 64 x 64 wave tiles
                                               ({'wavek_narrow_max_tiles': 1 << 20, 'wavek_tiny_max_tiles': 0}, 4, 3),  # 64 x 32
                                               ({'wavek_narrow_max_tiles': 1 << 20, 'wavek_tiny_max_tiles': 1 << 20}, 7, 3)])   # 32 x 32
def test_wave_split_k_tail_tiles_cut_in_k(shape_opts, tail, g, order):
    """Mid-size batches whose tile count does not fill the last round of blocks: the LAST `tail` tiles of a layer are cut `g` ways
    in K (their partials meet through the ticket of the tile, indexed from the first tail tile), the tiles in front of them stay
    whole -- one launch, head and tail remapped to the XCDs separately.  Every layer against the fp64 oracle (inside _run), the
    label names the cut, any block order gives the same bits, and the untouched head tiles have the bits of the un-cut layer."""
    cfg = EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128, True)
    opts = dict(shape_opts, wavek_max_tiles=1 << 12, wavek_target_blocks=1, wavek_force_tail_tiles=tail, wavek_force_tail_g=g, dense_gemv_max_batch=4)
    eb.set_block_order(order)
    try:
        labels = _run(cfg, 6, 91, wavek=1, options=opts)
        assert '_g1t%dx%d ' % (tail, g) in labels[1], labels
    finally:
        eb.set_block_order(0)


def test_wave_split_k_tail_split_leaves_the_head_tiles_bitwise_alone():
    cfg = EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128, True)
    w = synth.make_weights(seed=5, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides, latent=128, batch_norm=True)
    x = synth.make_crops(6, seed=6, shape=cfg.shape)
    outs = []
    for tail in (0, 5):
        enc = eb.EmuEncoder(w, cfg)
        for k, v in {'planner_cost_model': 0, 'wavek_max_tiles': 1 << 12, 'wavek_target_blocks': 1, 'wavek_narrow_max_tiles': 0, 'wavek_tiny_max_tiles': 0,
                     'wavek_force_tail_tiles': tail, 'wavek_force_tail_g': 3}.items():
            enc.set_option(k, v)
        enc.forward(x)
        assert ('t5x3' in enc.labels()[1]) == (tail == 5), enc.labels()
        outs.append(enc.activation(1).copy())                  # conv2 output [B, Ho, Wo, C]
        enc.close()
    a, b = (o.reshape(-1, o.shape[-1]) for o in outs)          # rows = output pixels; 64-row M tiles, 64-column N tiles
    tiles_m = -(-a.shape[0] // 64)
    tiles = tiles_m * (a.shape[1] // 64)
    same = [bool(np.array_equal(a[(t % tiles_m) * 64:(t % tiles_m + 1) * 64, (t // tiles_m) * 64:(t // tiles_m + 1) * 64],
                                b[(t % tiles_m) * 64:(t % tiles_m + 1) * 64, (t // tiles_m) * 64:(t // tiles_m + 1) * 64])) for t in range(tiles)]
    assert all(same[:tiles - 5]), same                        # head tiles: the same fma chains
    assert np.abs(a - b).max() / np.abs(a).max() < 2e-6       # tail tiles: another (fixed) summation order
