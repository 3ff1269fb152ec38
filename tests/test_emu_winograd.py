"""Polyphase Winograd convolution (csrc/kernels/conv_winograd_f32.h, encoder option "winograd") on the CPU fiber emulator against
the fp64 oracle: the four phase shapes (3 x 3, 3 x 2, 2 x 3 with the split along the columns, 2 x 2 with the uneven point split),
both block geometries (16 x 16-pixel regions of one image / four 8 x 8 images), window halo = 'SAME' padding, several 32-channel
stages, several blocks per image, ragged image groups, the accumulating output modes and the BN epilogue."""
import numpy as np
import pytest

import emu_backend as eb
from augmentedautoencoder_amd.weights import EncoderConfig
from oracle import reference_cpu as ref
from oracle import synth


def _run(cfg, B, seed, options=None, mode=1):
    w = synth.make_weights(seed=seed, shape=cfg.shape, num_filter=cfg.num_filter, strides=cfg.strides,
                           latent=cfg.latent_space_size, batch_norm=cfg.batch_norm, kernel_size=cfg.kernel_size)
    x = synth.make_crops(B, seed=seed + 1, shape=cfg.shape)
    enc = eb.EmuEncoder(w, cfg)
    enc.set_option('winograd_min_batch', 1)
    enc.set_option('winograd_min_blocks', 1)     # (the product rule: launches of at least three quarters of a round of blocks)
    enc.set_option('winograd', mode)
    for k, v in (options or {}).items():
        enc.set_option(k, v)
    z = enc.forward(x)
    z64, acts = ref.encoder_forward_np(ref.input_to_float(x), w, cfg.strides, cfg.batch_norm, return_activations=True)
    errs = []
    for i, a in enumerate(acts):
        err = np.abs(enc.activation(i) - a).max() / max(np.abs(a).max(), 1e-9)
        errs.append(err)
        assert err < 5e-6, 'layer %d rel err %.2e (%s)' % (i, err, enc.labels())
    assert np.abs(z - z64).max() / np.abs(z64).max() < 5e-6
    labels = enc.labels()
    enc.close()
    return labels, errs


# mode 1: one launch per layer (the four phases inside the block, 16-channel stages, outputs added up in LDS); mode 2: one launch per phase
# (32-channel stages, the phases add up in the output buffer)
@pytest.mark.parametrize('mode', [1, 2])
def test_one_region_per_image_one_stage(mode):
    # conv2: 32 x 32 x 32 -> 16 x 16 x 64: geometry 0, one block per image, one 32-channel stage (two 16-channel ones)
    labels, _ = _run(EncoderConfig((64, 64, 3), [32, 64], [2, 2], 5, 128), 2, 11, mode=mode)
    wino = [l for l in labels if 'conv_wino_f32' in l]
    assert all(l.startswith('conv2') for l in wino)
    if mode == 1:
        assert len(wino) == 1 and 'layer' in wino[0]
    else:
        assert [l.split('phase ')[1][:2] for l in wino] == ['11', '10', '01', '00']


@pytest.mark.parametrize('mode,wide', [(1, 0), (1, 1), (2, 0)])
def test_several_regions_per_image_two_stages_two_column_blocks(mode, wide):
    # conv2: 64 x 64 x 64 -> 32 x 32 x 128: 2 x 2 regions per image, two 32-channel stages, two 64-column blocks
    labels, _ = _run(EncoderConfig((128, 128, 3), [64, 128], [2, 2], 5, 64), 1, 23, mode=mode, options={'winograd_wide': wide})
    assert sum('conv_wino_f32' in l for l in labels) == (1 if mode == 1 else 4)


@pytest.mark.parametrize('B,mode,wide', [(1, 1, 0), (5, 1, 0), (5, 1, 1), (5, 2, 0)])
def test_four_images_per_block_ragged_groups(B, mode, wide):
    # conv2: 16 x 16 x 32 -> 8 x 8 x 64: geometry 1; B = 5: the second block holds one image and three empty slots
    labels, _ = _run(EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128), B, 31 + B, mode=mode, options={'winograd_wide': wide})
    assert sum('conv_wino_f32' in l for l in labels) == (1 if mode == 1 else 4)


@pytest.mark.parametrize('mode,wide', [(1, 0), (1, 1), (2, 0)])
def test_both_geometries_in_one_network_with_batch_norm(mode, wide):
    # wide = 1: blocks of four waves, each over both 32-channel halves of the block's 64 channels
    labels, _ = _run(EncoderConfig((64, 64, 3), [32, 64, 64], [2, 2, 2], 5, 128, True), 3, 47, mode=mode, options={'winograd_wide': wide})
    assert sum('conv_wino_f32' in l for l in labels) == (2 if mode == 1 else 8)
    assert labels[0].startswith('conv1') and 'wino' not in labels[0]


def test_min_batch_and_ineligible_layers_keep_the_direct_kernels():
    cfg = EncoderConfig((64, 64, 3), [32, 64], [2, 2], 5, 128)
    labels, _ = _run(cfg, 2, 5, options={'winograd_min_batch', 3} and {'winograd_min_batch': 3})
    assert not any('wino' in l for l in labels)
    # a 24 x 24 output is neither whole 16 x 16 regions nor 8 x 8: refused at option time, nothing changes
    w = synth.make_weights(seed=3, shape=(96, 96, 3), num_filter=[32, 64], strides=[2, 2], latent=128)
    enc = eb.EmuEncoder(w, EncoderConfig((96, 96, 3), [32, 64], [2, 2], 5, 128))
    with pytest.raises(Exception):
        enc.set_option('winograd', 1)
    enc.close()


@pytest.mark.parametrize('xcd_cols', [-1, 0, 1, 2])
def test_block_to_xcd_mappings_cover_every_region_and_column_block(xcd_cols):
    # conv2: 64 x 64 x 64 -> 32 x 32 x 128 at B = 3: 12 regions x 2 column blocks.  xcd_cols = 1: 2 column groups x 4 region groups (the grid
    # is padded to 8 * 1 * 3 = 24 blocks), 2 (= the default for two column blocks): 1 x 8 (8 * 2 * 2 = 32 blocks, eight of them leave at once)
    labels, _ = _run(EncoderConfig((128, 128, 3), [64, 128], [2, 2], 5, 64), 3, 61, options={'winograd_xcd_cols': xcd_cols})
    assert sum('conv_wino_f32' in l for l in labels) == 1


@pytest.mark.parametrize('xcd_cols', [0, 1])
def test_block_to_xcd_mapping_with_four_images_per_block(xcd_cols):
    # conv2: 16 x 16 x 32 -> 8 x 8 x 64 at B = 21: 6 regions (the last one ragged) x 1 column block over 8 region groups
    labels, _ = _run(EncoderConfig((32, 32, 3), [32, 64], [2, 2], 5, 128), 21, 67, options={'winograd_xcd_cols': xcd_cols})
    assert sum('conv_wino_f32' in l for l in labels) == 1
