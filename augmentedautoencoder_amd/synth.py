"""Seeded synthetic inputs for parity tests, smoke() and bench.py.

TEST / BENCH INFRASTRUCTURE -- no trained AAE weights or real crops exist in
this environment (download link only, /root/reference/README.md:184), so every
parity statement is on seeded synthetic data generated here.
"""
from __future__ import annotations

import numpy as np

DEFAULT_NUM_FILTER = [128, 256, 512, 512]     # auto_pose/ae/cfg/train_template.cfg:50
DEFAULT_STRIDES = [2, 2, 2, 2]                # :51
DEFAULT_KERNEL = 5                            # :52
DEFAULT_LATENT = 128                          # :49
DEFAULT_SHAPE = (128, 128, 3)                 # :6-8
DEFAULT_N = 2562 * 36                         # MIN_N_VIEWS * NUM_CYCLO (:40-41) = 92232


def _names(n, bn):
    convs = ['conv2d' if i == 0 else 'conv2d_%d' % i for i in range(n)]
    bns = ['batch_normalization' if i == 0 else 'batch_normalization_%d' % i for i in range(n)] if bn else []
    return convs, bns


def make_weights(seed=2024, shape=DEFAULT_SHAPE, num_filter=DEFAULT_NUM_FILTER, strides=DEFAULT_STRIDES,
                 kernel_size=DEFAULT_KERNEL, latent=DEFAULT_LATENT, batch_norm=False):
    """glorot-uniform kernels (the tf.layers default initialiser) + small non-zero
    biases (TF's zero default would leave the bias path untested)."""
    rng = np.random.default_rng(seed)
    H, W, C = shape
    w = {}
    convs, bns = _names(len(num_filter), batch_norm)
    cin = C
    for i, (co, s) in enumerate(zip(num_filter, strides)):
        fan_in, fan_out = kernel_size * kernel_size * cin, kernel_size * kernel_size * co
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        w[convs[i] + '/kernel'] = rng.uniform(-lim, lim, (kernel_size, kernel_size, cin, co)).astype(np.float32)
        w[convs[i] + '/bias'] = rng.uniform(-0.05, 0.05, (co,)).astype(np.float32)
        if batch_norm:
            w[bns[i] + '/gamma'] = rng.uniform(0.5, 1.5, (co,)).astype(np.float32)
            w[bns[i] + '/beta'] = rng.uniform(-0.1, 0.1, (co,)).astype(np.float32)
            w[bns[i] + '/moving_mean'] = rng.uniform(0.0, 0.2, (co,)).astype(np.float32)
            w[bns[i] + '/moving_variance'] = rng.uniform(0.5, 1.5, (co,)).astype(np.float32)
        cin = co
        H, W = -(-H // s), -(-W // s)
    flat = H * W * cin
    lim = np.sqrt(6.0 / (flat + latent))
    w['dense/kernel'] = rng.uniform(-lim, lim, (flat, latent)).astype(np.float32)
    w['dense/bias'] = rng.uniform(-0.05, 0.05, (latent,)).astype(np.float32)
    return w


def make_crops(B, seed=1234, shape=DEFAULT_SHAPE):
    """uint8 BGR crops: a mixture of uniform noise, structured low-frequency
    blobs/rectangles on black (diverse latents), and all-0 / all-255 edge crops."""
    rng = np.random.default_rng(seed)
    H, W, C = shape
    out = np.zeros((B, H, W, C), dtype=np.uint8)
    for b in range(B):
        kind = b % 8
        if kind == 0:
            out[b] = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
        elif kind == 6 and b >= 8:
            out[b] = 0 if (b // 8) % 2 else 255
        else:
            img = np.zeros((H, W, C), dtype=np.float64)
            for _ in range(int(rng.integers(2, 7))):
                y0, x0 = rng.integers(0, H), rng.integers(0, W)
                h, w_ = rng.integers(max(2, H // 16), max(3, H // 2)), rng.integers(max(2, W // 16), max(3, W // 2))
                col = rng.uniform(0, 255, (C,))
                if rng.random() < 0.5:
                    img[y0:y0 + h, x0:x0 + w_] = col
                else:
                    yy, xx = np.mgrid[0:H, 0:W]
                    m = np.exp(-(((yy - y0) / (0.5 * h + 1)) ** 2 + ((xx - x0) / (0.5 * w_ + 1)) ** 2))
                    img += m[..., None] * col
            img += rng.normal(0, 4.0, img.shape)
            out[b] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def make_codebook(N=DEFAULT_N, J=DEFAULT_LATENT, seed=7, planted_duplicates=64, num_cyclo=36):
    """N(0,1) rows normalised in float64 -> float32 (what update_embedding leaves in the
    variable), with exact duplicates planted at rows 36k / 36k+35: the reference's
    np.linspace(0, 2pi, 36) includes both endpoints (dataset.py:54), so those two
    rows are the same rotation and structurally (near-)tied."""
    rng = np.random.default_rng(seed)
    E = rng.standard_normal((N, J))
    E = (E / np.linalg.norm(E, axis=1, keepdims=True)).astype(np.float32)
    n_views = N // num_cyclo
    if planted_duplicates and n_views > 0 and num_cyclo > 1:
        ks = rng.choice(n_views, size=min(planted_duplicates, n_views), replace=False)
        for k in ks:
            E[num_cyclo * k + num_cyclo - 1] = E[num_cyclo * k]
    return E


def make_queries_near_rows(E, rows, noise=0.05, seed=99):
    """Latents whose nearest neighbour is known by construction: a scaled codebook
    row plus small noise (un-normalised, as the encoder would emit)."""
    rng = np.random.default_rng(seed)
    z = E[rows].astype(np.float64) * rng.uniform(0.5, 20.0, (len(rows), 1))
    z += noise * rng.standard_normal(z.shape) * np.linalg.norm(z, axis=1, keepdims=True) / np.sqrt(E.shape[1])
    return z.astype(np.float32)


def make_decoder_weights_for(cfg, seed=4242):
    """Synthetic decoder variables for a weights.DecoderConfig (glorot-uniform kernels, small biases, BN statistics)
    under the TF names cfg.variable_names() gives -- inputs of the decoder line of bench.py."""
    rng = np.random.default_rng(seed)
    dense, convs, final, bns = cfg.variable_names()
    dims = cfg.layer_dimensions()
    k = cfg.kernel_size
    w = {}

    def glorot(shape, fan_in, fan_out):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, shape).astype(np.float32)

    def add_bn(name, c):
        w[name + '/gamma'] = rng.uniform(0.5, 1.5, (c,)).astype(np.float32)
        w[name + '/beta'] = rng.uniform(-0.1, 0.1, (c,)).astype(np.float32)
        w[name + '/moving_mean'] = rng.uniform(0.0, 0.2, (c,)).astype(np.float32)
        w[name + '/moving_variance'] = rng.uniform(0.5, 1.5, (c,)).astype(np.float32)

    units = dims[0][0] * dims[0][1] * cfg.num_filters[0]
    w[dense + '/kernel'] = glorot((cfg.latent_space_size, units), cfg.latent_space_size, units)
    w[dense + '/bias'] = rng.uniform(0.0, 0.1, (units,)).astype(np.float32)
    if cfg.batch_norm:
        add_bn(bns[0], units)
    cin = cfg.num_filters[0]
    for i in range(1, cfg.num_layers):
        co = cfg.num_filters[i]
        w[convs[i - 1] + '/kernel'] = glorot((k, k, cin, co), k * k * cin, k * k * co)
        w[convs[i - 1] + '/bias'] = rng.uniform(-0.05, 0.05, (co,)).astype(np.float32)
        if cfg.batch_norm:
            add_bn(bns[i], co)
        cin = co
    c = cfg.shape[2]
    w[final + '/kernel'] = glorot((k, k, cin, c), k * k * cin, k * k * c)
    w[final + '/bias'] = rng.uniform(-0.05, 0.05, (c,)).astype(np.float32)
    return w
