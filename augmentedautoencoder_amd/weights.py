"""Encoder / codebook parameter plumbing (host side, NumPy only).

The reference keeps weights *and* the codebook in a TensorFlow checkpoint
(tf.train.Saver, /root/reference/auto_pose/ae/ae_embed.py:60,91; the codebook
is the non-trainable variable ``embedding_normalized`` plus ``embed_obj_bbs_var``,
auto_pose/ae/codebook.py:28-45).  TensorFlow is not available here, so the
native container is a flat ``.npz`` whose keys are the TF variable names
without the experiment scope:

    conv2d/kernel  conv2d/bias  conv2d_1/kernel ...   HWIO float32
    batch_normalization{,_i}/{gamma,beta,moving_mean,moving_variance}   (if BN)
    dense/kernel [Ho*Wo*C, J]  dense/bias [J]
    embedding_normalized [N, J] float32      (optional)
    embed_obj_bbs_var [N, 4] int32           (optional)
"""
from __future__ import annotations

import ast
import ctypes

import numpy as np

from . import _lib


def conv_names(num_layers):
    return ['conv2d' if i == 0 else 'conv2d_%d' % i for i in range(num_layers)]


def bn_names(num_layers):
    return ['batch_normalization' if i == 0 else 'batch_normalization_%d' % i for i in range(num_layers)]


def same_out(size, stride):
    return -(-int(size) // int(stride))


class EncoderConfig(object):
    """Shapes of the encoder, from the [Dataset]/[Network] cfg keys that
    ae_factory.build_encoder reads (/root/reference/auto_pose/ae/ae_factory.py:33-48)."""

    def __init__(self, shape=(128, 128, 3), num_filter=(128, 256, 512, 512), strides=(2, 2, 2, 2),
                 kernel_size=5, latent_space_size=128, batch_norm=False):
        self.shape = tuple(int(v) for v in shape)
        self.num_filter = [int(v) for v in num_filter]
        self.strides = [int(v) for v in strides]
        self.kernel_size = int(kernel_size)
        self.latent_space_size = int(latent_space_size)
        self.batch_norm = bool(batch_norm)
        if len(self.num_filter) != len(self.strides):
            raise ValueError('NUM_FILTER and STRIDES differ in length')
        if not 1 <= len(self.num_filter) <= _lib.AAE_MAX_LAYERS:
            raise ValueError('between 1 and %d conv layers supported' % _lib.AAE_MAX_LAYERS)

    @classmethod
    def from_cfg(cls, args):
        """args: configparser.ConfigParser of a train cfg (cfg/train_template.cfg layout).
        Values are parsed with ast.literal_eval (the reference eval()s them)."""
        return cls(
            shape=(args.getint('Dataset', 'H'), args.getint('Dataset', 'W'), args.getint('Dataset', 'C')),
            num_filter=ast.literal_eval(args.get('Network', 'NUM_FILTER')),
            strides=ast.literal_eval(args.get('Network', 'STRIDES')),
            kernel_size=args.getint('Network', 'KERNEL_SIZE_ENCODER'),
            latent_space_size=args.getint('Network', 'LATENT_SPACE_SIZE'),
            batch_norm=args.getboolean('Network', 'BATCH_NORMALIZATION'),
        )

    @property
    def num_layers(self):
        return len(self.num_filter)

    def layer_shapes(self):
        """[(H, W, Cin, Ho, Wo, Cout)] per conv layer."""
        H, W, C = self.shape
        out = []
        for co, s in zip(self.num_filter, self.strides):
            Ho, Wo = same_out(H, s), same_out(W, s)
            out.append((H, W, C, Ho, Wo, co))
            H, W, C = Ho, Wo, co
        return out

    @property
    def flatten_size(self):
        _, _, _, Ho, Wo, co = self.layer_shapes()[-1]
        return Ho * Wo * co

    def flops_per_crop(self):
        k = self.kernel_size
        f = 0
        for (_, _, ci, Ho, Wo, co) in self.layer_shapes():
            f += 2 * Ho * Wo * k * k * ci * co
        return f + 2 * self.flatten_size * self.latent_space_size

    def param_bytes(self):
        """fp32 bytes of the encoder's kernels + biases (59.4 MB for the default net: SURVEY.md section 8a)"""
        k = self.kernel_size
        n = sum(k * k * ci * co + co for (_, _, ci, _, _, co) in self.layer_shapes())
        return 4 * (n + self.flatten_size * self.latent_space_size + self.latent_space_size)

    def to_desc(self):
        d = _lib.EncoderDesc()
        d.in_h, d.in_w, d.in_c = self.shape
        d.num_layers = self.num_layers
        for i in range(self.num_layers):
            d.num_filters[i] = self.num_filter[i]
            d.strides[i] = self.strides[i]
        d.kernel_size = self.kernel_size
        d.latent_size = self.latent_space_size
        d.batch_norm = 1 if self.batch_norm else 0
        d.bn_eps = 1e-3
        return d


def ordered_weight_arrays(weights, cfg):
    """Flatten a {name: array} dict into the array order aae_encoder_create expects,
    validating every shape.  Returns a list of C-contiguous float32 arrays."""
    out = []
    convs, bns = conv_names(cfg.num_layers), bn_names(cfg.num_layers)
    k = cfg.kernel_size

    def take(name, shape):
        if name not in weights:
            raise ValueError('missing weight %r' % name)
        a = np.ascontiguousarray(np.asarray(weights[name], dtype=np.float32))
        if tuple(a.shape) != tuple(shape):
            raise ValueError('weight %r has shape %s, expected %s' % (name, a.shape, tuple(shape)))
        out.append(a)

    for i, (_, _, ci, _, _, co) in enumerate(cfg.layer_shapes()):
        take(convs[i] + '/kernel', (k, k, ci, co))
        take(convs[i] + '/bias', (co,))
        if cfg.batch_norm:
            for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                take(bns[i] + '/' + n, (co,))
    take('dense/kernel', (cfg.flatten_size, cfg.latent_space_size))
    take('dense/bias', (cfg.latent_space_size,))
    return out


def as_pointer_array(arrays):
    ptrs = (ctypes.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        ptrs[i] = a.ctypes.data
    return ptrs


def save_npz(path, weights, embedding_normalized=None, embed_obj_bbs=None):
    blob = {k: np.asarray(v) for k, v in weights.items()}
    if embedding_normalized is not None:
        blob['embedding_normalized'] = np.asarray(embedding_normalized, dtype=np.float32)
    if embed_obj_bbs is not None:
        blob['embed_obj_bbs_var'] = np.asarray(embed_obj_bbs, dtype=np.int32)
    np.savez(path, **blob)


def to_bf16_bits(a):
    """float32 array -> uint16 bfloat16 bit patterns, round-to-nearest-even (the storage
    format of a bf16 codebook, BASELINE config 5)."""
    u = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32)
    r = u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))
    return (r >> np.uint32(16)).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def load_npz(path):
    """Returns (weights dict, embedding_normalized or None, embed_obj_bbs or None)."""
    with np.load(path) as f:
        blob = {k: f[k] for k in f.files}
    emb = blob.pop('embedding_normalized', None)
    bbs = blob.pop('embed_obj_bbs_var', None)
    return blob, emb, bbs


class DecoderConfig(object):
    """Shapes of Decoder(reconstruction_target, latent_code, num_filters, kernel_size, strides, ...)
    as ae_factory.build_decoder fills it (/root/reference/auto_pose/ae/ae_factory.py:50-70):
    num_filter / strides are given in cfg (encoder) order and used reversed."""

    def __init__(self, shape=(128, 128, 3), num_filter=(128, 256, 512, 512), strides=(2, 2, 2, 2),
                 kernel_size=5, latent_space_size=128, batch_norm=False, auxiliary_mask=False):
        self.shape = tuple(int(v) for v in shape)
        self.num_filters = [int(v) for v in reversed(list(num_filter))]
        self.strides = [int(v) for v in reversed(list(strides))]
        self.kernel_size = int(kernel_size)
        self.latent_space_size = int(latent_space_size)
        self.batch_norm = bool(batch_norm)
        self.auxiliary_mask = bool(auxiliary_mask)
        if len(self.num_filters) != len(self.strides):
            raise ValueError('NUM_FILTER and STRIDES differ in length')
        if not 1 <= len(self.num_filters) <= _lib.AAE_MAX_LAYERS:
            raise ValueError('between 1 and %d layers supported' % _lib.AAE_MAX_LAYERS)

    @classmethod
    def from_cfg(cls, args):
        return cls(
            shape=(args.getint('Dataset', 'H'), args.getint('Dataset', 'W'), args.getint('Dataset', 'C')),
            num_filter=ast.literal_eval(args.get('Network', 'NUM_FILTER')),
            strides=ast.literal_eval(args.get('Network', 'STRIDES')),
            kernel_size=args.getint('Network', 'KERNEL_SIZE_DECODER'),
            latent_space_size=args.getint('Network', 'LATENT_SPACE_SIZE'),
            batch_norm=args.getboolean('Network', 'BATCH_NORMALIZATION'),
            auxiliary_mask=args.getboolean('Network', 'AUXILIARY_MASK', fallback=False),
        )

    @property
    def num_layers(self):
        return len(self.num_filters)

    def layer_dimensions(self):
        """[[h_i, w_i]] = int(H / prod(strides[i:]))  (decoder.py:41)."""
        h, w = self.shape[:2]
        out = []
        for i in range(self.num_layers):
            p = 1
            for s in self.strides[i:]:
                p *= s
            out.append([int(h / p), int(w / p)])
        return out

    def variable_names(self, weights=None):
        """(dense, [hidden convs], final conv, [batch norms]) -- TF auto-names in a graph whose
        encoder was built first (ae_factory.py:134-139).

        The decoder's dense layer is the second tf.layers.dense of the graph ('dense_1') unless the encoder was built
        with VARIATIONAL > 0: Encoder.q_sigma (encoder.py:70-80) then takes 'dense_1' and the decoder gets 'dense_2'.
        Given the restored ``weights``, the name is resolved by the kernel shape ([latent, h0*w0*F0] is the decoder's,
        [flatten, latent] would be q_sigma's)."""
        L = self.num_layers
        convs = ['conv2d_%d' % (L + i) for i in range(L - 1)]
        final = 'conv2d_%d' % (2 * L - 1 + (1 if self.auxiliary_mask else 0))
        bns = ['batch_normalization_%d' % (L + i) for i in range(L)] if self.batch_norm else []
        dense = 'dense_1'
        if weights is not None:
            dims = self.layer_dimensions()
            want = (self.latent_space_size, dims[0][0] * dims[0][1] * self.num_filters[0])
            for cand in ('dense_1', 'dense_2'):
                k = weights.get(cand + '/kernel')
                if k is not None and tuple(np.shape(k)) == want:
                    dense = cand
                    break
        return dense, convs, final, bns

    def flops_per_image(self):
        """Nominal multiply-add count x2 of Decoder.x (upsampled-resolution convolutions)."""
        dims = self.layer_dimensions()
        k = self.kernel_size
        f = 2 * self.latent_space_size * dims[0][0] * dims[0][1] * self.num_filters[0]
        cin = self.num_filters[0]
        for i in range(1, self.num_layers):
            f += 2 * dims[i][0] * dims[i][1] * k * k * cin * self.num_filters[i]
            cin = self.num_filters[i]
        return f + 2 * self.shape[0] * self.shape[1] * k * k * cin * self.shape[2]

    def to_desc(self):
        d = _lib.DecoderDesc()
        d.out_h, d.out_w, d.out_c = self.shape
        d.num_layers = self.num_layers
        for i in range(self.num_layers):
            d.num_filters[i] = self.num_filters[i]
            d.strides[i] = self.strides[i]
        d.kernel_size = self.kernel_size
        d.latent_size = self.latent_space_size
        d.batch_norm = 1 if self.batch_norm else 0
        d.bn_eps = 1e-3
        return d


def ordered_decoder_weight_arrays(weights, cfg):
    """{name: array} -> the array order aae_decoder_create expects, shapes validated."""
    out = []
    dense, convs, final, bns = cfg.variable_names(weights)
    dims = cfg.layer_dimensions()
    k = cfg.kernel_size

    def take(name, shape):
        if name not in weights:
            raise ValueError('missing decoder weight %r' % name)
        a = np.ascontiguousarray(np.asarray(weights[name], dtype=np.float32))
        if tuple(a.shape) != tuple(shape):
            raise ValueError('decoder weight %r has shape %s, expected %s' % (name, a.shape, tuple(shape)))
        out.append(a)

    def take_bn(name, c):
        for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            take(name + '/' + n, (c,))

    units = dims[0][0] * dims[0][1] * cfg.num_filters[0]
    take(dense + '/kernel', (cfg.latent_space_size, units))
    take(dense + '/bias', (units,))
    if cfg.batch_norm:
        take_bn(bns[0], units)
    cin = cfg.num_filters[0]
    for i in range(1, cfg.num_layers):
        take(convs[i - 1] + '/kernel', (k, k, cin, cfg.num_filters[i]))
        take(convs[i - 1] + '/bias', (cfg.num_filters[i],))
        if cfg.batch_norm:
            take_bn(bns[i], cfg.num_filters[i])
        cin = cfg.num_filters[i]
    take(final + '/kernel', (k, k, cin, cfg.shape[2]))
    take(final + '/bias', (cfg.shape[2],))
    return out
