"""CPU restatement of the reference decoder forward pass (Decoder.x).

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by the product package.
PARITY UNPINNED: the arithmetic lives in TensorFlow (tf.layers.dense / conv2d /
batch_normalization, tf.image.resize_nearest_neighbor), which is not installed here and
the reference ships no recorded outputs; this file follows the call sites line by line:

    /root/reference/auto_pose/ae/decoder.py:36-84      layer order, sizes, activations
    /root/reference/auto_pose/ae/ae_factory.py:50-70   NUM_FILTER / STRIDES are reversed,
                                                        KERNEL_SIZE_DECODER, BATCH_NORMALIZATION
[TF-semantics] resize_nearest_neighbor(align_corners=False, half_pixel_centers=False):
src = min(floor(dst * in / out), in - 1); conv2d 'same' stride 1 pads (k-1)/2 both sides;
batch norm after the activation with eps 1e-3; sigmoid on the output layer.
The upsampled tensors ARE materialised here (the HIP path never builds them).
"""
import numpy as np

from . import reference_cpu as ref


def decoder_layer_names(num_layers, batch_norm, aux_mask=False):
    """TF auto-names of the decoder variables in a graph whose encoder (num_layers convs + one
    dense) was built first (ae_factory.py:134-139): dense_1, conv2d_<L>.., final conv last."""
    L = num_layers
    convs = ['conv2d_%d' % (L + i) for i in range(L - 1)]
    final = 'conv2d_%d' % (2 * L - 1 + (1 if aux_mask else 0))
    bns = ['batch_normalization_%d' % (L + i) for i in range(L)] if batch_norm else []
    return 'dense_1', convs, final, bns


def layer_dimensions(out_hw, strides):
    h, w = out_hw
    return [[int(h / np.prod(strides[i:])), int(w / np.prod(strides[i:]))] for i in range(len(strides))]


def resize_nearest_neighbor_np(x, size):
    B, H, W, C = x.shape
    oh, ow = int(size[0]), int(size[1])
    iy = np.minimum((np.arange(oh) * H) // oh, H - 1)
    ix = np.minimum((np.arange(ow) * W) // ow, W - 1)
    return x[:, iy][:, :, ix]


def _bn(h, weights, name, dtype):
    return ref.batch_norm_inference_np(h, weights[name + '/gamma'], weights[name + '/beta'],
                                       weights[name + '/moving_mean'], weights[name + '/moving_variance'], dtype)


def decoder_forward_np(z, weights, out_shape, num_filters, strides, batch_norm=False, dtype=np.float64,
                       aux_mask=False, return_activations=False):
    """num_filters / strides in DECODER order (already reversed).  Returns x [B,H,W,C] in [0,1]."""
    h_img, w_img, c_img = out_shape
    L = len(num_filters)
    dense, convs, final, bns = decoder_layer_names(L, batch_norm, aux_mask)
    dims = layer_dimensions((h_img, w_img), strides)
    z = np.asarray(z).astype(np.float32).astype(dtype)
    x = np.maximum(z @ np.asarray(weights[dense + '/kernel'], dtype) + np.asarray(weights[dense + '/bias'], dtype), 0)
    if batch_norm:
        x = _bn(x, weights, bns[0], dtype)
    x = x.reshape(-1, dims[0][0], dims[0][1], num_filters[0])
    acts = [x]
    for i in range(1, L):
        x = resize_nearest_neighbor_np(x, dims[i])
        x = ref.conv2d_same_relu_np(x, weights[convs[i - 1] + '/kernel'], weights[convs[i - 1] + '/bias'], 1, dtype)
        if batch_norm:
            x = _bn(x, weights, bns[i], dtype)
        acts.append(x)
    x = resize_nearest_neighbor_np(x, (h_img, w_img))
    y = ref.conv2d_same_relu_np(x, weights[final + '/kernel'], weights[final + '/bias'], 1, dtype, relu=False)
    out = 1.0 / (1.0 + np.exp(-y))
    if return_activations:
        return out, acts
    return out


def decoder_forward_torch(z, weights, out_shape, num_filters, strides, batch_norm=False, aux_mask=False):
    """fp32 torch-CPU version (timed CPU baseline of the decoder measurements)."""
    import torch
    import torch.nn.functional as F
    h_img, w_img, c_img = out_shape
    L = len(num_filters)
    dense, convs, final, bns = decoder_layer_names(L, batch_norm, aux_mask)
    dims = layer_dimensions((h_img, w_img), strides)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))

    def bn(x, name, ch_last):
        g, b, m, v = (t(weights[name + '/' + k]) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance'))
        inv = g / torch.sqrt(v + ref.BN_EPS)
        shape = (1, -1) if not ch_last else (1, -1, 1, 1)
        return x * inv.reshape(shape) + (b - m * inv).reshape(shape)

    x = torch.relu(t(z) @ t(weights[dense + '/kernel']) + t(weights[dense + '/bias']))
    if batch_norm:
        x = bn(x, bns[0], False)
    x = x.reshape(-1, dims[0][0], dims[0][1], num_filters[0]).permute(0, 3, 1, 2)
    for i in range(1, L):
        x = F.interpolate(x, size=tuple(dims[i]), mode='nearest')
        k = t(weights[convs[i - 1] + '/kernel']).permute(3, 2, 0, 1)
        x = torch.relu(F.conv2d(x, k, t(weights[convs[i - 1] + '/bias']), padding=k.shape[-1] // 2))
        if batch_norm:
            x = bn(x, bns[i], True)
    x = F.interpolate(x, size=(h_img, w_img), mode='nearest')
    k = t(weights[final + '/kernel']).permute(3, 2, 0, 1)
    x = torch.sigmoid(F.conv2d(x, k, t(weights[final + '/bias']), padding=k.shape[-1] // 2))
    return x.permute(0, 2, 3, 1).contiguous().numpy()


def make_decoder_weights(seed=4242, out_shape=(128, 128, 3), num_filter=(128, 256, 512, 512), strides=(2, 2, 2, 2),
                         kernel_size=5, latent=128, batch_norm=False, aux_mask=False):
    """Synthetic decoder variables (glorot-uniform kernels, small biases) under their TF names.
    num_filter / strides in ENCODER (cfg) order; the decoder uses them reversed."""
    rng = np.random.default_rng(seed)
    nf, st = list(reversed(num_filter)), list(reversed(strides))
    L = len(nf)
    dense, convs, final, bns = decoder_layer_names(L, batch_norm, aux_mask)
    dims = layer_dimensions(out_shape[:2], st)
    w = {}

    def glorot(shape, fan_in, fan_out):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, shape).astype(np.float32)

    def add_bn(name, c):
        w[name + '/gamma'] = rng.uniform(0.5, 1.5, (c,)).astype(np.float32)
        w[name + '/beta'] = rng.uniform(-0.1, 0.1, (c,)).astype(np.float32)
        w[name + '/moving_mean'] = rng.uniform(0.0, 0.2, (c,)).astype(np.float32)
        w[name + '/moving_variance'] = rng.uniform(0.5, 1.5, (c,)).astype(np.float32)

    units = dims[0][0] * dims[0][1] * nf[0]
    w[dense + '/kernel'] = glorot((latent, units), latent, units)
    w[dense + '/bias'] = rng.uniform(0.0, 0.1, (units,)).astype(np.float32)
    if batch_norm:
        add_bn(bns[0], units)
    cin = nf[0]
    for i in range(1, L):
        w[convs[i - 1] + '/kernel'] = glorot((kernel_size, kernel_size, cin, nf[i]), kernel_size ** 2 * cin, kernel_size ** 2 * nf[i])
        w[convs[i - 1] + '/bias'] = rng.uniform(-0.05, 0.05, (nf[i],)).astype(np.float32)
        if batch_norm:
            add_bn(bns[i], nf[i])
        cin = nf[i]
    c = out_shape[2]
    w[final + '/kernel'] = glorot((kernel_size, kernel_size, cin, c), kernel_size ** 2 * cin, kernel_size ** 2 * c)
    w[final + '/bias'] = rng.uniform(-0.05, 0.05, (c,)).astype(np.float32)
    return w
