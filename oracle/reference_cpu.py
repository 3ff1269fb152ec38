"""CPU oracle: a restatement of the AugmentedAutoencoder orientation-inference hot path.

*** TEST INFRASTRUCTURE ONLY ***  Only tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py may import this module.  The product package
(augmentedautoencoder_amd) never imports it and fails loudly when the HIP
library is missing.

PARITY UNPINNED for the TensorFlow arithmetic: conv2d / dense / batch-norm /
l2_normalize / matmul live in TensorFlow (tensorflow 2.6.0 pinned in
/root/reference/aae_py37_tf26.yml:102-105), which is neither vendored in the
reference tree nor installed here, and the reference ships no golden vectors /
known-answer tests for this path.  Those functions are
pinned only against themselves through three independent implementations that
must agree (numpy-im2col fp64 here, torch.conv2d fp32/fp64 here, plain-C direct
loops in oracle/aae_oracle.c).
PINNED against outputs of the reference's own code run in the build container
(fixtures + generators under tests/golden/):
  (a) codebook-row -> rotation table: pysixd_stuff/view_sampler.py and
      Dataset.viewsphere_for_embedding (all 92232 x 3 x 3 doubles by digest);
  (b) everything between "similarity matrix" and "pose": input /255 and batch
      handling, np.argmax / upright / top-n index selection, squeeze behaviour,
      auto_pose6d geometry, batch iteration -- recorded from
      auto_pose/ae/{codebook,dataset,utils}.py with TF/cv2 stubbed and
      session.run answering from a provided similarity matrix
      (tests/golden/make_codebook_logic_golden.py, tests/test_golden_codebook_logic.py).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Semantics that come from TensorFlow's documented op
behaviour rather than from a file in the tree are tagged [TF-semantics].
"""
from __future__ import annotations

import numpy as np

L2_NORMALIZE_EPS = 1e-12   # tf.nn.l2_normalize default epsilon [TF-semantics]
BN_EPS = 1e-3              # tf.layers.batch_normalization default epsilon [TF-semantics]


# --------------------------------------------------------------------------
# a1: input normalisation  (auto_pose/ae/codebook.py:58-61, :137-141)
# --------------------------------------------------------------------------
def input_to_float(x: np.ndarray) -> np.ndarray:
    """uint8 -> x/255. in float64 (NumPy true division), add batch dim.

    The TF feed then casts to the placeholder dtype float32
    (ae_factory.py:133) -- done by the callers below via ``astype``.
    """
    x = np.asarray(x)
    if x.dtype == np.uint8:
        x = x / 255.
    if x.ndim == 3:
        x = np.expand_dims(x, 0)
    return x


def u8_lut_f32() -> np.ndarray:
    """The 256 float32 values a uint8 crop can take after x/255. -> float32 feed."""
    return (np.arange(256, dtype=np.float64) / 255.).astype(np.float32)


# --------------------------------------------------------------------------
# a2-a5: conv2d(k, stride, 'same', relu)   (auto_pose/ae/encoder.py:41-50)
# --------------------------------------------------------------------------
def same_pad(in_size: int, k: int, s: int):
    """[TF-semantics] SAME padding: out=ceil(in/s); total=max((out-1)*s+k-in,0);
    before=total//2, after=total-before (the extra pixel goes at the end)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv2d_same_relu_np(x, kernel, bias, stride, dtype=np.float64, relu=True):
    """NHWC x, HWIO kernel; im2col + matmul in ``dtype`` (fp64 = arbiter)."""
    x = np.asarray(x, dtype=dtype)
    kernel = np.asarray(kernel, dtype=dtype)
    B, H, W, C = x.shape
    kh, kw, ci, co = kernel.shape
    assert ci == C
    Ho, pt, pb = same_pad(H, kh, stride)
    Wo, pl, pr = same_pad(W, kw, stride)
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C), dtype=dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    cols = np.empty((B, Ho, Wo, kh, kw, C), dtype=dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + stride * Ho:stride, j:j + stride * Wo:stride, :]
    y = cols.reshape(B * Ho * Wo, kh * kw * C) @ kernel.reshape(kh * kw * C, co)
    y = y.reshape(B, Ho, Wo, co) + np.asarray(bias, dtype=dtype)
    if relu:
        y = np.maximum(y, 0)
    return y


def batch_norm_inference_np(x, gamma, beta, mean, var, dtype=np.float64, eps=BN_EPS):
    """tf.layers.batch_normalization(x, training=False), applied AFTER the ReLU
    (encoder.py:51-52).  [TF-semantics] tf.nn.batch_normalization:
    inv = rsqrt(var+eps)*gamma ; y = x*inv + (beta - mean*inv)."""
    inv = (1.0 / np.sqrt(np.asarray(var, dtype) + dtype(eps))) * np.asarray(gamma, dtype)
    return np.asarray(x, dtype) * inv + (np.asarray(beta, dtype) - np.asarray(mean, dtype) * inv)


def layer_names(num_layers: int, batch_norm: bool):
    """TF auto-names under the experiment scope (SURVEY.md section 8a inventory):
    conv2d, conv2d_1, ... ; batch_normalization, batch_normalization_1, ... ; dense."""
    convs = ['conv2d' if i == 0 else 'conv2d_%d' % i for i in range(num_layers)]
    bns = ['batch_normalization' if i == 0 else 'batch_normalization_%d' % i
           for i in range(num_layers)] if batch_norm else []
    return convs, bns


def encoder_forward_np(x, weights, strides, batch_norm=False, dtype=np.float64,
                       return_activations=False):
    """Encoder.encoder_out + Encoder.z (encoder.py:37-68) in numpy, ``dtype`` math.

    x: float NHWC in [0,1] (already through input_to_float).  The float32 feed
    cast is applied first, exactly as the TF placeholder does.
    """
    h = np.asarray(x).astype(np.float32).astype(dtype)
    convs, bns = layer_names(len(strides), batch_norm)
    acts = []
    for i, s in enumerate(strides):
        h = conv2d_same_relu_np(h, weights[convs[i] + '/kernel'], weights[convs[i] + '/bias'], s, dtype)
        if batch_norm:
            p = bns[i]
            h = batch_norm_inference_np(h, weights[p + '/gamma'], weights[p + '/beta'],
                                        weights[p + '/moving_mean'], weights[p + '/moving_variance'], dtype)
        acts.append(h)
    flat = h.reshape(h.shape[0], -1)                      # tf.layers.flatten, NHWC row-major (encoder.py:54)
    z = flat @ np.asarray(weights['dense/kernel'], dtype) + np.asarray(weights['dense/bias'], dtype)
    if return_activations:
        return z, acts
    return z


def encoder_forward_torch(x, weights, strides, batch_norm=False, dtype='float32',
                          return_activations=False):
    """Same graph through torch-CPU conv2d (oneDNN) -- the independent second
    implementation, and the timed CPU baseline (fp32, all host cores)."""
    import torch
    import torch.nn.functional as F
    td = getattr(torch, dtype)
    h = torch.from_numpy(np.ascontiguousarray(np.asarray(x).astype(np.float32))).to(td)
    h = h.permute(0, 3, 1, 2).contiguous()               # NHWC -> NCHW for torch
    convs, bns = layer_names(len(strides), batch_norm)
    acts = []
    for i, s in enumerate(strides):
        k = torch.from_numpy(np.asarray(weights[convs[i] + '/kernel'])).to(td)   # HWIO
        b = torch.from_numpy(np.asarray(weights[convs[i] + '/bias'])).to(td)
        kh, kw = k.shape[0], k.shape[1]
        _, pt, pb = same_pad(h.shape[2], kh, s)
        _, pl, pr = same_pad(h.shape[3], kw, s)
        h = F.pad(h, (pl, pr, pt, pb))
        h = F.relu(F.conv2d(h, k.permute(3, 2, 0, 1).contiguous(), b, stride=s))
        if batch_norm:
            p = bns[i]
            g, be, mu, var = (torch.from_numpy(np.asarray(weights[p + '/' + n])).to(td)
                              for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'))
            inv = torch.rsqrt(var + BN_EPS) * g
            h = h * inv[None, :, None, None] + (be - mu * inv)[None, :, None, None]
        acts.append(h)
    flat = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)  # back to NHWC order for flatten
    z = flat @ torch.from_numpy(np.asarray(weights['dense/kernel'])).to(td) \
        + torch.from_numpy(np.asarray(weights['dense/bias'])).to(td)
    if return_activations:
        return z.numpy(), [a.permute(0, 2, 3, 1).contiguous().numpy() for a in acts]
    return z.numpy()


# --------------------------------------------------------------------------
# a9-a13: codebook  (auto_pose/ae/codebook.py:27, 50-51, 55-75)
# --------------------------------------------------------------------------
def l2_normalize(z, dtype=np.float64, eps=L2_NORMALIZE_EPS):
    """tf.nn.l2_normalize(z, 1) = z * rsqrt(max(sum(z^2), eps)) [TF-semantics] (codebook.py:27)."""
    z = np.asarray(z, dtype)
    ss = np.sum(z * z, axis=1, keepdims=True)
    return z * (dtype(1.0) / np.sqrt(np.maximum(ss, dtype(eps))))


def cos_similarity(z, embedding_normalized, dtype=np.float64):
    """tf.matmul(l2_normalize(z), embedding_normalized, transpose_b=True) (codebook.py:50)."""
    return l2_normalize(z, dtype) @ np.asarray(embedding_normalized, dtype).T


def nearest_indices_reference(cosine_similarity, top_n=1, upright=False, num_cyclo=36):
    """Literal restatement of codebook.py:64-71 (host NumPy in the reference)."""
    if top_n == 1:
        if upright:
            idcs = np.argmax(cosine_similarity[:, ::int(num_cyclo)], axis=1) * int(num_cyclo)
        else:
            idcs = np.argmax(cosine_similarity, axis=1)
    else:
        unsorted_max_idcs = np.argpartition(-cosine_similarity.squeeze(), top_n)[:top_n]
        idcs = unsorted_max_idcs[np.argsort(-cosine_similarity.squeeze()[unsorted_max_idcs])]
    return idcs


def topk_canonical(cosine_similarity, k):
    """Deterministic top-k used as the tie-aware target: descending score,
    lowest index first among equal scores.  (The reference's order among exact
    ties is unspecified -- introselect + unstable argsort, codebook.py:70-71.)"""
    cs = np.asarray(cosine_similarity)
    out = np.empty((cs.shape[0], k), dtype=np.int64)
    for b in range(cs.shape[0]):
        order = np.lexsort((np.arange(cs.shape[1]), -cs[b].astype(np.float64)))
        out[b] = order[:k]
    return out


def normalize_codebook(embedding_z):
    """update_embedding tail (codebook.py:193,214-216): float64 buffer, row /
    np.linalg.norm (no epsilon), assigned to a float32 variable."""
    z = np.asarray(embedding_z, dtype=np.float64)
    return (z / np.linalg.norm(z, axis=1, keepdims=True)).astype(np.float32)


def batch_iteration_indices(N, batch_size):
    """auto_pose/ae/utils.py:20-26."""
    end = int(np.ceil(float(N) / float(batch_size)))
    for i in range(end):
        a = i * batch_size
        e = i * batch_size + batch_size
        e = e if e <= N else N
        yield (a, e)


def nearest_rotation_oracle(x, weights, strides, embedding_normalized, batch_norm=False,
                            top_n=1, upright=False, num_cyclo=36, dtype=np.float64):
    """Codebook.nearest_rotation(return_idcs=True) end to end (codebook.py:55-73).
    Returns (idcs, cosine_similarity)."""
    xf = input_to_float(x)
    if dtype == np.float64:
        z = encoder_forward_np(xf, weights, strides, batch_norm, np.float64) if xf.shape[0] * xf.shape[1] <= 512 \
            else encoder_forward_torch(xf, weights, strides, batch_norm, 'float64')
    else:
        z = encoder_forward_torch(xf, weights, strides, batch_norm, 'float32')
    cs = cos_similarity(z, embedding_normalized, dtype)
    return nearest_indices_reference(cs, top_n, upright, num_cyclo), cs


# --------------------------------------------------------------------------
# auto_pose6d geometry  (auto_pose/ae/codebook.py:79-129) -- "next" row N1, host fp64
# --------------------------------------------------------------------------
def auto_pose6d_geometry(idcs, Rs_all, embed_obj_bbs, predicted_bb, K_test, K_train, render_radius,
                         depth_pred=None):
    idcs = np.atleast_1d(idcs)
    top_n = len(idcs)
    Rs_est = Rs_all[idcs].copy()
    K_diag_ratio = np.sqrt(K_test[0, 0] ** 2 + K_test[1, 1] ** 2) / np.sqrt(K_train[0, 0] ** 2 + K_train[1, 1] ** 2)
    ts_est = np.empty((top_n, 3))
    for i, idx in enumerate(idcs):
        rendered_bb = embed_obj_bbs[idx].squeeze()
        if depth_pred is None:
            bb_diag_ratio = np.linalg.norm(np.float32(rendered_bb[2:])) / np.linalg.norm(np.float32(predicted_bb[2:]))
            z = bb_diag_ratio * K_diag_ratio * render_radius
        else:
            z = depth_pred
        cx_tr = rendered_bb[0] + rendered_bb[2] / 2. - K_train[0, 2]
        cy_tr = rendered_bb[1] + rendered_bb[3] / 2. - K_train[1, 2]
        cx_te = predicted_bb[0] + predicted_bb[2] / 2 - K_test[0, 2]
        cy_te = predicted_bb[1] + predicted_bb[3] / 2 - K_test[1, 2]
        tx = cx_te * z / K_test[0, 0] - cx_tr * render_radius / K_train[0, 0]
        ty = cy_te * z / K_test[1, 1] - cy_tr * render_radius / K_train[1, 1]
        t_est = np.array([tx, ty, z])
        ts_est[i] = t_est
        d_alpha_y = np.arctan(t_est[0] / np.sqrt(t_est[2] ** 2 + t_est[1] ** 2))
        d_alpha_x = - np.arctan(t_est[1] / t_est[2])
        R_corr_x = np.array([[1, 0, 0],
                             [0, np.cos(d_alpha_x), -np.sin(d_alpha_x)],
                             [0, np.sin(d_alpha_x), np.cos(d_alpha_x)]])
        R_corr_y = np.array([[np.cos(d_alpha_y), 0, np.sin(d_alpha_y)],
                             [0, 1, 0],
                             [-np.sin(d_alpha_y), 0, np.cos(d_alpha_y)]])
        Rs_est[i] = np.dot(R_corr_y, np.dot(R_corr_x, Rs_est[i]))
    return Rs_est, ts_est


# --------------------------------------------------------------------------
# N1 (caller side): detector crop extraction
#   AePoseEstimator.extract_square_patch(black_borders=True) + cv2.resize(INTER_LINEAR)
#   (auto_pose/m3_interface/ae_pose_estimator.py:106-131,157-162)
# cv2 is not installed here: the resize below restates OpenCV's uint8 INTER_LINEAR
# (fixed point, 11-bit coefficients: resize.cpp HResizeLinear / VResizeLinear<uchar>) --
# UNPINNED against a real cv2 build (the sampling convention is cross-checked against torch's float bilinear
# interpolation in tests/test_pose_estimator.py: within one grey level; the 11-bit rounding stays unpinned).
# --------------------------------------------------------------------------
def _cv_linear_coeffs(src, dst, horizontal=True):
    """Per output coordinate: (index of the first tap, short coefficients [a0, a1]).
    Horizontal taps are clamped with the fraction zeroed at the borders (resize.cpp xofs /
    ialpha set-up); vertical taps keep their fraction and the ROWS are clipped instead."""
    scale = 1.0 / (float(dst) / float(src))
    ofs = np.empty(dst, dtype=np.int64)
    coef = np.empty((dst, 2), dtype=np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if horizontal:
            if s < 0:
                f, s = np.float32(0), 0
            if s >= src - 1:
                f, s = np.float32(0), src - 1
        ofs[d] = s
        coef[d, 0] = int(np.rint(np.float32((np.float32(1.0) - f) * np.float32(2048.0))))
        coef[d, 1] = int(np.rint(np.float32(f * np.float32(2048.0))))
    return ofs, coef


def cv_resize_linear_u8(img, dsize):
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    xo, xa = _cv_linear_coeffs(sw, dw)
    yo, yb = _cv_linear_coeffs(sh, dh, horizontal=False)
    src = img.astype(np.int64)
    x1 = np.minimum(xo + 1, sw - 1)
    rows = src[:, xo, :] * xa[None, :, 0, None] + src[:, x1, :] * xa[None, :, 1, None]     # [sh, dw, C], scaled by 2048
    y0 = np.clip(yo, 0, sh - 1)
    y1 = np.clip(yo + 1, 0, sh - 1)
    s0, s1 = rows[y0], rows[y1]
    out = ((((yb[:, 0, None, None] * (s0 >> 4)) >> 16) + ((yb[:, 1, None, None] * (s1 >> 4)) >> 16) + 2) >> 2)
    return np.clip(out, 0, 255).astype(np.uint8)


def extract_square_patch_black_borders(scene_img, bb_xywh, pad_factor, resize=(128, 128)):
    """extract_square_patch(..., interpolation=INTER_LINEAR, black_borders=True)
    (ae_pose_estimator.py:106-131).  Parts of the box outside the image are black
    (the reference's slice assignment would raise there)."""
    x, y, w, h = np.array(bb_xywh).astype(np.int32)
    size = int(np.maximum(h, w) * pad_factor)
    crop = np.zeros((size, size, scene_img.shape[2]), dtype=np.uint8)
    oy, ox = (size - h) // 2, (size - w) // 2
    H, W = scene_img.shape[:2]
    for r in range(h):
        for c in range(w):
            if 0 <= y + r < H and 0 <= x + c < W and 0 <= oy + r < size and 0 <= ox + c < size:
                crop[oy + r, ox + c] = scene_img[y + r, x + c]
    return cv_resize_linear_u8(crop, resize)
