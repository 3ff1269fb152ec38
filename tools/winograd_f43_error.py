#!/usr/bin/env python
"""Gate of VERDICT r5 item 7 (stretch): would F(4 x 4, 3 x 3) for the 3 x 3-tap polyphase component keep the fp32 accuracy?
36 products per 4 x 4 outputs instead of 64 (F(2 x 2): 16 per 2 x 2) -- but the transforms carry 1/24 ... 8 (Lavin & Gray's
points 0, +-1, +-2, inf), where F(2 x 2, 3 x 3) uses 0, +-1/2, 1.  This script evaluates conv3's 3 x 3 component (256 -> 512
channels, post-ReLU-like activations, Glorot weights) on a sample of outputs in three ways, every operation rounded to float32
in the order the kernel would do it (input transform, per-point products accumulated over the channels in fp32, output
transform), against float64 direct evaluation:
    direct fp32 fma chain | Winograd F(2 x 2, 3 x 3) | Winograd F(4 x 4, 3 x 3)
and prints the largest error relative to the output scale.  Gate: F(4 x 4) <= 2e-6 of the output scale (SURVEY 8c's margin)."""
import json

import numpy as np

rng = np.random.default_rng(2024)
C, N = 256, 64                      # input channels, output channels sampled
T = 24                              # tiles sampled
f32 = np.float32

# F(2,3)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
# F(4,3) (Lavin & Gray 2015)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def mm32(A, X):
    """A (small constant matrix, exact in fp32 or rounded once) times X along axis 0, every add / multiply rounded to fp32, left to right"""
    A32 = A.astype(f32)
    out = np.zeros((A.shape[0],) + X.shape[1:], dtype=f32)
    for i in range(A.shape[0]):
        acc = np.zeros(X.shape[1:], dtype=f32)
        for j in range(A.shape[1]):
            if A32[i, j] != 0:
                acc = (acc + A32[i, j] * X[j]).astype(f32)
        out[i] = acc
    return out


def winograd(d, g, BT, G, AT, m):
    """d [T, a, a, C] patches (a = m + 2), g [3, 3, C, N] -> [T, m, m, N], fp32 rounding everywhere"""
    a = m + 2
    U = np.einsum('ik,klcn,jl->ijcn', G, g.astype(np.float64), G).astype(f32)            # G g G^T in float64, rounded once (host)
    out = np.zeros((len(d), m, m, g.shape[3]), dtype=f32)
    for t in range(len(d)):
        x = d[t].astype(f32)                                     # [a, a, C]
        v = mm32(BT, x)                                          # rows
        v = np.moveaxis(mm32(BT, np.moveaxis(v, 1, 0)), 0, 1)    # columns
        M = np.zeros((a, a, g.shape[3]), dtype=f32)
        for c in range(C):                                       # fp32 accumulation over the channels, channel order (the MFMA's fma chain)
            M = (M + v[:, :, c, None] * U[:, :, c, :]).astype(f32)
        y = mm32(AT, M)
        y = np.moveaxis(mm32(AT, np.moveaxis(y, 1, 0)), 0, 1)
        out[t] = y
    return out


def direct64(d, g, m):
    out = np.zeros((len(d), m, m, g.shape[3]))
    for i in range(m):
        for j in range(m):
            out[:, i, j, :] = np.einsum('tklc,klcn->tn', d[:, i:i + 3, j:j + 3, :].astype(np.float64), g.astype(np.float64))
    return out


def direct32(d, g, m):
    out = np.zeros((len(d), m, m, g.shape[3]), dtype=f32)
    for i in range(m):
        for j in range(m):
            acc = np.zeros((len(d), g.shape[3]), dtype=f32)
            for k in range(3):
                for l in range(3):
                    for c in range(C):
                        acc = (acc + d[:, i + k, j + l, c, None].astype(f32) * g[k, l, c, :].astype(f32)).astype(f32)
            out[:, i, j, :] = acc
    return out


r = rng.random((T, 6, 6, C))
d = np.where(r < 0.45, 0.0, 1.5 * (r - 0.45)).astype(f32)        # post-ReLU-like
lim = np.sqrt(6.0 / (25.0 * C + 25.0 * 512))
g = ((2 * rng.random((3, 3, C, N)) - 1) * lim).astype(f32)
ref4 = direct64(d, g, 4)
scale = float(np.abs(ref4).max())
w4 = winograd(d, g, BT4, G4, AT4, 4)
# F(2 x 2): the four 2 x 2 tiles of the same 4 x 4 outputs
w2 = np.zeros_like(w4)
for ty in range(2):
    for tx in range(2):
        w2[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2, :] = winograd(d[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4, :], g, BT2, G2, AT2, 2)
d32 = direct32(d, g, 4)
out = {'what': 'winograd_f4x4_3x3_error_gate', 'component': "conv3's 3 x 3-tap polyphase component: 256 input channels, %d of 512 output channels, %d tiles of 4 x 4 outputs" % (N, T),
       'output_scale': scale,
       'max_abs_err_over_scale': {'direct_fp32_fma_chain': float(np.abs(d32 - ref4).max() / scale), 'winograd_F2x2_3x3': float(np.abs(w2 - ref4).max() / scale),
                                  'winograd_F4x4_3x3': float(np.abs(w4 - ref4).max() / scale)},
       'gate': 'F(4 x 4, 3 x 3) error <= 2e-6 of the output scale', }
out['passes'] = out['max_abs_err_over_scale']['winograd_F4x4_3x3'] <= 2e-6
print(json.dumps(out))
