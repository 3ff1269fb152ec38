// Decoder-side kernels: nearest-neighbour upsampling fused into the convolution that follows it.
//
// Reference: /root/reference/auto_pose/ae/decoder.py:36-84 -- every decoder stage is
//     x = tf.image.resize_nearest_neighbor(x, size)      (align_corners=False: src = floor(dst*in/out))
//     x = tf.layers.conv2d(x, filters, k, padding='same', activation=relu | sigmoid)   [+ inference BN]
// The upsampled tensor is never materialised here.
//
// Exact 2x upsampling (the default STRIDES=[2,2,2,2]) folds further: output pixel (2m+py, 2n+px)
// only ever sees source pixels (m+dy, n+dx) with d = floor((p + k - pad)/2), so the KxK
// convolution on the upsampled image is, per output phase (py,px), a small convolution on the
// SOURCE image whose taps are sums of the original taps that land on the same source pixel
// (5x5 -> 3x3: 2.8x fewer multiply-adds).  Phase weights are summed in float64 on the host.
//   * wide layers (Cout > 4): the implicit-GEMM kernel in SCATTER mode (conv_igemm_f32.h),
//     four phase problems in one grid, each row stored at its (2m+py, 2n+px) position;
//   * the last layer (Cout = image channels <= 4, sigmoid): upconv2x_narrow_kernel below --
//     a 128-wide MFMA tile would waste 97 % of its columns, so this one runs on the vector
//     ALUs: one thread per source pixel, 4 phases x 4 channel accumulators, the input tile
//     staged in LDS once per 32-channel chunk, weights wave-uniform through scalar loads.
//   * anything else (other resize ratios, Cin not a multiple of 32): upconv_direct_kernel,
//     the shape-agnostic fallback.
#pragma once

namespace aae {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

struct UpconvDirectArgs {
    const float* x;         // [B,H,W,Cin] source resolution
    const float* w;         // HWIO [KS][KS][Cin][Cout]
    const float* bias;
    const float* bn_scale;  // or nullptr
    const float* bn_shift;
    float* out;             // [B,UH,UW,Cout]
    int H, W, Cin, UH, UW, Cout, KS, pad, act;
    long long total;        // B*UH*UW*Cout
};

// one thread per output element; k-ordered fp32 fma chain (kh, kw, ci ascending)
__global__ __launch_bounds__(256) void upconv_direct_kernel(const UpconvDirectArgs p) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < p.total; e += stride) {
        const int co = (int)(e % p.Cout);
        long long m = e / p.Cout;
        const int ow = (int)(m % p.UW); m /= p.UW;
        const int oh = (int)(m % p.UH);
        const long long b = m / p.UH;
        float acc = 0.f;
        for (int kh = 0; kh < p.KS; ++kh) {
            const int uh = oh - p.pad + kh;                       // row of the (virtual) upsampled image
            if ((unsigned)uh >= (unsigned)p.UH) continue;
            const int ih = min((int)(((long long)uh * p.H) / p.UH), p.H - 1);
            for (int kw = 0; kw < p.KS; ++kw) {
                const int uw = ow - p.pad + kw;
                if ((unsigned)uw >= (unsigned)p.UW) continue;
                const int iw = min((int)(((long long)uw * p.W) / p.UW), p.W - 1);
                const float* xp = p.x + ((b * p.H + ih) * p.W + iw) * p.Cin;
                const float* wp = p.w + ((long long)(kh * p.KS + kw) * p.Cin) * p.Cout + co;
                for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(xp[ci], wp[(long long)ci * p.Cout], acc);
            }
        }
        acc = apply_act(acc + p.bias[co], p.act);
        if (p.bn_scale) acc = acc * p.bn_scale[co] + p.bn_shift[co];
        p.out[e] = acc;
    }
}

struct UpconvNarrowArgs {
    const float* x;         // [B,H,W,Cin], Cin % 32 == 0
    const float* wq;        // [U*U taps][Cin][4 phases][4 channels] phase-combined, zero padded
    const float* bias;      // [4] zero padded
    float* out;             // [B,2H,2W,Cout]
    int H, W, Cin, Cout;    // Cout <= 4
    int U, dmin;            // taps per axis of the phase problems (union over phases), first offset
    int act;
    int tiles_y, tiles_x;   // 16x16-pixel tiles per image
    int row_stride;         // floats between staged tile rows: roundup((16+U-1)*36, 64)
};

constexpr int kNarrowPixStride = 36;     // 32 channels + 4 pad floats: 16 lanes of a ds_read_b128 group hit 16 distinct bank quads

__host__ __device__ inline int narrow_row_stride(int U) { return ((16 + U - 1) * kNarrowPixStride + 63) / 64 * 64; }
__host__ __device__ inline int narrow_smem_bytes(int U) { return (16 + U - 1) * narrow_row_stride(U) * (int)sizeof(float); }

// Block = one 16x16 tile of source pixels (+ halo) of one image.  Per 32-channel chunk the tile is
// staged once in LDS (each input element fetched once per block instead of once per tap), then every
// thread walks its U x U neighbourhood with ds_read_b128 and multiplies against wave-uniform
// weights (scalar loads): 16 accumulators = 4 output phases x up to 4 channels.
__global__ __launch_bounds__(256) void upconv2x_narrow_kernel(const UpconvNarrowArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* tile = reinterpret_cast<float*>(smem_raw);

    const int tid = threadIdx.x;
    const int txi = blockIdx.x % p.tiles_x;
    const int tyi = (blockIdx.x / p.tiles_x) % p.tiles_y;
    const long long b = blockIdx.x / (p.tiles_x * p.tiles_y);
    const int ty0 = tyi * 16, tx0 = txi * 16;
    const int py = tid >> 4, px = tid & 15;
    const int m = ty0 + py, n = tx0 + px;
    const bool live = m < p.H && n < p.W;
    const int side = 16 + p.U - 1;
    const int total = side * side * 8;               // float4 elements of one staged chunk

    float acc[4][4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[ph][c] = 0.f;

    for (int cc = 0; cc < p.Cin; cc += 32) {
        __syncthreads();                             // previous chunk's readers are done
        for (int e0 = 0; e0 < total; e0 += 1024) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = e0 + tid + 256 * j;
                const int g = e & 7, pix = e >> 3;
                const int r = pix / side, c = pix - r * side;
                const int sy = ty0 + r + p.dmin, sx = tx0 + c + p.dmin;
                v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (e < total && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W)
                    v[j] = *reinterpret_cast<const f32x4*>(p.x + ((b * p.H + sy) * p.W + sx) * p.Cin + cc + 4 * g);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = e0 + tid + 256 * j;
                const int g = e & 7, pix = e >> 3;
                const int r = pix / side, c = pix - r * side;
                if (e < total) *reinterpret_cast<f32x4*>(tile + r * p.row_stride + c * kNarrowPixStride + 4 * g) = v[j];
            }
        }
        __syncthreads();
        for (int ty = 0; ty < p.U; ++ty)
            for (int tx = 0; tx < p.U; ++tx) {
                const float* xp = tile + (py + ty) * p.row_stride + (px + tx) * kNarrowPixStride;
                const float* wt = p.wq + ((long long)(ty * p.U + tx) * p.Cin + cc) * 16;     // wave-uniform
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* w16 = wt + (g * 4 + j) * 16;
#pragma unroll
                        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[ph][c] = fmaf(xv[j], w16[ph * 4 + c], acc[ph][c]);
                    }
                }
            }
    }
    if (!live) return;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int oy = 2 * m + (ph >> 1), ox = 2 * n + (ph & 1);
        float* o = p.out + ((b * (2 * p.H) + oy) * (2LL * p.W) + ox) * p.Cout;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < p.Cout) o[c] = apply_act(acc[ph][c] + p.bias[c], p.act);
    }
}

}  // namespace aae
