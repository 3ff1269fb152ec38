// Dense layer for tiny batches (B <= 4: the reference's per-detection usage): z = flatten(x) . W + b as a
// weight-streaming GEMV (/root/reference/auto_pose/ae/encoder.py:58-68).  At M <= 4 a 128 x 128 MFMA tile is
// 97 % padding and the layer is the 16.8 MB read of W: each block streams a 128-k chunk of the packed weights
// [K/4][CoutPad][4] -- thread n reads 16 B = four consecutive k of ITS output column, 2 KB contiguous per
// slot row across the block -- against the activation chunk staged in LDS, and writes one partial row per
// batch element; splitk_reduce_kernel (fixed order) adds the chunks, bias and the optional BN.
#pragma once

namespace aae {

struct DenseGemvArgs {
    const float* x;        // [B][K]
    const float* wp;       // [K/4][CoutPad][4]
    unsigned wp_bytes;
    float* partial;        // [chunks][B][Cout]
    int B, K, Cout, CoutPad;
};

constexpr int kGemvChunk = 128;            // k per block = 32 slot rows: 64 KB of weights at CoutPad = 128

template <int MQ>
__global__ __launch_bounds__(256) void dense_gemv_f32_kernel(const DenseGemvArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* xs = reinterpret_cast<float*>(smem_raw);              // [MQ][128] activation chunk
    float* red = xs + MQ * kGemvChunk;                           // [MQ][128] second half's sums
    const int tid = threadIdx.x;
    const int n = blockIdx.y * 128 + (tid & 127), half = tid >> 7;   // two k-halves of 16 slot rows each
    const int k0 = blockIdx.x * kGemvChunk;
    for (int e = tid; e < MQ * kGemvChunk; e += 256) {
        const int m = e / kGemvChunk, k = e - m * kGemvChunk;
        xs[e] = (m < p.B && k0 + k < p.K) ? p.x[(long long)m * p.K + k0 + k] : 0.f;
    }
    const buffer_rsrc wbuf = make_buffer(p.wp, p.wp_bytes);
    f32x4 w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {                               // all 16 loads in flight before the first use
        const int slot = (k0 >> 2) + half * 16 + j;
        w[j] = buffer_load4(wbuf, (slot * 4 < p.K && n < p.CoutPad) ? (unsigned)((slot * p.CoutPad + n) * 16) : kOobOffset);
    }
    __syncthreads();
    float acc[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m) acc[m] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int kk = (half * 16 + j) * 4;
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + m * kGemvChunk + kk);   // broadcast read
            acc[m] = fmaf(xv.x, w[j].x, acc[m]);
            acc[m] = fmaf(xv.y, w[j].y, acc[m]);
            acc[m] = fmaf(xv.z, w[j].z, acc[m]);
            acc[m] = fmaf(xv.w, w[j].w, acc[m]);
        }
    }
    if (half == 1) {
#pragma unroll
        for (int m = 0; m < MQ; ++m) red[m * 128 + (tid & 127)] = acc[m];
    }
    __syncthreads();
    if (half == 0 && n < p.Cout) {
#pragma unroll
        for (int m = 0; m < MQ; ++m)
            if (m < p.B) p.partial[((long long)blockIdx.x * p.B + m) * p.Cout + n] = acc[m] + red[m * 128 + (tid & 127)];
    }
}

}  // namespace aae
