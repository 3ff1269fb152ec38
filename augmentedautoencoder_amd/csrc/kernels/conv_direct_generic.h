// Shape-agnostic direct convolution / dense on the vector ALUs.
//
// Correctness fallback for encoder configurations the matrix-core kernels do
// not cover (Cin not a multiple of 32 after the first layer, first-layer
// kernel size / channel count without a conv_first instantiation, flatten size
// not a multiple of 32).  Same semantics as the fast kernels: NHWC, HWIO,
// TF 'SAME' zero padding, bias -> ReLU -> inference batch-norm
// (/root/reference/auto_pose/ae/encoder.py:41-52, 62-66).  One thread per
// output element, k-ordered fp32 fma chain (kh, kw, ci ascending).
#pragma once

namespace aae {

struct ConvDirectArgs {
    const void* x;          // [B,H,W,Cin] float32 (or uint8 when IN_U8)
    const float* lut;
    const float* w;         // HWIO
    const float* bias;
    const float* bn_scale;
    const float* bn_shift;
    float* out;             // [B,Ho,Wo,Cout]
    int H, W, Cin, Ho, Wo, Cout, KS, S, pt, pl, relu;
    long long total;        // B*Ho*Wo*Cout
};

template <bool IN_U8>
__global__ __launch_bounds__(256) void conv_direct_generic_kernel(const ConvDirectArgs p) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < p.total; e += stride) {
        const int co = (int)(e % p.Cout);
        long long m = e / p.Cout;
        const int ow = (int)(m % p.Wo); m /= p.Wo;
        const int oh = (int)(m % p.Ho);
        const long long b = m / p.Ho;
        float acc = 0.f;
        for (int kh = 0; kh < p.KS; ++kh) {
            const int ih = oh * p.S - p.pt + kh;
            if ((unsigned)ih >= (unsigned)p.H) continue;
            for (int kw = 0; kw < p.KS; ++kw) {
                const int iw = ow * p.S - p.pl + kw;
                if ((unsigned)iw >= (unsigned)p.W) continue;
                const long long xo = ((b * p.H + ih) * p.W + iw) * p.Cin;
                const float* wp = p.w + ((long long)(kh * p.KS + kw) * p.Cin) * p.Cout + co;
                for (int ci = 0; ci < p.Cin; ++ci) {
                    float xv;
                    if (IN_U8) xv = p.lut[reinterpret_cast<const unsigned char*>(p.x)[xo + ci]];
                    else xv = reinterpret_cast<const float*>(p.x)[xo + ci];
                    acc = fmaf(xv, wp[(long long)ci * p.Cout], acc);
                }
            }
        }
        acc += p.bias[co];
        if (p.relu) acc = fmaxf(acc, 0.f);
        if (p.bn_scale) acc = acc * p.bn_scale[co] + p.bn_shift[co];
        p.out[e] = acc;
    }
}

}  // namespace aae
