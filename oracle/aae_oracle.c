/* Plain-C restatement of the AugmentedAutoencoder inference hot path.
 *
 * TEST INFRASTRUCTURE ONLY (checker, never shipped / never measured as product).
 * PARITY UNPINNED: see the header of oracle/reference_cpu.py -- the reference has
 * no golden vectors for this path and its arithmetic lives in TensorFlow.
 * This file is the third, independent implementation (direct nested loops, no
 * im2col, no BLAS) that the numpy and torch oracles must agree with.
 *
 * Reference lines followed (relative to /root/reference):
 *   conv+relu(+bn) .......... auto_pose/ae/encoder.py:41-52
 *   flatten + dense ......... auto_pose/ae/encoder.py:54,58-68
 *   l2_normalize ............ auto_pose/ae/codebook.py:27     (eps 1e-12, TF semantics)
 *   matmul transpose_b ...... auto_pose/ae/codebook.py:50
 *   argmax / upright ........ auto_pose/ae/codebook.py:64-68  (first index wins on ties)
 *
 * Build: make -C oracle      (gcc -O2 -shared -fPIC)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static void same_pad(int in, int k, int s, int *out, int *before)
{
    int o = (in + s - 1) / s;
    int total = (o - 1) * s + k - in;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;          /* the extra pixel is padded at the end */
}

/* NHWC input, HWIO kernel, 'same' padding, optional ReLU, then optional
 * inference batch-norm as per-channel scale/shift (applied AFTER the ReLU,
 * encoder.py:51-52).  Accumulation order: kh, kw, ci ascending, bias added last
 * (the k order of the GPU implicit GEMM; fp64 makes the order immaterial). */
#define CONV_IMPL(NAME, T)                                                                      \
void NAME(const T *x, int B, int H, int W, int C, const T *kernel, int KH, int KW, int CO,      \
          const T *bias, int stride, int relu, const T *bn_scale, const T *bn_shift, T *out)     \
{                                                                                               \
    int Ho, Wo, pt, pl;                                                                         \
    same_pad(H, KH, stride, &Ho, &pt);                                                          \
    same_pad(W, KW, stride, &Wo, &pl);                                                          \
    for (int b = 0; b < B; ++b)                                                                 \
      for (int oh = 0; oh < Ho; ++oh)                                                           \
        for (int ow = 0; ow < Wo; ++ow)                                                         \
          for (int co = 0; co < CO; ++co) {                                                     \
            T acc = 0;                                                                          \
            for (int kh = 0; kh < KH; ++kh) {                                                   \
              int ih = oh * stride - pt + kh;                                                   \
              if (ih < 0 || ih >= H) continue;                                                  \
              for (int kw = 0; kw < KW; ++kw) {                                                 \
                int iw = ow * stride - pl + kw;                                                 \
                if (iw < 0 || iw >= W) continue;                                                \
                const T *xp = x + (((size_t)b * H + ih) * W + iw) * C;                          \
                const T *kp = kernel + ((size_t)(kh * KW + kw) * C) * CO + co;                  \
                for (int ci = 0; ci < C; ++ci) acc += xp[ci] * kp[(size_t)ci * CO];             \
              }                                                                                 \
            }                                                                                   \
            acc += bias[co];                                                                    \
            if (relu && acc < 0) acc = 0;                                                       \
            if (bn_scale) acc = acc * bn_scale[co] + bn_shift[co];                              \
            out[(((size_t)b * Ho + oh) * Wo + ow) * CO + co] = acc;                             \
          }                                                                                     \
}
CONV_IMPL(aae_oracle_conv2d_f64, double)
CONV_IMPL(aae_oracle_conv2d_f32, float)

#define DENSE_IMPL(NAME, T)                                                                     \
void NAME(const T *x, int B, int F, const T *kernel, int J, const T *bias, T *z)                \
{                                                                                               \
    for (int b = 0; b < B; ++b)                                                                 \
      for (int j = 0; j < J; ++j) {                                                             \
        T acc = 0;                                                                              \
        for (int f = 0; f < F; ++f) acc += x[(size_t)b * F + f] * kernel[(size_t)f * J + j];    \
        z[(size_t)b * J + j] = acc + bias[j];                                                   \
      }                                                                                         \
}
DENSE_IMPL(aae_oracle_dense_f64, double)
DENSE_IMPL(aae_oracle_dense_f32, float)

/* tf.nn.l2_normalize(z, 1): z * rsqrt(max(sum(z*z), 1e-12)) */
void aae_oracle_l2_normalize_f64(const double *z, int B, int J, double *q)
{
    for (int b = 0; b < B; ++b) {
        double ss = 0;
        for (int j = 0; j < J; ++j) ss += z[(size_t)b * J + j] * z[(size_t)b * J + j];
        double inv = 1.0 / sqrt(ss > 1e-12 ? ss : 1e-12);
        for (int j = 0; j < J; ++j) q[(size_t)b * J + j] = z[(size_t)b * J + j] * inv;
    }
}

/* cos = q . E^T ; argmax over columns 0, col_stride, 2*col_stride, ... with the
 * FIRST maximal index winning (np.argmax semantics).  cs may be NULL. */
void aae_oracle_cos_argmax_f64(const double *q, int B, const float *E, int N, int J, int col_stride,
                               double *cs, int64_t *idx, double *best)
{
    if (col_stride < 1) col_stride = 1;
    for (int b = 0; b < B; ++b) {
        double bv = -INFINITY;
        int64_t bi = 0;
        for (int n = 0; n < N; ++n) {
            double acc = 0;
            for (int j = 0; j < J; ++j) acc += q[(size_t)b * J + j] * (double)E[(size_t)n * J + j];
            if (cs) cs[(size_t)b * N + n] = acc;
            if (n % col_stride == 0 && acc > bv) { bv = acc; bi = n; }
        }
        idx[b] = bi;
        if (best) best[b] = bv;
    }
}
