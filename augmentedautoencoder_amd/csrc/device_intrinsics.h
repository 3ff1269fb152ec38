// gfx950 wave-level primitives used by the kernels under csrc/kernels/.
// Included (after <hip/hip_runtime.h>) by the product translation unit only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define AAE_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

namespace aae {

// v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, exact fp32 fma chain.
// lane l: a = A[l&31][l>>5], b = B[l>>5][l&31];
// c/d reg r: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31.
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int shfl_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }

}  // namespace aae
