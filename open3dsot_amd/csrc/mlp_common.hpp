// mlp_common.hpp -- pieces shared by the fp32-MFMA kernels (mlp.hip, mlp_direct.hip)
#pragma once
#include "o3d_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// accumulator register r of a 32x32 tile holds row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Reduce-scatter of 32 per-lane values over the 32 lanes of each half-wave: afterwards lane
// (l&31) holds in x[0] the sum over those 32 lanes of the value with index (l&31).
__device__ __forceinline__ void reduce_scatter32(float (&x)[32], int l31) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        const bool up = (l31 & m) != 0;
#pragma unroll
        for (int v = 0; v < m; ++v) {
            // operands made opaque first: otherwise LLVM rewrites select(up, x[v+m], x[v]) into a
            // dynamically indexed extract of the 32-wide array (a 32-way v_cndmask chain per access)
            float lo = x[v], hi = x[v + m];
            asm volatile("" : "+v"(lo), "+v"(hi));
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            x[v] = keep + __shfl_xor(send, m, 64);
        }
    }
}
