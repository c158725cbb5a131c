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

// ---- BatchNorm finalize arithmetic of one channel, shared by the finalize kernels (mlp.hip) and the split-K tile's in-kernel
// finalize (mlp_direct.hip): from the fp64 sums {sum y, sum (y-c)^2} resp. {sum g, sum g (y - mean)}.
// No fp64 division / square root: v_rsq_f32 + one fp64 Newton step (relative error ~1e-14), the reciprocal of the count once.
struct BnFwdOut { float mean, invstd, scale, shift, run_mean, run_var; };
__device__ __forceinline__ BnFwdOut bn_fwd_math(double s, double q, double count, double cs, float g, float bt, float eps,
                                                float momentum, float run_mean, float run_var) {
    const double ic = 1.0 / count;
    const double mean = s * ic;
    double var = q * ic - (mean - cs) * (mean - cs);
    if (var < 0.0) var = 0.0;
    const double v = var + (double)eps;
    double rd = (double)rsqrtf((float)v);
    rd = rd * (1.5 - 0.5 * v * rd * rd);
    BnFwdOut o;
    o.mean = (float)mean;
    o.invstd = (float)rd;
    o.scale = g * o.invstd;
    o.shift = bt - o.mean * o.scale;
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    o.run_mean = (1.f - momentum) * run_mean + momentum * o.mean;
    o.run_var = (1.f - momentum) * run_var + momentum * (float)unbiased;
    return o;
}
struct BnBwdOut { float dgamma, dbeta, a1, a2, a3; };
__device__ __forceinline__ BnBwdOut bn_bwd_math(double s, double q, double count, double g, double is, double mu) {
    const double a1 = g * is;
    const double ic = 1.0 / count;
    const double a2 = -a1 * is * is * q * ic;
    const double a3 = -a1 * s * ic - a2 * mu;
    BnBwdOut o;
    o.dbeta = (float)s; o.dgamma = (float)(q * is);
    o.a1 = (float)a1; o.a2 = (float)a2; o.a3 = (float)a3;
    return o;
}
