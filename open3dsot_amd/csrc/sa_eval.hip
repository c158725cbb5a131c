// sa_eval.hip -- the tracking-inference set abstraction: gather + three 1x1 convolutions with BatchNorm on running
// statistics + ReLU + max over the ball, in ONE kernel.
//
// The reference tracks one frame at a time (models/base_model.py:59-86: batch 1, eval mode), so an SA call is a few
// thousand ball slots and the step is pure launch latency: the training path's per-layer kernels (compaction, expand,
// GEMM, statistics finalize, pool) are ~12 launches per call.  In eval mode BatchNorm is a fixed per-channel affine map
// (pointnet2/utils/pytorch_utils.py:56-59 with training=False), so nothing forces a grid-wide dependency between the
// layers: a workgroup takes 32 ball slots through all three layers with the activations in LDS.
//   layer 0   on the points, as everywhere in this library: Z = W0.[xyz ; feats] is one small GEMM over the N points
//             (csrc/mlp_direct.hip); here Y0[c, slot] = Z[c, idx(slot)] - W0[c, 0:3].centre(ball) is a gather
//   layer 1-2 fp32 MFMA 32x32x2: A fragments (weights) stream from L2, B fragments (activations) from LDS laid out
//             [slot][channel] so a lane's four k's are one ds_read_b128; BatchNorm + ReLU applied in the epilogue
//   pool      max over the ball's nsample (<= 32, power of two) slots = xor-shuffles inside a 32-lane half wave
// Duplicate slots (ball_query's padding) are simply computed: max ignores them and at batch 1 the work is nothing.
#include <cstdint>
#include "mlp_common.hpp"

namespace {

struct SaEvalArgs {
    const float* Z; long ldz;             // (C0, ldz): W0 . [xyz ; feats] per point
    const int32_t* idx;                   // (B, np, ns) point index of every slot
    const float* centers;                 // (B*np, 3) ball centres (scaled like xyz) or NULL (no xyz channels)
    const float* W0; int ldw;             // layer-0 weights: columns 0..2 multiply the centre
    const float* v0; const float* v1; const float* v2;   // (4, C): mean, invstd, scale, shift of each layer
    const float* W1; const float* W2;     // (C1, C0), (C2, C1)
    int C0, C1, C2;
    int B, np, ns, ld; long pt_base;      // ld point columns per cloud in Z, first column of this set of clouds
    float* out;                           // (B, C2, np)
};

// one 1x1-conv layer on the workgroup's 32 slots: Ain [32][K+4] (LDS) -> POOL ? out : Aout [32][M+4] (LDS)
template <bool POOL>
__device__ __forceinline__ void sa_eval_layer(const SaEvalArgs& a, const float* __restrict__ Ain, int K,
                                              const float* __restrict__ W, const float* __restrict__ v, int M,
                                              float* __restrict__ Aout, long slot0, int lane, int wave) {
    const int l31 = lane & 31, h = lane >> 5;
    const float* scale = v + 2 * M;
    const float* shift = v + 3 * M;
    const float* ba = Ain + l31 * (K + 4) + 4 * h;
    const int G = K / 8;
    for (int mt = wave; mt < M / 32; mt += 4) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // the 16 rows of this lane: 32*mt + 8*q + 4*h + j -- their BatchNorm constants as four float4 each, loaded BEFORE the
        // k loop (per row in the epilogue they were 16 load -> wait round trips per tile)
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc4[q] = *reinterpret_cast<const float4*>(scale + mt * 32 + 8 * q + 4 * h);
            sh4[q] = *reinterpret_cast<const float4*>(shift + mt * 32 + 8 * q + 4 * h);
        }
        const float* wa = W + (long)(mt * 32 + l31) * K + 4 * h;
        // weights stream from L2 (~500 cycles) against 4 MFMAs (256 cycles) per group: three groups in flight
        // (a ring with STATIC slots, unrolled by 4: rotating registers that are targets of loads still in flight makes
        // the compiler wait for all of them)
        float4 w[4];
#pragma unroll
        for (int u = 0; u < 3; ++u) w[u] = *reinterpret_cast<const float4*>(wa + 8 * (u < G ? u : G - 1));
        for (int g = 0; g < G; g += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[(u + 3) & 3] = *reinterpret_cast<const float4*>(wa + 8 * (g + u + 3 < G ? g + u + 3 : G - 1));
                if (g + u < G) {
                    const float4 av = w[u];
                    const float4 bv = *reinterpret_cast<const float4*>(ba + 8 * (g + u));
                    acc = mfma32(av.x, bv.x, acc);
                    acc = mfma32(av.y, bv.y, acc);
                    acc = mfma32(av.z, bv.z, acc);
                    acc = mfma32(av.w, bv.w, acc);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + acc_row(r, h);
            const int q = r >> 2, j = r & 3;
            const float scv = j == 0 ? sc4[q].x : j == 1 ? sc4[q].y : j == 2 ? sc4[q].z : sc4[q].w;
            const float shv = j == 0 ? sh4[q].x : j == 1 ? sh4[q].y : j == 2 ? sh4[q].z : sh4[q].w;
            float y = fmaxf(fmaf(acc[r], scv, shv), 0.f);
            if (POOL) {       // max over the ball's ns (power of two <= 32) lanes: DPP inside a row of 16, one shuffle across
                if (a.ns > 1) y = fmaxf(y, dpp_f<0xB1>(y));
                if (a.ns > 2) y = fmaxf(y, dpp_f<0x4E>(y));
                if (a.ns > 4) y = fmaxf(y, dpp_f<0x141>(y));
                if (a.ns > 8) y = fmaxf(y, dpp_f<0x140>(y));
                if (a.ns > 16) y = fmaxf(y, __shfl_xor(y, 16, 64));
                if ((l31 & (a.ns - 1)) == 0) {
                    const long ball = (slot0 + l31) / a.ns;
                    const int b = (int)(ball / a.np), jb = (int)(ball - (long)b * a.np);
                    a.out[((long)b * M + m) * a.np + jb] = y;
                }
            } else {
                Aout[l31 * (M + 4) + m] = y;
            }
        }
    }
}

__global__ __launch_bounds__(256) void sa_eval_kernel(SaEvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* A0 = sm;
    float* A1 = sm + 32 * (a.C0 + 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long slot0 = (long)blockIdx.x * 32;
    {   // layer 0: gather + centre term + BatchNorm + ReLU -> A0[slot][channel]
        const int l = tid & 31;
        const long s = slot0 + l;
        const long ball = s / a.ns;
        const int b = (int)(ball / a.np);
        const long gp = a.pt_base + (long)b * a.ld + a.idx[s];
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (a.centers) { const float* c = a.centers + ball * 3; cx = c[0]; cy = c[1]; cz = c[2]; }
        const float* sc = a.v0 + 2 * a.C0;
        const float* sh = a.v0 + 3 * a.C0;
        for (int c = tid >> 5; c < a.C0; c += 8) {
            float y = a.Z[(long)c * a.ldz + gp];
            if (a.centers) {
                const float* w = a.W0 + (long)c * a.ldw;
                y -= fmaf(w[2], cz, fmaf(w[1], cy, w[0] * cx));
            }
            A0[l * (a.C0 + 4) + c] = fmaxf(fmaf(y, sc[c], sh[c]), 0.f);
        }
    }
    __syncthreads();
    sa_eval_layer<false>(a, A0, a.C0, a.W1, a.v1, a.C1, A1, slot0, lane, wave);
    __syncthreads();
    sa_eval_layer<true>(a, A1, a.C1, a.W2, a.v2, a.C2, nullptr, slot0, lane, wave);
}

}  // namespace

// out (B, C2, np) = max over nsample of relu(bn2(W2 . relu(bn1(W1 . relu(bn0(Y0)))))),
//   Y0[c, (b,j,k)] = Z[c, pt_base + b*ld + idx[b,j,k]] - W0[c*ldw + 0..2] . centers[b*np + j]      (centers may be NULL)
// BatchNorm on running statistics: v_l (4, C_l) = {mean, invstd, scale, shift} (o3d_bn_eval_consts).
// C0 % 8 == 0, C1 % 32 == 0, C2 % 32 == 0, ns a power of two <= 32, (B*np*ns) % 32 == 0.
extern "C" int o3d_sa_eval_fused(const float* Z, long ldz, const int32_t* idx, const float* centers, const float* W0,
                                 int ldw, const float* v0, const float* W1, const float* v1, const float* W2,
                                 const float* v2, int C0, int C1, int C2, int B, int np, int ns, int ld, long pt_base,
                                 float* out, void* stream) {
    const long slots = (long)B * np * ns;
    if (!Z || !idx || !v0 || !W1 || !v1 || !W2 || !v2 || !out || (centers && (!W0 || ldw < 3)) || C0 <= 0 || C0 % 8 ||
        C1 <= 0 || C1 % 32 || C2 <= 0 || C2 % 32 || B <= 0 || np <= 0 || ns < 1 || ns > 32 || (ns & (ns - 1)) || ld <= 0 ||
        pt_base < 0 || slots % 32 != 0 || ((uintptr_t)v1 & 15) || ((uintptr_t)v2 & 15))      // (v1, v2: read as float4)
        return O3D_EINVAL;
    SaEvalArgs a = {Z, ldz, idx, centers, W0, ldw, v0, v1, v2, W1, W2, C0, C1, C2, B, np, ns, ld, pt_base, out};
    const size_t lds = sizeof(float) * 32 * ((size_t)C0 + 4 + (size_t)C1 + 4);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(sa_eval_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return O3D_ELAUNCH;
    hipLaunchKernelGGL(sa_eval_kernel, dim3((unsigned)(slots / 32)), dim3(256), lds, o3d_stream(stream), a);
    return o3d_launch_status();
}
