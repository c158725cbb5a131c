// mlp_direct.hip -- the aligned fast path of the grouped-MLP GEMMs: fp32 MFMA fed straight from
// global memory, no LDS staging and no barriers.
//
// Replaces (for Cin % 16 == 0, Cout % 64 == 0, P % 128 == 0 -- every inner layer of the trackers)
// the LDS-staged conv_fwd / conv_dgrad kernels of mlp.hip; same C-ABI semantics, same numbers
// (the contraction order over k is identical: bit-identical outputs, tests/test_fused_gpu.py).
//
// Why no LDS: v_mfma_f32_32x32x2_f32 takes 64 cycles per instruction, so operand bandwidth is a
// non-issue (one dword of A and one of B per lane per MFMA); what costs time is every cycle the
// matrix pipe is NOT issued to.  Measured on MI355X (tools/exp/direct_fwd.hip, rocprofv3 SQ
// counters in profiles/): the LDS-staged kernel parks 35 % of its wave-cycles in s_waitcnt /
// s_barrier (one workgroup barrier per 32 MFMAs) and reaches 89 TFLOP/s on the 256->256 layer;
// a wave that owns its tile end to end reaches 105-108.  Layout trick that makes direct loads
// coalesced: a wave's 128 positions are split over the four 32-column MFMA tiles as
//     column l of n-tile t  <->  position 4*l + t
// so ONE global_load_dwordx4 per lane (4 consecutive positions of one k row, 512 contiguous bytes
// per half-wave) is the B fragment of all four n-tiles, and the accumulators of a row store back as
// float4.  The k permutation (MFMA step s consumes k = 8g+s on lanes 0-31 and 8g+4+s on lanes
// 32-63) turns the A fragment into one dwordx4 per lane per 8 k's as well.  Weights (<= 256 KB) and
// the per-channel constants live in L1/L2; the Cout/64 waves that share a position tile are
// neighbours in the grid (same workgroup or adjacent workgroups).
//
// Software pipeline: ring of 2 fragment sets per wave, loop unrolled by 2, loads clamped instead of
// predicated (branch-free, so the compiler can count vmcnt exactly), sched_barriers keep the
// prefetch ahead of the MFMAs; 2 waves per SIMD (<= 256 VGPRs incl. 128 accumulators).
#include "mlp_common.hpp"

namespace {

constexpr int DT_POS = 128;   // positions per wave tile
constexpr int DT_M = 64;      // output rows per wave tile

// raw B operand of one group of 8 k's (this lane: 4 of them) x 4 consecutive positions
struct RawB {
    float4 v[4];      // the main tensor (X, or dN)
    float4 y[4];      // DY: raw conv output Y
    float4 c1, c2, c3;  // per-k constants: (scale, shift, -) or (A1, A2, A3)
};
// pooled dY source: v[s].xy = {dOut masked by out > 0, bits(arg)} of this lane's ball for k row s

enum BMode { B_PLAIN = 0, B_XFORM = 1, B_DY = 2, B_DYPOOL = 3 };

struct DirectArgs {
    const float* A;        // (M, K) row-major: W for forward, W^T for the data gradient
    const float* X;        // B main tensor (B, K, P): X / dN (NULL when pooled)
    const float* Y;        // DY modes: raw output of this layer (B, K, P)
    const float* c1; const float* c2; const float* c3;   // per-k constants (K each)
    const float2* pk; int ns;   // pooled source (B, K, P/ns): {dOut masked by out > 0, bits(arg)}
    float* Out;            // (B, M, P)
    int M, K, P, B;
    // compact (distinct-neighbour) layout, csrc/compact.hip: per-position weights and the live column count
    const float* w;        // (P) or NULL: weights of the statistics (forward) / of the A2*Y+A3 term (DY modes)
    const int32_t* meta;   // device int: positions >= meta[0] are dead (tiles beyond it return) or NULL
    // epilogue
    float* part;           // [B*P/128][2][M] or NULL
    const float* stat_c;   // forward: shift of the second moment
    const float* Yprev; const float* scale_p; const float* shift_p; const float* mean_p;  // dgrad mask
};

template <int MODE>
__device__ __forceinline__ void load_b(const DirectArgs& a, const float* xb, const float* yb, long rowP, int kb,
                                       long pool_base, int np, RawB& f) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (MODE != B_DYPOOL) f.v[s] = *reinterpret_cast<const float4*>(xb + (long)(kb + s) * rowP);
        if (MODE >= B_DY) f.y[s] = *reinterpret_cast<const float4*>(yb + (long)(kb + s) * rowP);
        if (MODE == B_DYPOOL) {
            const long i = pool_base + (long)(kb + s) * np;
            const float2 t = a.pk[i];
            f.v[s].x = t.x; f.v[s].y = t.y;
        }
    }
    if (MODE != B_PLAIN) {
        f.c1 = *reinterpret_cast<const float4*>(a.c1 + kb);
        f.c2 = *reinterpret_cast<const float4*>(a.c2 + kb);
        if (MODE >= B_DY) f.c3 = *reinterpret_cast<const float4*>(a.c3 + kb);
    }
}

// one group: 8 k's (4 per half-wave) x (2 m-tiles x 4 n-tiles) = 32 MFMAs
template <int MODE>
__device__ __forceinline__ void compute_group(const RawB& f, const float4& a0v, const float4& a1v,
                                              int kk, const float (&wv)[4], f32x16 (&acc)[2][4]) {
    const float a0[4] = {a0v.x, a0v.y, a0v.z, a0v.w}, a1[4] = {a1v.x, a1v.y, a1v.z, a1v.w};
    const float c1[4] = {f.c1.x, f.c1.y, f.c1.z, f.c1.w}, c2[4] = {f.c2.x, f.c2.y, f.c2.z, f.c2.w};
    const float c3[4] = {f.c3.x, f.c3.y, f.c3.z, f.c3.w};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float bv[4];
        if (MODE == B_DYPOOL) {
            const float go = f.v[s].x;
            const int ak = __float_as_int(f.v[s].y);
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[t] = (kk + t == ak) ? go : 0.f;
        } else {
            bv[0] = f.v[s].x; bv[1] = f.v[s].y; bv[2] = f.v[s].z; bv[3] = f.v[s].w;
        }
        if (MODE == B_XFORM) {
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[t] = fmaxf(fmaf(bv[t], c1[s], c2[s]), 0.f);
        } else if (MODE >= B_DY) {
            const float y[4] = {f.y[s].x, f.y[s].y, f.y[s].z, f.y[s].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[t] = fmaf(c1[s], bv[t], wv[t] * fmaf(c2[s], y[t], c3[s]));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0][t] = mfma32(a0[s], bv[t], acc[0][t]);
            acc[1][t] = mfma32(a1[s], bv[t], acc[1][t]);
        }
    }
}

// EPI 0: forward (store raw output, statistics {sum y, sum (y-c)^2})
// EPI 1: data gradient (mask by the producer's ReLU, store, statistics {sum g, sum g*(yprev-mean)})
template <int WAVES, int MODE, int EPI>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void direct_gemm_kernel(DirectArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_per_b = a.P / DT_POS;
    const int tile = blockIdx.x;
    const int b = tile / tiles_per_b, p0 = (tile - b * tiles_per_b) * DT_POS;
    if (a.meta && (long)tile * DT_POS >= a.meta[0]) return;     // compact layout: dead tile
    const int m0 = (blockIdx.y * WAVES + wave) * DT_M;
    const int p = p0 + 4 * l31;
    float wv[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.w) {
        const float4 t4 = *reinterpret_cast<const float4*>(a.w + (long)b * a.P + p);
        wv[0] = t4.x; wv[1] = t4.y; wv[2] = t4.z; wv[3] = t4.w;
    }
    const long rowP = a.P;
    const float* xb = (MODE != B_DYPOOL) ? a.X + (long)b * a.K * rowP + p : nullptr;
    const float* yb = (MODE >= B_DY) ? a.Y + (long)b * a.K * rowP + p : nullptr;
    const float* wa = a.A + (long)(m0 + l31) * a.K + 4 * h;
    const long wstep = 32L * a.K;
    int np = 1, kk = 0;
    long pool_base = 0;
    if (MODE == B_DYPOOL) {
        np = a.P / a.ns;
        const int j = p / a.ns;
        kk = p - j * a.ns;
        pool_base = (long)b * a.K * np + j;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    const int G = a.K / 8;       // even (K % 16 == 0)
    RawB f[2];
    float4 wa0[2], wa1[2];
    auto load = [&](int st, int g) {
        const int kb = 8 * g + 4 * h;
        load_b<MODE>(a, xb, yb, rowP, kb, pool_base, np, f[st]);
        wa0[st] = *reinterpret_cast<const float4*>(wa + 8 * g);
        wa1[st] = *reinterpret_cast<const float4*>(wa + 8 * g + wstep);
    };
    load(0, 0);
    for (int g = 0; g < G; g += 2) {
        load(1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_group<MODE>(f[0], wa0[0], wa1[0], kk, wv, acc);
        __builtin_amdgcn_sched_barrier(0);
        load(0, g + 2 < G ? g + 2 : G - 1);       // tail: harmless re-load of the last group
        __builtin_amdgcn_sched_barrier(0);
        compute_group<MODE>(f[1], wa0[1], wa1[1], kk, wv, acc);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---------------- epilogue: store, then the two statistics one at a time (32 live values each)
    float red[32];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * i + acc_row(r, h);
            const long o = ((long)b * a.M + m) * rowP + p;
            float4 v = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
            if (EPI == 1) {
                const float4 yp = *reinterpret_cast<const float4*>(a.Yprev + o);
                const float sc = a.scale_p[m], sf = a.shift_p[m];
                v.x = fmaf(yp.x, sc, sf) > 0.f ? v.x : 0.f;
                v.y = fmaf(yp.y, sc, sf) > 0.f ? v.y : 0.f;
                v.z = fmaf(yp.z, sc, sf) > 0.f ? v.z : 0.f;
                v.w = fmaf(yp.w, sc, sf) > 0.f ? v.w : 0.f;
                acc[i][0][r] = v.x; acc[i][1][r] = v.y; acc[i][2][r] = v.z; acc[i][3][r] = v.w;
            }
            *reinterpret_cast<float4*>(a.Out + o) = v;
            red[i * 16 + r] = EPI == 0 ? fmaf(wv[0], v.x, fmaf(wv[1], v.y, fmaf(wv[2], v.z, wv[3] * v.w)))
                                       : (v.x + v.y) + (v.z + v.w);
        }
    if (!a.part) return;
    // lane l31 of half h ends up owning value index l31 -> (i = l31>>4, r = l31&15)
    float* dst = a.part + (long)tile * 2 * a.M + m0 + 32 * (l31 >> 4) + acc_row(l31 & 15, h);
    reduce_scatter32(red, l31);
    dst[0] = red[0];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * i + acc_row(r, h);
            const float4 v = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
            if (EPI == 0) {
                const float c = a.stat_c ? a.stat_c[m] : 0.f;
                red[i * 16 + r] = wv[0] * (v.x - c) * (v.x - c) + wv[1] * (v.y - c) * (v.y - c) +
                                  wv[2] * (v.z - c) * (v.z - c) + wv[3] * (v.w - c) * (v.w - c);
            } else {      // masked entries of v are zero, so the mask needs no second evaluation
                const float4 yp = *reinterpret_cast<const float4*>(a.Yprev + ((long)b * a.M + m) * rowP + p);
                const float mu = a.mean_p[m];
                red[i * 16 + r] = v.x * (yp.x - mu) + v.y * (yp.y - mu) + v.z * (yp.z - mu) + v.w * (yp.w - mu);
            }
        }
    reduce_scatter32(red, l31);
    dst[a.M] = red[0];
}

// All M/64 row slabs of a position tile run as the waves of ONE workgroup: they read the same B rows,
// so those come from HBM once and from L1/L2 for the other slabs (rocprofv3 FETCH_SIZE of the
// two-workgroup variant showed every slab re-reading HBM: 604 MB instead of 350 MB per launch).
template <int MODE, int EPI>
int launch_direct(const DirectArgs& a, hipStream_t st) {
    const int tiles = a.B * (a.P / DT_POS);
    const int slabs = a.M / DT_M;
    if (slabs % 4 == 0) {
        hipLaunchKernelGGL((direct_gemm_kernel<4, MODE, EPI>), dim3(tiles, slabs / 4), dim3(256), 0, st, a);
    } else if (slabs % 2 == 0) {
        hipLaunchKernelGGL((direct_gemm_kernel<2, MODE, EPI>), dim3(tiles, slabs / 2), dim3(128), 0, st, a);
    } else {
        hipLaunchKernelGGL((direct_gemm_kernel<1, MODE, EPI>), dim3(tiles, slabs), dim3(64), 0, st, a);
    }
    return o3d_launch_status();
}

}  // namespace

bool o3d_direct_ok(int M, int K, int P) { return M % DT_M == 0 && K % 16 == 0 && P % DT_POS == 0; }

// forward: Y = W . f(X), see o3d_mlp_conv_fwd
int o3d_direct_fwd(const float* X, const float* W, const float* in_scale, const float* in_shift, int B, int Cin,
                   int Cout, int P, float* Y, float* part, const float* stat_c, const float* w, const int32_t* meta,
                   hipStream_t st) {
    DirectArgs a = {};
    a.w = w; a.meta = meta;
    a.A = W; a.X = X; a.c1 = in_scale; a.c2 = in_shift; a.Out = Y; a.M = Cout; a.K = Cin; a.P = P; a.B = B;
    a.part = part; a.stat_c = stat_c; a.ns = 4;
    return in_scale ? launch_direct<B_XFORM, 0>(a, st) : launch_direct<B_PLAIN, 0>(a, st);
}

// data gradient with the weights already transposed: Wt (Cin, Cout); see o3d_mlp_conv_dgrad
int o3d_direct_dgrad(const float* dN, const float* pk, int ns,
                     const float* Y, const float* A1, const float* A2, const float* A3, const float* Wt, int B,
                     int Cin, int Cout, int P, const float* Yprev, const float* scale_p, const float* shift_p,
                     const float* mean_p, float* dNprev, float* part, const float* w, const int32_t* meta,
                     hipStream_t st) {
    DirectArgs a = {};
    a.w = w; a.meta = meta;
    a.A = Wt; a.X = dN; a.Y = Y; a.c1 = A1; a.c2 = A2; a.c3 = A3; a.pk = reinterpret_cast<const float2*>(pk); a.ns = ns;
    a.Out = dNprev; a.M = Cin; a.K = Cout; a.P = P; a.B = B; a.part = part;
    a.Yprev = Yprev; a.scale_p = scale_p; a.shift_p = shift_p; a.mean_p = mean_p;
    return dN ? launch_direct<B_DY, 1>(a, st) : launch_direct<B_DYPOOL, 1>(a, st);
}
