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
#include <type_traits>
#include "mlp_common.hpp"

namespace {

constexpr int DT_M = 64;      // output rows per wave tile; columns per wave tile = 32 * NT (NT = 4 or 2)

template <int NT>
__device__ __forceinline__ void ldv(float (&d)[NT], const float* p) {
    if constexpr (NT == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    } else if constexpr (NT == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        d[0] = t.x; d[1] = t.y;
    } else {
        d[0] = *p;
    }
}
template <int NT>
__device__ __forceinline__ void stv(float* p, const float (&d)[NT]) {
    if constexpr (NT == 4)      *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
    else if constexpr (NT == 2) *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]);
    else                        *p = d[0];
}

// raw B operand of one group of 8 k's (this lane: 4 of them) x NT consecutive positions
template <int NT>
struct RawB {
    float v[4][NT];   // the main tensor (X, or dN); pooled: v[s][0..1] = {dOut masked by out > 0, bits(arg)}
    float y[4][NT];   // DY: raw conv output Y
    float4 c1, c2, c3;  // per-k constants: (scale, shift, -) or (A1, A2, A3)
};

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
    const int32_t* meta;   // device ints, 4 per segment: meta[4*s] = live columns of segment s (or NULL)
    long start1;           // first column of segment 1 (0 = one segment).  A second segment is a second set
                           // of clouds pushed through the SAME weights with its OWN BatchNorm statistics
                           // (template and search branch of the backbone): its per-channel constants sit
                           // K (c1..c3) resp. M (stat_c, scale_p, shift_p, mean_p) floats after segment 0's
    // epilogue
    float* part;           // [B*P/(32*NT)][2][M] or NULL
    const float* stat_c;   // forward: shift of the second moment
    const float* Yprev; const float* scale_p; const float* shift_p; const float* mean_p;  // dgrad mask
    // EPI 2 (plain layer of a 1-D conv stack): out = acc + bias[m] + resid[m, p]   (either may be NULL)
    const float* bias;     // (M)
    const float* resid;    // (M, P), same layout as Out
    // EPI 3 (= EPI 0 plus) per-cloud bias cbias (M, cb_B): out[m, q] += cbias[m, q / cb_N] before the statistics -- a per-cloud
    // CONSTANT input channel block (the pooled feature SegPointNet broadcasts to every point, models/backbone/
    // pointnet.py:188-190) contributes W_b . pooled[b] to every column of cloud b: a bias, not 1024 GEMM rows
    const float* cbias; int cb_N, cb_B;
    // remainder tiles in finer units (mlp_common.hpp::tail_plan): resident column-tile slots of this launch (0: off) and the
    // first statistics row behind the launch's regular rows
    int tail_slots; long extra_row0;
};

template <int MODE, int NT>
__device__ __forceinline__ void load_b(const DirectArgs& a, const float* xb, const float* yb, long rowP, int kb,
                                       long pool_base, int np, RawB<NT>& f) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (MODE != B_DYPOOL) ldv<NT>(f.v[s], xb + (long)(kb + s) * rowP);
        if (MODE >= B_DY) ldv<NT>(f.y[s], yb + (long)(kb + s) * rowP);
        if constexpr (MODE == B_DYPOOL && NT >= 2) {
            const float2 t = a.pk[pool_base + (long)(kb + s) * np];
            f.v[s][0] = t.x; f.v[s][1] = t.y;
        }
    }
    if (MODE != B_PLAIN) {
        f.c1 = *reinterpret_cast<const float4*>(a.c1 + kb);
        f.c2 = *reinterpret_cast<const float4*>(a.c2 + kb);
        if (MODE >= B_DY) f.c3 = *reinterpret_cast<const float4*>(a.c3 + kb);
    }
}

// one group: 8 k's (4 per half-wave) x (MT m-tiles x NT n-tiles) = 4*MT*NT MFMAs
template <int MODE, int NT, int MT>
__device__ __forceinline__ void compute_group(const RawB<NT>& f, const float4& a0v, const float4& a1v, int kk,
                                              const float (&wv)[NT], f32x16 (&acc)[MT][NT]) {
    const float a0[4] = {a0v.x, a0v.y, a0v.z, a0v.w}, a1[4] = {a1v.x, a1v.y, a1v.z, a1v.w};
    const float c1[4] = {f.c1.x, f.c1.y, f.c1.z, f.c1.w}, c2[4] = {f.c2.x, f.c2.y, f.c2.z, f.c2.w};
    const float c3[4] = {f.c3.x, f.c3.y, f.c3.z, f.c3.w};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float bv[NT];
        if constexpr (MODE == B_DYPOOL && NT >= 2) {
            const float go = f.v[s][0];
            const int ak = __float_as_int(f.v[s][1]);
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = (kk + t == ak) ? go : 0.f;
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = f.v[s][t];
        }
        if (MODE == B_XFORM) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = fmaxf(fmaf(bv[t], c1[s], c2[s]), 0.f);
        } else if (MODE >= B_DY) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = fmaf(c1[s], bv[t], wv[t] * fmaf(c2[s], f.y[s][t], c3[s]));
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[0][t] = mfma32(a0[s], bv[t], acc[0][t]);
            if constexpr (MT == 2) acc[1][t] = mfma32(a1[s], bv[t], acc[1][t]);
        }
    }
}

// EPI 0: forward (store raw output, statistics {sum y, sum (y-c)^2})
// EPI 1: data gradient (mask by the producer's ReLU, store, statistics {sum g, sum g*(yprev-mean)})
// EPI 3: EPI 0 with a per-cloud bias added before the store and the statistics (its own instantiation: as a run-time
//        branch of EPI 0 it cost every forward launch a 64-bit division per row, 1600 register moves and 138 spills)
// EPI 2: plain store + per-row bias + residual tensor, no statistics (last layer of a Conv1d stack,
//        pytorch_utils.py:124-155 with bn=False / activation=None, and the input gradient of a stack)
// NT = 4: 128 columns per wave (one dwordx4 B load per k row); NT = 2: 64 columns (dwordx2) -- twice the waves
// of half the length for launches that do not fill the chip (everything after compaction at batch 48).
// MT = 2: 64 output rows per wave (MT = 1, 32 rows, is only used by the split-K tile further down).
template <int WAVES, int MODE, int EPI, int NT, int MT = 2>
__device__ __forceinline__ void direct_gemm_body(DirectArgs& a, const int bx, const int by, const long part_row = -1) {
    constexpr int POS = 32 * NT;
    constexpr int DT_MW = 32 * MT;      // output rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_per_b = a.P / POS;
    const int tile = bx;
    const int b = tile / tiles_per_b, p0 = (tile - b * tiles_per_b) * POS;
    if (a.meta) {                                             // compact layout: dead tile?  which segment?
        const long c0 = (long)tile * POS;
        const int seg = (a.start1 > 0 && c0 >= a.start1) ? 1 : 0;
        if (c0 - (seg ? a.start1 : 0) >= a.meta[4 * seg]) return;
        if (seg) {
            if (a.c1) a.c1 += a.K;
            if (a.c2) a.c2 += a.K;
            if (a.c3) a.c3 += a.K;
            if (a.stat_c) a.stat_c += a.M;
            if (a.scale_p) { a.scale_p += a.M; a.shift_p += a.M; a.mean_p += a.M; }
        }
    }
    const int m0 = (by * WAVES + wave) * DT_MW;
    const int p = p0 + NT * l31;
    float wv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wv[t] = 1.f;
    if (a.w) ldv<NT>(wv, a.w + (long)b * a.P + p);
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

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    // Ring of 2 fragment sets: one group of loads in flight ahead of the MFMAs.  A group of 8 k's is 8*MT*NT/2 MFMAs of 64
    // cycles; with 2 waves per SIMD and 64-row tiles one group ahead covers a ~2000-cycle miss.  (Deeper rings were measured
    // in rounds 2-3: 3-deep for the forward 7.17 -> 7.20 ms per step, 4-deep for 32-row tiles no change -- the small
    // launches those were meant for run on the split-K tile below since round 4.)
    const int G = a.K / 8;       // even (K % 16 == 0)
    RawB<NT> f[2];
    float4 wa0[2], wa1[2];
    auto load = [&](auto stc, int g) {
        constexpr int st = decltype(stc)::value;
        const int kb = 8 * g + 4 * h;
        load_b<MODE, NT>(a, xb, yb, rowP, kb, pool_base, np, f[st]);
        wa0[st] = *reinterpret_cast<const float4*>(wa + 8 * g);
        if constexpr (MT == 2) wa1[st] = *reinterpret_cast<const float4*>(wa + 8 * g + wstep);
        else wa1[st] = wa0[st];
    };
    {
        load(std::integral_constant<int, 0>{}, 0);
        for (int g = 0; g < G; g += 2) {
            load(std::integral_constant<int, 1>{}, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_group<MODE, NT, MT>(f[0], wa0[0], wa1[0], kk, wv, acc);
            __builtin_amdgcn_sched_barrier(0);
            load(std::integral_constant<int, 0>{}, g + 2 < G ? g + 2 : G - 1);       // tail: harmless re-load of the last group
            __builtin_amdgcn_sched_barrier(0);
            compute_group<MODE, NT, MT>(f[1], wa0[1], wa1[1], kk, wv, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---------------- epilogue, one 32-row m-tile at a time: every load of the m-tile (the producer's raw output for
    // the ReLU mask / the residual, the per-row constants as float4) is issued BEFORE the first dependent use --
    // written row by row the compiler serialises load -> s_waitcnt vmcnt(0) -> store 32 times (stores count in vmcnt
    // on gfx9), i.e. 32 exposed memory latencies per wave; then store + both statistics of a row in the same pass and
    // ONE 32-value reduce-scatter per m-tile (16 rows x {sum, second statistic})
    int cloud = 0;
    if constexpr (EPI == 3) cloud = (int)(((long)b * a.P + p0) / a.cb_N);
    const bool stats = EPI != 2 && a.part != nullptr;
    // per-row constants that do not come with a tensor load (statistics shift, bias, cloud bias): all rows of the wave
    // tile up front -- loaded per batch they put a wait (which on gfx9 also waits for the previous batch's stores)
    // in front of every batch of stores
    float4 kc1[MT][4], kc2[MT][4];
    if constexpr (EPI != 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mb = m0 + 32 * i + 4 * h + 8 * q;
                kc1[i][q] = kc2[i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (EPI == 2) {
                    if (a.bias) kc1[i][q] = *reinterpret_cast<const float4*>(a.bias + mb);
                } else {
                    if (a.stat_c) kc1[i][q] = *reinterpret_cast<const float4*>(a.stat_c + mb);
                    if constexpr (EPI == 3) {
                        kc2[i][q].x = a.cbias[(long)(mb + 0) * a.cb_B + cloud];
                        kc2[i][q].y = a.cbias[(long)(mb + 1) * a.cb_B + cloud];
                        kc2[i][q].z = a.cbias[(long)(mb + 2) * a.cb_B + cloud];
                        kc2[i][q].w = a.cbias[(long)(mb + 3) * a.cb_B + cloud];
                    }
                }
            }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float red[32];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {                       // 8 rows per batch: 16 at once spills the row addresses
            const int mb = m0 + 32 * i + 16 * hb + 4 * h;      // rows mb + 8*q + j, q < 2, j < 4
            const long ob = ((long)b * a.M + mb) * rowP + p;
            float ex[8][NT];                                   // EPI 1: Yprev rows, EPI 2: residual rows
            float4 k1[2], k2[2], k3[2];                        // per-row constants, 4 consecutive rows each
            if constexpr (EPI == 1 || EPI == 2) {
                const float* src = EPI == 1 ? a.Yprev : a.resid;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) ex[r][t] = 0.f;
                    if (EPI == 1 || src) ldv<NT>(ex[r], src + ob + (long)(8 * (r >> 2) + (r & 3)) * rowP);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                k3[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (EPI == 1) {
                    k1[q] = *reinterpret_cast<const float4*>(a.scale_p + mb + 8 * q);
                    k2[q] = *reinterpret_cast<const float4*>(a.shift_p + mb + 8 * q);
                    k3[q] = *reinterpret_cast<const float4*>(a.mean_p + mb + 8 * q);
                } else {
                    k1[q] = kc1[i][2 * hb + q];
                    k2[q] = kc2[i][2 * hb + q];
                }
            }
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int q = r8 >> 2, j = r8 & 3, r = 8 * hb + r8;      // accumulator register r <-> row 8*(r>>2) + (r&3)
                const float c1 = j == 0 ? k1[q].x : j == 1 ? k1[q].y : j == 2 ? k1[q].z : k1[q].w;
                const float c2 = j == 0 ? k2[q].x : j == 1 ? k2[q].y : j == 2 ? k2[q].z : k2[q].w;
                const float c3 = j == 0 ? k3[q].x : j == 1 ? k3[q].y : j == 2 ? k3[q].z : k3[q].w;
                float v[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) v[t] = acc[i][t][r];
                float s1 = 0.f, s2 = 0.f;
                if constexpr (EPI == 1) {          // mask by the producer's ReLU; {sum g, sum g*(yprev-mean)}
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        v[t] = fmaf(ex[r8][t], c1, c2) > 0.f ? v[t] : 0.f;
                        s1 += v[t];
                        s2 = fmaf(v[t], ex[r8][t] - c3, s2);
                    }
                } else if constexpr (EPI == 2) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) v[t] = (v[t] + c1) + ex[r8][t];
                } else {                           // {sum w*y, sum w*(y-c)^2}
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if constexpr (EPI == 3) v[t] += c2;
                        s1 = fmaf(wv[t], v[t], s1);
                        s2 += wv[t] * (v[t] - c1) * (v[t] - c1);
                    }
                }
                stv<NT>(a.Out + ob + (long)(8 * q + j) * rowP, v);
                red[r] = s1;
                red[16 + r] = s2;
            }
        }
        if (stats) {       // lane l31 of half h ends up owning value index l31 -> (statistic l31>>4, row r = l31&15)
            reduce_scatter32(red, l31);
            a.part[(part_row >= 0 ? part_row : (long)tile) * 2 * a.M + (long)(l31 >> 4) * a.M + m0 + 32 * i + acc_row(l31 & 15, h)] = red[0];
        }
    }
}


template <int WAVES, int MODE, int EPI, int NT, int MT = 2>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, (NT == 4 && MT == 2) ? 2 : 4)))
void direct_gemm_kernel(DirectArgs a) {
    direct_gemm_body<WAVES, MODE, EPI, NT, MT>(a, blockIdx.x, blockIdx.y);
}

// direct_gemm_kernel<WAVES, MODE, EPI, 4, 2> with the remainder tiles of every segment cut into 64- or 32-column blocks
// (mlp_common.hpp::tail_plan): workgroup index -> (tile, block) from the device-side live count; a block is the same body
// instantiated for 2 or 1 n-tiles on contiguous columns (NT = 1: lane l <-> column p0 + l).  Compact layout only (meta != NULL).
template <int WAVES, int MODE, int EPI>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void direct_gemm_tail_kernel(DirectArgs a) {
    const int tile = blockIdx.x;                                     // 128-column tile index over the whole operand
    const int seg = (a.start1 > 0 && (long)tile * 128 >= a.start1) ? 1 : 0;
    const int t0 = seg ? (int)(a.start1 / 128) : 0;                  // first tile of the segment
    const int cap = a.start1 > 0 ? (seg ? (int)((a.P - a.start1) / 128) : (int)(a.start1 / 128)) : a.P / 128;   // worst-case tiles of the segment
    const int T = a.meta[4 * seg] / 128;
    const TailPlan pl = tail_plan(T, a.tail_slots, cap);
    const int tl = tile - t0;
    if (tl < pl.full) { direct_gemm_body<WAVES, MODE, EPI, 4, 2>(a, tile, blockIdx.y); return; }
    const int u = tl - pl.full;
    if (u >= pl.R * pl.f) return;                                    // dead workgroup (pl.f == 1: R = 0)
    const int j = u / pl.f, blk = u - j * pl.f;
    const int tt = t0 + pl.full + j;                                 // the tail tile
    const long row = blk == 0 ? (long)tt : a.extra_row0 + (long)seg * a.tail_slots + (long)j * (pl.f - 1) + (blk - 1);
    if (pl.f == 4) direct_gemm_body<WAVES, MODE, EPI, 1, 2>(a, tt * 4 + blk, blockIdx.y, row);
    else           direct_gemm_body<WAVES, MODE, EPI, 2, 2>(a, tt * 2 + blk, blockIdx.y, row);
}

// Two independent GEMMs of the same shape class (same columns, same operand / epilogue mode, same wave tile) in one
// launch, blockIdx.z selects: the 1-D conv stacks of the RPN that read the same seeds (FC_layer_cla and vote_layer,
// models/head/rpn.py:16-28,44-54) are serial chains of sub-round launches -- side by side their workgroups fill the chip
// together.  Rows beyond a problem's own M return at once.
template <int WAVES, int MODE, int EPI, int NT, int MT>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, NT == 4 ? 2 : 4)))
void direct_gemm_pair_kernel(DirectArgs a0, DirectArgs a1) {
    if (blockIdx.z == 0) {
        if ((int)blockIdx.y * WAVES * 32 * MT < a0.M) direct_gemm_body<WAVES, MODE, EPI, NT, MT>(a0, blockIdx.x, blockIdx.y);
    } else {
        if ((int)blockIdx.y * WAVES * 32 * MT < a1.M) direct_gemm_body<WAVES, MODE, EPI, NT, MT>(a1, blockIdx.x, blockIdx.y);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-K tile for the launch-latency-bound problems (round 4): the 1-D conv stacks of the heads (256 x 256 x 6 144: 5 us of
// matrix-pipe work that took 20-23 us as 768 one-wave-per-SIMD chains of 256 MFMAs behind a cold prologue), the per-point
// layer-0 GEMMs and the set-abstraction launches with few worst-case columns.  One workgroup = one 32-row x 64-column
// output tile; its FOUR WAVES EACH CONTRACT A QUARTER OF K (a quarter of the chain, four times the waves: 3 072 waves of 64
// MFMAs for the problem above, three per SIMD), the four partial tiles meet in LDS (32 KB), and wave w finishes rows
// [8w, 8w + 8) of the tile: BatchNorm statistics / ReLU mask / bias + residual exactly as direct_gemm_body's epilogues
// (same partial-row layout: one row per 64 columns).  Summation order over k: the four quarters in fixed order --
// deterministic, but not bitwise the unsplit kernel's.  Operand modes: B_PLAIN / B_XFORM / B_DY (no pooled sources).
constexpr int SK_ROWS = 32, SK_COLS = 64;

// 8 per-lane values over the 32 lanes of a half-wave: afterwards lane l31 < 8 holds the sum of value index l31
__device__ __forceinline__ float reduce_scatter8(float (&x)[8], int l31) {
#pragma unroll
    for (int m = 4; m >= 1; m >>= 1) {
        const bool up = (l31 & m) != 0;
#pragma unroll
        for (int v = 0; v < m; ++v) {
            float lo = x[v], hi = x[v + m];
            asm volatile("" : "+v"(lo), "+v"(hi));
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            x[v] = keep + __shfl_xor(send, m, 64);
        }
    }
    float r = x[0];
    r += __shfl_xor(r, 8, 64);
    r += __shfl_xor(r, 16, 64);
    return r;
}

template <int MODE, int EPI>
__device__ __forceinline__ void splitk_body(DirectArgs& a, const int bx, const int by, float* __restrict__ red) {
    static_assert(MODE <= B_DY && EPI <= 2, "split-K tile: plain / transformed / BatchNorm-backward operands only");
    constexpr int NT = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_per_b = a.P / SK_COLS;
    const int tile = bx;
    const int b = tile / tiles_per_b, p0 = (tile - b * tiles_per_b) * SK_COLS;
    if (a.meta) {                                             // compact layout: dead tile?  which segment?
        const long c0 = (long)tile * SK_COLS;
        const int seg = (a.start1 > 0 && c0 >= a.start1) ? 1 : 0;
        if (c0 - (seg ? a.start1 : 0) >= a.meta[4 * seg]) return;      // (the whole workgroup: before any barrier)
        if (seg) {
            if (a.c1) a.c1 += a.K;
            if (a.c2) a.c2 += a.K;
            if (a.c3) a.c3 += a.K;
            if (a.stat_c) a.stat_c += a.M;
            if (a.scale_p) { a.scale_p += a.M; a.shift_p += a.M; a.mean_p += a.M; }
        }
    }
    const int m0 = by * SK_ROWS;
    const int p = p0 + NT * l31;
    float wv[NT] = {1.f, 1.f};
    if (a.w) ldv<NT>(wv, a.w + (long)b * a.P + p);
    const long rowP = a.P;
    const float* xb = a.X + (long)b * a.K * rowP + p;
    const float* yb = (MODE >= B_DY) ? a.Y + (long)b * a.K * rowP + p : nullptr;
    const float* wa = a.A + (long)(m0 + l31) * a.K + 4 * h;

    // ---- what the epilogue needs besides the accumulators: requested now, consumed after the LDS exchange
    const int mrow = m0 + 8 * wave + 4 * h;                  // this lane's rows: mrow + j, j < 4; positions p, p + 1
    const long ob = ((long)b * a.M + mrow) * rowP + p;
    float ex[4][NT];
    float4 k1 = make_float4(0.f, 0.f, 0.f, 0.f), k2 = k1, k3 = k1;
    if constexpr (EPI == 1 || EPI == 2) {
        const float* src = EPI == 1 ? a.Yprev : a.resid;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ex[j][0] = ex[j][1] = 0.f;
            if (EPI == 1 || src) ldv<NT>(ex[j], src + ob + (long)j * rowP);
        }
    }
    if constexpr (EPI == 1) {
        k1 = *reinterpret_cast<const float4*>(a.scale_p + mrow);
        k2 = *reinterpret_cast<const float4*>(a.shift_p + mrow);
        k3 = *reinterpret_cast<const float4*>(a.mean_p + mrow);
    } else if constexpr (EPI == 2) {
        if (a.bias) k1 = *reinterpret_cast<const float4*>(a.bias + mrow);
    } else {
        if (a.stat_c) k1 = *reinterpret_cast<const float4*>(a.stat_c + mrow);
    }

    f32x16 acc[1][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][t][r] = 0.f;

    // this wave's share of the K / 16 pairs of 8-k groups (contiguous; the first K/16 % 4 waves take one pair more)
    const int npair = a.K / 16;
    const int q = npair / 4, rem = npair - 4 * q;
    const int ps = wave * q + (wave < rem ? wave : rem);
    const int gs = 2 * ps, ge = 2 * (ps + q + (wave < rem ? 1 : 0));
    RawB<NT> f[2];
    float4 wa0[2];
    auto load = [&](auto stc, int g) {
        constexpr int st = decltype(stc)::value;
        load_b<MODE, NT>(a, xb, yb, rowP, 8 * g + 4 * h, 0, 1, f[st]);
        wa0[st] = *reinterpret_cast<const float4*>(wa + 8 * g);
    };
    if (gs < ge) {
        load(std::integral_constant<int, 0>{}, gs);
        for (int g = gs; g < ge; g += 2) {
            load(std::integral_constant<int, 1>{}, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_group<MODE, NT, 1>(f[0], wa0[0], wa0[0], 0, wv, acc);
            __builtin_amdgcn_sched_barrier(0);
            load(std::integral_constant<int, 0>{}, g + 2 < ge ? g + 2 : ge - 1);       // tail: harmless re-load
            __builtin_amdgcn_sched_barrier(0);
            compute_group<MODE, NT, 1>(f[1], wa0[1], wa0[1], 0, wv, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the four partial tiles meet in LDS: red[wave][row][column], a lane's two positions as one 8-byte access
#pragma unroll
    for (int r = 0; r < 16; ++r)
        *reinterpret_cast<float2*>(red + ((wave * SK_ROWS + acc_row(r, h)) * SK_COLS + NT * l31)) =
            make_float2(acc[0][0][r], acc[0][1][r]);
    __syncthreads();
    float x8[8];
    const bool stats = EPI != 2 && a.part != nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * wave + 4 * h + j;
        float v[NT] = {0.f, 0.f};
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
            const float2 t = *reinterpret_cast<const float2*>(red + ((w4 * SK_ROWS + row) * SK_COLS + NT * l31));
            v[0] += t.x; v[1] += t.y;
        }
        const float c1 = j == 0 ? k1.x : j == 1 ? k1.y : j == 2 ? k1.z : k1.w;
        const float c2 = j == 0 ? k2.x : j == 1 ? k2.y : j == 2 ? k2.z : k2.w;
        const float c3 = j == 0 ? k3.x : j == 1 ? k3.y : j == 2 ? k3.z : k3.w;
        float s1 = 0.f, s2 = 0.f;
        if constexpr (EPI == 1) {          // mask by the producer's ReLU; {sum g, sum g*(yprev-mean)}
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                v[t] = fmaf(ex[j][t], c1, c2) > 0.f ? v[t] : 0.f;
                s1 += v[t];
                s2 = fmaf(v[t], ex[j][t] - c3, s2);
            }
        } else if constexpr (EPI == 2) {
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = (v[t] + c1) + ex[j][t];
        } else {                           // {sum w*y, sum w*(y-c)^2}
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                s1 = fmaf(wv[t], v[t], s1);
                s2 += wv[t] * (v[t] - c1) * (v[t] - c1);
            }
        }
        stv<NT>(a.Out + ob + (long)j * rowP, v);
        x8[j] = s1;
        x8[4 + j] = s2;
    }
    if (stats) {       // lane l31 < 8 of half h: statistic l31 >> 2 of row mrow + (l31 & 3)
        const float r = reduce_scatter8(x8, l31);
        if (l31 < 8) a.part[(long)tile * 2 * a.M + (long)(l31 >> 2) * a.M + mrow + (l31 & 3)] = r;
    }
}

template <int MODE, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4)))
void splitk_gemm_kernel(DirectArgs a) {
    __shared__ float red[4 * SK_ROWS * SK_COLS];
    splitk_body<MODE, EPI>(a, blockIdx.x, blockIdx.y, red);
}

template <int MODE, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4)))
void splitk_gemm_pair_kernel(DirectArgs a0, DirectArgs a1) {
    __shared__ float red[4 * SK_ROWS * SK_COLS];
    if (blockIdx.z == 0) {
        if ((int)blockIdx.y * SK_ROWS < a0.M) splitk_body<MODE, EPI>(a0, blockIdx.x, blockIdx.y, red);
    } else {
        if ((int)blockIdx.y * SK_ROWS < a1.M) splitk_body<MODE, EPI>(a1, blockIdx.x, blockIdx.y, red);
    }
}

// which problems take the split-K tile: 64-column partial rows (the caller's `tile`), K in whole pairs of 8-k groups with
// at least one pair per wave, and at most splitk_max() output elements in the worst case (beyond that the 8 row slabs of a
// column tile re-reading their B panel from L2 costs more than the shorter chains gain: HISTORY.md section 7c)
// (measured, same-box A/B of a BAT step: 0 -> 5.73 / 5.76 ms, 4 M (the heads only) -> 5.60, 16 M -> 5.50 / 5.54)
static long splitk_max() { return 16L << 20; }
static bool splitk_ok(const DirectArgs& a, int tile) {
    return tile == 64 && a.K % 16 == 0 && a.K >= 64 && a.M % SK_ROWS == 0 && a.P % SK_COLS == 0 && !a.pk &&
           (long)a.M * a.B * a.P <= splitk_max();
}
template <int MODE, int EPI>
int launch_splitk(const DirectArgs& a, hipStream_t st) {
    if constexpr (MODE <= B_DY && EPI <= 2) {
        hipLaunchKernelGGL((splitk_gemm_kernel<MODE, EPI>), dim3(a.B * (a.P / SK_COLS), a.M / SK_ROWS), dim3(256), 0, st, a);
        return o3d_launch_status();
    } else {
        return O3D_EINVAL;
    }
}

// All M/64 row slabs of a position tile run as the waves of ONE workgroup: they read the same B rows,
// so those come from HBM once and from L1/L2 for the other slabs (rocprofv3 FETCH_SIZE of the
// two-workgroup variant showed every slab re-reading HBM: 604 MB instead of 350 MB per launch).
template <int MODE, int EPI, int NT, int MT = 2>
int launch_direct_nt(const DirectArgs& a, hipStream_t st) {
    const int tiles = a.B * (a.P / (32 * NT));
    const int slabs = a.M / (32 * MT);
    if constexpr (NT == 4 && MT == 2 && MODE != B_DYPOOL && EPI <= 1) {
        if (a.tail_slots > 0 && a.meta && a.B == 1) {      // remainder tiles in finer units (direct_gemm_tail_kernel)
            if (slabs % 4 == 0)      hipLaunchKernelGGL((direct_gemm_tail_kernel<4, MODE, EPI>), dim3(tiles, slabs / 4), dim3(256), 0, st, a);
            else if (slabs % 2 == 0) hipLaunchKernelGGL((direct_gemm_tail_kernel<2, MODE, EPI>), dim3(tiles, slabs / 2), dim3(128), 0, st, a);
            else                     hipLaunchKernelGGL((direct_gemm_tail_kernel<1, MODE, EPI>), dim3(tiles, slabs), dim3(64), 0, st, a);
            return o3d_launch_status();
        }
    }
    if (slabs % 4 == 0) {
        hipLaunchKernelGGL((direct_gemm_kernel<4, MODE, EPI, NT, MT>), dim3(tiles, slabs / 4), dim3(256), 0, st, a);
    } else if (slabs % 2 == 0) {
        hipLaunchKernelGGL((direct_gemm_kernel<2, MODE, EPI, NT, MT>), dim3(tiles, slabs / 2), dim3(128), 0, st, a);
    } else {
        hipLaunchKernelGGL((direct_gemm_kernel<1, MODE, EPI, NT, MT>), dim3(tiles, slabs), dim3(64), 0, st, a);
    }
    return o3d_launch_status();
}

static int pw_tile(long P, int M) { return o3d_direct_tile(P, M, 0); }

template <int MODE, int EPI>
int launch_direct_small(const DirectArgs& a, int tile, hipStream_t st) {
    if constexpr (MODE <= B_DY && EPI <= 2) {
        if (splitk_ok(a, tile)) return launch_splitk<MODE, EPI>(a, st);
    }
    // Narrow outputs (64 / 128 rows) over many columns -- M2-Track's per-point layers, 98 304 columns: HBM-bound, and as 64-row
    // waves only 768 / 1 536 of them, one per SIMD (0.22-0.53 of the HBM roof).  32-row waves of the same 128 columns: twice the
    // waves, 64 accumulator registers (four per SIMD fit), the second reader of a B panel hits L1/L2.  Same box, M2-Track step:
    // 6.487 -> 6.439 ms; on the compact launches of BAT (launch_direct below) the same rule measured +0.02 / 0.00 ms: not there.
    if constexpr (MODE != B_DYPOOL) {
        if (tile == 128 && a.M <= 128) return launch_direct_nt<MODE, EPI, 4, 1>(a, st);
    }
    return tile == 64 ? launch_direct_nt<MODE, EPI, 2>(a, st) : launch_direct_nt<MODE, EPI, 4>(a, st);
}

template <int MODE, int EPI>
int launch_direct(const DirectArgs& a, int tile, hipStream_t st) {
    if constexpr (MODE <= B_DY && EPI <= 2) {
        if (splitk_ok(a, tile)) return launch_splitk<MODE, EPI>(a, st);
    }
    return tile == 64 ? launch_direct_nt<MODE, EPI, 2>(a, st) : launch_direct_nt<MODE, EPI, 4>(a, st);
}

}  // namespace

bool o3d_direct_ok(int M, int K, int P) { return M % DT_M == 0 && K % 16 == 0 && P % 128 == 0; }

static int g_tail_override = -1;
// TEST hook: -1 = the resident-slot count of the device (default), 0 = no remainder split at all (what the split is tested and
// A/B-ed against), S > 0 = pretend S slots whatever the shape (small problems then meet every branch of tail_plan)
extern "C" int o3d_direct_tail_override(int slots) { g_tail_override = slots; return O3D_OK; }

extern "C" int o3d_direct_tail_slots(int M) {
    if (g_tail_override >= 0) return (M > 0 && M % DT_M == 0) ? g_tail_override : 0;
    // compute units of the CURRENT device, cached per device (round-5 advisor: one static cached the first caller's device)
    static int cus_of[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus_of[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus_of[dev] = n;
    }
    const int cus = cus_of[dev];
    if (M <= 0 || M % DT_M != 0) return 0;
    const int slabs = M / DT_M;
    const int waves = slabs % 4 == 0 ? 4 : (slabs % 2 == 0 ? 2 : 1);
    const int per_tile = slabs / waves;                      // workgroups per column tile (grid.y)
    const int slots = (8 / waves) * cus / per_tile;
    return slots >= 64 ? slots : 0;
}

// Columns per wave tile (= columns per statistics partial row) for a problem of P worst-case columns and
// M output rows: 64 while the expected number of waves (a quarter of the worst case is live after
// compaction) does not fill the 256 CUs a few times over, else 128.
extern "C" int o3d_direct_tile(long P, int M, int compact) {
    (void)M; (void)compact;
    // measured on the MI355X (batch 48, same run A/B): 64-column tiles for every launch lose 1.3 % (more loads
    // per MFMA); 64-column tiles only for the small launches (vote aggregation, BoxCloud xcorr: <= 64 K slots,
    // too few 128-column tiles for the 256 CUs) gain 0.8 % (7.705 -> 7.642 ms per step).  Round 4: with the split-K tile
    // behind the 64-column class, 300 K / 600 K slots (SA levels 2 / 1 on it) LOSE 0.15 / 0.57 ms per step
    // (profiles/r04_ab_splitk.txt)
    // Also measured (round 4, profiles/r04_ab_m2track_narrow_tiles.txt): M2-Track's 64- / 128-row layers over 98 304 columns
    // (768 / 1 536 waves of 128 columns) on the 64-column / split-K tiles instead: every data-gradient launch 20-120 % slower
    // (the 128 <- 1024 one 0.32 -> 0.71 ms), the step 6.92 -> 7.60 ms; only the 64 -> 64 forward gained (29 -> 24 us)
    return P <= 65536L ? 64 : 128;
}

// forward: Y = W . f(X), see o3d_mlp_conv_fwd
int o3d_direct_fwd(const float* X, const float* W, const float* in_scale, const float* in_shift, int B, int Cin,
                   int Cout, int P, float* Y, float* part, const float* stat_c, const float* w, const int32_t* meta,
                   long start1, int tile, hipStream_t st) {
    DirectArgs a = {};
    a.w = w; a.meta = meta; a.start1 = start1;
    a.A = W; a.X = X; a.c1 = in_scale; a.c2 = in_shift; a.Out = Y; a.M = Cout; a.K = Cin; a.P = P; a.B = B;
    a.part = part; a.stat_c = stat_c; a.ns = 4;
    if (meta && tile == 128 && B == 1 && in_scale) { a.tail_slots = o3d_direct_tail_slots(Cout); a.extra_row0 = P / 128; }
    return in_scale ? launch_direct<B_XFORM, 0>(a, tile, st) : launch_direct<B_PLAIN, 0>(a, tile, st);
}

// data gradient with the weights already transposed: Wt (Cin, Cout); see o3d_mlp_conv_dgrad
int o3d_direct_dgrad(const float* dN, const float* pk, int ns,
                     const float* Y, const float* A1, const float* A2, const float* A3, const float* Wt, int B,
                     int Cin, int Cout, int P, const float* Yprev, const float* scale_p, const float* shift_p,
                     const float* mean_p, float* dNprev, float* part, const float* w, const int32_t* meta,
                     long start1, int tile, hipStream_t st) {
    DirectArgs a = {};
    a.w = w; a.meta = meta; a.start1 = start1;
    a.A = Wt; a.X = dN; a.Y = Y; a.c1 = A1; a.c2 = A2; a.c3 = A3; a.pk = reinterpret_cast<const float2*>(pk); a.ns = ns;
    a.Out = dNprev; a.M = Cin; a.K = Cout; a.P = P; a.B = B; a.part = part;
    a.Yprev = Yprev; a.scale_p = scale_p; a.shift_p = shift_p; a.mean_p = mean_p;
    if (meta && tile == 128 && B == 1 && dN) { a.tail_slots = o3d_direct_tail_slots(Cin); a.extra_row0 = P / 128; }
    // Y == NULL: `dN` already IS dY (written once by o3d_mlp_conv_wgrad2_c_dy): one operand tensor, no constants
    if (dN && !Y) { a.c1 = a.c2 = a.c3 = nullptr; return launch_direct<B_PLAIN, 1>(a, tile, st); }
    return dN ? launch_direct<B_DY, 1>(a, tile, st) : launch_direct<B_DYPOOL, 1>(a, tile, st);
}

// ---- 1-D conv stacks (the trackers' heads) on the flat (C, P) layout, P = B*N columns -------------------------
// Same kernels, two more operand / epilogue combinations.  P % 64 == 0, Cin % 16 == 0, Cout % 64 == 0 (callers
// zero-pad); 64-column wave tiles while the problem is small (o3d_direct_tile).

// Y (Cout, P) = W (Cout, Cin) . f(X) [+ bias + resid];  f = relu(x*in_scale+in_shift) or identity (both NULL).
// part != NULL: BatchNorm statistics partials [P/tile][2][Cout] (then bias / resid must be NULL), tile =
// o3d_pw_tile(P, rows of the output) columns per partial row.
extern "C" int o3d_pw_tile(long P, int M) { return pw_tile(P, M); }

extern "C" int o3d_pw_fwd(const float* X, const float* W, const float* in_scale, const float* in_shift,
                          const float* bias, const float* resid, int Cin, int Cout, long P, float* Y, float* part,
                          const float* stat_c, void* stream) {
    if (!X || !W || !Y || P <= 0 || P > 0x7fffffff || P % 64 != 0 || Cin <= 0 || Cin % 16 != 0 || Cout <= 0 ||
        Cout % DT_M != 0 || (in_scale == nullptr) != (in_shift == nullptr) || (part && (bias || resid)))
        return O3D_EINVAL;
    const int tile = pw_tile(P, Cout);
    if (tile == 128 && P % 128 != 0) return O3D_EINVAL;
    DirectArgs a = {};
    a.A = W; a.X = X; a.c1 = in_scale; a.c2 = in_shift; a.Out = Y; a.M = Cout; a.K = Cin; a.P = (int)P; a.B = 1;
    a.part = part; a.stat_c = stat_c; a.ns = 4; a.bias = bias; a.resid = resid;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (part || (!bias && !resid))
        return in_scale ? launch_direct_small<B_XFORM, 0>(a, tile, st) : launch_direct_small<B_PLAIN, 0>(a, tile, st);
    return in_scale ? launch_direct_small<B_XFORM, 2>(a, tile, st) : launch_direct_small<B_PLAIN, 2>(a, tile, st);
}

// dNprev (Cin, P) = Wt (Cin, Cout) . dY,  dY = dN (Y == NULL) or A1*dN + A2*Y + A3 (BatchNorm backward folded);
// Yprev != NULL: masked by the producer's ReLU (relu(Yprev*scale_p+shift_p) > 0) with its BatchNorm-backward
// partials in `part` [P/tile][2][Cin]; Yprev == NULL: plain store (+ resid): the gradient of the stack's input.
extern "C" int o3d_pw_dgrad(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3,
                            const float* Wt, int Cin, int Cout, long P, const float* Yprev, const float* scale_p,
                            const float* shift_p, const float* mean_p, const float* resid, float* dNprev, float* part,
                            void* stream) {
    if (!dN || !Wt || !dNprev || P <= 0 || P > 0x7fffffff || P % 64 != 0 || Cin <= 0 || Cin % DT_M != 0 ||
        Cout <= 0 || Cout % 16 != 0 || (Y && (!A1 || !A2 || !A3)) ||
        (Yprev && (!scale_p || !shift_p || !mean_p || !part || resid)) || (!Yprev && part))
        return O3D_EINVAL;
    const int tile = pw_tile(P, Cin);
    if (tile == 128 && P % 128 != 0) return O3D_EINVAL;
    DirectArgs a = {};
    a.A = Wt; a.X = dN; a.Y = Y; a.c1 = A1; a.c2 = A2; a.c3 = A3; a.ns = 4;
    a.Out = dNprev; a.M = Cin; a.K = Cout; a.P = (int)P; a.B = 1; a.part = part;
    a.Yprev = Yprev; a.scale_p = scale_p; a.shift_p = shift_p; a.mean_p = mean_p; a.resid = resid;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (Yprev) return Y ? launch_direct_small<B_DY, 1>(a, tile, st) : launch_direct_small<B_PLAIN, 1>(a, tile, st);
    return Y ? launch_direct_small<B_DY, 2>(a, tile, st) : launch_direct_small<B_PLAIN, 2>(a, tile, st);
}

// ---- pairs ---------------------------------------------------------------------------------------------------------------
namespace {
template <int MODE, int EPI, int NT, int MT>
int launch_pair_nt(const DirectArgs& a, const DirectArgs& b, hipStream_t st) {
    const int tiles = a.P / (32 * NT);
    const int sa = a.M / (32 * MT), sb = b.M / (32 * MT);
    const int smax = sa > sb ? sa : sb;
    if (sa % 4 == 0 && sb % 4 == 0)
        hipLaunchKernelGGL((direct_gemm_pair_kernel<4, MODE, EPI, NT, MT>), dim3(tiles, smax / 4, 2), dim3(256), 0, st, a, b);
    else if (sa % 2 == 0 && sb % 2 == 0)
        hipLaunchKernelGGL((direct_gemm_pair_kernel<2, MODE, EPI, NT, MT>), dim3(tiles, smax / 2, 2), dim3(128), 0, st, a, b);
    else
        hipLaunchKernelGGL((direct_gemm_pair_kernel<1, MODE, EPI, NT, MT>), dim3(tiles, smax, 2), dim3(64), 0, st, a, b);
    return o3d_launch_status();
}

// the tile launch_direct_small picks: 2 = 64 x 64 wave tiles, 3 = 64 x 128, 4 = the split-K tile
static int small_class(const DirectArgs& a, int tile) {
    if (splitk_ok(a, tile)) return 4;
    return tile == 64 ? 2 : 3;
}

template <int MODE, int EPI>
int launch_pair(const DirectArgs& a, const DirectArgs& b, int cls, hipStream_t st) {
    if (cls == 4) {
        const int sa = a.M / SK_ROWS, sb = b.M / SK_ROWS;
        hipLaunchKernelGGL((splitk_gemm_pair_kernel<MODE, EPI>), dim3(a.P / SK_COLS, sa > sb ? sa : sb, 2), dim3(256), 0, st, a, b);
        return o3d_launch_status();
    }
    return cls == 2 ? launch_pair_nt<MODE, EPI, 2, 2>(a, b, st) : launch_pair_nt<MODE, EPI, 4, 2>(a, b, st);
}

static bool pw_fwd_args_ok(const o3d_pw_fwd_args& q) {
    return q.X && q.W && q.Y && q.P > 0 && q.P <= 0x7fffffff && q.P % 64 == 0 && q.Cin > 0 && q.Cin % 16 == 0 && q.Cout > 0 &&
           q.Cout % DT_M == 0 && (q.in_scale == nullptr) == (q.in_shift == nullptr) && !(q.part && (q.bias || q.resid));
}
static DirectArgs pw_fwd_direct(const o3d_pw_fwd_args& q) {
    DirectArgs a = {};
    a.A = q.W; a.X = q.X; a.c1 = q.in_scale; a.c2 = q.in_shift; a.Out = q.Y; a.M = q.Cout; a.K = q.Cin; a.P = (int)q.P; a.B = 1;
    a.part = q.part; a.stat_c = q.stat_c; a.ns = 4; a.bias = q.bias; a.resid = q.resid;
    return a;
}
static bool pw_dgrad_args_ok(const o3d_pw_dgrad_args& q) {
    return q.dN && q.Wt && q.dNprev && q.P > 0 && q.P <= 0x7fffffff && q.P % 64 == 0 && q.Cin > 0 && q.Cin % DT_M == 0 &&
           q.Cout > 0 && q.Cout % 16 == 0 && !(q.Y && (!q.A1 || !q.A2 || !q.A3)) &&
           !(q.Yprev && (!q.scale_p || !q.shift_p || !q.mean_p || !q.part || q.resid)) && !(!q.Yprev && q.part);
}
static DirectArgs pw_dgrad_direct(const o3d_pw_dgrad_args& q) {
    DirectArgs a = {};
    a.A = q.Wt; a.X = q.dN; a.Y = q.Y; a.c1 = q.A1; a.c2 = q.A2; a.c3 = q.A3; a.ns = 4;
    a.Out = q.dNprev; a.M = q.Cin; a.K = q.Cout; a.P = (int)q.P; a.B = 1; a.part = q.part;
    a.Yprev = q.Yprev; a.scale_p = q.scale_p; a.shift_p = q.shift_p; a.mean_p = q.mean_p; a.resid = q.resid;
    return a;
}
}  // namespace

// o3d_pw_fwd for two independent problems over the same number of columns: ONE launch when both take the same kernel
// instantiation (operand mode, epilogue, wave tile), else the two single launches -- the result is the same either way.
extern "C" int o3d_pw_fwd_pair(const o3d_pw_fwd_args* pa, const o3d_pw_fwd_args* pb, void* stream) {
    if (!pa || !pb || !pw_fwd_args_ok(*pa) || !pw_fwd_args_ok(*pb)) return O3D_EINVAL;
    const o3d_pw_fwd_args &x = *pa, &y = *pb;
    const int ta = pw_tile(x.P, x.Cout), tb = pw_tile(y.P, y.Cout);
    if ((ta == 128 && x.P % 128) || (tb == 128 && y.P % 128)) return O3D_EINVAL;
    DirectArgs a = pw_fwd_direct(x), b = pw_fwd_direct(y);
    const bool epi2a = !(x.part || (!x.bias && !x.resid)), epi2b = !(y.part || (!y.bias && !y.resid));
    const int ca = small_class(a, ta), cb = small_class(b, tb);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (x.P != y.P || ta != tb || ca != cb || epi2a != epi2b || (x.in_scale == nullptr) != (y.in_scale == nullptr)) {
        const int rc = o3d_pw_fwd(x.X, x.W, x.in_scale, x.in_shift, x.bias, x.resid, x.Cin, x.Cout, x.P, x.Y, x.part, x.stat_c, stream);
        if (rc != O3D_OK) return rc;
        return o3d_pw_fwd(y.X, y.W, y.in_scale, y.in_shift, y.bias, y.resid, y.Cin, y.Cout, y.P, y.Y, y.part, y.stat_c, stream);
    }
    if (!epi2a) return x.in_scale ? launch_pair<B_XFORM, 0>(a, b, ca, st) : launch_pair<B_PLAIN, 0>(a, b, ca, st);
    return x.in_scale ? launch_pair<B_XFORM, 2>(a, b, ca, st) : launch_pair<B_PLAIN, 2>(a, b, ca, st);
}

// o3d_pw_dgrad for two independent problems, likewise
extern "C" int o3d_pw_dgrad_pair(const o3d_pw_dgrad_args* pa, const o3d_pw_dgrad_args* pb, void* stream) {
    if (!pa || !pb || !pw_dgrad_args_ok(*pa) || !pw_dgrad_args_ok(*pb)) return O3D_EINVAL;
    const o3d_pw_dgrad_args &x = *pa, &y = *pb;
    const int ta = pw_tile(x.P, x.Cin), tb = pw_tile(y.P, y.Cin);
    if ((ta == 128 && x.P % 128) || (tb == 128 && y.P % 128)) return O3D_EINVAL;
    DirectArgs a = pw_dgrad_direct(x), b = pw_dgrad_direct(y);
    const int ca = small_class(a, ta), cb = small_class(b, tb);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (x.P != y.P || ta != tb || ca != cb || (x.Yprev == nullptr) != (y.Yprev == nullptr) || (x.Y == nullptr) != (y.Y == nullptr)) {
        const int rc = o3d_pw_dgrad(x.dN, x.Y, x.A1, x.A2, x.A3, x.Wt, x.Cin, x.Cout, x.P, x.Yprev, x.scale_p, x.shift_p, x.mean_p,
                                    x.resid, x.dNprev, x.part, stream);
        if (rc != O3D_OK) return rc;
        return o3d_pw_dgrad(y.dN, y.Y, y.A1, y.A2, y.A3, y.Wt, y.Cin, y.Cout, y.P, y.Yprev, y.scale_p, y.shift_p, y.mean_p, y.resid,
                            y.dNprev, y.part, stream);
    }
    if (x.Yprev) return x.Y ? launch_pair<B_DY, 1>(a, b, ca, st) : launch_pair<B_PLAIN, 1>(a, b, ca, st);
    return x.Y ? launch_pair<B_DY, 2>(a, b, ca, st) : launch_pair<B_PLAIN, 2>(a, b, ca, st);
}

// Layer 0 of a per-point stack whose input is [X (Cin rows) ; a per-cloud constant block]: Y (Cout, P) = W_a . X +
// cbias[:, cloud of the column], cbias (Cout, B) = W_b . pooled^T computed by the caller; statistics partials
// [P/tile][2][Cout] of the biased output.  P = B*N, N % 128 == 0, Cin % 16 == 0, Cout % 64 == 0.
extern "C" int o3d_pw_fwd_cloud(const float* X, const float* W, const float* cbias, int B, int N, int Cin, int Cout,
                                float* Y, float* part, const float* stat_c, void* stream) {
    const long P = (long)B * N;
    if (!X || !W || !cbias || !Y || B <= 0 || N <= 0 || N % 128 != 0 || P > 0x7fffffff || Cin <= 0 || Cin % 16 != 0 ||
        Cout <= 0 || Cout % DT_M != 0)
        return O3D_EINVAL;
    DirectArgs a = {};
    a.A = W; a.X = X; a.Out = Y; a.M = Cout; a.K = Cin; a.P = (int)P; a.B = 1; a.part = part; a.stat_c = stat_c; a.ns = 4;
    a.cbias = cbias; a.cb_N = N; a.cb_B = B;
    return launch_direct_nt<B_PLAIN, 3, 4>(a, reinterpret_cast<hipStream_t>(stream));
}
