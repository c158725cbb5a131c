// mlp.hip -- the grouped-MLP hot loop of the set-abstraction / xcorr stacks as fp32-MFMA
// GEMMs for gfx950, with everything around the contraction fused into its prologue/epilogue.
//
// Replaces, per SharedMLP layer (pointnet2/utils/pytorch_utils.py:12-37,68-121):
//   Conv2d(1x1, bias=False) -> BatchNorm2d (batch statistics in training) -> ReLU
// plus the QueryAndGroup gather that feeds layer 0 (pointnet2_utils.py:299-339), the
// max-pool over nsample (pointnet2_modules.py:70-73) and the whole backward of that chain.
//
// Why per-layer kernels: training-mode BatchNorm needs the per-channel mean/variance over
// ALL B*npoint*nsample positions of a layer before the next layer may consume it, so a layer
// boundary is a grid-wide dependency.  Each boundary is cut into
//     GEMM (raw conv output Y + per-tile partial statistics)  ->  tiny finalize kernel
// and BN+ReLU is applied by the CONSUMER while it stages its operand ("transform on load"),
// so a normalised activation is never written to HBM.  The grouped (B,3+C,npoint,nsample)
// tensor is never materialised either: layer 0 gathers straight from (B,C,N) while staging.
//
// Contraction: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157.3 TFLOP/s dense peak).
//   wave tile 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs), workgroup = 2..8 waves,
//   operands staged global -> registers (prefetch of chunk t+1 issued before the MFMAs of
//   chunk t) -> LDS double buffer -> one ds_read per fragment; one barrier per K chunk.
//   fp32 MFMA is slow relative to LDS/HBM (64 cycles per instruction), so a plain
//   double-buffered pipeline keeps the matrix pipe busy; occupancy 2 workgroups/CU lets one
//   workgroup's epilogue (VALU/stores) overlap the other's MFMA stream.
// K index permutation: inside each group of 8 k's, MFMA step s (0..3) consumes k = 8g+s on
// lanes 0-31 and k = 8g+4+s on lanes 32-63, so an operand stored [row][k] is fetched with one
// ds_read_b128 per 4 steps; an operand stored [k][row] uses ds_read_b32 with the same mapping.
#include "mlp_common.hpp"

#include <mutex>
#include <unordered_set>

// aligned fast path (mlp_direct.hip)
bool o3d_direct_ok(int M, int K, int P);
int o3d_direct_fwd(const float* X, const float* W, const float* in_scale, const float* in_shift, int B, int Cin,
                   int Cout, int P, float* Y, float* part, const float* stat_c, const float* w, const int32_t* meta,
                   long start1, int tile, hipStream_t st);
int o3d_direct_dgrad(const float* dN, const float* pk, int ns,
                     const float* Y, const float* A1, const float* A2, const float* A3, const float* Wt, int B,
                     int Cin, int Cout, int P, const float* Yprev, const float* scale_p, const float* shift_p,
                     const float* mean_p, float* dNprev, float* part, const float* w, const int32_t* meta,
                     long start1, int tile, hipStream_t st);

namespace {


constexpr int BN_POS = 128;  // positions per workgroup tile (forward / dgrad)
constexpr int BK = 16;       // channel chunk (forward / dgrad)
constexpr int LDMK = BK + 4; // [row][k] LDS stride (conflict-free ds_read_b128, see HISTORY.md section 5b)
constexpr int WBK = 32;      // position chunk of the weight-gradient kernel
constexpr int WLD = WBK + 4;

// One K chunk of CH k's for a 64x64 wave tile.  A_MK: A stored [m][k] (stride lda) else [k][m].
template <int CH, bool A_MK, bool B_MK>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ As, int lda,
                                          const float* __restrict__ Bs, int ldb, int wm0, int wn0,
                                          int l31, int h, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int g = 0; g < CH / 8; ++g) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (A_MK) {
                const float4 v = *reinterpret_cast<const float4*>(&As[(wm0 + 32 * t + l31) * lda + 8 * g + 4 * h]);
                a[t][0] = v.x; a[t][1] = v.y; a[t][2] = v.z; a[t][3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) a[t][s] = As[(8 * g + 4 * h + s) * lda + wm0 + 32 * t + l31];
            }
            if (B_MK) {
                const float4 v = *reinterpret_cast<const float4*>(&Bs[(wn0 + 32 * t + l31) * ldb + 8 * g + 4 * h]);
                b[t][0] = v.x; b[t][1] = v.y; b[t][2] = v.z; b[t][3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) b[t][s] = Bs[(8 * g + 4 * h + s) * ldb + wn0 + 32 * t + l31];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = mfma32(a[tm][s], b[tn][s], acc[tm][tn]);
    }
}

// ======================================================================================
// K1: forward layer   Y[b,co,p] = sum_ci W[co,ci] * f(X[b,ci,p])
// ======================================================================================
struct FwdArgs {
    const float* X;         // (B,Cin,P) pre-activation of the producer layer
    const float* W;         // (Cout,Cin)
    float* Y;               // (B,Cout,P) raw conv output
    const float* in_scale;  // (Cin) f(x) = max(x*scale+shift,0) when XFORM
    const float* in_shift;
    float* part;            // [B*P/128][2][Cout] per-tile {sum y, sum (y-c)^2} or NULL
    const float* stat_c;    // (Cout) shift c of the second moment (running_mean) or NULL (=0)
    int B, Cin, Cout, P;
};

template <int BM, bool XFORM>
__global__ __launch_bounds__(BM * 2) void conv_fwd_kernel(FwdArgs a) {
    constexpr int T = BM * 2;            // threads
    constexpr int NB4 = 512 / T;         // float4 of the X tile per thread (16x128 floats)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto As = [&](int buf) -> float* { return smem + buf * (BM * LDMK); };                     // [m][k]
    auto Bs = [&](int buf) -> float* { return smem + 2 * BM * LDMK + buf * (BK * BN_POS); };  // [k][p]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int tiles_per_b = a.P / BN_POS;
    const int tile = blockIdx.x;
    const int b = tile / tiles_per_b;
    const int p0 = (tile - b * tiles_per_b) * BN_POS;
    const int co0 = blockIdx.y * BM;
    const bool w_vec = (a.Cin & 3) == 0;

    // fixed per-thread staging coordinates
    const int bc4 = tid & 31;            // float4 column of the X tile (positions 4*bc4..+3)
    const int br0 = tid >> 5;            // first k row; further rows at +T/32

    // Operand staging is split in two so that nothing waits on HBM in front of the MFMAs:
    //   load_chunk  -- only issues the global loads of chunk t+1 (raw values + per-row constants)
    //   store_chunk -- after the MFMAs of chunk t: applies BN+ReLU / centre subtraction, writes LDS
    float4 ra[2], rb[NB4];
    float rc0[NB4], rc1[NB4];   // per-row constants: (scale, shift) or (centre coordinate, -)
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {    // W tile: BM rows x 16 k  ([m][k] image)
            const int f = tid + T * i, m = f >> 2, c4 = f & 3;
            const int co = co0 + m, k = k0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < a.Cout) {
                const float* src = a.W + (long)co * a.Cin + k;
                if (w_vec && k + 3 < a.Cin) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (k + 0 < a.Cin) v.x = src[0];
                    if (k + 1 < a.Cin) v.y = src[1];
                    if (k + 2 < a.Cin) v.z = src[2];
                    if (k + 3 < a.Cin) v.w = src[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {  // X tile: 16 k rows x 128 positions ([k][p] image)
            const int ci = k0 + br0 + (T / 32) * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            float c0 = 0.f, c1 = 0.f;
            if (ci < a.Cin) {
                v = *reinterpret_cast<const float4*>(&a.X[((long)b * a.Cin + ci) * a.P + p0 + 4 * bc4]);
                if (XFORM) { c0 = a.in_scale[ci]; c1 = a.in_shift[ci]; }
            }
            rb[i] = v; rc0[i] = c0; rc1[i] = c1;
        }
    };
    auto store_chunk = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + T * i, m = f >> 2, c4 = f & 3;
            *reinterpret_cast<float4*>(&As(buf)[m * LDMK + 4 * c4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int r = br0 + (T / 32) * i;
            const int ci = k0 + r;
            float4 v = rb[i];
            if (XFORM) {
                if (ci < a.Cin) {
                    const float sc = rc0[i], sh = rc1[i];
                    v.x = fmaxf(fmaf(v.x, sc, sh), 0.f); v.y = fmaxf(fmaf(v.y, sc, sh), 0.f);
                    v.z = fmaxf(fmaf(v.z, sc, sh), 0.f); v.w = fmaxf(fmaf(v.w, sc, sh), 0.f);
                }
            }
            *reinterpret_cast<float4*>(&Bs(buf)[r * BN_POS + 4 * bc4]) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (a.Cin + BK - 1) / BK;
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    for (int t = 0; t < nchunks; ++t) {
        if (t + 1 < nchunks) load_chunk((t + 1) * BK);
        mma_chunk<BK, true, false>(As(t & 1), LDMK, Bs(t & 1), BN_POS, wm0, wn0, l31, h, acc);
        if (t + 1 < nchunks) store_chunk((t + 1) & 1, (t + 1) * BK);
        __syncthreads();
    }

    // ---- epilogue: raw output
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm0 + 32 * tm + acc_row(r, h);
            if (co < a.Cout) {
                float* dst = a.Y + ((long)b * a.Cout + co) * a.P + p0 + wn0 + l31;
                dst[0] = acc[tm][0][r];
                dst[32] = acc[tm][1][r];
            }
        }
    // ---- epilogue: per-tile BatchNorm statistics
    if (a.part) {
        float s[32], q[32];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm0 + 32 * tm + acc_row(r, h);
                const float c = (a.stat_c && co < a.Cout) ? a.stat_c[co] : 0.f;
                const float y0 = acc[tm][0][r], y1 = acc[tm][1][r];
                s[tm * 16 + r] = y0 + y1;
                q[tm * 16 + r] = (y0 - c) * (y0 - c) + (y1 - c) * (y1 - c);
            }
        reduce_scatter32(s, l31);
        reduce_scatter32(q, l31);
        // lane l31 now owns value index l31 -> (tm = l31>>4, r = l31&15)
        float* red = smem;  // [2 (wn)][BM][2]; the staging buffers are free after the last barrier
        const int row = wm0 + 32 * (l31 >> 4) + acc_row(l31 & 15, h);
        red[((wave & 1) * BM + row) * 2 + 0] = s[0];
        red[((wave & 1) * BM + row) * 2 + 1] = q[0];
        __syncthreads();
        if (tid < BM && co0 + tid < a.Cout) {
            float* dst = a.part + (long)tile * 2 * a.Cout + co0 + tid;
            dst[0] = red[tid * 2 + 0] + red[(BM + tid) * 2 + 0];
            dst[a.Cout] = red[tid * 2 + 1] + red[(BM + tid) * 2 + 1];
        }
    }
}

// ======================================================================================
// BatchNorm statistics finalize: partials -> mean / invstd / scale / shift (+ running stats)
// ======================================================================================
// finalize kernels: FIN_CH channels x FIN_SL partial-list slices per 256-thread workgroup
// (round 3 also tried folding the FIN_SL slices with fp64 butterflies instead of the serial LDS walk and fetching the
// per-channel constants up front: +0.08 ms per step on the same box -- 16 ds_bpermute round trips per lane cost more than
// 128 pipelined LDS reads by four lanes.  Not kept.  Likewise requesting the per-channel constants first and the first
// partial rows beside the live count instead of behind it (the kernel looks like a chain of dependent global accesses):
// 6.04 vs 6.04 / 6.06 vs 5.97 ms per step, 8.6 us per launch in the trace -- no gain, removed.)
#ifndef O3D_FIN_FASTMATH
#define O3D_FIN_FASTMATH 1
#endif
constexpr int FIN_CH = 4, FIN_SL = 64;      // measured: 2x128 and 8x32 are both slower (0.24 / 0.36 vs 0.17 ms per step)

struct BnFinArgs {
    const float* part;  // [nparts][2][C]
    int nparts, C;
    const int32_t* meta; int tile;   // compact layout: only the first meta[0]/tile parts are live (or NULL)
    double count;       // positions reduced (B*P)
    const float* stat_c; // shift used by the producer for the second moment, or NULL
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;  // updated in place when momentum >= 0 (may be NULL)
    float momentum, eps;
    float* mean; float* invstd; float* scale; float* shift;  // outputs (C each)
    // second segment (nparts1 > 0): its partial rows follow segment 0's `nparts`, its outputs / stat_c sit at
    // +C, its live count at meta[4]; the running statistics see segment 0's update first
    int nparts1; double count1;
    // statistics rows of a direct GEMM launch whose remainder tiles were cut into column blocks (mlp_common.hpp::tail_plan):
    // the same plan, from the same live count, tells how many extra rows follow the regular ones
    int tail_slots; long extra_row0;
};

__device__ __forceinline__ void bn_finalize_body(const BnFinArgs& a, const int bx) {
    // FIN_CH channels x FIN_SL tile-slices per workgroup: the partial list (up to 13 824 tiles) is a
    // latency-bound strided read, so it is spread over many lanes with 8 loads in flight each
    __shared__ double sh[2][FIN_SL][FIN_CH + 1];
    const int cl = threadIdx.x % FIN_CH, sl = threadIdx.x / FIN_CH;
    const int c = bx * FIN_CH + cl;
    const int nseg = a.nparts1 > 0 ? 2 : 1;
    for (int seg = 0; seg < nseg; ++seg) {
        const float* part = a.part + (seg ? (long)a.nparts * 2 * a.C : 0);
        const double count = seg ? a.count1 : a.count;
        const int off = seg * a.C;
        double s = 0.0, q = 0.0;
        int nparts = seg ? a.nparts1 : a.nparts;
        int nextra = 0;
        if (a.meta) {
            const int cap = nparts;
            const int live = (a.meta[4 * seg] + a.tile - 1) / a.tile;    // (ceil: a 512-column row of o3d_pool_bwd_dense may be half live)
            nparts = live < nparts ? live : nparts;
            if (a.tail_slots > 0) { const TailPlan pl = tail_plan(nparts, a.tail_slots, cap); nextra = pl.R * (pl.f - 1); }
        }
        if (c < a.C) {
#pragma unroll 8
            for (int t = sl; t < nparts; t += FIN_SL) {
                s += (double)part[((long)t * 2 + 0) * a.C + c];
                q += (double)part[((long)t * 2 + 1) * a.C + c];
            }
            const float* extra = a.part + (a.extra_row0 + (long)seg * a.tail_slots) * 2 * a.C;
            for (int t = sl; t < nextra; t += FIN_SL) {
                s += (double)extra[((long)t * 2 + 0) * a.C + c];
                q += (double)extra[((long)t * 2 + 1) * a.C + c];
            }
        }
        if (seg) __syncthreads();
        sh[0][sl][cl] = s;
        sh[1][sl][cl] = q;
        __syncthreads();
        if (sl == 0 && c < a.C) {
            s = 0.0; q = 0.0;
            for (int i = 0; i < FIN_SL; ++i) { s += sh[0][i][cl]; q += sh[1][i][cl]; }
            const double cs = a.stat_c ? (double)a.stat_c[off + c] : 0.0;
#if O3D_FIN_FASTMATH
            // no fp64 division / square root (software sequences of ~40 instructions each, on one lane per channel: measured
            // -0.05 ms per BAT step over the 24 forward finalizes, same-box A/B): v_rsq_f32 + one fp64 Newton step
            // (relative error ~1e-14), the reciprocal of the count taken once
            const double ic = 1.0 / count;
            const double mean = s * ic;
            double var = q * ic - (mean - cs) * (mean - cs);
            if (var < 0.0) var = 0.0;
            const double v = var + (double)a.eps;
            double rd = (double)rsqrtf((float)v);
            rd = rd * (1.5 - 0.5 * v * rd * rd);
            const float invstd = (float)rd;
#else
            const double mean = s / count;
            double var = q / count - (mean - cs) * (mean - cs);
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
#endif
            const float g = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
            a.mean[off + c] = (float)mean;
            a.invstd[off + c] = invstd;
            const float sc = g * invstd;
            a.scale[off + c] = sc;
            a.shift[off + c] = bt - (float)mean * sc;
            if (a.running_mean && a.momentum >= 0.f) {
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
                a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
            }
        }
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(BnFinArgs a) { bn_finalize_body(a, blockIdx.x); }

// two independent finalizes (two BatchNorm layers of two conv stacks advancing side by side) in one launch
__global__ __launch_bounds__(256) void bn_finalize_pair_kernel(BnFinArgs a0, BnFinArgs a1) {
    if (blockIdx.y == 0) { if ((int)blockIdx.x * FIN_CH < a0.C) bn_finalize_body(a0, blockIdx.x); }
    else if ((int)blockIdx.x * FIN_CH < a1.C) bn_finalize_body(a1, blockIdx.x);
}

// Stage 1 of a long partial list: out[g][i] = sum over parts t = g, g+G, g+2G, ... of part[t][i],
// i over the 2*C (statistic, channel) columns; coalesced rows, G x ceil(2C/256) workgroups.  The
// finalize kernels then walk G parts instead of thousands from 1-4 workgroups.
__global__ __launch_bounds__(256) void partials_fold_kernel(const float* __restrict__ part, int nparts, int n2c,
                                                            int G, float* __restrict__ out) {
    const int i = blockIdx.y * 256 + threadIdx.x, g = blockIdx.x;
    if (i >= n2c) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = g;
    for (; t + 3 * G < nparts; t += 4 * G) {
        s0 += part[(long)t * n2c + i];
        s1 += part[(long)(t + G) * n2c + i];
        s2 += part[(long)(t + 2 * G) * n2c + i];
        s3 += part[(long)(t + 3 * G) * n2c + i];
    }
    for (; t < nparts; t += G) s0 += part[(long)t * n2c + i];
    out[(long)g * n2c + i] = (s0 + s1) + (s2 + s3);
}

constexpr int FOLD_G = 32;

// folds a long partial list into FOLD_G parts in caller scratch (FOLD_G*2*C floats); no-op when short
static const float* fold_partials(const float* part, int& nparts, int C, float* fold, hipStream_t s) {
    // Round 5: only lists beyond O3D_FOLD_ABOVE rows are folded first.  The finalize kernels walk a list with 64 lanes per
    // channel group and handle the compact layout's 6 000-row lists in ~8 us without a fold; the fold launch in front of every
    // finalize of a 768-row (M2-Track, 98 304 columns) or 3 072-row (P2B's xcorr) list was a 4.6 us launch each, 31 per M2-Track
    // step (threshold was 128 rows; same-box A/B in profiles/r05_ab_partials_fold.txt)
#ifndef O3D_FOLD_ABOVE
#define O3D_FOLD_ABOVE 16384
#endif
    if (!fold || nparts <= O3D_FOLD_ABOVE) return part;
    hipLaunchKernelGGL(partials_fold_kernel, dim3(FOLD_G, o3d_cdiv(2 * C, 256)), dim3(256), 0, s, part, nparts, 2 * C,
                       FOLD_G, fold);
    nparts = FOLD_G;
    return fold;
}

// ======================================================================================
// BN + ReLU + max over the ns neighbours (forward of the pooling that ends a grouped MLP)
// ======================================================================================
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ Y,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int C,
                                                       int npoint, int ns, long total,
                                                       float* __restrict__ out, int32_t* __restrict__ arg,
                                                       float* __restrict__ yarg) {
    // 8 lanes share one (b, c, centre): lane e reads float4 e, e+8, ... of the ns neighbours, so a
    // wave reads 8 x 128 contiguous bytes per instruction; first-maximum arg-max via 3 shuffles.
    const long gi = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;  // over (b, c, j)
    const int e = threadIdx.x & 7;
    const bool live = gi < total;
    const long i = live ? gi : total - 1;
    const int c = (int)((i / npoint) % C);
    const float sc = scale[c], sf = shift[c];
    const float4* src = reinterpret_cast<const float4*>(Y + i * ns);
    float best = -INFINITY, ybest = 0.f;
    int bk = 0x7fffffff;
    for (int k4 = e; k4 < ns / 4; k4 += 8) {
        const float4 v = src[k4];
        const float y[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float n = fmaf(y[t], sc, sf);
            if (n > best) { best = n; bk = 4 * k4 + t; ybest = y[t]; }
        }
    }
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64), oy = __shfl_xor(ybest, off, 64);
        const int ok = __shfl_xor(bk, off, 64);
        if (ob > best || (ob == best && ok < bk)) { best = ob; bk = ok; ybest = oy; }
    }
    if (live && e == 0) {
        out[i] = fmaxf(best, 0.f);
        if (arg) { arg[i] = bk; yarg[i] = ybest; }
    }
}

// Backward statistics of the pooled layer: partial {sum g, sum g*(yarg-mean)} with g = dOut where the pooled activation is
// positive (+ the packed {g, bits(arg-max slot)} pairs the pooled-source gradient kernels read), for ONE cloud of many
// balls (the flat (C, npoint) layout of the P2B fusion: npoint = B*N balls): a channel's
// balls are split over gridDim.y workgroups, part [gridDim.y][2][C].
__global__ __launch_bounds__(256) void pool_bwd_partials_split_kernel(const float* __restrict__ dOut,
                                                                      const float* __restrict__ out,
                                                                      const float* __restrict__ yarg,
                                                                      const float* __restrict__ mean, int C, int npoint,
                                                                      float* __restrict__ part,
                                                                      const int32_t* __restrict__ arg,
                                                                      float2* __restrict__ pk) {
    __shared__ float sh[2][4];
    const int c = blockIdx.x, k = blockIdx.y;
    const int share = (npoint + gridDim.y - 1) / gridDim.y;
    const int j0 = k * share, j1 = min(j0 + share, npoint);
    const float mu = mean[c];
    float s = 0.f, q = 0.f;
    for (int j = j0 + threadIdx.x; j < j1; j += 256) {
        const long i = (long)c * npoint + j;
        const float g = out[i] > 0.f ? dOut[i] : 0.f;
        if (pk) pk[i] = make_float2(g, __int_as_float(arg[i]));
        s += g;
        q += g * (yarg[i] - mu);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((long)k * 2 + 0) * C + c] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[((long)k * 2 + 1) * C + c] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

// BatchNorm backward finalize: partials {sum dN, sum dN*(Y-mean)} -> dgamma, dbeta and the
// coefficients of dY = A1*dN + A2*Y + A3 (per channel).
struct BnBwdFinArgs {
    const float* part; int nparts, C; double count;
    const int32_t* meta; int tile;
    const float* gamma; const float* mean; const float* invstd;
    float* dgamma; float* dbeta; float* A1; float* A2; float* A3;
    // second segment (nparts1 > 0): partial rows after segment 0's, mean / invstd / A1..A3 at +C, live count at
    // meta[4]; dgamma / dbeta (C each) are the SUM over the segments (the affine parameters are shared)
    int nparts1; double count1;
    int tail_slots; long extra_row0;      // as BnFinArgs
};

__device__ __forceinline__ void bn_bwd_finalize_body(const BnBwdFinArgs& a, const int bx) {
    __shared__ double sh[2][FIN_SL][FIN_CH + 1];
    const int cl = threadIdx.x % FIN_CH, sl = threadIdx.x / FIN_CH;
    const int c = bx * FIN_CH + cl;
    const int nseg = a.nparts1 > 0 ? 2 : 1;
    double dg = 0.0, db = 0.0;
    for (int seg = 0; seg < nseg; ++seg) {
        const float* part = a.part + (seg ? (long)a.nparts * 2 * a.C : 0);
        const double count = seg ? a.count1 : a.count;
        const int off = seg * a.C;
        double s = 0.0, q = 0.0;
        int nparts = seg ? a.nparts1 : a.nparts;
        int nextra = 0;
        if (a.meta) {
            const int cap = nparts;
            const int live = (a.meta[4 * seg] + a.tile - 1) / a.tile;    // (ceil: a 512-column row of o3d_pool_bwd_dense may be half live)
            nparts = live < nparts ? live : nparts;
            if (a.tail_slots > 0) { const TailPlan pl = tail_plan(nparts, a.tail_slots, cap); nextra = pl.R * (pl.f - 1); }
        }
        if (c < a.C) {
#pragma unroll 8
            for (int t = sl; t < nparts; t += FIN_SL) {
                s += (double)part[((long)t * 2 + 0) * a.C + c];
                q += (double)part[((long)t * 2 + 1) * a.C + c];
            }
            const float* extra = a.part + (a.extra_row0 + (long)seg * a.tail_slots) * 2 * a.C;
            for (int t = sl; t < nextra; t += FIN_SL) {
                s += (double)extra[((long)t * 2 + 0) * a.C + c];
                q += (double)extra[((long)t * 2 + 1) * a.C + c];
            }
        }
        if (seg) __syncthreads();
        sh[0][sl][cl] = s;
        sh[1][sl][cl] = q;
        __syncthreads();
        if (sl == 0 && c < a.C) {
            s = 0.0; q = 0.0;
            for (int i = 0; i < FIN_SL; ++i) { s += sh[0][i][cl]; q += sh[1][i][cl]; }
            const double g = a.gamma ? (double)a.gamma[c] : 1.0;
            const double is = (double)a.invstd[off + c], mu = (double)a.mean[off + c];
            db += s;
            dg += q * is;
            const double a1 = g * is;
#if O3D_FIN_FASTMATH
            const double ic = 1.0 / count;
            const double a2 = -a1 * is * is * q * ic;
            const double a3 = -a1 * s * ic - a2 * mu;
#else
            const double a2 = -a1 * is * is * q / count;
            const double a3 = -a1 * s / count - a2 * mu;
#endif
            a.A1[off + c] = (float)a1; a.A2[off + c] = (float)a2; a.A3[off + c] = (float)a3;
        }
    }
    if (sl == 0 && c < a.C) {
        a.dbeta[c] = (float)db;
        a.dgamma[c] = (float)dg;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(BnBwdFinArgs a) { bn_bwd_finalize_body(a, blockIdx.x); }

__global__ __launch_bounds__(256) void bn_bwd_finalize_pair_kernel(BnBwdFinArgs a0, BnBwdFinArgs a1) {
    if (blockIdx.y == 0) { if ((int)blockIdx.x * FIN_CH < a0.C) bn_bwd_finalize_body(a0, blockIdx.x); }
    else if ((int)blockIdx.x * FIN_CH < a1.C) bn_bwd_finalize_body(a1, blockIdx.x);
}

// ======================================================================================
// dY operand loader shared by the dgrad and wgrad kernels
//   dense : dN (B,Cout,P) stored by the consumer layer's dgrad epilogue
//   pooled: dN[b,co,j*ns+k] = (k == arg[b,co,j] && out[b,co,j] > 0) ? dOut[b,co,j] : 0
//   dY = A1[co]*dN + A2[co]*Y + A3[co]
// ======================================================================================
struct DyArgs {
    const float* dN;      // dense source or NULL
    const float* dOut;    // pooled source (B,Cout,npoint)
    const float* out;     // pooled activation (B,Cout,npoint)
    const int32_t* arg;   // (B,Cout,npoint)
    const float* Y;       // (B,Cout,P)
    const float* A1; const float* A2; const float* A3;
    int ns;
};

struct RawDy {
    float4 g;      // dense: dN;  pooled: {dOut, bits(arg), out, -}
    float4 y;      // raw conv output Y
    float a1, a2, a3;
    int k;         // pooled: neighbour index of the first of the four positions
};

template <bool POOLED>
__device__ __forceinline__ RawDy load_dy_raw(const DyArgs& d, long b, int co, int Cout, int P, int p) {
    RawDy r;
    const long row = b * Cout + co;
    r.y = *reinterpret_cast<const float4*>(&d.Y[row * P + p]);
    if (POOLED) {
        const int np = P / d.ns;
        const int j = p / d.ns;
        const long pi = row * np + j;
        r.k = p - j * d.ns;
        r.g = make_float4(d.dOut[pi], __int_as_float(d.arg[pi]), d.out[pi], 0.f);
    } else {
        r.k = 0;
        r.g = *reinterpret_cast<const float4*>(&d.dN[row * P + p]);
    }
    r.a1 = d.A1[co]; r.a2 = d.A2[co]; r.a3 = d.A3[co];
    return r;
}

template <bool POOLED>
__device__ __forceinline__ float4 finish_dy(const RawDy& r) {
    float4 g;
    if (POOLED) {
        const int ak = __float_as_int(r.g.y);
        const float go = r.g.z > 0.f ? r.g.x : 0.f;
        g.x = (r.k + 0 == ak) ? go : 0.f; g.y = (r.k + 1 == ak) ? go : 0.f;
        g.z = (r.k + 2 == ak) ? go : 0.f; g.w = (r.k + 3 == ak) ? go : 0.f;
    } else {
        g = r.g;
    }
    float4 o;
    o.x = fmaf(r.a1, g.x, fmaf(r.a2, r.y.x, r.a3)); o.y = fmaf(r.a1, g.y, fmaf(r.a2, r.y.y, r.a3));
    o.z = fmaf(r.a1, g.z, fmaf(r.a2, r.y.z, r.a3)); o.w = fmaf(r.a1, g.w, fmaf(r.a2, r.y.w, r.a3));
    return o;
}

// ======================================================================================
// K2: data gradient   G[b,ci,p] = sum_co W[co,c_lo+ci] * dY[b,co,p],  ci in [0,M)
//   EPI 0 (MASK):    dNprev = G * [Yprev*scale_p+shift_p > 0] -> store (B,M,P)
//                    + per-tile partials {sum dNprev, sum dNprev*(Yprev-mean_p)}
//   EPI 1 (STORE):   layer 0 of a grouped MLP: store G (B,M,P); scatter_rows_kernel then adds it
//                    through idx into (B,M,N) with LDS-privatised accumulation
// ======================================================================================
struct DgradArgs {
    DyArgs dy;
    const float* W;        // (Cout,Cin)
    int B, Cin, Cout, P, c_lo, M;
    // EPI 0
    const float* Yprev; const float* scale_p; const float* shift_p; const float* mean_p;
    float* dNprev; float* part;
    // EPI 1: plain store of G into dNprev (B,M,P)
};

template <int BM, bool POOLED, int EPI>
__global__ __launch_bounds__(BM * 2) void conv_dgrad_kernel(DgradArgs a) {
    constexpr int T = BM * 2;
    constexpr int NB4 = 512 / T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto As = [&](int buf) -> float* { return smem + buf * (BK * BM); };                     // [k][m]
    auto Bs = [&](int buf) -> float* { return smem + 2 * BK * BM + buf * (BK * BN_POS); };  // [k][p]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int tiles_per_b = a.P / BN_POS;
    const int tile = blockIdx.x;
    const int b = tile / tiles_per_b;
    const int p0 = (tile - b * tiles_per_b) * BN_POS;
    const int m0 = blockIdx.y * BM;
    const bool w_vec = ((a.Cin & 3) == 0) && ((a.c_lo & 3) == 0);
    const int bc4 = tid & 31, br0 = tid >> 5;

    float4 ra[2];
    RawDy rb[NB4];
    bool rbv[NB4];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {     // W^T tile: 16 k(co) rows x BM m(ci)
            const int f = tid + T * i, r = f / (BM / 4), m4 = f - r * (BM / 4);
            const int co = k0 + r, m = m0 + 4 * m4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < a.Cout) {
                const float* src = a.W + (long)co * a.Cin + a.c_lo + m;
                if (w_vec && m + 3 < a.M) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (m + 0 < a.M) v.x = src[0];
                    if (m + 1 < a.M) v.y = src[1];
                    if (m + 2 < a.M) v.z = src[2];
                    if (m + 3 < a.M) v.w = src[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {   // dY tile: 16 k(co) rows x 128 positions
            const int co = k0 + br0 + (T / 32) * i;
            rbv[i] = co < a.Cout;
            if (rbv[i]) rb[i] = load_dy_raw<POOLED>(a.dy, b, co, a.Cout, a.P, p0 + 4 * bc4);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + T * i, r = f / (BM / 4), m4 = f - r * (BM / 4);
            *reinterpret_cast<float4*>(&As(buf)[r * BM + 4 * m4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int r = br0 + (T / 32) * i;
            *reinterpret_cast<float4*>(&Bs(buf)[r * BN_POS + 4 * bc4]) =
                rbv[i] ? finish_dy<POOLED>(rb[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (a.Cout + BK - 1) / BK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int t = 0; t < nchunks; ++t) {
        if (t + 1 < nchunks) load_chunk((t + 1) * BK);
        if (EPI != 1) mma_chunk<BK, false, false>(As(t & 1), BM, Bs(t & 1), BN_POS, wm0, wn0, l31, h, acc);
        else          mma_chunk<BK, false, false>(Bs(t & 1), BN_POS, As(t & 1), BM, wn0, wm0, l31, h, acc);
        if (t + 1 < nchunks) store_chunk((t + 1) & 1);
        __syncthreads();
    }

    if (EPI == 0) {
        float s[32], q[32];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * tm + acc_row(r, h);
                float s_ = 0.f, q_ = 0.f;
                if (m < a.M) {
                    const float sc = a.scale_p[m], sf = a.shift_p[m], mu = a.mean_p[m];
                    const long o = ((long)b * a.M + m) * a.P + p0 + wn0 + l31;
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const float yp = a.Yprev[o + 32 * tn];
                        const float g = fmaf(yp, sc, sf) > 0.f ? acc[tm][tn][r] : 0.f;
                        a.dNprev[o + 32 * tn] = g;
                        s_ += g;
                        q_ += g * (yp - mu);
                    }
                }
                s[tm * 16 + r] = s_;
                q[tm * 16 + r] = q_;
            }
        reduce_scatter32(s, l31);
        reduce_scatter32(q, l31);
        float* red = smem;
        const int row = wm0 + 32 * (l31 >> 4) + acc_row(l31 & 15, h);
        red[((wave & 1) * BM + row) * 2 + 0] = s[0];
        red[((wave & 1) * BM + row) * 2 + 1] = q[0];
        __syncthreads();
        if (tid < BM && m0 + tid < a.M) {
            float* dst = a.part + (long)tile * 2 * a.M + m0 + tid;
            dst[0] = red[tid * 2 + 0] + red[(BM + tid) * 2 + 0];
            dst[a.M] = red[tid * 2 + 1] + red[(BM + tid) * 2 + 1];
        }
    } else if (EPI == 2) {
        // plain store of G (B,M,P): the gradient w.r.t. an operand that is not a BN+ReLU output
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * tm + acc_row(r, h);
                if (m < a.M) {
                    float* dst = a.dNprev + ((long)b * a.M + m) * a.P + p0 + wn0 + l31;
                    dst[0] = acc[tm][0][r];
                    dst[32] = acc[tm][1][r];
                }
            }
    } else {
        // operand roles were swapped for this epilogue: acc[tm][tn] holds positions (rows) x input
        // channels (lane = channel), so G^T (B,P,M) is stored with 128-byte channel rows.  The
        // scatter through the grouping indices is a separate CSR gather (scatter_gt_kernel):
        // global fp32 atomics from here ran at the L2 atomic rate (170 M atomics = 9 ms per BAT step).
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + wn0 + 32 * tm + acc_row(r, h);
                float* dst = a.dNprev + ((long)b * a.P + p) * a.M + m0 + wm0 + l31;
                if (m0 + wm0 + l31 < a.M) dst[0] = acc[tm][0][r];
                if (m0 + wm0 + 32 + l31 < a.M) dst[32] = acc[tm][1][r];
            }
    }
}
// ======================================================================================
// K3: weight gradient   dW[co,ci] = sum_{b,p} dY[b,co,p] * X[b,ci,p]   (split over positions)
//   X: transform-on-load of the producer's raw output, or the layer-0 gather.
//   Each workgroup reduces `chunks_per_block` chunks of 32 positions into a 128x128 tile of
//   partial dW and writes it to part[z][Cout][Cin]; wgrad_reduce_kernel sums the slices.
// ======================================================================================
struct WgradArgs {
    DyArgs dy;
    const float* X; const float* in_scale; const float* in_shift;
    int B, Cin, Cout, P;
    int chunks_per_block, total_chunks;
    int tiles_ci, tiles_co, nslices;
    float* part;   // [nslices][Cout][Cin]
};

template <bool POOLED, bool XFORM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto As = [&](int buf) -> float* { return smem + buf * (128 * WLD); };        // dY^T  [co][pos]
    auto Bs = [&](int buf) -> float* { return smem + (2 + buf) * (128 * WLD); };  // X^T   [ci][pos]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    // 1-D grid.  Workgroup w is dispatched to XCD w % 8: the tiles_ci x tiles_co output tiles of one
    // position slice read the same dN / Y / X rows, so they are made neighbours on ONE XCD (shared L2)
    // instead of round-robin neighbours on different XCDs (every tile re-reading HBM).
    const int ntile = a.tiles_ci * a.tiles_co;
    int tile_id, slice;
    if ((a.nslices & 7) == 0) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        tile_id = local % ntile;
        slice = (local / ntile) * 8 + xcd;
    } else {
        tile_id = blockIdx.x % ntile;
        slice = blockIdx.x / ntile;
    }
    const int ci0 = (tile_id % a.tiles_ci) * 128, co0 = (tile_id / a.tiles_ci) * 128;
    const int c_begin = slice * a.chunks_per_block;
    int c_end = c_begin + a.chunks_per_block;
    if (c_end > a.total_chunks) c_end = a.total_chunks;
    const int chunks_per_b = a.P / WBK;
    const int r0 = tid >> 3, c4 = tid & 7;   // row (+32*i) and float4 column inside a chunk

    // staging split like the forward kernel: raw global loads before the MFMAs, transforms after
    RawDy ra[4];
    float4 rb[4];
    float rc0[4], rc1[4];
    auto load_chunk = [&](int ch) {
        const long b = ch / chunks_per_b;
        const int p = (ch - (int)b * chunks_per_b) * WBK + 4 * c4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + r0 + 32 * i;
            if (co < a.Cout) ra[i] = load_dy_raw<POOLED>(a.dy, b, co, a.Cout, a.P, p);
            const int ci = ci0 + r0 + 32 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            float c0 = 0.f, c1 = 0.f;
            if (ci < a.Cin) {
                v = *reinterpret_cast<const float4*>(&a.X[(b * a.Cin + ci) * a.P + p]);
                if (XFORM) { c0 = a.in_scale[ci]; c1 = a.in_shift[ci]; }
            }
            rb[i] = v; rc0[i] = c0; rc1[i] = c1;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + r0 + 32 * i;
            *reinterpret_cast<float4*>(&As(buf)[(r0 + 32 * i) * WLD + 4 * c4]) =
                co < a.Cout ? finish_dy<POOLED>(ra[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int ci = ci0 + r0 + 32 * i;
            float4 v = rb[i];
            if (XFORM) {
                if (ci < a.Cin) {
                    const float sc = rc0[i], sh = rc1[i];
                    v.x = fmaxf(fmaf(v.x, sc, sh), 0.f); v.y = fmaxf(fmaf(v.y, sc, sh), 0.f);
                    v.z = fmaxf(fmaf(v.z, sc, sh), 0.f); v.w = fmaxf(fmaf(v.w, sc, sh), 0.f);
                }
            }
            *reinterpret_cast<float4*>(&Bs(buf)[(r0 + 32 * i) * WLD + 4 * c4]) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        __syncthreads();
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int t = ch - c_begin;
            if (ch + 1 < c_end) load_chunk(ch + 1);
            mma_chunk<WBK, true, true>(As(t & 1), WLD, Bs(t & 1), WLD, wm0, wn0, l31, h, acc);
            if (ch + 1 < c_end) store_chunk((t + 1) & 1);
            __syncthreads();
        }
    }
    float* dst = a.part + (long)slice * a.Cout * a.Cin;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm0 + 32 * tm + acc_row(r, h);
            if (co >= a.Cout) continue;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int ci = ci0 + wn0 + 32 * tn + l31;
                if (ci < a.Cin) dst[(long)co * a.Cin + ci] = acc[tm][tn][r];
            }
        }
}

// out[g][i] = sum over slices z in group g of part[z][i]; groups of `per` consecutive slices
// (grid.y = number of groups).  Called twice (nslices -> 16 groups -> 1) so that no thread walks
// hundreds of slices serially; the summation order is fixed.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nslices,
                                                           int per, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int z0 = blockIdx.y * per;
    int z1 = z0 + per;
    if (z1 > nslices) z1 = nslices;
    float s = 0.f;
#pragma unroll 8
    for (int z = z0; z < z1; ++z) s += part[(long)z * n + i];
    out[(long)blockIdx.y * n + i] = s;
}

// one launch: 32 outputs x 8 slice groups per workgroup, groups summed through LDS in a fixed order
__global__ __launch_bounds__(256) void wgrad_reduce1_kernel(const float* __restrict__ part, int nslices, long n,
                                                            float* __restrict__ out) {
    __shared__ float sh[8][32];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + col;
    const int per = (nslices + 7) / 8, z0 = grp * per;
    int z1 = z0 + per;
    if (z1 > nslices) z1 = nslices;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int z = z0;
        for (; z + 3 < z1; z += 4) {
            s0 += part[(long)z * n + i]; s1 += part[(long)(z + 1) * n + i];
            s2 += part[(long)(z + 2) * n + i]; s3 += part[(long)(z + 3) * n + i];
        }
        for (; z < z1; ++z) s0 += part[(long)z * n + i];
    }
    sh[grp][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += sh[g][col];
        out[i] = t;
    }
}

template <typename K, typename... A>
int launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    if (lds > 48 * 1024) {  // dynamic LDS beyond the default window must be opted into, once per kernel
        static std::mutex mu;
        static std::unordered_set<const void*> done;
        std::lock_guard<std::mutex> lock(mu);
        const void* key = reinterpret_cast<const void*>(kernel);
        if (!done.count(key)) {
            if (hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return O3D_ELAUNCH;
            done.insert(key);
        }
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
    return o3d_launch_status();
}

inline size_t fwd_lds(int BM) { return sizeof(float) * (2 * BM * LDMK + 2 * BK * BN_POS); }
inline size_t dgrad_lds(int BM) {
    size_t stage = sizeof(float) * (2 * BK * BM + 2 * BK * BN_POS), red = sizeof(float) * 4 * BM;
    return stage > red ? stage : red;
}

template <bool XFORM>
int launch_fwd(const FwdArgs& a, hipStream_t s) {
    const int tiles = a.B * (a.P / BN_POS);
    if (a.Cout > 128) {
        return launch(conv_fwd_kernel<256, XFORM>, dim3(tiles, o3d_cdiv(a.Cout, 256)), dim3(512),
                      fwd_lds(256), s, a);
    } else if (a.Cout > 64) {
        return launch(conv_fwd_kernel<128, XFORM>, dim3(tiles, 1), dim3(256), fwd_lds(128), s, a);
    }
    return launch(conv_fwd_kernel<64, XFORM>, dim3(tiles, 1), dim3(128), fwd_lds(64), s, a);
}

template <bool POOLED, int EPI>
int launch_dgrad(const DgradArgs& a, hipStream_t s) {
    const int tiles = a.B * (a.P / BN_POS);
    if (a.M > 128) {
        return launch(conv_dgrad_kernel<256, POOLED, EPI>, dim3(tiles, o3d_cdiv(a.M, 256)), dim3(512),
                      dgrad_lds(256), s, a);
    } else if (a.M > 64) {
        return launch(conv_dgrad_kernel<128, POOLED, EPI>, dim3(tiles, 1), dim3(256), dgrad_lds(128), s, a);
    }
    return launch(conv_dgrad_kernel<64, POOLED, EPI>, dim3(tiles, 1), dim3(128), dgrad_lds(64), s, a);
}

}  // namespace

// --------------------------------------------------------------------------------------
// C-ABI
// --------------------------------------------------------------------------------------
extern "C" int o3d_mlp_conv_fwd(const float* X, const float* W, const float* in_scale,
                                const float* in_shift, int B, int Cin, int Cout, int P, float* Y,
                                float* part, const float* stat_c, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || P <= 0 || P % BN_POS != 0 || !X || !W || !Y) return O3D_EINVAL;
    if (o3d_direct_ok(Cout, Cin, P) && (in_scale == nullptr) == (in_shift == nullptr))
        // (the statistics partials are one row per 128 columns by contract: narrower tiles only without them)
        return o3d_direct_fwd(X, W, in_scale, in_shift, B, Cin, Cout, P, Y, part, stat_c, nullptr, nullptr, 0,
                              part ? 128 : o3d_direct_tile((long)B * P, Cout, 0), o3d_stream(stream));
    FwdArgs a = {};
    a.X = X; a.W = W; a.Y = Y; a.in_scale = in_scale; a.in_shift = in_shift; a.part = part; a.stat_c = stat_c;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.P = P;
    return in_scale ? launch_fwd<true>(a, o3d_stream(stream)) : launch_fwd<false>(a, o3d_stream(stream));
}

extern "C" int o3d_bn_finalize(const float* part, int nparts, int C, double count, const float* stat_c,
                               const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, float* mean,
                               float* invstd, float* scale, float* shift, float* fold, void* stream) {
    if (!part || nparts <= 0 || C <= 0 || !mean || !invstd || !scale || !shift) return O3D_EINVAL;
    part = fold_partials(part, nparts, C, fold, o3d_stream(stream));
    BnFinArgs a = {part, nparts, C, nullptr, 1, count, stat_c, gamma, beta, running_mean, running_var, momentum, eps,
                   mean, invstd, scale, shift};
    return launch(bn_finalize_kernel, dim3(o3d_cdiv(C, FIN_CH)), dim3(256), 0, o3d_stream(stream), a);
}

// o3d_bn_finalize for two independent layers in one launch (short partial lists: no fold stage)
extern "C" int o3d_bn_finalize_pair(const o3d_bn_fin_args* pa, const o3d_bn_fin_args* pb, void* stream) {
    if (!pa || !pb) return O3D_EINVAL;
    BnFinArgs v[2];
    const o3d_bn_fin_args* q[2] = {pa, pb};
    for (int i = 0; i < 2; ++i) {
        const o3d_bn_fin_args& x = *q[i];
        if (!x.part || x.nparts <= 0 || x.C <= 0 || !x.mean || !x.invstd || !x.scale || !x.shift) return O3D_EINVAL;
        v[i] = BnFinArgs{x.part, x.nparts, x.C, nullptr, 1, x.count, x.stat_c, x.gamma, x.beta, x.running_mean, x.running_var,
                         x.momentum, x.eps, x.mean, x.invstd, x.scale, x.shift};
    }
    const int cmax = pa->C > pb->C ? pa->C : pb->C;
    hipLaunchKernelGGL(bn_finalize_pair_kernel, dim3(o3d_cdiv(cmax, FIN_CH), 2), dim3(256), 0, o3d_stream(stream), v[0], v[1]);
    return o3d_launch_status();
}

extern "C" int o3d_bn_bwd_finalize_pair(const o3d_bn_bwd_fin_args* pa, const o3d_bn_bwd_fin_args* pb, void* stream) {
    if (!pa || !pb) return O3D_EINVAL;
    BnBwdFinArgs v[2];
    const o3d_bn_bwd_fin_args* q[2] = {pa, pb};
    for (int i = 0; i < 2; ++i) {
        const o3d_bn_bwd_fin_args& x = *q[i];
        if (!x.part || x.nparts <= 0 || x.C <= 0 || !x.mean || !x.invstd || !x.dgamma || !x.dbeta || !x.A1 || !x.A2 || !x.A3)
            return O3D_EINVAL;
        v[i] = BnBwdFinArgs{x.part, x.nparts, x.C, x.count, nullptr, 1, x.gamma, x.mean, x.invstd, x.dgamma, x.dbeta, x.A1, x.A2,
                            x.A3};
    }
    const int cmax = pa->C > pb->C ? pa->C : pb->C;
    hipLaunchKernelGGL(bn_bwd_finalize_pair_kernel, dim3(o3d_cdiv(cmax, FIN_CH), 2), dim3(256), 0, o3d_stream(stream), v[0],
                       v[1]);
    return o3d_launch_status();
}

extern "C" int o3d_bn_relu_maxpool_fwd(const float* Y, const float* scale, const float* shift, int B,
                                       int C, int npoint, int ns, float* out, int32_t* arg,
                                       float* yarg, void* stream) {
    if (B <= 0 || C <= 0 || npoint <= 0 || ns <= 0 || ns % 4 != 0 || !Y || !scale || !shift || !out ||
        (arg && !yarg))
        return O3D_EINVAL;
    const long total = (long)B * C * npoint;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(o3d_cdiv(total * 8, 256)), dim3(256), 0, o3d_stream(stream), Y,
                       scale, shift, C, npoint, ns, total, out, arg, yarg);
    return o3d_launch_status();
}

// one cloud of npoint balls (flat (C, npoint) pooled tensors): part [nsplit][2][C], pk (C, npoint, 2) or NULL
extern "C" int o3d_pool_bwd_partials_split(const float* dOut, const float* out, const float* yarg, const float* mean,
                                           int C, int npoint, int nsplit, float* part, const int32_t* arg, float* pk,
                                           void* stream) {
    if (C <= 0 || npoint <= 0 || nsplit <= 0 || nsplit > 64 || !dOut || !out || !yarg || !mean || !part || (pk && !arg))
        return O3D_EINVAL;
    hipLaunchKernelGGL(pool_bwd_partials_split_kernel, dim3(C, nsplit), dim3(256), 0, o3d_stream(stream), dOut, out, yarg,
                       mean, C, npoint, part, arg, reinterpret_cast<float2*>(pk));
    return o3d_launch_status();
}

extern "C" int o3d_bn_bwd_finalize(const float* part, int nparts, int C, double count, const float* gamma,
                                   const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                   float* A1, float* A2, float* A3, float* fold, void* stream) {
    if (!part || nparts <= 0 || C <= 0 || !mean || !invstd || !dgamma || !dbeta || !A1 || !A2 || !A3)
        return O3D_EINVAL;
    part = fold_partials(part, nparts, C, fold, o3d_stream(stream));
    BnBwdFinArgs a = {part, nparts, C, count, nullptr, 1, gamma, mean, invstd, dgamma, dbeta, A1, A2, A3};
    return launch(bn_bwd_finalize_kernel, dim3(o3d_cdiv(C, FIN_CH)), dim3(256), 0, o3d_stream(stream), a);
}

// compact layout: only the first meta[0]/tile partial rows are live (tile = positions per partial row)
extern "C" int o3d_bn_finalize_c(const float* part, int nparts, int C, double count, const float* stat_c,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                                 const int32_t* meta, int tile, void* stream) {
    if (!part || nparts <= 0 || C <= 0 || !mean || !invstd || !scale || !shift || !meta || tile <= 0) return O3D_EINVAL;
    BnFinArgs a = {part, nparts, C, meta, tile, count, stat_c, gamma, beta, running_mean, running_var, momentum, eps,
                   mean, invstd, scale, shift};
    if (tile == 128) { a.tail_slots = o3d_direct_tail_slots(C); a.extra_row0 = nparts; }     // rows of a direct GEMM launch
    return launch(bn_finalize_kernel, dim3(o3d_cdiv(C, FIN_CH)), dim3(256), 0, o3d_stream(stream), a);
}

extern "C" int o3d_bn_bwd_finalize_c(const float* part, int nparts, int C, double count, const float* gamma,
                                     const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                     float* A1, float* A2, float* A3, const int32_t* meta, int tile, void* stream) {
    if (!part || nparts <= 0 || C <= 0 || !mean || !invstd || !dgamma || !dbeta || !A1 || !A2 || !A3 || !meta ||
        tile <= 0)
        return O3D_EINVAL;
    BnBwdFinArgs a = {part, nparts, C, count, meta, tile, gamma, mean, invstd, dgamma, dbeta, A1, A2, A3};
    if (tile == 128) { a.tail_slots = o3d_direct_tail_slots(C); a.extra_row0 = nparts; }
    return launch(bn_bwd_finalize_kernel, dim3(o3d_cdiv(C, FIN_CH)), dim3(256), 0, o3d_stream(stream), a);
}

// Two segments in one launch (see BnFinArgs): partial rows [nparts0 | nparts1], stat_c / outputs (2, C),
// meta (2, 4).  Equivalent to o3d_bn_finalize_c for segment 0 followed by segment 1.
extern "C" int o3d_bn_finalize_c2(const float* part, int nparts0, int nparts1, int C, double count0, double count1,
                                  const float* stat_c, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, float* mean, float* invstd,
                                  float* scale, float* shift, const int32_t* meta, int tile, void* stream) {
    if (!part || nparts0 <= 0 || nparts1 <= 0 || C <= 0 || !mean || !invstd || !scale || !shift || !meta || tile <= 0)
        return O3D_EINVAL;
    BnFinArgs a = {part, nparts0, C, meta, tile, count0, stat_c, gamma, beta, running_mean, running_var, momentum, eps,
                   mean, invstd, scale, shift, nparts1, count1};
    if (tile == 128) { a.tail_slots = o3d_direct_tail_slots(C); a.extra_row0 = (long)nparts0 + nparts1; }
    return launch(bn_finalize_kernel, dim3(o3d_cdiv(C, FIN_CH)), dim3(256), 0, o3d_stream(stream), a);
}

// Two segments in one launch; meta may be NULL (every partial row live).  dgamma / dbeta (C) = sum over segments.
extern "C" int o3d_bn_bwd_finalize_c2(const float* part, int nparts0, int nparts1, int C, double count0, double count1,
                                      const float* gamma, const float* mean, const float* invstd, float* dgamma,
                                      float* dbeta, float* A1, float* A2, float* A3, const int32_t* meta, int tile,
                                      void* stream) {
    if (!part || nparts0 <= 0 || nparts1 <= 0 || C <= 0 || !mean || !invstd || !dgamma || !dbeta || !A1 || !A2 || !A3 ||
        tile <= 0)
        return O3D_EINVAL;
    BnBwdFinArgs a = {part, nparts0, C, count0, meta, tile, gamma, mean, invstd, dgamma, dbeta, A1, A2, A3, nparts1,
                      count1};
    if (tile == 128 && meta) { a.tail_slots = o3d_direct_tail_slots(C); a.extra_row0 = (long)nparts0 + nparts1; }
    return launch(bn_bwd_finalize_kernel, dim3(o3d_cdiv(C, FIN_CH)), dim3(256), 0, o3d_stream(stream), a);
}

static int fill_dy(DyArgs& d, const float* dN, const float* dOut, const float* out, const int32_t* arg,
                   const float* Y, const float* A1, const float* A2, const float* A3, int ns) {
    if (!Y || !A1 || !A2 || !A3) return O3D_EINVAL;
    if (!dN && (!dOut || !out || !arg || ns <= 0 || ns % 4 != 0)) return O3D_EINVAL;
    d.dN = dN; d.dOut = dOut; d.out = out; d.arg = arg; d.Y = Y; d.A1 = A1; d.A2 = A2; d.A3 = A3;
    d.ns = ns > 0 ? ns : 4;
    return O3D_OK;
}

// data gradient w.r.t. the (BN+ReLU'd) input of a layer; see DgradArgs for the two epilogues
extern "C" int o3d_mlp_conv_dgrad(const float* dN, const float* dOut, const float* out, const int32_t* arg,
                                  int ns, const float* Y, const float* A1, const float* A2, const float* A3,
                                  const float* W, int B, int Cin, int Cout, int P,
                                  const float* Yprev, const float* scale_p, const float* shift_p,
                                  const float* mean_p, float* dNprev, float* part, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || P <= 0 || P % BN_POS != 0 || !W || !Yprev || !scale_p ||
        !shift_p || !mean_p || !dNprev || !part)
        return O3D_EINVAL;
    DgradArgs a = {};
    if (fill_dy(a.dy, dN, dOut, out, arg, Y, A1, A2, A3, ns) != O3D_OK) return O3D_EINVAL;
    a.W = W; a.B = B; a.Cin = Cin; a.Cout = Cout; a.P = P; a.c_lo = 0; a.M = Cin;
    a.Yprev = Yprev; a.scale_p = scale_p; a.shift_p = shift_p; a.mean_p = mean_p; a.dNprev = dNprev; a.part = part;
    return dN ? launch_dgrad<false, 0>(a, o3d_stream(stream)) : launch_dgrad<true, 0>(a, o3d_stream(stream));
}

// Same contract as o3d_mlp_conv_dgrad plus Wt = W^T (Cin,Cout) contiguous and, for the pooled source,
// pk (B,Cout,npoint) float2 = {dOut where out > 0 else 0, bits(arg)} from o3d_pool_bwd_partials: shapes with Cin % 64 == 0 and
// Cout % 16 == 0 take the LDS-free kernel of mlp_direct.hip (its A operand is read along Cout).
extern "C" int o3d_mlp_conv_dgrad_wt(const float* dN, const float* dOut, const float* out, const int32_t* arg,
                                     int ns, const float* Y, const float* A1, const float* A2, const float* A3,
                                     const float* W, const float* Wt, const float* pk, int B, int Cin, int Cout,
                                     int P, const float* Yprev, const float* scale_p, const float* shift_p,
                                     const float* mean_p, float* dNprev, float* part, void* stream) {
    if (Wt && o3d_direct_ok(Cin, Cout, P) && B > 0 && Yprev && scale_p && shift_p && mean_p && dNprev && part &&
        Y && A1 && A2 && A3 && (dN || (pk && ns > 0 && ns % 4 == 0)))
        return o3d_direct_dgrad(dN, pk, ns, Y, A1, A2, A3, Wt, B, Cin, Cout, P, Yprev, scale_p, shift_p,
                                mean_p, dNprev, part, nullptr, nullptr, 0, 128, o3d_stream(stream));
    return o3d_mlp_conv_dgrad(dN, dOut, out, arg, ns, Y, A1, A2, A3, W, B, Cin, Cout, P, Yprev, scale_p, shift_p,
                              mean_p, dNprev, part, stream);
}

// ---- compact (distinct-neighbour) layout, csrc/compact.hip: flat (C, ldp) matrices, per-position weights w
// (ldp) and the live column count meta[0] in device memory.  Aligned shapes only.
extern "C" int o3d_mlp_conv_fwd_c(const float* X, const float* W, const float* in_scale, const float* in_shift,
                                  int Cin, int Cout, long ldp, const float* w, const int32_t* meta, long start1,
                                  int tile, float* Y, float* part, const float* stat_c, void* stream) {
    if (!X || !W || !Y || !w || !meta || !in_scale || !in_shift || ldp <= 0 || ldp > 0x7fffffff ||
        !o3d_direct_ok(Cout, Cin, (int)ldp) || (tile != 64 && tile != 128))
        return O3D_EINVAL;
    return o3d_direct_fwd(X, W, in_scale, in_shift, 1, Cin, Cout, (int)ldp, Y, part, stat_c, w, meta, start1, tile,
                          o3d_stream(stream));
}

extern "C" int o3d_mlp_conv_dgrad_c(const float* dN, const float* Y, const float* A1, const float* A2,
                                    const float* A3, const float* Wt, int Cin, int Cout, long ldp, const float* w,
                                    const int32_t* meta, long start1, int tile, const float* Yprev,
                                    const float* scale_p,
                                    const float* shift_p, const float* mean_p, float* dNprev, float* part,
                                    void* stream) {
    // Y == NULL (then A1..A3 are ignored): dN is the finished dY of o3d_mlp_conv_wgrad2_c_dy
    if (!dN || (Y && (!A1 || !A2 || !A3)) || !Wt || !w || !meta || !Yprev || !scale_p || !shift_p || !mean_p ||
        !dNprev || !part || ldp <= 0 || ldp > 0x7fffffff || !o3d_direct_ok(Cin, Cout, (int)ldp) ||
        (tile != 64 && tile != 128))
        return O3D_EINVAL;
    return o3d_direct_dgrad(dN, nullptr, 4, Y, A1, A2, A3, Wt, 1, Cin, Cout, (int)ldp, Yprev, scale_p, shift_p,
                            mean_p, dNprev, part, w, meta, start1, tile, o3d_stream(stream));
}

// dX (B,Cin,P) = W^T dY with dY = A1*dN + A2*Y + A3, no mask, no statistics: the gradient w.r.t. an
// operand that is not the BN+ReLU output of a previous layer (the per-point operand of layer 0).
extern "C" int o3d_mlp_conv_dgrad_plain(const float* dN, const float* Y, const float* A1, const float* A2,
                                        const float* A3, const float* W, int B, int Cin, int Cout, int P,
                                        float* dX, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || P <= 0 || P % BN_POS != 0 || !W || !dN || !dX) return O3D_EINVAL;
    DgradArgs a = {};
    if (fill_dy(a.dy, dN, nullptr, nullptr, nullptr, Y, A1, A2, A3, 4) != O3D_OK) return O3D_EINVAL;
    a.W = W; a.B = B; a.Cin = Cin; a.Cout = Cout; a.P = P; a.c_lo = 0; a.M = Cin; a.dNprev = dX;
    return launch_dgrad<false, 2>(a, o3d_stream(stream));
}

// dW[i] = sum over slices of part[z][i], in a fixed order (two stages above 32 slices; scratch2 holds
// 16 * n floats)
void o3d_wgrad_reduce(const float* part, int nslices, long n, float* scratch2, float* dW, hipStream_t s) {
    (void)scratch2;
    hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3(o3d_cdiv(n, 32)), dim3(256), 0, s, part, nslices, n, dW);
}

// weight gradient.  `part` is scratch of (nslices+16)*Cout*Cin floats; dW (Cout,Cin) is overwritten.
// X source: (X, in_scale, in_shift) (in_scale NULL = identity).  The gather arguments (xyz ... inv_radius) belonged to
// the slot-wise layer-0 path retired in round 4 (layer 0 runs on the points, csrc/compact.hip): pass NULL / 0.
extern "C" int o3d_mlp_conv_wgrad(const float* dN, const float* dOut, const float* out, const int32_t* arg,
                                  int ns, const float* Y, const float* A1, const float* A2, const float* A3,
                                  const float* X, const float* in_scale, const float* in_shift,
                                  const float* xyz, const float* new_xyz, const float* feats,
                                  const int32_t* idx, int N, int C, int nxyz, float inv_radius,
                                  int B, int Cin, int Cout, int P, int nslices, float* part, float* dW,
                                  void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || P <= 0 || P % WBK != 0 || nslices <= 0 || !part || !dW)
        return O3D_EINVAL;
    (void)xyz; (void)new_xyz; (void)feats; (void)idx; (void)N; (void)C; (void)nxyz; (void)inv_radius;
    if (!X) return O3D_EINVAL;
    WgradArgs a = {};
    if (fill_dy(a.dy, dN, dOut, out, arg, Y, A1, A2, A3, ns) != O3D_OK) return O3D_EINVAL;
    a.X = X; a.in_scale = in_scale; a.in_shift = in_shift;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.P = P;
    a.total_chunks = B * (P / WBK);
    a.chunks_per_block = (a.total_chunks + nslices - 1) / nslices;
    a.part = part;
    a.tiles_ci = o3d_cdiv(Cin, 128); a.tiles_co = o3d_cdiv(Cout, 128); a.nslices = nslices;
    const dim3 grid(a.tiles_ci * a.tiles_co * nslices), block(256);
    const size_t lds = sizeof(float) * 4 * 128 * WLD;
    hipStream_t s = o3d_stream(stream);
    int rc;
    const bool pooled = dN == nullptr;
    if (in_scale) {
        rc = pooled ? launch(conv_wgrad_kernel<true, true>, grid, block, lds, s, a)
                    : launch(conv_wgrad_kernel<false, true>, grid, block, lds, s, a);
    } else {
        rc = pooled ? launch(conv_wgrad_kernel<true, false>, grid, block, lds, s, a)
                    : launch(conv_wgrad_kernel<false, false>, grid, block, lds, s, a);
    }
    if (rc != O3D_OK) return rc;
    o3d_wgrad_reduce(part, nslices, (long)Cout * Cin, part + (long)nslices * Cout * Cin, dW, s);
    return o3d_launch_status();
}
