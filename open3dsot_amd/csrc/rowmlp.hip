// rowmlp.hip -- Linear (+ BatchNorm1d over the rows) (+ ReLU) on a handful of rows: the heads of M2-Track.
//
// models/m2track.py:43-71 builds four heads `Linear(256,128) -> BatchNorm1d -> ReLU -> Linear(128,128) -> BatchNorm1d ->
// ReLU -> Linear(128, out)` and models/backbone/pointnet.py:118-126 two `Linear -> BatchNorm1d -> ReLU` rows behind the
// global max of each MiniPointNet: activations of B <= 64 rows (one per frame pair of the batch) x <= 512 features.  On torch
// ops a row is ~6 launches forward and ~8 backward (addmm, batch_norm statistics / transform / update, threshold, their
// mirrors): 144 launches, 0.85 ms of a 7.95 ms M2-Track step for 0.1 GFLOP (profiles/r03_steady_state_per_step_m2track.txt).
//
// A BatchNorm1d over the ROWS normalises each feature on its own: a workgroup that owns a slice of the output features
// sees all of their statistics -- no cross-workgroup reduction, no finalize, one launch per layer each way:
//   forward  (grid = Cout / 8): Z[:, f] = X . W[f, :] + b[f]  (all rows, its 8 features; the weight slice in LDS, the rows
//            straight from global memory / L2 with many loads in flight), batch mean / biased variance of those features over the rows (two passes over
//            registers), y = relu(gamma * (z - mean) * invstd + beta), running statistics updated like torch (momentum,
//            unbiased variance); saves Z, mean, invstd for the backward.
//   backward (grid = Cout / 8): the incoming gradient of its 8 features is either given (top of the stack) or computed
//            on the spot from the layer above, dY[:, f] = dZ_up . W_up[:, f] (so a stack's data gradients need no launch of
//            their own); ReLU mask, dbeta, dgamma, dZ = gamma * invstd * (dN - mean(dN) - xhat * mean(dN * xhat)) for its
//            features, then dW[f, :] = dZ[:, f]^T . X and db[f] = sum dZ[:, f].  The gradient of the stack's INPUT is the same
//            kernel in its "input" mode (first phase only).
// fp32 FMA on the vector ALUs: 25 MFLOP per layer at most, spread over 8-32 workgroups -- latency, not arithmetic; MFMA
// tiles of 32 rows would be half empty at 48 rows.  Everything is deterministic (fixed summation orders).
#include "o3d_common.hpp"

namespace {

constexpr int RM_FT = 8;         // output features per workgroup
constexpr int RM_RG = 256 / RM_FT;            // row groups: thread = (feature fl = tid % FT, row group rg = tid / FT)
constexpr int RM_RMAX = 64;      // rows
constexpr int RM_RPT = RM_RMAX / RM_RG;       // rows per thread: rg + RM_RG * i
constexpr int RM_KMAX = 1024;    // input features (the weight slice of a workgroup lives in LDS: KMAX * FT floats)

using RowFwd = o3d_row_fwd_args;      // include/o3dsot.h

// acc[i] += sum_k A[row(i)][k] * Wt[k][fl] for this thread's rows; A (R, K) row-major (lda), Wt element (k, f) at
// Wsrc[k * wk + f * wf] -- W[f][k] (wk = 1, wf = K) forward, W_up[k][f] (wk = ldw, wf = 1) backward.
// The workgroup's weight slice is staged in LDS ONCE (one barrier); the rows are read straight from global memory with every
// load of eight k-groups in flight before the first use -- the first version staged A through LDS in chunks of 64 k behind
// two barriers per chunk and took ~20 us per layer: eight exposed L2 round trips on a launch of 8-32 workgroups.
__device__ __forceinline__ void rows_times_slice(const float* __restrict__ A, int lda, int R, int K, const float* __restrict__ Wsrc,
                                                 long wk, long wf, int f0, int nf, float* Ws, float (&acc)[RM_RPT]) {
    const int tid = threadIdx.x, fl = tid % RM_FT, rg = tid / RM_FT;
    const int K4 = (K + 3) >> 2;
    for (int e = tid; e < K4 * 4 * RM_FT; e += 256) {       // Ws[(k / 4) * FT + f][k % 4], zero beyond K and nf
        int k, f;
        if (wk == 1) { f = e / (K4 * 4); k = e - f * (K4 * 4); }      // W[f][k]: consecutive threads along k
        else { k = e / RM_FT; f = e - k * RM_FT; }                    // W_up[k][f]: consecutive threads along f
        float v = 0.f;
        if (k < K && f < nf) v = Wsrc[(long)k * wk + (long)(f0 + f) * wf];
        Ws[((k >> 2) * RM_FT + f) * 4 + (k & 3)] = v;
    }
    __syncthreads();
    const float* row[RM_RPT];
#pragma unroll
    for (int i = 0; i < RM_RPT; ++i) {
        const int r = rg + RM_RG * i;
        row[i] = A + (long)(r < R ? r : R - 1) * lda;               // clamped: rows beyond R are computed and never used
    }
    const bool vec = (K & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0;
    if (vec) {
#pragma unroll 8
        for (int k4 = 0; k4 < K4; ++k4) {
            const float4 w = *reinterpret_cast<const float4*>(&Ws[(k4 * RM_FT + fl) * 4]);
#pragma unroll
            for (int i = 0; i < RM_RPT; ++i) {
                const float4 x = *reinterpret_cast<const float4*>(row[i] + 4 * k4);
                acc[i] = fmaf(x.x, w.x, acc[i]);
                acc[i] = fmaf(x.y, w.y, acc[i]);
                acc[i] = fmaf(x.z, w.z, acc[i]);
                acc[i] = fmaf(x.w, w.w, acc[i]);
            }
        }
    } else {
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float w = Ws[((k >> 2) * RM_FT + fl) * 4 + (k & 3)];
#pragma unroll
            for (int i = 0; i < RM_RPT; ++i) acc[i] = fmaf(row[i][k], w, acc[i]);
        }
    }
    __syncthreads();
}

// sum over the rows of a per-thread partial: the row groups of a feature meet in LDS, fixed order
__device__ __forceinline__ float rows_sum(float v, float* red) {
    const int tid = threadIdx.x, fl = tid % RM_FT, rg = tid / RM_FT;
    red[rg * RM_FT + fl] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < RM_RG; ++g) s += red[g * RM_FT + fl];
    __syncthreads();
    return s;
}

__device__ __forceinline__ void row_fwd_body(const RowFwd& a, float* Ws, float* red) {
    const int tid = threadIdx.x, fl = tid % RM_FT, rg = tid / RM_FT;
    const int f0 = blockIdx.x * RM_FT, f = f0 + fl;
    const int nf = a.Cout - f0 < RM_FT ? a.Cout - f0 : RM_FT;
    float acc[RM_RPT];
#pragma unroll
    for (int i = 0; i < RM_RPT; ++i) acc[i] = 0.f;
    rows_times_slice(a.X, a.ldx, a.R, a.Cin, a.W, 1, a.Cin, f0, nf, Ws, acc);
    const bool live = fl < nf;
    const float b = (live && a.bias) ? a.bias[f] : 0.f;
    float z[RM_RPT];
    bool rowok[RM_RPT];
#pragma unroll
    for (int i = 0; i < RM_RPT; ++i) { rowok[i] = rg + RM_RG * i < a.R; z[i] = acc[i] + b; }
    float sc = 1.f, sh = 0.f;
    if (a.gamma) {
        float mu, istd;
        if (a.training) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < RM_RPT; ++i) s += rowok[i] ? z[i] : 0.f;
            mu = rows_sum(s, red) / (float)a.R;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < RM_RPT; ++i) q += rowok[i] ? (z[i] - mu) * (z[i] - mu) : 0.f;
            const float var = rows_sum(q, red) / (float)a.R;
            istd = rsqrtf(var + a.eps);
            istd = istd * (1.5f - 0.5f * (var + a.eps) * istd * istd);          // one Newton step: full fp32 accuracy
            if (live && rg == 0 && a.running_mean) {
                const float unb = a.R > 1 ? var * (float)a.R / (float)(a.R - 1) : var;
                a.running_mean[f] = (1.f - a.momentum) * a.running_mean[f] + a.momentum * mu;
                a.running_var[f] = (1.f - a.momentum) * a.running_var[f] + a.momentum * unb;
            }
        } else {
            mu = live ? a.running_mean[f] : 0.f;
            const float v = (live ? a.running_var[f] : 1.f) + a.eps;
            istd = rsqrtf(v);
            istd = istd * (1.5f - 0.5f * v * istd * istd);
        }
        if (live && rg == 0 && a.mean) { a.mean[f] = mu; a.invstd[f] = istd; }
        sc = (live ? a.gamma[f] : 1.f) * istd;
        sh = fmaf(-mu, sc, live ? a.beta[f] : 0.f);        // (spelled out: the backward recomputes y with this very expression)
    }
    if (!live) return;
#pragma unroll
    for (int i = 0; i < RM_RPT; ++i) {
        if (!rowok[i]) continue;
        const long o = (long)(rg + RM_RG * i) * a.Cout + f;
        if (a.Z) a.Z[o] = z[i];
        float y = fmaf(z[i], sc, sh);
        if (a.relu) y = fmaxf(y, 0.f);
        a.Y[o] = y;
    }
}

__global__ __launch_bounds__(256) void row_mlp_fwd_kernel(RowFwd a) {
    __shared__ __attribute__((aligned(16))) float Ws[RM_KMAX * RM_FT];
    __shared__ float red[RM_RG * RM_FT];
    row_fwd_body(a, Ws, red);
}

// up to RM_GROUP independent layers in one launch (blockIdx.y picks the job): the heads of M2-Track that read the same
// feature advance side by side, 3 launches for 9
constexpr int RM_GROUP = 4;
struct RowFwdGroup { RowFwd a[RM_GROUP]; };
__global__ __launch_bounds__(256) void row_mlp_fwd_group_kernel(RowFwdGroup g) {
    __shared__ __attribute__((aligned(16))) float Ws[RM_KMAX * RM_FT];
    __shared__ float red[RM_RG * RM_FT];
    const RowFwd& a = g.a[blockIdx.y];
    if ((int)blockIdx.x * RM_FT >= a.Cout) return;
    row_fwd_body(a, Ws, red);
}

using RowBwd = o3d_row_bwd_args;      // include/o3dsot.h

struct RowBwdGroup { RowBwd a[RM_GROUP]; int n; };

// `more`: further gradient sources of the same output (input mode only): g += dZup_j . Wup_j for the jobs 1 .. n-1
__device__ __forceinline__ void row_bwd_body(const RowBwd& a, float* Ws, float* red, float* dZs, const RowBwdGroup* more) {
    const int tid = threadIdx.x, fl = tid % RM_FT, rg = tid / RM_FT;
    const int f0 = blockIdx.x * RM_FT, f = f0 + fl;
    const int nf = a.C - f0 < RM_FT ? a.C - f0 : RM_FT;
    const bool live = fl < nf;
    float g[RM_RPT];
    bool rowok[RM_RPT];
#pragma unroll
    for (int i = 0; i < RM_RPT; ++i) { g[i] = 0.f; rowok[i] = rg + RM_RG * i < a.R; }
    if (a.dZup) {
        rows_times_slice(a.dZup, a.Cup, a.R, a.Cup, a.Wup, a.C, 1, f0, nf, Ws, g);
    } else {
#pragma unroll
        for (int i = 0; i < RM_RPT; ++i)
            if (live && rowok[i]) g[i] = a.dY[(long)(rg + RM_RG * i) * a.lddy + f];
    }
    if (a.input_mode) {
        if (more)
            for (int j = 1; j < more->n; ++j)
                rows_times_slice(more->a[j].dZup, more->a[j].Cup, a.R, more->a[j].Cup, more->a[j].Wup, a.C, 1, f0, nf, Ws, g);
#pragma unroll
        for (int i = 0; i < RM_RPT; ++i)
            if (live && rowok[i]) a.dX[(long)(rg + RM_RG * i) * a.lddx + f] = g[i];
        return;
    }
    float dz[RM_RPT];
    if (a.gamma) {
        const float mu = live ? a.mean[f] : 0.f, istd = live ? a.invstd[f] : 1.f;
        const float gm = live ? a.gamma[f] : 1.f, bt = live ? a.beta[f] : 0.f;
        float xh[RM_RPT], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < RM_RPT; ++i) {
            const float zz = (live && rowok[i]) ? a.Z[(long)(rg + RM_RG * i) * a.C + f] : mu;
            xh[i] = (zz - mu) * istd;
            // ReLU mask from y recomputed EXACTLY as the forward computed it (sc = gamma * istd, sh = fma(-mu, sc, beta),
            // y = fma(z, sc, sh)): another association rounds differently and an output within rounding of 0 would get
            // its gradient passed where the forward clamped it (torch masks on the stored output)
            if (a.relu && !(fmaf(zz, gm * istd, fmaf(-mu, gm * istd, bt)) > 0.f)) g[i] = 0.f;
            if (!rowok[i]) g[i] = 0.f;
            s1 += g[i];
            s2 = fmaf(g[i], xh[i], s2);
        }
        s1 = rows_sum(s1, red);
        s2 = rows_sum(s2, red);
        if (live && rg == 0) {
            if (a.dbeta) a.dbeta[f] = s1;
            if (a.dgamma) a.dgamma[f] = s2;
        }
        const float inv_r = 1.f / (float)a.R;
#pragma unroll
        for (int i = 0; i < RM_RPT; ++i)
            dz[i] = a.training ? gm * istd * (g[i] - s1 * inv_r - xh[i] * s2 * inv_r) : gm * istd * g[i];
    } else {
#pragma unroll
        for (int i = 0; i < RM_RPT; ++i) dz[i] = rowok[i] ? g[i] : 0.f;      // (a ReLU without BatchNorm does not occur in these stacks)
    }
#pragma unroll
    for (int i = 0; i < RM_RPT; ++i) {
        const int r = rg + RM_RG * i;
        const float v = (live && rowok[i]) ? dz[i] : 0.f;
        dZs[r * RM_FT + fl] = v;
        if (live && rowok[i] && a.dZ) a.dZ[(long)r * a.C + f] = v;
    }
    __syncthreads();
    if (a.db && tid < nf) {
        float s = 0.f;
        for (int r = 0; r < a.R; ++r) s += dZs[r * RM_FT + tid];
        a.db[f0 + tid] = s;
    }
    if (a.dW) {      // dW[f0 + j][c] = sum_r dZs[r][j] * X[r][c]: thread = input feature c (+ 256 per pass), all 16 output features
        for (int c = tid; c < a.Cin; c += 256) {
            float w[RM_FT];
#pragma unroll
            for (int j = 0; j < RM_FT; ++j) w[j] = 0.f;
            for (int r = 0; r < a.R; ++r) {
                const float x = a.X[(long)r * a.ldx + c];
#pragma unroll
                for (int j4 = 0; j4 < RM_FT; j4 += 4) {
                    const float4 d = *reinterpret_cast<const float4*>(&dZs[r * RM_FT + j4]);
                    w[j4 + 0] = fmaf(d.x, x, w[j4 + 0]);
                    w[j4 + 1] = fmaf(d.y, x, w[j4 + 1]);
                    w[j4 + 2] = fmaf(d.z, x, w[j4 + 2]);
                    w[j4 + 3] = fmaf(d.w, x, w[j4 + 3]);
                }
            }
#pragma unroll
            for (int j = 0; j < RM_FT; ++j)
                if (j < nf) a.dW[(long)(f0 + j) * a.Cin + c] = w[j];
        }
    }
}

__global__ __launch_bounds__(256) void row_mlp_bwd_kernel(RowBwd a) {
    __shared__ __attribute__((aligned(16))) float Ws[RM_KMAX * RM_FT];
    __shared__ float red[RM_RG * RM_FT];
    __shared__ __attribute__((aligned(16))) float dZs[RM_RMAX * RM_FT];
    row_bwd_body(a, Ws, red, dZs, nullptr);
}

__global__ __launch_bounds__(256) void row_mlp_bwd_group_kernel(RowBwdGroup g) {
    __shared__ __attribute__((aligned(16))) float Ws[RM_KMAX * RM_FT];
    __shared__ float red[RM_RG * RM_FT];
    __shared__ __attribute__((aligned(16))) float dZs[RM_RMAX * RM_FT];
    const RowBwd& a = g.a[blockIdx.y];
    if ((int)blockIdx.x * RM_FT >= a.C) return;
    row_bwd_body(a, Ws, red, dZs, nullptr);
}

// the gradient of rows that several stacks read: dX = sum over the jobs of dZup_j . Wup_j (fixed order)
__global__ __launch_bounds__(256) void row_mlp_input_grad_kernel(RowBwdGroup g) {
    __shared__ __attribute__((aligned(16))) float Ws[RM_KMAX * RM_FT];
    row_bwd_body(g.a[0], Ws, nullptr, nullptr, &g);
}

static bool row_fwd_ok(const RowFwd& a) {
    return a.X && a.W && a.Y && a.R > 0 && a.R <= RM_RMAX && a.Cin > 0 && a.Cin <= RM_KMAX && a.Cout > 0 && a.ldx >= a.Cin &&
           !(a.gamma && !a.beta) && !(a.gamma && !a.training && (!a.running_mean || !a.running_var)) && !(a.mean && !a.invstd);
}

static bool row_bwd_ok(const RowBwd& a) {
    if (a.R <= 0 || a.R > RM_RMAX || a.C <= 0 || (!a.dY && (!a.dZup || !a.Wup || a.Cup <= 0 || a.Cup > RM_KMAX)) ||
        (a.dY && a.lddy < a.C))
        return false;
    if (a.input_mode) return a.dX && a.lddx >= a.C;
    return !((a.gamma && (!a.beta || !a.mean || !a.invstd || !a.Z)) || (a.dW && (!a.X || a.Cin <= 0 || a.ldx < a.Cin)) ||
             (a.relu && !a.gamma));
}

}  // namespace

extern "C" int o3d_row_mlp_fwd_group(const o3d_row_fwd_args* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0 || njobs > RM_GROUP) return O3D_EINVAL;
    RowFwdGroup g = {};
    int cmax = 0;
    for (int j = 0; j < njobs; ++j) {
        if (!row_fwd_ok(jobs[j])) return O3D_EINVAL;
        g.a[j] = jobs[j];
        cmax = jobs[j].Cout > cmax ? jobs[j].Cout : cmax;
    }
    hipLaunchKernelGGL(row_mlp_fwd_group_kernel, dim3(o3d_cdiv(cmax, RM_FT), njobs), dim3(256), 0, o3d_stream(stream), g);
    return o3d_launch_status();
}

extern "C" int o3d_row_mlp_bwd_group(const o3d_row_bwd_args* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0 || njobs > RM_GROUP) return O3D_EINVAL;
    RowBwdGroup g = {};
    int cmax = 0;
    for (int j = 0; j < njobs; ++j) {
        if (!row_bwd_ok(jobs[j]) || jobs[j].input_mode) return O3D_EINVAL;
        g.a[j] = jobs[j];
        cmax = jobs[j].C > cmax ? jobs[j].C : cmax;
    }
    g.n = njobs;
    hipLaunchKernelGGL(row_mlp_bwd_group_kernel, dim3(o3d_cdiv(cmax, RM_FT), njobs), dim3(256), 0, o3d_stream(stream), g);
    return o3d_launch_status();
}

// dX (R, C) = sum_j dZup_j (R, Cup_j) . Wup_j (Cup_j, C); R, C, dX, lddx from jobs[0]
extern "C" int o3d_row_mlp_input_grad(const o3d_row_bwd_args* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0 || njobs > RM_GROUP) return O3D_EINVAL;
    RowBwdGroup g = {};
    for (int j = 0; j < njobs; ++j) {
        RowBwd a = jobs[j];
        a.dY = nullptr; a.input_mode = 1; a.R = jobs[0].R; a.C = jobs[0].C; a.dX = jobs[0].dX; a.lddx = jobs[0].lddx;
        if (!row_bwd_ok(a)) return O3D_EINVAL;
        g.a[j] = a;
    }
    g.n = njobs;
    hipLaunchKernelGGL(row_mlp_input_grad_kernel, dim3(o3d_cdiv(jobs[0].C, RM_FT)), dim3(256), 0, o3d_stream(stream), g);
    return o3d_launch_status();
}

extern "C" int o3d_row_mlp_fwd(const float* X, int ldx, const float* W, const float* bias, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, int training, int relu,
                               int R, int Cin, int Cout, float* Z, float* Y, float* mean, float* invstd, void* stream) {
    if (!X || !W || !Y || R <= 0 || R > RM_RMAX || Cin <= 0 || Cin > RM_KMAX || Cout <= 0 || ldx < Cin || (gamma && !beta) ||
        (gamma && !training && (!running_mean || !running_var)) || (mean && !invstd))
        return O3D_EINVAL;
    RowFwd a = {X, ldx, W, bias, gamma, beta, running_mean, running_var, momentum, eps, training, relu, R, Cin, Cout, Z, Y, mean,
                invstd};
    hipLaunchKernelGGL(row_mlp_fwd_kernel, dim3(o3d_cdiv(Cout, RM_FT)), dim3(256), 0, o3d_stream(stream), a);
    return o3d_launch_status();
}

// One layer of the backward (see the header of this file).  Gradient source: dY (R, C; row stride lddy) or, dY == NULL,
// dZup (R, Cup) . Wup (Cup, C).  input_mode != 0: only that product, stored to dX (R, C; row stride lddx).
extern "C" int o3d_row_mlp_bwd(const float* dY, int lddy, const float* dZup, const float* Wup, int Cup, int input_mode,
                               float* dX, int lddx, const float* Z, const float* gamma, const float* beta, const float* mean,
                               const float* invstd, int training, int relu, const float* X, int ldx, int R, int Cin, int C,
                               float* dZ, float* dW, float* db, float* dgamma, float* dbeta, void* stream) {
    if (R <= 0 || R > RM_RMAX || C <= 0 || (!dY && (!dZup || !Wup || Cup <= 0 || Cup > RM_KMAX)) || (dY && lddy < C))
        return O3D_EINVAL;
    if (input_mode) {
        if (!dX || lddx < C) return O3D_EINVAL;
    } else if ((gamma && (!beta || !mean || !invstd || !Z)) || (dW && (!X || Cin <= 0 || ldx < Cin)) || (relu && !gamma)) {
        return O3D_EINVAL;
    }
    RowBwd a = {dY, lddy, dZup, Wup, Cup, input_mode, dX, lddx, Z, gamma, beta, mean, invstd, training, relu, X, ldx, R, Cin, C,
                dZ, dW, db, dgamma, dbeta};
    hipLaunchKernelGGL(row_mlp_bwd_kernel, dim3(o3d_cdiv(C, RM_FT)), dim3(256), 0, o3d_stream(stream), a);
    return o3d_launch_status();
}
