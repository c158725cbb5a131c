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

struct RowFwd {
    const float* X; int ldx;     // (R, Cin), row stride ldx
    const float* W;              // (Cout, Cin)
    const float* bias;           // (Cout) or NULL
    const float* gamma; const float* beta;     // BatchNorm affine or NULL (no BatchNorm)
    float* running_mean; float* running_var;   // updated when training (may be NULL)
    float momentum, eps;
    int training, relu;
    int R, Cin, Cout;
    float* Z;                    // (R, Cout) pre-BatchNorm output, or NULL (not needed: no BatchNorm / no backward)
    float* Y;                    // (R, Cout)
    float* mean; float* invstd;  // (Cout) each: the constants the backward normalises with (batch or running statistics)
};

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

__global__ __launch_bounds__(256) void row_mlp_fwd_kernel(RowFwd a) {
    __shared__ __attribute__((aligned(16))) float Ws[RM_KMAX * RM_FT];
    __shared__ float red[RM_RG * RM_FT];
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
        sh = (live ? a.beta[f] : 0.f) - mu * sc;
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

struct RowBwd {
    // incoming gradient of this layer's OUTPUT (R, C): given, or dZ_up (R, Cup) . W_up (Cup, C)
    const float* dY; int lddy;
    const float* dZup; const float* Wup; int Cup;
    int input_mode;              // 1: the "layer" is the stack's input: store the incoming gradient as dX and stop
    float* dX; int lddx;
    // this layer
    const float* Z;              // (R, C) pre-BatchNorm output (BatchNorm layers)
    const float* gamma; const float* beta; const float* mean; const float* invstd;     // NULL: no BatchNorm
    int training, relu;
    const float* X; int ldx;     // (R, Cin) layer input
    int R, Cin, C;
    float* dZ;                   // (R, C) out: gradient of the pre-BatchNorm output (read by the layer below)
    float* dW; float* db;        // (C, Cin), (C) or NULL
    float* dgamma; float* dbeta; // (C) or NULL
};

__global__ __launch_bounds__(256) void row_mlp_bwd_kernel(RowBwd a) {
    __shared__ __attribute__((aligned(16))) float Ws[RM_KMAX * RM_FT];
    __shared__ float red[RM_RG * RM_FT];
    __shared__ __attribute__((aligned(16))) float dZs[RM_RMAX * RM_FT];
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
            if (a.relu && !(fmaf(xh[i], gm, bt) > 0.f)) g[i] = 0.f;
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

}  // namespace

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
