// xcorr.hip -- layer 0 of P2B's template<->search fusion (models/head/xcorr.py:25-53) without the
// (B, 4+f, M, N) fusion tensor.
//
// The reference builds, for every (template point i, search point j) pair, the vector
//     x_ij = [ cos_sim(t_i, s_j) ; xyz_i ; feat_i ]            (xcorr.py:37-45, 1 + 3 + f channels)
// runs a SharedMLP over the M*N positions and max-pools over the template axis (xcorr.py:47-49).  Only the first
// channel depends on j, so the first 1x1 convolution splits exactly as in csrc/compact.hip:
//     Y0[c, (j,i)] = Z[c, i] + W0[c, 0] * sim[j, i],      Z = W0[:, 1:] . [xyz_i ; feat_i]   (a GEMM over the M points)
// and its backward is the transpose:
//     S[c, i]   = sum_j dY0[c, (j,i)]          -> dW0[:, 1:] = S . [xyz ; feat]^T,  d[xyz ; feat] = W0[:, 1:]^T . S
//     dsim[j,i] = sum_c W0[c, 0] * dY0[c, (j,i)]
//     dW0[c, 0] = sum_{j,i} dY0[c, (j,i)] * sim[j, i]
// with dY0 = A1*dN + A2*Y0 + A3 (BatchNorm backward folded, as everywhere in this library).
// Column order: q = (b*N + j)*M + i -- the M template points of one search point are one "ball" of M contiguous
// columns, so layers >= 1 and the max-pool run on the same kernels as the set-abstraction levels
// (csrc/mlp_direct.hip, csrc/mlp_wgrad.hip, pool_c_kernel of csrc/compact.hip).
#include "o3d_common.hpp"

namespace {

// workgroup = 256 columns x a channel range; wave w takes channels w, w+4, ...; lane = 4 columns
__global__ __launch_bounds__(256) void xcorr_expand_kernel(const float* __restrict__ Z, long ldz,
                                                           const float* __restrict__ sim,
                                                           const float* __restrict__ W0, int ldw, int C0, int M,
                                                           long colsPerCloud, long P, float* __restrict__ Y0,
                                                           float* __restrict__ part, const float* __restrict__ stat_c) {
    const long q0 = (long)blockIdx.x * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long q = q0 + 4 * lane;
    const long b = q / colsPerCloud;
    const int i = (int)(q % M);                                   // M % 4 == 0: the lane's 4 columns share the ball
    const float4 s4 = *reinterpret_cast<const float4*>(&sim[q]);
    const float* zb = Z + b * M + i;
    constexpr int EG = 8;
    const int cper = (C0 + gridDim.y - 1) / gridDim.y;
    const int cbeg = blockIdx.y * cper, cend = cbeg + cper < C0 ? cbeg + cper : C0;
    for (int g0 = cbeg + wave; g0 < cend; g0 += 4 * EG) {
        float4 y[EG];
        float wq[EG], cq[EG];
#pragma unroll
        for (int g = 0; g < EG; ++g) {
            const int co = g0 + 4 * g < cend ? g0 + 4 * g : cend - 1;
            y[g] = *reinterpret_cast<const float4*>(zb + (long)co * ldz);
            // per-channel constants with the row loads, not between the stores below (a load issued after a store makes
            // the wait for it wait for the store as well: vmcnt counts both on gfx9)
            wq[g] = W0[(long)co * ldw];
            cq[g] = (part && stat_c) ? stat_c[co] : 0.f;
        }
        float s[EG], v[EG];
#pragma unroll
        for (int g = 0; g < EG; ++g) {
            const int co = g0 + 4 * g;
            if (co >= cend) { s[g] = 0.f; v[g] = 0.f; continue; }
            const float w = wq[g];
            y[g].x = fmaf(w, s4.x, y[g].x); y[g].y = fmaf(w, s4.y, y[g].y);
            y[g].z = fmaf(w, s4.z, y[g].z); y[g].w = fmaf(w, s4.w, y[g].w);
            *reinterpret_cast<float4*>(&Y0[(long)co * P + q]) = y[g];
            const float c = cq[g];
            s[g] = (y[g].x + y[g].y) + (y[g].z + y[g].w);
            v[g] = (y[g].x - c) * (y[g].x - c) + (y[g].y - c) * (y[g].y - c) + (y[g].z - c) * (y[g].z - c) +
                   (y[g].w - c) * (y[g].w - c);
        }
        if (part) {
#pragma unroll
            for (int g = 0; g < EG; ++g) { s[g] = wave_sum(s[g]); v[g] = wave_sum(v[g]); }
            // the four waves of the workgroup hold disjoint channels: one partial row per 256 columns
            if (lane == 0)
#pragma unroll
                for (int g = 0; g < EG; ++g) {
                    const int co = g0 + 4 * g;
                    if (co < cend) {
                        part[((long)blockIdx.x * 2 + 0) * C0 + co] = s[g];
                        part[((long)blockIdx.x * 2 + 1) * C0 + co] = v[g];
                    }
                }
        }
    }
}

// workgroup = (cloud b, group of CS channels): 256 threads walk the cloud's N*M columns 1024 at a time; a thread's
// 4 columns keep the same template points i0..i0+3 in every pass (1024 % M == 0), so S accumulates in registers.
//   S (C0, B*M)              per template point
//   dsim_part (C0/CS, P)     partial sums over the group's channels (summed by the caller)
//   dw_part (B, C0)          per cloud (summed by the caller)
template <int CS>
__global__ __launch_bounds__(256) void xcorr_reduce_kernel(const float* __restrict__ dN, const float* __restrict__ Y0,
                                                           const float* __restrict__ A1, const float* __restrict__ A2,
                                                           const float* __restrict__ A3, const float* __restrict__ sim,
                                                           const float* __restrict__ W0, int ldw, int C0, int M,
                                                           long colsPerCloud, long P, float* __restrict__ S, long lds_,
                                                           float* __restrict__ dsim_part, float* __restrict__ dw_part) {
    __shared__ float red[16][64 + 1];
    __shared__ float wred[CS][4];
    const int groups = C0 / CS;
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int c0 = g * CS;
    const int tid = threadIdx.x;
    float a1[CS], a2[CS], a3[CS], w0[CS], sacc[CS][4], wacc[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        a1[c] = A1[c0 + c]; a2[c] = A2[c0 + c]; a3[c] = A3[c0 + c];
        w0[c] = W0[(long)(c0 + c) * ldw];
        wacc[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) sacc[c][t] = 0.f;
    }
    const long qb = (long)b * colsPerCloud;
    for (long off = 4L * tid; off < colsPerCloud; off += 1024) {
        const long q = qb + off;
        const float4 s4 = *reinterpret_cast<const float4*>(&sim[q]);
        float4 ds = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int cb = 0; cb < CS; cb += 4) {
            float4 d[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[u] = *reinterpret_cast<const float4*>(&dN[(long)(c0 + cb + u) * P + q]);
                y[u] = *reinterpret_cast<const float4*>(&Y0[(long)(c0 + cb + u) * P + q]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = cb + u;
                float4 dy;
                dy.x = fmaf(a1[c], d[u].x, fmaf(a2[c], y[u].x, a3[c]));
                dy.y = fmaf(a1[c], d[u].y, fmaf(a2[c], y[u].y, a3[c]));
                dy.z = fmaf(a1[c], d[u].z, fmaf(a2[c], y[u].z, a3[c]));
                dy.w = fmaf(a1[c], d[u].w, fmaf(a2[c], y[u].w, a3[c]));
                sacc[c][0] += dy.x; sacc[c][1] += dy.y; sacc[c][2] += dy.z; sacc[c][3] += dy.w;
                ds.x = fmaf(w0[c], dy.x, ds.x); ds.y = fmaf(w0[c], dy.y, ds.y);
                ds.z = fmaf(w0[c], dy.z, ds.z); ds.w = fmaf(w0[c], dy.w, ds.w);
                wacc[c] += (dy.x * s4.x + dy.y * s4.y) + (dy.z * s4.z + dy.w * s4.w);
            }
        }
        *reinterpret_cast<float4*>(&dsim_part[(long)g * P + q]) = ds;
    }
    // S: threads with the same (4*tid) % M hold the same template points.  M <= 64 here (M divides 1024, M % 4 == 0):
    // slot = tid / (M/4) enumerates the 1024/M search points a pass covers.
    const int per = M / 4;                       // threads per ball
    const int ii = tid % per, jj = tid / per;    // jj < 1024 / M
    const int nj = 256 / per;
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        // fold jj in rounds of 16 (the LDS scratch holds 16 slots)
        for (int base = 0; base < nj; base += 16) {
            __syncthreads();
            if (jj >= base && jj < base + 16) {
#pragma unroll
                for (int t = 0; t < 4; ++t) red[jj - base][4 * ii + t] = sacc[c][t];
            }
            __syncthreads();
            if (tid < M) {
                float a = 0.f;
                const int lim = nj - base < 16 ? nj - base : 16;
                for (int k = 0; k < lim; ++k) a += red[k][tid];
                float* dst = &S[(long)(c0 + c) * lds_ + (long)b * M + tid];
                if (base == 0) *dst = a; else *dst += a;
            }
        }
    }
    // dW0[:, 0] partial of this cloud
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        float v = wacc[c];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((tid & 63) == 0) wred[c][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < CS) dw_part[(long)b * C0 + c0 + tid] = (wred[tid][0] + wred[tid][1]) + (wred[tid][2] + wred[tid][3]);
}

}  // namespace

static bool xc_shape_ok(int B, int M, int N, int C0) {
    const long cpc = (long)M * N;
    return B > 0 && M >= 4 && M <= 64 && (M & (M - 1)) == 0 && N > 0 && C0 > 0 && C0 % 16 == 0 && cpc % 1024 == 0 &&
           (B * cpc) % 256 == 0 && B * cpc <= 0x7fffffffL;
}

// Y0 (C0, P), P = B*N*M, column q = (b*N + j)*M + i:  Y0[c,q] = Z[c, b*M + i] + W0[c*ldw] * sim[q]
// Z (C0, ldz >= B*M); sim (P) = the (B, N, M) similarity map; part [P/256][2][C0] or NULL.
extern "C" int o3d_xcorr_expand(const float* Z, long ldz, const float* sim, const float* W0, int ldw, int B, int M, int N,
                                int C0, float* Y0, float* part, const float* stat_c, void* stream) {
    if (!Z || !sim || !W0 || !Y0 || ldw < 1 || !xc_shape_ok(B, M, N, C0) || ldz < (long)B * M) return O3D_EINVAL;
    const long cpc = (long)M * N, P = B * cpc;
    const int ysplit = C0 >= 128 ? 4 : C0 >= 64 ? 2 : 1;
    hipLaunchKernelGGL(xcorr_expand_kernel, dim3((unsigned)(P / 256), ysplit), dim3(256), 0, o3d_stream(stream), Z, ldz,
                       sim, W0, ldw, C0, M, cpc, P, Y0, part, stat_c);
    return o3d_launch_status();
}

// floats of `dsim_part` scratch: (C0 / 16) rows of P
extern "C" long o3d_xcorr_reduce_groups(int C0) { return C0 > 0 && C0 % 16 == 0 ? C0 / 16 : -1; }

// S (C0, lds >= B*M), dsim_part (C0/16, P), dw_part (B, C0) from dY0 = A1*dN + A2*Y0 + A3 (see the file header)
extern "C" int o3d_xcorr_reduce(const float* dN, const float* Y0, const float* A1, const float* A2, const float* A3,
                                const float* sim, const float* W0, int ldw, int B, int M, int N, int C0, float* S,
                                long lds, float* dsim_part, float* dw_part, void* stream) {
    if (!dN || !Y0 || !A1 || !A2 || !A3 || !sim || !W0 || !S || !dsim_part || !dw_part || ldw < 1 ||
        !xc_shape_ok(B, M, N, C0) || lds < (long)B * M)
        return O3D_EINVAL;
    const long cpc = (long)M * N, P = B * cpc;
    hipLaunchKernelGGL(xcorr_reduce_kernel<16>, dim3(B * (C0 / 16)), dim3(256), 0, o3d_stream(stream), dN, Y0, A1, A2, A3,
                       sim, W0, ldw, C0, M, cpc, P, S, lds, dsim_part, dw_part);
    return o3d_launch_status();
}

// =====================================================================================================================
// Cosine similarity map of P2B_XCorr (models/head/xcorr.py:37-38: nn.CosineSimilarity(dim=1, eps=1e-8) on the expanded
// (B,f,M,N) features): sim[b,j,i] = <t[b,:,i], s[b,:,j]> / (max(|t_i|, eps) * max(|s_j|, eps)).  Round 4: one kernel each
// way instead of torch.bmm + norm + clamp + outer product + division (and autograd's mirror of them): 4 rocBLAS launches and
// ~12 elementwise ones per step.  The products are small (f = 256, M = 64, N = 128 per pair: 4 MFLOP forward) -- plain
// LDS-tiled FMA, no MFMA: the work is latency, not arithmetic.
//   forward : workgroup = (pair b, 32 search points): t/s chunks of 32 channels through LDS, thread = 2 search x 4 template
//             points; the two norms ride in the same pass.
//   backward: workgroup = (pair b, 32 channels): G[j,i] = dsim/(tn_i sn_j) and the row / column sums of dsim*sim in LDS;
//             dt[c,i] = sum_j G[j,i] s[c,j] - t[c,i] a_i / tn_i^2,  ds[c,j] = sum_i G[j,i] t[c,i] - s[c,j] b_j / sn_j^2
//             (a_i = sum_j dsim*sim, b_j = sum_i dsim*sim; a clamped norm has no gradient through the norm).
// t (B,f,M), s (B,f,N): arbitrary element strides (the features are views of the flat conv_final output).
// =====================================================================================================================
namespace {
constexpr int CS_J = 32, CS_K = 32, CS_MMAX = 64, CS_NMAX = 128;
constexpr float CS_EPS = 1e-8f;

struct CosArgs {
    const float* t; long tsb, tsc, tsn;
    const float* s; long ssb, ssc, ssn;
    int B, f, M, N;
    float* sim; float* tn; float* sn;            // (B,N,M), (B,M), (B,N)
    const float* dsim; float* dt; float* ds;     // backward: (B,N,M) in, (B,f,M), (B,f,N) out (contiguous)
};

__global__ __launch_bounds__(256) void cosine_sim_fwd_kernel(CosArgs a) {
    __shared__ float T[CS_K][CS_MMAX + 4], S[CS_K][CS_J + 4];
    __shared__ float tns[CS_MMAX], sns[CS_J];
    const int b = blockIdx.x, j0 = blockIdx.y * CS_J;
    const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;          // template points 4ti..4ti+3, search 2tj, 2tj+1
    const float* tb = a.t + (long)b * a.tsb;
    const float* sb = a.s + (long)b * a.ssb;
    float acc[2][4] = {};
    float n2t[4] = {}, n2s[2] = {};
    for (int k0 = 0; k0 < a.f; k0 += CS_K) {
        for (int e = tid; e < CS_K * CS_MMAX; e += 256) {
            const int k = e / CS_MMAX, i = e - k * CS_MMAX;
            T[k][i] = (i < a.M) ? tb[(long)(k0 + k) * a.tsc + (long)i * a.tsn] : 0.f;
        }
        for (int e = tid; e < CS_K * CS_J; e += 256) {
            const int k = e / CS_J, j = e - k * CS_J;
            S[k][j] = (j0 + j < a.N) ? sb[(long)(k0 + k) * a.ssc + (long)(j0 + j) * a.ssn] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < CS_K; ++k) {
            const float4 tv = *reinterpret_cast<const float4*>(&T[k][4 * ti]);
            const float2 sv = *reinterpret_cast<const float2*>(&S[k][2 * tj]);
            const float t4[4] = {tv.x, tv.y, tv.z, tv.w}, s2[2] = {sv.x, sv.y};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fmaf(s2[u], t4[v], acc[u][v]);
            if (tj == 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) n2t[v] = fmaf(t4[v], t4[v], n2t[v]);
            }
            if (ti == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) n2s[u] = fmaf(s2[u], s2[u], n2s[u]);
            }
        }
        __syncthreads();
    }
    if (tj == 0)
        for (int v = 0; v < 4; ++v) tns[4 * ti + v] = fmaxf(sqrtf(n2t[v]), CS_EPS);
    if (ti == 0)
        for (int u = 0; u < 2; ++u) sns[2 * tj + u] = fmaxf(sqrtf(n2s[u]), CS_EPS);
    __syncthreads();
    if (blockIdx.y == 0 && tid < a.M) a.tn[(long)b * a.M + tid] = tns[tid];
    if (tid < CS_J && j0 + tid < a.N) a.sn[(long)b * a.N + j0 + tid] = sns[tid];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = j0 + 2 * tj + u;
        if (j >= a.N) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = 4 * ti + v;
            if (i < a.M) a.sim[((long)b * a.N + j) * a.M + i] = acc[u][v] / (sns[2 * tj + u] * tns[i]);
        }
    }
}

__global__ __launch_bounds__(256) void cosine_sim_bwd_kernel(CosArgs a) {
    extern __shared__ float lds[];
    const int b = blockIdx.x, c0 = blockIdx.y * 32;
    const int tid = threadIdx.x;
    const int M = a.M, N = a.N;
    float* G = lds;                       // [N][M + 1]
    float* Sc = G + N * (M + 1);          // [32][N + 1]
    float* Tc = Sc + 32 * (N + 1);        // [32][M + 1]
    float* ai = Tc + 32 * (M + 1);        // [M]  sum_j dsim*sim / tn_i^2
    float* bj = ai + M;                   // [N]  sum_i dsim*sim / sn_j^2
    const float* dsim = a.dsim + (long)b * N * M;
    const float* sim = a.sim + (long)b * N * M;
    const float* tn = a.tn + (long)b * M;
    const float* sn = a.sn + (long)b * N;
    for (int e = tid; e < N * M; e += 256) {
        const int j = e / M, i = e - j * M;
        G[j * (M + 1) + i] = dsim[e] * sim[e];              // first the products for the two sums ...
    }
    const float* tb = a.t + (long)b * a.tsb;
    const float* sb = a.s + (long)b * a.ssb;
    for (int e = tid; e < 32 * N; e += 256) {
        const int c = e / N, j = e - c * N;
        Sc[c * (N + 1) + j] = sb[(long)(c0 + c) * a.ssc + (long)j * a.ssn];
    }
    for (int e = tid; e < 32 * M; e += 256) {
        const int c = e / M, i = e - c * M;
        Tc[c * (M + 1) + i] = tb[(long)(c0 + c) * a.tsc + (long)i * a.tsn];
    }
    __syncthreads();
    if (tid < M) {                       // a clamped norm (|x| <= eps) is a constant: no gradient through it
        float s = 0.f;
        for (int j = 0; j < N; ++j) s += G[j * (M + 1) + tid];
        const float n = tn[tid];
        ai[tid] = n > CS_EPS ? s / (n * n) : 0.f;
    }
    if (tid >= 64 && tid - 64 < N) {
        const int j = tid - 64;
        float s = 0.f;
        for (int i = 0; i < M; ++i) s += G[j * (M + 1) + i];
        const float n = sn[j];
        bj[j] = n > CS_EPS ? s / (n * n) : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < N * M; e += 256) {                // ... then G itself
        const int j = e / M, i = e - j * M;
        G[j * (M + 1) + i] = dsim[e] / (tn[i] * sn[j]);
    }
    __syncthreads();
    // dt[c, i]: 32 x M outputs, thread = (channel c = tid / 8, template points i = tid % 8 + 8 v)
    {
        const int c = tid >> 3, i8 = tid & 7;
        float acc[CS_MMAX / 8] = {};
        for (int j = 0; j < N; ++j) {
            const float sv = Sc[c * (N + 1) + j];
#pragma unroll
            for (int v = 0; v < CS_MMAX / 8; ++v)
                if (i8 + 8 * v < M) acc[v] = fmaf(G[j * (M + 1) + i8 + 8 * v], sv, acc[v]);
        }
#pragma unroll
        for (int v = 0; v < CS_MMAX / 8; ++v) {
            const int i = i8 + 8 * v;
            if (i < M) a.dt[((long)b * a.f + c0 + c) * M + i] = acc[v] - Tc[c * (M + 1) + i] * ai[i];
        }
    }
    // ds[c, j]: 32 x N outputs, thread = (channel c = tid / 8, search points j = tid % 8 + 8 v)
    {
        const int c = tid >> 3, j8 = tid & 7;
        float acc[CS_NMAX / 8] = {};
        for (int i = 0; i < M; ++i) {
            const float tv = Tc[c * (M + 1) + i];
#pragma unroll
            for (int v = 0; v < CS_NMAX / 8; ++v)
                if (j8 + 8 * v < N) acc[v] = fmaf(G[(j8 + 8 * v) * (M + 1) + i], tv, acc[v]);
        }
#pragma unroll
        for (int v = 0; v < CS_NMAX / 8; ++v) {
            const int j = j8 + 8 * v;
            if (j < N) a.ds[((long)b * a.f + c0 + c) * N + j] = acc[v] - Sc[c * (N + 1) + j] * bj[j];
        }
    }
}
}  // namespace

extern "C" int o3d_cosine_sim_fwd(const float* t, long tsb, long tsc, long tsn, const float* s, long ssb, long ssc, long ssn,
                                  int B, int f, int M, int N, float* sim, float* tn, float* sn, void* stream) {
    if (!t || !s || !sim || !tn || !sn || B <= 0 || B > 65535 || f <= 0 || f % CS_K != 0 || M <= 0 || M > CS_MMAX || M % 4 != 0 ||
        N <= 0 || N > CS_NMAX)
        return O3D_EINVAL;
    CosArgs a = {t, tsb, tsc, tsn, s, ssb, ssc, ssn, B, f, M, N, sim, tn, sn, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(cosine_sim_fwd_kernel, dim3(B, o3d_cdiv(N, CS_J)), dim3(256), 0, o3d_stream(stream), a);
    return o3d_launch_status();
}

extern "C" int o3d_cosine_sim_bwd(const float* dsim, const float* sim, const float* tn, const float* sn, const float* t,
                                  long tsb, long tsc, long tsn, const float* s, long ssb, long ssc, long ssn, int B, int f, int M,
                                  int N, float* dt, float* ds, void* stream) {
    if (!dsim || !sim || !tn || !sn || !t || !s || !dt || !ds || B <= 0 || B > 65535 || f <= 0 || f % 32 != 0 || M <= 0 ||
        M > CS_MMAX || N <= 0 || N > CS_NMAX)
        return O3D_EINVAL;
    CosArgs a = {t, tsb, tsc, tsn, s, ssb, ssc, ssn, B, f, M, N, const_cast<float*>(sim), const_cast<float*>(tn),
                 const_cast<float*>(sn), dsim, dt, ds};
    const size_t lds = sizeof(float) * ((size_t)N * (M + 1) + 32 * (N + 1) + 32 * (M + 1) + M + N);
    const void* fn = reinterpret_cast<const void*>(cosine_sim_bwd_kernel);
    if (lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return O3D_ELAUNCH;
    hipLaunchKernelGGL(cosine_sim_bwd_kernel, dim3(B, f / 32), dim3(256), lds, o3d_stream(stream), a);
    return o3d_launch_status();
}
