// pointwise.hip -- ends of a per-point MLP chain on the flat (C, P) layout, P = B*N columns.
//
// The M2-Track stacks (models/backbone/pointnet.py:91-204: Conv1d 1x1 -> BatchNorm1d -> ReLU, global
// max-pool) are the grouped MLP with ONE ball per cloud, so they run on the GEMM kernels of
// mlp_direct.hip / mlp_wgrad.hip unchanged (BN+ReLU applied by the consumer on load).  What is left is
// the two ways a chain ends:
//   * materialised activation  out = relu(Y*scale+shift)            (bn_relu_apply / act_bwd_partials)
//   * global max over the N points of a cloud (AdaptiveMaxPool1d(1)) (gmax_fwd; backward = the pooled
//     kernels of mlp.hip / compact.hip with one ball per cloud)
#include "o3d_common.hpp"

namespace {

__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float* __restrict__ Y,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, long P,
                                                            float* __restrict__ out) {
    const int c = blockIdx.y;
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= P) return;
    const float sc = scale[c], sf = shift[c];
    const float4 y = *reinterpret_cast<const float4*>(&Y[(long)c * P + p]);
    float4 o;
    o.x = fmaxf(fmaf(y.x, sc, sf), 0.f); o.y = fmaxf(fmaf(y.y, sc, sf), 0.f);
    o.z = fmaxf(fmaf(y.z, sc, sf), 0.f); o.w = fmaxf(fmaf(y.w, sc, sf), 0.f);
    *reinterpret_cast<float4*>(&out[(long)c * P + p]) = o;
}

// dN = g where the activation is positive, plus the BatchNorm-backward partials of that layer per
// 128-column tile: part[tile][0][c] = sum dN, part[tile][1][c] = sum dN*(Y-mean).
// block = 8 channels x 32 lanes (float4 each).
__global__ __launch_bounds__(256) void act_bwd_partials_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ Y,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean, int C, long P,
                                                               float* __restrict__ dN, float* __restrict__ part) {
    const int lane = threadIdx.x & 31, c = blockIdx.y * 8 + (threadIdx.x >> 5);
    const long tile = blockIdx.x, p = tile * 128 + 4 * lane;
    const bool live = c < C;
    const int cc = live ? c : C - 1;
    const float sc = scale[cc], sf = shift[cc], mu = mean[cc];
    const float4 y = *reinterpret_cast<const float4*>(&Y[(long)cc * P + p]);
    float4 v = *reinterpret_cast<const float4*>(&g[(long)cc * P + p]);
    v.x = fmaf(y.x, sc, sf) > 0.f ? v.x : 0.f; v.y = fmaf(y.y, sc, sf) > 0.f ? v.y : 0.f;
    v.z = fmaf(y.z, sc, sf) > 0.f ? v.z : 0.f; v.w = fmaf(y.w, sc, sf) > 0.f ? v.w : 0.f;
    if (live) *reinterpret_cast<float4*>(&dN[(long)c * P + p]) = v;
    float s = (v.x + v.y) + (v.z + v.w);
    float q = v.x * (y.x - mu) + v.y * (y.y - mu) + v.z * (y.z - mu) + v.w * (y.w - mu);
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) { s += __shfl_xor(s, m, 64); q += __shfl_xor(q, m, 64); }
    if (live && lane == 0) {
        part[(tile * 2 + 0) * C + c] = s;
        part[(tile * 2 + 1) * C + c] = q;
    }
}

// out[b,c] = max over the cloud's N columns of relu(Y*scale+shift); argq = column of the (first) maximum,
// yarg = raw Y there.  block = 4 channels x 64 lanes.
__global__ __launch_bounds__(256) void gmax_fwd_kernel(const float* __restrict__ Y,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int C, int N, long P,
                                                       float* __restrict__ out, int32_t* __restrict__ argq,
                                                       float* __restrict__ yarg) {
    const int lane = threadIdx.x & 63, b = blockIdx.x;
    int c = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool live = c < C;
    if (!live) c = C - 1;
    const float sc = scale[c], sf = shift[c];
    const float* y = Y + (long)c * P + (long)b * N;
    float best = -INFINITY, yb = 0.f;
    int bq = 0x7fffffff;
    for (int i = 4 * lane; i < N; i += 256) {
        const float4 v4 = *reinterpret_cast<const float4*>(&y[i]);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float n = fmaf(v[t], sc, sf);
            if (n > best) { best = n; bq = i + t; yb = v[t]; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64), oy = __shfl_xor(yb, off, 64);
        const int oq = __shfl_xor(bq, off, 64);
        if (ob > best || (ob == best && oq < bq)) { best = ob; bq = oq; yb = oy; }
    }
    if (live && lane == 0) {
        const long o = (long)b * C + c;                 // (B, C): the pooled layout of compact.hip with npoint = 1
        out[o] = fmaxf(best, 0.f);
        if (argq) { argq[o] = b * N + bq; yarg[o] = yb; }
    }
}

// per-cloud sums of dY = A1*dN + A2*Y + A3 (the gradient of a per-cloud bias): out[c, b] = sum over the N columns of
// cloud b.  One workgroup per (channel, cloud), fixed order.
__global__ __launch_bounds__(256) void cloud_sum_dy_kernel(const float* __restrict__ dN, const float* __restrict__ Y,
                                                           const float* __restrict__ A1, const float* __restrict__ A2,
                                                           const float* __restrict__ A3, int N, long P, int B,
                                                           float* __restrict__ out) {
    __shared__ float sh[4];
    const int c = blockIdx.y, b = blockIdx.x;
    const float a1 = A1[c], a2 = A2[c], a3 = A3[c];
    const float* d = dN + (long)c * P + (long)b * N;
    const float* y = Y + (long)c * P + (long)b * N;
    float s = 0.f;
    for (int i = 4 * threadIdx.x; i < N; i += 1024) {
        const float4 dv = *reinterpret_cast<const float4*>(&d[i]);
        const float4 yv = *reinterpret_cast<const float4*>(&y[i]);
        s += (fmaf(a1, dv.x, fmaf(a2, yv.x, a3)) + fmaf(a1, dv.y, fmaf(a2, yv.y, a3))) +
             (fmaf(a1, dv.z, fmaf(a2, yv.z, a3)) + fmaf(a1, dv.w, fmaf(a2, yv.w, a3)));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(long)c * B + b] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

}  // namespace

// out (C, B) = per-cloud column sums of dY = A1*dN + A2*Y + A3 over the flat (C, B*N) layout; N % 4 == 0
extern "C" int o3d_cloud_sum_dy(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3, int C,
                                int B, int N, float* out, void* stream) {
    if (!dN || !Y || !A1 || !A2 || !A3 || !out || C <= 0 || B <= 0 || N <= 0 || N % 4 != 0) return O3D_EINVAL;
    hipLaunchKernelGGL(cloud_sum_dy_kernel, dim3(B, C), dim3(256), 0, o3d_stream(stream), dN, Y, A1, A2, A3, N,
                       (long)B * N, B, out);
    return o3d_launch_status();
}

extern "C" int o3d_bn_relu_apply(const float* Y, const float* scale, const float* shift, int C, long P, float* out,
                                 void* stream) {
    if (!Y || !scale || !shift || !out || C <= 0 || P <= 0 || P % 4 != 0) return O3D_EINVAL;
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(o3d_cdiv(P / 4, 256), C), dim3(256), 0, o3d_stream(stream), Y, scale,
                       shift, P, out);
    return o3d_launch_status();
}

extern "C" int o3d_act_bwd_partials(const float* g, const float* Y, const float* scale, const float* shift,
                                    const float* mean, int C, long P, float* dN, float* part, void* stream) {
    if (!g || !Y || !scale || !shift || !mean || !dN || !part || C <= 0 || P <= 0 || P % 128 != 0) return O3D_EINVAL;
    hipLaunchKernelGGL(act_bwd_partials_kernel, dim3((unsigned)(P / 128), o3d_cdiv(C, 8)), dim3(256), 0,
                       o3d_stream(stream), g, Y, scale, shift, mean, C, P, dN, part);
    return o3d_launch_status();
}

extern "C" int o3d_gmax_fwd(const float* Y, const float* scale, const float* shift, int B, int C, int N, float* out,
                            int32_t* argq, float* yarg, void* stream) {
    if (!Y || !scale || !shift || !out || B <= 0 || C <= 0 || N <= 0 || N % 4 != 0 || (argq && !yarg)) return O3D_EINVAL;
    hipLaunchKernelGGL(gmax_fwd_kernel, dim3(B, o3d_cdiv(C, 4)), dim3(256), 0, o3d_stream(stream), Y, scale, shift, C, N,
                       (long)B * N, out, argq, yarg);
    return o3d_launch_status();
}

// ---- backward of a THIN layer 0 (round 4) ---------------------------------------------------------------------------------
// The per-point stacks of M2-Track start from 12-14 input channels (xyz + time flag + BoxCloud, models/m2track.py:45-56):
// layer 0 is 64 x <= 16 x P.  Its weight gradient dW (64, Cin) = dY . X^T and the stack's input gradient dX (Cin, P) =
// W^T . dY went through the LDS-staged MFMA kernels of mlp.hip built for wide layers: 58 + 25 us on 98 304 columns where the
// 55 MB they read take 7 us (profiles/r04_per_launch_roofline_m2track.txt, pw_conv_wgrad / pw_conv_dgrad with Cin 12-14).
// One pass over dN and Y instead, on the vector ALUs (2 x 16 FMAs per element of dY: 0.2 GFLOP in all): a workgroup of 16
// waves owns 64 columns at a time, wave w the rows [4w, 4w + 4) of dY = A1*dN + A2*Y + A3; lane = column.  Each lane keeps
// 4 x 16 sums of dW, reduced over the lanes once at the end (recursive halving: 63 shuffles) into one partial per workgroup;
// the dX sums of the 16 waves meet in LDS.  A second small launch adds the partials in a fixed order.
namespace {

constexpr int TH_K = 64, TH_M = 16, TH_WV = 16, TH_KW = TH_K / TH_WV;
constexpr int TH_GRID = 256;

struct ThinBwd {
    const float* dN; const float* Y; const float* A1; const float* A2; const float* A3;   // (64, P), per-row constants
    const float* X;      // (Cin, P)
    const float* W;      // (64, Cin)
    int Cin; long P;
    float* wpart;        // [gridDim.x][64][16]
    float* dX;           // (Cin, P) or unused
};

template <bool DX>
__global__ __launch_bounds__(TH_WV * 64) void thin_bwd_kernel(ThinBwd a) {
    __shared__ float xs[2][TH_M][64];
    __shared__ float red[DX ? TH_WV : 1][TH_M][64];
    __shared__ float Ws[TH_K][TH_M];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (DX) {
        const int k = tid >> 4, m = tid & 15;
        Ws[k][m] = m < a.Cin ? a.W[k * a.Cin + m] : 0.f;
    }
    float c1[TH_KW], c2[TH_KW], c3[TH_KW];
#pragma unroll
    for (int j = 0; j < TH_KW; ++j) {
        const int k = wave * TH_KW + j;
        c1[j] = a.A1[k]; c2[j] = a.A2[k]; c3[j] = a.A3[k];
    }
    float acc[TH_KW * TH_M];
#pragma unroll
    for (int i = 0; i < TH_KW * TH_M; ++i) acc[i] = 0.f;
    const long ngroups = a.P / 64;
    int buf = 0;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x, buf ^= 1) {
        const long p = g * 64 + lane;
        {   // the group's inputs: one value per thread (row = wave)
            xs[buf][wave][lane] = wave < a.Cin ? a.X[(long)wave * a.P + p] : 0.f;
        }
        float dy[TH_KW];
#pragma unroll
        for (int j = 0; j < TH_KW; ++j) {
            const long o = (long)(wave * TH_KW + j) * a.P + p;
            dy[j] = fmaf(c1[j], a.dN[o], fmaf(c2[j], a.Y[o], c3[j]));
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < TH_M; ++m) {
            const float x = xs[buf][m][lane];
#pragma unroll
            for (int j = 0; j < TH_KW; ++j) acc[j * TH_M + m] = fmaf(dy[j], x, acc[j * TH_M + m]);
            if (DX) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < TH_KW; ++j) d = fmaf(Ws[wave * TH_KW + j][m], dy[j], d);
                red[wave][m][lane] = d;
            }
        }
        if (DX) {
            __syncthreads();
            if (wave < a.Cin) {          // thread (row = wave, column = lane) of the 16 x 64 outputs
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < TH_WV; ++w) s += red[w][wave][lane];
                a.dX[(long)wave * a.P + p] = s;
            }
        }
    }
    // sums over the 64 lanes: recursive halving, afterwards lane l holds the total of acc index l (= row l / 16, input l % 16)
#pragma unroll
    for (int half = 32; half >= 1; half >>= 1) {
        const bool up = (lane & half) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const float keep = up ? acc[i + half] : acc[i];
            const float send = up ? acc[i] : acc[i + half];
            acc[i] = keep + __shfl_xor(send, half, 64);
        }
    }
    a.wpart[((long)blockIdx.x * TH_K + wave * TH_KW + (lane >> 4)) * TH_M + (lane & 15)] = acc[0];
}

// dW (64, Cin) = sum over the workgroups' partials, fixed order: 64 threads per element, 4 elements per workgroup
__global__ __launch_bounds__(256) void thin_reduce_kernel(const float* __restrict__ wpart, int nparts, int Cin,
                                                          float* __restrict__ dW) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;     // e over 64 x 16
    float s = 0.f;
    for (int g = lane; g < nparts; g += 64) s += wpart[(long)g * (TH_K * TH_M) + e];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const int k = e >> 4, m = e & 15;
    if (lane == 0 && m < Cin) dW[k * Cin + m] = s;
}

}  // namespace

extern "C" long o3d_thin_bwd_scratch(void) { return (long)TH_GRID * TH_K * TH_M; }

// Layer 0 of a per-point stack with Cout == 64 and Cin <= 16 on the flat (C, P) layout, P % 64 == 0:
// dY = A1*dN + A2*Y + A3 (the BatchNorm backward folded, as o3d_mlp_conv_wgrad);  dW (64, Cin) = dY . X^T;
// dX (Cin, P) = W^T . dY when dX != NULL.  scratch: o3d_thin_bwd_scratch() floats.
extern "C" int o3d_thin_bwd(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3, const float* X,
                            const float* W, int Cin, int Cout, long P, float* scratch, float* dW, float* dX, void* stream) {
    if (!dN || !Y || !A1 || !A2 || !A3 || !X || !W || !scratch || !dW || Cout != TH_K || Cin <= 0 || Cin > TH_M || P <= 0 ||
        P % 64 != 0)
        return O3D_EINVAL;
    ThinBwd a{dN, Y, A1, A2, A3, X, W, Cin, P, scratch, dX};
    const long ngroups = P / 64;
    const int grid = (int)(ngroups < TH_GRID ? ngroups : TH_GRID);
    if (dX) hipLaunchKernelGGL(thin_bwd_kernel<true>, dim3(grid), dim3(TH_WV * 64), 0, o3d_stream(stream), a);
    else hipLaunchKernelGGL(thin_bwd_kernel<false>, dim3(grid), dim3(TH_WV * 64), 0, o3d_stream(stream), a);
    hipLaunchKernelGGL(thin_reduce_kernel, dim3(TH_K * TH_M / 4), dim3(256), 0, o3d_stream(stream), scratch, grid, Cin, dW);
    return o3d_launch_status();
}

// ---- backward of the global max without the dense gradient (round 4) ---------------------------------------------------------
// dN of a "gmax" stack's last layer is zero except at one column per (cloud, channel): materialising it cost a zero fill of
// (C, P) -- 400 MB for M2-Track's 1024-channel layer -- a scatter, and a dense read in each of the two GEMMs behind it.  The
// GEMM kernels can build that operand on the fly from pk (C, B) = {dOut where out > 0, bits(arg - first column of the cloud)}
// (the pooled-source modes of mlp_direct.hip / mlp_wgrad.hip: one "ball" of N columns per cloud): this kernel writes pk and
// the BatchNorm-backward sums {sum g, sum g * (yarg - mean)} (one partial row), thread per channel, fixed order.
namespace {
__global__ __launch_bounds__(64) void gmax_bwd_pk_kernel(const float* __restrict__ dOut, const float* __restrict__ out,
                                                         const int32_t* __restrict__ argq, const float* __restrict__ yarg,
                                                         const float* __restrict__ mean, int B, int C, int N,
                                                         float2* __restrict__ pk, float* __restrict__ part) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const float mu = mean[c];
    float s = 0.f, q = 0.f;
    for (int b = 0; b < B; ++b) {
        const long o = (long)b * C + c;
        const float g = out[o] > 0.f ? dOut[o] : 0.f;
        pk[(long)c * B + b] = make_float2(g, __int_as_float(argq[o] - b * N));
        s += g;
        q = fmaf(g, yarg[o] - mu, q);
    }
    part[c] = s;
    part[C + c] = q;
}
}  // namespace

// dOut, out, yarg (B, C), argq (B, C) as o3d_gmax_fwd wrote them -> pk (C, B, 2), part [1][2][C]
extern "C" int o3d_gmax_bwd_pk(const float* dOut, const float* out, const int32_t* argq, const float* yarg, const float* mean,
                               int B, int C, int N, float* pk, float* part, void* stream) {
    if (!dOut || !out || !argq || !yarg || !mean || !pk || !part || B <= 0 || C <= 0 || N <= 0) return O3D_EINVAL;
    hipLaunchKernelGGL(gmax_bwd_pk_kernel, dim3(o3d_cdiv(C, 64)), dim3(64), 0, o3d_stream(stream), dOut, out, argq, yarg, mean,
                       B, C, N, reinterpret_cast<float2*>(pk), part);
    return o3d_launch_status();
}
