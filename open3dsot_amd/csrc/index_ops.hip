// index_ops.hip -- ball query, group / gather (+grad), 3-NN, 3-interpolate (+grad), stable kNN
// for gfx950.  HBM/latency-bound integer+gather work: coalesced along the contiguous output
// axis, wave64 ballot compaction for the ball query, hardware fp32 atomics for the scatters.
//
// Reference interfaces replaced (pointnet2/utils/pointnet2_utils.py):
//   :268 _ext.ball_query        :217/:237 _ext.group_points[_grad]
//   :92/:98 _ext.gather_points[_grad]   :125 _ext.three_nn
//   :162/:184 _ext.three_interpolate[_grad]
//   models/head/xcorr.py:81,87 + pointnet2_utils.py:399-400  cdist+argsort -> o3d_knn
// Semantics restated in oracle/pointnet2_oracle.c.
#include "o3d_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// Ball query: one wave per centre.  The wave sweeps the cloud 64 points at a time in index
// order; a ballot gives the in-radius mask, a prefix popcount gives each hit its output
// slot, so the "first nsample hits in ascending index order" rule falls out without any
// serial per-thread scan.  The sweep stops as soon as nsample hits are found.
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ new_xyz,
                                                         const float* __restrict__ xyz, long total,
                                                         int N, int npoint, float r2, int nsample,
                                                         int32_t* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (long c = (long)blockIdx.x * 4 + wave; c < total; c += (long)gridDim.x * 4) {
        const long b = c / npoint;
        const float* p = xyz + b * N * 3;
        const float cx = new_xyz[c * 3 + 0], cy = new_xyz[c * 3 + 1], cz = new_xyz[c * 3 + 2];
        int32_t* out = idx + c * nsample;
        int cnt = 0, first = 0;
        for (int base = 0; base < N && cnt < nsample; base += 64) {
            const int k = base + lane;
            bool hit = false;
            if (k < N) {
                const float d2 = o3d_sqdist3(cx, cy, cz, p[3 * k], p[3 * k + 1], p[3 * k + 2]);
                hit = d2 < r2;
            }
            const unsigned long long m = __ballot(hit);
            if (m) {
                if (cnt == 0) first = base + (__ffsll((long long)m) - 1);
                const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                if (hit && pos < nsample) out[pos] = k;
                cnt += __popcll(m);
            }
        }
        if (cnt > nsample) cnt = nsample;
        for (int l = cnt + lane; l < nsample; l += 64) out[l] = first;  // pad (0 when empty)
    }
}

// ---------------------------------------------------------------------------------------
// Sampling gather + ball query of one or two sets of clouds in ONE launch (round 5).  A set-abstraction level starts with
// new_xyz = xyz[sample_idx] (farthest-point indices, or the arange(npoint) prefix: pointnet2_modules.py:52-62) followed by
// ball_query(new_xyz, xyz) (:64, pointnet2_utils.py:268), for the template and for the search cloud: 2 gathers (or strided
// copies) + 2 ball queries + the concatenation of the centres for the fused path = 5 launches per level.  Here a wave takes
// a centre of either set: fetches it through the sampling index, writes it to `centers` ((nballs + 1, 3): set 0's centres,
// set 1's, then the origin row of the padding columns' dummy ball -- new_xyz of both sets are views of it) and runs the sweep of
// ball_query_kernel on it (same arithmetic, same order: bit-identical indices).
struct SampleQuerySet { const float* xyz; const int32_t* sidx; int N, npoint; int32_t* idx; };   // sidx NULL: prefix

__global__ __launch_bounds__(256) void sample_query_kernel(SampleQuerySet s0, SampleQuerySet s1, long total0, long total,
                                                           float r2, int nsample, float* __restrict__ centers) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x < 3) centers[total * 3 + threadIdx.x] = 0.f;      // the dummy ball's centre
    for (long c = (long)blockIdx.x * 4 + wave; c < total; c += (long)gridDim.x * 4) {
        const bool second = c >= total0;
        const SampleQuerySet& st = second ? s1 : s0;
        const long cl = second ? c - total0 : c;
        const int N = st.N;
        const long b = cl / st.npoint;
        const int j = (int)(cl - b * st.npoint);
        const float* p = st.xyz + b * N * 3;
        const int src = st.sidx ? st.sidx[cl] : j;
        const float cx = p[3 * src + 0], cy = p[3 * src + 1], cz = p[3 * src + 2];
        if (lane < 3) centers[c * 3 + lane] = lane == 0 ? cx : lane == 1 ? cy : cz;
        int32_t* out = st.idx + cl * nsample;
        int cnt = 0, first = 0;
        for (int base = 0; base < N && cnt < nsample; base += 64) {
            const int k = base + lane;
            bool hit = false;
            if (k < N) {
                const float d2 = o3d_sqdist3(cx, cy, cz, p[3 * k], p[3 * k + 1], p[3 * k + 2]);
                hit = d2 < r2;
            }
            const unsigned long long m = __ballot(hit);
            if (m) {
                if (cnt == 0) first = base + (__ffsll((long long)m) - 1);
                const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                if (hit && pos < nsample) out[pos] = k;
                cnt += __popcll(m);
            }
        }
        if (cnt > nsample) cnt = nsample;
        for (int l = cnt + lane; l < nsample; l += 64) out[l] = first;  // pad (0 when empty)
    }
}

// ---------------------------------------------------------------------------------------
// group / gather: out[b,c,q] = feats[b,c,idx[b,q]], q over npoint*nsample (contiguous writes)
constexpr int GCT = 16;  // channels per workgroup
__global__ __launch_bounds__(256) void group_points_kernel(const float* __restrict__ feats,
                                                           const int32_t* __restrict__ idx, int C,
                                                           int N, int Q, float* __restrict__ out) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    const long b = blockIdx.z;
    const int c0 = blockIdx.y * GCT;
    const int c1 = c0 + GCT < C ? c0 + GCT : C;
    const int id = idx[b * Q + q];
    const float* src = feats + (b * C + c0) * (long)N + id;
    float* dst = out + (b * C + c0) * (long)Q + q;
    for (int c = c0; c < c1; ++c, src += N, dst += Q) *dst = *src;
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(const float* __restrict__ grad_out,
                                                                const int32_t* __restrict__ idx,
                                                                int C, int N, int Q,
                                                                float* __restrict__ grad_feats) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    const long b = blockIdx.z;
    const int c0 = blockIdx.y * GCT;
    const int c1 = c0 + GCT < C ? c0 + GCT : C;
    const int id = idx[b * Q + q];
    float* dst = grad_feats + (b * C + c0) * (long)N + id;
    const float* src = grad_out + (b * C + c0) * (long)Q + q;
    for (int c = c0; c < c1; ++c, src += Q, dst += N) unsafeAtomicAdd(dst, *src);
}

// ---------------------------------------------------------------------------------------
// 3-NN: one thread per unknown point, known points streamed through L1 (broadcast reads).
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ unknown,
                                                       const float* __restrict__ known, int n,
                                                       int m, float* __restrict__ dist2,
                                                       int32_t* __restrict__ idx) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const long b = blockIdx.y;
    const float* u = unknown + (b * n + j) * 3;
    const float* kn = known + b * m * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    double best1 = 1e40, best2 = 1e40, best3 = 1e40;  // upstream literal is a double
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < m; ++k) {
        const double d = (double)o3d_sqdist3(ux, uy, uz, kn[3 * k], kn[3 * k + 1], kn[3 * k + 2]);
        if (d < best1) {
            best3 = best2; i3 = i2; best2 = best1; i2 = i1; best1 = d; i1 = k;
        } else if (d < best2) {
            best3 = best2; i3 = i2; best2 = d; i2 = k;
        } else if (d < best3) {
            best3 = d; i3 = k;
        }
    }
    float* dd = dist2 + (b * n + j) * 3;
    int32_t* ii = idx + (b * n + j) * 3;
    dd[0] = (float)best1; dd[1] = (float)best2; dd[2] = (float)best3;
    ii[0] = i1; ii[1] = i2; ii[2] = i3;
}

__global__ __launch_bounds__(256) void three_interpolate_kernel(const float* __restrict__ feats,
                                                                const int32_t* __restrict__ idx,
                                                                const float* __restrict__ weight,
                                                                int c, int m, int n,
                                                                float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const long b = blockIdx.z;
    const int c0 = blockIdx.y * GCT;
    const int c1 = c0 + GCT < c ? c0 + GCT : c;
    const int32_t* ii = idx + (b * n + j) * 3;
    const float* w = weight + (b * n + j) * 3;
    const int a0 = ii[0], a1 = ii[1], a2 = ii[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    for (int ch = c0; ch < c1; ++ch) {
        const float* src = feats + (b * c + ch) * (long)m;
        out[(b * c + ch) * (long)n + j] =
            __fmaf_rn(src[a2], w2, __fmaf_rn(src[a1], w1, __fmul_rn(src[a0], w0)));
    }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    const float* __restrict__ grad_out, const int32_t* __restrict__ idx,
    const float* __restrict__ weight, int c, int n, int m, float* __restrict__ grad_feats) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const long b = blockIdx.z;
    const int c0 = blockIdx.y * GCT;
    const int c1 = c0 + GCT < c ? c0 + GCT : c;
    const int32_t* ii = idx + (b * n + j) * 3;
    const float* w = weight + (b * n + j) * 3;
    const int a0 = ii[0], a1 = ii[1], a2 = ii[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    for (int ch = c0; ch < c1; ++ch) {
        const float g = grad_out[(b * c + ch) * (long)n + j];
        float* dst = grad_feats + (b * c + ch) * (long)m;
        unsafeAtomicAdd(dst + a0, __fmul_rn(g, w0));
        unsafeAtomicAdd(dst + a1, __fmul_rn(g, w1));
        unsafeAtomicAdd(dst + a2, __fmul_rn(g, w2));
    }
}

// ---------------------------------------------------------------------------------------
// Stable k-smallest: one thread per query, k best kept in registers (fully unrolled
// insertion; strict '<' keeps the lower reference index in front on ties).
// DT > 0: the feature dimension at compile time (9 = the BoxCloud descriptor of BoxAwareXCorr, models/head/xcorr.py:81)
// -- the query in registers, the cloud's references staged once in LDS (uniform broadcast reads), the distance chain
// unrolled; the generic form (DT = 0) re-reads both operands from global memory inside a serial 64 x 9 load -> fma chain
// (58 us for 48 x 128 queries).  Same summation order over the dimension in both: same bits.
template <int KMAX, int DT>
__global__ __launch_bounds__(128) void knn_kernel(const float* __restrict__ query,
                                                  const float* __restrict__ ref, int Q, int R,
                                                  int D, int k, int32_t* __restrict__ idx) {
    extern __shared__ float s_ref[];           // DT > 0: R * DT floats
    const int q = blockIdx.x * 128 + threadIdx.x;
    const long b = blockIdx.y;
    const float* rf = ref + b * R * (long)D;
    if (DT > 0) {
        for (int i = threadIdx.x; i < R * DT; i += 128) s_ref[i] = rf[i];
        __syncthreads();
    }
    if (q >= Q) return;
    const float* qq = query + (b * Q + q) * (long)D;
    float qv[DT > 0 ? DT : 1];
    if (DT > 0) {
#pragma unroll
        for (int t = 0; t < DT; ++t) qv[t] = qq[t];
    }
    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { bd[t] = 0.f; bi[t] = 0; }
    int cnt = 0;
    for (int r = 0; r < R; ++r) {
        float d = 0.f;
        if (DT > 0) {
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const float df = qv[t] - s_ref[r * DT + t];
                d = __fmaf_rn(df, df, d);
            }
        } else {
            for (int t = 0; t < D; ++t) {
                const float df = qq[t] - rf[(long)r * D + t];
                d = __fmaf_rn(df, df, d);
            }
        }
        int cur;
        if (cnt < k) {
            cur = cnt++;
        } else {
            bool lt = false;
#pragma unroll
            for (int t = 0; t < KMAX; ++t) if (t == k - 1) lt = d < bd[t];
            if (!lt) continue;
            cur = k - 1;
        }
#pragma unroll
        for (int t = 0; t < KMAX; ++t) if (t == cur) { bd[t] = d; bi[t] = r; }
#pragma unroll
        for (int t = KMAX - 1; t >= 1; --t) {
            if (t <= cur && bd[t] < bd[t - 1]) {
                const float fd = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = fd;
                const int fi = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = fi;
            }
        }
    }
    int32_t* out = idx + (b * Q + q) * (long)k;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) if (t < k) out[t] = bi[t];
}

inline bool bad(const void* p, long count) { return count > 0 && p == nullptr; }

}  // namespace

extern "C" int o3d_ball_query(const float* new_xyz, const float* xyz, int B, int N, int npoint,
                              float radius, int nsample, int32_t* idx, void* stream) {
    if (B < 0 || N < 0 || npoint < 0 || nsample < 0) return O3D_EINVAL;
    const long total = (long)B * npoint;
    if (total == 0 || nsample == 0) return O3D_OK;
    if (!new_xyz || !idx || bad(xyz, N)) return O3D_EINVAL;
    const float r2 = radius * radius;
    long blocks = (total + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(ball_query_kernel, dim3((unsigned)blocks), dim3(256), 0, o3d_stream(stream),
                       new_xyz, xyz, total, N, npoint, r2, nsample, idx);
    return o3d_launch_status();
}

extern "C" int o3d_sample_query(const float* xyz0, const int32_t* sidx0, int N0, int npoint0, int32_t* idx0,
                                const float* xyz1, const int32_t* sidx1, int N1, int npoint1, int32_t* idx1, int B,
                                float radius, int nsample, float* centers, void* stream) {
    if (B <= 0 || N0 <= 0 || npoint0 <= 0 || nsample <= 0 || !xyz0 || !idx0 || !centers || npoint1 < 0 ||
        (npoint1 > 0 && (!xyz1 || !idx1 || N1 <= 0)) || (!sidx0 && npoint0 > N0) || (npoint1 > 0 && !sidx1 && npoint1 > N1))
        return O3D_EINVAL;
    const long total0 = (long)B * npoint0, total = total0 + (long)B * npoint1;
    SampleQuerySet s0 = {xyz0, sidx0, N0, npoint0, idx0}, s1 = {xyz1, sidx1, N1, npoint1 > 0 ? npoint1 : 1, idx1};
    long blocks = (total + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(sample_query_kernel, dim3((unsigned)blocks), dim3(256), 0, o3d_stream(stream), s0, s1, total0, total,
                       radius * radius, nsample, centers);
    return o3d_launch_status();
}

extern "C" int o3d_group_points(const float* feats, const int32_t* idx, int B, int C, int N,
                                int npoint, int nsample, float* out, void* stream) {
    if (B < 0 || C < 0 || N < 0 || npoint < 0 || nsample < 0 || B > 65535) return O3D_EINVAL;
    const long Q = (long)npoint * nsample;
    if (B == 0 || C == 0 || Q == 0) return O3D_OK;
    if (!feats || !idx || !out || Q > 0x7fffffffL) return O3D_EINVAL;
    hipLaunchKernelGGL(group_points_kernel, dim3(o3d_cdiv(Q, 256), o3d_cdiv(C, GCT), B), dim3(256), 0,
                       o3d_stream(stream), feats, idx, C, N, (int)Q, out);
    return o3d_launch_status();
}

extern "C" int o3d_group_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N,
                                     int npoint, int nsample, float* grad_feats, void* stream) {
    if (B < 0 || C < 0 || N < 0 || npoint < 0 || nsample < 0 || B > 65535) return O3D_EINVAL;
    const long Q = (long)npoint * nsample;
    const size_t bytes = sizeof(float) * (size_t)B * C * N;
    if (bytes == 0) return O3D_OK;
    if (!grad_feats) return O3D_EINVAL;
    if (hipMemsetAsync(grad_feats, 0, bytes, o3d_stream(stream)) != hipSuccess) return O3D_ELAUNCH;
    if (Q == 0) return O3D_OK;
    if (!grad_out || !idx || Q > 0x7fffffffL) return O3D_EINVAL;
    hipLaunchKernelGGL(group_points_grad_kernel, dim3(o3d_cdiv(Q, 256), o3d_cdiv(C, GCT), B),
                       dim3(256), 0, o3d_stream(stream), grad_out, idx, C, N, (int)Q, grad_feats);
    return o3d_launch_status();
}

// rows of a point-major tensor: out[b, j, :] = src[b, idx[b, j], :]  (src (B,N,D), idx (B,npoint), out (B,npoint,D)).
// The sampled ball centres new_xyz = gather_operation(xyz^T, idx)^T (pointnet2_modules.py:52-62) without the two
// transposed copies around the channel-major gather.
namespace {
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                          int N, int D, int npoint, long total, float* __restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;      // (b, j, d), d fastest
    if (t >= total) return;
    const long bj = t / D;
    const int d = (int)(t - bj * D);
    const long b = bj / npoint;
    out[t] = src[(b * N + idx[bj]) * D + d];
}
// the same for two tensors over the same points through one index whose rows may be a prefix of a longer index
// (ld_idx >= npoint: idx[b * ld_idx + j]): the seed labels of the trackers (models/bat.py:96-97,132-133)
__global__ __launch_bounds__(256) void gather_rows2_kernel(const float* __restrict__ a, int Da, const float* __restrict__ b_,
                                                           int Db, const int32_t* __restrict__ idx, long ld_idx, int N,
                                                           int npoint, long total, float* __restrict__ outa,
                                                           float* __restrict__ outb) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;      // (b, j, d), d fastest over Da + Db
    if (t >= total) return;
    const int D = Da + Db;
    const long bj = t / D;
    const int d = (int)(t - bj * D);
    const long b = bj / npoint;
    const int j = (int)(bj - b * npoint);
    const long row = b * N + idx[b * ld_idx + j];
    if (d < Da) outa[bj * Da + d] = a[row * Da + d];
    else outb[bj * Db + (d - Da)] = b_[row * Db + (d - Da)];
}
}  // namespace

extern "C" int o3d_gather_rows2(const float* a, int Da, const float* b, int Db, const int32_t* idx, long ld_idx, int B, int N,
                                int npoint, float* outa, float* outb, void* stream) {
    if (B < 0 || N <= 0 || Da <= 0 || Db < 0 || npoint < 0 || ld_idx < npoint || !a || !outa || !idx || (Db > 0 && (!b || !outb)))
        return O3D_EINVAL;
    const long total = (long)B * npoint * (Da + Db);
    if (total == 0) return O3D_OK;
    hipLaunchKernelGGL(gather_rows2_kernel, dim3((unsigned)o3d_cdiv(total, 256)), dim3(256), 0, o3d_stream(stream), a, Da, b, Db,
                       idx, ld_idx, N, npoint, total, outa, outb);
    return o3d_launch_status();
}

extern "C" int o3d_gather_rows(const float* src, const int32_t* idx, int B, int N, int D, int npoint, float* out,
                               void* stream) {
    if (B < 0 || N < 0 || D <= 0 || npoint < 0) return O3D_EINVAL;
    const long total = (long)B * npoint * D;
    if (total == 0) return O3D_OK;
    if (!src || !idx || !out || N == 0) return O3D_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)o3d_cdiv(total, 256)), dim3(256), 0, o3d_stream(stream), src, idx,
                       N, D, npoint, total, out);
    return o3d_launch_status();
}

extern "C" int o3d_gather_points(const float* feats, const int32_t* idx, int B, int C, int N,
                                 int npoint, float* out, void* stream) {
    return o3d_group_points(feats, idx, B, C, N, npoint, 1, out, stream);
}

extern "C" int o3d_gather_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N,
                                      int npoint, float* grad_feats, void* stream) {
    return o3d_group_points_grad(grad_out, idx, B, C, N, npoint, 1, grad_feats, stream);
}

extern "C" int o3d_three_nn(const float* unknown, const float* known, int B, int n, int m,
                            float* dist2, int32_t* idx, void* stream) {
    if (B < 0 || n < 0 || m < 0 || B > 65535) return O3D_EINVAL;
    if (B == 0 || n == 0) return O3D_OK;
    if (!unknown || !dist2 || !idx || bad(known, m)) return O3D_EINVAL;
    hipLaunchKernelGGL(three_nn_kernel, dim3(o3d_cdiv(n, 256), B), dim3(256), 0, o3d_stream(stream),
                       unknown, known, n, m, dist2, idx);
    return o3d_launch_status();
}

extern "C" int o3d_three_interpolate(const float* feats, const int32_t* idx, const float* weight,
                                     int B, int c, int m, int n, float* out, void* stream) {
    if (B < 0 || c < 0 || m < 0 || n < 0 || B > 65535) return O3D_EINVAL;
    if (B == 0 || c == 0 || n == 0) return O3D_OK;
    if (!feats || !idx || !weight || !out) return O3D_EINVAL;
    hipLaunchKernelGGL(three_interpolate_kernel, dim3(o3d_cdiv(n, 256), o3d_cdiv(c, GCT), B),
                       dim3(256), 0, o3d_stream(stream), feats, idx, weight, c, m, n, out);
    return o3d_launch_status();
}

extern "C" int o3d_three_interpolate_grad(const float* grad_out, const int32_t* idx,
                                          const float* weight, int B, int c, int n, int m,
                                          float* grad_feats, void* stream) {
    if (B < 0 || c < 0 || m < 0 || n < 0 || B > 65535) return O3D_EINVAL;
    const size_t bytes = sizeof(float) * (size_t)B * c * m;
    if (bytes == 0) return O3D_OK;
    if (!grad_feats) return O3D_EINVAL;
    if (hipMemsetAsync(grad_feats, 0, bytes, o3d_stream(stream)) != hipSuccess) return O3D_ELAUNCH;
    if (n == 0) return O3D_OK;
    if (!grad_out || !idx || !weight) return O3D_EINVAL;
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(o3d_cdiv(n, 256), o3d_cdiv(c, GCT), B),
                       dim3(256), 0, o3d_stream(stream), grad_out, idx, weight, c, n, m, grad_feats);
    return o3d_launch_status();
}

extern "C" int o3d_knn(const float* query, const float* ref, int B, int Q, int R, int D, int k,
                       int32_t* idx, void* stream) {
    if (B < 0 || Q < 0 || R < 0 || D < 0 || k < 1 || k > 32 || k > R || B > 65535) return O3D_EINVAL;
    if (B == 0 || Q == 0) return O3D_OK;
    if (!idx || bad(query, D) || bad(ref, D)) return O3D_EINVAL;
    const dim3 grid(o3d_cdiv(Q, 128), B), block(128);
    hipStream_t s = o3d_stream(stream);
    if (D == 9 && k <= 4 && (long)R * 9 * 4 <= 48 * 1024)      // BoxAwareXCorr's call (k = 4 on the 9-dim BoxCloud)
        hipLaunchKernelGGL((knn_kernel<4, 9>), grid, block, (size_t)R * 9 * sizeof(float), s, query, ref, Q, R, D, k, idx);
    else if (k <= 4) hipLaunchKernelGGL((knn_kernel<4, 0>), grid, block, 0, s, query, ref, Q, R, D, k, idx);
    else if (k <= 8) hipLaunchKernelGGL((knn_kernel<8, 0>), grid, block, 0, s, query, ref, Q, R, D, k, idx);
    else if (k <= 16) hipLaunchKernelGGL((knn_kernel<16, 0>), grid, block, 0, s, query, ref, Q, R, D, k, idx);
    else hipLaunchKernelGGL((knn_kernel<32, 0>), grid, block, 0, s, query, ref, Q, R, D, k, idx);
    return o3d_launch_status();
}
