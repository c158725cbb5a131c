// group.hip -- layer 0 of a grouped MLP without the 16-32x redundant contraction.
//
// The reference gathers first and convolves second (QueryAndGroup -> SharedMLP layer 0,
// pointnet2_utils.py:299-339 + pytorch_utils.py:12-37): every source point is multiplied by W0
// once per ball it falls into (P = npoint*nsample positions, N = 2*npoint points: 16x at
// nsample 32).  A 1x1 convolution commutes with a gather, so here the order is swapped:
//
//   forward   Z  = W0 . [xyz ; feats]                      one small GEMM over the N points
//             Y0[b,co,p] = Z[b,co,idx[b,p]] - W0[co,0:3] . new_xyz[b,j(p)]     (expand, HBM write-bound)
//   backward  S[b,co,n] = sum_{p : idx[b,p]==n} dY0[b,co,p]                   (reduce, HBM read-bound)
//             T[b,co,j] = sum_k dY0[b,co,j*ns+k]
//             dW0 = S . [xyz;feats]^T - T . new_xyz^T,   d[xyz;feats] = W0^T . S,   dnew_xyz = -W0x^T . T
//   and because dY0 = A1*dN0 + A2*Y0 + A3 (BatchNorm backward folded into per-channel constants,
//   see mlp.hip) is affine in dN0 and Y0, the sums of Y0 never touch the (B,C0,P) tensor again:
//             sum_{p in list(n)} Y0[co,p] = cnt[n]*Z[co,n] - W0[co,0:3] . R[n],   R[n] = sum of the
//             centres of the balls that contain n;      sum_k Y0[co,j,k] = GY[co,j] (kept by expand).
//
// These are byte-moving kernels: coalesced float4 rows along the position axis, the scattered side
// (Z rows of <= 4 KB, per-cloud accumulators) lives in L1/L2 or LDS.  No MFMA here.
#include "o3d_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// cnt[b,n] = number of positions that reference point n;  R[b,n,:] = sum of their centres
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_meta_kernel(const int32_t* __restrict__ idx,
                                                         const float* __restrict__ new_xyz, int N, int ld,
                                                         int npoint, int ns, float* __restrict__ cnt,
                                                         float* __restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [N] counts, [3N] centre sums
    const int b = blockIdx.x, P = npoint * ns;
    for (int i = threadIdx.x; i < 4 * N; i += 256) sm[i] = 0.f;
    __syncthreads();
    const int32_t* id = idx + (long)b * P;
    // A ball lists distinct points and pads with its first hit (Appendix A.2 of SURVEY.md): all
    // references to the first hit are counted by the ball's first lane and added once -- without
    // this the padded slots of one ball serialise on a single LDS address.
    const int lane = threadIdx.x & 63;
    if (ns <= 64 && (ns & (ns - 1)) == 0 && (P & 255) == 0) {
        for (int p = threadIdx.x; p < P; p += 256) {
            const int n = id[p];
            const int first = __shfl(n, lane & ~(ns - 1), 64);
            const bool is_first = n == first;
            unsigned long long m = __ballot(is_first);
            const int base = lane & ~(ns - 1);
            const unsigned long long seg = ns == 64 ? ~0ull : (((1ull << ns) - 1) << base);
            const int k = __popcll(m & seg);
            const float* c = new_xyz ? new_xyz + ((long)b * npoint + p / ns) * 3 : nullptr;
            if (lane == base) {
                atomicAdd(&sm[n], (float)k);
                if (c) {
                    atomicAdd(&sm[N + 3 * n + 0], k * c[0]);
                    atomicAdd(&sm[N + 3 * n + 1], k * c[1]);
                    atomicAdd(&sm[N + 3 * n + 2], k * c[2]);
                }
            } else if (!is_first) {
                atomicAdd(&sm[n], 1.f);
                if (c) {
                    atomicAdd(&sm[N + 3 * n + 0], c[0]);
                    atomicAdd(&sm[N + 3 * n + 1], c[1]);
                    atomicAdd(&sm[N + 3 * n + 2], c[2]);
                }
            }
        }
    } else {
        for (int p = threadIdx.x; p < P; p += 256) {
            const int n = id[p];
            atomicAdd(&sm[n], 1.f);
            if (new_xyz) {
                const float* c = new_xyz + ((long)b * npoint + p / ns) * 3;
                atomicAdd(&sm[N + 3 * n + 0], c[0]);
                atomicAdd(&sm[N + 3 * n + 1], c[1]);
                atomicAdd(&sm[N + 3 * n + 2], c[2]);
            }
        }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < ld; n += 256) cnt[(long)b * ld + n] = n < N ? sm[n] : 0.f;
    if (R)
        for (int i = threadIdx.x; i < 3 * ld; i += 256) R[(long)b * ld * 3 + i] = i < 3 * N ? sm[N + i] : 0.f;
}

// ---------------------------------------------------------------------------------------
// expand: Y0 = Z[idx] - W0x . new_xyz, per-tile BatchNorm partials, per-ball sums GY
//   workgroup = (cloud b, 256 positions), 4 waves; wave w handles channels w, w+4, ...;
//   a lane owns 4 consecutive positions (one float4 store per channel row).
// ---------------------------------------------------------------------------------------
constexpr int EXP_TP = 256;

__global__ __launch_bounds__(256) void group_expand_kernel(const float* __restrict__ Z, int ldz,
                                                           const int32_t* __restrict__ idx,
                                                           const float* __restrict__ new_xyz,
                                                           const float* __restrict__ W0, int ldw, int C0,
                                                           int npoint, int ns, float* __restrict__ Y0,
                                                           float* __restrict__ part,
                                                           const float* __restrict__ stat_c,
                                                           float* __restrict__ GY) {
    const int P = npoint * ns;
    const int tiles_per_b = P / EXP_TP;
    const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x;
    const int p0 = (blockIdx.x - b * tiles_per_b) * EXP_TP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = p0 + 4 * lane;
    const int4 id = *reinterpret_cast<const int4*>(&idx[(long)b * P + p]);
    const int j = p / ns;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (new_xyz) {
        const float* c = new_xyz + ((long)b * npoint + j) * 3;
        cx = c[0]; cy = c[1]; cz = c[2];
    }
    const int seg = ns / 4;   // lanes per ball (power of two <= 64)
    for (int co = wave; co < C0; co += 4) {
        const float* z = Z + ((long)b * C0 + co) * ldz;
        float cc = 0.f;
        if (new_xyz) {
            const float* w = W0 + (long)co * ldw;
            cc = fmaf(w[2], cz, fmaf(w[1], cy, w[0] * cx));
        }
        float4 y;
        y.x = z[id.x] - cc; y.y = z[id.y] - cc; y.z = z[id.z] - cc; y.w = z[id.w] - cc;
        *reinterpret_cast<float4*>(&Y0[((long)b * C0 + co) * P + p]) = y;
        if (part) {
            const float c = stat_c ? stat_c[co] : 0.f;
            float s = (y.x + y.y) + (y.z + y.w);
            float q = (y.x - c) * (y.x - c) + (y.y - c) * (y.y - c) + (y.z - c) * (y.z - c) + (y.w - c) * (y.w - c);
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                if (m == seg && GY && (lane & (seg - 1)) == 0) GY[((long)b * C0 + co) * npoint + j] = s;
                s += __shfl_xor(s, m, 64);
                q += __shfl_xor(q, m, 64);
            }
            if (seg == 64 && GY && lane == 0) GY[((long)b * C0 + co) * npoint + j] = s;
            if (lane == 0) {
                part[((long)tile * 2 + 0) * C0 + co] = s;
                part[((long)tile * 2 + 1) * C0 + co] = q;
            }
        } else if (GY) {
            float s = (y.x + y.y) + (y.z + y.w);
            for (int m = 1; m < seg; m <<= 1) s += __shfl_xor(s, m, 64);
            if ((lane & (seg - 1)) == 0) GY[((long)b * C0 + co) * npoint + j] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------
// reduce: SdN[b,co,n] = sum_{p in list(n)} dN[b,co,p]   TdN[b,co,j] = sum_k dN[b,co,j*ns+k]
//   workgroup = (cloud b, CS channels); per-channel accumulators [CS][ld] in LDS, fp32 LDS atomics
//   (upstream scatters with global atomics too, pointnet2_utils.py:237; the summation order inside a
//   point's list is not fixed).  A ball lists distinct points and pads with its first hit (SURVEY.md
//   Appendix A.2): everything that refers to the first hit is summed over the ball with shuffles and
//   added once -- otherwise the padded slots of one ball serialise on a single LDS address.
//   The workgroup of channel slab 0 also produces the index-only quantities cnt[b,n] (references to
//   point n) and R[b,n,:] (sum of the centres of the balls referencing n) as four virtual channels.
// ---------------------------------------------------------------------------------------
#ifdef O3D_EXP_NOATOMIC   // experiment only (tools/exp): how much of the kernel is LDS-atomic time
#define O3D_LDS_ADD(p, v) (*(p) = (v))
#else
#define O3D_LDS_ADD(p, v) atomicAdd((p), (v))
#endif
__device__ __forceinline__ float scatter4(float* a, const float4& v, const int4& id, int first, bool fx, bool fy,
                                          bool fz, bool fw, bool leader, int seg) {
    float f = (fx ? v.x : 0.f) + (fy ? v.y : 0.f) + (fz ? v.z : 0.f) + (fw ? v.w : 0.f);
    float s = (v.x + v.y) + (v.z + v.w);
    for (int m = 1; m < seg; m <<= 1) { f += __shfl_xor(f, m, 64); s += __shfl_xor(s, m, 64); }
    if (leader) O3D_LDS_ADD(&a[first], f);
    if (!fx) O3D_LDS_ADD(&a[id.x], v.x);
    if (!fy) O3D_LDS_ADD(&a[id.y], v.y);
    if (!fz) O3D_LDS_ADD(&a[id.z], v.z);
    if (!fw) O3D_LDS_ADD(&a[id.w], v.w);
    return s;      // ball sum (valid on every lane of the ball)
}

template <int CS>
__global__ __launch_bounds__(256) void group_reduce_kernel(const float* __restrict__ dN,
                                                           const int32_t* __restrict__ idx, int C0, int ld,
                                                           int npoint, int ns, float* __restrict__ SdN,
                                                           float* __restrict__ TdN,
                                                           const float* __restrict__ new_xyz,
                                                           float* __restrict__ cnt, float* __restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [CS (+4 for slab 0)][ld]
    const int P = npoint * ns;
    const int slabs = (C0 + CS - 1) / CS;
    const int b = blockIdx.x / slabs, c0 = (blockIdx.x - b * slabs) * CS;
    const int lane = threadIdx.x & 63;
    const bool meta = cnt != nullptr && c0 == 0;
    const int rows = CS + (meta ? 4 : 0);
    for (int i = threadIdx.x; i < rows * ld; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int32_t* id_b = idx + (long)b * P;
    const int seg = ns / 4;
    for (int q4 = threadIdx.x; q4 < P / 4; q4 += 256) {
        const int4 id = *reinterpret_cast<const int4*>(&id_b[4 * q4]);
        const int j = (4 * q4) / ns;
        const int first = __shfl(id.x, lane & ~(seg - 1), 64);    // first hit of this lane's ball
        const bool fx = id.x == first, fy = id.y == first, fz = id.z == first, fw = id.w == first;
        const bool leader = (lane & (seg - 1)) == 0;
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            if (c0 + c >= C0) break;
            const float4 v = *reinterpret_cast<const float4*>(&dN[((long)b * C0 + c0 + c) * P + 4 * q4]);
            const float s = scatter4(acc + c * ld, v, id, first, fx, fy, fz, fw, leader, seg);
            if (TdN && leader) TdN[((long)b * C0 + c0 + c) * npoint + j] = s;
        }
        if (meta) {
            scatter4(acc + CS * ld, make_float4(1.f, 1.f, 1.f, 1.f), id, first, fx, fy, fz, fw, leader, seg);
            if (R) {
                const float* ctr = new_xyz + ((long)b * npoint + j) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float cv = ctr[k];
                    scatter4(acc + (CS + 1 + k) * ld, make_float4(cv, cv, cv, cv), id, first, fx, fy, fz, fw, leader, seg);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CS * ld; i += 256) {
        const int c = i / ld, n = i - c * ld;
        if (c0 + c < C0) SdN[((long)b * C0 + c0 + c) * ld + n] = acc[i];
    }
    if (meta) {
        for (int n = threadIdx.x; n < ld; n += 256) {
            cnt[(long)b * ld + n] = acc[CS * ld + n];
            if (R) {
                float* r = R + ((long)b * ld + n) * 3;
                r[0] = acc[(CS + 1) * ld + n]; r[1] = acc[(CS + 2) * ld + n]; r[2] = acc[(CS + 3) * ld + n];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// combine: S = A1*SdN + A2*(cnt*Z - W0x.R) + A3*cnt     T = A1*TdN + A2*GY + A3*ns   (in place)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_combine_kernel(float* __restrict__ S, float* __restrict__ T,
                                                            const float* __restrict__ Z,
                                                            const float* __restrict__ GY,
                                                            const float* __restrict__ cnt,
                                                            const float* __restrict__ R,
                                                            const float* __restrict__ W0, int ldw,
                                                            const float* __restrict__ A1,
                                                            const float* __restrict__ A2,
                                                            const float* __restrict__ A3, int C0, int ld,
                                                            int npoint, int ns, long nS, long nT) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nS) {
        const int n = (int)(i % ld);
        const long bc = i / ld;
        const int co = (int)(bc % C0);
        const long b = bc / C0;
        const float k = cnt[b * ld + n];
        float ly = k * Z[i];
        if (R) {
            const float* r = R + (b * ld + n) * 3;
            const float* w = W0 + (long)co * ldw;
            ly -= fmaf(w[2], r[2], fmaf(w[1], r[1], w[0] * r[0]));
        }
        S[i] = fmaf(A1[co], S[i], fmaf(A2[co], ly, A3[co] * k));
    } else if (i < nS + nT) {
        const long t = i - nS;
        const int co = (int)((t / npoint) % C0);
        T[t] = fmaf(A1[co], T[t], fmaf(A2[co], GY[t], A3[co] * (float)ns));
    }
}

}  // namespace

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int o3d_group_meta(const int32_t* idx, const float* new_xyz, int B, int N, int ld, int npoint,
                              int ns, float* cnt, float* R, void* stream) {
    if (!idx || !cnt || B <= 0 || N <= 0 || ld < N || N > 8192 || npoint <= 0 || ns <= 0 || (R && !new_xyz))
        return O3D_EINVAL;
    const size_t lds = sizeof(float) * 4 * (size_t)N;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(group_meta_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return O3D_ELAUNCH;
    }
    hipLaunchKernelGGL(group_meta_kernel, dim3(B), dim3(256), lds, o3d_stream(stream), idx, R ? new_xyz : nullptr,
                       N, ld, npoint, ns, cnt, R);
    return o3d_launch_status();
}

extern "C" int o3d_group_expand_fwd(const float* Z, int ldz, const int32_t* idx, const float* new_xyz,
                                    const float* W0, int ldw, int B, int C0, int npoint, int ns, float* Y0,
                                    float* part, const float* stat_c, float* GY, void* stream) {
    const long P = (long)npoint * ns;
    if (!Z || !idx || !Y0 || B <= 0 || C0 <= 0 || npoint <= 0 || ns < 4 || ns > 256 || !pow2(ns) ||
        P % EXP_TP != 0 || (new_xyz && (!W0 || ldw < 3)))
        return O3D_EINVAL;
    hipLaunchKernelGGL(group_expand_kernel, dim3((unsigned)(B * (P / EXP_TP))), dim3(256), 0, o3d_stream(stream), Z,
                       ldz, idx, new_xyz, W0, ldw, C0, npoint, ns, Y0, part, stat_c, GY);
    return o3d_launch_status();
}

template <int CS>
static int launch_reduce(const float* dN, const int32_t* idx, int B, int C0, int ld, int npoint, int ns, float* SdN,
                         float* TdN, const float* new_xyz, float* cnt, float* R, hipStream_t s) {
    const size_t lds = sizeof(float) * (CS + (cnt ? 4 : 0)) * (size_t)ld;
    if (lds > 64 * 1024) return O3D_EINVAL;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(group_reduce_kernel<CS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return O3D_ELAUNCH;
    const int slabs = (C0 + CS - 1) / CS;
    hipLaunchKernelGGL(group_reduce_kernel<CS>, dim3(B * slabs), dim3(256), lds, s, dN, idx, C0, ld, npoint, ns, SdN,
                       TdN, new_xyz, cnt, R);
    return o3d_launch_status();
}

extern "C" int o3d_group_reduce_bwd(const float* dN, const int32_t* idx, int B, int C0, int ld, int npoint,
                                    int ns, float* SdN, float* TdN, const float* new_xyz, float* cnt, float* R,
                                    void* stream) {
    if (!dN || !idx || !SdN || B <= 0 || C0 <= 0 || ld <= 0 || npoint <= 0 || ns < 4 || ns > 256 || !pow2(ns) ||
        ((long)npoint * ns) % 256 != 0 || (R && (!cnt || !new_xyz)))
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    // enough workgroups to fill 256 CUs several times over: fewer channels per workgroup on small layers
    const long rows = (long)B * C0;
    if (rows >= 8 * 2048) return launch_reduce<8>(dN, idx, B, C0, ld, npoint, ns, SdN, TdN, new_xyz, cnt, R, s);
    if (rows >= 4 * 2048) return launch_reduce<4>(dN, idx, B, C0, ld, npoint, ns, SdN, TdN, new_xyz, cnt, R, s);
    return launch_reduce<2>(dN, idx, B, C0, ld, npoint, ns, SdN, TdN, new_xyz, cnt, R, s);
}

extern "C" int o3d_group_bwd_combine(float* S, float* T, const float* Z, const float* GY, const float* cnt,
                                     const float* R, const float* W0, int ldw, const float* A1, const float* A2,
                                     const float* A3, int B, int C0, int ld, int npoint, int ns, void* stream) {
    if (!S || !Z || !cnt || !A1 || !A2 || !A3 || B <= 0 || C0 <= 0 || ld <= 0 || (T && !GY) || (R && !W0))
        return O3D_EINVAL;
    const long nS = (long)B * C0 * ld, nT = T ? (long)B * C0 * npoint : 0;
    hipLaunchKernelGGL(group_combine_kernel, dim3(o3d_cdiv(nS + nT, 256)), dim3(256), 0, o3d_stream(stream), S, T, Z,
                       GY, cnt, R, W0, ldw, A1, A2, A3, C0, ld, npoint, ns, nS, nT);
    return o3d_launch_status();
}
