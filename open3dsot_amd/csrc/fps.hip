// fps.hip -- furthest point sampling for gfx950 (wave64).
//
// Replaces _ext.furthest_point_sampling (pointnet2/utils/pointnet2_utils.py:56; upstream
// sampling_gpu.cu, semantics restated in oracle/pointnet2_oracle.c and SURVEY.md A.1).
//
// Design (MI355X-first, not the upstream block-of-512 + shared-memory tree):
//   * the op is LATENCY bound: npoint-1 strictly serial arg-max rounds per cloud, a few
//     KB of data.  One workgroup owns one cloud and keeps the whole cloud -- coordinates
//     and the running min-distance -- in VGPRs for all rounds (PPT points per lane), so a
//     round touches no memory except one uniform 12-byte LDS read of the new centre.
//   * N <= 2048: ONE wave per cloud, no barrier anywhere in the round loop; the arg-max
//     is a DPP row reduction (quad_perm / row_half_mirror / row_mirror) + 4 v_readlane.
//     2048 < N <= 8192: 8 waves per cloud, ONE LDS hand-off and one barrier per round (every wave publishes its
//     maximum and its best key at that maximum).
//     N > 8192: generic strided kernel with the min-distance array in caller scratch.
//   * bit-identical indices: the upstream winner among equal maxima is decided by its
//     thread-strided scan + tree reduction, i.e. by
//         rank(k) = bitrev_L(k mod bs) * ceil(N/bs) + (k div bs),  bs = opt_n_threads(N),
//     lowest rank wins (L = log2 bs).  The reduction here is two-phase: wave/block max of
//     the distance, then max of ((0xFFFF - rank) << 16 | k) over the lanes that hold that
//     maximum -- independent of how points are laid over lanes.
#include <cstdlib>
#include "o3d_common.hpp"

namespace {

__device__ __forceinline__ unsigned dpp_max_row_u32(unsigned v) {
    // all-reduce max inside each row of 16 lanes
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false); // row_half_mirror
    v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false); // row_mirror
    v = v > t ? v : t;
    return v;
}

template <bool USE_DPP>
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    if (USE_DPP) {
        v = dpp_max_row_u32(v);
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
        const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
        const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
        const unsigned d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
        return ab > cd ? ab : cd;
    } else {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned t = (unsigned)__shfl_xor((int)v, off, 64);
            v = v > t ? v : t;
        }
        return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    }
}

__device__ __forceinline__ unsigned fps_rank(unsigned k, int bs_log2, unsigned cpb) {
    const unsigned tid = k & ((1u << bs_log2) - 1u);
    const unsigned rev = bs_log2 == 0 ? 0u : (__brev(tid) >> (32 - bs_log2));
    return rev * cpb + (k >> bs_log2);
}

// A second, independent set of clouds sampled by the same launch (workgroups >= B0): the template and the
// search cloud of a tracker step are one wave per cloud each -- 2 x 48 workgroups on 256 CUs -- so the two
// samplings run side by side instead of back to back.  xyz == NULL: one set.
struct FpsSet1 {
    const float* xyz;
    int32_t* idx;
    int N, npoint, bs_log2, cpb, B0;
};

// One workgroup (NW waves) per cloud, PPT points per lane in registers.  N <= 64*NW*PPT,
// N < 65535.  LDS: N*3 floats (cloud copy) + 4*NW words (cross-wave exchange).
template <int PPT, int NW, bool USE_DPP>
__global__ __launch_bounds__(64 * NW) void fps_reg_kernel(const float* __restrict__ xyz, int N,
                                                          int npoint, int bs_log2, int cpb,
                                                          int32_t* __restrict__ idx, FpsSet1 s1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int cloud = blockIdx.x;
    if (s1.xyz && cloud >= s1.B0) {
        cloud -= s1.B0;
        xyz = s1.xyz; idx = s1.idx; N = s1.N; npoint = s1.npoint; bs_log2 = s1.bs_log2; cpb = s1.cpb;
    }
    float* s_xyz = smem;
    unsigned* s_x = reinterpret_cast<unsigned*>(smem + (size_t)N * 3);  // [2][2][NW]: {maximum, key} per wave, two rounds
    constexpr int T = 64 * NW;
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const float* p = xyz + (size_t)cloud * N * 3;
    int32_t* out = idx + (size_t)cloud * npoint;

    for (int t = tid; t < N * 3; t += T) s_xyz[t] = p[t];
    __syncthreads();

    float px[PPT], py[PPT], pz[PPT], tmp[PPT];
    unsigned lo[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + T * i;
        const bool in = k < N;
        const float x = in ? s_xyz[3 * k + 0] : 0.f;
        const float y = in ? s_xyz[3 * k + 1] : 0.f;
        const float z = in ? s_xyz[3 * k + 2] : 0.f;
        const float mag = __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x)));
        const bool valid = in && !o3d_fps_near_origin(mag);
        // skipped / padded slots: min-distance pinned at +0 and key 0 => never a winner
        px[i] = x; py[i] = y; pz[i] = z;
        tmp[i] = valid ? 1e10f : 0.f;
        lo[i] = valid ? (((0xFFFFu - fps_rank((unsigned)k, bs_log2, (unsigned)cpb)) << 16) | (unsigned)k) : 0u;
    }

    if (tid == 0) out[0] = 0;
    int old = 0;
    for (int j = 1; j < npoint; ++j) {
        const float x1 = s_xyz[3 * old + 0], y1 = s_xyz[3 * old + 1], z1 = s_xyz[3 * old + 2];
        float hb = 0.f;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float d = o3d_sqdist3(px[i], py[i], pz[i], x1, y1, z1);
            const float d2 = fminf(d, tmp[i]);
            tmp[i] = d2;
            hb = fmaxf(hb, d2);
        }
        // phase 1: maximum distance over the WAVE's points (non-negative floats order like u32)
        const unsigned M = wave_max_u32<USE_DPP>(__float_as_uint(hb));
        // phase 2: lowest upstream rank among the wave's points at that distance
        unsigned c = 0u;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const unsigned cand = (__float_as_uint(tmp[i]) == M) ? lo[i] : 0u;
            c = c > cand ? c : cand;
        }
        unsigned L = wave_max_u32<USE_DPP>(c);
        if (NW > 1) {
            // ONE hand-off per round: every wave publishes (its maximum, its best key at that maximum); the cloud's
            // winner is the best key among the waves that hold the overall maximum.  Double-buffered by the round's
            // parity, so one barrier per round is enough: a wave can only overwrite a buffer two rounds later, i.e.
            // after the barrier of the round in between, which every reader of the old contents has reached.
            unsigned* exM = s_x + ((j & 1) ? 2 * NW : 0);
            unsigned* exL = exM + NW;
            if ((tid & 63) == 0) { exM[wave] = M; exL[wave] = L; }
            __syncthreads();
            unsigned Mg = 0u, Lg = 0u;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const unsigned m = exM[w], l = exL[w];
                if (m > Mg || (m == Mg && l > Lg)) { Mg = m; Lg = l; }
            }
            L = Lg;
        }
        old = (int)(L & 0xFFFFu);  // L == 0 (nothing selectable) -> index 0, as upstream
        if (tid == 0) out[j] = old;
    }
}

// Generic fallback, any N: 1024 threads per cloud, min-distance array in global scratch.
__global__ __launch_bounds__(1024) void fps_generic_kernel(const float* __restrict__ xyz, int N,
                                                           int npoint, int bs_log2, int cpb,
                                                           float* __restrict__ temp,
                                                           int32_t* __restrict__ idx) {
    __shared__ unsigned long long s_key[2][16];
    const int tid = threadIdx.x, wave = tid >> 6;
    const float* p = xyz + (size_t)blockIdx.x * N * 3;
    float* tmp = temp + (size_t)blockIdx.x * N;
    int32_t* out = idx + (size_t)blockIdx.x * npoint;
    for (int k = tid; k < N; k += 1024) tmp[k] = 1e10f;
    if (tid == 0) out[0] = 0;
    __syncthreads();
    int old = 0;
    for (int j = 1; j < npoint; ++j) {
        const float x1 = p[3 * old], y1 = p[3 * old + 1], z1 = p[3 * old + 2];
        unsigned long long best = 0ull;
        for (int k = tid; k < N; k += 1024) {
            const float x = p[3 * k], y = p[3 * k + 1], z = p[3 * k + 2];
            const float mag = __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x)));
            if (o3d_fps_near_origin(mag)) continue;
            const float d2 = fminf(o3d_sqdist3(x, y, z, x1, y1, z1), tmp[k]);
            tmp[k] = d2;
            // key: distance bits, then inverted rank (rank < 2^31 so the low word is never 0)
            const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) |
                                           (0xFFFFFFFFu - fps_rank((unsigned)k, bs_log2, (unsigned)cpb));
            best = key > best ? key : best;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long t = __shfl_xor(best, off, 64);
            best = t > best ? t : best;
        }
        if ((tid & 63) == 0) s_key[j & 1][wave] = best;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 16; ++w) { const unsigned long long t = s_key[j & 1][w]; best = t > best ? t : best; }
        if (best == 0ull) {
            old = 0;
        } else {
            const unsigned r = 0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull);
            const unsigned t = r / (unsigned)cpb, q = r % (unsigned)cpb;
            const unsigned rev = bs_log2 == 0 ? 0u : (__brev(t) >> (32 - bs_log2));
            old = (int)(rev + (q << bs_log2));
        }
        if (tid == 0) out[j] = old;
        __syncthreads();  // temp[] writes of this round visible before the next one reads them
    }
}

template <int PPT, int NW, bool USE_DPP>
int launch_reg(const float* xyz, int B, int N, int npoint, int bs_log2, int cpb, int32_t* idx,
               hipStream_t s) {
    const size_t lds = (size_t)N * 3 * sizeof(float) + 4 * NW * sizeof(unsigned);
    hipLaunchKernelGGL((fps_reg_kernel<PPT, NW, USE_DPP>), dim3(B), dim3(64 * NW), lds, s, xyz, N,
                       npoint, bs_log2, cpb, idx, FpsSet1{});
    return o3d_launch_status();
}

static void fps_rank_params(int N, int& bs_log2, int& cpb) {
    bs_log2 = 0;
    while ((2 << bs_log2) <= N && bs_log2 < 9) ++bs_log2;  // bs = opt_n_threads(N)
    cpb = (N + (1 << bs_log2) - 1) >> bs_log2;
}

// waves per cloud for 256 < N <= 2048: 1 (no hand-off).  The 4-wave form (a quarter of the per-round distance updates per
// wave, one LDS hand-off + barrier per round) measured SLOWER on the MI355X, same-box A/B at 512 / 1024 points: BAT step
// 6.45 vs 6.43 ms, batch-1 frame 1.021 vs 1.006 ms -- the round is ~100 VALU instructions shorter, the barrier and the LDS
// round trip cost more.  Multi-wave instantiations stay for N > 2048 (the cloud no longer fits one wave's registers).

// both sets in one launch of the register kernel sized for the larger cloud
template <int PPT, int NW>
int launch_pair(const float* xyz0, int N0, int np0, int32_t* idx0, const float* xyz1, int N1, int np1,
                int32_t* idx1, int B, hipStream_t s) {
    int l0, c0, l1, c1;
    fps_rank_params(N0, l0, c0);
    fps_rank_params(N1, l1, c1);
    const int Nmax = N0 > N1 ? N0 : N1;
    const size_t lds = (size_t)Nmax * 3 * sizeof(float) + 4 * NW * sizeof(unsigned);
    const FpsSet1 s1 = {xyz1, idx1, N1, np1, l1, c1, B};
    hipLaunchKernelGGL((fps_reg_kernel<PPT, NW, true>), dim3(2 * B), dim3(64 * NW), lds, s, xyz0, N0, np0, l0, c0, idx0,
                       s1);
    return o3d_launch_status();
}

template <bool USE_DPP>
int fps_dispatch(const float* xyz, int B, int N, int npoint, float* temp, int32_t* idx,
                 hipStream_t s) {
    int bs_log2 = 0;
    while ((2 << bs_log2) <= N && bs_log2 < 9) ++bs_log2;  // bs = opt_n_threads(N)
    const int bs = 1 << bs_log2;
    const int cpb = (N + bs - 1) / bs;
    if (N <= 64) return launch_reg<1, 1, USE_DPP>(xyz, B, N, npoint, bs_log2, cpb, idx, s);
    if (N <= 128) return launch_reg<2, 1, USE_DPP>(xyz, B, N, npoint, bs_log2, cpb, idx, s);
    if (N <= 256) return launch_reg<4, 1, USE_DPP>(xyz, B, N, npoint, bs_log2, cpb, idx, s);
    if (N <= 512) return launch_reg<8, 1, USE_DPP>(xyz, B, N, npoint, bs_log2, cpb, idx, s);
    if (N <= 1024) return launch_reg<16, 1, USE_DPP>(xyz, B, N, npoint, bs_log2, cpb, idx, s);
    if (N <= 2048) return launch_reg<32, 1, USE_DPP>(xyz, B, N, npoint, bs_log2, cpb, idx, s);
    if (N <= 4096) return launch_reg<8, 8, USE_DPP>(xyz, B, N, npoint, bs_log2, cpb, idx, s);
    if (N <= 8192) return launch_reg<16, 8, USE_DPP>(xyz, B, N, npoint, bs_log2, cpb, idx, s);
    // beyond 8192 points the cloud no longer fits the LDS staging of the register kernels (N*12 B against
    // 160 KB per CU from N = 13649 on): generic kernel, running min-distance in the caller's `temp`
    if (!temp) return O3D_EINVAL;
    hipLaunchKernelGGL(fps_generic_kernel, dim3(B), dim3(1024), 0, s, xyz, N, npoint, bs_log2, cpb,
                       temp, idx);
    return o3d_launch_status();
}

}  // namespace

extern "C" int o3d_furthest_point_sampling(const float* xyz, int B, int N, int npoint, float* temp,
                                           int32_t* idx, void* stream) {
    if (B < 0 || N <= 0 || npoint < 0 || (B > 0 && npoint > 0 && (!xyz || !idx))) return O3D_EINVAL;
    if (B == 0 || npoint == 0) return O3D_OK;
    return fps_dispatch<true>(xyz, B, N, npoint, temp, idx, o3d_stream(stream));
}

// Two independent sets of B clouds (N0 / N1 points, npoint0 / npoint1 samples) in ONE launch; each set's
// result is bit-identical to o3d_furthest_point_sampling on it.  max(N0, N1) <= 2048 (one wave per cloud),
// npoint >= 1; otherwise O3D_EINVAL -- make the two calls.
extern "C" int o3d_furthest_point_sampling_pair(const float* xyz0, int N0, int npoint0, int32_t* idx0,
                                                const float* xyz1, int N1, int npoint1, int32_t* idx1, int B,
                                                void* stream) {
    const int Nmax = N0 > N1 ? N0 : N1;
    if (B <= 0 || N0 <= 0 || N1 <= 0 || npoint0 <= 0 || npoint1 <= 0 || Nmax > 2048 || !xyz0 || !xyz1 || !idx0 || !idx1)
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    if (Nmax <= 256) return launch_pair<4, 1>(xyz0, N0, npoint0, idx0, xyz1, N1, npoint1, idx1, B, s);
    if (Nmax <= 512) return launch_pair<8, 1>(xyz0, N0, npoint0, idx0, xyz1, N1, npoint1, idx1, B, s);
    if (Nmax <= 1024) return launch_pair<16, 1>(xyz0, N0, npoint0, idx0, xyz1, N1, npoint1, idx1, B, s);
    return launch_pair<32, 1>(xyz0, N0, npoint0, idx0, xyz1, N1, npoint1, idx1, B, s);
}

// Test hook: same op through the ds_bpermute (__shfl_xor) reduction instead of DPP.
extern "C" int o3d_furthest_point_sampling_shfl(const float* xyz, int B, int N, int npoint,
                                                float* temp, int32_t* idx, void* stream) {
    if (B < 0 || N <= 0 || npoint < 0 || (B > 0 && npoint > 0 && (!xyz || !idx))) return O3D_EINVAL;
    if (B == 0 || npoint == 0) return O3D_OK;
    return fps_dispatch<false>(xyz, B, N, npoint, temp, idx, o3d_stream(stream));
}
