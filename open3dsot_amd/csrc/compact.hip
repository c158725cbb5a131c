// compact.hip -- the grouped MLP on DISTINCT neighbours only.
//
// ball_query pads every ball to nsample slots by repeating its first hit (SURVEY.md Appendix A.2;
// pointnet2_utils.py:268).  All copies of one neighbour in one ball see the same input column, hence
// the same value at every layer (1x1 convolutions, per-channel BatchNorm, ReLU), and max-pooling
// ignores copies.  On KITTI-like crops 70-80 % of the slots are such copies (tools/exp/group_probe.py),
// so the reference's kernels -- and a slot-faithful port of them -- spend three quarters of their
// FLOPs and bytes on duplicates.  Here every ball keeps its cnt distinct entries as columns of one
// flat (C, Ptot) matrix; the first hit carries the weight w = 1 + (nsample - cnt) of its copies:
//     BatchNorm statistics        sum_q w_q*y_q, sum_q w_q*(y_q - c)^2          (count stays B*npoint*nsample)
//     backward                    dY_q = A1*dN_q + w_q*(A2*Y_q + A3)            (dN_q = class sum of the copies' dN)
// which is exactly the slot-wise sum, because everything downstream of dY is linear in it (weight
// gradient, W^T dY, ReLU mask identical for copies, next BatchNorm's sums).  Ptot is data dependent:
// buffers and grids are sized for the worst case (all slots distinct), meta[0] = Ptot rounded up to
// 256 lives in device memory, and tiles beyond it return at once -- no host synchronisation, so the
// step stays capturable in a HIP graph.
//
// Layout: positions of cloud b, ball j are contiguous and ordered (b, j, k); per column
//   gp[q]    = b*ld + idx[b,j,k]      source point, as a column of the per-point matrices (C, B*ld)
//   cball[q] = b*npoint + j           ball id (dummy id B*npoint for the padding columns)
//   cw[q]    = weight
#include "o3d_common.hpp"

namespace {

// one lane per slot; balls of ns (power of two <= 64) consecutive lanes
__global__ __launch_bounds__(256) void compact_count_kernel(const int32_t* __restrict__ idx, long nslots, int ns,
                                                            int32_t* __restrict__ ball_cnt) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int v = s < nslots ? idx[s] : -1;
    const int base = lane & ~(ns - 1), k = lane - base;
    const int first = __shfl(v, base, 64);
    // copies = the maximal SUFFIX of slots equal to the first hit (ball_query's padding); an index list
    // without that structure (e.g. k-NN output) simply keeps all its slots
    const unsigned long long m = __ballot(v != first);
    const unsigned long long seg = ns == 64 ? ~0ull : (((1ull << ns) - 1) << base);
    const unsigned long long other = m & seg;
    if (k == 0 && s < nslots) ball_cnt[s / ns] = other ? (63 - __clzll((long long)other)) - base + 1 : 1;
}

// exclusive scan of ball_cnt (n <= 65536) by one workgroup; meta = {Ptot rounded up to 256, Ptot, n, 0}
__global__ __launch_bounds__(1024) void compact_scan_kernel(const int32_t* __restrict__ ball_cnt, int n,
                                                            int32_t* __restrict__ ball_off,
                                                            int32_t* __restrict__ meta) {
    __shared__ int sh[1024];
    const int per = (n + 1023) / 1024;
    const int i0 = threadIdx.x * per;
    int local = 0;
    for (int i = i0; i < i0 + per && i < n; ++i) local += ball_cnt[i];
    sh[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    int run = sh[threadIdx.x] - local;
    for (int i = i0; i < i0 + per && i < n; ++i) {
        ball_off[i] = run;
        run += ball_cnt[i];
    }
    if (threadIdx.x == 1023) {
        const int tot = sh[1023];
        ball_off[n] = tot;
        meta[0] = (tot + 255) & ~255;
        meta[1] = tot;
        meta[2] = n;
        meta[3] = 0;
    }
}

__global__ __launch_bounds__(256) void compact_fill_kernel(const int32_t* __restrict__ idx, long nslots, int ns,
                                                           int npoint, int ld,
                                                           const int32_t* __restrict__ ball_cnt,
                                                           const int32_t* __restrict__ ball_off,
                                                           const int32_t* __restrict__ meta,
                                                           int32_t* __restrict__ gp, int32_t* __restrict__ cball,
                                                           float* __restrict__ cw) {
    if (blockIdx.x == gridDim.x - 1) {       // padding columns [Ptot, Ptot_pad)
        const int q = meta[1] + threadIdx.x;
        if (q < meta[0]) { gp[q] = 0; cball[q] = meta[2]; cw[q] = 0.f; }
        return;
    }
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= nslots) return;
    const int ball = (int)(s / ns), k = (int)(s - (long)ball * ns);
    const int cnt = ball_cnt[ball];
    if (k >= cnt) return;
    const int q = ball_off[ball] + k;
    gp[q] = (ball / npoint) * ld + idx[s];
    cball[q] = ball;
    cw[q] = k == 0 ? (float)(1 + ns - cnt) : 1.f;
}

// ---------------------------------------------------------------------------------------
// expand: Y0[c,q] = Z[c,gp[q]] - W0[c,0:3].centre[cball[q]], weighted statistics partials
//   workgroup = 256 columns, 4 waves; wave w takes channels w, w+4, ...; lane = 4 columns
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_c_kernel(const float* __restrict__ Z, long ldz,
                                                       const int32_t* __restrict__ gp,
                                                       const int32_t* __restrict__ cball,
                                                       const float* __restrict__ cw,
                                                       const float* __restrict__ centers,
                                                       const float* __restrict__ W0, int ldw, int C0,
                                                       const int32_t* __restrict__ meta, long ldp,
                                                       float* __restrict__ Y0, float* __restrict__ part,
                                                       const float* __restrict__ stat_c) {
    const int q0 = blockIdx.x * 256;
    if (q0 >= meta[0]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = q0 + 4 * lane;
    const int4 id = *reinterpret_cast<const int4*>(&gp[q]);
    const float4 w = *reinterpret_cast<const float4*>(&cw[q]);
    float cx[4] = {0.f, 0.f, 0.f, 0.f}, cy[4] = {0.f, 0.f, 0.f, 0.f}, cz[4] = {0.f, 0.f, 0.f, 0.f};
    if (centers) {
        const int4 bj = *reinterpret_cast<const int4*>(&cball[q]);
        const int bb[4] = {bj.x, bj.y, bj.z, bj.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float* c = centers + (long)bb[t] * 3;
            cx[t] = c[0]; cy[t] = c[1]; cz[t] = c[2];
        }
    }
    for (int co = wave; co < C0; co += 4) {
        const float* z = Z + (long)co * ldz;
        float w0 = 0.f, w1 = 0.f, w2 = 0.f;
        if (centers) { const float* wr = W0 + (long)co * ldw; w0 = wr[0]; w1 = wr[1]; w2 = wr[2]; }
        float4 y;
        y.x = z[id.x] - fmaf(w2, cz[0], fmaf(w1, cy[0], w0 * cx[0]));
        y.y = z[id.y] - fmaf(w2, cz[1], fmaf(w1, cy[1], w0 * cx[1]));
        y.z = z[id.z] - fmaf(w2, cz[2], fmaf(w1, cy[2], w0 * cx[2]));
        y.w = z[id.w] - fmaf(w2, cz[3], fmaf(w1, cy[3], w0 * cx[3]));
        *reinterpret_cast<float4*>(&Y0[(long)co * ldp + q]) = y;
        if (part) {
            const float c = stat_c ? stat_c[co] : 0.f;
            float s = fmaf(w.x, y.x, fmaf(w.y, y.y, fmaf(w.z, y.z, w.w * y.w)));
            float v = w.x * (y.x - c) * (y.x - c) + w.y * (y.y - c) * (y.y - c) + w.z * (y.z - c) * (y.z - c) +
                      w.w * (y.w - c) * (y.w - c);
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { s += __shfl_xor(s, m, 64); v += __shfl_xor(v, m, 64); }
            if (lane == 0) {
                part[((long)blockIdx.x * 2 + 0) * C0 + co] = s;
                part[((long)blockIdx.x * 2 + 1) * C0 + co] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// pool: out[b,c,j] = max over the ball's columns of relu(Y*scale+shift); argq = column of the max
//   8 lanes per ball (its columns are contiguous, 8 on average): a wave reads the columns of 8
//   consecutive balls, i.e. one nearly contiguous row segment; first-maximum arg-max via 3 shuffles.
//   workgroup = 32 balls x POOL_CH channels.
// ---------------------------------------------------------------------------------------
constexpr int POOL_CH = 8;

__global__ __launch_bounds__(256) void pool_c_kernel(const float* __restrict__ Y, long ldp,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift,
                                                     const int32_t* __restrict__ ball_off, int C, int npoint,
                                                     int nballs, float* __restrict__ out,
                                                     int32_t* __restrict__ argq, float* __restrict__ yarg) {
    const int e = threadIdx.x & 7;
    int ball = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool live = ball < nballs;
    if (!live) ball = nballs - 1;
    const int q0 = ball_off[ball], q1 = ball_off[ball + 1];
    const int b = ball / npoint, j = ball - b * npoint;
    const int c0 = blockIdx.y * POOL_CH;
    for (int cc = 0; cc < POOL_CH && c0 + cc < C; ++cc) {
        const int c = c0 + cc;
        const float sc = scale[c], sf = shift[c];
        const float* y = Y + (long)c * ldp;
        float best = -INFINITY, yb = 0.f;
        int bq = 0x7fffffff;
        for (int q = q0 + e; q < q1; q += 8) {
            const float v = y[q], n = fmaf(v, sc, sf);
            if (n > best) { best = n; bq = q; yb = v; }
        }
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64), oy = __shfl_xor(yb, off, 64);
            const int oq = __shfl_xor(bq, off, 64);
            if (ob > best || (ob == best && oq < bq)) { best = ob; bq = oq; yb = oy; }
        }
        if (live && e == 0) {
            const long o = ((long)b * C + c) * npoint + j;
            out[o] = fmaxf(best, 0.f);
            if (argq) { argq[o] = bq; yarg[o] = yb; }
        }
    }
}

// dense gradient of the pooled layer: zero the live columns, then one value per (c, ball)
__global__ __launch_bounds__(256) void zero_cols_kernel(float* __restrict__ D, long ldp,
                                                        const int32_t* __restrict__ meta) {
    const int q = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (q >= meta[0]) return;
    *reinterpret_cast<float4*>(&D[(long)blockIdx.y * ldp + q]) = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void pool_scatter_c_kernel(const float* __restrict__ dOut,
                                                             const float* __restrict__ out,
                                                             const int32_t* __restrict__ argq, int C,
                                                             int npoint, long total, long ldp,
                                                             float* __restrict__ D) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;     // (b, c, j)
    if (o >= total) return;
    const int c = (int)((o / npoint) % C);
    if (out[o] > 0.f) D[(long)c * ldp + argq[o]] = dOut[o];
}

// ---------------------------------------------------------------------------------------
// layer-0 backward reduce: dY = A1*dN + w*(A2*Y0 + A3);  S[c, b*ld+n] = sum of dY over the columns that
// reference point n, T[c, ball] = sum of dY over the ball.  Workgroup = (cloud, CS channels), LDS fp32
// atomics (no duplicates inside a ball any more, so no same-address pile-ups).
// ---------------------------------------------------------------------------------------
template <int CS>
__global__ __launch_bounds__(256) void reduce_c_kernel(const float* __restrict__ dN, const float* __restrict__ Y0,
                                                       long ldp, const float* __restrict__ A1,
                                                       const float* __restrict__ A2,
                                                       const float* __restrict__ A3,
                                                       const int32_t* __restrict__ gp,
                                                       const int32_t* __restrict__ cball,
                                                       const float* __restrict__ cw,
                                                       const int32_t* __restrict__ ball_off, int npoint, int ld,
                                                       int C0, long lds_row, float* __restrict__ S,
                                                       float* __restrict__ T, int nballs) {
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [CS][ld] then [CS][npoint]
    const int slabs = (C0 + CS - 1) / CS;
    const int b = blockIdx.x / slabs, c0 = (blockIdx.x - b * slabs) * CS;
    float* tacc = acc + CS * ld;
    for (int i = threadIdx.x; i < CS * (ld + npoint); i += 256) acc[i] = 0.f;
    __syncthreads();
    const int q0 = ball_off[b * npoint], q1 = ball_off[(b + 1) * npoint];
    float a1[CS], a2[CS], a3[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        const int cc = c0 + c < C0 ? c0 + c : C0 - 1;
        a1[c] = A1[cc]; a2[c] = A2[cc]; a3[c] = A3[cc];
    }
    // A ball's columns are consecutive, so its sum T is a segmented reduction over lanes (inclusive
    // segmented scan by ball id, the last lane of each run adds the run total once): per-element LDS
    // atomics on T would pile the ~8 columns of a ball onto one address (ds_add_f32 retires roughly
    // one lane per 8 cycles per CU, measured with tools/exp/group_probe.py).
    const int lane = threadIdx.x & 63;
    const int span = q1 - q0;
    for (int i0 = 0; i0 < span; i0 += 256) {
        const int q = q0 + i0 + threadIdx.x;
        const bool live = q < q1;
        const int n = live ? gp[q] - b * ld : 0, j = live ? cball[q] - b * npoint : -1;
        const float w = live ? cw[q] : 0.f;
        unsigned same = 0;           // bit s: lane - 2^s belongs to the same ball
        if (T) {
#pragma unroll
            for (int sft = 0; sft < 6; ++sft) {
                const int ju = __shfl_up(j, 1 << sft, 64);
                if (lane >= (1 << sft) && ju == j) same |= 1u << sft;
            }
        }
        const int jn = __shfl_down(j, 1, 64);
        const bool tail = live && (lane == 63 || jn != j);
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            if (c0 + c >= C0) break;
            float dy = 0.f;
            if (live) {
                const long o = (long)(c0 + c) * ldp + q;
                dy = fmaf(a1[c], dN[o], w * fmaf(a2[c], Y0[o], a3[c]));
                atomicAdd(&acc[c * ld + n], dy);
            }
            if (T) {
                float run = dy;
#pragma unroll
                for (int sft = 0; sft < 6; ++sft) {
                    const float up = __shfl_up(run, 1 << sft, 64);
                    if (same & (1u << sft)) run += up;
                }
                if (tail) atomicAdd(&tacc[c * npoint + j], run);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CS * ld; i += 256) {
        const int c = i / ld, n = i - c * ld;
        if (c0 + c < C0) S[(long)(c0 + c) * lds_row + (long)b * ld + n] = acc[i];
    }
    if (T)
        for (int i = threadIdx.x; i < CS * npoint; i += 256) {
            const int c = i / npoint, j = i - c * npoint;
            if (c0 + c < C0) T[(long)(c0 + c) * nballs + b * npoint + j] = tacc[i];
        }
}

}  // namespace

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// idx (B,npoint,ns) -> ball_cnt (B*npoint), ball_off (B*npoint+1), gp/cball/cw (B*npoint*ns worst case),
// meta (4 ints).  ld = row length of the per-point matrices per cloud (N rounded up).
extern "C" int o3d_compact_build(const int32_t* idx, int B, int npoint, int ns, int ld, int32_t* ball_cnt,
                                 int32_t* ball_off, int32_t* gp, int32_t* cball, float* cw, int32_t* meta,
                                 void* stream) {
    const long nballs = (long)B * npoint, nslots = nballs * ns;
    if (!idx || !ball_cnt || !ball_off || !gp || !cball || !cw || !meta || B <= 0 || npoint <= 0 || ns < 1 ||
        ns > 64 || !pow2(ns) || nballs > 65536 || nslots % 256 != 0 || ld <= 0)
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    const int blocks = (int)(nslots / 256);
    hipLaunchKernelGGL(compact_count_kernel, dim3(blocks), dim3(256), 0, s, idx, nslots, ns, ball_cnt);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, ball_cnt, (int)nballs, ball_off, meta);
    hipLaunchKernelGGL(compact_fill_kernel, dim3(blocks + 1), dim3(256), 0, s, idx, nslots, ns, npoint, ld, ball_cnt,
                       ball_off, meta, gp, cball, cw);
    return o3d_launch_status();
}

// Y0 (C0, ldp) from Z (C0, ldz); centers ((nballs+1), 3) or NULL; part [ldp/256][2][C0] or NULL
extern "C" int o3d_group_expand_c(const float* Z, long ldz, const int32_t* gp, const int32_t* cball,
                                  const float* cw, const float* centers, const float* W0, int ldw, int C0,
                                  const int32_t* meta, long ldp, float* Y0, float* part, const float* stat_c,
                                  void* stream) {
    if (!Z || !gp || !cball || !cw || !meta || !Y0 || C0 <= 0 || ldp <= 0 || ldp % 256 != 0 || (centers && (!W0 || ldw < 3)))
        return O3D_EINVAL;
    hipLaunchKernelGGL(expand_c_kernel, dim3((unsigned)(ldp / 256)), dim3(256), 0, o3d_stream(stream), Z, ldz, gp,
                       cball, cw, centers, W0, ldw, C0, meta, ldp, Y0, part, stat_c);
    return o3d_launch_status();
}

extern "C" int o3d_pool_fwd_c(const float* Y, long ldp, const float* scale, const float* shift,
                              const int32_t* ball_off, int B, int C, int npoint, float* out, int32_t* argq,
                              float* yarg, void* stream) {
    if (!Y || !scale || !shift || !ball_off || !out || B <= 0 || C <= 0 || npoint <= 0 || (argq && !yarg))
        return O3D_EINVAL;
    const int nballs = B * npoint;
    hipLaunchKernelGGL(pool_c_kernel, dim3(o3d_cdiv(nballs, 32), o3d_cdiv(C, POOL_CH)), dim3(256), 0,
                       o3d_stream(stream), Y, ldp, scale, shift, ball_off, C, npoint, nballs, out, argq, yarg);
    return o3d_launch_status();
}

// D (C, ldp): zero on the live columns, then D[c, argq[b,c,j]] = dOut[b,c,j] where out > 0
extern "C" int o3d_pool_bwd_dense_c(const float* dOut, const float* out, const int32_t* argq, int B, int C,
                                    int npoint, const int32_t* meta, long ldp, float* D, void* stream) {
    if (!dOut || !out || !argq || !meta || !D || B <= 0 || C <= 0 || npoint <= 0 || ldp <= 0 || ldp % 4 != 0)
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)o3d_cdiv(ldp, 1024), C), dim3(256), 0, s, D, ldp, meta);
    const long total = (long)B * C * npoint;
    hipLaunchKernelGGL(pool_scatter_c_kernel, dim3(o3d_cdiv(total, 256)), dim3(256), 0, s, dOut, out, argq, C, npoint,
                       total, ldp, D);
    return o3d_launch_status();
}

template <int CS>
static int launch_reduce_c(const float* dN, const float* Y0, long ldp, const float* A1, const float* A2,
                           const float* A3, const int32_t* gp, const int32_t* cball, const float* cw,
                           const int32_t* ball_off, int B, int npoint, int ld, int C0, float* S, float* T,
                           hipStream_t s) {
    const size_t lds = sizeof(float) * CS * (size_t)(ld + npoint);
    if (lds > 64 * 1024) return O3D_EINVAL;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(reduce_c_kernel<CS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return O3D_ELAUNCH;
    const int slabs = (C0 + CS - 1) / CS;
    hipLaunchKernelGGL(reduce_c_kernel<CS>, dim3(B * slabs), dim3(256), lds, s, dN, Y0, ldp, A1, A2, A3, gp, cball, cw,
                       ball_off, npoint, ld, C0, (long)B * ld, S, T, B * npoint);
    return o3d_launch_status();
}

// S (C0, B*ld), T (C0, B*npoint) or NULL
extern "C" int o3d_group_reduce_c(const float* dN, const float* Y0, long ldp, const float* A1, const float* A2,
                                  const float* A3, const int32_t* gp, const int32_t* cball, const float* cw,
                                  const int32_t* ball_off, int B, int npoint, int ld, int C0, float* S, float* T,
                                  void* stream) {
    if (!dN || !Y0 || !A1 || !A2 || !A3 || !gp || !cball || !cw || !ball_off || !S || B <= 0 || npoint <= 0 ||
        ld <= 0 || C0 <= 0)
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    if ((long)B * C0 >= 8192 && sizeof(float) * 4 * (size_t)(ld + npoint) <= 64 * 1024)
        return launch_reduce_c<4>(dN, Y0, ldp, A1, A2, A3, gp, cball, cw, ball_off, B, npoint, ld, C0, S, T, s);
    return launch_reduce_c<2>(dN, Y0, ldp, A1, A2, A3, gp, cball, cw, ball_off, B, npoint, ld, C0, S, T, s);
}
