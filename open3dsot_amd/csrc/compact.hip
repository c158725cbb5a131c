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
#include <cstdlib>
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

// exclusive scan of ball_cnt (any n: ceil(n / 1024) balls per thread) by one workgroup; meta = {Ptot rounded up to 256, Ptot, n, 0}
__global__ __launch_bounds__(1024) void compact_scan_kernel(const int32_t* __restrict__ ball_cnt, int n,
                                                            int col_base, int32_t* __restrict__ ball_off,
                                                            int32_t* __restrict__ meta) {
    __shared__ int sh[1024];
    const int per = (n + 1023) / 1024;
    const int i0 = threadIdx.x * per;
    int local = 0;
    for (int i = i0; i < i0 + per && i < n; ++i) local += ball_cnt[i];
    sh[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    int run = col_base + sh[threadIdx.x] - local;
    for (int i = i0; i < i0 + per && i < n; ++i) {
        ball_off[i] = run;
        run += ball_cnt[i];
    }
    if (threadIdx.x == 1023) {
        const int tot = sh[1023];
        meta[0] = (tot + 255) & ~255;
        meta[1] = tot;
        meta[2] = n;
        meta[3] = 0;
    }
}

__global__ __launch_bounds__(256) void compact_fill_kernel(const int32_t* __restrict__ idx, long nslots, int ns,
                                                           int npoint, int ld, int col_base, int pt_base,
                                                           int ball_base, int dummy_ball,
                                                           const int32_t* __restrict__ ball_cnt,
                                                           const int32_t* __restrict__ ball_off,
                                                           const int32_t* __restrict__ meta,
                                                           int32_t* __restrict__ gp, int32_t* __restrict__ cball,
                                                           float* __restrict__ cw) {
    if (blockIdx.x == gridDim.x - 1) {       // padding columns [Ptot, Ptot_pad)
        const int q = meta[1] + threadIdx.x;
        if (q < meta[0]) { gp[col_base + q] = 0; cball[col_base + q] = dummy_ball; cw[col_base + q] = 0.f; }
        return;
    }
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= nslots) return;
    const int ball = (int)(s / ns), k = (int)(s - (long)ball * ns);
    const int cnt = ball_cnt[ball];
    if (k >= cnt) return;
    const int q = ball_off[ball] + k;                     // ball_off already includes col_base
    gp[q] = pt_base + (ball / npoint) * ld + idx[s];
    cball[q] = ball_base + ball;
    cw[q] = k == 0 ? (float)(1 + ns - cnt) : 1.f;
}

// ---------------------------------------------------------------------------------------
// expand: Y0[c,q] = Z[c,gp[q]] - W0[c,0:3].centre[cball[q]], weighted statistics partials
//   workgroup = 256 columns, 4 waves; wave w takes channels w, w+4, ...; lane = 4 columns
// ---------------------------------------------------------------------------------------
// DIRECT (round 3): an xyz-only layer 0 (SA level 0: no features) needs no per-point GEMM at all --
// Y0[c,q] = W0[c,0:3].(xyz[gp[q]] - centre[cball[q]]): Z then IS the packed coordinate operand (3 rows), gathered once per
// column, and the "centre term" carries the whole product (the relative coordinate first, as the reference computes it)
template <bool CENTERS, bool DIRECT = false>      // compile-time: a run-time `if (centers)` around a load makes the compiler drain vmcnt at the join
__global__ __launch_bounds__(256) void expand_c_kernel(const float* __restrict__ Z, long ldz,
                                                       const int32_t* __restrict__ gp,
                                                       const int32_t* __restrict__ cball,
                                                       const float* __restrict__ cw,
                                                       const float* __restrict__ centers,
                                                       const float* __restrict__ W0, int ldw, int C0,
                                                       const int32_t* __restrict__ meta, long start1, long ldp,
                                                       float* __restrict__ Y0, float* __restrict__ part,
                                                       const float* __restrict__ stat_c) {
    const int q0 = blockIdx.x * 256;
    const int seg = (start1 > 0 && q0 >= start1) ? 1 : 0;
    if (q0 - (seg ? start1 : 0) >= meta[4 * seg]) return;
    if (stat_c && seg) stat_c += C0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = q0 + 4 * lane;
    const int4 id = *reinterpret_cast<const int4*>(&gp[q]);
    const float4 w = *reinterpret_cast<const float4*>(&cw[q]);
    float cx[4] = {0.f, 0.f, 0.f, 0.f}, cy[4] = {0.f, 0.f, 0.f, 0.f}, cz[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (CENTERS) {
        const int4 bj = *reinterpret_cast<const int4*>(&cball[q]);
        const int bb[4] = {bj.x, bj.y, bj.z, bj.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float* c = centers + (long)bb[t] * 3;
            cx[t] = c[0]; cy[t] = c[1]; cz[t] = c[2];
        }
        if constexpr (DIRECT) {
            const int ids[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) { cx[t] -= Z[ids[t]]; cy[t] -= Z[ldz + ids[t]]; cz[t] -= Z[2 * ldz + ids[t]]; }
        }
    }
    // channels in groups of EG per wave: all 4*EG gathers in flight first, and the EG x 2 statistics butterflies
    // interleave (one channel at a time the 12 dependent shuffles are a ~300-cycle chain per channel)
    constexpr int EG = 8;
    const int cper = (C0 + gridDim.y - 1) / gridDim.y;           // channel range of this workgroup
    const int cbeg = blockIdx.y * cper, cend = cbeg + cper < C0 ? cbeg + cper : C0;
    for (int g0 = cbeg + wave; g0 < cend; g0 += 4 * EG) {
        float4 y[EG];
        float w0[EG], w1[EG], w2[EG], sc_[EG];
#pragma unroll
        for (int g = 0; g < EG; ++g) {
            const int co = g0 + 4 * g < cend ? g0 + 4 * g : cend - 1;
            if constexpr (DIRECT) {
                y[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const float* z = Z + (long)co * ldz;
                y[g].x = z[id.x]; y[g].y = z[id.y]; y[g].z = z[id.z]; y[g].w = z[id.w];
            }
            // the per-channel constants with the gathers, not between the stores below: a load issued after a store
            // makes the wait for it wait for the store as well (vmcnt counts both on gfx9) -- EG round trips per pass
            w0[g] = w1[g] = w2[g] = 0.f;
            if constexpr (CENTERS) { const float* wr = W0 + (long)co * ldw; w0[g] = wr[0]; w1[g] = wr[1]; w2[g] = wr[2]; }
            sc_[g] = (part && stat_c) ? stat_c[co] : 0.f;
        }
        float s[EG], v[EG];
#pragma unroll
        for (int g = 0; g < EG; ++g) {
            const int co = g0 + 4 * g;
            if (co >= cend) { s[g] = 0.f; v[g] = 0.f; continue; }
            y[g].x -= fmaf(w2[g], cz[0], fmaf(w1[g], cy[0], w0[g] * cx[0]));
            y[g].y -= fmaf(w2[g], cz[1], fmaf(w1[g], cy[1], w0[g] * cx[1]));
            y[g].z -= fmaf(w2[g], cz[2], fmaf(w1[g], cy[2], w0[g] * cx[2]));
            y[g].w -= fmaf(w2[g], cz[3], fmaf(w1[g], cy[3], w0[g] * cx[3]));
            *reinterpret_cast<float4*>(&Y0[(long)co * ldp + q]) = y[g];
            const float c = sc_[g];
            s[g] = fmaf(w.x, y[g].x, fmaf(w.y, y[g].y, fmaf(w.z, y[g].z, w.w * y[g].w)));
            v[g] = w.x * (y[g].x - c) * (y[g].x - c) + w.y * (y[g].y - c) * (y[g].y - c) +
                   w.z * (y[g].z - c) * (y[g].z - c) + w.w * (y[g].w - c) * (y[g].w - c);
        }
        if (part) {
#pragma unroll
            for (int g = 0; g < EG; ++g) { s[g] = wave_sum(s[g]); v[g] = wave_sum(v[g]); }
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

// ---------------------------------------------------------------------------------------
// pool: out[c,ball] = max over the ball's columns of relu(Y*scale+shift); argq = column of the max
//   8 lanes per ball (its columns are contiguous, 8 on average): a wave reads the columns of 8
//   consecutive balls, i.e. one nearly contiguous row segment; first-maximum arg-max via 3 shuffles.
//   workgroup = 32 balls x POOL_CH channels.
// ---------------------------------------------------------------------------------------
constexpr int POOL_BWD_SPLIT = 8;       // = O3D_POOL_BWD_SPLIT (include/o3dsot.h)

// pooled tensors are stored per segment in the reference's (B, C, npoint) layout, segment 1's block after
// segment 0's: element (c, ball)
__device__ __forceinline__ long pool_index(int c, int ball, int C, int seg1_ball, int np0, int np1) {
    const bool s1 = ball >= seg1_ball;
    const int local = s1 ? ball - seg1_ball : ball, np = s1 ? np1 : np0;
    const int b = local / np, j = local - b * np;
    return (s1 ? (long)seg1_ball * C : 0) + ((long)b * C + c) * np + j;
}

// (dpp_f / dpp_i of o3d_common.hpp) lane <- lane ^ 1, lane ^ 2 (quad permutes) and lane <- 7 - lane within each group of
// 8 (row_half_mirror): after the three steps lane 0 of every 8-lane group has combined all 8 lanes.
template <int POOL_CH>
__global__ __launch_bounds__(256) void pool_c_kernel(const float* __restrict__ Y, long ldp,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift,
                                                     const int32_t* __restrict__ ball_off,
                                                     const int32_t* __restrict__ ball_cnt, int C, int seg1_ball,
                                                     int np0, int np1, int nballs, float* __restrict__ out,
                                                     int32_t* __restrict__ argq, float* __restrict__ yarg) {
    const int e = threadIdx.x & 7;
    int ball = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool live = ball < nballs;
    if (!live) ball = nballs - 1;
    const int q0 = ball_off[ball], q1 = q0 + ball_cnt[ball];
    const int c0 = blockIdx.y * POOL_CH;
    if (ball >= seg1_ball) { scale += C; shift += C; }       // second segment: its own BatchNorm constants
    // ALL loads of the workgroup's channels first -- the 4 columns per lane and channel that cover a 32-column ball
    // (clamped, not predicated) and the per-channel constants -- then the compares, then ALL stores: written channel by
    // channel the kernel is a chain of store -> load constants -> s_waitcnt vmcnt(0) round trips (stores count in vmcnt
    // on gfx9, so every wait for the next channel's constants also waited for the previous channel's stores)
    constexpr int PU = 4;
    float v[POOL_CH][PU], sc[POOL_CH], sf[POOL_CH];
#pragma unroll
    for (int cc = 0; cc < POOL_CH; ++cc) {
        const int c = c0 + cc < C ? c0 + cc : C - 1;
        const float* y = Y + (long)c * ldp;
#pragma unroll
        for (int i = 0; i < PU; ++i) {
            const int q = q0 + e + 8 * i;
            v[cc][i] = y[q < q1 ? q : q1 - 1];
        }
        sc[cc] = scale[c];
        sf[cc] = shift[c];
    }
    float rbest[POOL_CH], ryb[POOL_CH];
    int rbq[POOL_CH];
#pragma unroll
    for (int cc = 0; cc < POOL_CH; ++cc) {
        float best = -INFINITY, yb = 0.f;
        int bq = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PU; ++i) {
            const int q = q0 + e + 8 * i;
            const float n = fmaf(v[cc][i], sc[cc], sf[cc]);
            if (q < q1 && n > best) { best = n; bq = q; yb = v[cc][i]; }
        }
        if (q1 - q0 > 8 * PU) {                               // balls of more than 32 columns (no tracker has them)
            const float* y = Y + (long)(c0 + cc < C ? c0 + cc : C - 1) * ldp;
            for (int q = q0 + e + 8 * PU; q < q1; q += 8) {
                const float w = y[q], n = fmaf(w, sc[cc], sf[cc]);
                if (n > best) { best = n; bq = q; yb = w; }
            }
        }
        // first maximum of the 8 lanes (largest value, smallest column on ties): xor 1, xor 2, mirror
        {
            const float ob = dpp_f<0xB1>(best), oy = dpp_f<0xB1>(yb);
            const int oq = dpp_i<0xB1>(bq);
            if (ob > best || (ob == best && oq < bq)) { best = ob; bq = oq; yb = oy; }
        }
        {
            const float ob = dpp_f<0x4E>(best), oy = dpp_f<0x4E>(yb);
            const int oq = dpp_i<0x4E>(bq);
            if (ob > best || (ob == best && oq < bq)) { best = ob; bq = oq; yb = oy; }
        }
        {
            const float ob = dpp_f<0x141>(best), oy = dpp_f<0x141>(yb);
            const int oq = dpp_i<0x141>(bq);
            if (ob > best || (ob == best && oq < bq)) { best = ob; bq = oq; yb = oy; }
        }
        rbest[cc] = best; ryb[cc] = yb; rbq[cc] = bq;
    }
    if (live && e == 0) {
#pragma unroll
        for (int cc = 0; cc < POOL_CH; ++cc) {
            const int c = c0 + cc;
            if (c < C) {
                const long o = pool_index(c, ball, C, seg1_ball, np0, np1);
                out[o] = fmaxf(rbest[cc], 0.f);
                if (argq) { argq[o] = rbq[cc]; yarg[o] = ryb[cc]; }
            }
        }
    }
}

// The same pooling with the tile TRANSPOSED through LDS (round 3).  pool_c_kernel gives every ball 8 lanes x 4 clamped
// loads per channel -- 32 slots for a ball that holds ~10 distinct neighbours on the tracker crops: the SQ counters show it
// issue-bound (30 % active, 16 % waiting for memory), two thirds of the lane-operations spent on clamped duplicates.
// Here a workgroup loads a (64 channels) x (128 + 32 columns) tile of Y with coalesced float4 rows into LDS, then
// lane = channel and a wave walks ONE ball's columns at a time (a wave-uniform trip count, no idle lanes): one ds_read, one
// fma and a compare / select per element.  A ball belongs to the 256-column chunk its first column lies in (it may run up
// to 31 columns into the next chunk: the tile is 160 wide).  Strict `>` over ascending columns keeps the first maximum,
// as pool_c_kernel does.  Results leave through a second LDS staging so that the stores run along the ball index.
// C % 64 == 0, balls of at most 32 columns.
constexpr int PT_CH = 32, PT_COLS = 128, PT_OVER = 32, PT_LD = PT_COLS + PT_OVER + 3;     // odd stride: conflict-free; +3: the 4-wide reads
constexpr int PT_RB = 32, PT_RLD = PT_CH + 1;         // balls per output batch; padded row of the result staging

// one work item of pool_t_kernel: the PT_COLS-column chunk at q0 x the PT_CH channels from c0
__device__ __forceinline__ void pool_t_item(const float* __restrict__ Y, long ldp,
                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                            const int32_t* __restrict__ ball_off,
                                            const int32_t* __restrict__ ball_cnt,
                                            const int32_t* __restrict__ cball,
                                            const int32_t* __restrict__ meta, long start1, int C,
                                            int seg1_ball, int np0, int np1, int nballs,
                                            float* __restrict__ out, int32_t* __restrict__ argq,
                                            float* __restrict__ yarg, const long q0, const int c0, float* pt_tile) {
    // LDS: tile [PT_CH][PT_LD] | ball table off / cnt / output base [PT_COLS] each | results [3][PT_RB][PT_RLD]
    int* pt_off = reinterpret_cast<int*>(pt_tile + PT_CH * PT_LD);
    int* pt_cnt = pt_off + PT_COLS;
    int* pt_ob = pt_cnt + PT_COLS;
    float* pt_res = reinterpret_cast<float*>(pt_ob + PT_COLS);
    const int seg = (start1 > 0 && q0 >= start1) ? 1 : 0;
    const long sbase = seg ? start1 : 0;
    if (q0 - sbase >= meta[4 * seg]) return;            // dead chunk
    const long qend = sbase + meta[4 * seg + 1];        // one past the segment's last real column
    if (q0 >= qend) return;                             // only padding columns: no ball starts here
    // balls that START in this chunk: [bfirst, blast] (at most PT_COLS: one per column)
    int bfirst = cball[q0];
    if (ball_off[bfirst] < q0) ++bfirst;
    const long qlast = (q0 + PT_COLS < qend ? q0 + PT_COLS : qend) - 1;
    const int blast = cball[qlast];
    const int nb = blast - bfirst + 1;
    const int np = seg ? np1 : np0;
    const long segbase = seg ? (long)seg1_ball * C : 0;
    const int ball0 = seg ? seg1_ball : 0;
    // ---- tile: rows c0 .. c0+63, columns q0 .. q0+159 (clamped into the buffer), coalesced float4 along the columns;
    // the ball table of the chunk rides along (a per-ball global load inside the walk below was a ~1 us latency per ball)
    constexpr int F4 = (PT_COLS + PT_OVER) / 4;         // 40 float4 per row
    constexpr int NLD = PT_CH * F4 / 256;               // 10 float4 per thread: ALL in flight before the first LDS store
    static_assert(PT_CH * F4 % 256 == 0, "tile loads must divide evenly over the workgroup");
    {                                                   // (a load -> wait -> store loop was one exposed HBM latency per row)
        float4 v[NLD];
        int bo = 0, bc = 0;
        if ((int)threadIdx.x < nb) { bo = ball_off[bfirst + threadIdx.x]; bc = ball_cnt[bfirst + threadIdx.x]; }
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = threadIdx.x + 256 * u;
            const int r = i / F4, f = i - r * F4;
            long q = q0 + 4 * f;
            if (q + 4 > ldp) q = ldp - 4;                // beyond the buffer: columns no ball of this chunk reaches
            v[u] = *reinterpret_cast<const float4*>(&Y[(long)(c0 + r) * ldp + q]);
        }
        __builtin_amdgcn_sched_barrier(0);              // (the scheduler otherwise sinks the stores between the loads)
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = threadIdx.x + 256 * u;
            const int r = i / F4, f = i - r * F4;
            float* d = &pt_tile[r * PT_LD + 4 * f];
            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
        if ((int)threadIdx.x < PT_COLS) {
            const int local = bfirst + (int)threadIdx.x - ball0;       // pooled tensors: (B, C, npoint) blocks per segment
            const int b = local / np, j = local - b * np;
            pt_off[threadIdx.x] = bo;
            pt_cnt[threadIdx.x] = bc;
            pt_ob[threadIdx.x] = (int)(segbase + (long)b * C * np + j);   // + c * np: element (c, ball); < 2^31 (nballs*C)
        }
    }
    // lane = (ball of a pair, channel): 32 channels per workgroup keep four workgroups resident per CU (a 64-channel tile
    // with its staging is 68 KB: two), each half-wave walks its own ball
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = lane & (PT_CH - 1), half = lane / PT_CH;
    constexpr int BPW = 64 / PT_CH;                     // balls a wave walks at a time
    const float sc = scale[seg * C + c0 + ch], sf = shift[seg * C + c0 + ch];
    __syncthreads();
    const float* row = &pt_tile[ch * PT_LD];
    const int wj = threadIdx.x & 31, wc = threadIdx.x >> 5;      // write-out: lane bits 0-4 = ball of the batch, 8 channel lanes
    for (int j0 = 0; j0 < nb; j0 += PT_RB) {
        // ---- this wave's 8 balls of the batch: lane = channel, a wave-uniform walk over the ball's columns
#pragma unroll 1
        for (int t = 0; t < PT_RB / 4; t += BPW) {
            const int jl = j0 + wave * (PT_RB / 4) + t + half;
            if (jl - half >= nb) break;
            const bool has = jl < nb;
            const int off = has ? pt_off[jl] : 0, cnt = has ? pt_cnt[jl] : 0;        // LDS broadcasts
            const int rel = (int)(off - q0);
            float best = -INFINITY, yb = 0.f;
            int bq = 0x7fffffff;
            for (int k = 0; k < cnt; k += 4) {          // four LDS reads in flight (within the row: rel + k + 3 <= 162)
                const float w0 = row[rel + k], w1 = row[rel + k + 1], w2 = row[rel + k + 2], w3 = row[rel + k + 3];
                const float n0 = fmaf(w0, sc, sf), n1 = fmaf(w1, sc, sf), n2 = fmaf(w2, sc, sf), n3 = fmaf(w3, sc, sf);
                if (n0 > best) { best = n0; bq = off + k; yb = w0; }
                if (k + 1 < cnt && n1 > best) { best = n1; bq = off + k + 1; yb = w1; }
                if (k + 2 < cnt && n2 > best) { best = n2; bq = off + k + 2; yb = w2; }
                if (k + 3 < cnt && n3 > best) { best = n3; bq = off + k + 3; yb = w3; }
            }
            if (has) {
                float* r = &pt_res[(jl - j0) * PT_RLD + ch];
                r[0] = fmaxf(best, 0.f);
                r[PT_RB * PT_RLD] = __int_as_float(bq);
                r[2 * PT_RB * PT_RLD] = yb;
            }
        }
        __syncthreads();
        // ---- write-out with lanes along the balls: (B, C, npoint) rows are contiguous in the ball index, so a wave
        // stores two 128-byte runs per instruction (lane = channel scattered 64 four-byte writes over 64 lines: 16x
        // the write requests, measured slower than the kernel it replaces)
        if (j0 + wj < nb) {
            const int ob = pt_ob[j0 + wj];
#pragma unroll
            for (int cc = wc; cc < PT_CH; cc += 8) {
                const long o = (long)ob + (long)(c0 + cc) * np;
                const float* r = &pt_res[wj * PT_RLD + cc];
                out[o] = r[0];
                if (argq) { argq[o] = __float_as_int(r[PT_RB * PT_RLD]); yarg[o] = r[2 * PT_RB * PT_RLD]; }
            }
        }
        __syncthreads();
    }
}

// PERSISTENT grid (round 5, as pool_bwd_dense_kernel): a fixed number of workgroups walks the LIVE (chunk, channel group) items
// -- chunks that hold at least one real column, from the device-side meta -- instead of a worst-case grid two thirds of which
// are workgroups that read the live count and return
constexpr int PT_GRID = 4096;
__global__ __launch_bounds__(256) void pool_t_kernel(const float* __restrict__ Y, long ldp,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const int32_t* __restrict__ ball_off,
                                                     const int32_t* __restrict__ ball_cnt,
                                                     const int32_t* __restrict__ cball,
                                                     const int32_t* __restrict__ meta, long start1, int C,
                                                     int seg1_ball, int np0, int np1, int nballs,
                                                     float* __restrict__ out, int32_t* __restrict__ argq,
                                                     float* __restrict__ yarg) {
    extern __shared__ float pt_tile[];
    const int ngrp = C / PT_CH;
    const int n0 = (meta[1] + PT_COLS - 1) / PT_COLS;
    const int n1 = start1 > 0 ? (meta[5] + PT_COLS - 1) / PT_COLS : 0;
    const long items = (long)(n0 + n1) * ngrp;
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const int ch = (int)(it / ngrp), cg = (int)(it - (long)ch * ngrp);
        const long q0 = ch < n0 ? (long)ch * PT_COLS : start1 + (long)(ch - n0) * PT_COLS;
        pool_t_item(Y, ldp, scale, shift, ball_off, ball_cnt, cball, meta, start1, C, seg1_ball, np0, np1, nballs, out, argq, yarg, q0,
                    cg * PT_CH, pt_tile);
        __syncthreads();
    }
}

// dense gradient of the pooled layer: zero the live columns, then one value per (c, ball)
__global__ __launch_bounds__(256) void zero_cols_kernel(float* __restrict__ D, long ldp,
                                                        const int32_t* __restrict__ meta, long start1) {
    const long q = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (q >= ldp) return;
    const int seg = (start1 > 0 && q >= start1) ? 1 : 0;
    if (q - (seg ? start1 : 0) >= meta[4 * seg]) return;
    *reinterpret_cast<float4*>(&D[(long)blockIdx.y * ldp + q]) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// BatchNorm-backward partials of the pooled layer, POOL_BWD_SPLIT rows per segment:
// part[seg*SPLIT + k][0][c] = sum g, part[..][1][c] = sum g*(yarg - mean), g = dOut where out > 0, over the
// k-th share of the segment's balls.  grid (C, nseg, SPLIT).
__global__ __launch_bounds__(256) void pool_bwd_partials_c_kernel(const float* __restrict__ dOut,
                                                                  const float* __restrict__ out,
                                                                  const float* __restrict__ yarg,
                                                                  const float* __restrict__ mean, int C,
                                                                  int nballs, int seg1_ball, int np0, int np1,
                                                                  float* __restrict__ part,
                                                                  const int32_t* __restrict__ argq,
                                                                  float* __restrict__ D, long ldp) {
    // the scatter of the dense gradient D[c, argq] = g rides along -- the same (channel, ball) walk as the statistics (D's
    // live columns were zeroed by the launch in front of this one)
    __shared__ float sh[2][4];
    const int c = blockIdx.x, seg = blockIdx.y, k = blockIdx.z;
    const int s0 = seg ? seg1_ball : 0, s1 = seg ? nballs : seg1_ball;
    const int share = (s1 - s0 + POOL_BWD_SPLIT - 1) / POOL_BWD_SPLIT;
    const int b0 = s0 + k * share, b1 = min(b0 + share, s1);
    const float mu = mean[seg * C + c];
    float s = 0.f, q = 0.f;
    for (int ball = b0 + threadIdx.x; ball < b1; ball += 256) {
        const long o = pool_index(c, ball, C, seg1_ball, np0, np1);
        const bool pos = out[o] > 0.f;
        const float g = pos ? dOut[o] : 0.f;
        if (D && pos) D[(long)c * ldp + argq[o]] = g;
        s += g;
        q += g * (yarg[o] - mu);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { s += __shfl_xor(s, m, 64); q += __shfl_xor(q, m, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long row = (long)seg * POOL_BWD_SPLIT + k;
        part[(row * 2 + 0) * C + c] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[(row * 2 + 1) * C + c] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

// Backward of the pool in ONE pass (round 5): the dense gradient D (C, ldp) of the pooled layer is written column by column --
// D[c, q] = dOut[c, ball(q)] where q is the ball's arg-max column and out > 0, else 0 -- instead of a zero fill of the
// live columns (0.52 GB of zeros per BAT step) followed by a scatter.  Workgroup = 512 columns x 8 channels: the {gradient,
// arg-max column} pairs of the chunk's balls (<= 512: every ball has at least one column) are staged in LDS with loads that
// run along the ball index of the (B, C, npoint) tensors, then every lane writes float4 runs of four channel rows and looks
// its columns' balls up in LDS (round 2's single pass gathered dOut / out / argq from global memory per column: 12 gathers
// per float4 store, 0.41 ms).  The BatchNorm-backward sums ride along: a ball counts in the chunk that holds its first
// column, one partial row {sum g, sum g (yarg - mean)} per chunk (live rows only, like the GEMM epilogues').
constexpr int PBD_COLS = 512, PBD_CH = 8, PBD_BALLS = 128, PBD_GRID = 4096;

struct PoolGrad { const float* p; long sb, sc; };     // a segment's pooled-output gradient (B, C, npoint), strides in floats
                                                       // (innermost 1); p == NULL: no gradient arrived (zeros)

struct PoolBwdDense {
    PoolGrad g0, g1;
    const float* out; const int32_t* argq; const float* yarg; const float* mean; const int32_t* cball; const int32_t* ball_off;
    const int32_t* meta; long start1, ldp; int C, dummy_ball, seg1_ball, np0, np1;
    float* D; float* part;
};

// one work item: the 512-column chunk at q0 x the PBD_CH channels from c0
__device__ __forceinline__ void pool_bwd_dense_item(const PoolBwdDense& a, const long q0, const int c0, float2 (&stg)[PBD_CH][PBD_BALLS],
                                                    int& sh_hi) {
    const PoolGrad g0 = a.g0, g1 = a.g1;
    const float* __restrict__ out = a.out; const int32_t* __restrict__ argq = a.argq; const float* __restrict__ yarg = a.yarg;
    const float* __restrict__ mean = a.mean; const int32_t* __restrict__ cball = a.cball;
    const int32_t* __restrict__ ball_off = a.ball_off; const int32_t* __restrict__ meta = a.meta;
    const long start1 = a.start1, ldp = a.ldp;
    const int C = a.C, dummy_ball = a.dummy_ball, seg1_ball = a.seg1_ball, np0 = a.np0, np1 = a.np1;
    float* __restrict__ D = a.D; float* __restrict__ part = a.part;
    // staged per pass: PBD_BALLS balls (a 512-column chunk of the KITTI-like crops holds ~50, of the k-NN level 128, of
    // full balls 16; more -> more passes), so that 8 workgroups fit a CU: the kernel is a latency chain per workgroup
    // (index loads -> gathers along the balls -> LDS -> stores) and lives on the bytes in flight per CU
    const long chunk = q0 / PBD_COLS;              // the chunk's statistics row
    const int seg = (start1 > 0 && q0 >= start1) ? 1 : 0;
    const long local0 = q0 - (seg ? start1 : 0);
    // every index load of the prologue is issued at once (none depends on another): the live count, the chunk's first
    // ball, this lane's four column -> ball entries (allocated up to ldp; only used when the columns are live)
    const int qi = threadIdx.x & 127, half = threadIdx.x >> 7;
    const long q = q0 + 4 * qi;
    const int live = meta[4 * seg];                  // columns of the segment rounded up to 256
    const int b_lo = cball[q0];
    const int4 cb = *reinterpret_cast<const int4*>(&cball[q]);
    if (threadIdx.x == 0) sh_hi = -1;
    if (local0 >= live) return;                      // (uniform)
    const int ncols = (int)min((long)PBD_COLS, live - local0);     // 256 or 512
    const bool mine = 4 * qi < ncols;
    __syncthreads();
    if (half == 0) {               // (wave-uniform) last real ball of the chunk: padding columns carry the dummy ball, the
        int hi = cb.x;             // largest id, and follow every real column
        hi = cb.y != dummy_ball ? cb.y : hi;
        hi = cb.z != dummy_ball ? cb.z : hi;
        hi = cb.w != dummy_ball ? cb.w : hi;
        if (!mine || cb.x == dummy_ball) hi = -1;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) hi = max(hi, __shfl_xor(hi, m, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&sh_hi, hi);
    }
    __syncthreads();
    const int nb = sh_hi - b_lo + 1;                 // >= 1: a live chunk starts with a real column
    const int cl = threadIdx.x >> 5, bl = threadIdx.x & 31;
    const PoolGrad gs = seg ? g1 : g0;
    const int qq = (int)q;
  {
    const float mu = mean[seg * C + c0 + cl];
    float s = 0.f, sq = 0.f;
    float4 v[PBD_CH / 2];
#pragma unroll
    for (int cc = 0; cc < PBD_CH / 2; ++cc) v[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < nb; base += PBD_BALLS) {
        if (base) __syncthreads();
        {   // ---- stage PBD_BALLS balls: thread = (channel, ball lane); the loads run along the ball index
            const int c = c0 + cl;
            float ov[PBD_BALLS / 32], gv[PBD_BALLS / 32], yv[PBD_BALLS / 32];
            int av[PBD_BALLS / 32], fo[PBD_BALLS / 32];
#pragma unroll
            for (int k = 0; k < PBD_BALLS / 32; ++k) {
                const int i = base + bl + 32 * k;
                const int ball = b_lo + (i < nb ? i : nb - 1);           // clamped, not predicated: loads stay in flight
                const int local = seg ? ball - seg1_ball : ball, np = seg ? np1 : np0;      // (a chunk lies in one segment)
                const int b = local / np, j = local - b * np;
                const long o = (seg ? (long)seg1_ball * C : 0) + ((long)b * C + c) * np + j;     // = pool_index(c, ball, ...)
                ov[k] = out[o]; yv[k] = yarg[o]; av[k] = argq[o]; fo[k] = ball_off[ball];
                gv[k] = gs.p ? gs.p[b * gs.sb + c * gs.sc + j] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < PBD_BALLS / 32; ++k) {
                const int i = base + bl + 32 * k;
                const bool pos = ov[k] > 0.f;
                const float g = pos ? gv[k] : 0.f;
                stg[cl][bl + 32 * k] = make_float2(g, __int_as_float(pos ? av[k] : -1));
                if (i < nb && fo[k] >= q0) { s += g; sq += g * (yv[k] - mu); }   // the ball's first column lies in this chunk
            }
        }
        __syncthreads();
        if (mine) {   // ---- this lane's four columns of four channel rows: look the balls up
            const int i0 = cb.x - b_lo - base, i1 = cb.y - b_lo - base, i2 = cb.z - b_lo - base, i3 = cb.w - b_lo - base;
            const int lim = min(PBD_BALLS, nb - base);          // (padding columns: dummy ball -> beyond nb)
            // branch-free: clamped indices, the arg-max test carries the validity (a staged arg is a real column of ITS ball,
            // so a clamped look-up of another ball can never equal this lane's column)
            const int j0 = min(max(i0, 0), lim - 1), j1 = min(max(i1, 0), lim - 1);
            const int j2 = min(max(i2, 0), lim - 1), j3 = min(max(i3, 0), lim - 1);
#pragma unroll
            for (int cc = 0; cc < PBD_CH / 2; ++cc) {
                const int ch = half * (PBD_CH / 2) + cc;
                const float2 e0 = stg[ch][j0], e1 = stg[ch][j1], e2 = stg[ch][j2], e3 = stg[ch][j3];
                v[cc].x = __float_as_int(e0.y) == qq + 0 ? e0.x : v[cc].x;
                v[cc].y = __float_as_int(e1.y) == qq + 1 ? e1.x : v[cc].y;
                v[cc].z = __float_as_int(e2.y) == qq + 2 ? e2.x : v[cc].z;
                v[cc].w = __float_as_int(e3.y) == qq + 3 ? e3.x : v[cc].w;
            }
        }
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) { s += __shfl_xor(s, m, 64); sq += __shfl_xor(sq, m, 64); }
    if (bl == 0) {
        part[(chunk * 2 + 0) * C + c0 + cl] = s;
        part[(chunk * 2 + 1) * C + c0 + cl] = sq;
    }
    if (mine) {
#pragma unroll
        for (int cc = 0; cc < PBD_CH / 2; ++cc)
            *reinterpret_cast<float4*>(&D[(long)(c0 + half * (PBD_CH / 2) + cc) * ldp + q]) = v[cc];
    }
  }
}

// PERSISTENT grid (round 5): two thirds of a worst-case grid over (chunks x channel groups) are workgroups whose chunk is dead on
// real crops -- up to 49 000 per launch here, and a dead workgroup still costs its dispatch (section 7d of HISTORY.md).  A fixed
// number of workgroups walks the LIVE items instead (counts from the device-side meta), channel group fastest so that
// neighbouring workgroups share a chunk's index lines.
__global__ __launch_bounds__(256) void pool_bwd_dense_kernel(PoolBwdDense a) {
    __shared__ float2 stg[PBD_CH][PBD_BALLS];       // {g, bits(arg-max column) | -1}
    __shared__ int sh_hi;
    const int ngrp = a.C / PBD_CH;
    const int n0 = (a.meta[0] + PBD_COLS - 1) / PBD_COLS;
    const int n1 = a.start1 > 0 ? (a.meta[4] + PBD_COLS - 1) / PBD_COLS : 0;
    const long items = (long)(n0 + n1) * ngrp;
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const int ch = (int)(it / ngrp), cg = (int)(it - (long)ch * ngrp);
        const long q0 = ch < n0 ? (long)ch * PBD_COLS : a.start1 + (long)(ch - n0) * PBD_COLS;
        pool_bwd_dense_item(a, q0, cg * PBD_CH, stg, sh_hi);
        __syncthreads();                                // the item's look-ups are done with `stg` / `sh_hi`
    }
}

// ---------------------------------------------------------------------------------------
// layer-0 backward reduce: dY = A1*dN + w*(A2*Y0 + A3);  S[c, b*ld+n] = sum of dY over the columns that
// reference point n, T[c, ball] = sum of dY over the ball.  Workgroup = (cloud, CS channels), LDS fp32
// atomics (no duplicates inside a ball any more, so no same-address pile-ups).
// ---------------------------------------------------------------------------------------
struct SegParams { int npoint, ld, pt_base, ball_base; };   // per segment: balls / points per cloud, first point
                                                             // column and first ball of the segment

template <int CS, int BT>
__global__ __launch_bounds__(BT) void reduce_c_kernel(const float* __restrict__ dN, const float* __restrict__ Y0,
                                                       long ldp, const float* __restrict__ A1,
                                                       const float* __restrict__ A2,
                                                       const float* __restrict__ A3,
                                                       const int32_t* __restrict__ gp,
                                                       const int32_t* __restrict__ cball,
                                                       const float* __restrict__ cw,
                                                       const int32_t* __restrict__ ball_off,
                                                       const int32_t* __restrict__ ball_cnt, int B, SegParams sp0,
                                                       SegParams sp1, int C0, long lds_row,
                                                       float* __restrict__ S, float* __restrict__ T, int nballs) {
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [CS][ld] then [CS][npoint]
    const int slabs = (C0 + CS - 1) / CS;
    const int cloud = blockIdx.x / slabs, c0 = (blockIdx.x - cloud * slabs) * CS;
    const int seg = cloud >= B ? 1 : 0, b = cloud - seg * B;
    const SegParams sp = seg ? sp1 : sp0;
    const int npoint = sp.npoint, ld = sp.ld;
    const int pbase = sp.pt_base + b * ld, bbase = sp.ball_base + b * npoint;
    float* tacc = acc + CS * ld;
    for (int i = threadIdx.x; i < CS * (ld + npoint); i += BT) acc[i] = 0.f;
    __syncthreads();
    const int q0 = ball_off[bbase], q1 = ball_off[bbase + npoint - 1] + ball_cnt[bbase + npoint - 1];
    float a1[CS], a2[CS], a3[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        const int cc = seg * C0 + (c0 + c < C0 ? c0 + c : C0 - 1);     // segment 1: its own BN-backward constants
        a1[c] = A1[cc]; a2[c] = A2[cc]; a3[c] = A3[cc];
    }
    // lane = 4 consecutive columns (one dwordx4 per operand row and per metadata array): 1024 columns per pass,
    // all 2*CS row loads in flight before the first LDS atomic.  A ball's columns are consecutive, so a lane first
    // folds its own columns of the same ball into one term for the ball sum T (the ~8 columns of a ball piling onto
    // one LDS address was the slow part: ds_add_f32 retires roughly one lane per 8 cycles per CU on a shared address).
    for (int base = q0 & ~3; base < q1; base += 4 * BT) {
        const int q = base + 4 * threadIdx.x;
        const bool any = q < q1;
        int4 g4 = make_int4(0, 0, 0, 0), b4 = make_int4(0, 0, 0, 0);
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (any) {
            g4 = *reinterpret_cast<const int4*>(&gp[q]);
            b4 = *reinterpret_cast<const int4*>(&cball[q]);
            w4 = *reinterpret_cast<const float4*>(&cw[q]);
        }
        const int gq[4] = {g4.x, g4.y, g4.z, g4.w}, bq[4] = {b4.x, b4.y, b4.z, b4.w};
        const float wq[4] = {w4.x, w4.y, w4.z, w4.w};
        bool lv[4];
        int n[4], j[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            lv[t] = q + t >= q0 && q + t < q1;
            n[t] = lv[t] ? gq[t] - pbase : 0;
            j[t] = lv[t] ? bq[t] - bbase : -1;
        }
        float dy[CS][4];
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f), y = d;
            if (any && c0 + c < C0) {
                const long o = (long)(c0 + c) * ldp + q;
                d = *reinterpret_cast<const float4*>(&dN[o]);
                y = *reinterpret_cast<const float4*>(&Y0[o]);
            }
            const float dv[4] = {d.x, d.y, d.z, d.w}, yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) dy[c][t] = lv[t] ? fmaf(a1[c], dv[t], wq[t] * fmaf(a2[c], yv[t], a3[c])) : 0.f;
        }
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            if (c0 + c >= C0) break;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (lv[t]) atomicAdd(&acc[c * ld + n[t]], dy[c][t]);
            if (T) {
                float run = dy[c][0];
                int jr = j[0];
#pragma unroll
                for (int t = 1; t < 4; ++t) {
                    if (j[t] == jr) {
                        run += dy[c][t];
                    } else {
                        if (jr >= 0) atomicAdd(&tacc[c * npoint + jr], run);
                        jr = j[t];
                        run = dy[c][t];
                    }
                }
                if (jr >= 0) atomicAdd(&tacc[c * npoint + jr], run);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CS * ld; i += BT) {
        const int c = i / ld, n = i - c * ld;
        if (c0 + c < C0) S[(long)(c0 + c) * lds_row + pbase + n] = acc[i];
    }
    if (T)
        for (int i = threadIdx.x; i < CS * npoint; i += BT) {
            const int c = i / npoint, j = i - c * npoint;
            if (c0 + c < C0) T[(long)(c0 + c) * nballs + bbase + j] = tacc[i];
        }
}

// ---------------------------------------------------------------------------------------
// The same reduce WITHOUT float atomics (tools/exp/reduce_gather_probe.hip: 138.6 vs 297.7 us on the SA1 shape).
//   csr_build_kernel   per cloud, once per SA call: the cloud's columns sorted by (column chunk, point, column) in
//                      `perm` (counting sort in LDS, lists sorted -> fixed summation order) and the list starts
//                      poff[cloud][chunk*ld + n] (relative to the cloud's first column), total at [nchunk*ld].
//   reduce_gather_kernel  per (cloud, CS channels): the chunk's dY staged in LDS with plain stores; thread n
//                      gathers its own list, thread j sums its ball's contiguous range.
// The default since round 2; reduce_c_kernel (one LDS atomic per column) remains for clouds whose index does not fit.
// ---------------------------------------------------------------------------------------
constexpr int RG_CH = 2048;     // columns per chunk (16-bit list entries: <= 65536)
// channels per workgroup: 2 (4 re-read the chunk's index half as often but double the LDS footprint: measured on the
// MI355X, same run A/B, 0.55 vs 0.50 ms per step)
static int rg_cs() { return 2; }

__global__ __launch_bounds__(1024) void csr_build_kernel(const int32_t* __restrict__ gp,
                                                         const int32_t* __restrict__ ball_off,
                                                         const int32_t* __restrict__ ball_cnt, int B, SegParams sp0,
                                                         SegParams sp1, int nchunk, int poff_stride,
                                                         int32_t* __restrict__ perm, int32_t* __restrict__ poff) {
    extern __shared__ int csr_sh[];         // cnt[nchunk*ld + 1], scan scratch [1024]
    const int cloud = blockIdx.x;
    const int seg = cloud >= B ? 1 : 0, b = cloud - seg * B;
    const SegParams sp = seg ? sp1 : sp0;
    const int npoint = sp.npoint, ld = sp.ld;
    const int pbase = sp.pt_base + b * ld, bbase = sp.ball_base + b * npoint;
    const int q0 = ball_off[bbase], q1 = ball_off[bbase + npoint - 1] + ball_cnt[bbase + npoint - 1];
    const int q0a = q0 & ~3;                // chunk 0 starts at the float4-aligned column at or before q0
    const int nbin = nchunk * ld;
    int* cnt = csr_sh;
    int* part = csr_sh + nbin + 1;
    for (int i = threadIdx.x; i <= nbin; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int q = q0 + threadIdx.x; q < q1; q += 1024) atomicAdd(&cnt[((q - q0a) / RG_CH) * ld + (gp[q] - pbase)], 1);
    __syncthreads();
    const int per = (nbin + 1023) / 1024, i0 = threadIdx.x * per;       // exclusive scan of the bin counts
    int local = 0;
    for (int i = i0; i < i0 + per && i < nbin; ++i) local += cnt[i];
    part[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - local;
    for (int i = i0; i < i0 + per && i < nbin; ++i) {
        const int c = cnt[i];
        cnt[i] = run;
        run += c;
    }
    if (threadIdx.x == 1023) cnt[nbin] = part[1023];
    __syncthreads();
    int32_t* po = poff + (long)cloud * poff_stride;
    for (int i = threadIdx.x; i <= nbin; i += 1024) po[i] = cnt[i];
    __syncthreads();
    // fill: entry = (point << 16) | (column - chunk start); the order inside a list is whatever the LDS atomics
    // hand out (all entries of a list belong to the same point, so only the summation order depends on it)
    for (int q = q0 + threadIdx.x; q < q1; q += 1024) {
        const int ch = (q - q0a) / RG_CH, n = gp[q] - pbase;
        const int k = atomicAdd(&cnt[ch * ld + n], 1);
        perm[q0 + k] = (n << 16) | (q - q0a - ch * RG_CH);
    }
}

template <int CS>
__global__ __launch_bounds__(256) void reduce_gather_kernel(const float* __restrict__ dN, const float* __restrict__ Y0,
                                                            long ldp, const float* __restrict__ A1,
                                                            const float* __restrict__ A2, const float* __restrict__ A3,
                                                            const float* __restrict__ cw,
                                                            const int32_t* __restrict__ ball_off,
                                                            const int32_t* __restrict__ ball_cnt,
                                                            const int32_t* __restrict__ perm,
                                                            const int32_t* __restrict__ poff, int poff_stride, int B,
                                                            SegParams sp0, SegParams sp1, int C0, long lds_row,
                                                            float* __restrict__ S, float* __restrict__ T, int nballs) {
    extern __shared__ float rg_sm[];        // dy[CS][CH], sacc[CS][ld], tacc[CS][npoint], list[CH] packed (point, column)
    constexpr int CH = RG_CH;
    const int slabs = (C0 + CS - 1) / CS;
    const int cloud = blockIdx.x / slabs, c0 = (blockIdx.x - cloud * slabs) * CS;
    const int seg = cloud >= B ? 1 : 0, b = cloud - seg * B;
    const SegParams sp = seg ? sp1 : sp0;
    const int npoint = sp.npoint, ld = sp.ld;
    const int pbase = sp.pt_base + b * ld, bbase = sp.ball_base + b * npoint;
    float* dy = rg_sm;
    float* sacc = dy + CS * CH;
    float* tacc = sacc + CS * ld;
    int* lst = reinterpret_cast<int*>(tacc + CS * npoint);
    const int q0 = ball_off[bbase], q1 = ball_off[bbase + npoint - 1] + ball_cnt[bbase + npoint - 1];
    const int q0a = q0 & ~3;
    const int32_t* po = poff + (long)cloud * poff_stride;
    for (int i = threadIdx.x; i < CS * (ld + npoint); i += 256) sacc[i] = 0.f;
    float a1[CS], a2[CS], a3[CS];
    int crow[CS];                            // channel rows (clamped: an odd C0 repeats its last row, never stored)
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        crow[c] = c0 + c < C0 ? c0 + c : C0 - 1;
        const int cc = seg * C0 + crow[c];   // segment 1: its own BatchNorm-backward constants
        a1[c] = A1[cc]; a2[c] = A2[cc]; a3[c] = A3[cc];
    }
    for (int k = 0, lo = q0a; lo < q1; ++k, lo += CH) {
        const int hi = lo + CH < q1 ? lo + CH : q1;
        // phase A: every global load of the chunk in flight before the first wait -- the chunk's slice of the sorted
        // index (<= CH entries: 8 per thread, clamped) and the dN / Y0 rows of both column quads of the thread --
        // then dY and the index into LDS with plain stores
        const int l0 = po[k * ld], l1 = po[(k + 1) * ld];         // this chunk's slice of perm (sorted by point)
        const int cnt = l1 - l0;
        int ent[CH / 256];
#pragma unroll
        for (int u = 0; u < CH / 256; ++u) {
            const int i = threadIdx.x + 256 * u;
            ent[u] = perm[cnt > 0 ? q0 + l0 + (i < cnt ? i : cnt - 1) : 0];
        }
        float4 d[CH / 1024][CS], y[CH / 1024][CS], w4[CH / 1024];
#pragma unroll
        for (int it = 0; it < CH / 1024; ++it) {
            const int q = lo + 4 * threadIdx.x + 1024 * it;
            const int qc = q < hi ? q : lo;                       // clamped, not predicated (lo < hi always)
            w4[it] = *reinterpret_cast<const float4*>(&cw[qc]);
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                d[it][c] = *reinterpret_cast<const float4*>(&dN[(long)crow[c] * ldp + qc]);
                y[it][c] = *reinterpret_cast<const float4*>(&Y0[(long)crow[c] * ldp + qc]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // (keeps the machine scheduler from pulling the first quad's arithmetic, and
                                            // with it a full wait, in front of the remaining loads)
        __syncthreads();                    // previous chunk fully consumed (first pass: the zero fill)
#pragma unroll
        for (int it = 0; it < CH / 1024; ++it) {
            // UNCONDITIONAL stores (columns at or beyond `hi` get the clamped column's values: no index entry and no
            // ball range refers to them): under an `if` the compiler sinks the loads into the branch, behind the barrier
            const int i = 4 * threadIdx.x + 1024 * it;
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                float4 o;
                o.x = fmaf(a1[c], d[it][c].x, w4[it].x * fmaf(a2[c], y[it][c].x, a3[c]));
                o.y = fmaf(a1[c], d[it][c].y, w4[it].y * fmaf(a2[c], y[it][c].y, a3[c]));
                o.z = fmaf(a1[c], d[it][c].z, w4[it].z * fmaf(a2[c], y[it][c].z, a3[c]));
                o.w = fmaf(a1[c], d[it][c].w, w4[it].w * fmaf(a2[c], y[it][c].w, a3[c]));
                *reinterpret_cast<float4*>(&dy[c * CH + i]) = o;
            }
        }
#pragma unroll
        for (int u = 0; u < CH / 256; ++u) lst[threadIdx.x + 256 * u] = ent[u];      // (entries >= cnt: never read)
        __syncthreads();
        // phase B, balanced: every thread takes an equal share of the chunk's entries (a point referenced by hundreds
        // of balls no longer serialises on one thread) and flushes a run of equal points with one LDS atomic
        {
            const int per = (cnt + 255) / 256;
            const int e0 = threadIdx.x * per, e1 = e0 + per < cnt ? e0 + per : cnt;
            int cur = -1;
            float sr[CS];
#pragma unroll
            for (int c = 0; c < CS; ++c) sr[c] = 0.f;
            for (int a = e0; a < e1; ++a) {
                const int e = lst[a];
                const int n = e >> 16, i = e & 0xffff;
                if (n != cur) {
                    if (cur >= 0) {
#pragma unroll
                        for (int c = 0; c < CS; ++c) atomicAdd(&sacc[c * ld + cur], sr[c]);
                    }
                    cur = n;
#pragma unroll
                    for (int c = 0; c < CS; ++c) sr[c] = 0.f;
                }
#pragma unroll
                for (int c = 0; c < CS; ++c) sr[c] += dy[c * CH + i];
            }
            if (cur >= 0) {
#pragma unroll
                for (int c = 0; c < CS; ++c) atomicAdd(&sacc[c * ld + cur], sr[c]);
            }
        }
        if (T)
            for (int j = threadIdx.x; j < npoint; j += 256) {     // thread j: its ball's part of the chunk
                const int b0 = ball_off[bbase + j], b1 = b0 + ball_cnt[bbase + j];
                const int s0 = b0 > lo ? b0 : lo, s1 = b1 < hi ? b1 : hi;
                if (s0 >= s1) continue;
                float t[CS];
#pragma unroll
                for (int c = 0; c < CS; ++c) t[c] = 0.f;
                for (int q = s0; q < s1; ++q)
#pragma unroll
                    for (int c = 0; c < CS; ++c) t[c] += dy[c * CH + (q - lo)];
#pragma unroll
                for (int c = 0; c < CS; ++c) tacc[c * npoint + j] += t[c];
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CS * ld; i += 256) {
        const int c = i / ld, n = i - c * ld;
        if (c0 + c < C0) S[(long)(c0 + c) * lds_row + pbase + n] = sacc[i];
    }
    if (T)
        for (int i = threadIdx.x; i < CS * npoint; i += 256) {
            const int c = i / npoint, j = i - c * npoint;
            if (c0 + c < C0) T[(long)(c0 + c) * nballs + bbase + j] = tacc[i];
        }
}

// dW[c, 0..2] -= sum over balls of T[c, ball] * centers[ball, :]: the centre term of
// grouped_xyz = xyz[idx] - new_xyz (pointnet2_utils.py:322-324) in the layer-0 weight gradient.
// One 1024-thread workgroup per channel, fixed summation order.
// out != NULL (round 3): the finished row goes to the COMPACT (C0, ncols) gradient instead -- out[c, k] = dW[c, k] (- the
// centre term for k < 3) -- which was a strided torch copy of dW[:, :ncols] behind this launch.
__global__ __launch_bounds__(1024) void center_term_kernel(const float* __restrict__ T,
                                                           const float* __restrict__ centers, int nballs, int ldw,
                                                           float* __restrict__ dW, float* __restrict__ out = nullptr,
                                                           int ncols = 0) {
    __shared__ float red[16][3];
    const int c = blockIdx.x, tid = threadIdx.x;
    const float* t = T + (long)c * nballs;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int ball = tid; ball < nballs; ball += 1024) {
        const float v = t[ball];
        s0 = fmaf(v, centers[3 * ball], s0);
        s1 = fmaf(v, centers[3 * ball + 1], s1);
        s2 = fmaf(v, centers[3 * ball + 2], s2);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s0 += __shfl_xor(s0, off, 64);
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
    }
    if ((tid & 63) == 0) { red[tid >> 6][0] = s0; red[tid >> 6][1] = s1; red[tid >> 6][2] = s2; }
    __syncthreads();
    if (tid < 3) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) a += red[w][tid];
        if (out) out[(long)c * ncols + tid] = dW[(long)c * ldw + tid] - a;
        else dW[(long)c * ldw + tid] -= a;
    }
    if (out)
        for (int k = 3 + tid; k < ncols; k += 1024) out[(long)c * ncols + k] = dW[(long)c * ldw + k];
}

// Per-point operand of layer 0: X0 (rows, ldz) = [xyz * inv_radius ; feats ; zero rows] with every cloud's N points
// padded to ld columns by zeros, for one or two segments side by side (segment 1's columns start at B*ld0).
// Replaces a zero fill and up to four strided torch copies per SA call.
struct PackSeg { const float* xyz; const float* feats; int N, ld; };

__global__ __launch_bounds__(256) void pack_points_kernel(PackSeg s0, PackSeg s1, int B, int nxyz, int C, float inv_radius,
                                                          long ldz, float* __restrict__ X0) {
    const long col = (long)blockIdx.x * 256 + threadIdx.x;
    if (col >= ldz) return;
    const int r = blockIdx.y;
    const long base1 = (long)B * s0.ld;
    const bool second = col >= base1;
    const PackSeg sg = second ? s1 : s0;
    const long local = second ? col - base1 : col;
    const int b = (int)(local / sg.ld), n = (int)(local - (long)b * sg.ld);
    float v = 0.f;
    if (n < sg.N) {
        if (r < nxyz) v = sg.xyz[((long)b * sg.N + n) * 3 + r] * inv_radius;
        else if (r < nxyz + C) v = sg.feats[((long)b * C + (r - nxyz)) * sg.N + n];
    }
    X0[(long)r * ldz + col] = v;
}

}  // namespace

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
// balls per segment: bounded only by 32-bit column arithmetic (balls * nsample slots); BAT_CAR_NUSCENES at its YAML batch
// (100 pairs x 1024 balls of the 2048-point search cloud, cfgs/BAT_CAR_NUSCENES.yaml:11,56) has 102 400
constexpr long O3D_MAX_BALLS = 1L << 20;

// One segment of the compact layout: idx (B,npoint,ns) -> ball_cnt (B*npoint), ball_off (B*npoint+1, absolute
// columns), and per column gp / cball / cw written at [col_base, col_base + live); meta (4 ints) = {live rounded
// up to 256, live, B*npoint, 0}.  pt_base / ball_base = first point column / first ball id of the segment,
// dummy_ball = ball id of the padding columns (a zero row of the centre table).  ld = point columns per cloud.
extern "C" int o3d_compact_build(const int32_t* idx, int B, int npoint, int ns, int ld, int col_base, int pt_base,
                                 int ball_base, int dummy_ball, int32_t* ball_cnt, int32_t* ball_off, int32_t* gp,
                                 int32_t* cball, float* cw, int32_t* meta, void* stream) {
    const long nballs = (long)B * npoint, nslots = nballs * ns;
    if (!idx || !ball_cnt || !ball_off || !gp || !cball || !cw || !meta || B <= 0 || npoint <= 0 || ns < 1 ||
        ns > 64 || !pow2(ns) || nballs > O3D_MAX_BALLS || nslots % 256 != 0 || ld <= 0 || col_base < 0 || col_base % 256 != 0)
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    const int blocks = (int)(nslots / 256);
    hipLaunchKernelGGL(compact_count_kernel, dim3(blocks), dim3(256), 0, s, idx, nslots, ns, ball_cnt);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, ball_cnt, (int)nballs, col_base, ball_off, meta);
    hipLaunchKernelGGL(compact_fill_kernel, dim3(blocks + 1), dim3(256), 0, s, idx, nslots, ns, npoint, ld, col_base,
                       pt_base, ball_base, dummy_ball, ball_cnt, ball_off, meta, gp, cball, cw);
    return o3d_launch_status();
}

// ---------------------------------------------------------------------------------------
// Weight gradient of a layer 0 whose input is xyz ONLY and whose input gradient nobody needs (SA level 0 of the backbone,
// models/backbone/pointnet.py:30-38: features None, the input cloud carries no gradient):
//     dW0[c,k] = sum_q dY0[c,q] * (xyz[gp[q],k] - centre[cball[q],k]),   dY0 = A1*dN + w*(A2*Y0 + A3)
// straight from the columns -- one streaming pass over dN / Y0 -- instead of the per-point list sums S (LDS atomics,
// reduce_gather_kernel), the per-ball sums T, a K = 3 GEMM S.xyz^T on the generic kernel and the centre term.
//   workgroup = 256 columns (lane = 4 consecutive columns, float4 rows), wave w takes channels w, w+4, ...; per lane
//   3 accumulators per channel, folded over the wave at the end; partial rows [block][C0][3] (dead tiles write zeros),
//   summed in a fixed order by dw0_reduce_kernel.
// ---------------------------------------------------------------------------------------
namespace {
template <int CPW>        // channels per wave (C0 = 4 * CPW per pass)
__global__ __launch_bounds__(256) void dw0_xyz_kernel(const float* __restrict__ dN, const float* __restrict__ Y0, long ldp,
                                                      const float* __restrict__ A1, const float* __restrict__ A2,
                                                      const float* __restrict__ A3, const int32_t* __restrict__ gp,
                                                      const int32_t* __restrict__ cball, const float* __restrict__ cw,
                                                      const float* __restrict__ X, long ldz,
                                                      const float* __restrict__ centers,
                                                      const int32_t* __restrict__ meta, long start1, int C0,
                                                      float* __restrict__ part) {
    const long q0 = (long)blockIdx.x * 256;
    const int seg = (start1 > 0 && q0 >= start1) ? 1 : 0;
    float* prow = part + (long)blockIdx.x * C0 * 3;
    if (q0 - (seg ? start1 : 0) >= meta[4 * seg]) {       // dead tile: a zero row (the reduction sums every row)
        for (int i = threadIdx.x; i < C0 * 3; i += 256) prow[i] = 0.f;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long q = q0 + 4 * lane;
    const int4 id = *reinterpret_cast<const int4*>(&gp[q]);
    const int4 bj = *reinterpret_cast<const int4*>(&cball[q]);
    const float4 w = *reinterpret_cast<const float4*>(&cw[q]);
    const int ids[4] = {id.x, id.y, id.z, id.w}, bb[4] = {bj.x, bj.y, bj.z, bj.w};
    const float wv[4] = {w.x, w.y, w.z, w.w};
    float rel[4][3];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 3; ++k)      // (padding columns of the last live tile carry weight 0: they contribute nothing)
            rel[t][k] = wv[t] != 0.f ? X[(long)k * ldz + ids[t]] - centers[(long)bb[t] * 3 + k] : 0.f;
    const float* a1 = A1 + seg * C0;
    const float* a2 = A2 + seg * C0;
    const float* a3 = A3 + seg * C0;
    for (int cbase = 0; cbase < C0; cbase += 4 * CPW) {
        float acc[CPW][3];
        float4 d[CPW], y[CPW];
#pragma unroll
        for (int i = 0; i < CPW; ++i) {            // all rows of the pass in flight first
            const int c = cbase + wave + 4 * i;
            const int cc = c < C0 ? c : C0 - 1;
            d[i] = *reinterpret_cast<const float4*>(&dN[(long)cc * ldp + q]);
            y[i] = *reinterpret_cast<const float4*>(&Y0[(long)cc * ldp + q]);
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int c = cbase + wave + 4 * i;
            const int cc = c < C0 ? c : C0 - 1;
            const float k1 = a1[cc], k2 = a2[cc], k3 = a3[cc];
            const float dv[4] = {d[i].x, d[i].y, d[i].z, d[i].w}, yv[4] = {y[i].x, y[i].y, y[i].z, y[i].w};
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float dy = wv[t] != 0.f ? fmaf(k1, dv[t], wv[t] * fmaf(k2, yv[t], k3)) : 0.f;
                s0 = fmaf(dy, rel[t][0], s0); s1 = fmaf(dy, rel[t][1], s1); s2 = fmaf(dy, rel[t][2], s2);
            }
            acc[i][0] = s0; acc[i][1] = s1; acc[i][2] = s2;
        }
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int c = cbase + wave + 4 * i;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float t = wave_sum(acc[i][k]);
                if (lane == 0 && c < C0) prow[c * 3 + k] = t;
            }
        }
    }
}

// out[i] = sum over the nblk partial rows of part[r][i], i < n: one workgroup per output, rows strided over the threads,
// wave sums + 4 LDS entries -- a fixed order
__global__ __launch_bounds__(256) void dw0_reduce_kernel(const float* __restrict__ part, int nblk, int n,
                                                         float* __restrict__ out) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < nblk; r += 256) s += part[(long)r * n + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[i] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
}  // namespace

// dW0 (C0, 3) of an xyz-only layer 0 straight from the columns (see dw0_xyz_kernel); X = the packed per-point operand
// (rows 0..2 = the scaled coordinates, ldz columns), centers ((nballs+1), 3) scaled alike, A1..A3 (nseg, C0);
// part: scratch of (ldp/256) * C0 * 3 floats.
extern "C" int o3d_group_dw0_xyz(const float* dN, const float* Y0, long ldp, const float* A1, const float* A2,
                                 const float* A3, const int32_t* gp, const int32_t* cball, const float* cw,
                                 const float* X, long ldz, const float* centers, const int32_t* meta, long start1, int C0,
                                 float* part, float* dW, void* stream) {
    if (!dN || !Y0 || !A1 || !A2 || !A3 || !gp || !cball || !cw || !X || !centers || !meta || !part || !dW || C0 <= 0 ||
        ldp <= 0 || ldp % 256 != 0 || ldz <= 0 || start1 < 0 || start1 % 256 != 0)
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    const int nblk = (int)(ldp / 256);
    hipLaunchKernelGGL(dw0_xyz_kernel<8>, dim3(nblk), dim3(256), 0, s, dN, Y0, ldp, A1, A2, A3, gp, cball, cw, X, ldz, centers,
                       meta, start1, C0, part);
    hipLaunchKernelGGL(dw0_reduce_kernel, dim3(C0 * 3), dim3(256), 0, s, part, nblk, C0 * 3, dW);
    return o3d_launch_status();
}

// Both segments of a paired call (template + search cloud through one shared module) in THREE launches instead of six:
// every kernel covers segment 0's workgroups first, then segment 1's.  Arguments per segment as o3d_compact_build.
namespace {
struct CompactSeg {
    const int32_t* idx; long nslots; int npoint, ld, col_base, pt_base, ball_base, nballs;
    int32_t* ball_cnt; int32_t* ball_off; int32_t* meta;
};

__global__ __launch_bounds__(256) void compact_count2_kernel(CompactSeg s0, CompactSeg s1, int ns, int blocks0) {
    const bool second = (int)blockIdx.x >= blocks0;
    const CompactSeg& sg = second ? s1 : s0;
    const long s = (long)(blockIdx.x - (second ? blocks0 : 0)) * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int v = s < sg.nslots ? sg.idx[s] : -1;
    const int base = lane & ~(ns - 1), k = lane - base;
    const int first = __shfl(v, base, 64);
    const unsigned long long m = __ballot(v != first);
    const unsigned long long seg = ns == 64 ? ~0ull : (((1ull << ns) - 1) << base);
    const unsigned long long other = m & seg;
    if (k == 0 && s < sg.nslots) sg.ball_cnt[s / ns] = other ? (63 - __clzll((long long)other)) - base + 1 : 1;
}

__global__ __launch_bounds__(1024) void compact_scan2_kernel(CompactSeg s0, CompactSeg s1) {
    const CompactSeg& sg = blockIdx.x ? s1 : s0;
    const int n = sg.nballs;
    __shared__ int sh[1024];
    const int per = (n + 1023) / 1024;
    const int i0 = threadIdx.x * per;
    int local = 0;
    for (int i = i0; i < i0 + per && i < n; ++i) local += sg.ball_cnt[i];
    sh[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    int run = sg.col_base + sh[threadIdx.x] - local;
    for (int i = i0; i < i0 + per && i < n; ++i) {
        sg.ball_off[i] = run;
        run += sg.ball_cnt[i];
    }
    if (threadIdx.x == 1023) {
        const int tot = sh[1023];
        sg.meta[0] = (tot + 255) & ~255;
        sg.meta[1] = tot;
        sg.meta[2] = n;
        sg.meta[3] = 0;
    }
}

__global__ __launch_bounds__(256) void compact_fill2_kernel(CompactSeg s0, CompactSeg s1, int ns, int blocks0, int dummy_ball,
                                                            int32_t* __restrict__ gp, int32_t* __restrict__ cball,
                                                            float* __restrict__ cw) {
    const bool second = (int)blockIdx.x >= blocks0 + 1;
    const CompactSeg& sg = second ? s1 : s0;
    const int blk = blockIdx.x - (second ? blocks0 + 1 : 0);
    const int nblk = (int)(sg.nslots / 256);
    if (blk == nblk) {       // padding columns [Ptot, Ptot_pad) of this segment
        const int q = sg.meta[1] + threadIdx.x;
        if (q < sg.meta[0]) { gp[sg.col_base + q] = 0; cball[sg.col_base + q] = dummy_ball; cw[sg.col_base + q] = 0.f; }
        return;
    }
    const long s = (long)blk * 256 + threadIdx.x;
    if (s >= sg.nslots) return;
    const int ball = (int)(s / ns), k = (int)(s - (long)ball * ns);
    const int cnt = sg.ball_cnt[ball];
    if (k >= cnt) return;
    const int q = sg.ball_off[ball] + k;                  // ball_off already includes col_base
    gp[q] = sg.pt_base + (ball / sg.npoint) * sg.ld + sg.idx[s];
    cball[q] = sg.ball_base + ball;
    cw[q] = k == 0 ? (float)(1 + ns - cnt) : 1.f;
}
}  // namespace

extern "C" int o3d_compact_build2(const int32_t* idx0, int npoint0, int ld0, const int32_t* idx1, int npoint1, int ld1,
                                  int B, int ns, int col_base1, int pt_base1, int dummy_ball, int32_t* ball_cnt,
                                  int32_t* ball_off, int32_t* gp, int32_t* cball, float* cw, int32_t* meta, void* stream) {
    const long nb0 = (long)B * npoint0, nb1 = (long)B * npoint1;
    if (!idx0 || !idx1 || !ball_cnt || !ball_off || !gp || !cball || !cw || !meta || B <= 0 || npoint0 <= 0 ||
        npoint1 <= 0 || ns < 1 || ns > 64 || !pow2(ns) || nb0 > O3D_MAX_BALLS || nb1 > O3D_MAX_BALLS || (nb0 * ns) % 256 != 0 ||
        (nb1 * ns) % 256 != 0 || ld0 <= 0 || ld1 <= 0 || col_base1 < 0 || col_base1 % 256 != 0 || pt_base1 < 0)
        return O3D_EINVAL;
    CompactSeg s0 = {idx0, nb0 * ns, npoint0, ld0, 0, 0, 0, (int)nb0, ball_cnt, ball_off, meta};
    CompactSeg s1 = {idx1, nb1 * ns, npoint1, ld1, col_base1, pt_base1, (int)nb0, (int)nb1, ball_cnt + nb0, ball_off + nb0,
                     meta + 4};
    hipStream_t s = o3d_stream(stream);
    const int b0 = (int)(s0.nslots / 256), b1 = (int)(s1.nslots / 256);
    hipLaunchKernelGGL(compact_count2_kernel, dim3(b0 + b1), dim3(256), 0, s, s0, s1, ns, b0);
    hipLaunchKernelGGL(compact_scan2_kernel, dim3(2), dim3(1024), 0, s, s0, s1);
    hipLaunchKernelGGL(compact_fill2_kernel, dim3(b0 + b1 + 2), dim3(256), 0, s, s0, s1, ns, b0, dummy_ball, gp, cball, cw);
    return o3d_launch_status();
}

// Y0 (C0, ldp) from Z (C0, ldz); centers ((nballs+1), 3) or NULL; part [ldp/256][2][C0] or NULL
extern "C" int o3d_group_expand_c(const float* Z, long ldz, const int32_t* gp, const int32_t* cball,
                                  const float* cw, const float* centers, const float* W0, int ldw, int C0,
                                  const int32_t* meta, long start1, long ldp, float* Y0, float* part,
                                  const float* stat_c, void* stream) {
    if (!Z || !gp || !cball || !cw || !meta || !Y0 || C0 <= 0 || ldp <= 0 || ldp % 256 != 0 || start1 < 0 ||
        start1 % 256 != 0 || (centers && (!W0 || ldw < 3)))
        return O3D_EINVAL;
    // channel ranges per column tile: measured on the MI355X (same run A/B, ms per step of this kernel) 1 range
    // 0.26, 2 ranges 0.20, 4 ranges 0.19; a range keeps >= 32 channels = one full pass of its 4 waves
    const int ysplit = C0 >= 128 ? 4 : C0 >= 64 ? 2 : 1;
    if (centers)
        hipLaunchKernelGGL(expand_c_kernel<true>, dim3((unsigned)(ldp / 256), ysplit), dim3(256), 0, o3d_stream(stream), Z,
                           ldz, gp, cball, cw, centers, W0, ldw, C0, meta, start1, ldp, Y0, part, stat_c);
    else
        hipLaunchKernelGGL(expand_c_kernel<false>, dim3((unsigned)(ldp / 256), ysplit), dim3(256), 0, o3d_stream(stream), Z,
                           ldz, gp, cball, cw, centers, W0, ldw, C0, meta, start1, ldp, Y0, part, stat_c);
    return o3d_launch_status();
}

// o3d_group_expand_c for an xyz-only layer 0 WITHOUT the per-point GEMM: X3 = the packed coordinate operand (3 rows of
// ldz point columns, scaled like `centers`), Y0[c,q] = W0[c,0:3].(X3[:, gp[q]] - centers[cball[q]])
extern "C" int o3d_group_expand_c3(const float* X3, long ldz, const int32_t* gp, const int32_t* cball, const float* cw,
                                   const float* centers, const float* W0, int ldw, int C0, const int32_t* meta,
                                   long start1, long ldp, float* Y0, float* part, const float* stat_c, void* stream) {
    if (!X3 || !gp || !cball || !cw || !centers || !W0 || ldw < 3 || !meta || !Y0 || C0 <= 0 || ldp <= 0 ||
        ldp % 256 != 0 || start1 < 0 || start1 % 256 != 0)
        return O3D_EINVAL;
    const int ysplit = C0 >= 128 ? 4 : C0 >= 64 ? 2 : 1;
    hipLaunchKernelGGL((expand_c_kernel<true, true>), dim3((unsigned)(ldp / 256), ysplit), dim3(256), 0, o3d_stream(stream),
                       X3, ldz, gp, cball, cw, centers, W0, ldw, C0, meta, start1, ldp, Y0, part, stat_c);
    return o3d_launch_status();
}

// out / argq / yarg: per segment a (B, C, npoint_s) block (segment 1's block after segment 0's); balls
// >= seg1_ball = B*npoint0 belong to segment 1 (scale/shift at +C); npoint1 = 0 for one segment
extern "C" int o3d_pool_fwd_c(const float* Y, long ldp, const float* scale, const float* shift,
                              const int32_t* ball_off, const int32_t* ball_cnt, int B, int C, int npoint0,
                              int npoint1, float* out, int32_t* argq, float* yarg, void* stream) {
    if (!Y || !scale || !shift || !ball_off || !ball_cnt || !out || B <= 0 || C <= 0 || npoint0 <= 0 || npoint1 < 0 ||
        (argq && !yarg))
        return O3D_EINVAL;
    const int seg1_ball = B * npoint0, nballs = B * (npoint0 + npoint1);
    // 8 channels per workgroup (4 / 16 measured: 0.37 / 0.42 vs 0.36 ms per step)
    constexpr int POOL_CH = 8;
    hipLaunchKernelGGL(pool_c_kernel<8>, dim3(o3d_cdiv(nballs, 32), o3d_cdiv(C, POOL_CH)), dim3(256), 0,
                       o3d_stream(stream), Y, ldp, scale, shift, ball_off, ball_cnt, C, seg1_ball, npoint0,
                       npoint1 > 0 ? npoint1 : npoint0, nballs, out, argq, yarg);
    return o3d_launch_status();
}

// o3d_pool_fwd_c on the LDS-transposed tile (pool_t_kernel): additionally needs the column -> ball map and the live
// counts; C % 64 == 0 and balls of at most 32 columns (nsample <= 32), else O3D_EINVAL (use o3d_pool_fwd_c)
extern "C" int o3d_pool_fwd_ct(const float* Y, long ldp, const float* scale, const float* shift, const int32_t* ball_off,
                               const int32_t* ball_cnt, const int32_t* cball, const int32_t* meta, long start1, int B, int C,
                               int npoint0, int npoint1, int nsample, float* out, int32_t* argq, float* yarg, void* stream) {
    if (!Y || !scale || !shift || !ball_off || !ball_cnt || !cball || !meta || !out || B <= 0 || C <= 0 || C % PT_CH != 0 ||
        npoint0 <= 0 || npoint1 < 0 || nsample < 1 || nsample > PT_OVER || (argq && !yarg) || ldp < PT_COLS + PT_OVER ||
        ldp % PT_COLS != 0 || start1 < 0 || start1 % PT_COLS != 0)
        return O3D_EINVAL;
    const int seg1_ball = B * npoint0, nballs = B * (npoint0 + npoint1);
    const size_t lds = sizeof(float) * (PT_CH * PT_LD + 3 * PT_RB * PT_RLD) + sizeof(int) * 3 * PT_COLS;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(pool_t_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!attr_ok) return O3D_ELAUNCH;
    const long worst = (ldp / PT_COLS) * (C / PT_CH);
    hipLaunchKernelGGL(pool_t_kernel, dim3((unsigned)(worst < PT_GRID ? worst : PT_GRID)), dim3(256), lds, o3d_stream(stream), Y, ldp,
                       scale, shift, ball_off, ball_cnt, cball, meta, start1, C, seg1_ball, npoint0,
                       npoint1 > 0 ? npoint1 : npoint0, nballs, out, argq, yarg);
    return o3d_launch_status();
}

// Backward of the pool (pooled tensors in the per-segment (B,C,npoint) blocks of o3d_pool_fwd_c): D (C, ldp) zero on the live columns then D[c, argq] = dOut
// where out > 0; part [nseg][2][C] = BatchNorm-backward partials of the pooled layer per segment.
extern "C" int o3d_pool_bwd_c(const float* dOut, const float* out, const int32_t* argq, const float* yarg,
                              const float* mean, int B, int C, int npoint0, int npoint1, const int32_t* meta,
                              long start1, long ldp, float* D, float* part, void* stream) {
    if (!dOut || !out || !argq || !yarg || !mean || !meta || !D || !part || B <= 0 || C <= 0 || npoint0 <= 0 ||
        npoint1 < 0 || ldp <= 0 || ldp % 4 != 0)
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    const int nseg = npoint1 > 0 ? 2 : 1;
    const int seg1_ball = B * npoint0, nballs = B * (npoint0 + npoint1), np1 = npoint1 > 0 ? npoint1 : npoint0;
    // zero the live columns, then statistics + scatter in one walk over (channel, ball)
    hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)o3d_cdiv(ldp, 1024), C), dim3(256), 0, s, D, ldp, meta, start1);
    hipLaunchKernelGGL(pool_bwd_partials_c_kernel, dim3(C, nseg, POOL_BWD_SPLIT), dim3(256), 0, s, dOut, out, yarg, mean, C,
                       nballs, seg1_ball, npoint0, np1, part, argq, D, ldp);
    return o3d_launch_status();
}

// o3d_pool_bwd_c in one pass (pool_bwd_dense_kernel): no zero fill.  The pooled-output gradient comes per segment as a
// (B, C, npoint_s) tensor with arbitrary batch / channel strides (sb, sc in floats; NULL = zeros): the template and the
// search gradient of a paired level, or a (C, B * npoint) flat view, are read where they lie (were: 6 strided torch copies
// per step into one buffer).  Needs the column -> ball map and the ball offsets of o3d_compact_build; C % 8 == 0, ldp % 512 == 0, start1 % 512 == 0 (else O3D_EINVAL: use o3d_pool_bwd_c).  part: one row
// {sum g, sum g (yarg - mean)} per 512-column chunk, [ldp / 512][2][C], live rows only (finalize with tile = 512).
extern "C" int o3d_pool_bwd_dense(const float* dOut0, long sb0, long sc0, const float* dOut1, long sb1, long sc1,
                                  const float* out, const int32_t* argq, const float* yarg,
                                  const float* mean, const int32_t* cball, const int32_t* ball_off, const int32_t* meta,
                                  long start1, long ldp, int B, int C, int npoint0, int npoint1, float* D, float* part,
                                  void* stream) {
    if (!out || !argq || !yarg || !mean || !cball || !ball_off || !meta || !D || !part || B <= 0 || C <= 0 ||
        C % PBD_CH != 0 || npoint0 <= 0 || npoint1 < 0 || ldp <= 0 || ldp % PBD_COLS != 0 || start1 < 0 ||
        start1 % PBD_COLS != 0 || ldp > 0x7fffffffL)
        return O3D_EINVAL;
    const int seg1_ball = B * npoint0, np1 = npoint1 > 0 ? npoint1 : npoint0;
    const PoolGrad g0 = {dOut0, sb0, sc0}, g1 = {dOut1, sb1, sc1};
    // (measured, round 5: 4 channel groups per workgroup of a worst-case grid -- a quarter of the workgroups, the index prologue
    // paid once -- took 0.31 ms per step against 0.23: 1.5 rounds of long workgroups instead of 6 of short ones; the persistent
    // grid below has neither the dead workgroups nor that quantisation)
    PoolBwdDense a = {g0, g1, out, argq, yarg, mean, cball, ball_off, meta, npoint1 > 0 ? start1 : 0, ldp, C,
                      B * (npoint0 + npoint1), seg1_ball, npoint0, np1, D, part};
    const long worst = (ldp / PBD_COLS) * (C / PBD_CH);
    const unsigned grid = (unsigned)(worst < PBD_GRID ? worst : PBD_GRID);
    hipLaunchKernelGGL(pool_bwd_dense_kernel, dim3(grid), dim3(256), 0, o3d_stream(stream), a);
    return o3d_launch_status();
}

template <int CS, int BT = 256>
static int launch_reduce_c(const float* dN, const float* Y0, long ldp, const float* A1, const float* A2,
                           const float* A3, const int32_t* gp, const int32_t* cball, const float* cw,
                           const int32_t* ball_off, const int32_t* ball_cnt, int B, int nseg, SegParams sp0,
                           SegParams sp1, int C0,
                           long lds_row, int nballs, float* S, float* T, hipStream_t s) {
    const int span = (sp0.ld + sp0.npoint) > (sp1.ld + sp1.npoint) ? (sp0.ld + sp0.npoint) : (sp1.ld + sp1.npoint);
    const size_t lds = sizeof(float) * CS * (size_t)span;
    if (lds > 64 * 1024) return O3D_EINVAL;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(reduce_c_kernel<CS, BT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return O3D_ELAUNCH;
    const int slabs = (C0 + CS - 1) / CS;
    hipLaunchKernelGGL((reduce_c_kernel<CS, BT>), dim3(B * nseg * slabs), dim3(BT), lds, s, dN, Y0, ldp, A1, A2, A3, gp, cball,
                       cw, ball_off, ball_cnt, B, sp0, sp1, C0, lds_row, S, T, nballs);
    return o3d_launch_status();
}

// S (C0, lds_row) per source point, T (C0, nballs) per ball or NULL.  One or two segments of B clouds each
// (segment s: npoint[s] balls and ld[s] point columns per cloud; segment 1's points start at column B*ld[0],
// its balls at B*npoint[0], its constants A1..A3 at +C0).
extern "C" int o3d_group_reduce_c(const float* dN, const float* Y0, long ldp, const float* A1, const float* A2,
                                  const float* A3, const int32_t* gp, const int32_t* cball, const float* cw,
                                  const int32_t* ball_off, const int32_t* ball_cnt, int B, int nseg, int npoint0,
                                  int ld0, int npoint1, int ld1, int C0, float* S, float* T, void* stream) {
    if (!dN || !Y0 || !A1 || !A2 || !A3 || !gp || !cball || !cw || !ball_off || !ball_cnt || !S || B <= 0 || npoint0 <= 0 ||
        ld0 <= 0 || C0 <= 0 || (nseg != 1 && nseg != 2) || (nseg == 2 && (npoint1 <= 0 || ld1 <= 0)))
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    const SegParams sp0 = {npoint0, ld0, 0, 0};
    const SegParams sp1 = nseg == 2 ? SegParams{npoint1, ld1, B * ld0, B * npoint0} : sp0;
    const long lds_row = (long)B * ld0 + (nseg == 2 ? (long)B * ld1 : 0);
    const int nballs = B * npoint0 + (nseg == 2 ? B * npoint1 : 0);
    // channels per workgroup: 1 (measured on the MI355X, same run A/B, ms per step of this kernel: 4 channels
    // 0.76, 2 channels 0.71, 1 channel 0.65 -- more, smaller workgroups win over amortising the column metadata loads)
    return launch_reduce_c<1>(dN, Y0, ldp, A1, A2, A3, gp, cball, cw, ball_off, ball_cnt, B, nseg, sp0, sp1, C0,
                              lds_row, nballs, S, T, s);
}

// dW (C0, ldw): columns 0..2 -= T (C0, nballs) . centers (nballs, 3)
extern "C" int o3d_center_term(const float* T, const float* centers, int C0, int nballs, int ldw, float* dW,
                               void* stream) {
    if (!T || !centers || !dW || C0 <= 0 || nballs <= 0 || ldw < 3) return O3D_EINVAL;
    hipLaunchKernelGGL(center_term_kernel, dim3(C0), dim3(1024), 0, o3d_stream(stream), T, centers, nballs, ldw, dW,
                       static_cast<float*>(nullptr), 0);
    return o3d_launch_status();
}

// o3d_center_term writing the finished gradient compactly: out (C0, ncols) = dW[:, :ncols] with the centre term taken off
// columns 0..2 (dW (C0, ldw), ldw >= ncols >= 3, is left as it was)
extern "C" int o3d_center_term_out(const float* T, const float* centers, int C0, int nballs, int ldw, const float* dW,
                                   int ncols, float* out, void* stream) {
    if (!T || !centers || !dW || !out || C0 <= 0 || nballs <= 0 || ncols < 3 || ldw < ncols) return O3D_EINVAL;
    hipLaunchKernelGGL(center_term_kernel, dim3(C0), dim3(1024), 0, o3d_stream(stream), T, centers, nballs, ldw,
                       const_cast<float*>(dW), out, ncols);
    return o3d_launch_status();
}

// Gradient of the ball centres (grouped_xyz = xyz[idx] - new_xyz, pointnet2_utils.py:319-320): out (3, nballs)[k, ball] =
// scale * sum_c W0[c, k] * T[c, ball], T (C0, nballs) = per-ball sums of dY0, W0 (C0, ldw).  Was a torch.addmm (one rocBLAS
// launch per level whose centres carry a gradient -- the vote aggregation, models/head/rpn.py:55-60 -- and the last vendor-BLAS
// kernel of the BAT / P2B steps); thread per ball, T read coalesced, the three weight columns broadcast.
namespace {
__global__ __launch_bounds__(256) void center_grad_kernel(const float* __restrict__ T, const float* __restrict__ W0, int ldw, int C0,
                                                          int nballs, float scale, float* __restrict__ out) {
    const int ball = blockIdx.x * 256 + threadIdx.x;
    if (ball >= nballs) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int c = 0; c < C0; ++c) {
        const float t = T[(long)c * nballs + ball];
        s0 = fmaf(W0[(long)c * ldw + 0], t, s0);
        s1 = fmaf(W0[(long)c * ldw + 1], t, s1);
        s2 = fmaf(W0[(long)c * ldw + 2], t, s2);
    }
    out[ball] = scale * s0;
    out[(long)nballs + ball] = scale * s1;
    out[2L * nballs + ball] = scale * s2;
}
}  // namespace

extern "C" int o3d_center_grad(const float* T, const float* W0, int ldw, int C0, int nballs, float scale, float* out, void* stream) {
    if (!T || !W0 || !out || C0 <= 0 || nballs <= 0 || ldw < 3) return O3D_EINVAL;
    hipLaunchKernelGGL(center_grad_kernel, dim3(o3d_cdiv(nballs, 256)), dim3(256), 0, o3d_stream(stream), T, W0, ldw, C0, nballs,
                       scale, out);
    return o3d_launch_status();
}

// X0 (rows, ldz), ldz = B*(ld0 + ld1): rows [0,nxyz) = xyz^T * inv_radius, [nxyz, nxyz+C) = feats, the rest zero;
// columns of cloud b of segment s: [base_s + b*ld_s, +N_s) live, the padding up to ld_s zero.  xyz_s (B,N_s,3) (NULL
// when nxyz == 0), feats_s (B,C,N_s) (NULL when C == 0); N1 = 0: one segment.
extern "C" int o3d_pack_points(const float* xyz0, const float* feats0, int N0, int ld0, const float* xyz1,
                               const float* feats1, int N1, int ld1, int B, int nxyz, int C, float inv_radius, int rows,
                               float* X0, void* stream) {
    if (!X0 || B <= 0 || N0 <= 0 || ld0 < N0 || N1 < 0 || (N1 > 0 && ld1 < N1) || (nxyz != 0 && nxyz != 3) || C < 0 ||
        rows < nxyz + C || (nxyz && (!xyz0 || (N1 > 0 && !xyz1))) || (C && (!feats0 || (N1 > 0 && !feats1))))
        return O3D_EINVAL;
    const long ldz = (long)B * ld0 + (N1 > 0 ? (long)B * ld1 : 0);
    const PackSeg s0 = {xyz0, feats0, N0, ld0}, s1 = {xyz1, feats1, N1, N1 > 0 ? ld1 : 1};
    hipLaunchKernelGGL(pack_points_kernel, dim3((unsigned)o3d_cdiv(ldz, 256), rows), dim3(256), 0, o3d_stream(stream), s0, s1,
                       B, nxyz, C, inv_radius, ldz, X0);
    return o3d_launch_status();
}

// ---- the reduce through a transposed index (see csr_build_kernel); same result as o3d_group_reduce_c ----------
static int rg_nchunk(int spanmax) { return (spanmax + 4 + RG_CH - 1) / RG_CH; }

// int32 elements of `poff` scratch for o3d_group_reduce_gather, or -1 when the shape does not fit its LDS budget
// (the caller then uses o3d_group_reduce_c).  spanmax = npoint*nsample of the largest segment.
extern "C" long o3d_group_reduce_gather_scratch(int B, int nseg, int npoint0, int ld0, int npoint1, int ld1, int spanmax) {
    if (B <= 0 || (nseg != 1 && nseg != 2) || npoint0 <= 0 || ld0 <= 0 || spanmax <= 0 ||
        (nseg == 2 && (npoint1 <= 0 || ld1 <= 0)))
        return -1;
    const int ldm = nseg == 2 && ld1 > ld0 ? ld1 : ld0, npm = nseg == 2 && npoint1 > npoint0 ? npoint1 : npoint0;
    const long nbin = (long)rg_nchunk(spanmax) * ldm;
    const size_t lds_csr = ((size_t)nbin + 1 + 1024) * 4;
    const size_t lds_red = ((size_t)rg_cs() * RG_CH + (size_t)rg_cs() * (ldm + npm)) * 4 + (size_t)RG_CH * 4 + 8;
    if (lds_csr > 128 * 1024 || lds_red > 80 * 1024) return -1;
    return (long)B * nseg * (nbin + 1);
}

extern "C" int o3d_group_reduce_gather(const float* dN, const float* Y0, long ldp, const float* A1, const float* A2,
                                       const float* A3, const int32_t* gp, const float* cw, const int32_t* ball_off,
                                       const int32_t* ball_cnt, int B, int nseg, int npoint0, int ld0, int npoint1,
                                       int ld1, int C0, int spanmax, int32_t* perm, int32_t* poff, float* S, float* T,
                                       void* stream) {
    if (!dN || !Y0 || !A1 || !A2 || !A3 || !gp || !cw || !ball_off || !ball_cnt || !perm || !poff || !S || C0 <= 0 ||
        ldp <= 0 || ldp % 4 != 0 || o3d_group_reduce_gather_scratch(B, nseg, npoint0, ld0, npoint1, ld1, spanmax) < 0)
        return O3D_EINVAL;
    hipStream_t s = o3d_stream(stream);
    const SegParams sp0 = {npoint0, ld0, 0, 0};
    const SegParams sp1 = nseg == 2 ? SegParams{npoint1, ld1, B * ld0, B * npoint0} : sp0;
    const long lds_row = (long)B * ld0 + (nseg == 2 ? (long)B * ld1 : 0);
    const int nballs = B * npoint0 + (nseg == 2 ? B * npoint1 : 0);
    const int ldm = nseg == 2 && ld1 > ld0 ? ld1 : ld0, npm = nseg == 2 && npoint1 > npoint0 ? npoint1 : npoint0;
    const int nchunk = rg_nchunk(spanmax);
    const int stride = nchunk * ldm + 1;
    const size_t lds_csr = ((size_t)nchunk * ldm + 1 + 1024) * 4;
    const int cs = rg_cs();
    const size_t lds_red = ((size_t)cs * RG_CH + (size_t)cs * (ldm + npm)) * 4 + (size_t)RG_CH * 4 + 8;
    if (lds_csr > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(csr_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_csr) != hipSuccess)
        return O3D_ELAUNCH;
    const void* rgk = reinterpret_cast<const void*>(reduce_gather_kernel<2>);
    if (lds_red > 48 * 1024 &&
        hipFuncSetAttribute(rgk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_red) != hipSuccess)
        return O3D_ELAUNCH;
    hipLaunchKernelGGL(csr_build_kernel, dim3(B * nseg), dim3(1024), lds_csr, s, gp, ball_off, ball_cnt, B, sp0, sp1, nchunk,
                       stride, perm, poff);
    const int slabs = (C0 + cs - 1) / cs;
    hipLaunchKernelGGL(reduce_gather_kernel<2>, dim3(B * nseg * slabs), dim3(256), lds_red, s, dN, Y0, ldp, A1, A2, A3, cw,
                           ball_off, ball_cnt, perm, poff, stride, B, sp0, sp1, C0, lds_row, S, T, nballs);
    return o3d_launch_status();
}
