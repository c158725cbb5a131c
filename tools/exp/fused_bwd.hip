// tools/exp/fused_bwd.hip -- EXPERIMENT (not built into the library): data gradient + weight gradient of one
// aligned inner layer of the grouped MLP in ONE kernel, for the HBM-bound 64-channel layers of SA1
// (HISTORY.md (round 1-2) section 9.2).
//
// Today (csrc/mlp_direct.hip + csrc/mlp_wgrad.hip) the two gradients are two launches that each read dN and Y
// (2*Cout rows) plus Yprev (Cin rows): (4*Cout + 2*Cin + Cin) * 4 B per column.  Here one staging of
//     dY = A1*dN + w*(A2*Y + A3)     (Cout rows)     and     Xt = relu(sc*Yprev + sh)     (Cin rows)
// in LDS feeds both MFMA chains: (2*Cout + 2*Cin) * 4 B per column.
//   weight gradient   dW[co][ci] += sum_pos dY[co][pos] * Xt[ci][pos]            (exactly wgrad2_kernel's inner loop)
//   data gradient     dX[ci][pos]  = sum_co Wt[ci][co] * dY[co][pos]   masked by Xt > 0
//       A fragments = Wt rows, constant over the whole launch: loaded ONCE into registers (Cout/8 float4 per lane);
//       B fragments = dY[co][pos] read back from the staging buffer, one ds_read_b32 per MFMA
//   BatchNorm-backward partials of the data gradient, {sum g, sum g*(yprev - mean)}: the second from
//       sum g*Xt  (Xt is what the LDS holds):  sum g*(yprev-mean) = (sum g*Xt - beta*sum g) / sc,  beta = sh + sc*mean
//       accumulated per lane (its one column of every chunk) in 32 registers, folded over the lanes once at the end
// Registers: ~225-240 VGPRs + 80 AGPRs -> ONE workgroup (4 waves) per CU; the whole next chunk (49-65 KB per CU) is in
// flight while the current one is multiplied, which is more than latency x bandwidth needs (1 us x 5 TB/s = 5 MB < 12 MB).
// Scope: Cin == 64, Cout in {64, 128}; dense dN; compact layout (w, meta, start1) or plain (all NULL / 0).
// Output partial rows: part_w [nslices*WK][Cout][Cin] as wgrad2; part_s [2 segments][nslices*NPT][2][Cin]
// (a slice writes its sums into its own segment's block and zeros into the other).
//
// Build + check: tools/exp/fused_bwd_check.py.  Verified on an MI355X (profiles/r02_fused_bwd_experiment.txt): same
// errors against fp64 as the production pair (dW 1e-7, dX 3e-7, statistics 5e-8); 64 -> 64 at the benchmarked size 0.138 ms
// (+ ~5 us partial-tile reduction) against 0.154 ms for the pair, 64 -> 128 0.295 against 0.244 ms (slower: issue-bound).
#include <cstdint>
#include <hip/hip_runtime.h>
#include "mlp_common.hpp"

namespace {

struct FusedBwdArgs {
    const float* dN; const float* Y;                      // (Cout, P)
    const float* A1; const float* A2; const float* A3;    // (nseg, Cout)
    const float* X;                                       // Yprev (Cin, P): raw output of the producer layer
    const float* in_scale; const float* in_shift; const float* in_mean;   // (nseg, Cin): BatchNorm of the producer
    const float* Wt;                                      // (Cin, Cout) = W^T
    int Cin, Cout, P;
    int chunks_per_block, total_chunks, nslices;
    const float* w; const int32_t* meta; long start1;     // compact layout (or NULL, NULL, 0)
    float* part_w;                                        // [nslices*WK][Cout][Cin]
    float* part_s;                                        // [2][nslices*NPT][2][Cin]
    float* dX;                                            // (Cin, P)
};

template <int TM>      // TM = Cout (64 | 128); Cin = 64
__global__ __launch_bounds__(256) void fused_bwd_kernel(FusedBwdArgs a) {   // ~305 registers: one workgroup per CU (see header)
    constexpr int TN = 64;
    constexpr int WM = TM / 64, WK = 4 / WM;              // waves over output rows / over the chunk's positions
    constexpr int CP = (TM + TN == 128) ? 64 : 32;        // positions per staged chunk
    constexpr int NPT = CP / 32;                          // 32-position tiles of the data gradient per chunk
    constexpr int LD = CP + 4;
    constexpr int F = CP / 4, RPP = 256 / F, PA = TM / RPP, PB = TN / RPP;
    constexpr int NG = CP / WK / 8;                       // k groups of 8 positions per wave per chunk (weight gradient)
    constexpr int GD = TM / 8;                            // k groups of 8 output channels (data gradient)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto As = [&](int buf) -> float* { return smem + buf * ((TM + TN) * LD); };
    auto Bs = [&](int buf) -> float* { return smem + buf * ((TM + TN) * LD) + TM * LD; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wk = wave / WM, wm0 = (wave % WM) * 64;     // weight gradient: k split, 64-row block
    const int dci = (wave & 1) * 32, dpt = wave >> 1;     // data gradient: 32-row ci tile, 32-position tile
    const bool dactive = dpt < NPT;
    const int slice = blockIdx.x;
    int c_begin, c_end, seg = 0;
    long col0 = 0;
    if (a.meta) {       // live chunks of each segment split evenly over the slices; a slice never straddles segments
        const int n0 = a.meta[0] / CP, n1 = a.start1 > 0 ? a.meta[4] / CP : 0;
        int nsl0 = a.nslices;
        if (n1 > 0) {
            nsl0 = (int)(((long)a.nslices * n0 + (n0 + n1) / 2) / (n0 + n1));
            nsl0 = nsl0 < 1 ? 1 : (nsl0 > a.nslices - 1 ? a.nslices - 1 : nsl0);
        }
        seg = slice >= nsl0 ? 1 : 0;
        const int nsl = seg ? a.nslices - nsl0 : nsl0, ls = seg ? slice - nsl0 : slice, n = seg ? n1 : n0;
        const int per = (n + nsl - 1) / nsl;
        c_begin = ls * per;
        c_end = c_begin + per < n ? c_begin + per : n;
        col0 = seg ? a.start1 : 0;
    } else {
        c_begin = slice * a.chunks_per_block;
        c_end = c_begin + a.chunks_per_block < a.total_chunks ? c_begin + a.chunks_per_block : a.total_chunks;
    }
    const float* A1 = a.A1 + (seg ? a.Cout : 0);
    const float* A2 = a.A2 + (seg ? a.Cout : 0);
    const float* A3 = a.A3 + (seg ? a.Cout : 0);
    const float* in_scale = a.in_scale + (seg ? a.Cin : 0);
    const float* in_shift = a.in_shift + (seg ? a.Cin : 0);
    const float* in_mean = a.in_mean + (seg ? a.Cin : 0);
    const int r0 = tid / F, c4 = tid % F;

    float ka1[PA], ka2[PA], ka3[PA], ksc[PB], ksh[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) { const int co = r0 + RPP * i; ka1[i] = A1[co]; ka2[i] = A2[co]; ka3[i] = A3[co]; }
#pragma unroll
    for (int i = 0; i < PB; ++i) { const int ci = r0 + RPP * i; ksc[i] = in_scale[ci]; ksh[i] = in_shift[ci]; }

    // data-gradient A fragments: row ci = dci + l31 of Wt, MFMA step s of group g consumes co = 8g + 4h + s
    float4 wt[GD];
#pragma unroll
    for (int g = 0; g < GD; ++g) wt[g] = *reinterpret_cast<const float4*>(a.Wt + (long)(dci + l31) * TM + 8 * g + 4 * h);

    float4 rg[PA], ry[PA], rx[PB];
    float4 rw = make_float4(1.f, 1.f, 1.f, 1.f);
    auto load_chunk = [&](int chl) {
        const long p = (col0 / CP + chl) * CP + 4 * c4;
        if (a.w) rw = *reinterpret_cast<const float4*>(&a.w[p]);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const long row = r0 + RPP * i;
            ry[i] = *reinterpret_cast<const float4*>(&a.Y[row * a.P + p]);
            rg[i] = *reinterpret_cast<const float4*>(&a.dN[row * a.P + p]);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) rx[i] = *reinterpret_cast<const float4*>(&a.X[(long)(r0 + RPP * i) * a.P + p]);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            float4 o;
            o.x = fmaf(ka1[i], rg[i].x, rw.x * fmaf(ka2[i], ry[i].x, ka3[i]));
            o.y = fmaf(ka1[i], rg[i].y, rw.y * fmaf(ka2[i], ry[i].y, ka3[i]));
            o.z = fmaf(ka1[i], rg[i].z, rw.z * fmaf(ka2[i], ry[i].z, ka3[i]));
            o.w = fmaf(ka1[i], rg[i].w, rw.w * fmaf(ka2[i], ry[i].w, ka3[i]));
            *reinterpret_cast<float4*>(&As(buf)[(r0 + RPP * i) * LD + 4 * c4]) = o;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            float4 v = rx[i];
            v.x = fmaxf(fmaf(v.x, ksc[i], ksh[i]), 0.f); v.y = fmaxf(fmaf(v.y, ksc[i], ksh[i]), 0.f);
            v.z = fmaxf(fmaf(v.z, ksc[i], ksh[i]), 0.f); v.w = fmaxf(fmaf(v.w, ksc[i], ksh[i]), 0.f);
            *reinterpret_cast<float4*>(&Bs(buf)[(r0 + RPP * i) * LD + 4 * c4]) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float sacc[32];            // data-gradient statistics of this lane's column share: [statistic][r]
#pragma unroll
    for (int r = 0; r < 32; ++r) sacc[r] = 0.f;

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        __syncthreads();
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int t = ch - c_begin;
            if (ch + 1 < c_end) load_chunk(ch + 1);
            const float* A_ = As(t & 1);
            const float* B_ = Bs(t & 1);
            // ---- weight gradient: this wave's 64 x 64 block over its share of the chunk's positions
            {
                const float* Aw = A_ + wk * (CP / WK);
                const float* Bw = B_ + wk * (CP / WK);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float av[2][4], bv[2][4];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float4 x = *reinterpret_cast<const float4*>(&Aw[(wm0 + 32 * u + l31) * LD + 8 * g + 4 * h]);
                        av[u][0] = x.x; av[u][1] = x.y; av[u][2] = x.z; av[u][3] = x.w;
                        const float4 y = *reinterpret_cast<const float4*>(&Bw[(32 * u + l31) * LD + 8 * g + 4 * h]);
                        bv[u][0] = y.x; bv[u][1] = y.y; bv[u][2] = y.z; bv[u][3] = y.w;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                            for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = mfma32(av[tm][s], bv[tn][s], acc[tm][tn]);
                }
            }
            // ---- data gradient: ci tile dci, positions 32*dpt .. +31 of the chunk
            if (dactive) {
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
                const float* Bd = A_ + 32 * dpt + l31;          // dY[co][pos]: row co, this lane's position
#pragma unroll
                for (int g = 0; g < GD; ++g) {
                    const float b0 = Bd[(8 * g + 4 * h + 0) * LD], b1 = Bd[(8 * g + 4 * h + 1) * LD];
                    const float b2 = Bd[(8 * g + 4 * h + 2) * LD], b3 = Bd[(8 * g + 4 * h + 3) * LD];
                    d = mfma32(wt[g].x, b0, d);
                    d = mfma32(wt[g].y, b1, d);
                    d = mfma32(wt[g].z, b2, d);
                    d = mfma32(wt[g].w, b3, d);
                }
                const long q = (col0 / CP + ch) * CP + 32 * dpt + l31;      // this lane's column
                // mask, store, and this lane's share (its one column per chunk) of the two sums of every row: at one workgroup
                // per CU the register file has room for 32 accumulators per lane; they are folded over the lanes once, at the end
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = acc_row(r, h);
                    const float xt = B_[(dci + row) * LD + 32 * dpt + l31];
                    const float gq = xt > 0.f ? d[r] : 0.f;
                    a.dX[(long)(dci + row) * a.P + q] = gq;
                    sacc[r] += gq;
                    sacc[16 + r] = fmaf(gq, xt, sacc[16 + r]);
                }
            }
            if (ch + 1 < c_end) store_chunk((t + 1) & 1);
            __syncthreads();
        }
    }
    // ---- weight-gradient partial tile of this (slice, k-split wave)
    float* dst = a.part_w + (long)(slice * WK + wk) * TM * TN;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wm0 + 32 * tm + acc_row(r, h);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) dst[(long)co * TN + 32 * tn + l31] = acc[tm][tn][r];
        }
    // ---- BatchNorm-backward partials of the data gradient: row (slice*NPT + dpt) of this slice's segment block, zeros in
    // the other segment's block.  After the reduce-scatter lane l31 of half h owns (statistic l31>>4, row acc_row(l31&15, h)).
    if (dactive) {
        reduce_scatter32(sacc, l31);
        const int ci = dci + acc_row(l31 & 15, h);
        const int which = l31 >> 4;
        const float other = __shfl_xor(sacc[0], 16, 64);      // the other statistic of the same row sits 16 lanes away
        const float s1 = which ? other : sacc[0], s2x = which ? sacc[0] : other;    // sum g, sum g*Xt
        const float sc = in_scale[ci], beta = fmaf(sc, in_mean[ci], in_shift[ci]);
        const float s2 = sc != 0.f ? (s2x - beta * s1) / sc : 0.f;                  // sum g*(yprev - mean)
        const long rows = (long)a.nslices * NPT, row = (long)slice * NPT + dpt;
        float* mine = a.part_s + ((long)seg * rows + row) * 2 * TN;
        float* theirs = a.part_s + ((long)(1 - seg) * rows + row) * 2 * TN;
        mine[which * TN + ci] = which ? s2 : s1;
        theirs[which * TN + ci] = 0.f;
    }
}

template <int TM>
int launch(const FusedBwdArgs& a, hipStream_t s) {
    constexpr int CP = (TM + 64 == 128) ? 64 : 32;
    const size_t lds = sizeof(float) * 2 * (TM + 64) * (CP + 4);
    const void* fn = reinterpret_cast<const void*>(fused_bwd_kernel<TM>);
    if (lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -3;
    hipLaunchKernelGGL((fused_bwd_kernel<TM>), dim3(a.nslices), dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

// nslices (= workgroups) for a problem of P columns: one resident round (one workgroup per CU) unless the problem is smaller
extern "C" int o3d_exp_bwd_fused_slices(int Cout, long P) {
    const int CP = (Cout + 64 == 128) ? 64 : 32;
    const long total = P / CP;
    int nsl = 256;
    while (nsl > 8 && total / nsl < 4) nsl -= 8;
    if (nsl > total) nsl = (int)total;
    return nsl < 1 ? 1 : nsl;
}

// dX (Cin, P), part_w [nslices * (4 / (Cout/64))][Cout][Cin], part_s [2][nslices * NPT][2][Cin] (NPT = 2 for Cout 64, 1 for 128)
extern "C" int o3d_exp_bwd_fused(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3,
                                 const float* X, const float* in_scale, const float* in_shift, const float* in_mean,
                                 const float* Wt, int Cin, int Cout, long P, const float* w, const int32_t* meta,
                                 long start1, int nslices, float* part_w, float* part_s, float* dX, void* stream) {
    if (!dN || !Y || !A1 || !A2 || !A3 || !X || !in_scale || !in_shift || !in_mean || !Wt || !part_w || !part_s || !dX ||
        Cin != 64 || (Cout != 64 && Cout != 128) || P <= 0 || P % 64 != 0 || P > 0x7fffffff || nslices <= 0 ||
        (meta != nullptr) != (w != nullptr) || start1 < 0 || start1 % 256 != 0)
        return -1;
    const int CP = (Cout + 64 == 128) ? 64 : 32;
    FusedBwdArgs a = {};
    a.dN = dN; a.Y = Y; a.A1 = A1; a.A2 = A2; a.A3 = A3; a.X = X; a.in_scale = in_scale; a.in_shift = in_shift;
    a.in_mean = in_mean; a.Wt = Wt; a.Cin = Cin; a.Cout = Cout; a.P = (int)P;
    a.total_chunks = (int)(P / CP);
    a.chunks_per_block = (a.total_chunks + nslices - 1) / nslices;
    a.nslices = nslices;
    a.w = w; a.meta = meta; a.start1 = start1; a.part_w = part_w; a.part_s = part_s; a.dX = dX;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return Cout == 64 ? launch<64>(a, s) : launch<128>(a, s);
}
