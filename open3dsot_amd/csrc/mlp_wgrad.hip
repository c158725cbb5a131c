// mlp_wgrad.hip -- weight gradient of the aligned inner layers (Cin, Cout multiples of 64):
//     dW[co,ci] = sum_{b,p} dY[b,co,p] * f(X[b,ci,p]),   dY = A1*dN + A2*Y + A3,  f = BN+ReLU of the producer
// (backward of Conv2d 1x1 + BatchNorm2d, pointnet2/utils/pytorch_utils.py:12-37,68-121).
//
// The contraction runs over positions, so both MFMA operands are rows that are contiguous along k:
// here LDS staging IS the right tool (a lane of an MFMA fragment wants 16 bytes of ONE row; a coalesced
// global load wants 128 bytes of a row from 8 neighbouring lanes) -- unlike the forward / data-gradient
// kernels of mlp_direct.hip.  What this version fixes over the generic kernel in mlp.hip:
//   * the workgroup tile matches the layer (64x64, 128x64, 64x128, 128x128): the 64-channel layers of
//     SA1 no longer multiply 75 % zeros; waves left over split the chunk's positions (k) instead;
//   * no bounds checks, per-row constants hoisted out of the position loop;
//   * exactly one resident round of workgroups (2 per CU x 256 CUs) instead of 1.5;
//   * the output tiles of one position slice sit on one XCD (they re-read the same rows: L2 hits);
//   * pooled source read as the packed {gradient, arg} pairs of o3d_pool_bwd_partials.
// Partial tiles [slice][Cout][Cin] are summed in a fixed order by wgrad_reduce_kernel (mlp.hip).
#include "mlp_common.hpp"

namespace {

struct Wgrad2Args {
    const float* dN;       // (B,Cout,P) dense source or NULL
    const float2* pk;      // pooled source (B,Cout,P/ns): {masked dOut, bits(arg-max slot)} per (channel, ball)
    int ns;
    const float* Y;        // (B,Cout,P)
    const float* A1; const float* A2; const float* A3;
    const float* X;        // (B,Cin,P) raw output of the producer
    const float* in_scale; const float* in_shift;
    int B, Cin, Cout, P;
    int chunks_per_block, total_chunks, nslices;
    float* part;           // [nslices*WK][Cout][Cin]
    const float* w;        // compact layout: per-position weight of the A2*Y+A3 term, or NULL
    const int32_t* meta;   // compact layout: 4 device ints per segment, meta[4*s] = live columns of segment s
    long start1;           // first column of segment 1 (0 = one segment); its per-channel constants follow
                           // segment 0's (A1..A3 after Cout floats, in_scale/in_shift after Cin floats)
    float* dY_out;         // (Cout, P) or NULL: the staged operand dY = A1*dN + w*(A2*Y + A3) of the live columns written out
                           // ONCE (by the workgroups of input tile 0), so that the data gradient can load one tensor
};

template <int TM, int TN, int POOLED>
__device__ __forceinline__ void wgrad2_body(const Wgrad2Args& a, const int bid, float* smem) {
    constexpr int WM = TM / 64, WN = TN / 64, WK = 4 / (WM * WN);
    constexpr int CP = (TM + TN == 128) ? 64 : 32;      // positions per staged chunk
    constexpr int LD = CP + 4;                          // [row][pos] stride: conflict-free ds_read_b128
    constexpr int F = CP / 4;                           // float4 per row
    constexpr int RPP = 256 / F;                        // rows staged per pass
    constexpr int PA = TM / RPP, PB = TN / RPP;         // passes
    constexpr int NG = CP / WK / 8;                     // k groups of 8 per wave per chunk
    auto As = [&](int buf) -> float* { return smem + buf * ((TM + TN) * LD); };
    auto Bs = [&](int buf) -> float* { return smem + buf * ((TM + TN) * LD) + TM * LD; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wk = wave / (WM * WN), wmn = wave - wk * (WM * WN);
    const int wm0 = (wmn / WN) * 64, wn0 = (wmn % WN) * 64;
    const int tiles_ci = a.Cin / TN, ntile = tiles_ci * (a.Cout / TM);
    int tile_id, slice;
    if ((a.nslices & 7) == 0) {      // keep the tiles of one position slice on one XCD (shared L2)
        const int xcd = bid & 7, local = bid >> 3;
        tile_id = local % ntile;
        slice = (local / ntile) * 8 + xcd;
    } else {
        tile_id = bid % ntile;
        slice = bid / ntile;
    }
    const int ci0 = (tile_id % tiles_ci) * TN, co0 = (tile_id / tiles_ci) * TM;
    int c_begin, c_end;
    int seg = 0;
    long col0 = 0;                 // first column of the range this workgroup's chunks are counted from
    if (a.meta) {
        // data-dependent column counts: the live chunks of each segment are split evenly over the slices,
        // and a slice never straddles segments (its per-channel constants belong to one of them)
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
    const bool plain_x = a.in_scale == nullptr;          // X as stored (no BatchNorm + ReLU on load)
    const float* in_scale = plain_x ? nullptr : a.in_scale + (seg ? a.Cin : 0);
    const float* in_shift = plain_x ? nullptr : a.in_shift + (seg ? a.Cin : 0);
    const float x_floor = plain_x ? -INFINITY : 0.f;
    const int chunks_per_b = a.P / CP;
    const int r0 = tid / F, c4 = tid % F;
    const int np = POOLED == 1 ? a.P / a.ns : 1;

    // per-row constants of this thread's staging rows
    float ka1[PA], ka2[PA], ka3[PA], ksc[PB], ksh[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int co = co0 + r0 + RPP * i;
        ka1[i] = A1[co]; ka2[i] = A2[co]; ka3[i] = A3[co];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int ci = ci0 + r0 + RPP * i;
        ksc[i] = plain_x ? 1.f : in_scale[ci]; ksh[i] = plain_x ? 0.f : in_shift[ci];
    }

    float4 rg[PA], ry[PA], rx[PB];
    float4 rw = make_float4(1.f, 1.f, 1.f, 1.f);
    int rk = 0;
    float* const dyo = (!POOLED && ci0 == 0) ? a.dY_out : nullptr;
    long dy_ofs = 0;                                 // element (first staging row of this thread, its 4 columns) of the loaded chunk
    auto load_chunk = [&](int chl) {
        const long chg = col0 / CP + chl;            // chunk index in the whole column space
        const long b = chg / chunks_per_b;
        const int p = (int)(chg - b * chunks_per_b) * CP + 4 * c4;
        dy_ofs = (b * a.Cout + co0 + r0) * (long)a.P + p;
        if (a.w) rw = *reinterpret_cast<const float4*>(&a.w[b * a.P + p]);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const long row = b * a.Cout + co0 + r0 + RPP * i;
            ry[i] = *reinterpret_cast<const float4*>(&a.Y[row * a.P + p]);
            if (POOLED) {
                const int j = p / a.ns;
                const float2 t = a.pk[row * np + j];
                rg[i].x = t.x; rg[i].y = t.y;
                rk = p - j * a.ns;
            } else {
                rg[i] = *reinterpret_cast<const float4*>(&a.dN[row * a.P + p]);
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i)
            rx[i] = *reinterpret_cast<const float4*>(&a.X[(b * a.Cin + ci0 + r0 + RPP * i) * a.P + p]);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            float4 g;
            if (POOLED) {
                const int ak = __float_as_int(rg[i].y);
                const float go = rg[i].x;
                g.x = (rk + 0 == ak) ? go : 0.f; g.y = (rk + 1 == ak) ? go : 0.f;
                g.z = (rk + 2 == ak) ? go : 0.f; g.w = (rk + 3 == ak) ? go : 0.f;
            } else {
                g = rg[i];
            }
            float4 o;
            o.x = fmaf(ka1[i], g.x, rw.x * fmaf(ka2[i], ry[i].x, ka3[i])); o.y = fmaf(ka1[i], g.y, rw.y * fmaf(ka2[i], ry[i].y, ka3[i]));
            o.z = fmaf(ka1[i], g.z, rw.z * fmaf(ka2[i], ry[i].z, ka3[i])); o.w = fmaf(ka1[i], g.w, rw.w * fmaf(ka2[i], ry[i].w, ka3[i]));
            *reinterpret_cast<float4*>(&As(buf)[(r0 + RPP * i) * LD + 4 * c4]) = o;
            if (dyo) *reinterpret_cast<float4*>(&dyo[dy_ofs + (long)(RPP * i) * a.P]) = o;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            float4 v = rx[i];
            v.x = fmaxf(fmaf(v.x, ksc[i], ksh[i]), x_floor); v.y = fmaxf(fmaf(v.y, ksc[i], ksh[i]), x_floor);
            v.z = fmaxf(fmaf(v.z, ksc[i], ksh[i]), x_floor); v.w = fmaxf(fmaf(v.w, ksc[i], ksh[i]), x_floor);
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

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
        __syncthreads();
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int t = ch - c_begin;
            if (ch + 1 < c_end) load_chunk(ch + 1);
            const float* A_ = As(t & 1) + wk * (CP / WK);
            const float* B_ = Bs(t & 1) + wk * (CP / WK);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float av[2][4], bv[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 x = *reinterpret_cast<const float4*>(&A_[(wm0 + 32 * u + l31) * LD + 8 * g + 4 * h]);
                    av[u][0] = x.x; av[u][1] = x.y; av[u][2] = x.z; av[u][3] = x.w;
                    const float4 y = *reinterpret_cast<const float4*>(&B_[(wn0 + 32 * u + l31) * LD + 8 * g + 4 * h]);
                    bv[u][0] = y.x; bv[u][1] = y.y; bv[u][2] = y.z; bv[u][3] = y.w;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = mfma32(av[tm][s], bv[tn][s], acc[tm][tn]);
            }
            if (ch + 1 < c_end) store_chunk((t + 1) & 1);
            __syncthreads();
        }
    }
    // every k-split wave writes its own partial slice (summed later in a fixed order)
    float* dst = a.part + (long)(slice * WK + wk) * a.Cout * a.Cin;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm0 + 32 * tm + acc_row(r, h);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) dst[(long)co * a.Cin + ci0 + wn0 + 32 * tn + l31] = acc[tm][tn][r];
        }
}

template <int TM, int TN, int POOLED>
__global__ __launch_bounds__(256) void wgrad2_kernel(Wgrad2Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    wgrad2_body<TM, TN, POOLED>(a, blockIdx.x, smem);
}

// Several independent weight gradients (dense dN, flat layout) in ONE launch: the weight gradients of a 1-D conv stack
// (models/head/rpn.py:16-39 ...) depend only on operands the data-gradient chain has already produced, and each of them is
// a sub-round launch (192 workgroups of 4 short chunks for a 256 x 256 x 6144 problem: 22 us for 6 us of matrix work, plus
// its slice reduction).  Grouped, their workgroups fill the chip together and overlap each other's staging latencies.
constexpr int WG_MAXJOBS = 8;
struct Wgrad2Group {
    Wgrad2Args job[WG_MAXJOBS];
    int first[WG_MAXJOBS + 1];     // first workgroup of every job
    int kind[WG_MAXJOBS];          // 0: <128,128>  1: <128,64>  2: <64,128>  3: <64,64>  4: row sums (bias gradient)
    int njobs;
};

// kind 4: out[c] = sum_p G[c, p] -- the bias gradient of a stack's last layer rides in the stack's grouped launch (it was a
// launch of its own, csrc/heads.hip::row_sum_kernel: same arithmetic, same order).  Wgrad2Args: dN = G (Cout rows of P
// columns), part = out.
__device__ __forceinline__ void row_sum_body(const Wgrad2Args& a, const int bid, float* smem) {
    const float* g = a.dN + (long)bid * a.P;
    float s = 0.f;
    for (long p = 4L * threadIdx.x; p < a.P; p += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(g + p);
        s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.part[bid] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

__global__ __launch_bounds__(256) void wgrad2_group_kernel(Wgrad2Group g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int j = 0;
    while (j + 1 < g.njobs && (int)blockIdx.x >= g.first[j + 1]) ++j;
    const int bid = blockIdx.x - g.first[j];
    switch (g.kind[j]) {
        case 0: wgrad2_body<128, 128, 0>(g.job[j], bid, smem); break;
        case 1: wgrad2_body<128, 64, 0>(g.job[j], bid, smem); break;
        case 2: wgrad2_body<64, 128, 0>(g.job[j], bid, smem); break;
        case 3: wgrad2_body<64, 64, 0>(g.job[j], bid, smem); break;
        default: row_sum_body(g.job[j], bid, smem); break;
    }
}

struct ReduceGroup {
    const float* part[WG_MAXJOBS]; float* out[WG_MAXJOBS];
    long n[WG_MAXJOBS]; int nslices[WG_MAXJOBS]; int first[WG_MAXJOBS + 1]; int njobs;
    // compact output: element (row, col) of the padded (rows_p, ld[j]) gradient goes to out[row * ocols[j] + col] when
    // row < orows[j] and col < ocols[j] -- the gradient autograd receives is then contiguous in the parameter's own
    // shape (a strided slice of the padded buffer made AccumulateGrad clone it: one more launch per padded layer)
    int ld[WG_MAXJOBS], orows[WG_MAXJOBS], ocols[WG_MAXJOBS];
};

// the slice reductions of a group in one launch: wgrad_reduce1_kernel's arithmetic (mlp.hip), same fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(ReduceGroup r) {
    __shared__ float sh[8][32];
    int j = 0;
    while (j + 1 < r.njobs && (int)blockIdx.x >= r.first[j + 1]) ++j;
    const float* __restrict__ part = r.part[j];
    const long n = r.n[j];
    const int nslices = r.nslices[j];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long i = (long)(blockIdx.x - r.first[j]) * 32 + col;
    const int per = (nslices + 7) / 8, z0 = grp * per;
    int z1 = z0 + per;
    if (z1 > nslices) z1 = nslices;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int z = z0;
        for (; z + 3 < z1; z += 4) {
            s0 += part[(long)z * n + i]; s1 += part[(long)(z + 1) * n + i];
            s2 += part[(long)(z + 2) * n + i]; s3 += part[(long)(z + 3) * n + i];
        }
        for (; z < z1; ++z) s0 += part[(long)z * n + i];
    }
    sh[grp][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += sh[g][col];
        const int ld = r.ld[j];
        const long row = i / ld;
        const int cc = (int)(i - row * ld);
        if (row < r.orows[j] && cc < r.ocols[j]) r.out[j][row * r.ocols[j] + cc] = t;
    }
}

template <int TM, int TN>
int launch_wgrad2(const Wgrad2Args& a, hipStream_t s) {
    constexpr int CP = (TM + TN == 128) ? 64 : 32;
    const size_t lds = sizeof(float) * 2 * (TM + TN) * (CP + 4);
    const void* fn = a.dN ? reinterpret_cast<const void*>(wgrad2_kernel<TM, TN, 0>)
                          : reinterpret_cast<const void*>(wgrad2_kernel<TM, TN, 1>);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return O3D_ELAUNCH;
    const int ntile = (a.Cout / TM) * (a.Cin / TN);
    if (a.dN) hipLaunchKernelGGL((wgrad2_kernel<TM, TN, 0>), dim3(ntile * a.nslices), dim3(256), lds, s, a);
    else      hipLaunchKernelGGL((wgrad2_kernel<TM, TN, 1>), dim3(ntile * a.nslices), dim3(256), lds, s, a);
    return o3d_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Data gradient + weight gradient of one aligned inner layer in ONE kernel (Cin == 64, Cout in {64, 128}; dense dN):
// one staging of dY = A1*dN + w*(A2*Y + A3) and Xt = relu(sc*Yprev + sh) in LDS feeds both MFMA chains, so a column
// costs (2*Cout + 2*Cin) * 4 B instead of the (4*Cout + 3*Cin) * 4 B of the wgrad2 + direct-dgrad pair.
//   weight gradient   dW[co][ci] += sum_pos dY[co][pos] * Xt[ci][pos]          (wgrad2_kernel's inner loop)
//   data gradient     dX[ci][pos]  = sum_co Wt[ci][co] * dY[co][pos], masked by Xt > 0: A fragments = Wt rows, constant
//                     over the launch, loaded ONCE into registers; B fragments = dY read back from the staging buffer
//   statistics        {sum g, sum g*(yprev-mean)} of the data gradient, the second from sum g*Xt:
//                     sum g*(yprev-mean) = (sum g*Xt - beta*sum g) / sc, beta = sh + sc*mean; one column per lane and
//                     chunk accumulated in 32 registers, folded over the lanes once at the end
// ~245 VGPRs + 80 AGPRs: ONE workgroup per CU, the whole next chunk (49-65 KB per CU) in flight while the current one is
// multiplied.  Measured on the MI355X at SA1's size (384 000 live columns, tools/exp/fused_bwd_check.py,
// profiles/r02_fused_bwd_experiment.txt): 64 -> 64 0.101 ms against 0.167 ms for the pair, 64 -> 128 0.22 against 0.23.
// Partial rows: part_w [nslices*WK][Cout][Cin] as wgrad2; part_s [2 segments][nslices*NPT][2][Cin] (a slice writes its sums
// into its own segment's block and zeros into the other).
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
        float s2 = sc != 0.f ? (s2x - beta * s1) / sc : 0.f;                        // sum g*(yprev - mean)
        if (sc == 0.f && which) {
            // gamma == 0 (zero-initialised / pruned channel): relu(bn(yprev)) is the constant beta, the sum cannot be
            // recovered from the staged value.  Rare, so the owning lane simply re-reads its row: the masked gradient this
            // wave stored above and the producer's raw output, over this wave's 32-position tile of every chunk of the slice
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float mu = in_mean[ci];
            float t2 = 0.f;
            for (int ch = c_begin; ch < c_end; ++ch) {
                const long qb = (col0 / CP + ch) * CP + 32 * dpt;
                for (int j = 0; j < 32; ++j)
                    t2 = fmaf(a.dX[(long)ci * a.P + qb + j], a.X[(long)ci * a.P + qb + j] - mu, t2);
            }
            s2 = t2;
        }
        const long rows = (long)a.nslices * NPT, row = (long)slice * NPT + dpt;
        float* mine = a.part_s + ((long)seg * rows + row) * 2 * TN;
        float* theirs = a.part_s + ((long)(1 - seg) * rows + row) * 2 * TN;
        mine[which * TN + ci] = which ? s2 : s1;
        theirs[which * TN + ci] = 0.f;
    }
}

template <int TM>
int launch_fused_bwd(const FusedBwdArgs& a, hipStream_t s) {
    constexpr int CP = (TM + 64 == 128) ? 64 : 32;
    const size_t lds = sizeof(float) * 2 * (TM + 64) * (CP + 4);
    const void* fn = reinterpret_cast<const void*>(fused_bwd_kernel<TM>);
    if (lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -3;
    hipLaunchKernelGGL((fused_bwd_kernel<TM>), dim3(a.nslices), dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

void o3d_wgrad_reduce(const float* part, int nslices, long n, float* scratch2, float* dW, hipStream_t s);

static void wgrad2_plan(int Cin, int Cout, int B, int P, int& TM, int& TN, int& WK, int& CP, int& nsl) {
    TM = Cout % 128 == 0 ? 128 : 64;
    TN = Cin % 128 == 0 ? 128 : 64;
    WK = 4 / ((TM / 64) * (TN / 64));
    CP = (TM + TN == 128) ? 64 : 32;
    const int ntile = (Cout / TM) * (Cin / TN);
    const long total = (long)B * (P / CP);
    constexpr int wgs = 512;          // (256 / 384 / 1024 workgroups measured in rounds 1-2: 7.65 / +0.14 / 7.87 vs 7.68 ms per step)
    nsl = wgs / ntile;                 // 512: one resident round, 2 workgroups per CU x 256 CUs
    if (nsl < 8) nsl = 8;
    while (nsl > 8 && total / nsl < 4) nsl -= 8;
    if (nsl > total) nsl = (int)total;
}

// floats of scratch needed by o3d_mlp_conv_wgrad2 (partial tiles + second-stage groups)
extern "C" long o3d_mlp_conv_wgrad2_scratch(int B, int Cin, int Cout, int P) {
    if (Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64 || P <= 0 || P % 64 || B <= 0) return -1;
    int TM, TN, WK, CP, nsl;
    wgrad2_plan(Cin, Cout, B, P, TM, TN, WK, CP, nsl);
    return ((long)nsl * WK + 16) * Cout * Cin;
}

// dW (Cout,Cin) = sum_{b,p} dY * f(X); dY from dN (dense) or pk (pooled, needs ns); X raw producer output
// with (in_scale,in_shift) (both NULL: X as stored).  Cin, Cout, P multiples of 64 (the staged chunks are 32 / 64 columns).
static int wgrad2_impl(const float* dN, const float* pk, int ns, const float* Y, const float* A1, const float* A2,
                       const float* A3, const float* X, const float* in_scale, const float* in_shift, int B, int Cin,
                       int Cout, int P, const float* w, const int32_t* meta, long start1, float* scratch, float* dW,
                       void* stream, float* dY_out = nullptr);

extern "C" int o3d_mlp_conv_wgrad2(const float* dN, const float* pk, int ns, const float* Y, const float* A1,
                                   const float* A2, const float* A3, const float* X, const float* in_scale,
                                   const float* in_shift, int B, int Cin, int Cout, int P, float* scratch,
                                   float* dW, void* stream) {
    return wgrad2_impl(dN, pk, ns, Y, A1, A2, A3, X, in_scale, in_shift, B, Cin, Cout, P, nullptr, nullptr, 0, scratch,
                       dW, stream);
}

// compact layout (csrc/compact.hip): flat (C, ldp) operands, weights w (ldp), live columns meta[0]
extern "C" int o3d_mlp_conv_wgrad2_c(const float* dN, const float* Y, const float* A1, const float* A2,
                                     const float* A3, const float* X, const float* in_scale, const float* in_shift,
                                     int Cin, int Cout, long ldp, const float* w, const int32_t* meta, long start1,
                                     float* scratch, float* dW, void* stream) {
    if (!dN || !w || !meta || ldp <= 0 || ldp > 0x7fffffff || start1 < 0 || start1 % 256 != 0) return O3D_EINVAL;
    return wgrad2_impl(dN, nullptr, 4, Y, A1, A2, A3, X, in_scale, in_shift, 1, Cin, Cout, (int)ldp, w, meta, start1,
                       scratch, dW, stream);
}

// the same launch, which also writes the operand it stages -- dY = A1*dN + w*(A2*Y + A3), (Cout, ldp), live columns only --
// so that o3d_mlp_conv_dgrad_c can be given ONE tensor (dN = dY, Y = NULL) instead of rebuilding dY from two
extern "C" int o3d_mlp_conv_wgrad2_c_dy(const float* dN, const float* Y, const float* A1, const float* A2,
                                        const float* A3, const float* X, const float* in_scale, const float* in_shift,
                                        int Cin, int Cout, long ldp, const float* w, const int32_t* meta, long start1,
                                        float* scratch, float* dW, float* dY, void* stream) {
    if (!dN || !w || !meta || !dY || ldp <= 0 || ldp > 0x7fffffff || start1 < 0 || start1 % 256 != 0) return O3D_EINVAL;
    return wgrad2_impl(dN, nullptr, 4, Y, A1, A2, A3, X, in_scale, in_shift, 1, Cin, Cout, (int)ldp, w, meta, start1,
                       scratch, dW, stream, dY);
}

static int wgrad2_impl(const float* dN, const float* pk, int ns, const float* Y, const float* A1, const float* A2,
                       const float* A3, const float* X, const float* in_scale, const float* in_shift, int B, int Cin,
                       int Cout, int P, const float* w, const int32_t* meta, long start1, float* scratch, float* dW,
                       void* stream, float* dY_out) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64 || P <= 0 || P % 64 || !Y || !A1 || !A2 || !A3 ||
        !X || (in_scale == nullptr) != (in_shift == nullptr) || !scratch || !dW || (!dN && (!pk || ns < 4 || ns % 4)))
        return O3D_EINVAL;
    int TM, TN, WK, CP, nsl;
    wgrad2_plan(Cin, Cout, B, P, TM, TN, WK, CP, nsl);
    Wgrad2Args a = {};
    a.dN = dN; a.pk = reinterpret_cast<const float2*>(pk); a.ns = ns > 0 ? ns : 4; a.Y = Y;
    a.A1 = A1; a.A2 = A2; a.A3 = A3; a.X = X; a.in_scale = in_scale; a.in_shift = in_shift;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.P = P;
    a.total_chunks = B * (P / CP);
    a.chunks_per_block = (a.total_chunks + nsl - 1) / nsl;
    a.nslices = nsl;
    a.part = scratch;
    a.w = w; a.meta = meta; a.start1 = start1; a.dY_out = dY_out;
    hipStream_t s = o3d_stream(stream);
    int rc;
    if (TM == 128 && TN == 128) rc = launch_wgrad2<128, 128>(a, s);
    else if (TM == 128)         rc = launch_wgrad2<128, 64>(a, s);
    else if (TN == 128)         rc = launch_wgrad2<64, 128>(a, s);
    else                        rc = launch_wgrad2<64, 64>(a, s);
    if (rc != O3D_OK) return rc;
    const long n = (long)Cout * Cin;
    o3d_wgrad_reduce(scratch, nsl * WK, n, scratch + (long)nsl * WK * n, dW, s);
    return o3d_launch_status();
}


// Up to WG_MAXJOBS independent weight gradients of the flat (C, P) layout in one launch + one reduction launch: job i is
// o3d_mlp_conv_wgrad2(dN, NULL, 4, Y, A1, A2, A3, X, in_scale, in_shift, 1, Cin, Cout, P, scratch, dW) (dense dN; Y == dN
// with A = (1, 0, 0) for a plain layer), scratch sized by o3d_mlp_conv_wgrad2_scratch(1, Cin, Cout, P).  Same numbers as
// the single launches (same tile plan, same slice order).
extern "C" int o3d_mlp_conv_wgrad2_group(const o3d_wgrad_job* jobs, int njobs, void* stream) {
    if (!jobs || njobs < 1 || njobs > WG_MAXJOBS) return O3D_EINVAL;
    Wgrad2Group g = {};
    ReduceGroup r = {};
    g.njobs = r.njobs = njobs;
    size_t lds = 0;
    int nblk = 0, rblk = 0;
    for (int i = 0; i < njobs; ++i) {
        const o3d_wgrad_job& q = jobs[i];
        if (q.dN && !q.X && !q.Y && q.dW && q.Cout > 0 && q.P > 0 && q.P % 4 == 0 && q.P <= 0x7fffffff) {
            // a row-sum job (the bias gradient): dW (Cout) = row sums of dN (Cout, P); no reduction stage
            Wgrad2Args& a = g.job[i];
            a.dN = q.dN; a.P = (int)q.P; a.Cout = q.Cout; a.part = q.dW;
            g.kind[i] = 4;
            g.first[i] = nblk;
            nblk += q.Cout;
            r.part[i] = q.dW; r.out[i] = q.dW; r.n[i] = 0; r.nslices[i] = 0; r.ld[i] = 1; r.orows[i] = r.ocols[i] = 0;
            r.first[i] = rblk;
            continue;
        }
        if (!q.dN || !q.Y || !q.A1 || !q.A2 || !q.A3 || !q.X || (q.in_scale == nullptr) != (q.in_shift == nullptr) ||
            !q.scratch || !q.dW || q.Cin <= 0 || q.Cout <= 0 || q.Cin % 64 || q.Cout % 64 || q.P <= 0 || q.P % 64 ||
            q.P > 0x7fffffff)
            return O3D_EINVAL;
        int TM, TN, WK, CP, nsl;
        wgrad2_plan(q.Cin, q.Cout, 1, (int)q.P, TM, TN, WK, CP, nsl);
        Wgrad2Args& a = g.job[i];
        a.dN = q.dN; a.ns = 4; a.Y = q.Y; a.A1 = q.A1; a.A2 = q.A2; a.A3 = q.A3; a.X = q.X;
        a.in_scale = q.in_scale; a.in_shift = q.in_shift; a.B = 1; a.Cin = q.Cin; a.Cout = q.Cout; a.P = (int)q.P;
        a.total_chunks = (int)(q.P / CP);
        a.chunks_per_block = (a.total_chunks + nsl - 1) / nsl;
        a.nslices = nsl;
        a.part = q.scratch;
        g.kind[i] = TM == 128 ? (TN == 128 ? 0 : 1) : (TN == 128 ? 2 : 3);
        g.first[i] = nblk;
        nblk += (q.Cout / TM) * (q.Cin / TN) * nsl;
        const size_t l = sizeof(float) * 2 * (TM + TN) * (CP + 4);
        lds = l > lds ? l : lds;
        if (q.out_rows < 0 || q.out_rows > q.Cout || q.out_cols < 0 || q.out_cols > q.Cin) return O3D_EINVAL;
        r.part[i] = q.scratch; r.out[i] = q.dW; r.n[i] = (long)q.Cout * q.Cin; r.nslices[i] = nsl * WK;
        r.ld[i] = q.Cin; r.orows[i] = q.out_rows > 0 ? q.out_rows : q.Cout; r.ocols[i] = q.out_cols > 0 ? q.out_cols : q.Cin;
        r.first[i] = rblk;
        rblk += o3d_cdiv(r.n[i], 32);
    }
    g.first[njobs] = nblk;
    r.first[njobs] = rblk;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad2_group_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    if (!attr_ok || lds > 80 * 1024) return O3D_ELAUNCH;
    hipStream_t s = o3d_stream(stream);
    if (lds < 16) lds = 16;            // (row-sum jobs only: four floats of staging)
    hipLaunchKernelGGL(wgrad2_group_kernel, dim3(nblk), dim3(256), lds, s, g);
    if (rblk > 0) hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3(rblk), dim3(256), 0, s, r);
    return o3d_launch_status();
}


// ---- fused data + weight gradient (fused_bwd_kernel above) ----------------------------------------------------------
static int fused_bwd_slices(int Cout, long P) {
    constexpr int wgs = 256;          // one workgroup per CU (~305 registers)
    const int CP = (Cout + 64 == 128) ? 64 : 32;
    const long total = P / CP;
    int nsl = wgs;
    while (nsl > 8 && total / nsl < 4) nsl -= 8;
    if (nsl > total) nsl = (int)total;
    return nsl < 1 ? 1 : nsl;
}
static bool fused_bwd_ok(int Cin, int Cout, long P) {
    return Cin == 64 && (Cout == 64 || Cout == 128) && P > 0 && P % 64 == 0 && P <= 0x7fffffff;
}

// rows per segment block of part_s (-1: shape not supported), floats of weight-gradient partial tiles in `scratch`
extern "C" int o3d_mlp_conv_bwd_fused_rows(int Cin, int Cout, long P) {
    return fused_bwd_ok(Cin, Cout, P) ? fused_bwd_slices(Cout, P) * (Cout == 64 ? 2 : 1) : -1;
}
extern "C" long o3d_mlp_conv_bwd_fused_scratch(int Cin, int Cout, long P) {
    return fused_bwd_ok(Cin, Cout, P) ? (long)fused_bwd_slices(Cout, P) * (4 / (Cout / 64)) * Cout * Cin : -1;
}

// dW (Cout,Cin), dNprev (Cin,P) and the BatchNorm-backward partials part_s [2][rows][2][Cin] of dNprev in one launch
// (+ the fixed-order reduction of the weight-gradient partial tiles); operands as o3d_mlp_conv_wgrad2_c /
// o3d_mlp_conv_dgrad_c: A1..A3 (nseg,Cout), in_scale / in_shift / in_mean (nseg,Cin) of the producer layer, Wt (Cin,Cout).
extern "C" int o3d_mlp_conv_bwd_fused_c(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3,
                                        const float* X, const float* in_scale, const float* in_shift, const float* in_mean,
                                        const float* Wt, int Cin, int Cout, long ldp, const float* w, const int32_t* meta,
                                        long start1, float* scratch, float* dW, float* part_s, float* dNprev, void* stream) {
    if (!dN || !Y || !A1 || !A2 || !A3 || !X || !in_scale || !in_shift || !in_mean || !Wt || !scratch || !dW || !part_s ||
        !dNprev || !fused_bwd_ok(Cin, Cout, ldp) || (meta != nullptr) != (w != nullptr) || start1 < 0 || start1 % 256 != 0)
        return O3D_EINVAL;
    const int CP = (Cout + 64 == 128) ? 64 : 32;
    const int nsl = fused_bwd_slices(Cout, ldp);
    FusedBwdArgs a = {};
    a.dN = dN; a.Y = Y; a.A1 = A1; a.A2 = A2; a.A3 = A3; a.X = X; a.in_scale = in_scale; a.in_shift = in_shift;
    a.in_mean = in_mean; a.Wt = Wt; a.Cin = Cin; a.Cout = Cout; a.P = (int)ldp;
    a.total_chunks = (int)(ldp / CP);
    a.chunks_per_block = (a.total_chunks + nsl - 1) / nsl;
    a.nslices = nsl;
    a.w = w; a.meta = meta; a.start1 = start1; a.part_w = scratch; a.part_s = part_s; a.dX = dNprev;
    hipStream_t s = o3d_stream(stream);
    const int rc = Cout == 64 ? launch_fused_bwd<64>(a, s) : launch_fused_bwd<128>(a, s);
    if (rc != 0) return O3D_ELAUNCH;
    const long n = (long)Cout * Cin;
    o3d_wgrad_reduce(scratch, nsl * (4 / (Cout / 64)), n, nullptr, dW, s);
    return o3d_launch_status();
}
