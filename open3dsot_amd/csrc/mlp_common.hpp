// mlp_common.hpp -- pieces shared by the fp32-MFMA kernels (mlp.hip, mlp_direct.hip)
#pragma once
#include "o3d_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// accumulator register r of a 32x32 tile holds row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Reduce-scatter of 32 per-lane values over the 32 lanes of each half-wave: afterwards lane
// (l&31) holds in x[0] the sum over those 32 lanes of the value with index (l&31).
__device__ __forceinline__ void reduce_scatter32(float (&x)[32], int l31) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        const bool up = (l31 & m) != 0;
#pragma unroll
        for (int v = 0; v < m; ++v) {
            // operands made opaque first: otherwise LLVM rewrites select(up, x[v+m], x[v]) into a
            // dynamically indexed extract of the 32-wide array (a 32-way v_cndmask chain per access)
            float lo = x[v], hi = x[v + m];
            asm volatile("" : "+v"(lo), "+v"(hi));
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            x[v] = keep + __shfl_xor(send, m, 64);
        }
    }
}

// ---- remainder tiles of a compact-layout GEMM launch in finer units (round 5) ---------------------------------------------
// A launch of T live 128-column tiles on S resident workgroup slots runs floor(T / S) full rounds and a last round that holds
// only R = T mod S tiles but lasts as long as a full one (profiles/r05_sq_dense_vs_compact.txt: the whole gap between the
// compact launches and the dense shapes is this tail).  The last R tiles are therefore cut into f = 4 (R <= S/4) or 2 (R <= S/2)
// column blocks of 32 / 64 columns, each a workgroup of the same launch: the tail round is then a quarter / half as long.
// Decided on the DEVICE from the live count (no host sync) by the GEMM and, identically, by the BatchNorm finalize that reads
// the launch's statistics rows: a tail tile's block 0 writes the tile's own row, its blocks 1..f-1 write extra rows behind the
// launch's regular rows (extra_row0 + segment * S + tile_in_tail * (f - 1) + block - 1).  `cap` = the segment's worst-case
// tile count (= workgroups the grid holds for it): no split when the extra workgroups would not fit.
struct TailPlan { int full, R, f; };
__host__ __device__ __forceinline__ TailPlan tail_plan(int T, int S, int cap) {
    TailPlan p = {T, 0, 1};
    if (S <= 0 || T <= 0) return p;
    const int R = T % S;
    if (R == 0) return p;
    const int f = 4 * R <= S ? 4 : (2 * R <= S ? 2 : 1);
    if (f == 1 || T + R * (f - 1) > cap) return p;
    p.full = T - R; p.R = R; p.f = f;
    return p;
}
// resident column-tile slots of direct_gemm_kernel<WAVES, *, *, 4, 2> for M output rows (2 waves per SIMD, 4 SIMDs per CU:
// 8 / WAVES workgroups per CU, M / (64 * WAVES) of them per column tile); 0: no tail split for this shape.  Host side.
extern "C" int o3d_direct_tail_slots(int M);
