// o3d_common.hpp -- shared helpers for the gfx950 kernels of libo3dsot_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/o3dsot.h"

#define O3D_WAVE 64

static inline int o3d_launch_status() {
    return hipGetLastError() == hipSuccess ? O3D_OK : O3D_ELAUNCH;
}

static inline hipStream_t o3d_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Canonical squared distance: the FMA contraction of the left-associated upstream
// expression dx*dx + dy*dy + dz*dz; the oracle uses the same chain.
__device__ __forceinline__ float o3d_sqdist3(float ax, float ay, float az, float bx, float by,
                                             float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

// Farthest-point sampling's near-origin rule (Appendix A.1): upstream writes `if (mag <= 1e-3) continue;` -- the float
// magnitude against the DOUBLE literal.  (float)1e-3 = 0x3A83126F lies above the double 0.001, so that one float is kept
// (`mag <= 1e-3f` would skip it).  The compare is spelled as upstream spells it; one cvt + one fp64 compare per point.
__device__ __forceinline__ bool o3d_fps_near_origin(float mag) { return (double)mag <= 1e-3; }

static inline int o3d_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- cross-lane moves on the DPP path (no LDS round trip like ds_bpermute / __shfl_xor) ---------------------------
// CTRL: 0xB1 quad_perm [1,0,3,2] (lane ^ 1), 0x4E quad_perm [2,3,0,1] (lane ^ 2), 0x141 row_half_mirror (lane -> 7 - lane
// within 8), 0x140 row_mirror (lane -> 15 - lane within 16)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }

// sum over the 64 lanes of a wave, the same value in every lane: four DPP steps inside each row of 16, then the four
// row sums through v_readlane (fixed order: deterministic)
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
