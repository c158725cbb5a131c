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

static inline int o3d_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
