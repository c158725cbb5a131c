// BoxCloud: the distance of every point to the centre and the eight corners of its target box.
//
// Restates datasets/points_utils.py:127-143 (get_point_to_box_distance: scipy cdist of the points
// against [centre | Box.corners()]) and datasets/data_classes.py:226-250 (Box.corners: +-l/2, +-w/2,
// +-h/2 in the order x = l/2*[1,1,1,1,-1,-1,-1,-1], y = w/2*[1,-1,-1,1,1,-1,-1,1],
// z = h/2*[1,1,-1,-1,1,1,-1,-1], rotated by the box orientation, translated by the centre).
// The reference runs this in numpy (fp64) inside DataLoader workers and, at inference, on the
// critical path of every frame (models/bat.py:41-55); here it is one launch on the clouds that are
// already resident: 12 B in, 36 B out per point -- pure HBM streaming.
#include "o3d_common.hpp"

namespace {

__global__ __launch_bounds__(256) void boxcloud_kernel(const float* __restrict__ points,
                                                       const float* __restrict__ center,
                                                       const float* __restrict__ wlh,
                                                       const float* __restrict__ rot, float wlh_factor, int N,
                                                       float* __restrict__ out) {
    __shared__ float lm[9][3];       // landmarks: centre, then the 8 corners
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid < 9) {
        const float cx = center[3 * b], cy = center[3 * b + 1], cz = center[3 * b + 2];
        if (tid == 0) {
            lm[0][0] = cx; lm[0][1] = cy; lm[0][2] = cz;
        } else {
            const int k = tid - 1;
            const float w = wlh[3 * b] * wlh_factor, l = wlh[3 * b + 1] * wlh_factor, h = wlh[3 * b + 2] * wlh_factor;
            const float x = 0.5f * l * (k < 4 ? 1.f : -1.f);
            const float y = 0.5f * w * (((k & 3) == 0 || (k & 3) == 3) ? 1.f : -1.f);
            const float z = 0.5f * h * ((k & 2) ? -1.f : 1.f);
            const float* R = rot + 9 * b;
            lm[tid][0] = R[0] * x + R[1] * y + R[2] * z + cx;
            lm[tid][1] = R[3] * x + R[4] * y + R[5] * z + cy;
            lm[tid][2] = R[6] * x + R[7] * y + R[8] * z + cz;
        }
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    if (n >= N) return;
    const float* p = points + ((long)b * N + n) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    float* o = out + ((long)b * N + n) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float dx = px - lm[k][0], dy = py - lm[k][1], dz = pz - lm[k][2];
        o[k] = sqrtf(dx * dx + dy * dy + dz * dz);
    }
}

}  // namespace

// points (B,N,3), center (B,3), wlh (B,3) = width, length, height, rot (B,3,3) row-major rotation matrix of
// the box orientation -> out (B,N,9): distance to the centre (channel 0) and to corner k (channel 1+k).
extern "C" int o3d_boxcloud(const float* points, const float* center, const float* wlh, const float* rot,
                            float wlh_factor, int B, int N, float* out, void* stream) {
    if (B < 0 || N < 0 || (B > 0 && N > 0 && (!points || !center || !wlh || !rot || !out))) return O3D_EINVAL;
    if (B == 0 || N == 0) return O3D_OK;
    hipLaunchKernelGGL(boxcloud_kernel, dim3(o3d_cdiv(N, 256), B), dim3(256), 0, o3d_stream(stream), points, center, wlh,
                       rot, wlh_factor, N, out);
    return o3d_launch_status();
}

// ---- M2-Track, between its two stages (models/m2track.py:120-137) ----------------------------------------------------------
// The reference chains datasets/points_utils.py's tensor helpers on the masked points: get_offset_box_tensor (:420-436) makes
// the first-stage box `aux` = previous box moved by the motion; get_offset_points_tensor (:390-417) carries the previous
// frame's half of the points along that motion; remove_transform_points_tensor (:439-452) expresses both halves in the frame
// of `aux`.  ~45 launches forward (rotation matrices, batched 3x3 GEMMs, concatenations) and twice that backward, on 12 B per
// point.  Here: one launch each way.  Forward follows the reference's chain step by step; the backward uses that z rotations
// commute, which collapses the chain to  y = Rz(-prev_t)(x - prev_c)  for the previous half and  y = Rz(-aux_t)(x - aux_c)
// for the current half (the same function in exact arithmetic), and reduces the four per-cloud sums in one workgroup.
namespace {

struct Rz { float c, s; };
__device__ __forceinline__ Rz rz(float t) { Rz r; sincosf(t, &r.s, &r.c); return r; }
// (x, y) rotated by the angle of r; inv: by minus that angle
__device__ __forceinline__ void rot2(const Rz& r, float x, float y, float& ox, float& oy) { ox = r.c * x - r.s * y; oy = r.s * x + r.c * y; }
__device__ __forceinline__ void rot2i(const Rz& r, float x, float y, float& ox, float& oy) { ox = r.c * x + r.s * y; oy = -r.s * x + r.c * y; }

struct MotionArgs {
    const float* pts; long bstride, cstride;   // channel c of point n of cloud b at pts[b * bstride + c * cstride + n]
    const float* prev;                         // (B,4) previous box or NULL (zeros)
    const float* motion;                       // (B,4)
    int B, N;
};

__global__ __launch_bounds__(256) void motion_merge_fwd_kernel(MotionArgs a, float* __restrict__ merged, float* __restrict__ aux) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    float pc[3] = {0.f, 0.f, 0.f}, pt = 0.f;
    if (a.prev) { pc[0] = a.prev[4 * b]; pc[1] = a.prev[4 * b + 1]; pc[2] = a.prev[4 * b + 2]; pt = a.prev[4 * b + 3]; }
    const float mc[3] = {a.motion[4 * b], a.motion[4 * b + 1], a.motion[4 * b + 2]}, mt = a.motion[4 * b + 3];
    const Rz rp = rz(pt), rm = rz(mt);
    float ac[3], at = pt + mt;
    rot2(rp, mc[0], mc[1], ac[0], ac[1]);
    ac[0] += pc[0]; ac[1] += pc[1]; ac[2] = mc[2] + pc[2];
    const Rz ra = rz(at);
    if (n == 0) { aux[4 * b] = ac[0]; aux[4 * b + 1] = ac[1]; aux[4 * b + 2] = ac[2]; aux[4 * b + 3] = at; }
    if (n >= a.N) return;
    const float* p = a.pts + (long)b * a.bstride + n;
    float x = p[0], y = p[a.cstride], z = p[2 * a.cstride];
    if (n < a.N / 2) {          // previous frame's points: into the previous box, along the motion, back to the world
        float qx, qy, rx, ry;
        rot2i(rp, x - pc[0], y - pc[1], qx, qy);
        rot2(rm, qx, qy, rx, ry);
        rx += mc[0]; ry += mc[1];
        const float rzz = (z - pc[2]) + mc[2];
        rot2(rp, rx, ry, x, y);
        x += pc[0]; y += pc[1]; z = rzz + pc[2];
    }
    float ox, oy;
    rot2i(ra, x - ac[0], y - ac[1], ox, oy);
    float* o = merged + (long)b * 3 * a.N + n;
    o[0] = ox; o[a.N] = oy; o[2 * (long)a.N] = z - ac[2];
}

// per cloud: S = sum gy, T = sum gy . d/dtheta [Rz(-theta)(x - c)] over each half; then the chain through aux = prev (+) motion
__global__ __launch_bounds__(256) void motion_merge_bwd_kernel(MotionArgs a, const float* __restrict__ gy,
                                                               const float* __restrict__ g_aux, float* __restrict__ g_prev,
                                                               float* __restrict__ g_motion) {
    __shared__ float red[4][8];
    const int b = blockIdx.x, tid = threadIdx.x;
    float pc[3] = {0.f, 0.f, 0.f}, pt = 0.f;
    if (a.prev) { pc[0] = a.prev[4 * b]; pc[1] = a.prev[4 * b + 1]; pc[2] = a.prev[4 * b + 2]; pt = a.prev[4 * b + 3]; }
    const float mc[3] = {a.motion[4 * b], a.motion[4 * b + 1], a.motion[4 * b + 2]}, mt = a.motion[4 * b + 3];
    const Rz rp = rz(pt);
    float ac[3], at = pt + mt;
    rot2(rp, mc[0], mc[1], ac[0], ac[1]);
    ac[0] += pc[0]; ac[1] += pc[1]; ac[2] = mc[2] + pc[2];
    const Rz ra = rz(at);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // half 0: S (3), T; half 1: S (3), T
    for (int n = tid; n < a.N; n += 256) {
        const int h = n >= a.N / 2;
        const float* p = a.pts + (long)b * a.bstride + n;
        const float* g = gy + (long)b * 3 * a.N + n;
        const float gx = g[0], gyy = g[a.N], gz = g[2 * (long)a.N];
        const Rz& r = h ? ra : rp;
        const float* c = h ? ac : pc;
        const float dx = p[0] - c[0], dy = p[a.cstride] - c[1];
        // y = (c dx + s dy, -s dx + c dy): d/dtheta = (-s dx + c dy, -c dx - s dy)
        const float tx = -r.s * dx + r.c * dy, ty = -r.c * dx - r.s * dy;
        s[4 * h + 0] += gx; s[4 * h + 1] += gyy; s[4 * h + 2] += gz; s[4 * h + 3] += gx * tx + gyy * ty;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[k] += __shfl_xor(s[k], off, 64);
    if ((tid & 63) == 0)
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid >> 6][k] = s[k];
    __syncthreads();
    if (tid != 0) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    // y = Rz(-theta)(x - c): dL/dc = -Rz(theta) S (z: -S_z), dL/dtheta = T
    float gpc[3], gac[3], gpt = s[3], gat = s[7];
    rot2(rp, -s[0], -s[1], gpc[0], gpc[1]); gpc[2] = -s[2];
    rot2(ra, -s[4], -s[5], gac[0], gac[1]); gac[2] = -s[6];
    if (g_aux) { gac[0] += g_aux[4 * b]; gac[1] += g_aux[4 * b + 1]; gac[2] += g_aux[4 * b + 2]; gat += g_aux[4 * b + 3]; }
    // aux_c = Rz(prev_t) motion_c + prev_c, aux_t = prev_t + motion_t
    float gm[2];
    rot2i(rp, gac[0], gac[1], gm[0], gm[1]);
    g_motion[4 * b] = gm[0]; g_motion[4 * b + 1] = gm[1]; g_motion[4 * b + 2] = gac[2]; g_motion[4 * b + 3] = gat;
    if (g_prev) {
        // d/dprev_t [Rz(prev_t) mc] = (-s mx - c my, c mx - s my)
        const float dax = -rp.s * mc[0] - rp.c * mc[1], day = rp.c * mc[0] - rp.s * mc[1];
        g_prev[4 * b] = gpc[0] + gac[0]; g_prev[4 * b + 1] = gpc[1] + gac[1]; g_prev[4 * b + 2] = gpc[2] + gac[2];
        g_prev[4 * b + 3] = gpt + gat + gac[0] * dax + gac[1] * day;
    }
}

// get_offset_box_tensor (datasets/points_utils.py:420-436) on its own: box = ref moved by `off` given in ref's frame;
// thread per box.  g_box == NULL: forward (box out), else backward (g_ref / g_off out, either may be NULL).
__global__ __launch_bounds__(64) void offset_box_kernel(const float* __restrict__ ref, const float* __restrict__ off, int B,
                                                        float* __restrict__ box, const float* __restrict__ g_box,
                                                        float* __restrict__ g_ref, float* __restrict__ g_off) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float rt = ref[4 * b + 3], ox = off[4 * b], oy = off[4 * b + 1];
    const Rz r = rz(rt);
    if (!g_box) {
        float cx, cy;
        rot2(r, ox, oy, cx, cy);
        box[4 * b] = cx + ref[4 * b]; box[4 * b + 1] = cy + ref[4 * b + 1]; box[4 * b + 2] = off[4 * b + 2] + ref[4 * b + 2];
        box[4 * b + 3] = rt + off[4 * b + 3];
        return;
    }
    const float gx = g_box[4 * b], gy = g_box[4 * b + 1], gz = g_box[4 * b + 2], gt = g_box[4 * b + 3];
    if (g_ref) {
        g_ref[4 * b] = gx; g_ref[4 * b + 1] = gy; g_ref[4 * b + 2] = gz;
        g_ref[4 * b + 3] = gt + gx * (-r.s * ox - r.c * oy) + gy * (r.c * ox - r.s * oy);
    }
    if (g_off) {
        float mx, my;
        rot2i(r, gx, gy, mx, my);
        g_off[4 * b] = mx; g_off[4 * b + 1] = my; g_off[4 * b + 2] = gz; g_off[4 * b + 3] = gt;
    }
}

}  // namespace

// ref (B,4), off (B,4) -> box (B,4) [g_box == NULL]; or g_box (B,4) -> g_ref | NULL, g_off | NULL
extern "C" int o3d_offset_box(const float* ref, const float* off, int B, float* box, const float* g_box, float* g_ref, float* g_off,
                              void* stream) {
    if (!ref || !off || B <= 0 || (!g_box && !box)) return O3D_EINVAL;
    hipLaunchKernelGGL(offset_box_kernel, dim3(o3d_cdiv(B, 64)), dim3(64), 0, o3d_stream(stream), ref, off, B, box, g_box, g_ref, g_off);
    return o3d_launch_status();
}

// pts: the masked points, channel-major per cloud (xyz = channels 0..2; the first N/2 points belong to the previous frame);
// prev (B,4) | NULL (no previous-box refinement: zeros), motion (B,4) -> merged (B,3,N) in the frame of the first-stage box,
// aux (B,4) the first-stage box
extern "C" int o3d_motion_merge_fwd(const float* pts, long bstride, long cstride, const float* prev, const float* motion, int B,
                                    int N, float* merged, float* aux, void* stream) {
    if (!pts || !motion || !merged || !aux || B <= 0 || N <= 0 || N % 2 != 0) return O3D_EINVAL;
    MotionArgs a{pts, bstride, cstride, prev, motion, B, N};
    hipLaunchKernelGGL(motion_merge_fwd_kernel, dim3(o3d_cdiv(N, 256), B), dim3(256), 0, o3d_stream(stream), a, merged, aux);
    return o3d_launch_status();
}

// g_merged (B,3,N), g_aux (B,4) | NULL -> g_prev (B,4) | NULL, g_motion (B,4)
extern "C" int o3d_motion_merge_bwd(const float* pts, long bstride, long cstride, const float* prev, const float* motion, int B,
                                    int N, const float* g_merged, const float* g_aux, float* g_prev, float* g_motion, void* stream) {
    if (!pts || !motion || !g_merged || !g_motion || B <= 0 || N <= 0 || N % 2 != 0) return O3D_EINVAL;
    MotionArgs a{pts, bstride, cstride, prev, motion, B, N};
    hipLaunchKernelGGL(motion_merge_bwd_kernel, dim3(B), dim3(256), 0, o3d_stream(stream), a, g_merged, g_aux, g_prev, g_motion);
    return o3d_launch_status();
}
