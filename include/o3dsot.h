/*
 * o3dsot.h -- C-ABI of libo3dsot_hip.so, the MI355X (gfx950) native replacement for the
 * `pointnet2_ops._ext` operator set that Open3DSOT's hot path calls.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer (HIP), row-major contiguous, fp32 / int32;
 *   - the call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream): no allocation, no host synchronisation, caller owns every buffer;
 *   - return value: O3D_OK (0) or a negative O3D_E* code; nothing is thrown;
 *   - stateless and thread-safe.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * Open3DSOT tree).  The reference binds these through the pybind module
 * `pointnet2_ops._ext` (pointnet2/utils/pointnet2_utils.py:17); INTEGRATION.md shows the
 * ctypes stub that re-creates that module on top of this header.
 */
#ifndef O3DSOT_H_
#define O3DSOT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define O3D_OK 0
#define O3D_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define O3D_ELAUNCH (-2)  /* the HIP runtime refused a launch or memset     */

/* Library identification: returns a static string "o3dsot-hip <ver> gfx950". */
const char* o3d_version(void);

/* ---- furthest point sampling -------------------------------------------------------
 * replaces _ext.furthest_point_sampling(xyz, npoint)      pointnet2_utils.py:56
 * xyz (B,N,3) f32 -> idx (B,npoint) i32.  Iterative FPS from index 0; points with
 * |p|^2 <= 1e-3 are never selected; tie order identical to the upstream thread-block
 * reduction (block = opt_n_threads(N)).  `temp` is a (B,N) f32 scratch that is only
 * touched when N > 16384 (may be NULL otherwise). */
int o3d_furthest_point_sampling(const float* xyz, int B, int N, int npoint, float* temp,
                                int32_t* idx, void* stream);

/* ---- gather ---------------------------------------------------------------------------
 * replaces _ext.gather_points(features, idx)              pointnet2_utils.py:92
 * feats (B,C,N), idx (B,npoint) -> out (B,C,npoint) */
int o3d_gather_points(const float* feats, const int32_t* idx, int B, int C, int N, int npoint,
                      float* out, void* stream);
/* replaces _ext.gather_points_grad(grad_out, idx, N)      pointnet2_utils.py:98
 * grad_out (B,C,npoint) -> grad_feats (B,C,N); grad_feats is zero-filled by the call. */
int o3d_gather_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N,
                           int npoint, float* grad_feats, void* stream);

/* ---- ball query -----------------------------------------------------------------------
 * replaces _ext.ball_query(new_xyz, xyz, radius, nsample) pointnet2_utils.py:268
 * new_xyz (B,npoint,3), xyz (B,N,3) -> idx (B,npoint,nsample) i32: first `nsample`
 * indices k (ascending) with d^2 < radius^2, padded with the first hit, zeros if none. */
int o3d_ball_query(const float* new_xyz, const float* xyz, int B, int N, int npoint, float radius,
                   int nsample, int32_t* idx, void* stream);

/* ---- grouping -------------------------------------------------------------------------
 * replaces _ext.group_points(features, idx)               pointnet2_utils.py:217
 * feats (B,C,N), idx (B,npoint,nsample) -> out (B,C,npoint,nsample) */
int o3d_group_points(const float* feats, const int32_t* idx, int B, int C, int N, int npoint,
                     int nsample, float* out, void* stream);
/* replaces _ext.group_points_grad(grad_out, idx, N)       pointnet2_utils.py:237
 * grad_out (B,C,npoint,nsample) -> grad_feats (B,C,N); zero-filled by the call. */
int o3d_group_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N,
                          int npoint, int nsample, float* grad_feats, void* stream);

/* ---- 3-NN + inverse-distance interpolation ------------------------------------------
 * replaces _ext.three_nn(unknown, known)                  pointnet2_utils.py:125
 * unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) f32 SQUARED distances, idx (B,n,3) */
int o3d_three_nn(const float* unknown, const float* known, int B, int n, int m, float* dist2,
                 int32_t* idx, void* stream);
/* replaces _ext.three_interpolate(features, idx, weight)  pointnet2_utils.py:162
 * feats (B,c,m), idx (B,n,3), weight (B,n,3) -> out (B,c,n) */
int o3d_three_interpolate(const float* feats, const int32_t* idx, const float* weight, int B,
                          int c, int m, int n, float* out, void* stream);
/* replaces _ext.three_interpolate_grad(grad_out, idx, weight, m)  pointnet2_utils.py:184
 * grad_out (B,c,n) -> grad_feats (B,c,m); zero-filled by the call. */
int o3d_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight,
                               int B, int c, int n, int m, float* grad_feats, void* stream);

/* ---- stable k-nearest selection -----------------------------------------------------
 * replaces torch.cdist + torch.argsort(...)[:k]   models/head/xcorr.py:81,87 and
 *                                                 pointnet2_utils.py:399-400 (knn_point)
 * query (B,Q,D), ref (B,R,D) -> idx (B,Q,k) i32, ascending squared distance, ties ->
 * lowest ref index (argsort leaves tie order unspecified; this pins it).  1 <= k <= 32. */
int o3d_knn(const float* query, const float* ref, int B, int Q, int R, int D, int k,
            int32_t* idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* O3DSOT_H_ */
