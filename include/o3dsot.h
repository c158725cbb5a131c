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
 * touched when N > 8192 (may be NULL otherwise). */
int o3d_furthest_point_sampling(const float* xyz, int B, int N, int npoint, float* temp,
                                int32_t* idx, void* stream);

/* Test hook: the same sampling with the wave arg-max through ds_bpermute shuffles instead of DPP (the two
 * reductions must agree bit for bit, tests/test_index_ops_gpu.py). */
int o3d_furthest_point_sampling_shfl(const float* xyz, int B, int N, int npoint, float* temp, int32_t* idx,
                                     void* stream);

/* Two independent sets of B clouds in ONE launch (the template and the search cloud of a tracker step,
 * models/bat.py:89-90: 2 x B one-wave workgroups side by side instead of back to back).  Each set's indices are
 * bit-identical to o3d_furthest_point_sampling on it.  max(N0,N1) <= 2048, else O3D_EINVAL (make two calls). */
int o3d_furthest_point_sampling_pair(const float* xyz0, int N0, int npoint0, int32_t* idx0, const float* xyz1,
                                     int N1, int npoint1, int32_t* idx1, int B, void* stream);

/* ---- gather ---------------------------------------------------------------------------
 * replaces _ext.gather_points(features, idx)              pointnet2_utils.py:92
 * feats (B,C,N), idx (B,npoint) -> out (B,C,npoint) */
int o3d_gather_points(const float* feats, const int32_t* idx, int B, int C, int N, int npoint,
                      float* out, void* stream);
/* replaces _ext.gather_points_grad(grad_out, idx, N)      pointnet2_utils.py:98
 * grad_out (B,C,npoint) -> grad_feats (B,C,N); grad_feats is zero-filled by the call. */
int o3d_gather_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N,
                           int npoint, float* grad_feats, void* stream);

/* rows of a point-major tensor: src (B,N,D), idx (B,npoint) -> out (B,npoint,D), out[b,j,:] = src[b,idx[b,j],:].
 * The ball centres of a set abstraction, new_xyz = gather_operation(xyz^T, fps_idx)^T (pointnet2_modules.py:52-62),
 * without the two transposed copies. */
int o3d_gather_rows(const float* src, const int32_t* idx, int B, int N, int D, int npoint, float* out, void* stream);

/* o3d_gather_rows for one or two point-major tensors (a (B,N,Da), b (B,N,Db) | NULL with Db = 0) through the same index, whose
 * rows may be a prefix of a longer index (idx[b * ld_idx + j], ld_idx >= npoint): the seed labels / BoxCloud rows of the
 * trackers (models/bat.py:96-97,132-133: `.long()` + expand + torch.gather per tensor). */
int o3d_gather_rows2(const float* a, int Da, const float* b, int Db, const int32_t* idx, long ld_idx, int B, int N, int npoint,
                     float* outa, float* outb, void* stream);

/* ---- ball query -----------------------------------------------------------------------
 * replaces _ext.ball_query(new_xyz, xyz, radius, nsample) pointnet2_utils.py:268
 * new_xyz (B,npoint,3), xyz (B,N,3) -> idx (B,npoint,nsample) i32: first `nsample`
 * indices k (ascending) with d^2 < radius^2, padded with the first hit, zeros if none. */
int o3d_ball_query(const float* new_xyz, const float* xyz, int B, int N, int npoint, float radius,
                   int nsample, int32_t* idx, void* stream);

/* Sampling gather + ball query of one or two sets of clouds in one launch: centres = xyz[b, sidx[b, j]] (sidx (B, npoint)
 * int32, or NULL: the first npoint points, pointnet2_modules.py:56) written to `centers` ((B*npoint0 + B*npoint1 + 1, 3):
 * set 0's, set 1's, then an origin row), idx_s (B, npoint_s, nsample) = ball_query around them (bit-identical to
 * o3d_ball_query on the gathered centres).  npoint1 = 0: one set.  Replaces gather_operation + ball_query of
 * pointnet2_modules.py:52-64 / pointnet2_utils.py:92,268 for both clouds of a backbone level. */
int o3d_sample_query(const float* xyz0, const int32_t* sidx0, int N0, int npoint0, int32_t* idx0, const float* xyz1,
                     const int32_t* sidx1, int N1, int npoint1, int32_t* idx1, int B, float radius, int nsample,
                     float* centers, void* stream);

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


/* ======================================================================================
 * Fused grouped-MLP layers (fp32-MFMA GEMMs, open3dsot_amd/csrc/mlp.hip).
 * Replace, per SharedMLP layer, Conv2d(1x1,bias=False) + BatchNorm2d + ReLU
 *                                   pointnet2/utils/pytorch_utils.py:12-37,68-121
 * the QueryAndGroup gather feeding layer 0          pointnet2_utils.py:299-339
 * the max-pool over nsample                          pointnet2_modules.py:69-73
 * BoxAwareXCorr's group + mlp + max                  models/head/xcorr.py:89-100
 * and the autograd backward of that chain.  P = npoint*nsample positions per batch item,
 * P % 128 == 0 and nsample % 4 == 0.  "Raw" tensors Y are conv outputs BEFORE BatchNorm;
 * consumers apply BN+ReLU on load via per-channel (scale, shift).
 * ====================================================================================== */

/* Y[b,co,p] = sum_ci W[co,ci] * f(X[b,ci,p]);  f = identity when in_scale == NULL, else
 * f(x) = max(x*in_scale[ci] + in_shift[ci], 0).  X (B,Cin,P), W (Cout,Cin), Y (B,Cout,P).
 * part != NULL: receives per-tile partial statistics [B*P/128][2][Cout] = {sum y,
 * sum (y - stat_c[co])^2} (stat_c == NULL means 0). */
int o3d_mlp_conv_fwd(const float* X, const float* W, const float* in_scale, const float* in_shift,
                     int B, int Cin, int Cout, int P, float* Y, float* part, const float* stat_c,
                     void* stream);

/* (fold: optional scratch of 64*C floats; long partial lists are first folded into 32 parts by a
 * wide kernel so the finalize does not walk thousands of rows from a handful of workgroups.)
 * Training-mode BatchNorm statistics from the partials: mean, invstd = 1/sqrt(var_biased+eps),
 * scale = gamma*invstd, shift = beta - mean*scale (C each); when running_mean != NULL and
 * momentum >= 0 the running statistics are updated like torch.nn.BatchNorm (unbiased var). */
int o3d_bn_finalize(const float* part, int nparts, int C, double count, const float* stat_c,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                    float* fold, void* stream);

/* out[b,c,j] = max_k relu(Y[b,c,j*ns+k]*scale[c] + shift[c]); optional arg (k of the max) and
 * yarg (raw Y at that k) for the backward pass. */
int o3d_bn_relu_maxpool_fwd(const float* Y, const float* scale, const float* shift, int B, int C,
                            int npoint, int ns, float* out, int32_t* arg, float* yarg, void* stream);

/* The same for ONE cloud of npoint balls in the flat (C, npoint) layout (the P2B fusion: npoint = B*N), the balls of
 * a channel split over nsplit workgroups: part [nsplit][2][C]. */
int o3d_pool_bwd_partials_split(const float* dOut, const float* out, const float* yarg, const float* mean, int C,
                                int npoint, int nsplit, float* part, const int32_t* arg, float* pk, void* stream);

/* BatchNorm backward from partials {sum dN, sum dN*(Y-mean)}: dgamma, dbeta and the per-channel
 * coefficients of dY = A1*dN + A2*Y + A3. */
int o3d_bn_bwd_finalize(const float* part, int nparts, int C, double count, const float* gamma,
                        const float* mean, const float* invstd, float* dgamma, float* dbeta,
                        float* A1, float* A2, float* A3, float* fold, void* stream);

/* Data gradient of an inner layer.  dY comes from dN (dense, (B,Cout,P)) or, when dN == NULL,
 * from the pooled triple (dOut, out, arg) of o3d_bn_relu_maxpool_fwd.  Writes
 * dNprev (B,Cin,P) = (W^T dY) masked by the producer's ReLU (Yprev*scale_p+shift_p > 0) and the
 * producer's BN-backward partials [B*P/128][2][Cin]. */
int o3d_mlp_conv_dgrad(const float* dN, const float* dOut, const float* out, const int32_t* arg,
                       int ns, const float* Y, const float* A1, const float* A2, const float* A3,
                       const float* W, int B, int Cin, int Cout, int P, const float* Yprev,
                       const float* scale_p, const float* shift_p, const float* mean_p,
                       float* dNprev, float* part, void* stream);

/* o3d_mlp_conv_dgrad with the transposed weights Wt (Cin,Cout) supplied as well: aligned shapes
 * (Cin % 64 == 0, Cout % 16 == 0) run the LDS-free kernel, which reads its A operand along Cout
 * and, for the pooled source, the packed pk of o3d_pool_bwd_partials_split instead of (dOut,out,arg). */
int o3d_mlp_conv_dgrad_wt(const float* dN, const float* dOut, const float* out, const int32_t* arg,
                          int ns, const float* Y, const float* A1, const float* A2, const float* A3,
                          const float* W, const float* Wt, const float* pk, int B, int Cin, int Cout, int P,
                          const float* Yprev, const float* scale_p, const float* shift_p,
                          const float* mean_p, float* dNprev, float* part, void* stream);

/* dX (B,Cin,P) = W^T (A1*dN + A2*Y + A3): gradient w.r.t. an operand that is not a BN+ReLU output
 * (the per-point operand [xyz;feats] of layer 0).  No mask, no statistics. */
int o3d_mlp_conv_dgrad_plain(const float* dN, const float* Y, const float* A1, const float* A2,
                             const float* A3, const float* W, int B, int Cin, int Cout, int P,
                             float* dX, void* stream);

/* Weight gradient dW (Cout,Cin) = sum_{b,p} dY[b,co,p] * X[b,ci,p]; X = f(X raw) as in
 * o3d_mlp_conv_fwd.  part: scratch of (nslices+16)*Cout*Cin floats (split over positions, reduced in a fixed
 * order).  xyz .. inv_radius: arguments of the slot-wise layer-0 gather retired in round 4 (layer 0 runs on the
 * points, csrc/compact.hip); pass NULL / 0, X must not be NULL. */
int o3d_mlp_conv_wgrad(const float* dN, const float* dOut, const float* out, const int32_t* arg, int ns,
                       const float* Y, const float* A1, const float* A2, const float* A3,
                       const float* X, const float* in_scale, const float* in_shift, const float* xyz,
                       const float* new_xyz, const float* feats, const int32_t* idx, int N, int C,
                       int nxyz, float inv_radius, int B, int Cin, int Cout, int P, int nslices,
                       float* part, float* dW, void* stream);

/* ---- compact (distinct-neighbour) layout, csrc/compact.hip ------------------------------------------
 * ball_query pads a ball with copies of its first hit (pointnet2_utils.py:268); copies have identical
 * values at every layer, so each ball keeps its cnt distinct entries as columns of flat (C, ldp)
 * matrices (ldp = worst-case column count) and the first hit carries the weight 1+nsample-cnt.
 * meta: 4 device ints per SEGMENT = {live columns rounded up to 256, live columns, balls, 0}; kernels
 * skip tiles beyond the live range -- no host synchronisation.
 * Segments: one fused call can carry two independent sets of clouds through the SAME weights with
 * SEPARATE BatchNorm statistics (template and search branch of the backbone, models/bat.py:89-90).
 * Segment 1's columns start at `start1` (a multiple of 256, = worst-case size of segment 0; 0 means one
 * segment), its point columns after segment 0's, its balls after segment 0's, and every per-channel
 * constant array holds segment 1's values right after segment 0's. */

/* One segment: idx (B,npoint,ns) -> ball_cnt (B*npoint), ball_off (B*npoint+1, absolute columns), per
 * column gp = pt_base + b*ld + point, cball = ball_base + ball (dummy_ball for padding columns), cw;
 * written at [col_base, col_base+live).  ns a power of two <= 64, B*npoint <= 65536. */
int o3d_compact_build(const int32_t* idx, int B, int npoint, int ns, int ld, int col_base, int pt_base,
                      int ball_base, int dummy_ball, int32_t* ball_cnt, int32_t* ball_off, int32_t* gp,
                      int32_t* cball, float* cw, int32_t* meta, void* stream);

/* Weight gradient of an xyz-only layer 0 whose input gradient is not needed (set abstraction level 0 of the backbone:
 * features None, models/backbone/pointnet.py:30-38), straight from the columns of the compact layout:
 * dW0[c,k] = sum_q (A1*dN + w*(A2*Y0 + A3))[c,q] * (X[k, gp[q]] - centers[cball[q], k]), k < 3.  Replaces the list-sum
 * reduction (o3d_group_reduce_gather), the K = 3 GEMM and the centre term.  part: (ldp/256)*C0*3 floats of scratch. */
int o3d_group_dw0_xyz(const float* dN, const float* Y0, long ldp, const float* A1, const float* A2, const float* A3,
                      const int32_t* gp, const int32_t* cball, const float* cw, const float* X, long ldz,
                      const float* centers, const int32_t* meta, long start1, int C0, float* part, float* dW, void* stream);

/* o3d_compact_build for the two segments of a paired call (template + search cloud through one shared module,
 * models/bat.py:89-90) in three launches instead of six: segment 0 at column / point / ball base 0, segment 1 at
 * col_base1 / pt_base1 / ball base B*npoint0; ball_cnt, ball_off: B*(npoint0+npoint1) (+1) entries, meta: 8 ints. */
int o3d_compact_build2(const int32_t* idx0, int npoint0, int ld0, const int32_t* idx1, int npoint1, int ld1, int B,
                       int ns, int col_base1, int pt_base1, int dummy_ball, int32_t* ball_cnt, int32_t* ball_off,
                       int32_t* gp, int32_t* cball, float* cw, int32_t* meta, void* stream);

/* Y0[c,q] = Z[c,gp[q]] - W0[c,0:3].centers[cball[q]] (QueryAndGroup + layer 0 after the per-point GEMM
 * Z = W0.[xyz;feats], pointnet2_utils.py:299-339); centers (balls+1, 3) or NULL; weighted statistics
 * partials part [ldp/256][2][C0] or NULL (stat_c: C0 floats per segment). */
int o3d_group_expand_c(const float* Z, long ldz, const int32_t* gp, const int32_t* cball, const float* cw,
                       const float* centers, const float* W0, int ldw, int C0, const int32_t* meta, long start1,
                       long ldp, float* Y0, float* part, const float* stat_c, void* stream);

/* o3d_group_expand_c for an xyz-only layer 0 (set abstraction level 0: features None) without the per-point GEMM:
 * Y0[c,q] = W0[c,0:3] . (X3[:, gp[q]] - centers[cball[q]]) -- the relative coordinate first, as pointnet2_utils.py:319-320
 * computes it; X3 = the packed coordinate operand (3 rows, ldz columns). */
int o3d_group_expand_c3(const float* X3, long ldz, const int32_t* gp, const int32_t* cball, const float* cw,
                        const float* centers, const float* W0, int ldw, int C0, const int32_t* meta, long start1, long ldp,
                        float* Y0, float* part, const float* stat_c, void* stream);

/* Inner layers on the compact layout: forward (BN+ReLU of the producer on load, weighted statistics),
 * data gradient (dY = A1*dN + w*(A2*Y + A3), ReLU mask, statistics; Wt = W^T), weight gradient.
 * tile = columns per wave tile = columns per statistics partial row: o3d_direct_tile(ldp, M, 1). */
int o3d_direct_tile(long P, int M, int compact);
/* Round 5: a 128-column direct GEMM launch over the compact layout (o3d_mlp_conv_fwd_c / o3d_mlp_conv_dgrad_c with tile = 128)
 * cuts the remainder tiles of its last resident round (T mod S live tiles, S = o3d_direct_tail_slots(output rows) resident
 * column-tile slots) into 2 or 4 column blocks that run as workgroups of the same launch, decided on the device from the live
 * count.  Their statistics go to EXTRA rows behind the regular ones: `part` must hold ldp/128 + nseg * S rows; the
 * o3d_bn_finalize_c[2] / o3d_bn_bwd_finalize_c[2] calls with tile = 128 follow the same plan (C = the launch's output rows). */
int o3d_direct_tail_slots(int M);
/* test hook: -1 automatic (default), 0 no remainder split, S > 0 pretend S slots for every shape */
int o3d_direct_tail_override(int slots);
int o3d_mlp_conv_fwd_c(const float* X, const float* W, const float* in_scale, const float* in_shift, int Cin,
                       int Cout, long ldp, const float* w, const int32_t* meta, long start1, int tile, float* Y,
                       float* part, const float* stat_c, void* stream);
int o3d_mlp_conv_dgrad_c(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3,
                         const float* Wt, int Cin, int Cout, long ldp, const float* w, const int32_t* meta,
                         long start1, int tile, const float* Yprev, const float* scale_p, const float* shift_p,
                         const float* mean_p, float* dNprev, float* part, void* stream);
int o3d_mlp_conv_wgrad2_c(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3,
                          const float* X, const float* in_scale, const float* in_shift, int Cin, int Cout,
                          long ldp, const float* w, const int32_t* meta, long start1, float* scratch, float* dW,
                          void* stream);
/* Round 6 (experiment of the round-5 review's item 2-ii, kept as a tested switch): the same weight-gradient launch, which also
 * WRITES the operand it stages -- dY = A1*dN + w*(A2*Y + A3), (Cout, ldp), live columns only -- so that the data gradient
 * o3d_mlp_conv_dgrad_c can be given that one tensor (dN = dY, Y = NULL: A1..A3 ignored) instead of rebuilding dY from dN and Y. */
int o3d_mlp_conv_wgrad2_c_dy(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3,
                             const float* X, const float* in_scale, const float* in_shift, int Cin, int Cout,
                             long ldp, const float* w, const int32_t* meta, long start1, float* scratch, float* dW,
                             float* dY, void* stream);

/* BatchNorm finalize kernels reading only the live partial rows (meta[0] / tile); per segment: pass the
 * segment's first partial row and its meta block. */
int o3d_bn_finalize_c(const float* part, int nparts, int C, double count, const float* stat_c,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                      const int32_t* meta, int tile, void* stream);
int o3d_bn_bwd_finalize_c(const float* part, int nparts, int C, double count, const float* gamma,
                          const float* mean, const float* invstd, float* dgamma, float* dbeta, float* A1,
                          float* A2, float* A3, const int32_t* meta, int tile, void* stream);

/* out[b,c,j] = max over the ball's columns of relu(Y*scale+shift) (max_pool2d over nsample,
 * pointnet2_modules.py:69-73); argq = column of the maximum, yarg = raw Y there.  Pooled tensors hold
 * one (B,C,npoint_s) block per segment, segment 1's after segment 0's (npoint1 = 0: one segment). */
int o3d_pool_fwd_c(const float* Y, long ldp, const float* scale, const float* shift, const int32_t* ball_off,
                   const int32_t* ball_cnt, int B, int C, int npoint0, int npoint1, float* out, int32_t* argq,
                   float* yarg, void* stream);

/* o3d_pool_fwd_c with the tile transposed through LDS (lane = channel, a wave walks one ball's columns: no idle lanes for
 * the small balls the distinct-neighbour layout produces).  cball (ldp) = ball of every column, meta = the live counts of
 * o3d_compact_build; C % 64 == 0, nsample <= 32 (else O3D_EINVAL: use o3d_pool_fwd_c).  Same results. */
int o3d_pool_fwd_ct(const float* Y, long ldp, const float* scale, const float* shift, const int32_t* ball_off,
                    const int32_t* ball_cnt, const int32_t* cball, const int32_t* meta, long start1, int B, int C,
                    int npoint0, int npoint1, int nsample, float* out, int32_t* argq, float* yarg, void* stream);

/* Backward of the pool: D (C,ldp) = dense class-sum gradient (zero on live columns, D[c,argq] = dOut where
 * out > 0) and the BatchNorm-backward partials of the pooled layer, O3D_POOL_BWD_SPLIT rows per segment:
 * part [nseg][O3D_POOL_BWD_SPLIT][2][C]. */
#define O3D_POOL_BWD_SPLIT 8
int o3d_pool_bwd_c(const float* dOut, const float* out, const int32_t* argq, const float* yarg,
                   const float* mean, int B, int C, int npoint0, int npoint1, const int32_t* meta, long start1,
                   long ldp, float* D, float* part, void* stream);

/* o3d_pool_bwd_c in ONE pass (round 5): D's live columns are written once, column by column (gradient at the ball's arg-max
 * column, zero elsewhere), no zero fill.  cball / ball_off / meta from o3d_compact_build; C % 8 == 0, ldp % 512 == 0,
 * start1 % 512 == 0, else O3D_EINVAL (use o3d_pool_bwd_c).  part [ldp/512][2][C]: one partial row per 512-column chunk
 * (segment 1's rows after segment 0's Pmax0/512), only the live ones written: finalize with meta and tile = 512.
 * dOut0 / dOut1: the pooled-output gradient of segment 0 / 1 as (B, C, npoint_s) with batch / channel strides sb / sc in
 * floats, innermost stride 1; NULL = no gradient (zeros).
 * Replaces max_pool2d's backward of pointnet2_modules.py:70-73 together with o3d_pool_bwd_c. */
int o3d_pool_bwd_dense(const float* dOut0, long sb0, long sc0, const float* dOut1, long sb1, long sc1, const float* out,
                       const int32_t* argq, const float* yarg, const float* mean,
                       const int32_t* cball, const int32_t* ball_off, const int32_t* meta, long start1, long ldp, int B,
                       int C, int npoint0, int npoint1, float* D, float* part, void* stream);

/* The same sums as o3d_group_reduce_c without float atomics: the cloud's columns are sorted by (column chunk,
 * point) once per call (perm: ldp ints; poff: o3d_group_reduce_gather_scratch(...) ints, -1 = shape not covered,
 * use o3d_group_reduce_c) and every sum is a gather from an LDS-staged chunk in a fixed order (bitwise
 * reproducible).  spanmax = npoint*nsample of the largest segment. */
long o3d_group_reduce_gather_scratch(int B, int nseg, int npoint0, int ld0, int npoint1, int ld1, int spanmax);
int o3d_group_reduce_gather(const float* dN, const float* Y0, long ldp, const float* A1, const float* A2,
                            const float* A3, const int32_t* gp, const float* cw, const int32_t* ball_off,
                            const int32_t* ball_cnt, int B, int nseg, int npoint0, int ld0, int npoint1, int ld1, int C0,
                            int spanmax, int32_t* perm, int32_t* poff, float* S, float* T, void* stream);

/* Per-point operand of layer 0 (QueryAndGroup's inputs before the gather, pointnet2_utils.py:318-333):
 * X0 (rows, ldz), ldz = B*(ld0 + ld1): rows [0,nxyz) = xyz^T * inv_radius, rows [nxyz, nxyz+C) = feats, further
 * rows zero; cloud b of segment s occupies columns [base_s + b*ld_s, +N_s), padded to ld_s by zeros.
 * xyz_s (B,N_s,3), feats_s (B,C,N_s); N1 = 0: one segment. */
int o3d_pack_points(const float* xyz0, const float* feats0, int N0, int ld0, const float* xyz1, const float* feats1,
                    int N1, int ld1, int B, int nxyz, int C, float inv_radius, int rows, float* X0, void* stream);

/* Centre term of the layer-0 weight gradient (grouped_xyz = xyz[idx] - new_xyz, pointnet2_utils.py:322-324):
 * dW (C0,ldw) columns 0..2 -= T (C0,nballs) . centers (nballs,3) */
int o3d_center_term(const float* T, const float* centers, int C0, int nballs, int ldw, float* dW, void* stream);
/* the same into a compact gradient: out (C0, ncols) = dW[:, :ncols], columns 0..2 minus the centre term (dW untouched) */
int o3d_center_term_out(const float* T, const float* centers, int C0, int nballs, int ldw, const float* dW, int ncols,
                        float* out, void* stream);

/* Gradient of the ball centres: out (3, nballs)[k, ball] = scale * sum_c W0[c, k] * T[c, ball] (grouped_xyz = xyz[idx] -
 * new_xyz, pointnet2_utils.py:319-320; live where the centres carry a gradient: the vote aggregation, rpn.py:55-60). */
int o3d_center_grad(const float* T, const float* W0, int ldw, int C0, int nballs, float scale, float* out, void* stream);

/* Layer-0 backward sums of dY = A1*dN + w*(A2*Y0 + A3): S (C0, point columns) per source point
 * (= group_points_grad, pointnet2_utils.py:237), T (C0, balls) per ball (may be NULL). */
int o3d_group_reduce_c(const float* dN, const float* Y0, long ldp, const float* A1, const float* A2,
                       const float* A3, const int32_t* gp, const int32_t* cball, const float* cw,
                       const int32_t* ball_off, const int32_t* ball_cnt, int B, int nseg, int npoint0, int ld0,
                       int npoint1, int ld1, int C0, float* S, float* T, void* stream);

/* ---- ends of a per-point MLP chain on the flat (C, P = B*N) layout (csrc/pointwise.hip): the M2-Track
 * stacks (models/backbone/pointnet.py:91-204) run on the GEMM kernels above; these finish a chain. */
/* out = relu(Y*scale+shift) */
int o3d_bn_relu_apply(const float* Y, const float* scale, const float* shift, int C, long P, float* out,
                      void* stream);
/* dN = g masked by the activation, part [P/128][2][C] = {sum dN, sum dN*(Y-mean)} */
int o3d_act_bwd_partials(const float* g, const float* Y, const float* scale, const float* shift,
                         const float* mean, int C, long P, float* dN, float* part, void* stream);
/* AdaptiveMaxPool1d(1) of relu(bn(Y)) per cloud: out (B,C), argq (B,C) column of the first maximum,
 * yarg raw Y there (backward: o3d_pool_bwd_c with one ball per cloud) */
int o3d_gmax_fwd(const float* Y, const float* scale, const float* shift, int B, int C, int N, float* out,
                 int32_t* argq, float* yarg, void* stream);

/* Weight gradient of an aligned inner layer (Cin, Cout multiples of 64; P multiple of 128), workgroup
 * tile matched to the layer: dW (Cout,Cin) = sum dY * f(X), dY = A1*dN + A2*Y + A3 from dN (dense) or,
 * when dN == NULL, from the packed pooled source pk of o3d_pool_bwd_partials_split; f(x) =
 * max(x*in_scale+in_shift, 0), or x when in_scale == in_shift == NULL.  scratch:
 * o3d_mlp_conv_wgrad2_scratch(...) floats. */
long o3d_mlp_conv_wgrad2_scratch(int B, int Cin, int Cout, int P);
int o3d_mlp_conv_wgrad2(const float* dN, const float* pk, int ns, const float* Y, const float* A1,
                        const float* A2, const float* A3, const float* X, const float* in_scale,
                        const float* in_shift, int B, int Cin, int Cout, int P, float* scratch, float* dW,
                        void* stream);

/* BatchNorm finalize / backward finalize of TWO segments in one launch (template + search cloud through a
 * shared module, models/bat.py:89-90): partial rows [nparts0 | nparts1], stat_c / mean / invstd / scale /
 * shift / A1..A3 laid out (2,C), meta (2,4) (NULL in the backward variant: every row live).  Same numbers as
 * the one-segment entry points called for segment 0, then 1; dgamma / dbeta (C) are summed over the segments. */
int o3d_bn_finalize_c2(const float* part, int nparts0, int nparts1, int C, double count0, double count1,
                       const float* stat_c, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                       float* shift, const int32_t* meta, int tile, void* stream);
int o3d_bn_bwd_finalize_c2(const float* part, int nparts0, int nparts1, int C, double count0, double count1,
                           const float* gamma, const float* mean, const float* invstd, float* dgamma,
                           float* dbeta, float* A1, float* A2, float* A3, const int32_t* meta, int tile,
                           void* stream);

/* ---- 1-D conv stacks of the heads on the flat (C, P = B*N) layout -------------------------------------
 * Replaces pt_utils.Seq / Conv1d (+BatchNorm1d +ReLU) stacks, pointnet2/utils/pytorch_utils.py:124-155,300-457, as
 * used by models/head/rpn.py:16-39 (FC_layer_cla, vote_layer, FC_proposal), models/head/xcorr.py:14-17 (fea_layer)
 * and models/bat.py:22-26 (conv_final, mlp_bc): hidden layers conv -> BatchNorm -> ReLU (BatchNorm + ReLU applied
 * by the consumer on load, statistics partials from the producer), last layer conv + bias.
 * P % 64 == 0; contraction sizes % 16 == 0 and output rows % 64 == 0 (callers zero-pad, see o3d_pack_rows /
 * o3d_prep_weights).  Statistics partial rows are per o3d_pw_tile(P, output rows) columns. */
typedef struct { const float* p; long sb, sc, sn; int C; } o3d_rows_src;

/* X (rows, B*N) <- up to 4 sources of shape (B, C_i, N) with arbitrary strides (in floats) stacked along the rows
 * (torch.cat(dim=1) of e.g. [xyz^T ; features]: rpn.py:50, bat.py:94); rows beyond sum C_i are zero.
 * srcs: HOST array. */
int o3d_pack_rows(const o3d_rows_src* srcs, int nsrc, int B, int N, int rows, float* X, void* stream);
/* the same into a column block of a wider operand (X = the block's first column, ld = the operand's row stride): two sets
 * of clouds of different sizes as the columns of ONE GEMM -- conv_final on the template and on the search feature,
 * models/bat.py:91-92 */
int o3d_pack_rows_ld(const o3d_rows_src* srcs, int nsrc, int B, int N, int rows, float* X, long ld, void* stream);

/* The element-wise glue of the vote head, models/head/rpn.py:47-56 (round 5; strides in floats, sources (B, C, N) views):
 * score (B,N) = sigmoid(cla), vote_xyz (B,N,3) = vote[:, :3]^T, vote_feature (B,1+f,N) = cat(score, vote[:, 3:]) in one
 * launch (were sigmoid + a transposing copy + torch.cat); and its backward: dvote (3+f, B*N) = [d vote_xyz^T ; d vote_feature
 * [:, 1:]] in the flat layout of the conv stacks, dcla (B,N) = d vote_feature[:, 0] * s (1 - s); a NULL gradient is zeros. */
int o3d_rpn_votes_fwd(const float* cla, long cla_sb, long cla_sn, const float* vote, long v_sb, long v_sc, long v_sn, int B, int N,
                      int f, float* score, float* vote_xyz, float* vote_feature, void* stream);
int o3d_rpn_votes_bwd(const float* score, const float* dvf, long f_sb, long f_sc, long f_sn, const float* dvx, long x_sb, long x_sc,
                      long x_sn, int B, int N, int f, float* dcla, float* dvote, void* stream);
/* boxes (B,P,5) = transpose(cat(offsets[:, :3] + centers^T, offsets[:, 3:])) of models/head/rpn.py:62-66 (offsets (B,5,P) with
 * any strides, centers (B,P,3) contiguous): one launch (were add + cat + a transposing copy; the backward is two views). */
int o3d_box_assemble(const float* offsets, long o_sb, long o_sc, long o_sn, const float* centers, int B, int P, float* boxes,
                     void* stream);

/* Every padded / transposed weight copy of a step in one launch.  jobs: DEVICE array of njobs x 6 longs
 * {src ptr, dst ptr, rows, cols, dst_ld, transpose}: src (rows, cols) row-major -> dst[r*ld + c], or
 * dst[c*ld + r] when transpose != 0; the padding of dst is not written. */
int o3d_prep_weights(const long* jobs, int njobs, void* stream);

/* out (C) = row sums of G (C, P): bias gradient of a plain Conv1d layer. */
int o3d_row_sum(const float* G, int C, long P, float* out, void* stream);

int o3d_pw_tile(long P, int M);

/* Data gradient + weight gradient of one aligned inner layer in ONE launch (csrc/mlp_wgrad.hip::fused_bwd_kernel):
 * replaces the pair o3d_mlp_conv_wgrad2_c + o3d_mlp_conv_dgrad_c for Cin == 64, Cout in {64, 128} and a dense dN --
 * dY and relu(bn(Yprev)) are staged once for both MFMA chains.  A1..A3 (nseg,Cout); in_scale / in_shift / in_mean
 * (nseg,Cin): BatchNorm of the producer layer; Wt (Cin,Cout) = W^T; w / meta / start1: compact layout (or NULL, NULL, 0).
 * scratch: o3d_mlp_conv_bwd_fused_scratch(...) floats.  part_s [2][rows][2][Cin], rows = o3d_mlp_conv_bwd_fused_rows(...)
 * (-1: shape not supported): BatchNorm-backward partials {sum g, sum g*(yprev-mean)} of dNprev, segment 1's block after
 * segment 0's -- finalize with o3d_bn_bwd_finalize (one segment) / o3d_bn_bwd_finalize_c2(part_s, rows, rows, ..., meta
 * = NULL, tile = 1). */
int o3d_mlp_conv_bwd_fused_rows(int Cin, int Cout, long P);
long o3d_mlp_conv_bwd_fused_scratch(int Cin, int Cout, long P);
int o3d_mlp_conv_bwd_fused_c(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3,
                             const float* X, const float* in_scale, const float* in_shift, const float* in_mean,
                             const float* Wt, int Cin, int Cout, long ldp, const float* w, const int32_t* meta,
                             long start1, float* scratch, float* dW, float* part_s, float* dNprev, void* stream);

/* Two independent 1-D conv launches over the same number of columns as ONE launch (blockIdx.z selects): the stacks of
 * the RPN that read the same seeds -- FC_layer_cla and vote_layer, models/head/rpn.py:16-28,44-54 -- advance layer by layer
 * side by side.  Fields = the arguments of o3d_pw_fwd / o3d_pw_dgrad.  When the two problems do not take the same kernel
 * instantiation the entry issues the two single launches instead; the numbers are the same either way. */
typedef struct {      /* the arguments of o3d_bn_finalize (no fold scratch: partial lists of <= a few hundred rows) */
    const float* part; int nparts, C; double count; const float* stat_c; const float* gamma; const float* beta;
    float* running_mean; float* running_var; float momentum, eps; float* mean; float* invstd; float* scale; float* shift;
} o3d_bn_fin_args;
typedef struct {      /* the arguments of o3d_bn_bwd_finalize */
    const float* part; int nparts, C; double count; const float* gamma; const float* mean; const float* invstd;
    float* dgamma; float* dbeta; float* A1; float* A2; float* A3;
} o3d_bn_bwd_fin_args;
int o3d_bn_finalize_pair(const o3d_bn_fin_args* a, const o3d_bn_fin_args* b, void* stream);
int o3d_bn_bwd_finalize_pair(const o3d_bn_bwd_fin_args* a, const o3d_bn_bwd_fin_args* b, void* stream);
typedef struct {
    const float* X; const float* W; const float* in_scale; const float* in_shift; const float* bias; const float* resid;
    int Cin, Cout; long P;
    float* Y; float* part; const float* stat_c;
} o3d_pw_fwd_args;
typedef struct {
    const float* dN; const float* Y; const float* A1; const float* A2; const float* A3; const float* Wt;
    int Cin, Cout; long P;
    const float* Yprev; const float* scale_p; const float* shift_p; const float* mean_p; const float* resid;
    float* dNprev; float* part;
} o3d_pw_dgrad_args;
int o3d_pw_fwd_pair(const o3d_pw_fwd_args* a, const o3d_pw_fwd_args* b, void* stream);
int o3d_pw_dgrad_pair(const o3d_pw_dgrad_args* a, const o3d_pw_dgrad_args* b, void* stream);

/* Several independent weight gradients of the flat (C, P) layout in one launch (+ one reduction launch): the 1-D conv
 * stacks of the heads (models/head/rpn.py:16-39, models/head/xcorr.py:14-17, models/bat.py:22-26).  Job i computes what
 * o3d_mlp_conv_wgrad2(dN, NULL, 4, Y, A1, A2, A3, X, in_scale, in_shift, 1, Cin, Cout, P, scratch, dW, stream) computes
 * (same tile plan and summation order); njobs <= 8; scratch: o3d_mlp_conv_wgrad2_scratch(1, Cin, Cout, P) floats each.
 * A job with X == NULL and Y == NULL is a ROW-SUM job (the bias gradient of a stack's last layer, o3d_row_sum):
 * dW (Cout) = row sums of dN (Cout, P). */
typedef struct {
    const float* dN; const float* Y; const float* A1; const float* A2; const float* A3;
    const float* X; const float* in_scale; const float* in_shift;
    int Cin, Cout; long P;
    float* scratch; float* dW;
    int out_rows, out_cols;      /* 0, 0: dW is (Cout, Cin); else dW is the COMPACT (out_rows, out_cols) top-left block of it
                                  * (the parameter's own shape when Cout / Cin are its zero-padded sizes) */
} o3d_wgrad_job;
int o3d_mlp_conv_wgrad2_group(const o3d_wgrad_job* jobs, int njobs, void* stream);

/* torch.optim.Adam's update (models/base_model.py:32-33) for all parameters in one launch: parameters and moments in
 * flat buffers, gradients found through a DEVICE job table of njobs x 3 longs {gradient ptr, offset, n};
 * bc1 = 1 - beta1^t, bc2 = 1 - beta2^t.  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps), g += weight_decay*p first. */
int o3d_adam_step(const long* jobs, int njobs, float* params, float* exp_avg, float* exp_avg_sq, double lr,
                  double beta1, double beta2, double eps, double weight_decay, double bc1, double bc2, void* stream);

/* Best proposal of a tracked frame: replaces the host-side selection of MatchingBaseModel.evaluate_one_sample
 * (models/base_model.py:44-57: `estimation_box.cpu().numpy()`, `[:, 4].argmax()`, `[best, 0:4]`) -- SURVEY.md section
 * 8f-4 lists that device->host copy as part of the per-frame latency path.  boxes (B, P, 5) contiguous;
 * out (B, 4) = boxes[b, argmax_p boxes[b, p, 4], 0:4] with numpy's tie rule (first maximum); out_idx (B) int32 or NULL. */
int o3d_best_proposal(const float* boxes, int B, int P, float* out, int32_t* out_idx, void* stream);

/* A per-point stack whose input is [X ; a per-cloud CONSTANT block] (SegPointNet: the pooled feature broadcast to every
 * point and concatenated, models/backbone/pointnet.py:188-190): the constant block contributes W_b . pooled[b] to every
 * column of cloud b -- a per-cloud bias cbias (Cout, B), not 1024 more GEMM rows.
 * o3d_pw_fwd_cloud: Y (Cout, B*N) = W_a (Cout, Cin) . X + cbias[:, cloud], statistics partials [P/128][2][Cout] of it.
 * o3d_cloud_sum_dy: out (C, B) = per-cloud sums of dY = A1*dN + A2*Y + A3, the gradient of cbias. */
int o3d_pw_fwd_cloud(const float* X, const float* W, const float* cbias, int B, int N, int Cin, int Cout, float* Y,
                     float* part, const float* stat_c, void* stream);
int o3d_cloud_sum_dy(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3, int C, int B,
                     int N, float* out, void* stream);

/* Eval-mode BatchNorm constants in one launch: vec (4, nrep, C) = {mean, invstd, scale, shift} with
 * invstd = 1/sqrt(running_var + eps), scale = gamma*invstd, shift = beta - mean*scale (pytorch_utils.py:56-59 in
 * eval mode); conv_bias (C) or NULL is a bias of the convolution in front, folded into the mean. */
int o3d_bn_eval_consts(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                       const float* conv_bias, float eps, int C, int nrep, float* vec, void* stream);

/* Y (Cout, P) = W (Cout, Cin) . f(X),  f = relu(x*in_scale + in_shift) per input row, or identity (both NULL).
 * part != NULL: BatchNorm statistics partials [P/tile][2][Cout] = {sum y, sum (y - stat_c)^2};
 * part == NULL: Y += bias[row] + resid (Cout, P), either may be NULL (last layer of a stack; the residual is the
 * `seeds + vote_layer(seeds)` of rpn.py:53). */
int o3d_pw_fwd(const float* X, const float* W, const float* in_scale, const float* in_shift, const float* bias,
               const float* resid, int Cin, int Cout, long P, float* Y, float* part, const float* stat_c,
               void* stream);

/* dNprev (Cin, P) = Wt (Cin, Cout) . dY,  dY = dN (Y == NULL) or A1*dN + A2*Y + A3.
 * Yprev != NULL: masked by relu(Yprev*scale_p + shift_p) > 0, BatchNorm-backward partials of the producer
 * [P/tile][2][Cin] = {sum g, sum g*(Yprev - mean_p)} in `part`;
 * Yprev == NULL: plain store (+ resid (Cin, P)): the gradient of the stack's input. */
int o3d_pw_dgrad(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3,
                 const float* Wt, int Cin, int Cout, long P, const float* Yprev, const float* scale_p,
                 const float* shift_p, const float* mean_p, const float* resid, float* dNprev, float* part,
                 void* stream);

/* ---- P2B template<->search fusion, layer 0 (models/head/xcorr.py:25-53) ---------------------------------
 * The reference materialises x_ij = [cos_sim(t_i, s_j) ; xyz_i ; feat_i] for all M*N (template, search) pairs and
 * runs SharedMLP + max over the template axis.  Only channel 0 depends on j, so with Z = W0[:,1:].[xyz;feat]
 * (a GEMM over the M template points):   Y0[c, q] = Z[c, b*M + i] + W0[c*ldw] * sim[q],   q = (b*N + j)*M + i.
 * M a power of two in [4, 64], (M*N) % 1024 == 0, C0 % 16 == 0.  part [P/256][2][C0] = {sum y, sum (y-stat_c)^2}. */
int o3d_xcorr_expand(const float* Z, long ldz, const float* sim, const float* W0, int ldw, int B, int M, int N, int C0,
                     float* Y0, float* part, const float* stat_c, void* stream);

/* Backward of the above with dY0 = A1*dN + A2*Y0 + A3:  S (C0, lds) = sum over j (-> dW0[:,1:], d[xyz;feat] through
 * the per-point GEMMs), dsim_part (o3d_xcorr_reduce_groups(C0), P) = per-channel-group partial sums of
 * W0[c,0]*dY0 (the caller adds the rows), dw_part (B, C0) = per-cloud partial sums of dY0*sim (-> dW0[:,0]). */
long o3d_xcorr_reduce_groups(int C0);
int o3d_xcorr_reduce(const float* dN, const float* Y0, const float* A1, const float* A2, const float* A3,
                     const float* sim, const float* W0, int ldw, int B, int M, int N, int C0, float* S, long lds,
                     float* dsim_part, float* dw_part, void* stream);

/* ---- tracking inference: one set abstraction in ONE kernel ---------------------
 * Eval mode (models/base_model.py:59-86 tracks frame by frame, batch 1, BatchNorm on running statistics): gather +
 * centre subtraction + three 1x1 convolutions with BatchNorm + ReLU + max over nsample, activations in LDS.
 * out (B, C2, np) from Z (C0, ldz) = W0 . [xyz ; feats] per point (o3d_pack_points + o3d_mlp_conv_fwd),
 * idx (B, np, ns), centers (B*np, 3) or NULL, v_l (4, C_l) = o3d_bn_eval_consts of layer l.
 * C0 % 8 == 0, C1 % 32 == 0, C2 % 32 == 0, ns a power of two <= 32, (B*np*ns) % 32 == 0; ld = point columns per
 * cloud in Z, pt_base = first column of this set of clouds. */
int o3d_sa_eval_fused(const float* Z, long ldz, const int32_t* idx, const float* centers, const float* W0, int ldw,
                      const float* v0, const float* W1, const float* v1, const float* W2, const float* v2, int C0, int C1,
                      int C2, int B, int np, int ns, int ld, long pt_base, float* out, void* stream);

/* ---- P2B_XCorr's cosine similarity map (models/head/xcorr.py:37-38, nn.CosineSimilarity(dim=1, eps=1e-8)) ----------
 * sim (B,N,M)[b,j,i] = <t[b,:,i], s[b,:,j]> / (max(|t_i|, eps) * max(|s_j|, eps));  t (B,f,M), s (B,f,N) with ELEMENT
 * strides (batch, channel, point) -- the features are views of a flat conv output; tn (B,M), sn (B,N) = the clamped
 * norms (kept for the backward).  f % 32 == 0, M <= 64, M % 4 == 0, N <= 128.
 * Backward: dt (B,f,M), ds (B,f,N) contiguous from dsim (B,N,M). */
int o3d_cosine_sim_fwd(const float* t, long tsb, long tsc, long tsn, const float* s, long ssb, long ssc, long ssn, int B, int f,
                       int M, int N, float* sim, float* tn, float* sn, void* stream);
int o3d_cosine_sim_bwd(const float* dsim, const float* sim, const float* tn, const float* sn, const float* t, long tsb, long tsc,
                       long tsn, const float* s, long ssb, long ssc, long ssn, int B, int f, int M, int N, float* dt, float* ds,
                       void* stream);

/* ---- Linear (+ BatchNorm1d over the rows) (+ ReLU) on R <= 64 rows: the heads of M2-Track (models/m2track.py:43-71) and
 * the hidden rows of MiniPointNet (models/backbone/pointnet.py:118-126), one launch per layer each way (csrc/rowmlp.hip).
 * Forward: Y (R, Cout) = act(bn(X (R, Cin; row stride ldx) . W^T (Cout, Cin) + bias)); gamma == NULL: no BatchNorm;
 * training: batch statistics over the rows (biased variance), running statistics updated with `momentum` (unbiased
 * variance), else the running statistics.  Z (R, Cout) = the pre-BatchNorm output, mean / invstd (Cout) = the constants the
 * normalisation used: kept for the backward (any may be NULL).
 * Backward of one layer: gradient of the layer's OUTPUT = dY (R, C; row stride lddy) or, dY == NULL, dZup (R, Cup) . Wup
 * (Cup, C) (the layer above, computed on the spot); -> dZ (R, C) gradient of the pre-BatchNorm output, dW (C, Cin), db (C),
 * dgamma / dbeta (C).  input_mode != 0: only the product, stored to dX (R, C; row stride lddx): the stack's input gradient. */
int o3d_row_mlp_fwd(const float* X, int ldx, const float* W, const float* bias, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, int training, int relu, int R, int Cin,
                    int Cout, float* Z, float* Y, float* mean, float* invstd, void* stream);
int o3d_row_mlp_bwd(const float* dY, int lddy, const float* dZup, const float* Wup, int Cup, int input_mode, float* dX, int lddx,
                    const float* Z, const float* gamma, const float* beta, const float* mean, const float* invstd, int training,
                    int relu, const float* X, int ldx, int R, int Cin, int C, float* dZ, float* dW, float* db, float* dgamma,
                    float* dbeta, void* stream);

/* Several independent layers in one launch (njobs <= 4; the heads of M2-Track that read the same feature, models/m2track.py:
 * 60-71): job j is what o3d_row_mlp_fwd / o3d_row_mlp_bwd do with the same arguments (input_mode must be 0).
 * o3d_row_mlp_input_grad: dX (R, C) = sum over the jobs of dZup_j (R, Cup_j) . Wup_j (Cup_j, C), fixed order; R, C, dX, lddx
 * are taken from jobs[0], the other fields are ignored. */
typedef struct {
    const float* X; int ldx; const float* W; const float* bias; const float* gamma; const float* beta;
    float* running_mean; float* running_var; float momentum, eps; int training, relu; int R, Cin, Cout;
    float* Z; float* Y; float* mean; float* invstd;
} o3d_row_fwd_args;
typedef struct {
    const float* dY; int lddy; const float* dZup; const float* Wup; int Cup; int input_mode; float* dX; int lddx;
    const float* Z; const float* gamma; const float* beta; const float* mean; const float* invstd; int training, relu;
    const float* X; int ldx; int R, Cin, C; float* dZ; float* dW; float* db; float* dgamma; float* dbeta;
} o3d_row_bwd_args;
int o3d_row_mlp_fwd_group(const o3d_row_fwd_args* jobs, int njobs, void* stream);
int o3d_row_mlp_bwd_group(const o3d_row_bwd_args* jobs, int njobs, void* stream);
int o3d_row_mlp_input_grad(const o3d_row_bwd_args* jobs, int njobs, void* stream);

/* ---- tracker losses (next row of SURVEY.md section 8f: the loss as one launch) ----------------------
 * MatchingBaseModel.compute_loss (models/base_model.py:122-164) + the BoxCloud term (models/bat.py:57-65)
 * + the weighted total (models/bat.py:131-137, models/p2b.py:69-74) and the gradients of the total.
 * losses[6] = {total, objective, box, seg, vote, bc}.  bc_pred == NULL: no BoxCloud term (P2B).
 * g_* == NULL: losses only.  scratch: 512 floats. */
int o3d_track_loss(const float* cla, const float* seg, const float* vote, const float* box_label,
                   const float* centers, const float* boxes, const float* bc_pred, const float* bc_label, int B,
                   int N, int P, int K, float w_obj, float w_box, float w_seg, float w_vote, float w_bc,
                   float* scratch, float* losses, float* g_cla, float* g_vote, float* g_boxes, float* g_bc, void* stream);

/* M2-Track between its two stages (models/m2track.py:120-137 over datasets/points_utils.py:390-452): aux = the previous box
 * `prev` (B,4 = x, y, z, yaw; NULL: zeros) moved by `motion` (B,4) [get_offset_box_tensor]; the first N/2 points carried along
 * that motion [get_offset_points_tensor], then all N points expressed in the frame of aux [remove_transform_points_tensor].
 * pts: channel c of point n of cloud b at pts[b*bstride + c*cstride + n] (channels 0..2 = xyz).  merged (B,3,N), aux (B,4).
 * _bwd: g_merged (B,3,N), g_aux (B,4) | NULL -> g_prev (B,4) | NULL, g_motion (B,4); no gradient to the points. */
int o3d_motion_merge_fwd(const float* pts, long bstride, long cstride, const float* prev, const float* motion, int B, int N,
                         float* merged, float* aux, void* stream);
int o3d_motion_merge_bwd(const float* pts, long bstride, long cstride, const float* prev, const float* motion, int B, int N,
                         const float* g_merged, const float* g_aux, float* g_prev, float* g_motion, void* stream);

/* Backward of o3d_gmax_fwd without the dense gradient: pk (C, B) float2 = {dOut where out > 0 else 0, bits(column of the
 * maximum relative to its cloud)} -- the pooled operand of o3d_mlp_conv_dgrad_wt / o3d_mlp_conv_wgrad2 with one ball of
 * ns = N columns per cloud -- and the BatchNorm-backward sums part[0][0][c] = sum g, part[0][1][c] = sum g * (yarg - mean). */
int o3d_gmax_bwd_pk(const float* dOut, const float* out, const int32_t* argq, const float* yarg, const float* mean, int B, int C,
                    int N, float* pk, float* part, void* stream);

/* Backward of a thin first layer of a per-point stack (models/backbone/pointnet.py:91-204 with 12-14 input channels) on the
 * flat (C, P) layout, Cout == 64, Cin <= 16, P % 64 == 0: dY = A1*dN + A2*Y + A3 (per-row constants, the BatchNorm backward
 * folded); dW (64, Cin) = dY . X^T; dX (Cin, P) = W^T . dY when dX != NULL.  One pass over dN and Y.
 * scratch: o3d_thin_bwd_scratch() floats. */
long o3d_thin_bwd_scratch(void);
int o3d_thin_bwd(const float* dN, const float* Y, const float* A1, const float* A2, const float* A3, const float* X,
                 const float* W, int Cin, int Cout, long P, float* scratch, float* dW, float* dX, void* stream);

/* get_offset_box_tensor (datasets/points_utils.py:420-436): box = `ref` (B,4) moved by `off` (B,4) given in ref's frame.
 * g_box == NULL: forward, writes box; else backward: g_ref / g_off (either may be NULL) from g_box. */
int o3d_offset_box(const float* ref, const float* off, int B, float* box, const float* g_box, float* g_ref, float* g_off,
                   void* stream);

/* M2-Track's loss (models/m2track.py:153-231) and the gradients of its weighted total: segmentation cross entropy with class
 * weights (cw0, cw1) = (0.5, 2.0), motion-state cross entropy, (centre, angle) smooth-L1 pairs of the refined / previous /
 * first-stage box and of the motion (the latter over the moving samples when `state` is given), BoxCloud smooth-L1 against
 * the concatenation of bc_a and bc_b (B, N/2, K each).  NULL predictions switch their terms off: bc_pred (box_aware), motion_cls +
 * state (use_motion_cls), est (use_second_stage), prev (use_prev_refinement).  losses[12] = {total, motion_cls, center, angle,
 * center_prev, angle_prev, seg, center_aux, center_motion, angle_aux, angle_motion, bc}.  g_seg == NULL: losses only.
 * scratch: 1024 floats. */
int o3d_m2track_loss(const float* seg_logits, const int64_t* seg_label, const float* bc_pred, const float* bc_a, const float* bc_b,
                     const float* motion_cls, const int64_t* state, const float* motion, const float* motion_lab, const float* aux,
                     const float* est, const float* prev, const float* box_lab, const float* prev_lab, int B, int N, int K,
                     float w_center, float w_angle, float w_seg, float w_bc, float w_mcls, float cw0, float cw1, float* scratch,
                     float* losses, float* g_seg, float* g_bc, float* g_mcls, float* g_motion, float* g_aux, float* g_est,
                     float* g_prev, void* stream);

/* ---- BoxCloud (next row of SURVEY.md section 8f-2) -------------------------------------------------
 * get_point_to_box_distance (datasets/points_utils.py:127-143) with Box.corners (datasets/data_classes.py:
 * 226-250): out (B,N,9) = distance of every point to the box centre (channel 0) and to the 8 corners
 * (channels 1..8, the corner order of Box.corners).  wlh = (width, length, height); rot (B,3,3) row-major
 * rotation matrix of the box orientation. */
int o3d_boxcloud(const float* points, const float* center, const float* wlh, const float* rot, float wlh_factor,
                 int B, int N, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* O3DSOT_H_ */
