// heads.hip -- adapters of the trackers' 1-D conv stacks (models/head/rpn.py:16-39, models/head/xcorr.py:14-17,
// models/bat.py:22-26) to the flat (C, P = B*N) layout of the GEMM kernels.
//
//   pack_rows_kernel     X (rows, P) <- up to four (B, C_i, N) sources with ARBITRARY strides stacked along the
//                        rows (the reference's torch.cat along dim 1 of e.g. [xyz^T ; features], rpn.py:50,
//                        bat.py:94), zero rows up to `rows` (the GEMM kernels want K % 16 == 0, the weight-gradient
//                        kernel K % 64 == 0).  One launch instead of transpose + contiguous + cat + permute.
//   prep_weights_kernel  every padded / transposed weight copy of a step (W zero-padded to aligned shapes, W^T for
//                        the data-gradient GEMMs) in ONE launch from a job table in device memory; the weights only
//                        change in the optimizer step, so the table is static and the launch is graph-capturable.
#include "o3d_common.hpp"

namespace {

struct PackRowsArgs {
    o3d_rows_src s[4];
    int nsrc, B, N, rows;
    float* X;
    long ld;           // row stride of X (0: B*N) -- a column block of a wider operand
};

__global__ __launch_bounds__(256) void pack_rows_kernel(PackRowsArgs a) {
    const long P = (long)a.B * a.N;
    const long col = (long)blockIdx.x * 256 + threadIdx.x;
    if (col >= P) return;
    const int b = (int)(col / a.N), n = (int)(col - (long)b * a.N);
    int r = blockIdx.y;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < a.nsrc) {
            if (r >= 0 && r < a.s[i].C) v = a.s[i].p[b * a.s[i].sb + r * a.s[i].sc + n * a.s[i].sn];
            r -= a.s[i].C;
        }
    }
    a.X[(long)blockIdx.y * (a.ld ? a.ld : P) + col] = v;
}

// job = {src, dst, rows, cols, dst_ld, transpose}: src (rows, cols) row-major ->
//   dst[r*ld + c] (transpose == 0)  or  dst[c*ld + r] (transpose != 0); padding of dst is never written
__global__ __launch_bounds__(256) void prep_weights_kernel(const long* __restrict__ jobs) {
    const long* j = jobs + 6L * blockIdx.x;
    const float* src = reinterpret_cast<const float*>(j[0]);
    float* dst = reinterpret_cast<float*>(j[1]);
    const long rows = j[2], cols = j[3], ld = j[4], tr = j[5];
    const long n = rows * cols;
    for (long o = (long)blockIdx.y * 256 + threadIdx.x; o < n; o += 256L * gridDim.y) {
        if (tr) {
            const long c = o / rows, r = o - c * rows;
            dst[c * ld + r] = src[r * cols + c];
        } else {
            const long r = o / cols, c = o - r * cols;
            dst[r * ld + c] = src[o];
        }
    }
}

// row sums of a (C, P) matrix: the bias gradient of a plain Conv1d layer.  One workgroup per row, fixed order.
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ G, long P, float* __restrict__ out) {
    __shared__ float sh[4];
    const float* g = G + (long)blockIdx.x * P;
    float s = 0.f;
    for (long p = 4L * threadIdx.x; p < P; p += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(&g[p]);
        s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// eval-mode BatchNorm as the four per-channel vectors the fused kernels consume (one launch instead of four
// elementwise torch launches per layer): mean = running_mean - bias_shift, invstd, scale = gamma*invstd,
// shift = beta - mean*scale
__global__ __launch_bounds__(256) void bn_eval_consts_kernel(const float* __restrict__ rm, const float* __restrict__ rv,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ conv_bias, float eps, int C,
                                                             int nrep, float* __restrict__ vec) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float mu = rm[c] - (conv_bias ? conv_bias[c] : 0.f);
    const float is = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
    for (int r = 0; r < nrep; ++r) {            // vec laid out (4, nrep, C): one copy per segment
        vec[((long)0 * nrep + r) * C + c] = mu;
        vec[((long)1 * nrep + r) * C + c] = is;
        vec[((long)2 * nrep + r) * C + c] = sc;
        vec[((long)3 * nrep + r) * C + c] = sh;
    }
}

}  // namespace

// vec (4, nrep, C) = {mean, invstd, scale, shift} of an eval-mode BatchNorm (F.batch_norm with training=False,
// pytorch_utils.py:56-59): invstd = 1/sqrt(running_var + eps).  conv_bias (C) or NULL: a bias of the convolution in
// front, folded into the mean.
extern "C" int o3d_bn_eval_consts(const float* running_mean, const float* running_var, const float* gamma,
                                  const float* beta, const float* conv_bias, float eps, int C, int nrep, float* vec,
                                  void* stream) {
    if (!running_mean || !running_var || !gamma || !beta || !vec || C <= 0 || nrep <= 0) return O3D_EINVAL;
    hipLaunchKernelGGL(bn_eval_consts_kernel, dim3(o3d_cdiv(C, 256)), dim3(256), 0, o3d_stream(stream), running_mean,
                       running_var, gamma, beta, conv_bias, eps, C, nrep, vec);
    return o3d_launch_status();
}

// X (rows, B*N): rows [0, sum C_i) = the sources stacked in order, element (b, c, n) of source i read at
// p + b*sb + c*sc + n*sn (strides in floats); remaining rows zero.  srcs is a HOST array of nsrc <= 4 entries.
extern "C" int o3d_pack_rows(const o3d_rows_src* srcs, int nsrc, int B, int N, int rows, float* X, void* stream) {
    if (!srcs || nsrc < 1 || nsrc > 4 || B <= 0 || N <= 0 || rows <= 0 || !X) return O3D_EINVAL;
    PackRowsArgs a = {};
    int total = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (!srcs[i].p || srcs[i].C <= 0) return O3D_EINVAL;
        a.s[i] = srcs[i];
        total += srcs[i].C;
    }
    if (total > rows) return O3D_EINVAL;
    a.nsrc = nsrc; a.B = B; a.N = N; a.rows = rows; a.X = X;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)o3d_cdiv((long)B * N, 256), rows), dim3(256), 0,
                       o3d_stream(stream), a);
    return o3d_launch_status();
}

// o3d_pack_rows into a column block of a wider operand: X points at the block's first column, ld = the operand's row
// stride (two sets of clouds of different sizes side by side as the columns of ONE GEMM: conv_final on the template and
// on the search feature, models/bat.py:91-92)
extern "C" int o3d_pack_rows_ld(const o3d_rows_src* srcs, int nsrc, int B, int N, int rows, float* X, long ld, void* stream) {
    if (!srcs || nsrc < 1 || nsrc > 4 || B <= 0 || N <= 0 || rows <= 0 || !X || ld < (long)B * N) return O3D_EINVAL;
    PackRowsArgs a = {};
    int total = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (!srcs[i].p || srcs[i].C <= 0) return O3D_EINVAL;
        a.s[i] = srcs[i];
        total += srcs[i].C;
    }
    if (total > rows) return O3D_EINVAL;
    a.nsrc = nsrc; a.B = B; a.N = N; a.rows = rows; a.X = X; a.ld = ld;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)o3d_cdiv((long)B * N, 256), rows), dim3(256), 0,
                       o3d_stream(stream), a);
    return o3d_launch_status();
}

// jobs: DEVICE array of njobs x 6 longs {src, dst, rows, cols, dst_ld, transpose} (see prep_weights_kernel)
extern "C" int o3d_prep_weights(const long* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0) return O3D_EINVAL;
    hipLaunchKernelGGL(prep_weights_kernel, dim3(njobs, 32), dim3(256), 0, o3d_stream(stream), jobs);
    return o3d_launch_status();
}

// out (C) = row sums of G (C, P); P % 4 == 0
extern "C" int o3d_row_sum(const float* G, int C, long P, float* out, void* stream) {
    if (!G || !out || C <= 0 || P <= 0 || P % 4 != 0) return O3D_EINVAL;
    hipLaunchKernelGGL(row_sum_kernel, dim3(C), dim3(256), 0, o3d_stream(stream), G, P, out);
    return o3d_launch_status();
}

// ---- Adam on flat buffers ------------------------------------------------------------------------------------
// torch.optim.Adam's update (models/base_model.py:32-33: betas (0.5, 0.999), eps 1e-6) for every parameter of the model
// in ONE launch: parameters and both moments live in flat buffers (each nn.Parameter is a view), the gradients stay
// where autograd left them and are found through a device job table {grad ptr, offset, n}.  torch's fused multi-tensor
// Adam spends 3 launches / 0.13 ms per step on this model's 76 small tensors; this is one ~10 us launch.
namespace {
__global__ __launch_bounds__(256) void adam_step_kernel(const long* __restrict__ jobs, float* __restrict__ P,
                                                        float* __restrict__ M, float* __restrict__ V, float lr_bc1,
                                                        float beta1, float omb1, float beta2, float omb2, float eps,
                                                        float wd, float inv_sqrt_bc2) {
    const long* j = jobs + 3L * blockIdx.x;
    const float* g = reinterpret_cast<const float*>(j[0]);
    const long off = j[1], n = j[2];
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < n; i += 256L * gridDim.y) {
        const long k = off + i;
        float p = P[k];
        float gi = g[i];
        if (wd != 0.f) gi = fmaf(wd, p, gi);
        const float m = fmaf(beta1, M[k], omb1 * gi);
        const float v = fmaf(beta2, V[k], omb2 * gi * gi);
        M[k] = m; V[k] = v;
        P[k] = p - lr_bc1 * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
    }
}
}  // namespace

// One Adam step: jobs = DEVICE array of njobs x 3 longs {gradient pointer, offset of the parameter in the flat buffers,
// element count}; bias corrections bc1 = 1 - beta1^t, bc2 = 1 - beta2^t computed by the caller (t = step count).
//   m = beta1*m + (1-beta1)*g;  v = beta2*v + (1-beta2)*g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// (g += weight_decay * p first when weight_decay != 0) -- torch.optim.Adam without amsgrad / maximize.
extern "C" int o3d_adam_step(const long* jobs, int njobs, float* params, float* exp_avg, float* exp_avg_sq, double lr,
                             double beta1, double beta2, double eps, double weight_decay, double bc1, double bc2,
                             void* stream) {
    // hyper-parameters in double like torch's Python scalars: 1 - 0.999f is 4.7e-5 away from 0.001
    if (!jobs || njobs <= 0 || !params || !exp_avg || !exp_avg_sq || bc1 <= 0. || bc2 <= 0.) return O3D_EINVAL;
    hipLaunchKernelGGL(adam_step_kernel, dim3(njobs, 32), dim3(256), 0, o3d_stream(stream), jobs, params, exp_avg,
                       exp_avg_sq, (float)(lr / bc1), (float)beta1, (float)(1. - beta1), (float)beta2, (float)(1. - beta2),
                       (float)eps, (float)weight_decay, (float)(1.0 / sqrt(bc2)));
    return o3d_launch_status();
}


// ---- best proposal of a tracked frame (models/base_model.py:44-57) ----------------------------------------------------
// The reference ships the (P, 5) proposals of a frame to the host and takes `estimation_box_cpu[:, 4].argmax()` there
// (numpy: the FIRST maximum), then keeps columns 0..3.  Here one wave per frame does the same on the device: lanes stride
// over the proposals keeping (score, index) with "greater, or equal and lower index" as the order, a butterfly folds the
// 64 lanes, lane 0 writes the four box values and the index.  A NaN score is never greater, like numpy's argmax only when
// no NaN is present -- the trackers' heads emit finite values (checked by the tests).
namespace {
__global__ __launch_bounds__(64) void best_proposal_kernel(const float* __restrict__ boxes, int P, float* __restrict__ out,
                                                           int32_t* __restrict__ out_idx) {
    const float* bx = boxes + (long)blockIdx.x * P * 5;
    const int lane = threadIdx.x;
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int j = lane; j < P; j += 64) {
        const float s = bx[5 * j + 4];
        if (s > best || (s == best && j < bi) || bi == 0x7fffffff) { best = s; bi = j; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float os = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || os > best || (os == best && oi < bi))) { best = os; bi = oi; }
    }
    if (lane < 4) out[4 * blockIdx.x + lane] = bx[5 * bi + lane];
    if (lane == 0 && out_idx) out_idx[blockIdx.x] = bi;
}
}  // namespace

// boxes (B, P, 5) contiguous -> out (B, 4) = boxes[b, argmax_p boxes[b, p, 4], 0:4], out_idx (B) int32 (may be NULL)
extern "C" int o3d_best_proposal(const float* boxes, int B, int P, float* out, int32_t* out_idx, void* stream) {
    if (!boxes || !out || B <= 0 || P <= 0) return O3D_EINVAL;
    hipLaunchKernelGGL(best_proposal_kernel, dim3(B), dim3(64), 0, o3d_stream(stream), boxes, P, out, out_idx);
    return o3d_launch_status();
}

// ---- the element-wise glue of the vote head (models/head/rpn.py:44-66) as three launches (round 5) ------------------------
//   rpn_votes_fwd   sigmoid of the seed scores, the votes' coordinates point-major, cat(score, vote features) (:47-56):
//                   were sigmoid + a transposing copy + a concatenation
//   rpn_votes_bwd   their backward: the packed gradient of `vote` (rows: d xyz, d features) and d scores * s (1 - s):
//                   were a concatenation + sigmoid_backward (+ the views autograd made contiguous)
//   box_assemble    boxes (B,P,5) = [offsets[:, :3] + centres ; offsets[:, 3:]] transposed (:62-66): were add + cat + copy;
//                   its backward is two VIEWS of the incoming gradient (no launch)
namespace {
struct Src3 { const float* p; long sb, sc, sn; };     // a (B, C, N) tensor with arbitrary strides (floats)

__global__ __launch_bounds__(256) void rpn_votes_fwd_kernel(Src3 cla /* (B,1,N) */, Src3 vote /* (B,3+f,N) */, int B, int N, int f,
                                                            float* __restrict__ score, float* __restrict__ vxyz,
                                                            float* __restrict__ vfeat) {
    const long col = (long)blockIdx.x * 256 + threadIdx.x;
    if (col >= (long)B * N) return;
    const int b = (int)(col / N), n = (int)(col - (long)b * N);
    const int r = blockIdx.y;                 // 0: the score row; 1..f: feature rows; f + 1: the three coordinates
    if (r == 0) {
        const float x = cla.p[b * cla.sb + n * cla.sn];
        const float s = 1.f / (1.f + expf(-x));
        score[col] = s;
        vfeat[((long)b * (1 + f)) * N + n] = s;
    } else if (r <= f) {
        vfeat[((long)b * (1 + f) + r) * N + n] = vote.p[b * vote.sb + (long)(2 + r) * vote.sc + n * vote.sn];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) vxyz[col * 3 + k] = vote.p[b * vote.sb + k * vote.sc + n * vote.sn];
    }
}

__global__ __launch_bounds__(256) void rpn_votes_bwd_kernel(const float* __restrict__ score, Src3 dvf /* (B,1+f,N) | NULL */,
                                                            Src3 dvx /* (B,3,N) view of d vote_xyz | NULL */, int B, int N, int f,
                                                            float* __restrict__ dcla /* (B,N) */,
                                                            float* __restrict__ dvote /* (3+f, B*N) */) {
    const long col = (long)blockIdx.x * 256 + threadIdx.x;
    const long P = (long)B * N;
    if (col >= P) return;
    const int b = (int)(col / N), n = (int)(col - (long)b * N);
    const int r = blockIdx.y;                 // 0..2+f: rows of dvote; 3 + f: d scores
    if (r < 3) {
        dvote[(long)r * P + col] = dvx.p ? dvx.p[b * dvx.sb + r * dvx.sc + n * dvx.sn] : 0.f;
    } else if (r < 3 + f) {
        dvote[(long)r * P + col] = dvf.p ? dvf.p[b * dvf.sb + (long)(r - 2) * dvf.sc + n * dvf.sn] : 0.f;
    } else {
        const float s = score[col];
        dcla[col] = dvf.p ? dvf.p[b * dvf.sb + n * dvf.sn] * s * (1.f - s) : 0.f;
    }
}

__global__ __launch_bounds__(256) void box_assemble_kernel(Src3 off /* (B,5,P) */, const float* __restrict__ centers /* (B,P,3) */,
                                                           long total, int P, float* __restrict__ boxes /* (B,P,5) */) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;       // (b, p, k), k fastest
    if (t >= total) return;
    const long bp = t / 5;
    const int k = (int)(t - bp * 5);
    const int b = (int)(bp / P), p = (int)(bp - (long)b * P);
    float v = off.p[b * off.sb + k * off.sc + p * off.sn];
    if (k < 3) v += centers[bp * 3 + k];
    boxes[t] = v;
}
}  // namespace

extern "C" int o3d_rpn_votes_fwd(const float* cla, long cla_sb, long cla_sn, const float* vote, long v_sb, long v_sc, long v_sn,
                                 int B, int N, int f, float* score, float* vote_xyz, float* vote_feature, void* stream) {
    if (!cla || !vote || !score || !vote_xyz || !vote_feature || B <= 0 || N <= 0 || f <= 0) return O3D_EINVAL;
    const Src3 c = {cla, cla_sb, 0, cla_sn}, v = {vote, v_sb, v_sc, v_sn};
    hipLaunchKernelGGL(rpn_votes_fwd_kernel, dim3((unsigned)o3d_cdiv((long)B * N, 256), f + 2), dim3(256), 0, o3d_stream(stream),
                       c, v, B, N, f, score, vote_xyz, vote_feature);
    return o3d_launch_status();
}

extern "C" int o3d_rpn_votes_bwd(const float* score, const float* dvf, long f_sb, long f_sc, long f_sn, const float* dvx,
                                 long x_sb, long x_sc, long x_sn, int B, int N, int f, float* dcla, float* dvote, void* stream) {
    if (!score || !dcla || !dvote || B <= 0 || N <= 0 || f <= 0) return O3D_EINVAL;
    const Src3 a = {dvf, f_sb, f_sc, f_sn}, x = {dvx, x_sb, x_sc, x_sn};
    hipLaunchKernelGGL(rpn_votes_bwd_kernel, dim3((unsigned)o3d_cdiv((long)B * N, 256), f + 4), dim3(256), 0, o3d_stream(stream),
                       score, a, x, B, N, f, dcla, dvote);
    return o3d_launch_status();
}

extern "C" int o3d_box_assemble(const float* offsets, long o_sb, long o_sc, long o_sn, const float* centers, int B, int P,
                                float* boxes, void* stream) {
    if (!offsets || !centers || !boxes || B <= 0 || P <= 0) return O3D_EINVAL;
    const Src3 o = {offsets, o_sb, o_sc, o_sn};
    const long total = (long)B * P * 5;
    hipLaunchKernelGGL(box_assemble_kernel, dim3((unsigned)o3d_cdiv(total, 256)), dim3(256), 0, o3d_stream(stream), o, centers, total,
                       P, boxes);
    return o3d_launch_status();
}
