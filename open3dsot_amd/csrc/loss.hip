// Siamese matching losses of the trackers and their gradients in two small launches.
//
// Restates models/base_model.py:122-164 (MatchingBaseModel.compute_loss: segmentation BCE, vote
// smooth-L1 masked by the segmentation label, objectness BCE with pos_weight 2 on the proposals
// within 0.3 m / masked beyond 0.6 m, box smooth-L1 on the near proposals), models/bat.py:57-65
// (BoxCloud smooth-L1) and the weighted sum of models/bat.py:131-137 / models/p2b.py:69-74.
// As torch ops this is ~50 forward + ~60 backward launches of a few microseconds each on
// 6144-element tensors; here two launches of 64 workgroups walk the B*N seeds and B*P proposals (block
// sums, then the gradients with the denominators known).  Fixed summation order: deterministic.
// (One 1024-thread workgroup doing both passes measured 87 us on the MI355X: a single CU's load
// bandwidth; 64 workgroups bring each pass to launch latency.)
#include "o3d_common.hpp"

namespace {

constexpr int LT = 256;       // threads per workgroup
constexpr int LG = 64;        // workgroups
constexpr int NS = 8;         // block sums

struct LossArgs {
    const float* cla;         // (B,N)   segmentation logits
    const float* seg;         // (B,N)   segmentation label (0/1)
    const float* vote;        // (B,N,3) voted centres
    const float* box_label;   // (B,4)
    const float* centers;     // (B,P,3) proposal centres
    const float* boxes;       // (B,P,5) proposals: x,y,z,ry,objectness logit
    const float* bc_pred;     // (B,N,K) predicted BoxCloud or NULL
    const float* bc_label;    // (B,N,K)
    int B, N, P, K;
    float w_obj, w_box, w_seg, w_vote, w_bc;
    float* partial;           // [LG][NS] block sums
    float* losses;            // [6] total, objective, box, seg, vote, bc
    float* g_cla;             // (B,N)
    float* g_vote;            // (B,N,3)
    float* g_boxes;           // (B,P,5)
    float* g_bc;              // (B,N,K) or NULL
};

__device__ __forceinline__ float smooth_l1(float d) {
    const float a = fabsf(d);
    return a < 1.f ? 0.5f * d * d : a - 0.5f;
}

__device__ __forceinline__ float softplus_neg_abs(float x) { return log1pf(expf(-fabsf(x))); }

// every thread returns the NS block totals (wave shuffles, then a fixed-order pass over the 16 wave rows)
__device__ __forceinline__ void block_sums(float (&v)[NS], float (*red)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0)
#pragma unroll
        for (int k = 0; k < NS; ++k) red[tid >> 6][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < LT / 64; ++w) a += red[w][k];
        v[k] = a;
    }
}

__global__ __launch_bounds__(LT) void track_loss_sums_kernel(LossArgs a) {
    __shared__ float red[LT / 64][NS];
    const int tid = threadIdx.x, gtid = blockIdx.x * LT + threadIdx.x, gstride = LG * LT;
    const int nseed = a.B * a.N, nprop = a.B * a.P;
    // sums: 0 seg count, 1 near count, 2 mask count, 3 seg BCE, 4 vote, 5 bc, 6 objectness BCE, 7 box
    float s[NS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = gtid; i < nseed; i += gstride) {
        const int b = i / a.N;
        const float x = a.cla[i], y = a.seg[i];
        s[0] += y;
        s[3] += fmaxf(x, 0.f) - x * y + softplus_neg_abs(x);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) v += smooth_l1(a.vote[3 * (long)i + k] - a.box_label[4 * b + k]);
        s[4] += v * (1.f / 3.f) * y;
        if (a.bc_pred) {
            float c = 0.f;
            for (int k = 0; k < a.K; ++k) c += smooth_l1(a.bc_pred[(long)i * a.K + k] - a.bc_label[(long)i * a.K + k]);
            s[5] += c / (float)a.K * y;
        }
    }
    for (int i = gtid; i < nprop; i += gstride) {
        const int b = i / a.P;
        float d2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = a.centers[3 * (long)i + k] - a.box_label[4 * b + k];
            d2 += d * d;
        }
        const float dist = sqrtf(d2 + 1e-6f);
        const float near = dist < 0.3f ? 1.f : 0.f;
        const float mask = fminf(near + (dist > 0.6f ? 1.f : 0.f), 1.f);
        s[1] += near;
        s[2] += mask;
        const float x = a.boxes[5 * (long)i + 4];
        // binary_cross_entropy_with_logits with pos_weight 2: (1-y) x + (1 + y) (log1p(exp(-|x|)) + max(-x, 0))
        s[6] += (1.f - near) * x + (1.f + near) * (softplus_neg_abs(x) + fmaxf(-x, 0.f));
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) v += smooth_l1(a.boxes[5 * (long)i + k] - a.box_label[4 * b + k]);
        s[7] += v * 0.25f * near;
    }
    block_sums(s, red);
    if (tid < NS) {
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) mine = k == tid ? s[k] : mine;
        a.partial[blockIdx.x * NS + tid] = mine;
    }
}

__global__ __launch_bounds__(LT) void track_loss_grads_kernel(LossArgs a) {
    __shared__ float tot[NS];
    const int tid = threadIdx.x, gtid = blockIdx.x * LT + threadIdx.x, gstride = LG * LT;
    const int nseed = a.B * a.N, nprop = a.B * a.P;
    if (tid < NS) {
        float t = 0.f;
        for (int g = 0; g < LG; ++g) t += a.partial[g * NS + tid];
        tot[tid] = t;
    }
    __syncthreads();
    float s[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] = tot[k];
    const float inv_seg = 1.f / (s[0] + 1e-6f), inv_near = 1.f / (s[1] + 1e-6f);
    const float mask_ratio = s[2] / (s[2] + 1e-6f);      // the reference's 'mean' BCE is a scalar: the mask rescales it
    const float l_seg = s[3] / (float)nseed;
    const float l_vote = s[4] * inv_seg;
    const float l_bc = a.bc_pred ? s[5] * inv_seg : 0.f;
    const float l_obj = s[6] / (float)nprop * mask_ratio;
    const float l_box = s[7] * inv_near;
    if (gtid == 0) {
        a.losses[0] = l_obj * a.w_obj + l_box * a.w_box + l_seg * a.w_seg + l_vote * a.w_vote + l_bc * a.w_bc;
        a.losses[1] = l_obj; a.losses[2] = l_box; a.losses[3] = l_seg; a.losses[4] = l_vote; a.losses[5] = l_bc;
    }
    if (!a.g_cla) return;
    // gradients of the weighted total
    const float c_seg = a.w_seg / (float)nseed, c_vote = a.w_vote * inv_seg * (1.f / 3.f);
    const float c_bc = a.bc_pred ? a.w_bc * inv_seg / (float)a.K : 0.f;
    for (int i = gtid; i < nseed; i += gstride) {
        const int b = i / a.N;
        const float x = a.cla[i], y = a.seg[i];
        a.g_cla[i] = c_seg * (1.f / (1.f + expf(-x)) - y);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = a.vote[3 * (long)i + k] - a.box_label[4 * b + k];
            a.g_vote[3 * (long)i + k] = c_vote * y * fminf(fmaxf(d, -1.f), 1.f);
        }
        if (a.bc_pred)
            for (int k = 0; k < a.K; ++k) {
                const float d = a.bc_pred[(long)i * a.K + k] - a.bc_label[(long)i * a.K + k];
                a.g_bc[(long)i * a.K + k] = c_bc * y * fminf(fmaxf(d, -1.f), 1.f);
            }
    }
    const float c_obj = a.w_obj * mask_ratio / (float)nprop, c_box = a.w_box * inv_near * 0.25f;
    for (int i = gtid; i < nprop; i += gstride) {
        const int b = i / a.P;
        float d2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = a.centers[3 * (long)i + k] - a.box_label[4 * b + k];
            d2 += d * d;
        }
        const float near = sqrtf(d2 + 1e-6f) < 0.3f ? 1.f : 0.f;
        const float x = a.boxes[5 * (long)i + 4];
        const float sig = 1.f / (1.f + expf(-x));
        a.g_boxes[5 * (long)i + 4] = c_obj * ((1.f - near) - (1.f + near) * (1.f - sig));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = a.boxes[5 * (long)i + k] - a.box_label[4 * b + k];
            a.g_boxes[5 * (long)i + k] = c_box * near * fminf(fmaxf(d, -1.f), 1.f);
        }
    }
}

}  // namespace

// losses[6] = {weighted total, objective, box, seg, vote, bc}; gradients of the TOTAL w.r.t. cla, vote,
// boxes and bc_pred (all four pointers or none; the proposal centres enter through comparisons only and
// get no gradient).  bc_pred == NULL: P2B (no BoxCloud term).  scratch: 512 floats.
extern "C" int o3d_track_loss(const float* cla, const float* seg, const float* vote, const float* box_label,
                              const float* centers, const float* boxes, const float* bc_pred,
                              const float* bc_label, int B, int N, int P, int K, float w_obj, float w_box,
                              float w_seg, float w_vote, float w_bc, float* scratch, float* losses, float* g_cla,
                              float* g_vote, float* g_boxes, float* g_bc, void* stream) {
    if (!scratch || !cla || !seg || !vote || !box_label || !centers || !boxes || !losses || B <= 0 || N <= 0 || P <= 0 ||
        (bc_pred && (!bc_label || K <= 0)) || (g_cla && (!g_vote || !g_boxes || (bc_pred && !g_bc))))
        return O3D_EINVAL;
    LossArgs a{cla, seg, vote, box_label, centers, boxes, bc_pred, bc_label, B, N, P, K, w_obj, w_box, w_seg,
               w_vote, w_bc, scratch, losses, g_cla, g_vote, g_boxes, g_bc};
    hipLaunchKernelGGL(track_loss_sums_kernel, dim3(LG), dim3(LT), 0, o3d_stream(stream), a);
    hipLaunchKernelGGL(track_loss_grads_kernel, dim3(LG), dim3(LT), 0, o3d_stream(stream), a);
    return o3d_launch_status();
}

// =====================================================================================================================
// M2-Track's loss (models/m2track.py:153-231) and its gradients in two small launches (round 4): segmentation cross
// entropy with class weights (0.5, 2.0), motion-state cross entropy, four (centre smooth-L1 over x,y,z ; angle smooth-L1 on
// sin) pairs -- refined box, previous-frame box, first-stage box, motion (the latter averaged over the MOVING samples only)
// -- and the BoxCloud smooth-L1.  As torch ops ~45 launches forward and ~60 backward of a few microseconds each.
// Optional terms are switched off by NULL pointers: motion_cls / state (use_motion_cls), est (use_second_stage), prev
// (use_prev_refinement), bc_pred (box_aware).  Fixed summation order: deterministic.
// =====================================================================================================================
namespace {

constexpr int MS = 16;        // block sums of the M2-Track loss

struct M2LossArgs {
    const float* seg_logits;  // (B,2,N)
    const int64_t* seg_label; // (B,N)
    const float* bc_pred;     // (B,N,K) or NULL
    const float* bc_a;        // (B,N/2,K) BoxCloud label of the previous frame's half ...
    const float* bc_b;        // (B,N/2,K) ... and of this frame's half (the reference concatenates them, :224)
    const float* motion_cls;  // (B,2) or NULL
    const int64_t* state;     // (B) motion state label or NULL
    const float* motion;      // (B,4) motion_pred
    const float* motion_lab;  // (B,4)
    const float* aux;         // (B,4) first-stage box
    const float* est;         // (B,4) refined box or NULL
    const float* prev;        // (B,4) previous-frame box or NULL
    const float* box_lab;     // (B,4)
    const float* prev_lab;    // (B,4)
    int B, N, K;
    float w_center, w_angle, w_seg, w_bc, w_mcls, cw0, cw1;
    float* partial;           // [LG][MS]
    float* losses;            // [12]: total, motion_cls, center, angle, center_prev, angle_prev, seg, center_aux,
                              //       center_motion, angle_aux, angle_motion, bc
    float* g_seg; float* g_bc; float* g_mcls; float* g_motion; float* g_aux; float* g_est; float* g_prev;
};

__device__ __forceinline__ void block_sums16(float (&v)[MS], float (*red)[MS]) {
#pragma unroll
    for (int k = 0; k < MS; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0)
#pragma unroll
        for (int k = 0; k < MS; ++k) red[tid >> 6][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MS; ++k) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < LT / 64; ++w) a += red[w][k];
        v[k] = a;
    }
}

// 2-class log-softmax pieces: returns -log softmax(x)[y], p1 = softmax(x)[1]
__device__ __forceinline__ float nll2(float x0, float x1, int y, float& p1) {
    const float m = fmaxf(x0, x1);
    const float e0 = expf(x0 - m), e1 = expf(x1 - m);
    const float lse = m + logf(e0 + e1);
    p1 = e1 / (e0 + e1);
    return lse - (y ? x1 : x0);
}

// sums: 0 sum of class weights, 1 weighted seg NLL, 2 BoxCloud smooth-L1, 3 moving samples, 4 motion centre (moving),
// 5 motion angle (moving), 6/7 refined centre / angle, 8/9 previous, 10/11 first stage, 12 motion-state NLL
__global__ __launch_bounds__(LT) void m2_loss_sums_kernel(M2LossArgs a) {
    __shared__ float red[LT / 64][MS];
    const int tid = threadIdx.x;
    float s[MS];
#pragma unroll
    for (int k = 0; k < MS; ++k) s[k] = 0.f;
    // a cloud per workgroup (no per-element division: as a flat grid-stride loop with b = i / N this kernel took 40 us)
    const int half = (a.N / 2) * a.K, per = a.N * a.K;
    for (int b = blockIdx.x; b < a.B; b += LG) {
        const float* l0 = a.seg_logits + (long)b * 2 * a.N;
        const int64_t* lab = a.seg_label + (long)b * a.N;
        for (int n = tid; n < a.N; n += LT) {
            const int y = lab[n] != 0;
            float p1;
            const float nll = nll2(l0[n], l0[a.N + n], y, p1);
            const float w = y ? a.cw1 : a.cw0;
            s[0] += w;
            s[1] += w * nll;
        }
        if (a.bc_pred) {
            const float* pr = a.bc_pred + (long)b * per;
            const float* la = a.bc_a + (long)b * half;
            const float* lb = a.bc_b + (long)b * half - half;
            for (int r = tid; r < per; r += LT) s[2] += smooth_l1(pr[r] - (r < half ? la[r] : lb[r]));
        }
    }
    if (blockIdx.x == 0) {
        for (int b = tid; b < a.B; b += LT) {
            const float st = a.state ? (float)(a.state[b] != 0) : 1.f;
            s[3] += st;
            float c = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) c += smooth_l1(a.motion[4 * b + k] - a.motion_lab[4 * b + k]);
            s[4] += st * c * (1.f / 3.f);
            s[5] += st * smooth_l1(sinf(a.motion[4 * b + 3]) - sinf(a.motion_lab[4 * b + 3]));
            const float* rows[3] = {a.est, a.prev, a.aux};
            const float* labs[3] = {a.box_lab, a.prev_lab, a.box_lab};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (!rows[q]) continue;
                float cc = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) cc += smooth_l1(rows[q][4 * b + k] - labs[q][4 * b + k]);
                s[6 + 2 * q] += cc;
                s[7 + 2 * q] += smooth_l1(sinf(rows[q][4 * b + 3]) - sinf(labs[q][4 * b + 3]));
            }
            if (a.motion_cls) {
                float p1;
                s[12] += nll2(a.motion_cls[2 * b], a.motion_cls[2 * b + 1], a.state[b] != 0, p1);
            }
        }
    }
    block_sums16(s, red);
    if (tid < MS) {
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < MS; ++k) mine = k == tid ? s[k] : mine;
        a.partial[blockIdx.x * MS + tid] = mine;
    }
}

__global__ __launch_bounds__(LT) void m2_loss_grads_kernel(M2LossArgs a) {
    __shared__ float tot[MS];
    const int tid = threadIdx.x, gtid = blockIdx.x * LT + threadIdx.x;
    if (tid < MS) {
        float t = 0.f;
        for (int g = 0; g < LG; ++g) t += a.partial[g * MS + tid];
        tot[tid] = t;
    }
    __syncthreads();
    float s[MS];
#pragma unroll
    for (int k = 0; k < MS; ++k) s[k] = tot[k];
    const float fB = (float)a.B;
    const float l_seg = s[1] / s[0];
    const float l_bc = a.bc_pred ? s[2] / ((float)a.B * a.N * a.K) : 0.f;
    // the motion pair: averaged over the moving samples (state given), else plain means over the batch
    const float inv_mov = a.state ? 1.f / (s[3] + 1e-6f) : 1.f / fB;
    const float l_cm = s[4] * inv_mov, l_am = s[5] * inv_mov;
    const float l_c = a.est ? s[6] / (3.f * fB) : 0.f, l_a = a.est ? s[7] / fB : 0.f;
    const float l_cp = a.prev ? s[8] / (3.f * fB) : 0.f, l_ap = a.prev ? s[9] / fB : 0.f;
    const float l_cx = s[10] / (3.f * fB), l_ax = s[11] / fB;
    const float l_mcls = a.motion_cls ? s[12] / fB : 0.f;
    if (gtid == 0) {
        a.losses[0] = l_mcls * a.w_mcls + (l_c + l_cp + l_cx + l_cm) * a.w_center + (l_a + l_ap + l_ax + l_am) * a.w_angle +
                      l_seg * a.w_seg + l_bc * a.w_bc;
        a.losses[1] = l_mcls; a.losses[2] = l_c; a.losses[3] = l_a; a.losses[4] = l_cp; a.losses[5] = l_ap; a.losses[6] = l_seg;
        a.losses[7] = l_cx; a.losses[8] = l_cm; a.losses[9] = l_ax; a.losses[10] = l_am; a.losses[11] = l_bc;
    }
    if (!a.g_seg) return;
    const float c_seg = a.w_seg / s[0];
    const int half = (a.N / 2) * a.K, per = a.N * a.K;
    const float c_bc = a.w_bc / ((float)a.B * a.N * a.K);
    for (int b = blockIdx.x; b < a.B; b += LG) {
        const float* l0 = a.seg_logits + (long)b * 2 * a.N;
        const int64_t* lab = a.seg_label + (long)b * a.N;
        float* g0 = a.g_seg + (long)b * 2 * a.N;
        for (int n = tid; n < a.N; n += LT) {
            const int y = lab[n] != 0;
            float p1;
            nll2(l0[n], l0[a.N + n], y, p1);
            const float w = (y ? a.cw1 : a.cw0) * c_seg;
            g0[n] = w * ((1.f - p1) - (y ? 0.f : 1.f));
            g0[a.N + n] = w * (p1 - (y ? 1.f : 0.f));
        }
        if (a.bc_pred) {
            const float* pr = a.bc_pred + (long)b * per;
            const float* la = a.bc_a + (long)b * half;
            const float* lb = a.bc_b + (long)b * half - half;
            float* g = a.g_bc + (long)b * per;
            for (int r = tid; r < per; r += LT) g[r] = c_bc * fminf(fmaxf(pr[r] - (r < half ? la[r] : lb[r]), -1.f), 1.f);
        }
    }
    if (blockIdx.x == 0) {
        for (int b = tid; b < a.B; b += LT) {
            const float st = a.state ? (float)(a.state[b] != 0) : 1.f;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                a.g_motion[4 * b + k] = a.w_center * st * inv_mov * (1.f / 3.f) *
                                        fminf(fmaxf(a.motion[4 * b + k] - a.motion_lab[4 * b + k], -1.f), 1.f);
            {
                const float p = a.motion[4 * b + 3];
                a.g_motion[4 * b + 3] = a.w_angle * st * inv_mov * cosf(p) *
                                        fminf(fmaxf(sinf(p) - sinf(a.motion_lab[4 * b + 3]), -1.f), 1.f);
            }
            const float* rows[3] = {a.est, a.prev, a.aux};
            const float* labs[3] = {a.box_lab, a.prev_lab, a.box_lab};
            float* outs[3] = {a.g_est, a.g_prev, a.g_aux};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (!rows[q]) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    outs[q][4 * b + k] = a.w_center / (3.f * fB) * fminf(fmaxf(rows[q][4 * b + k] - labs[q][4 * b + k], -1.f), 1.f);
                const float p = rows[q][4 * b + 3];
                outs[q][4 * b + 3] = a.w_angle / fB * cosf(p) * fminf(fmaxf(sinf(p) - sinf(labs[q][4 * b + 3]), -1.f), 1.f);
            }
            if (a.motion_cls) {
                float p1;
                const int y = a.state[b] != 0;
                nll2(a.motion_cls[2 * b], a.motion_cls[2 * b + 1], y, p1);
                a.g_mcls[2 * b + 0] = a.w_mcls / fB * ((1.f - p1) - (y ? 0.f : 1.f));
                a.g_mcls[2 * b + 1] = a.w_mcls / fB * (p1 - (y ? 1.f : 0.f));
            }
        }
    }
}

}  // namespace

// losses[12] = {weighted total, motion_cls, center, angle, center_prev, angle_prev, seg, center_aux, center_motion, angle_aux,
// angle_motion, bc}; the g_* = gradients of the TOTAL w.r.t. the corresponding prediction (g_seg == NULL: losses only).
// scratch: 64 * 16 floats.  NULL predictions switch their terms off (see the header above).
extern "C" int o3d_m2track_loss(const float* seg_logits, const int64_t* seg_label, const float* bc_pred, const float* bc_a,
                                const float* bc_b, const float* motion_cls, const int64_t* state, const float* motion,
                                const float* motion_lab, const float* aux, const float* est, const float* prev,
                                const float* box_lab, const float* prev_lab, int B, int N, int K, float w_center, float w_angle,
                                float w_seg, float w_bc, float w_mcls, float cw0, float cw1, float* scratch, float* losses,
                                float* g_seg, float* g_bc, float* g_mcls, float* g_motion, float* g_aux, float* g_est,
                                float* g_prev, void* stream) {
    if (!seg_logits || !seg_label || !motion || !motion_lab || !aux || !box_lab || !scratch || !losses || B <= 0 || N <= 0 ||
        (bc_pred && (!bc_a || !bc_b || K <= 0 || N % 2 != 0)) || (motion_cls && !state) || (prev && !prev_lab) ||
        (g_seg && (!g_motion || !g_aux || (bc_pred && !g_bc) || (motion_cls && !g_mcls) || (est && !g_est) || (prev && !g_prev))))
        return O3D_EINVAL;
    M2LossArgs a{seg_logits, seg_label, bc_pred, bc_a, bc_b, motion_cls, state, motion, motion_lab, aux, est, prev, box_lab,
                 prev_lab, B, N, K, w_center, w_angle, w_seg, w_bc, w_mcls, cw0, cw1, scratch, losses, g_seg, g_bc, g_mcls,
                 g_motion, g_aux, g_est, g_prev};
    hipLaunchKernelGGL(m2_loss_sums_kernel, dim3(LG), dim3(LT), 0, o3d_stream(stream), a);
    hipLaunchKernelGGL(m2_loss_grads_kernel, dim3(LG), dim3(LT), 0, o3d_stream(stream), a);
    return o3d_launch_status();
}
