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
