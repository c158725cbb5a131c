// gemm_probe.hip -- standalone micro-benchmarks (GPU box only; NOT part of the product):
//   1. fp32-MFMA issue-rate ceiling on this box (no memory traffic)
//   2. the product's conv_fwd / conv_dgrad / conv_wgrad C-ABI entries at the BAT shapes
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/gemm_probe.hip -o tools/exp/gemm_probe \
//        -Lopen3dsot_amd/_lib -lo3dsot_hip -Wl,-rpath,'$ORIGIN/../../open3dsot_amd/_lib'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "../../include/o3dsot.h"

extern "C" int o3d_mlp_conv_fwd(const float*, const float*, const float*, const float*, int, int, int, int, float*,
                                float*, const float*, void*);
extern "C" int o3d_mlp_conv_dgrad(const float* dN, const float* dOut, const float* out, const int32_t* arg, int ns,
                                  const float* Y, const float* A1, const float* A2, const float* A3, const float* W,
                                  int B, int Cin, int Cout, int P, const float* Yprev, const float* scale_p,
                                  const float* shift_p, const float* mean_p, float* dNprev, float* part, void* stream);
extern "C" int o3d_mlp_conv_dgrad_wt(const float* dN, const float* dOut, const float* out, const int32_t* arg, int ns,
                                     const float* Y, const float* A1, const float* A2, const float* A3, const float* W,
                                     const float* Wt, const float* pk, int B, int Cin, int Cout, int P, const float* Yprev,
                                     const float* scale_p, const float* shift_p, const float* mean_p, float* dNprev,
                                     float* part, void* stream);
extern "C" int o3d_mlp_conv_wgrad(const float* dN, const float* dOut, const float* out, const int32_t* arg, int ns,
                                  const float* Y, const float* A1, const float* A2, const float* A3, const float* X,
                                  const float* in_scale, const float* in_shift, const float* xyz, const float* new_xyz,
                                  const float* feats, const int32_t* idx, int N, int C, int nxyz, float inv_radius,
                                  int B, int Cin, int Cout, int P, int nslices, float* part, float* dW, void* stream);

extern "C" long o3d_mlp_conv_wgrad2_scratch(int B, int Cin, int Cout, int P);
extern "C" int o3d_mlp_conv_wgrad2(const float* dN, const float* pk, int ns, const float* Y, const float* A1,
                                   const float* A2, const float* A3, const float* X, const float* in_scale,
                                   const float* in_shift, int B, int Cin, int Cout, int P, float* scratch, float* dW,
                                   void* stream);
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

// variants of the issue-rate probe: 8 accumulators (128 acc registers) like the direct kernel,
// MODE 1: + a dependent v_fma/v_max producing the B operand of every MFMA pair,
// MODE 2: MODE 1 + four dwordx4 loads per 32 MFMAs (L2-resident buffer)
template <int MODE>
__global__ __launch_bounds__(64) void mfma_peak8_kernel(float* out, const float* src, int iters) {
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    float a0 = threadIdx.x * 1e-3f, a1 = blockIdx.x * 1e-3f, sc = 1.0001f, sh = 1e-3f;
    float4 b[4];
    for (int s = 0; s < 4; ++s) b[s] = make_float4(0.1f * s, 0.2f, 0.3f, 0.4f);
    const float4* p = reinterpret_cast<const float4*>(src) + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int s = 0; s < 4; ++s) b[s] = p[(it * 4 + s) * 64 % 4096];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float bv[4] = {b[s].x, b[s].y, b[s].z, b[s].w};
            if (MODE >= 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[t] = fmaxf(fmaf(bv[t], sc, sh), 0.f);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[t], acc[1][t], 0, 0, 0);
            }
            if (MODE == 1) { b[s].x = bv[1]; b[s].y = bv[2]; b[s].z = bv[3]; b[s].w = bv[0]; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[i][t][r];
    if (s == 123.456f) out[0] = s;
}

static float* dev_rand(size_t n, float scale = 1.f) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() & 0xffff) / 32768.f - 1.f);
    float* d; CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

template <typename F>
static float time_ms(F f, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : "all";
    int reps = argc > 2 ? atoi(argv[2]) : 10;
    if (!strcmp(only, "all") || !strcmp(only, "peak")) {
        float* out; CK(hipMalloc(&out, 4));
        for (int wpc : {4, 8, 16}) {  // waves per CU: blocks of 4 waves, 256 CUs
            const int blocks = 256 * wpc / 4, iters = 4096;
            float ms = time_ms([&] { hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
            double flops = (double)blocks * 4 * iters * 32.0 * (2.0 * 32 * 32 * 2);
            printf("mfma_peak waves/CU=%2d: %.3f ms  %.1f TFLOP/s\n", wpc, ms, flops / ms * 1e-9);
        }
        float* src = dev_rand(1 << 16);
        for (int wpc : {4, 8}) {
            const int blocks = 256 * wpc, iters = 1024;
            const double flops = (double)blocks * iters * 32.0 * 4096.0;
            float m0 = time_ms([&] { hipLaunchKernelGGL(mfma_peak8_kernel<0>, dim3(blocks), dim3(64), 0, 0, out, src, iters); }, 5);
            float m1 = time_ms([&] { hipLaunchKernelGGL(mfma_peak8_kernel<1>, dim3(blocks), dim3(64), 0, 0, out, src, iters); }, 5);
            float m2 = time_ms([&] { hipLaunchKernelGGL(mfma_peak8_kernel<2>, dim3(blocks), dim3(64), 0, 0, out, src, iters); }, 5);
            printf("mfma_peak8 waves/CU=%d: plain %.1f TF | +valu %.1f TF | +valu+loads %.1f TF\n", wpc, flops / m0 * 1e-9,
                   flops / m1 * 1e-9, flops / m2 * 1e-9);
        }
    }
    struct Shape { const char* name; int B, Cin, Cout, P; };
    const Shape shapes[] = {{"S-SA3 256->256", 48, 256, 256, 4096}, {"T-SA3 256->256", 48, 256, 256, 2048},
                            {"S-SA2 128->128", 48, 128, 128, 8192}, {"S-SA2 128->256", 48, 128, 256, 8192},
                            {"S-SA1 64->64", 48, 64, 64, 16384},    {"S-SA1 64->128", 48, 64, 128, 16384},
                            {"RPN 256->256", 48, 256, 256, 1024},   {"xcorr 256->256", 48, 256, 256, 512}};
    for (const Shape& s : shapes) {
        if (strcmp(only, "all") && strcmp(only, "gemm") && !strstr(s.name, only)) continue;
        const size_t nx = (size_t)s.B * s.Cin * s.P, ny = (size_t)s.B * s.Cout * s.P;
        float *X = dev_rand(nx), *W = dev_rand((size_t)s.Cin * s.Cout, 0.1f), *Y = dev_rand(ny), *dN = dev_rand(ny);
        float *sc = dev_rand(s.Cin), *sh = dev_rand(s.Cin), *mu = dev_rand(s.Cin), *c = dev_rand(s.Cout);
        float *A1 = dev_rand(s.Cout), *A2 = dev_rand(s.Cout, 0.01f), *A3 = dev_rand(s.Cout, 0.01f);
        const int ntiles = s.B * (s.P / 128);
        float *part; CK(hipMalloc(&part, sizeof(float) * (size_t)ntiles * 2 * 256));
        float *dNp; CK(hipMalloc(&dNp, nx * sizeof(float)));
        const int nsl = 768 / (((s.Cin + 127) / 128) * ((s.Cout + 127) / 128));
        float *wpart; CK(hipMalloc(&wpart, sizeof(float) * (size_t)(nsl + 16) * s.Cin * s.Cout));
        float *dW; CK(hipMalloc(&dW, sizeof(float) * s.Cin * s.Cout));
        const double gf = 2.0 * s.Cin * s.Cout * (double)s.B * s.P * 1e-9;
        float t_f = time_ms([&] { o3d_mlp_conv_fwd(X, W, sc, sh, s.B, s.Cin, s.Cout, s.P, Y, part, c, 0); }, reps);
        float t_f0 = time_ms([&] { o3d_mlp_conv_fwd(X, W, sc, sh, s.B, s.Cin, s.Cout, s.P, Y, nullptr, nullptr, 0); }, reps);
        float t_d = time_ms([&] { o3d_mlp_conv_dgrad_wt(dN, 0, 0, 0, 32, Y, A1, A2, A3, W, W, 0, s.B, s.Cin, s.Cout, s.P, X, sc, sh, mu, dNp, part, 0); }, reps);
        float t_w = time_ms([&] { o3d_mlp_conv_wgrad(dN, 0, 0, 0, 32, Y, A1, A2, A3, X, sc, sh, 0, 0, 0, 0, 0, 0, 0, 1.f, s.B, s.Cin, s.Cout, s.P, nsl, wpart, dW, 0); }, reps);
        float* w2s; CK(hipMalloc(&w2s, sizeof(float) * o3d_mlp_conv_wgrad2_scratch(s.B, s.Cin, s.Cout, s.P)));
        float* dW2; CK(hipMalloc(&dW2, sizeof(float) * s.Cin * s.Cout));
        float t_w2 = time_ms([&] { o3d_mlp_conv_wgrad2(dN, 0, 32, Y, A1, A2, A3, X, sc, sh, s.B, s.Cin, s.Cout, s.P, w2s, dW2, 0); }, reps);
        {
            std::vector<float> h1((size_t)s.Cin * s.Cout), h2(h1.size());
            CK(hipMemcpy(h1.data(), dW, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), dW2, h2.size() * 4, hipMemcpyDeviceToHost));
            double md = 0, mx = 0;
            for (size_t i = 0; i < h1.size(); ++i) { md = fmax(md, fabs((double)h1[i] - h2[i])); mx = fmax(mx, fabs((double)h1[i])); }
            printf("[wgrad2 %.3f ms %5.1f TF relerr %.1e] ", t_w2, gf / t_w2, md / mx);
        }
        CK(hipFree(w2s)); CK(hipFree(dW2));
        printf("%-16s %6.2f GF | fwd-nostat %.3f ms | fwd %.3f ms %6.1f TF | dgrad %.3f ms %6.1f TF | wgrad %.3f ms %6.1f TF\n", s.name, gf,
               t_f0, t_f, gf / t_f, t_d, gf / t_d, t_w, gf / t_w);
        for (float* p : {X, W, Y, dN, sc, sh, mu, c, A1, A2, A3, part, dNp, wpart, dW}) CK(hipFree(p));
    }
    return 0;
}
