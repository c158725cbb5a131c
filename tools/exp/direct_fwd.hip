// direct_fwd.hip -- experiment (GPU box only): fp32-MFMA forward layer with NO LDS staging.
// Each wave owns 64 output channels x 128 positions; A (weights, float4 along k) and B (activations,
// float4 along positions, lane l31 <-> positions 4*l31+t of n-tile t) fragments come straight from
// global memory (L1/L2) into VGPRs in MFMA layout; BN+ReLU of the producer applied in registers.
// build: tools/exp/build.sh     run: tools/exp/direct_fwd [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

extern "C" int o3d_mlp_conv_fwd(const float*, const float*, const float*, const float*, int, int, int, int, float*,
                                float*, const float*, void*);
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct Frag { float4 b[4]; float4 a[2]; float4 sc, sh; };

template <int WAVES, bool XFORM, int STAGES, int MAP = 0>
__global__ __launch_bounds__(WAVES * 64) void direct_fwd_kernel(const float* __restrict__ X,
                                                                const float* __restrict__ W,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int Cin,
                                                                int Cout, int P, float* __restrict__ Y, int mode, long long* __restrict__ ts) {
    const long long t0 = clock64();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_per_b = P / 128;
    int tile = blockIdx.x, slab = blockIdx.y;
    if (MAP == 1) {   // 1-D grid; workgroup w runs on XCD w%8: keep the Cout/64 slabs of one position tile on one XCD
        const int nslab = Cout / (64 * WAVES);
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        slab = local % nslab;
        tile = (local / nslab) * 8 + xcd;
    }
    const int b = tile / tiles_per_b, p0 = (tile - b * tiles_per_b) * 128;
    const int co0 = (slab * WAVES + wave) * 64;
    const float* xb = (mode & 2) ? X + 4 * l31 : X + ((long)b * Cin) * P + p0 + 4 * l31;   // + k*P
    const float* wa0 = W + (long)(co0 + l31) * Cin + 4 * h;      // + 8g  (m-tile 0), +32*Cin (m-tile 1)
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    auto load = [&](Frag& f, int g) {
        const int kb = 8 * g + 4 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) f.b[s] = *reinterpret_cast<const float4*>(xb + (long)(kb + s) * P);
        f.a[0] = *reinterpret_cast<const float4*>(wa0 + 8 * g);
        f.a[1] = *reinterpret_cast<const float4*>(wa0 + 8 * g + 32 * (long)Cin);
        if (XFORM) {
            f.sc = *reinterpret_cast<const float4*>(scale + kb);
            f.sh = *reinterpret_cast<const float4*>(shift + kb);
        }
    };
    auto compute = [&](Frag& f) {
        const float sc[4] = {f.sc.x, f.sc.y, f.sc.z, f.sc.w}, sh[4] = {f.sh.x, f.sh.y, f.sh.z, f.sh.w};
        const float a0[4] = {f.a[0].x, f.a[0].y, f.a[0].z, f.a[0].w}, a1[4] = {f.a[1].x, f.a[1].y, f.a[1].z, f.a[1].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float bv[4] = {f.b[s].x, f.b[s].y, f.b[s].z, f.b[s].w};
            if (XFORM) {
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[t] = fmaxf(fmaf(bv[t], sc[s], sh[s]), 0.f);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], bv[t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], bv[t], acc[1][t], 0, 0, 0);
            }
        }
    };
    const int G = Cin / 8;     // G % STAGES == 0; branch-free ring so the compiler can count vmcnt exactly
    Frag f[STAGES];
#pragma unroll
    for (int i = 0; i < STAGES - 1; ++i) load(f[i], i);
    const long long t1 = clock64();
    for (int g = 0; g < G; g += STAGES) {
#pragma unroll
        for (int i = 0; i < STAGES; ++i) {
            const int gn = g + i + STAGES - 1;
            load(f[(i + STAGES - 1) % STAGES], gn < G ? gn : G - 1);   // tail: harmless re-load
            __builtin_amdgcn_sched_barrier(0);
            compute(f[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t2 = clock64();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * i + acc_row(r, h);
            float4 v = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
            if ((mode & 1) && v.x != 123.456f) continue;
            *reinterpret_cast<float4*>(&Y[((long)b * Cout + co) * P + p0 + 4 * l31]) = v;
        }
    if (ts && lane == 0) {
        const long long t3 = clock64();
        long long* d = ts + 4 * ((long)(blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave);
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3;
    }
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
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    int reps = argc > 1 ? atoi(argv[1]) : 10;
    struct Shape { const char* name; int B, Cin, Cout, P; };
    const Shape shapes[] = {{"S-SA3 256->256", 48, 256, 256, 4096}, {"S-SA2 128->128", 48, 128, 128, 8192},
                            {"S-SA2 128->256", 48, 128, 256, 8192}, {"S-SA1 64->64", 48, 64, 64, 16384},
                            {"S-SA1 64->128", 48, 64, 128, 16384},  {"RPN 256->256", 48, 256, 256, 1024}};
    for (const Shape& s : shapes) {
        const size_t nx = (size_t)s.B * s.Cin * s.P, ny = (size_t)s.B * s.Cout * s.P;
        float *X = dev_rand(nx), *W = dev_rand((size_t)s.Cin * s.Cout, 0.1f), *sc = dev_rand(s.Cin), *sh = dev_rand(s.Cin);
        float *Y0, *Y1; CK(hipMalloc(&Y0, ny * 4)); CK(hipMalloc(&Y1, ny * 4));
        const double gf = 2.0 * s.Cin * s.Cout * (double)s.B * s.P * 1e-9;
        float t_ref = time_ms([&] { o3d_mlp_conv_fwd(X, W, sc, sh, s.B, s.Cin, s.Cout, s.P, Y0, nullptr, nullptr, 0); }, reps);
        const int tiles = s.B * (s.P / 128);
        int mode = 0;
        long long* ts = nullptr;
        auto run = [&](int waves) {
            const dim3 grid(tiles, s.Cout / (64 * (waves == 4 ? 4 : 1))), grid4(tiles, s.Cout / 256);
            if (waves == 4) hipLaunchKernelGGL((direct_fwd_kernel<4, true, 4>), grid, dim3(256), 0, 0, X, W, sc, sh, s.Cin, s.Cout, s.P, Y1, mode, ts);
            else if (waves == 2) hipLaunchKernelGGL((direct_fwd_kernel<1, true, 2>), dim3(tiles, s.Cout / 64), dim3(64), 0, 0, X, W, sc, sh, s.Cin, s.Cout, s.P, Y1, mode, ts);
            else if (waves == 5) hipLaunchKernelGGL((direct_fwd_kernel<4, true, 2>), grid4, dim3(256), 0, 0, X, W, sc, sh, s.Cin, s.Cout, s.P, Y1, mode, ts);
            else if (waves == 6) hipLaunchKernelGGL((direct_fwd_kernel<1, true, 2, 1>), dim3(tiles * (s.Cout / 64)), dim3(64), 0, 0, X, W, sc, sh, s.Cin, s.Cout, s.P, Y1, mode, ts);
            else if (waves == 7) hipLaunchKernelGGL((direct_fwd_kernel<2, true, 2>), dim3(tiles, s.Cout / 128), dim3(128), 0, 0, X, W, sc, sh, s.Cin, s.Cout, s.P, Y1, mode, ts);
            else hipLaunchKernelGGL((direct_fwd_kernel<1, true, 4>), grid, dim3(64), 0, 0, X, W, sc, sh, s.Cin, s.Cout, s.P, Y1, mode, ts);
        };
        printf("%-16s %6.2f GF | lds-staged %.3f ms %6.1f TF |", s.name, gf, t_ref, gf / t_ref);
        for (int waves : {2, 5, 6, 7}) {   // 2: 1 wave/WG 2 stages; 5: 4 waves/WG 2 stages; 6: variant 2 + XCD map; 7: 2 waves/WG
            if (waves == 5 && s.Cout % 256) continue;
            if (waves == 7 && s.Cout % 128) continue;
            if (waves == 6 && (tiles % 8)) continue;
            float t = time_ms([&] { run(waves); }, reps);
            printf(" direct w%d %.3f ms %6.1f TF |", waves, t, gf / t);
        }
        for (int m : {3}) {
            mode = m;
            float t = time_ms([&] { run(1); }, reps);
            printf(" mode%d w1 %.3f ms %6.1f TF |", m, t, gf / t);
        }
        mode = 0;
        {   // per-wave cycle stamps (s_memtime ticks at 100 MHz on gfx9: report in microseconds)
            const size_t nw = (size_t)tiles * (s.Cout / 64);
            CK(hipMalloc(&ts, nw * 4 * sizeof(long long)));
            for (int v : {2}) {
                run(v); CK(hipDeviceSynchronize());
                std::vector<long long> h(nw * 4);
                CK(hipMemcpy(h.data(), ts, nw * 4 * sizeof(long long), hipMemcpyDeviceToHost));
                double pro = 0, loop = 0, epi = 0; long long tmin = h[0], tmax = h[3]; (void)0;
                for (size_t i = 0; i < nw; ++i) { pro += h[4*i+1]-h[4*i]; loop += h[4*i+2]-h[4*i+1]; epi += h[4*i+3]-h[4*i+2]; tmin = std::min(tmin, h[4*i]); tmax = std::max(tmax, h[4*i+3]); }
                float tk = time_ms([&] { run(v); }, 3);
                printf("\n    stamps v%d: per wave prologue %.0f loop %.0f epilogue %.0f ticks; kernel span %lld ticks = %.3f ms -> %.2f GHz; waves %zu", v, pro / nw, loop / nw, epi / nw, tmax - tmin, tk, (tmax - tmin) / (tk * 1e6), nw);
            }
            printf("\n   ");
            CK(hipFree(ts)); ts = nullptr;
        }
        run(6);
        // correctness of the last variant
        std::vector<float> a(ny), bb(ny);
        CK(hipMemcpy(a.data(), Y0, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bb.data(), Y1, ny * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t i = 0; i < ny; ++i) { md = fmax(md, fabs((double)a[i] - bb[i])); mx = fmax(mx, fabs((double)a[i])); }
        printf(" maxdiff %.2e (scale %.2e)\n", md, mx);
        for (float* p : {X, W, sc, sh, Y0, Y1}) CK(hipFree(p));
    }
    return 0;
}
