// atomic_probe.hip -- standalone micro-benchmark (GPU box only; NOT part of the product).
// Question (round 4): can the BatchNorm statistics of a GEMM launch be accumulated with global fp64
// atomics from the producers' epilogues (no partial rows, no finalize launch), and what does that cost?
//   mode 0: T workgroups x 256 threads, each thread stores 2 floats of a [T][2][256] partial table   (today's epilogue)
//   mode 1: each thread issues 2 non-returning fp64 atomic adds into acc[shard][2][256], shard = block % S
//   mode 2: the same with fp32 atomics
//   mode 3: the same fp64 atomics, but only after a wave-level pre-reduction is skipped -- 4 waves hit the SAME 128 addresses
//           (what 4 row slabs of different column tiles do): contention pattern of the real kernel
// Each mode is bracketed by a dummy streaming body (reads 64 KB per workgroup) so the atomics overlap real traffic.
// Also: cost of a dependent tiny kernel behind a big one in a captured graph (the launch seam a finalize costs today).
// build: tools/exp/build.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void stats_kernel(const float4* src, long nsrc, float* part, double* acc64, float* acc32,
                                                    int shards, int body) {
    const int t = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    // body: `body` dwordx4 loads per thread of a streaming read
    const long base = ((long)blockIdx.x * 256 + t) % (nsrc - 256L * body);
    for (int i = 0; i < body; ++i) {
        const float4 v = src[base + 256L * i];
        s1 += v.x + v.z;
        s2 += v.y * v.w;
    }
    if constexpr (MODE == 0) {
        part[((long)blockIdx.x * 2 + 0) * 256 + t] = s1;
        part[((long)blockIdx.x * 2 + 1) * 256 + t] = s2;
    } else if constexpr (MODE == 1) {
        double* a = acc64 + (long)(blockIdx.x % shards) * 512;
        __hip_atomic_fetch_add(a + t, (double)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a + 256 + t, (double)s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (MODE == 2) {
        float* a = acc32 + (long)(blockIdx.x % shards) * 512;
        __hip_atomic_fetch_add(a + t, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a + 256 + t, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        double* a = acc64 + (long)(blockIdx.x % shards) * 512;
        __hip_atomic_fetch_add(a + (t & 127), (double)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a + 256 + (t & 127), (double)s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void tiny_kernel(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

// consumer-side prologue cost: every workgroup derives 256 channels' (scale, shift) from S shards of fp64 sums
__global__ __launch_bounds__(256) void consume_kernel(const double* acc64, int shards, const float* gamma, const float* beta,
                                                      float* out, double inv_count) {
    __shared__ float sc[256], sh[256];
    const int t = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < shards; ++s) { s1 += acc64[(long)s * 512 + t]; s2 += acc64[(long)s * 512 + 256 + t]; }
    const double mean = s1 * inv_count;
    const double var = s2 * inv_count - mean * mean;
    const float istd = rsqrtf((float)var + 1e-5f);
    sc[t] = gamma[t] * istd;
    sh[t] = beta[t] - (float)mean * sc[t];
    __syncthreads();
    out[(long)blockIdx.x * 256 + t] = sc[t] + sh[(t + 1) & 255];
}

int main() {
    const long nsrc = 64L << 20;     // 1 GB of float4
    float4* src; float* part; double* acc64; float* acc32; float* gamma; float* out;
    CK(hipMalloc(&src, nsrc * sizeof(float4)));
    CK(hipMemset(src, 0, nsrc * sizeof(float4)));
    CK(hipMalloc(&part, 16384L * 512 * 4));
    CK(hipMalloc(&acc64, 256L * 512 * 8));
    CK(hipMalloc(&acc32, 256L * 512 * 4));
    CK(hipMalloc(&gamma, 1024 * 4));
    CK(hipMalloc(&out, 16384L * 256 * 4));
    CK(hipMemset(acc64, 0, 256L * 512 * 8));
    CK(hipMemset(acc32, 0, 256L * 512 * 4));
    CK(hipMemset(gamma, 0, 1024 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto fn, int reps) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / reps;
    };
    printf("# T workgroups x 256 threads; body = dwordx4 loads per thread; us per launch (back-to-back launches, incl. launch gap)\n");
    printf("%6s %5s %7s | %9s | %9s %9s %9s | %9s %9s | %9s\n", "T", "body", "", "store", "f64 S=1", "f64 S=8", "f64 S=64", "f32 S=8",
           "f32 S=64", "f64x4 S=8");
    for (int body : {0, 16, 64}) {
        for (int T : {96, 192, 768, 1536, 3072, 6144}) {
            auto run = [&](int mode, int S) {
                return timeit([&] {
                    switch (mode) {
                        case 0: hipLaunchKernelGGL(stats_kernel<0>, dim3(T), dim3(256), 0, st, src, nsrc, part, acc64, acc32, S, body); break;
                        case 1: hipLaunchKernelGGL(stats_kernel<1>, dim3(T), dim3(256), 0, st, src, nsrc, part, acc64, acc32, S, body); break;
                        case 2: hipLaunchKernelGGL(stats_kernel<2>, dim3(T), dim3(256), 0, st, src, nsrc, part, acc64, acc32, S, body); break;
                        default: hipLaunchKernelGGL(stats_kernel<3>, dim3(T), dim3(256), 0, st, src, nsrc, part, acc64, acc32, S, body); break;
                    }
                }, 50);
            };
            printf("%6d %5d %7s | %9.2f | %9.2f %9.2f %9.2f | %9.2f %9.2f | %9.2f\n", T, body, "", run(0, 1), run(1, 1), run(1, 8),
                   run(1, 64), run(2, 8), run(2, 64), run(3, 8));
        }
    }
    printf("# consumer prologue: 256 channels from S shards, T workgroups (us per launch)\n");
    for (int S : {1, 8, 64})
        for (int T : {192, 1536})
            printf("S=%2d T=%5d %9.2f\n", S, T, timeit([&] {
                hipLaunchKernelGGL(consume_kernel, dim3(T), dim3(256), 0, st, acc64, S, gamma, gamma + 256, out, 1e-5);
            }, 50));
    // launch seam inside a captured graph: big -> tiny -> big -> tiny ... vs big -> big
    for (int with_tiny = 0; with_tiny < 2; ++with_tiny) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 40; ++i) {
            hipLaunchKernelGGL(stats_kernel<0>, dim3(1536), dim3(256), 0, st, src, nsrc, part, acc64, acc32, 1, 16);
            if (with_tiny) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st, (float*)acc32);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const float us = timeit([&] { CK(hipGraphLaunch(ge, st)); }, 20);
        printf("graph of 40 x (1536-workgroup kernel%s): %.2f us per pair\n", with_tiny ? " + tiny dependent kernel" : "", us / 40);
    }
    return 0;
}
