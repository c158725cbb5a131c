// Probe for HISTORY.md (round 1-2) section 9.2: the layer-0 backward reduce (S[c,n] = sum of dY0 over the columns that reference
// point n, T[c,ball] = sum over the ball's columns) WITHOUT LDS float atomics.
//
//   variant A (what csrc/compact.hip::reduce_c_kernel does today): per column one ds_add_f32 into S, in-lane folded
//              ds_add_f32 into T.  Measured in production: ~0.4 atomic lanes / clk / CU -> 0.65 ms per step.
//   variant B (candidate): a per-cloud TRANSPOSED index built once per SA call by a counting sort in LDS
//              (keys = (column chunk, point)), then per (cloud, channel pair): stage the chunk's dY in LDS with plain
//              stores and let thread n gather its own list from LDS; T = thread per ball over its contiguous range.
//              No float atomics; summation order fixed (lists are sorted) -> bitwise reproducible.
//
// Self-checking: both variants are compared with a double-precision host reference; prints max errors and the
// average time of each variant.
//   hipcc --offload-arch=gfx950 -O3 [-DPROBE_CH=2048 -DPROBE_CS=2] -o /tmp/reduce_probe tools/exp/reduce_gather_probe.hip
// Measured on the MI355X (round 1, SA1-like shape: 96 clouds x 1024 points x 512 balls, 64 channels, 661 K live
// columns; both variants match the reference to 2e-6):
//   atomic (production structure)                     297.7 us per launch
//   gather, CH 4096 / 2048 / 1024, CS 2 / 2 / 4       185.1 / 138.6 / 142.9 us   (CH 2048 CS 4: 149.6, CH 4096 CS 1: 179.5)
//   csr_build (once per SA call, all channels)         ~20 us
// -> 2.1x on this kernel (0.65 ms per step in production); NOT yet integrated into csrc/compact.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#ifndef PROBE_CH
#define PROBE_CH 4096
#endif
#ifndef PROBE_CS
#define PROBE_CS 2
#endif
constexpr int CH = PROBE_CH;     // columns per chunk (variant B), <= 65536 (16-bit list entries)
constexpr int CS = PROBE_CS;     // channels per workgroup (variant B)

// ---------------------------------------------------------------- variant A: LDS atomics (one channel per workgroup)
__global__ __launch_bounds__(256) void reduce_atomic(const float* __restrict__ dN, const float* __restrict__ Y0, long ldp,
                                                     const float* __restrict__ A1, const float* __restrict__ A2,
                                                     const float* __restrict__ A3, const int* __restrict__ gp,
                                                     const int* __restrict__ cball, const float* __restrict__ cw,
                                                     const int* __restrict__ ball_off, const int* __restrict__ ball_cnt,
                                                     int npoint, int ld, int C, float* __restrict__ S,
                                                     float* __restrict__ T, long ldz, int nballs) {
    extern __shared__ float acc[];          // [ld] then [npoint]
    const int cloud = blockIdx.x / C, c = blockIdx.x % C;
    const int pbase = cloud * ld, bbase = cloud * npoint;
    float* tacc = acc + ld;
    for (int i = threadIdx.x; i < ld + npoint; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int q0 = ball_off[bbase], q1 = ball_off[bbase + npoint - 1] + ball_cnt[bbase + npoint - 1];
    const float a1 = A1[c], a2 = A2[c], a3 = A3[c];
    for (int base = q0 & ~3; base < q1; base += 1024) {
        const int q = base + 4 * threadIdx.x;
        if (q >= q1) continue;
        const int4 g4 = *reinterpret_cast<const int4*>(&gp[q]);
        const int4 b4 = *reinterpret_cast<const int4*>(&cball[q]);
        const float4 w4 = *reinterpret_cast<const float4*>(&cw[q]);
        const float4 d = *reinterpret_cast<const float4*>(&dN[(long)c * ldp + q]);
        const float4 y = *reinterpret_cast<const float4*>(&Y0[(long)c * ldp + q]);
        const int gq[4] = {g4.x, g4.y, g4.z, g4.w}, bq[4] = {b4.x, b4.y, b4.z, b4.w};
        const float wq[4] = {w4.x, w4.y, w4.z, w4.w}, dv[4] = {d.x, d.y, d.z, d.w}, yv[4] = {y.x, y.y, y.z, y.w};
        float run = 0.f;
        int jr = -1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (q + t < q0 || q + t >= q1) continue;
            const float dy = fmaf(a1, dv[t], wq[t] * fmaf(a2, yv[t], a3));
            atomicAdd(&acc[gq[t] - pbase], dy);
            const int j = bq[t] - bbase;
            if (j == jr) {
                run += dy;
            } else {
                if (jr >= 0) atomicAdd(&tacc[jr], run);
                jr = j;
                run = dy;
            }
        }
        if (jr >= 0) atomicAdd(&tacc[jr], run);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ld; i += 256) S[(long)c * ldz + pbase + i] = acc[i];
    for (int i = threadIdx.x; i < npoint; i += 256) T[(long)c * nballs + bbase + i] = tacc[i];
}

// ---------------------------------------------------------------- variant B: transposed index + gather from LDS
// One workgroup per cloud.  Output: perm[q0 .. q1) = the cloud's columns sorted by (chunk, point, column);
// poff[cloud][k][n] = start (relative to q0) of point n's list inside chunk k, poff[cloud][k][ld] = its end.
// `nchunk_max` chunks of CH columns cover the worst-case span npoint*ns.
__global__ __launch_bounds__(1024) void csr_build(const int* __restrict__ gp, const int* __restrict__ ball_off,
                                                  const int* __restrict__ ball_cnt, int npoint, int ld, int nchunk_max,
                                                  int* __restrict__ perm, int* __restrict__ poff) {
    extern __shared__ int sh[];             // cnt[nchunk_max*ld + 1], scan scratch [1024]
    const int cloud = blockIdx.x;
    const int pbase = cloud * ld, bbase = cloud * npoint;
    const int q0 = ball_off[bbase], q1 = ball_off[bbase + npoint - 1] + ball_cnt[bbase + npoint - 1];
    const int q0a = q0 & ~3;                // chunk 0 starts at the float4-aligned column at or before q0
    const int nbin = nchunk_max * ld;
    int* cnt = sh;
    int* part = sh + nbin + 1;
    for (int i = threadIdx.x; i <= nbin; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int q = q0 + threadIdx.x; q < q1; q += 1024) atomicAdd(&cnt[((q - q0a) / CH) * ld + (gp[q] - pbase)], 1);
    __syncthreads();
    // exclusive scan of cnt[0 .. nbin) -> start offsets (kept in cnt), total in cnt[nbin]
    const int per = (nbin + 1023) / 1024, i0 = threadIdx.x * per;
    int local = 0;
    for (int i = i0; i < i0 + per && i < nbin; ++i) local += cnt[i];
    part[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - local;
    for (int i = i0; i < i0 + per && i < nbin; ++i) {
        const int c = cnt[i];
        cnt[i] = run;
        run += c;
    }
    if (threadIdx.x == 1023) cnt[nbin] = part[1023];
    __syncthreads();
    int* po = poff + (long)cloud * (nbin + 1);
    for (int i = threadIdx.x; i <= nbin; i += 1024) po[i] = cnt[i];
    __syncthreads();
    // fill: cursor = cnt (advanced by int atomics); order inside a list is restored by the sort below
    for (int q = q0 + threadIdx.x; q < q1; q += 1024) {
        const int k = atomicAdd(&cnt[((q - q0a) / CH) * ld + (gp[q] - pbase)], 1);
        perm[q0 + k] = q;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbin; i += 1024) {      // insertion sort of each (short) list: fixed summation order
        const int s = po[i], e = po[i + 1];
        for (int a = s + 1; a < e; ++a) {
            const int v = perm[q0 + a];
            int b = a - 1;
            while (b >= s && perm[q0 + b] > v) { perm[q0 + b + 1] = perm[q0 + b]; --b; }
            perm[q0 + b + 1] = v;
        }
    }
}

__global__ __launch_bounds__(256) void reduce_gather(const float* __restrict__ dN, const float* __restrict__ Y0, long ldp,
                                                     const float* __restrict__ A1, const float* __restrict__ A2,
                                                     const float* __restrict__ A3, const float* __restrict__ cw,
                                                     const int* __restrict__ ball_off, const int* __restrict__ ball_cnt,
                                                     const int* __restrict__ perm, const int* __restrict__ poff,
                                                     int npoint, int ld, int nchunk_max, int C, float* __restrict__ S,
                                                     float* __restrict__ T, long ldz, int nballs) {
    extern __shared__ float sm[];           // dy[CS][CH], sacc[CS][ld], tacc[CS][npoint], offs [ld+1] ints, list [CH] u16
    const int slabs = C / CS;
    const int cloud = blockIdx.x / slabs, c0 = (blockIdx.x % slabs) * CS;
    const int pbase = cloud * ld, bbase = cloud * npoint;
    float* dy = sm;
    float* sacc = dy + CS * CH;
    float* tacc = sacc + CS * ld;
    int* offs = reinterpret_cast<int*>(tacc + CS * npoint);
    uint16_t* lst = reinterpret_cast<uint16_t*>(offs + ld + 1);
    const int q0 = ball_off[bbase], q1 = ball_off[bbase + npoint - 1] + ball_cnt[bbase + npoint - 1];
    const int q0a = q0 & ~3;
    const int nbin = nchunk_max * ld;
    const int* po = poff + (long)cloud * (nbin + 1);
    for (int i = threadIdx.x; i < CS * (ld + npoint); i += 256) sacc[i] = 0.f;
    float a1[CS], a2[CS], a3[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) { a1[c] = A1[c0 + c]; a2[c] = A2[c0 + c]; a3[c] = A3[c0 + c]; }
    for (int k = 0, lo = q0a; lo < q1; ++k, lo += CH) {
        const int hi = lo + CH < q1 ? lo + CH : q1;
        __syncthreads();                    // previous chunk fully consumed (and the zero fill on the first pass)
        // phase A: dY of the chunk into LDS, plain stores; this chunk's lists and offsets beside it
        for (int i = 4 * threadIdx.x; i < CH; i += 1024) {
            const int q = lo + i;
            if (q >= hi) break;
            const float4 w4 = *reinterpret_cast<const float4*>(&cw[q]);
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                const float4 d = *reinterpret_cast<const float4*>(&dN[(long)(c0 + c) * ldp + q]);
                const float4 y = *reinterpret_cast<const float4*>(&Y0[(long)(c0 + c) * ldp + q]);
                float4 o;
                o.x = fmaf(a1[c], d.x, w4.x * fmaf(a2[c], y.x, a3[c]));
                o.y = fmaf(a1[c], d.y, w4.y * fmaf(a2[c], y.y, a3[c]));
                o.z = fmaf(a1[c], d.z, w4.z * fmaf(a2[c], y.z, a3[c]));
                o.w = fmaf(a1[c], d.w, w4.w * fmaf(a2[c], y.w, a3[c]));
                *reinterpret_cast<float4*>(&dy[c * CH + i]) = o;
            }
        }
        const int l0 = po[k * ld], l1 = po[(k + 1) * ld];       // this chunk's slice of perm
        for (int i = threadIdx.x; i < l1 - l0; i += 256) lst[i] = (uint16_t)(perm[q0 + l0 + i] - lo);
        for (int i = threadIdx.x; i <= ld; i += 256) offs[i] = po[k * ld + i] - l0;
        __syncthreads();
        // phase B: thread n walks its own list; thread j walks its ball's intersection with the chunk
        for (int n = threadIdx.x; n < ld; n += 256) {
            float s[CS];
#pragma unroll
            for (int c = 0; c < CS; ++c) s[c] = 0.f;
            for (int a = offs[n]; a < offs[n + 1]; ++a) {
                const int i = lst[a];
#pragma unroll
                for (int c = 0; c < CS; ++c) s[c] += dy[c * CH + i];
            }
#pragma unroll
            for (int c = 0; c < CS; ++c) sacc[c * ld + n] += s[c];
        }
        for (int j = threadIdx.x; j < npoint; j += 256) {
            const int b0 = ball_off[bbase + j], b1 = b0 + ball_cnt[bbase + j];
            const int s0 = b0 > lo ? b0 : lo, s1 = b1 < hi ? b1 : hi;
            if (s0 >= s1) continue;
            float t[CS];
#pragma unroll
            for (int c = 0; c < CS; ++c) t[c] = 0.f;
            for (int q = s0; q < s1; ++q)
#pragma unroll
                for (int c = 0; c < CS; ++c) t[c] += dy[c * CH + (q - lo)];
#pragma unroll
            for (int c = 0; c < CS; ++c) tacc[c * npoint + j] += t[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CS * ld; i += 256) {
        const int c = i / ld, n = i - c * ld;
        S[(long)(c0 + c) * ldz + pbase + n] = sacc[i];
    }
    for (int i = threadIdx.x; i < CS * npoint; i += 256) {
        const int c = i / npoint, j = i - c * npoint;
        T[(long)(c0 + c) * nballs + bbase + j] = tacc[i];
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 96, ld = 1024, npoint = 512, ns = 32, C = 64;    // SA1, both segments as clouds
    const int nballs = B * npoint;
    const long ldp = (long)nballs * ns, ldz = (long)B * ld;
    std::mt19937 rng(7);
    std::vector<int> ball_cnt(nballs), ball_off(nballs), gp(ldp, 0), cball(ldp, nballs);
    std::vector<float> cw(ldp, 0.f);
    long q = 0;
    for (int ball = 0; ball < nballs; ++ball) {
        const int cnt = 4 + rng() % 20;                       // ~13 distinct neighbours on average
        const int cloud = ball / npoint;
        std::vector<int> pts;
        const int centre = rng() % ld;
        while ((int)pts.size() < cnt) {                       // neighbours cluster around the centre like a real ball
            const int n = (centre + (int)(rng() % 96) - 48 + ld) % ld;
            if (std::find(pts.begin(), pts.end(), n) == pts.end()) pts.push_back(n);
        }
        std::sort(pts.begin(), pts.end());
        ball_cnt[ball] = cnt;
        ball_off[ball] = (int)q;
        for (int k = 0; k < cnt; ++k, ++q) {
            gp[q] = cloud * ld + pts[k];
            cball[q] = ball;
            cw[q] = k == 0 ? (float)(1 + ns - cnt) : 1.f;
        }
    }
    const long live = q;
    printf("CH %d CS %d | ", CH, CS);
    printf("clouds %d, live columns %ld of %ld slots (%.1f per ball)\n", B, live, ldp, (double)live / nballs);
    std::vector<float> dN((size_t)C * ldp), Y0((size_t)C * ldp), A1(C), A2(C), A3(C);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    for (auto& v : dN) v = U(rng);
    for (auto& v : Y0) v = U(rng);
    for (int c = 0; c < C; ++c) { A1[c] = 1.f + 0.3f * U(rng); A2[c] = 0.05f * U(rng); A3[c] = 0.05f * U(rng); }
    // host reference (two channels are enough for the check)
    const int CHK = 2;
    std::vector<double> Sref((size_t)CHK * ldz, 0.0), Tref((size_t)CHK * nballs, 0.0);
    for (int c = 0; c < CHK; ++c)
        for (long k = 0; k < live; ++k) {
            const double dyv = (double)A1[c] * dN[(size_t)c * ldp + k] + (double)cw[k] * ((double)A2[c] * Y0[(size_t)c * ldp + k] + A3[c]);
            Sref[(size_t)c * ldz + gp[k]] += dyv;
            Tref[(size_t)c * nballs + cball[k]] += dyv;
        }
    float *d_dN, *d_Y0, *d_A1, *d_A2, *d_A3, *d_cw, *d_S, *d_T;
    int *d_gp, *d_cball, *d_off, *d_cnt, *d_perm, *d_poff;
    const int nchunk_max = (npoint * ns + 4 + CH - 1) / CH;
    CK(hipMalloc(&d_dN, dN.size() * 4)); CK(hipMalloc(&d_Y0, Y0.size() * 4));
    CK(hipMalloc(&d_A1, C * 4)); CK(hipMalloc(&d_A2, C * 4)); CK(hipMalloc(&d_A3, C * 4));
    CK(hipMalloc(&d_cw, ldp * 4)); CK(hipMalloc(&d_gp, ldp * 4)); CK(hipMalloc(&d_cball, ldp * 4));
    CK(hipMalloc(&d_off, nballs * 4)); CK(hipMalloc(&d_cnt, nballs * 4)); CK(hipMalloc(&d_perm, ldp * 4));
    CK(hipMalloc(&d_poff, (size_t)B * (nchunk_max * ld + 1) * 4));
    CK(hipMalloc(&d_S, (size_t)C * ldz * 4)); CK(hipMalloc(&d_T, (size_t)C * nballs * 4));
    CK(hipMemcpy(d_dN, dN.data(), dN.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_Y0, Y0.data(), Y0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_A1, A1.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_A2, A2.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_A3, A3.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cw, cw.data(), ldp * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_gp, gp.data(), ldp * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cball, cball.data(), ldp * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, ball_off.data(), nballs * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cnt, ball_cnt.data(), nballs * 4, hipMemcpyHostToDevice));
    const size_t ldsA = (size_t)(ld + npoint) * 4;
    const size_t ldsCsr = ((size_t)nchunk_max * ld + 1 + 1024) * 4;
    const size_t ldsB = ((size_t)CS * CH + (size_t)CS * (ld + npoint)) * 4 + ((size_t)ld + 1) * 4 + (size_t)CH * 2 + 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(csr_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsCsr));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(reduce_gather), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
    printf("LDS per workgroup: atomic %zu B, csr build %zu B, gather %zu B\n", ldsA, ldsCsr, ldsB);
    std::vector<float> S((size_t)C * ldz), T((size_t)C * nballs);
    auto check = [&](const char* name) {
        CK(hipMemcpy(S.data(), d_S, S.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(T.data(), d_T, T.size() * 4, hipMemcpyDeviceToHost));
        double es = 0, et = 0, ms = 0;
        for (size_t i = 0; i < Sref.size(); ++i) { es = std::max(es, std::fabs(S[i] - Sref[i])); ms = std::max(ms, std::fabs(Sref[i])); }
        for (size_t i = 0; i < Tref.size(); ++i) et = std::max(et, std::fabs(T[i] - Tref[i]));
        printf("%-8s max |S - ref| %.3e  max |T - ref| %.3e  (|S| max %.2f)  %s\n", name, es, et, ms,
               es < 1e-4 * ms && et < 1e-4 * ms ? "OK" : "MISMATCH");
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    const int it = 20;
    // ---- A
    CK(hipMemset(d_S, 0, S.size() * 4)); CK(hipMemset(d_T, 0, T.size() * 4));
    hipLaunchKernelGGL(reduce_atomic, dim3(B * C), dim3(256), ldsA, 0, d_dN, d_Y0, ldp, d_A1, d_A2, d_A3, d_gp, d_cball, d_cw,
                       d_off, d_cnt, npoint, ld, C, d_S, d_T, ldz, nballs);
    CK(hipDeviceSynchronize());
    check("atomic");
    CK(hipEventRecord(e0));
    for (int i = 0; i < it; ++i)
        hipLaunchKernelGGL(reduce_atomic, dim3(B * C), dim3(256), ldsA, 0, d_dN, d_Y0, ldp, d_A1, d_A2, d_A3, d_gp, d_cball,
                           d_cw, d_off, d_cnt, npoint, ld, C, d_S, d_T, ldz, nballs);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("atomic  : %.1f us per launch\n", 1e3 * ms / it);
    // ---- B
    CK(hipMemset(d_S, 0, S.size() * 4)); CK(hipMemset(d_T, 0, T.size() * 4));
    hipLaunchKernelGGL(csr_build, dim3(B), dim3(1024), ldsCsr, 0, d_gp, d_off, d_cnt, npoint, ld, nchunk_max, d_perm, d_poff);
    hipLaunchKernelGGL(reduce_gather, dim3(B * (C / CS)), dim3(256), ldsB, 0, d_dN, d_Y0, ldp, d_A1, d_A2, d_A3, d_cw, d_off,
                       d_cnt, d_perm, d_poff, npoint, ld, nchunk_max, C, d_S, d_T, ldz, nballs);
    CK(hipDeviceSynchronize());
    check("gather");
    CK(hipEventRecord(e0));
    for (int i = 0; i < it; ++i)
        hipLaunchKernelGGL(csr_build, dim3(B), dim3(1024), ldsCsr, 0, d_gp, d_off, d_cnt, npoint, ld, nchunk_max, d_perm, d_poff);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("csr     : %.1f us per launch (once per SA call)\n", 1e3 * ms / it);
    CK(hipEventRecord(e0));
    for (int i = 0; i < it; ++i)
        hipLaunchKernelGGL(reduce_gather, dim3(B * (C / CS)), dim3(256), ldsB, 0, d_dN, d_Y0, ldp, d_A1, d_A2, d_A3, d_cw,
                           d_off, d_cnt, d_perm, d_poff, npoint, ld, nchunk_max, C, d_S, d_T, ldz, nballs);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("gather  : %.1f us per launch\n", 1e3 * ms / it);
    return 0;
}
