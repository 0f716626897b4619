// Microbenchmark: how MFMA phases and VALU phases of the two waves of a SIMD overlap (gfx950).
// 512-thread workgroups (2 waves per SIMD), one per CU.  Each wave runs ITER iterations of a body:
//   mode 0: M MFMAs (6 accumulators, dependent triples)            -> cycles per MFMA
//   mode 1: V VALU FMAs (16 independent chains)                    -> cycles per VALU
//   mode 2: M MFMAs then V VALU, every wave the same order          (phases in sequence inside a wave)
//   mode 3: as 2, waves 4-7 start with the VALU phase               (the two waves of a SIMD in opposite phases)
//   mode 4: waves 0-3 only MFMA (2 bodies' worth), waves 4-7 only VALU (split roles)
//   mode 5: as 3 with s_setprio 1 around the MFMA phase
//   mode 6: M MFMAs with V VALU interleaved in the same wave (V/M per MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int MODE, int M, int V>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[6];
    for (int b = 0; b < 6; ++b) for (int r = 0; r < 16; ++r) acc[b][r] = (float)(lane + b + r) * 1e-3f;
    half8 a, bb;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); bb[j] = (_Float16)(0.02f * (lane - j)); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 1.0f + 1e-3f * (lane + j);
    const float c1 = 1.0000001f, c2 = 1e-7f;
    auto mf = [&]() {
#pragma unroll
        for (int i = 0; i < M; ++i) acc[(i / 3) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, acc[(i / 3) % 6], 0, 0, 0);
    };
    auto va = [&]() {
#pragma unroll
        for (int i = 0; i < V; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], c1, c2);
    };
    const bool hiw = wave >= 4;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) mf();
        else if (MODE == 1) va();
        else if (MODE == 2) { mf(); __builtin_amdgcn_sched_barrier(0); va(); __builtin_amdgcn_sched_barrier(0); }
        else if (MODE == 3 || MODE == 5) {
            if (hiw) { va(); __builtin_amdgcn_sched_barrier(0); if (MODE == 5) __builtin_amdgcn_s_setprio(1); mf(); if (MODE == 5) __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
            else { if (MODE == 5) __builtin_amdgcn_s_setprio(1); mf(); if (MODE == 5) __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); va(); __builtin_amdgcn_sched_barrier(0); }
        } else if (MODE == 4) {
            if (hiw) { va(); va(); } else { mf(); mf(); }
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                acc[(i / 3) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, acc[(i / 3) % 6], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < V / M; ++j) v[(i * (V / M) + j) % 16] = __builtin_fmaf(v[(i * (V / M) + j) % 16], c1, c2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int b = 0; b < 6; ++b) for (int r = 0; r < 16; ++r) s += acc[b][r];
    for (int j = 0; j < 16; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int M, int V>
void run(const char* name, float* out, long long* cyc, long long* hc) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, M, V>), dim3(blocks), dim3(512), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, M, V>), dim3(blocks), dim3(512), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, cyc, sizeof(long long) * blocks * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)hc[b * 8 + w];
    lo /= blocks * 4.0 * iters; hi /= blocks * 4.0 * iters;
    printf("%-44s M=%2d V=%3d  ms=%7.3f  us/iter=%7.3f  cyc/iter waves0-3=%8.1f waves4-7=%8.1f\n", name, M, V, ms, ms * 1e3 / iters, lo, hi);
}

int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    long long* hc = (long long*)malloc(256 * 8 * 8);
    run<0, 18, 0>("MFMA only", out, cyc, hc);
    run<1, 0, 128>("VALU only", out, cyc, hc);
    run<2, 18, 128>("MFMA then VALU, same order", out, cyc, hc);
    run<3, 18, 128>("MFMA then VALU, partner opposite phase", out, cyc, hc);
    run<5, 18, 128>("  + s_setprio 1 on the MFMA phase", out, cyc, hc);
    run<4, 18, 128>("split roles (waves 0-3 MFMA, 4-7 VALU)", out, cyc, hc);
    run<6, 18, 36>("interleaved in one wave, 2 VALU per MFMA", out, cyc, hc);
    run<6, 18, 72>("interleaved in one wave, 4 VALU per MFMA", out, cyc, hc);
    run<6, 18, 108>("interleaved in one wave, 6 VALU per MFMA", out, cyc, hc);
    run<6, 18, 144>("interleaved in one wave, 8 VALU per MFMA", out, cyc, hc);
    run<2, 18, 72>("MFMA then VALU (72)", out, cyc, hc);
    run<3, 18, 72>("MFMA then VALU (72), opposite", out, cyc, hc);
    return 0;
}
