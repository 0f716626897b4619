// stand-alone timing harness of the vec_proj product (adf_gemm_f16x3_kernel<0,3,2,3,...>) with the G16_ABL ablation bits of
// gemm16.hip: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DG16_ABL=<bits> -o g16 gemm16_vecproj_harness.hip
// (ADF_GEMM_WREG=0 / ADF_GEMM_W8=0 select the earlier forms at run time)
#include <stdarg.h>
#include "../../adsorbdiff_amd/csrc/gemm16.hip"
#include "../../adsorbdiff_amd/csrc/mlp16.hip"
void adf_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fprintf(stderr, "\n"); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fillf(float* p, size_t n, float a, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = a * ((h & 0xffff) / 32768.0f - 1.0f);
    }
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 200000, H = 512;
    float *vec, *w, *v1, *dot, *cat, *mag, *isc;
    _Float16 *hi, *lo; void* frag; unsigned int* scratch;
    CK(hipMalloc(&vec, (size_t)N * 3 * H * 4)); CK(hipMalloc(&w, (size_t)2 * H * H * 4)); CK(hipMalloc(&v1, (size_t)N * 3 * H * 4));
    CK(hipMalloc(&dot, (size_t)N * H * 4)); CK(hipMalloc(&cat, (size_t)N * H * 4)); CK(hipMalloc(&mag, (size_t)3 * N * 4));
    CK(hipMalloc(&hi, (size_t)2 * H * H * 2)); CK(hipMalloc(&lo, (size_t)2 * H * H * 2)); CK(hipMalloc(&frag, (size_t)2 * H * H * 4));
    CK(hipMalloc(&isc, 4)); CK(hipMalloc(&scratch, 4));
    fillf<<<1024, 256>>>(vec, (size_t)N * 3 * H, 1.f, 1); fillf<<<256, 256>>>(w, (size_t)2 * H * H, .05f, 2);
    adf_w16 W = {hi, lo, isc, nullptr, nullptr};
    if (adf_split_weight(w, (long long)2 * H * H, &W, scratch, 0, H, H, nullptr, 2) != ADF_OK) return 2;
    if (adf_pack_frag(&W, 2 * H, H, frag, 0) != ADF_OK) return 2;
    adf_lift lf = {mag, 3ll * N};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wr = 0; wr < 2; ++wr) {
        adf_epi ep = {};
        ep.v1 = v1; ep.dotw = dot; ep.cat = cat; ep.H = H; W.frag = wr ? frag : nullptr;
        if (adf_launch_rowmag(vec, H, H, nullptr, 0, 3ll * N, mag, 0) != ADF_OK) return 2;
        ep.rmag = mag;
        for (int it = 0; it < 2; ++it) if (adf_launch_gemm16_fused(vec, H, &W, N, H, H, 3, &ep, 0, nullptr) != ADF_OK) return 2;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        const int reps = 5;
        for (int it = 0; it < reps; ++it) adf_launch_gemm16_fused(vec, H, &W, N, H, H, 3, &ep, 0, nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)N * 3 * 1024.0 * 512 * 6;
        printf("G16_ABL=%d WR=%d N=%d: %.3f ms per launch, %.0f TFLOP/s issued\n", G16_ABL, wr, N, ms / reps, flops / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}
