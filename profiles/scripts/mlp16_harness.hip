// stand-alone timing harness of adf_mlp16_kernel (synthetic operands): hipcc -DML_ABL=x ... mlp_harness.hip
#include <stdarg.h>
#include "../../adsorbdiff_amd/csrc/mlp16.hip"
void adf_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fprintf(stderr, "\n"); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(float* p, size_t n, float a, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = a * ((h & 0xffff) / 32768.0f - 1.0f);
    }
}
__global__ void fillh(_Float16* p, size_t n, float a, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (_Float16)(a * ((h & 0xffff) / 32768.0f - 1.0f));
    }
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 200000, H = 512;
    float *A1, *A2, *rmag, *b0, *b2, *isc, *vec, *rec, *x, *dot, *vv;
    _Float16 *W0f, *W2f;
    CK(hipMalloc(&A1, (size_t)M * H * 4)); CK(hipMalloc(&A2, (size_t)M * H * 4)); CK(hipMalloc(&rmag, (size_t)M * 4));
    CK(hipMalloc(&b0, H * 4)); CK(hipMalloc(&b2, 3 * H * 4)); CK(hipMalloc(&isc, 4));
    CK(hipMalloc(&vec, (size_t)M * 3 * H * 4)); CK(hipMalloc(&rec, (size_t)(M + 1) * 5 * H * 4)); CK(hipMalloc(&x, (size_t)M * H * 4));
    CK(hipMalloc(&dot, (size_t)M * H * 4)); CK(hipMalloc(&vv, (size_t)M * 3 * H * 4));
    CK(hipMalloc(&W0f, (size_t)H * 2 * H * 4)); CK(hipMalloc(&W2f, (size_t)3 * H * H * 4));
    fill<<<1024, 256>>>(A1, (size_t)M * H, 1.f, 1); fill<<<1024, 256>>>(A2, (size_t)M * H, 1.f, 2);
    fill<<<1024, 256>>>(rmag, M, 0.f, 3); fill<<<64, 256>>>(b0, H, .1f, 4); fill<<<64, 256>>>(b2, 3 * H, .1f, 5);
    fill<<<1024, 256>>>(vec, (size_t)M * 3 * H, 1.f, 6); fill<<<1024, 256>>>(x, (size_t)M * H, 1.f, 7);
    fill<<<1024, 256>>>(dot, (size_t)M * H, 1.f, 8); fill<<<1024, 256>>>(vv, (size_t)M * 3 * H, 1.f, 9);
    fillh<<<1024, 256>>>(W0f, (size_t)H * 2 * H * 2, 600.f, 10); fillh<<<1024, 256>>>(W2f, (size_t)3 * H * H * 2, 600.f, 11);
    float one = 1.0f / 1024; CK(hipMemcpy(isc, &one, 4, hipMemcpyHostToDevice));
    float rm1 = 1.0f;   // rmag = 1 for every row
    { std::vector<float> r(M, rm1); CK(hipMemcpy(rmag, r.data(), (size_t)M * 4, hipMemcpyHostToDevice)); }
    adf_w16 w0 = {W0f, W0f, isc, nullptr}, w2 = {W2f, W2f, isc, b2};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int epi = 1; epi <= 2; ++epi) {
        adf_epi ep = {};
        ep.H = H; ep.lift_y = 1;
        float* pb; CK(hipMalloc(&pb, 128)); ep.cat = pb;
        if (epi == 1) { ep.vec_in = vec; ep.rec = rec; }
        else { ep.x = x; ep.vec = vec; ep.dot = dot; ep.vv = vv; ep.scale = 1.0f; }
        const float* a2 = epi == 2 ? A2 : nullptr;
        for (int it = 0; it < 2; ++it)
            if (adf_launch_mlp16(epi == 2 ? x : A1, a2, H, rmag, W0f, &w0, b0, W2f, &w2, M, H, epi, &ep, 0) != ADF_OK) return 2;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        const int reps = 5;
        for (int it = 0; it < reps; ++it) adf_launch_mlp16(epi == 2 ? x : A1, a2, H, rmag, W0f, &w0, b0, W2f, &w2, M, H, epi, &ep, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
#if ML_PROF
        {
            CK(hipMemset(pb, 0, 128));
            adf_launch_mlp16(epi == 2 ? x : A1, a2, H, rmag, W0f, &w0, b0, W2f, &w2, M, H, epi, &ep, 0);
            CK(hipDeviceSynchronize());
            unsigned long long c[16]; CK(hipMemcpy(c, pb, 128, hipMemcpyDeviceToHost));
            const double nt = (M + 63) / 64;
            const char* nm[8] = {"load A", "product 1", "y hand-off", "product 2 g0", "epilogue g0", "product 2 g1", "epilogue g1", ""};
            for (int w = 0; w < 2; ++w) { printf("  wave %d us per tile:", 4 * w); for (int k = 0; k < 7; ++k) printf(" %s %.1f |", nm[k], c[8 * w + k] / nt / 100.0); printf("\n"); }
        }
#endif
        const double flops = (double)M * ((epi == 2 ? 1024.0 : 512.0) * 512 + 512.0 * 1536) * 6;
        printf("ABL=%d EPI %d M=%d: %.3f ms per launch, %.0f TFLOP/s issued\n", ML_ABL, epi, M, ms / reps, flops / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}
