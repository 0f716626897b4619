// Two dense layers of a PaiNN node MLP in ONE kernel, f16x3 arithmetic (gemm16.hip), hidden width H = 512:
//
//   y   = ScaledSiLU(A W0^T + b0)          A [M, K0] fp32 (K0 = 512, or 1024 from two sources [A1 | A2]), W0 [512, K0]
//   out = y W2^T + b2                      W2 [1536, 512] row-permuted as in gemm16.hip (three H-wide parts of 32 channels
//                                          side by side), consumed on the accumulators by the same two epilogues:
//     EPI 1  x_proj:    LayerNorm(x) -> x_proj.0 -> x_proj.2 -> gather records of the message kernel
//                       (painn_denoising.py:531; replaces adf_gemm_f16x3<ssilu> + adf_gemm_f16x3<EPI 1>)
//     EPI 2  xvec_proj: [x | ||v2||] -> xvec_proj.0 -> xvec_proj.2 -> gating + residuals + ScaleFactor
//                       (painn_denoising.py:614-623, 449-451; replaces adf_gemm_f16x3<ssilu> + adf_gemm_f16x3<EPI 2>)
//
// Why: between the two products of such a pair the [M, 512] intermediate went to HBM and back (0.8 GB per pair and layer
// at 200 000 atoms, plus its row magnitudes), and each product ran a 2-barriers-per-K-step LDS pipeline at 30-40 % matrix-
// core utilisation.  Here a 512-thread workgroup owns 64 rows: the lifted fp16 hi / lo image of its A rows, and then of
// its y rows, lives in 130 KB of LDS (row stride 1040 B: conflict-free ds_read_b128 fragments); the weights are never
// staged - every wave owns its own output columns, so a weight element is used by exactly one wave and its MFMA B
// fragments are loaded STRAIGHT FROM L2 INTO REGISTERS from a fragment-ordered image (one contiguous KB per wave
// instruction, packed once at adf_painn_set_weights).  No barrier inside a K loop; 8 waves = 2 per SIMD, so that one
// wave's epilogue (global loads / stores) runs beside its partner's matrix phase.
//
// Arithmetic: bit-identical to the two-kernel path (same per-row lifts from the same row magnitudes, same order of the
// three split products per 16-deep k-step, same epilogue expressions; tests/test_gpu_parity.py), so choosing it
// (adf_painn_set_fused_mlp) does not touch the reproducibility guarantees (sharded / incremental / subset runs).
//
// STATUS (round 6): measured 5-12 % SLOWER than the two gemm16.hip products it replaces (2.06 / 2.09 ms per launch against
// 1.83 / 2.04 at 1000 systems) and therefore OFF by default.  130 KB of LDS = one workgroup per CU, and a tile's phases that are
// not matrix work (the load of its rows, the ScaledSiLU / row-maximum / split hand-off, the 1 MB of epilogue traffic at
// 12 KB in flight per wave) have nothing to overlap with; profiles/NOTES.md has the phase times.  Its weight-fragment image
// and its K loop (B fragments straight from L2 into registers, pinned prefetch) are what gemm16.hip's products now run on.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#ifndef ML_ABL
#define ML_ABL 0   // timing experiments only (wrong results): 1 = epilogues without global loads / stores, 2 = K loops without
                   // weight loads, 4 = every weight load from k-step 0 (cache-hot), 8 = no second product, 16 = no first product
#endif
#ifndef ML_VAR
#define ML_VAR 0   // schedule variants under measurement: 1 = weight requests interleaved with the products (sched_group_barrier),
                   // 2 = waves 4-7 at raised priority in the second product (their partners on the same SIMDs fall behind)
#endif
#ifndef ML_PROF
#define ML_PROF 0  // 1: waves 0 and 4 of every workgroup add their phase times (100 MHz ticks) into ep.rec[0..15] as uint64 (harness only)
#endif
#if ML_PROF
#define ML_STAMP(k_) do { if (lane == 0 && (wave & 3) == 0) { const unsigned long long t_ = wall_clock64(); \
        atomicAdd(reinterpret_cast<unsigned long long*>(ml_prof) + (wave >> 2) * 8 + (k_), t_ - ml_t); ml_t = t_; } } while (0)
#else
#define ML_STAMP(k_) do { } while (0)
#endif
#define ML_H 512
#define ML_TM 64                       // rows per workgroup
#define ML_YLD 1040                    // bytes per row of an LDS plane: 512 halves + 16 B (16 rows of a lane group -> 16 bank quads)
#define ML_PLANE (ML_TM * ML_YLD)      // 66 560 B
#define ML_TFLOATS 768                 // per-wave transposition scratch: [8 rows][96] floats
#define ML_LDS_BYTES (2 * ML_PLANE + 8 * ML_TFLOATS * 4 + 4 * ML_TM * 4)   // 158 720 B of the CU's 163 840

__device__ __forceinline__ float ml_pow2_lift(float mx) {   // == adf_pow2_lift (gemm16.hip)
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(mx, &e);
    e = 15 - e;
    e = e > 120 ? 120 : (e < -120 ? -120 : e);
    return ldexpf(1.0f, e);
}

__device__ __forceinline__ float ml_ssilu(float x) {        // == ssilu16 (gemm16.hip)
    float s = x / (1.0f + expf(-x));
    return s * 1.6666666666666667f;
}

__device__ __forceinline__ float ml_row16_max(float v) {    // == adf_row16_max (gemm16.hip)
#define ML_ROR(n_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n_), 0xf, 0xf, false))
    v = fmaxf(v, ML_ROR(8));
    v = fmaxf(v, ML_ROR(4));
    v = fmaxf(v, ML_ROR(2));
    v = fmaxf(v, ML_ROR(1));
#undef ML_ROR
    return v;
}

// Fragment-ordered weight image: for 32-column block cb, k-step s (16 k) and plane (0 = hi, 1 = lo) the 64 lanes' B operands
// of v_mfma_f32_32x32x16_f16 lie in one contiguous KB: lane l holds W[32 cb + (l & 31)][16 s + 8 (l >> 5) .. + 7].
__global__ void adf_pack_frag_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, half8* __restrict__ out,
                                     int N, int K) {
    const int nks = K / 16;
    const long long total = (long long)(N / 32) * nks * 2 * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), plane = (int)((i >> 6) & 1);
        const long long cs = i >> 7;
        const int s = (int)(cs % nks), cb = (int)(cs / nks);
        const _Float16* src = (plane ? lo : hi) + (size_t)(32 * cb + (lane & 31)) * K + 16 * s + 8 * (lane >> 5);
        out[i] = *reinterpret_cast<const half8*>(src);
    }
}

int32_t adf_pack_frag(const adf_w16* w, int N, int K, void* out, hipStream_t s) {
    if (N % 32 || K % 16) { adf_set_error("pack_frag: N %% 32 or K %% 16"); return ADF_EINVAL; }
    hipLaunchKernelGGL(adf_pack_frag_kernel, dim3(256), dim3(256), 0, s, (const _Float16*)w->hi, (const _Float16*)w->lo,
                       (half8*)out, N, K);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

template <int EPI, bool TWO>
__global__ __launch_bounds__(512, 1) void adf_mlp16_kernel(const float* __restrict__ A1, const float* __restrict__ A2, int lda,
                                                           const float* __restrict__ rmag, const half8* __restrict__ W0f,
                                                           const float* __restrict__ isc0p, const float* __restrict__ bias0,
                                                           const half8* __restrict__ W2f, const float* __restrict__ isc2p,
                                                           const float* __restrict__ bias2, int Mh, adf_epi ep) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[ML_LDS_BYTES];
    const int M = ep.m_dev ? min(Mh, (int)*ep.m_dev) : Mh;
    const int m0 = blockIdx.x * ML_TM;
    if (m0 >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((ML_VAR & 12) && blockIdx.x < 256) {   // experiment: the chip's first round of workgroups starts out of phase
        const int ph = (ML_VAR & 8) ? (int)((blockIdx.x >> 3) & 7) : (int)((blockIdx.x >> 3) & 3);
        for (int d = 0; d < ph * ((ML_VAR & 8) ? 3 : 5); ++d) __builtin_amdgcn_s_sleep(127);
    }
#if ML_PROF
    unsigned long long ml_t = wall_clock64();
    float* const ml_prof = ep.cat;   // (unused by these epilogues: the harness hangs its counter buffer there)
#endif
    unsigned char* const Yhi = lds;
    unsigned char* const Ylo = lds + ML_PLANE;
    float* const T = reinterpret_cast<float*>(lds + 2 * ML_PLANE) + wave * ML_TFLOATS;
    float* const liftA = reinterpret_cast<float*>(lds + 2 * ML_PLANE + 8 * ML_TFLOATS * 4);
    float* const rinvA = liftA + ML_TM;
    float* const rinvY = rinvA + ML_TM;
    unsigned int* const ymax = reinterpret_cast<unsigned int*>(rinvY + ML_TM);
    constexpr int H = ML_H;
    constexpr int NKS = (TWO ? 2 * H : H) / 16;   // k-steps of the first product

    // ---- the workgroup's A rows -> lifted fp16 hi / lo planes.  16 float4 per thread: f = tid + 512 i -> row f >> 7
    float4 av[16];
    auto request_rows = [&](const float* __restrict__ src) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = tid + 512 * i;
            const int row = f >> 7, c4 = f & 127;
            av[i] = *reinterpret_cast<const float4*>(src + (size_t)min(m0 + row, M - 1) * lda + 4 * c4);
        }
    };
    auto stage_rows = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = tid + 512 * i;
            const int row = f >> 7, c4 = f & 127;
            const float rs = liftA[row];
            const float sx = av[i].x * rs, sy = av[i].y * rs, sz = av[i].z * rs, sw = av[i].w * rs;
            half4 h, l;
            h[0] = (_Float16)sx; h[1] = (_Float16)sy; h[2] = (_Float16)sz; h[3] = (_Float16)sw;
            l[0] = (_Float16)(sx - (float)h[0]); l[1] = (_Float16)(sy - (float)h[1]);
            l[2] = (_Float16)(sz - (float)h[2]); l[3] = (_Float16)(sw - (float)h[3]);
            *reinterpret_cast<half4*>(Yhi + row * ML_YLD + 8 * c4) = h;
            *reinterpret_cast<half4*>(Ylo + row * ML_YLD + 8 * c4) = l;
        }
    };
    request_rows(A1);
    if (tid < ML_TM) {
        const int g = min(m0 + tid, M - 1);
        const float lf = rmag ? ml_pow2_lift(rmag[g]) : 1.0f;
        liftA[tid] = lf;
        rinvA[tid] = rmag ? 1.0f / ml_pow2_lift(rmag[g]) : 1.0f;
        ymax[tid] = 0u;
    }
    __syncthreads();
    stage_rows();
    if (TWO) request_rows(A2);   // the second K half: in flight behind the first half's products
    __syncthreads();
    ML_STAMP(0);

    // A fragment of row block i, k-step s: lane -> row 32 i + (lane & 31), halves 16 s + 8 (lane >> 5) .. + 7
    const int a_off = (lane & 31) * ML_YLD + (lane >> 5) * 16;

    // ---- first product: wave w owns columns [64 w, 64 w + 64): 2 column blocks x 2 row blocks
    f32x16 acc0[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[i][j][r] = 0.f;
    {
        const half8* const w0 = W0f + lane;
        auto ldB = [&](int s_, half8 (&b)[4]) {
            if ((ML_ABL & 2) && s_ != 0) return;
            const int s = (ML_ABL & 4) ? 0 : s_;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const size_t base = ((size_t)(2 * wave + j) * NKS + s) * 128;
                b[2 * j] = w0[base];
                b[2 * j + 1] = w0[base + 64];
            }
        };
        auto step = [&](int sl, const half8 (&b)[4]) {   // sl: k-step inside the LDS-resident half
            half8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(Yhi + a_off + i * 32 * ML_YLD + sl * 32);
                al[i] = *reinterpret_cast<const half8*>(Ylo + a_off + i * 32 * ML_YLD + sl * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], b[2 * j], acc0[i][j], 0, 0, 0);
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b[2 * j + 1], acc0[i][j], 0, 0, 0);
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b[2 * j], acc0[i][j], 0, 0, 0);
                }
        };
        half8 b0[4], b1[4];
        if (ML_ABL & 2) { ldB(0, b0); ldB(0, b1); }
#pragma unroll 1
        for (int half = 0; half < ((ML_ABL & 16) ? 0 : (TWO ? 2 : 1)); ++half) {
            const int ks0 = half * (H / 16);
            if (half == 1) {
                __syncthreads();   // every wave is done with the first half's fragments
                stage_rows();
                __syncthreads();
            }
            // The requests for k-step s + 1 must stay IN FRONT of the products of k-step s: left alone, hipcc's scheduler sinks
            // them to just before their first use (shorter live ranges) and every k-step then waits out an L2 round trip
            // (first build of this kernel: 2.1 ms per launch instead of 0.9).  sched_barrier(0) pins the order of the groups.
            ldB(ks0, b0);
#pragma unroll 1
            for (int s = 0; s < H / 16; s += 2) {
                ldB(ks0 + s + 1, b1);
                if (!(ML_VAR & 1)) __builtin_amdgcn_sched_barrier(0);
                step(s, b0);
                if (ML_VAR & 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 3, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
                ldB(ks0 + min(s + 2, H / 16 - 1), b0);   // (the last request re-reads a fragment: no branch around a load)
                if (!(ML_VAR & 1)) __builtin_amdgcn_sched_barrier(0);
                step(s + 1, b1);
                if (ML_VAR & 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 3, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    ML_STAMP(1);
    // ---- y = ssilu(acc / lifts + b0) in registers; its row maxima through LDS; then its lifted hi / lo planes over A's
    {
        const float isc0 = *isc0p;
        const int q = lane & 31;
        float bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = bias0[64 * wave + 32 * j + q];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float sc = isc0 * rinvA[row];
                float v0 = acc0[i][0][r] * sc + bv[0], v1 = acc0[i][1][r] * sc + bv[1];
                v0 = ml_ssilu(v0); v1 = ml_ssilu(v1);
                acc0[i][0][r] = v0; acc0[i][1][r] = v1;
                float mg = fmaxf(fabsf(v0), fabsf(v1));
                mg = ml_row16_max(mg);
                mg = fmaxf(mg, __shfl_xor(mg, 16));
                if (q == 0) atomicMax(ymax + row, __float_as_uint(mg));
            }
        __syncthreads();   // all maxima in; every wave is done reading A's planes
        if (tid < ML_TM) {
            const float mx = __uint_as_float(ymax[tid]);
            liftA[tid] = ep.lift_y ? ml_pow2_lift(mx) : 1.0f;
            rinvY[tid] = ep.lift_y ? 1.0f / ml_pow2_lift(mx) : 1.0f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float rs = liftA[row];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float sx = acc0[i][j][r] * rs;
                    const _Float16 hh = (_Float16)sx;
                    const _Float16 ll = (_Float16)(sx - (float)hh);
                    const int col = 64 * wave + 32 * j + q;
                    *reinterpret_cast<_Float16*>(Yhi + row * ML_YLD + 2 * col) = hh;
                    *reinterpret_cast<_Float16*>(Ylo + row * ML_YLD + 2 * col) = ll;
                }
            }
        __syncthreads();
    }

    ML_STAMP(2);
    // ---- second product: wave w owns the channel groups {w, w + 8}: 3 column blocks (the three parts) x 2 row blocks
    const float isc2 = *isc2p;
    const half8* const w2 = W2f + lane;
    if ((ML_VAR & 2) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int gi = 0; gi < ((ML_ABL & 8) ? 0 : 2); ++gi) {
        const int g = wave + 8 * gi;
        f32x16 acc[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;
        auto ldB = [&](int s_, half8 (&b)[6]) {
            if ((ML_ABL & 2) && s_ != 0) return;
            const int s = (ML_ABL & 4) ? 0 : s_;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const size_t base = ((size_t)(3 * g + p) * (H / 16) + s) * 128;
                b[2 * p] = w2[base];
                b[2 * p + 1] = w2[base + 64];
            }
        };
        auto step = [&](int s, const half8 (&b)[6]) {
            half8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(Yhi + a_off + i * 32 * ML_YLD + s * 32);
                al[i] = *reinterpret_cast<const half8*>(Ylo + a_off + i * 32 * ML_YLD + s * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], b[2 * p], acc[i][p], 0, 0, 0);
                    acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b[2 * p + 1], acc[i][p], 0, 0, 0);
                    acc[i][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], b[2 * p], acc[i][p], 0, 0, 0);
                }
        };
        {
            half8 b0[6], b1[6];
            if (ML_ABL & 2) ldB(0, b1);
            ldB(0, b0);
#pragma unroll 1
            for (int s = 0; s < H / 16; s += 2) {
                ldB(s + 1, b1);
                if (!(ML_VAR & 1)) __builtin_amdgcn_sched_barrier(0);
                step(s, b0);
                if (ML_VAR & 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int u = 0; u < 6; ++u) { __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 3, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
                ldB(min(s + 2, H / 16 - 1), b0);
                if (!(ML_VAR & 1)) __builtin_amdgcn_sched_barrier(0);
                step(s + 1, b1);
                if (ML_VAR & 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int u = 0; u < 6; ++u) { __builtin_amdgcn_sched_group_barrier(0x20, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 3, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        ML_STAMP(3 + 2 * gi);
        // epilogue on 8-row bands through the wave's transposition scratch: lane = (row lr, 4 channels c4)
        // (an opaque copy of the lane index: everything the epilogue derives from it is then computed HERE, not hoisted in
        // front of the K loop where 8 bands x several 64-bit addresses would sit beside the accumulators and the ring)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int q = lane_e & 31;
        const float bb0 = bias2[96 * g + q], bb1 = bias2[96 * g + 32 + q], bb2 = bias2[96 * g + 64 + q];
        const int lr = lane_e >> 3, c4 = lane_e & 7;
        const int c = 32 * g + 4 * c4;
        // uniform bases + 32-bit byte offsets (the launcher checks the spans)
        const char* const vin_b = reinterpret_cast<const char*>(ep.vec_in);
        const char* const dot_b = reinterpret_cast<const char*>(ep.dot);
        char* const x_b = reinterpret_cast<char*>(ep.x);
        char* const vec_b = reinterpret_cast<char*>(ep.vec);
        const char* const vv_b = reinterpret_cast<const char*>(ep.vv);
        char* const rec_b = reinterpret_cast<char*>(ep.rec);
        // The global operands of a whole batch of bands are requested before the batch's first store (a store may alias a
        // later load for all the compiler knows: band by band the epilogue is 8 dependent HBM round trips, 20 us of a 130 us
        // tile).  EPI 1: 4 bands per batch (3 float4 each); EPI 2: 2 bands (8 float4 each).
        constexpr int NB = EPI == 1 ? 4 : 2;        // bands per batch (more: the operands spill beside the 96 accumulators)
        constexpr int NE = EPI == 1 ? 3 : 8;        // float4 operands per band
#pragma unroll
        for (int bt = 0; bt < 8 / NB; ++bt) {
            float4 eo[NB][NE];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const int band = bt * NB + bb;
                const int n = min(m0 + 8 * band + lr, M - 1);
                if (ML_ABL & 1) {
#pragma unroll
                    for (int k = 0; k < NE; ++k) eo[bb][k] = make_float4(1.f, 2.f, 3.f, 4.f);
                } else if constexpr (EPI == 1) {
                    if (!ep.vec_is_zero) {
                        const unsigned int vo = ((unsigned int)n * (3u * H) + (unsigned int)c) * 4u;
                        eo[bb][0] = *reinterpret_cast<const float4*>(vin_b + vo);
                        eo[bb][1] = *reinterpret_cast<const float4*>(vin_b + vo + 4u * H);
                        eo[bb][2] = *reinterpret_cast<const float4*>(vin_b + vo + 8u * H);
                    }
                } else {
                    const unsigned int xo = ((unsigned int)n * (unsigned int)H + (unsigned int)c) * 4u;
                    const unsigned int vo = ((unsigned int)n * (3u * H) + (unsigned int)c) * 4u;
                    eo[bb][3] = *reinterpret_cast<const float4*>(dot_b + xo);
                    eo[bb][4] = *reinterpret_cast<const float4*>(x_b + xo);
                    eo[bb][5] = *reinterpret_cast<const float4*>(vv_b + vo);
                    eo[bb][6] = *reinterpret_cast<const float4*>(vv_b + vo + 4u * H);
                    eo[bb][7] = *reinterpret_cast<const float4*>(vv_b + vo + 8u * H);
                    eo[bb][0] = *reinterpret_cast<const float4*>(vec_b + vo);
                    eo[bb][1] = *reinterpret_cast<const float4*>(vec_b + vo + 4u * H);
                    eo[bb][2] = *reinterpret_cast<const float4*>(vec_b + vo + 8u * H);
                }
            }
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const int band = bt * NB + bb;
                const int i = band >> 2, qd = band & 3;
                const int n_raw = m0 + 8 * band + lr;
                const int n = min(n_raw, M - 1);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * qd + rr;
                    const int l8 = rr + 4 * (lane >> 5);
                    const float sc = isc2 * rinvY[8 * band + l8];
                    T[l8 * 96 + q] = acc[i][0][r] * sc + bb0;
                    T[l8 * 96 + 32 + q] = acc[i][1][r] * sc + bb1;
                    T[l8 * 96 + 64 + q] = acc[i][2][r] * sc + bb2;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float4 p0 = *reinterpret_cast<const float4*>(T + lr * 96 + 4 * c4);
                const float4 p1 = *reinterpret_cast<const float4*>(T + lr * 96 + 32 + 4 * c4);
                const float4 p2 = *reinterpret_cast<const float4*>(T + lr * 96 + 64 + 4 * c4);
                const float4 e0 = eo[bb][0], e1 = eo[bb][1], e2 = eo[bb][2];
                if (ML_ABL & 1) {
                    const float sk = p0.x + p1.y + p2.z + e0.x + e1.y + e2.z;
                    asm volatile("" ::"v"(sk));
                } else if constexpr (EPI == 1) {
                    if (n_raw < M) {
                        const unsigned int ro = ((unsigned int)(ep.row_map ? ep.row_map[n] : n) * (unsigned int)(H / 32) + (unsigned int)g) * 640u;
                        float* rec = reinterpret_cast<float*>(rec_b + ro);
                        float4* ra_ = reinterpret_cast<float4*>(rec + 16 * c4);
                        if (!ep.vec_is_zero) {
                            ra_[0] = make_float4(e0.x * p1.x, e1.x * p1.x, e2.x * p1.x, p0.x);
                            ra_[1] = make_float4(e0.y * p1.y, e1.y * p1.y, e2.y * p1.y, p0.y);
                            ra_[2] = make_float4(e0.z * p1.z, e1.z * p1.z, e2.z * p1.z, p0.z);
                            ra_[3] = make_float4(e0.w * p1.w, e1.w * p1.w, e2.w * p1.w, p0.w);
                        } else {
                            ra_[0] = make_float4(0.f, 0.f, 0.f, p0.x);
                            ra_[1] = make_float4(0.f, 0.f, 0.f, p0.y);
                            ra_[2] = make_float4(0.f, 0.f, 0.f, p0.z);
                            ra_[3] = make_float4(0.f, 0.f, 0.f, p0.w);
                        }
                        *reinterpret_cast<float4*>(rec + 128 + 4 * c4) = p2;
                    }
                } else {
                    const float4 d4 = eo[bb][EPI == 1 ? 0 : 3], x4 = eo[bb][EPI == 1 ? 0 : 4];
                    const float4 w10 = eo[bb][EPI == 1 ? 0 : 5], w11 = eo[bb][EPI == 1 ? 0 : 6], w12 = eo[bb][EPI == 1 ? 0 : 7];
                    const float k2 = 0.70710678118654752f, sc = ep.scale;
                    float4 xo4 = x4;
                    xo4.x = (xo4.x + (p0.x + p1.x * d4.x) * k2) * sc;
                    xo4.y = (xo4.y + (p0.y + p1.y * d4.y) * k2) * sc;
                    xo4.z = (xo4.z + (p0.z + p1.z * d4.z) * k2) * sc;
                    xo4.w = (xo4.w + (p0.w + p1.w * d4.w) * k2) * sc;
                    if (n_raw < M) {
                        *reinterpret_cast<float4*>(x_b + ((unsigned int)n * (unsigned int)H + (unsigned int)c) * 4u) = xo4;
                        float* vr = reinterpret_cast<float*>(vec_b + ((unsigned int)n * (3u * H) + (unsigned int)c) * 4u);
                        float4 t = e0;
                        t.x += p2.x * w10.x; t.y += p2.y * w10.y; t.z += p2.z * w10.z; t.w += p2.w * w10.w;
                        *reinterpret_cast<float4*>(vr) = t;
                        t = e1;
                        t.x += p2.x * w11.x; t.y += p2.y * w11.y; t.z += p2.z * w11.z; t.w += p2.w * w11.w;
                        *reinterpret_cast<float4*>(vr + H) = t;
                        t = e2;
                        t.x += p2.x * w12.x; t.y += p2.y * w12.y; t.z += p2.z * w12.z; t.w += p2.w * w12.w;
                        *reinterpret_cast<float4*>(vr + 2 * H) = t;
                    }
                }
                __builtin_amdgcn_wave_barrier();   // T is rewritten by the next band
            }
        }
        ML_STAMP(4 + 2 * gi);
    }
}

// y = ssilu(A W0^T + b0) ; (y W2^T + b2) into epilogue `epi` (1: gather records, 2: update gating), see the kernel comment.
// W0f / W2f: fragment images (adf_pack_frag) of the split weights W0 / W2; A2 != null: K0 = 1024 from [A1 | A2].
int32_t adf_launch_mlp16(const float* A1, const float* A2, int lda, const float* rmag, const void* W0f, const adf_w16* W0,
                         const float* bias0, const void* W2f, const adf_w16* W2, int M, int H, int epi, const adf_epi* ep_in,
                         hipStream_t s) {
    if (M <= 0) return ADF_OK;
    if (H != ML_H || (lda & 3) || (epi != 1 && epi != 2) || !W2->bias_perm) {
        adf_set_error("mlp16: needs hidden width %d, lda %% 4 == 0 and a fused 3H-wide second layer", ML_H);
        return ADF_EINVAL;
    }
    // the epilogues address vec / v1 [M,3,H] and the record table [rows+1, 5H] with 32-bit byte offsets
    if ((long long)M * 3 * H * 4 >= (1ll << 32) || (epi == 1 && ((long long)(ep_in->rec_rows > 0 ? ep_in->rec_rows : M) + 1) * 5 * H * 4 >= (1ll << 32))) {
        adf_set_error("mlp16: %d rows exceed the 32-bit offset range of the fused epilogue", M);
        return ADF_EOOM;
    }
    if ((long long)M * lda * 4 >= (1ll << 32)) {
        adf_set_error("mlp16: A operand of %d rows x %d exceeds the 32-bit offset range, split the batch", M, lda);
        return ADF_EOOM;
    }
    const dim3 grid((unsigned)((M + ML_TM - 1) / ML_TM));
#define ML_LAUNCH(EPI_, TWO_)                                                                                              \
    hipLaunchKernelGGL((adf_mlp16_kernel<EPI_, TWO_>), grid, dim3(512), 0, s, A1, A2, lda, rmag, (const half8*)W0f,      \
                       W0->inv_scale, bias0, (const half8*)W2f, W2->inv_scale, W2->bias_perm, M, *ep_in)
    if (epi == 1) { if (A2) ML_LAUNCH(1, true); else ML_LAUNCH(1, false); }
    else { if (A2) ML_LAUNCH(2, true); else ML_LAUNCH(2, false); }
#undef ML_LAUNCH
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
