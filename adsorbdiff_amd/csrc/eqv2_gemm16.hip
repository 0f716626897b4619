// Dense products of the EquiformerV2 path on the f16 matrix cores with the 3-product split of gemm16.hip
// (a = a_hi + a_lo, w s = w_hi + w_lo, three v_mfma_f32_32x32x16_f16 per fp32 product, fp32 accumulate), plus what the
// PaiNN kernel does not have:
//   * a power-of-two scale PER ROW of A applied before the split (and divided out in the epilogue).  The matrix core
//     flushes fp16 subnormals, so an unscaled element below ~0.1 loses its a_lo term (2^-11 instead of 2^-22 relative);
//     EquiformerV2's operands are products of O(1e-3) embeddings and would sit there.  With the row's largest element
//     lifted to [2^14, 2^15) every element down to 8e-6 of the row maximum keeps both terms, and smaller ones contribute
//     less than 4e-9 of the row's scale.  The scale is a function of the row alone, so results do not depend on which
//     rows share a launch (sharded / chunked runs reproduce bit for bit).
//   * two-level row addressing of A and C (eq_rowmap): the per-degree maps of an SO3_LinearV2 work on [N, S, C] in place;
//   * accumulate-into-C (residual connections) and plain SiLU.
// Tiling as gemm16.hip: 128 x (64 NJ) x 32 per 256-thread workgroup, 4 waves as 2 x 2, fp32 A split while staged, weights
// pre-split at set_weights, one K-tile of global loads in flight, XCD-aware tile order, 16-byte stores through LDS.
#include <stdlib.h>

#include "eqv2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define GK 32
#define GLD 40  // halves per LDS row (32 + 8 pad: conflict-free ds_read_b128)
#define GTLD 64

__device__ __forceinline__ float eq16_silu(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ float eq16_lift(float mx) {
    // 2^e with mx 2^e in [2^14, 2^15); 1 for mx == 0 or non-finite
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(mx, &e);
    e = 15 - e;
    e = e > 120 ? 120 : (e < -120 ? -120 : e);
    return ldexpf(1.0f, e);
}

// rs[r] = max_k |A[r, k]| (the product kernels turn it into the row's power-of-two lift).  Sixteen lanes per row (four rows per
// wave, sixteen per workgroup), the row's maximum by four DPP row rotations (round 6; was one wave per row with a six-step
// ds_bpermute butterfly: four times the workgroups and a chain of LDS-crossbar round trips per row).
__global__ __launch_bounds__(256) void eq_rowscale_kernel(const float* __restrict__ A, eq_rowmap am, long long M, int K,
                                                          float* __restrict__ rs) {
    const int lane = threadIdx.x & 63, j = lane & 15;
    const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + (lane >> 4);
    const long long rc = r < M ? r : M - 1;   // (whole waves stay in the rotations; rows past M are not stored)
    const float* a;
    if (am.period == 1) {
        a = A + rc * am.outer;
    } else {   // (32-bit division: the launcher bounds M)
        const unsigned int q = (unsigned int)rc / (unsigned int)am.period, rem = (unsigned int)rc - q * (unsigned int)am.period;
        a = A + (long long)q * am.outer + (long long)rem * am.inner;
    }
    float mx = 0.f;
    for (int k = j * 4; k < K; k += 64) {
        const float4 v = *reinterpret_cast<const float4*>(a + k);
        mx = fmaxf(fmaxf(fmaxf(mx, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#define EQ_ROR(n_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x120 + (n_), 0xf, 0xf, false))
    mx = fmaxf(mx, EQ_ROR(8));
    mx = fmaxf(mx, EQ_ROR(4));
    mx = fmaxf(mx, EQ_ROR(2));
    mx = fmaxf(mx, EQ_ROR(1));
#undef EQ_ROR
    if (j == 0 && r < M) rs[r] = mx;
}

int32_t eq_launch_rowscale(const float* A, const eq_rowmap* am, long long M, int K, float* rs, hipStream_t s) {
    if (M <= 0) return ADF_OK;
    if (M >= (1ll << 32)) { adf_set_error("eq_rowscale: %lld rows", M); return ADF_EINVAL; }
    hipLaunchKernelGGL(eq_rowscale_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, s, A, *am, M, K, rs);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

template <int ACT, int NJ, bool ACCUM>
__global__ __launch_bounds__(256, 2) void eq_gemm16_kernel(const float* __restrict__ A, eq_rowmap am,
                                                           const float* __restrict__ rscale,
                                                           const _Float16* __restrict__ Whi,
                                                           const _Float16* __restrict__ Wlo,
                                                           const float* __restrict__ inv_scale,
                                                           const float* __restrict__ bias, float* __restrict__ Cm,
                                                           eq_rowmap cm, long long M, int N, int K, int tiles_n,
                                                           unsigned int* __restrict__ out_mag, int rs_div) {
    constexpr int MI = 2;
    constexpr int TM = 64 * MI, TN = 64 * NJ, NA = TM / 32;
    __shared__ __attribute__((aligned(16))) _Float16 lds[(2 * TM + 2 * TN) * GLD];
    __shared__ float rinv[TM];
    _Float16* Ahi = lds;
    _Float16* Alo = Ahi + TM * GLD;
    _Float16* Bhi = Alo + TM * GLD;
    _Float16* Blo = Bhi + TN * GLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * (32 * NJ);
    const int id = blockIdx.x, xcd = id & 7, qd = id >> 3;
    const long long tile_m = (long long)(qd / tiles_n) * 8 + xcd;
    const int tile_n = qd % tiles_n;
    const long long m0 = tile_m * TM;
    const int n0 = tile_n * TN;
    if (m0 >= M) return;

    const float* a_ptr[NA];
    float a_rs[NA];
    int a_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i;
        const int row = f >> 3, kq = f & 7;
        long long grow = m0 + row;
        if (grow > M - 1) grow = M - 1;
        a_ptr[i] = A + (grow / am.period) * am.outer + (grow % am.period) * (long long)am.inner + kq * 4;
        a_rs[i] = rscale ? eq16_lift(rscale[rs_div > 1 ? grow / rs_div : grow]) : 1.0f;
        a_off[i] = row * GLD + kq * 4;
    }
    if (tid < TM) {
        long long grow = m0 + tid;
        if (grow > M - 1) grow = M - 1;
        rinv[tid] = rscale ? 1.0f / eq16_lift(rscale[rs_div > 1 ? grow / rs_div : grow]) : 1.0f;
    }
    int w_src[NJ], w_off[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int f = tid + 256 * i;
        const int row = f >> 2, part = f & 3;
        w_src[i] = min(n0 + row, N - 1) * K + part * 8;
        w_off[i] = row * GLD + part * 8;
    }
    float4 ra[NA];
    half8 rwh[NJ], rwl[NJ];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4*>(a_ptr[i]);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        rwh[i] = *reinterpret_cast<const half8*>(Whi + w_src[i]);
        rwl[i] = *reinterpret_cast<const half8*>(Wlo + w_src[i]);
    }
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / GK;
    const int fa = (wm + (lane & 31)) * GLD + (lane >> 5) * 8;
    const int fb = (wn + (lane & 31)) * GLD + (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float sx = ra[i].x * a_rs[i], sy = ra[i].y * a_rs[i], sz = ra[i].z * a_rs[i], sw = ra[i].w * a_rs[i];
            half4 h, l;
            h[0] = (_Float16)sx; h[1] = (_Float16)sy; h[2] = (_Float16)sz; h[3] = (_Float16)sw;
            l[0] = (_Float16)(sx - (float)h[0]); l[1] = (_Float16)(sy - (float)h[1]);
            l[2] = (_Float16)(sz - (float)h[2]); l[3] = (_Float16)(sw - (float)h[3]);
            *reinterpret_cast<half4*>(Ahi + a_off[i]) = h;
            *reinterpret_cast<half4*>(Alo + a_off[i]) = l;
        }
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            *reinterpret_cast<half8*>(Bhi + w_off[i]) = rwh[i];
            *reinterpret_cast<half8*>(Blo + w_off[i]) = rwl[i];
        }
        __syncthreads();
        if (kt + 1 < nk) {
            const int k1 = (kt + 1) * GK;
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + k1);
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                rwh[i] = *reinterpret_cast<const half8*>(Whi + w_src[i] + k1);
                rwl[i] = *reinterpret_cast<const half8*>(Wlo + w_src[i] + k1);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 bh[NJ], bl[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                bh[j] = *reinterpret_cast<const half8*>(Bhi + fb + j * 32 * GLD + ks * 16);
                bl[j] = *reinterpret_cast<const half8*>(Blo + fb + j * 32 * GLD + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const half8 ah = *reinterpret_cast<const half8*>(Ahi + fa + i * 32 * GLD + ks * 16);
                const half8 al = *reinterpret_cast<const half8*>(Alo + fa + i * 32 * GLD + ks * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }

    const float isc = *inv_scale;
    const int q = lane & 31;
    __syncthreads();  // all waves are done reading the operand tiles
    float* T = reinterpret_cast<float*>(lds) + wave * (32 * GTLD);  // [32 rows][64] floats per wave
#pragma unroll
    for (int jj = 0; jj < NJ / 2; ++jj) {
        const int cb = n0 + wn + 64 * jj;
        const float bv0 = (bias && cb + q < N) ? bias[cb + q] : 0.f;
        const float bv1 = (bias && cb + 32 + q < N) ? bias[cb + 32 + q] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float sc = isc * rinv[wm + 32 * i + lr];
                float v0 = acc[i][2 * jj][r] * sc + bv0, v1 = acc[i][2 * jj + 1][r] * sc + bv1;
                if (ACT == 2) { v0 = eq16_silu(v0); v1 = eq16_silu(v1); }
                T[lr * GTLD + q] = v0;
                T[lr * GTLD + 32 + q] = v1;
                if (out_mag) {  // magnitude of the output row (all lanes of a half-wave hold the same row)
                    float mg = fmaxf(cb + q < N ? fabsf(v0) : 0.f, cb + 32 + q < N ? fabsf(v1) : 0.f);
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) mg = fmaxf(mg, __shfl_xor(mg, o));
                    const long long orow = m0 + wm + 32 * i + lr;
                    if (q == 0 && orow < M) atomicMax(out_mag + orow, __float_as_uint(mg));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int item = lane + 64 * it;
                const int lr = item >> 4, c4 = item & 15;
                const long long row = m0 + wm + 32 * i + lr;
                const int col = cb + 4 * c4;
                if (row < M && col < N) {
                    float* cp = Cm + (row / cm.period) * cm.outer + (row % cm.period) * (long long)cm.inner + col;
                    float4 v = *reinterpret_cast<const float4*>(T + lr * GTLD + 4 * c4);
                    if (ACCUM) {
                        const float4 o = *reinterpret_cast<const float4*>(cp);
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *reinterpret_cast<float4*>(cp) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// 256 x 256 x 32 tile, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 (4 x 2 MFMA blocks): per k-tile the workgroup
// stages 32 KB of A and 32 KB of W for 384 MFMAs — a third of the operand bytes per MFMA of the 128 x 256 kernel above
// (whose two co-resident workgroups ask the L2 for ~60 B/clk/CU at the matrix pipe's pace).  One workgroup per CU,
// two waves per SIMD.
template <int ACT, bool ACCUM>
__global__ __launch_bounds__(512, 2) void eq_gemm16_256_kernel(const float* __restrict__ A, eq_rowmap am,
                                                               const float* __restrict__ rscale,
                                                               const _Float16* __restrict__ Whi,
                                                               const _Float16* __restrict__ Wlo,
                                                               const float* __restrict__ inv_scale,
                                                               const float* __restrict__ bias, float* __restrict__ Cm,
                                                               eq_rowmap cm, long long M, int N, int K, int tiles_n,
                                                               unsigned int* __restrict__ out_mag, int rs_div) {
    constexpr int MI = 4, NJ = 2, TM = 256, TN = 256, NA = 4, NW = 2;
    __shared__ __attribute__((aligned(16))) _Float16 lds[(2 * TM + 2 * TN) * GLD];
    __shared__ float rinv[TM];
    _Float16* Ahi = lds;
    _Float16* Alo = Ahi + TM * GLD;
    _Float16* Bhi = Alo + TM * GLD;
    _Float16* Blo = Bhi + TN * GLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const int id = blockIdx.x, xcd = id & 7, qd = id >> 3;
    const long long tile_m = (long long)(qd / tiles_n) * 8 + xcd;
    const int tile_n = qd % tiles_n;
    const long long m0 = tile_m * TM;
    const int n0 = tile_n * TN;
    if (m0 >= M) return;

    const float* a_ptr[NA];
    float a_rs[NA];
    int a_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + 512 * i;
        const int row = f >> 3, kq = f & 7;
        long long grow = m0 + row;
        if (grow > M - 1) grow = M - 1;
        a_ptr[i] = A + (grow / am.period) * am.outer + (grow % am.period) * (long long)am.inner + kq * 4;
        a_rs[i] = rscale ? eq16_lift(rscale[rs_div > 1 ? grow / rs_div : grow]) : 1.0f;
        a_off[i] = row * GLD + kq * 4;
    }
    if (tid < TM) {
        long long grow = m0 + tid;
        if (grow > M - 1) grow = M - 1;
        rinv[tid] = rscale ? 1.0f / eq16_lift(rscale[rs_div > 1 ? grow / rs_div : grow]) : 1.0f;
    }
    int w_src[NW], w_off[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int f = tid + 512 * i;
        const int row = f >> 2, part = f & 3;
        w_src[i] = min(n0 + row, N - 1) * K + part * 8;
        w_off[i] = row * GLD + part * 8;
    }
    float4 ra[NA];
    half8 rwh[NW], rwl[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4*>(a_ptr[i]);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        rwh[i] = *reinterpret_cast<const half8*>(Whi + w_src[i]);
        rwl[i] = *reinterpret_cast<const half8*>(Wlo + w_src[i]);
    }
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / GK;
    const int fa = (wm + (lane & 31)) * GLD + (lane >> 5) * 8;
    const int fb = (wn + (lane & 31)) * GLD + (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float sx = ra[i].x * a_rs[i], sy = ra[i].y * a_rs[i], sz = ra[i].z * a_rs[i], sw = ra[i].w * a_rs[i];
            half4 h, l;
            h[0] = (_Float16)sx; h[1] = (_Float16)sy; h[2] = (_Float16)sz; h[3] = (_Float16)sw;
            l[0] = (_Float16)(sx - (float)h[0]); l[1] = (_Float16)(sy - (float)h[1]);
            l[2] = (_Float16)(sz - (float)h[2]); l[3] = (_Float16)(sw - (float)h[3]);
            *reinterpret_cast<half4*>(Ahi + a_off[i]) = h;
            *reinterpret_cast<half4*>(Alo + a_off[i]) = l;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            *reinterpret_cast<half8*>(Bhi + w_off[i]) = rwh[i];
            *reinterpret_cast<half8*>(Blo + w_off[i]) = rwl[i];
        }
        __syncthreads();
        if (kt + 1 < nk) {
            const int k1 = (kt + 1) * GK;
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + k1);
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                rwh[i] = *reinterpret_cast<const half8*>(Whi + w_src[i] + k1);
                rwl[i] = *reinterpret_cast<const half8*>(Wlo + w_src[i] + k1);
            }
        }
        // Fragment reads run one row block ahead of the MFMAs that consume them (the compiler's own order waits on a read
        // right after issuing it, ~14 exposed LDS round trips per k-tile): the 6 MFMAs of block i cover the reads of i + 1.
        half8 bh[2][NJ], bl[2][NJ], ah[2], al[2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bh[0][j] = *reinterpret_cast<const half8*>(Bhi + fb + j * 32 * GLD);
            bl[0][j] = *reinterpret_cast<const half8*>(Blo + fb + j * 32 * GLD);
        }
        ah[0] = *reinterpret_cast<const half8*>(Ahi + fa);
        al[0] = *reinterpret_cast<const half8*>(Alo + fa);
#pragma unroll
        for (int step = 0; step < 2 * MI; ++step) {
            const int ks = step / MI, i = step % MI, cur = step & 1, nxt = cur ^ 1;
            if (step + 1 < 2 * MI) {
                const int ks2 = (step + 1) / MI, i2 = (step + 1) % MI;
                ah[nxt] = *reinterpret_cast<const half8*>(Ahi + fa + i2 * 32 * GLD + ks2 * 16);
                al[nxt] = *reinterpret_cast<const half8*>(Alo + fa + i2 * 32 * GLD + ks2 * 16);
                if (i2 == 0) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        bh[ks2][j] = *reinterpret_cast<const half8*>(Bhi + fb + j * 32 * GLD + ks2 * 16);
                        bl[ks2][j] = *reinterpret_cast<const half8*>(Blo + fb + j * 32 * GLD + ks2 * 16);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bh[ks][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bl[ks][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bh[ks][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    const float isc = *inv_scale;
    const int q = lane & 31;
    __syncthreads();  // all waves are done reading the operand tiles
    float* T = reinterpret_cast<float*>(lds) + wave * (32 * GTLD);  // [32 rows][64] floats per wave (8 x 8 KB <= 80 KB)
    const int cb = n0 + wn;
    const float bv0 = (bias && cb + q < N) ? bias[cb + q] : 0.f;
    const float bv1 = (bias && cb + 32 + q < N) ? bias[cb + 32 + q] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float sc = isc * rinv[wm + 32 * i + lr];
            float v0 = acc[i][0][r] * sc + bv0, v1 = acc[i][1][r] * sc + bv1;
            if (ACT == 2) { v0 = eq16_silu(v0); v1 = eq16_silu(v1); }
            T[lr * GTLD + q] = v0;
            T[lr * GTLD + 32 + q] = v1;
            if (out_mag) {
                float mg = fmaxf(cb + q < N ? fabsf(v0) : 0.f, cb + 32 + q < N ? fabsf(v1) : 0.f);
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) mg = fmaxf(mg, __shfl_xor(mg, o));
                const long long orow = m0 + wm + 32 * i + lr;
                if (q == 0 && orow < M) atomicMax(out_mag + orow, __float_as_uint(mg));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = lane + 64 * it;
            const int lr = item >> 4, c4 = item & 15;
            const long long row = m0 + wm + 32 * i + lr;
            const int col = cb + 4 * c4;
            if (row < M && col < N) {
                float* cp = Cm + (row / cm.period) * cm.outer + (row % cm.period) * (long long)cm.inner + col;
                float4 v = *reinterpret_cast<const float4*>(T + lr * GTLD + 4 * c4);
                if (ACCUM) {
                    const float4 o = *reinterpret_cast<const float4*>(cp);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *reinterpret_cast<float4*>(cp) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Pre-split A operand: the producer (rotate-in) already wrote fp16 hi / lo images of the rows, lifted by the row's own
// power of two (mag[row] = the row's max |a| before the lift), so staging is a pure 16-byte copy — the fp32 kernel above
// spends ~4 VALU instructions per MFMA on the conversion (SQ counters, profiles/).  256 x 256 x 32 tile, 8 waves; LDS rows
// are 64 bytes, unpadded, with the 16-byte chunk index XOR-ed by (row >> 2) & 3 (conflict-free ds_write_b128 and
// ds_read_b128); TWO LDS buffers: tile t+1 is written and tile t+2 requested while tile t is multiplied, one barrier
// per k-tile; fragment reads run one row block ahead of their MFMAs.
template <int ACT>
__global__ __launch_bounds__(512, 2) void eq_gemm16p_kernel(const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo,
                                                            const float* __restrict__ mag, const _Float16* __restrict__ Whi,
                                                            const _Float16* __restrict__ Wlo, const float* __restrict__ inv_scale,
                                                            const float* __restrict__ bias, float* __restrict__ Cm, int ldc,
                                                            long long M, int N, int K, int tiles_n) {
    constexpr int MI = 4, NJ = 2, TM = 256, TN = 256, RB = 32;  // RB halves (64 B) per LDS row
    constexpr int BUF = (2 * TM + 2 * TN) * RB;                  // halves per buffer (64 KB)
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsp[];
    __shared__ float rinv[TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const int id = blockIdx.x, xcd = id & 7, qd = id >> 3;
    const long long tile_m = (long long)(qd / tiles_n) * 8 + xcd;
    const int tile_n = qd % tiles_n;
    const long long m0 = tile_m * TM;
    const int n0 = tile_n * TN;
    if (m0 >= M) return;

    // staging: 2 chunks of 16 B per thread and image: f = tid + 512 i -> row f >> 2, chunk f & 3
    size_t a_src[2];
    int w_src[2], st_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 512 * i;
        const int row = f >> 2, part = f & 3;
        long long grow = m0 + row;
        if (grow > M - 1) grow = M - 1;
        a_src[i] = (size_t)grow * K + part * 8;
        w_src[i] = min(n0 + row, N - 1) * K + part * 8;
        st_off[i] = row * RB + ((part ^ ((row >> 2) & 3)) * 8);
    }
    if (tid < TM) {
        long long grow = m0 + tid;
        if (grow > M - 1) grow = M - 1;
        rinv[tid] = 1.0f / eq16_lift(mag[grow]);
    }
    half8 rah[2], ral[2], rwh[2], rwl[2];
    auto load_tile = [&](int k1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rah[i] = *reinterpret_cast<const half8*>(Ahi + a_src[i] + k1);
            ral[i] = *reinterpret_cast<const half8*>(Alo + a_src[i] + k1);
            rwh[i] = *reinterpret_cast<const half8*>(Whi + w_src[i] + k1);
            rwl[i] = *reinterpret_cast<const half8*>(Wlo + w_src[i] + k1);
        }
    };
    auto store_tile = [&](_Float16* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<half8*>(buf + st_off[i]) = rah[i];
            *reinterpret_cast<half8*>(buf + TM * RB + st_off[i]) = ral[i];
            *reinterpret_cast<half8*>(buf + 2 * TM * RB + st_off[i]) = rwh[i];
            *reinterpret_cast<half8*>(buf + (2 * TM + TN) * RB + st_off[i]) = rwl[i];
        }
    };
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / GK;
    // fragment addresses: row (lane & 31) of a 32-row block, chunk c = 2 ks + (lane >> 5), swizzled by (row >> 2) & 3
    // (the 32-row blocks start at multiples of 32, so the swizzle depends on lane only)
    const int frow = lane & 31, fsw = (frow >> 2) & 3, fkh = lane >> 5;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * RB + (((2 * ks + fkh) ^ fsw) * 8);
    load_tile(0);
    store_tile(ldsp);
    if (nk > 1) load_tile(GK);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        _Float16* cur = ldsp + (kt & 1) * BUF;
        _Float16* nxt = ldsp + ((kt + 1) & 1) * BUF;
        if (kt + 1 < nk) store_tile(nxt);          // tile kt+1 (in registers since the previous iteration)
        if (kt + 2 < nk) load_tile((kt + 2) * GK);  // tile kt+2: its latency hides under this tile's MFMAs
        const _Float16* Ah = cur + wm * RB;
        const _Float16* Al = cur + TM * RB + wm * RB;
        const _Float16* Bh = cur + 2 * TM * RB + wn * RB;
        const _Float16* Bl = cur + (2 * TM + TN) * RB + wn * RB;
        half8 bh[2][NJ], bl[2][NJ], ah[2], al[2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bh[0][j] = *reinterpret_cast<const half8*>(Bh + j * 32 * RB + foff[0]);
            bl[0][j] = *reinterpret_cast<const half8*>(Bl + j * 32 * RB + foff[0]);
        }
        ah[0] = *reinterpret_cast<const half8*>(Ah + foff[0]);
        al[0] = *reinterpret_cast<const half8*>(Al + foff[0]);
#pragma unroll
        for (int step = 0; step < 2 * MI; ++step) {
            const int ks = step / MI, i = step % MI, c = step & 1, nx = c ^ 1;
            if (step + 1 < 2 * MI) {
                const int ks2 = (step + 1) / MI, i2 = (step + 1) % MI;
                ah[nx] = *reinterpret_cast<const half8*>(Ah + i2 * 32 * RB + foff[ks2]);
                al[nx] = *reinterpret_cast<const half8*>(Al + i2 * 32 * RB + foff[ks2]);
                if (i2 == 0) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        bh[ks2][j] = *reinterpret_cast<const half8*>(Bh + j * 32 * RB + foff[ks2]);
                        bl[ks2][j] = *reinterpret_cast<const half8*>(Bl + j * 32 * RB + foff[ks2]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[c], bh[ks][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bl[ks][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bh[ks][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    const float isc = *inv_scale;
    const int q = lane & 31;
    float* T = reinterpret_cast<float*>(ldsp) + wave * (32 * GTLD);  // [32 rows][64] floats per wave
    const int cb = n0 + wn;
    const float bv0 = (bias && cb + q < N) ? bias[cb + q] : 0.f;
    const float bv1 = (bias && cb + 32 + q < N) ? bias[cb + 32 + q] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float sc = isc * rinv[wm + 32 * i + lr];
            float v0 = acc[i][0][r] * sc + bv0, v1 = acc[i][1][r] * sc + bv1;
            if (ACT == 2) { v0 = eq16_silu(v0); v1 = eq16_silu(v1); }
            T[lr * GTLD + q] = v0;
            T[lr * GTLD + 32 + q] = v1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = lane + 64 * it;
            const int lr = item >> 4, c4 = item & 15;
            const long long row = m0 + wm + 32 * i + lr;
            const int col = cb + 4 * c4;
            if (row < M && col < N)
                *reinterpret_cast<float4*>(Cm + row * (long long)ldc + col) = *reinterpret_cast<const float4*>(T + lr * GTLD + 4 * c4);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int32_t eq_launch_gemm16p(const void* Ahi, const void* Alo, const float* mag, const adf_w16* W, const float* bias, float* Cm,
                          int ldc, long long M, int N, int K, int act, hipStream_t s) {
    if (M <= 0 || N <= 0) return ADF_OK;
    if (K % GK != 0 || (N & 3) || (ldc & 3)) { adf_set_error("eq_gemm16p: bad shape"); return ADF_EINVAL; }
    const int tiles_n = (N + 255) / 256;
    const long long tiles_m8 = ((M + 255) / 256 + 7) / 8 * 8;
    const long long nb = tiles_m8 * tiles_n;
    if (nb > 0x7fffffffLL) { adf_set_error("eq_gemm16p: grid too large"); return ADF_EINVAL; }
    const size_t dyn = 2 * (size_t)(2 * 256 + 2 * 256) * 32 * sizeof(_Float16);  // 128 KB
    if (act == 2) {
        ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&eq_gemm16p_kernel<2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(eq_gemm16p_kernel<2>, dim3((unsigned)nb), dim3(512), dyn, s, (const _Float16*)Ahi, (const _Float16*)Alo,
                           mag, (const _Float16*)W->hi, (const _Float16*)W->lo, W->inv_scale, bias, Cm, ldc, M, N, K, tiles_n);
    } else {
        ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&eq_gemm16p_kernel<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(eq_gemm16p_kernel<0>, dim3((unsigned)nb), dim3(512), dyn, s, (const _Float16*)Ahi, (const _Float16*)Alo,
                           mag, (const _Float16*)W->hi, (const _Float16*)W->lo, W->inv_scale, bias, Cm, ldc, M, N, K, tiles_n);
    }
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// The same product with the WEIGHTS STREAMED AS MFMA FRAGMENTS (round 6; the form gemm16.hip's node products took this
// round): every wave loads the B operands of its own two column blocks straight from the fragment-ordered image
// (adf_pack_frag, mlp16.hip) into a ring of four register sets, two k-steps ahead; LDS holds only the A tile (two buffers,
// hi + lo, 64-byte rows with the chunk XOR-swizzle of the kernel above), one barrier per K tile, the next tile's rows are
// requested two tiles ahead in two register sets and copied into the other buffer between the products of the second
// k-step.  Per K tile a CU's LDS moves (TM / 256) x (128 KB of fragment reads + 32 KB of writes) instead of 196 + 64 KB.
// Same products in the same order as eq_gemm16p_kernel: same bits.
//   NWN = 4: 8 waves as 2 (M) x 4 (N), tile (64 MI) x 256, one workgroup per CU.
//   NWN = 2: 4 waves as 2 x 2, tile (64 MI) x 128, two workgroups per CU: the remainder columns of an N that is not a
//   multiple of 256 (order 2 of config 4: 640 = 2 x 256 + 128) without idle column waves.
// Columns [col0, col0 + tiles_n * TN) of C; needs K / 32 even and N % 32 == 0 (launcher).
// TMI: tile rows / 64; NWN: waves / 2 (tile columns 64 NWN); the 2 NWN waves sit as (2 NWN / WCOLS) x WCOLS, each on MI row blocks
// x 2 column blocks: MI = TMI, WCOLS = NWN is the full tile; MI = 2, WCOLS = 2 on eight waves the 256 x 128 HALF tile (below)
template <int ACT, int TMI, int NWN, int MI, int WCOLS>
__device__ __forceinline__ void eq_gemm16pw_tile(
    const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo, const float* __restrict__ mag,
    const half8* __restrict__ Wf, const float* __restrict__ inv_scale, const float* __restrict__ bias, float* __restrict__ Cm,
    int ldc, long long M, int N, int K, int tiles_n, int col0, _Float16* ldsw, float* rinv) {
    constexpr int NJ = 2, NT = 128 * NWN, TM = 64 * TMI, TN = 64 * NWN, RB = 32;
    static_assert((2 * NWN / WCOLS) * 32 * MI == TM, "the waves' row blocks cover the tile");
    constexpr int NA = TM * 8 / NT;          // 16-byte chunks per thread and K tile (hi and lo planes together)
    constexpr int PLANE = TM * RB;           // halves per plane
    constexpr int BUF = 2 * PLANE;           // halves per buffer
    static_assert(TM * 8 % NT == 0, "whole chunks per thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WCOLS) * (32 * MI), wn = (wave % WCOLS) * 64;
    const int id = blockIdx.x, xcd = id & 7, qd = id >> 3;
    const long long tile_m = (long long)(qd / tiles_n) * 8 + xcd;
    const int tile_n = qd % tiles_n;
    const long long m0 = tile_m * TM;
    const int n0 = col0 + tile_n * TN;
    if (m0 >= M) return;

    // staging: chunk f = tid + NT i -> plane f / (4 TM), row (f % (4 TM)) >> 2, 16-byte part f & 3
    const _Float16* a_src[NA];
    int st_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + NT * i;
        const int plane = f / (4 * TM), g = f % (4 * TM);
        const int row = g >> 2, part = g & 3;
        long long grow = m0 + row;
        if (grow > M - 1) grow = M - 1;
        a_src[i] = (plane ? Alo : Ahi) + (size_t)grow * K + part * 8;
        st_off[i] = plane * PLANE + row * RB + ((part ^ ((row >> 2) & 3)) * 8);
    }
    if (tid < TM) {
        long long grow = m0 + tid;
        if (grow > M - 1) grow = M - 1;
        rinv[tid] = 1.0f / eq16_lift(mag[grow]);
    }
    half8 ra[NA], rb[NA];
    auto request = [&](half8 (&r)[NA], int kt1) {
#pragma unroll
        for (int i = 0; i < NA; ++i) r[i] = *reinterpret_cast<const half8*>(a_src[i] + (size_t)kt1 * GK);
    };
    auto store = [&](const half8 (&r)[NA], int boff) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<half8*>(ldsw + boff + st_off[i]) = r[i];
    };
    // B fragments of one 16-deep k-step for this wave's two column blocks: [column block][hi | lo]; a column block past N
    // (the dead half of a last tile) reads block 0 and is never stored
    const int nks = K / 16;
    const int cbw = (n0 + wn) / 32;
    const half8* wfr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wfr[j] = Wf + (size_t)((cbw + j) * 32 < N ? cbw + j : 0) * nks * 128 + lane;
    auto load_wf = [&](int s_, half8 (&wf)[NJ][2]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            wf[j][0] = wfr[j][(size_t)s_ * 128];
            wf[j][1] = wfr[j][(size_t)s_ * 128 + 64];
        }
    };
    half8 wfa[NJ][2], wfb[NJ][2], wfc[NJ][2], wfd[NJ][2];
    request(ra, 0);
    load_wf(0, wfa);
    load_wf(1, wfb);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / GK;
    const int frow = lane & 31, fsw = (frow >> 2) & 3, fkh = lane >> 5;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = (wm + frow) * RB + (((2 * ks + fkh) ^ fsw) * 8);
    auto kstep = [&](int boff, int ks, const half8 (&cur)[NJ][2]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const half8 ah = *reinterpret_cast<const half8*>(ldsw + boff + i * 32 * RB + foff[ks]);
            const half8 al = *reinterpret_cast<const half8*>(ldsw + boff + PLANE + i * 32 * RB + foff[ks]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, cur[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[j][0], acc[i][j], 0, 0, 0);
            }
        }
    };
    // tile kt: products on buffer `cur`; `nxt` (tile kt + 1, in registers) is copied into the other buffer between the
    // products of the second k-step; (w0, w1) are this tile's fragments, the next tile's are requested into (w2, w3).
    // sched_barrier(0) between request group and product group: left alone, hipcc sinks the requests to their first use.
    auto tile = [&](int kt, int cur, const half8 (&nxt)[NA], const half8 (&w0)[NJ][2], const half8 (&w1)[NJ][2],
                    half8 (&w2)[NJ][2], half8 (&w3)[NJ][2]) {
        load_wf(min(2 * kt + 2, 2 * nk - 1), w2);
        __builtin_amdgcn_sched_barrier(0);
        kstep(cur, 0, w0);
        __builtin_amdgcn_sched_barrier(0);
        load_wf(min(2 * kt + 3, 2 * nk - 1), w3);
        __builtin_amdgcn_sched_barrier(0);
        kstep(cur, 1, w1);
        store(nxt, BUF - cur);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * MI, 0);
#pragma unroll
        for (int u = 0; u < 3 * MI * NJ; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if (u % 3 == 2 && u / 3 < NA) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    store(ra, 0);                          // tile 0
    request(ra, min(1, nk - 1));
    request(rb, min(2, nk - 1));
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {   // (nk is even: launcher)
        tile(kt, 0, ra, wfa, wfb, wfc, wfd);
        request(ra, min(kt + 3, nk - 1));
        __syncthreads();
        tile(kt + 1, BUF, rb, wfc, wfd, wfa, wfb);
        request(rb, min(kt + 4, nk - 1));
        __syncthreads();
    }

    const float isc = *inv_scale;
    const int q = lane & 31;
    float* T = reinterpret_cast<float*>(ldsw) + wave * (32 * GTLD);  // [32 rows][64] floats per wave
    const int cb = n0 + wn;
    if (cb >= N) return;   // (no barrier below)
    const float bv0 = (bias && cb + q < N) ? bias[cb + q] : 0.f;
    const float bv1 = (bias && cb + 32 + q < N) ? bias[cb + 32 + q] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float sc = isc * rinv[wm + 32 * i + lr];
            float v0 = acc[i][0][r] * sc + bv0, v1 = acc[i][1][r] * sc + bv1;
            if (ACT == 2) { v0 = eq16_silu(v0); v1 = eq16_silu(v1); }
            T[lr * GTLD + q] = v0;
            T[lr * GTLD + 32 + q] = v1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = lane + 64 * it;
            const int lr = item >> 4, c4 = item & 15;
            const long long row = m0 + wm + 32 * i + lr;
            const int col = cb + 4 * c4;
            if (row < M && col < N)
                *reinterpret_cast<float4*>(Cm + row * (long long)ldc + col) = *reinterpret_cast<const float4*>(T + lr * GTLD + 4 * c4);
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// A last column tile with at most 128 live columns (order 2 of config 4: N = 640 = 2 x 256 + 128) runs the HALF layout: the same
// 256-row A tile - shared through L2 with the workgroups of the full column tiles beside it - on eight waves as 4 (M) x 2 (N), two row
// blocks each: no idle column waves and no second pass over the operand rows (a separate four-wave launch for those columns re-read
// all of A: 3.4 GB for a fifth of the columns, MfmaUtil 32 % beside 56 %).
template <int ACT, int MI, int NWN>
__global__ __launch_bounds__(128 * NWN, NWN == 4 ? 1 : 2) void eq_gemm16pw_kernel(
    const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo, const float* __restrict__ mag,
    const half8* __restrict__ Wf, const float* __restrict__ inv_scale, const float* __restrict__ bias, float* __restrict__ Cm,
    int ldc, long long M, int N, int K, int tiles_n, int col0) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsw[];
    __shared__ float rinv[64 * MI];
    if constexpr (NWN == 4 && MI == 4) {
        const int n0 = col0 + (int)((blockIdx.x >> 3) % tiles_n) * 256;
        if (N - n0 <= 128) {
            eq_gemm16pw_tile<ACT, 4, 4, 2, 2>(Ahi, Alo, mag, Wf, inv_scale, bias, Cm, ldc, M, N, K, tiles_n, col0, ldsw, rinv);
            return;
        }
    }
    eq_gemm16pw_tile<ACT, MI, NWN, MI, NWN>(Ahi, Alo, mag, Wf, inv_scale, bias, Cm, ldc, M, N, K, tiles_n, col0, ldsw, rinv);
}

template <int ACT, int MI, int NWN>
static int32_t eq_gemm16pw_go(const void* Ahi, const void* Alo, const float* mag, const adf_w16* W, const float* bias, float* Cm,
                              int ldc, long long M, int N, int K, int col0, int ncols, hipStream_t s) {
    constexpr int TM = 64 * MI, TN = 64 * NWN;
    const int tiles_n = (ncols + TN - 1) / TN;
    const long long tiles_m8 = ((M + TM - 1) / TM + 7) / 8 * 8;
    const long long nb = tiles_m8 * tiles_n;
    if (nb > 0x7fffffffLL) { adf_set_error("eq_gemm16pw: grid too large"); return ADF_EINVAL; }
    // two A buffers; the epilogue's per-wave transposition scratch (2 NWN waves x 8 KB) lives in the same bytes
    size_t dyn = (size_t)2 * 2 * TM * 32 * sizeof(_Float16);
    const size_t scratch = (size_t)2 * NWN * 32 * GTLD * sizeof(float);
    if (dyn < scratch) dyn = scratch;
    auto kern = &eq_gemm16pw_kernel<ACT, MI, NWN>;
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(128 * NWN), dyn, s, (const _Float16*)Ahi, (const _Float16*)Alo, mag,
                       (const half8*)W->frag, W->inv_scale, bias, Cm, ldc, M, N, K, tiles_n, col0);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// whether eq_launch_gemm16pw takes the shape (the weight needs its fragment image)
bool eq_gemm16pw_ok(const adf_w16* W, int N, int K) { return W->frag && N % 32 == 0 && K % 64 == 0 && N >= 128; }

int32_t eq_launch_gemm16pw(const void* Ahi, const void* Alo, const float* mag, const adf_w16* W, const float* bias, float* Cm,
                           int ldc, long long M, int N, int K, int act, hipStream_t s) {
    if (M <= 0 || N <= 0) return ADF_OK;
    if (!eq_gemm16pw_ok(W, N, K)) { adf_set_error("eq_gemm16pw: shape or fragment image"); return ADF_EINVAL; }
    static int mi = -1;   // rows per tile: 256 (default; 1.5 % faster than 192 on config 4's shapes), ADF_EQV2_PW_MI=3: 192
    if (mi < 0) { const char* e = getenv("ADF_EQV2_PW_MI"); mi = (e && atoi(e) == 3) ? 3 : 4; }
    // whole 256-column tiles on eight waves; a remainder of at most 128 columns on the four-wave form
    const int rem = N % 256;
    // 256-row tiles: a last tile of at most 128 columns runs the kernel's half layout; 192-row tiles (ADF_EQV2_PW_MI=3): the
    // four-wave remainder launch
    const int wide = (mi == 3 && rem > 0 && rem <= 128) ? N - rem : N;
#define EQ_PW(ACT_, MI_, NWN_, C0_, NC_) eq_gemm16pw_go<ACT_, MI_, NWN_>(Ahi, Alo, mag, W, bias, Cm, ldc, M, N, K, C0_, NC_, s)
    if (wide > 0) {
        if (act == 2) { if (mi == 4) ADF_TRY(EQ_PW(2, 4, 4, 0, wide)); else ADF_TRY(EQ_PW(2, 3, 4, 0, wide)); }
        else { if (mi == 4) ADF_TRY(EQ_PW(0, 4, 4, 0, wide)); else ADF_TRY(EQ_PW(0, 3, 4, 0, wide)); }
    }
    if (wide < N) {
        if (act == 2) ADF_TRY(EQ_PW(2, 3, 2, wide, N - wide)); else ADF_TRY(EQ_PW(0, 3, 2, wide, N - wide));
    }
#undef EQ_PW
    return ADF_OK;
}

// A [M, K] fp32 + row magnitudes -> lifted fp16 hi / lo images (unit-test / benchmark path of eq_gemm16p_kernel; in the
// model the producer kernel writes them directly)
__global__ void eq_presplit_kernel(const float* __restrict__ A, const float* __restrict__ mag, long long M, int K,
                                   _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * K) return;
    const float sv = A[t] * eq16_lift(mag[t / K]);
    const _Float16 h = (_Float16)sv;
    hi[t] = h;
    lo[t] = (_Float16)(sv - (float)h);
}

int32_t eq_launch_presplit(const float* A, const float* mag, long long M, int K, void* hi, void* lo, hipStream_t s) {
    const long long n = M * K;
    hipLaunchKernelGGL(eq_presplit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A, mag, M, K, (_Float16*)hi,
                       (_Float16*)lo);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// shapes the kernel takes: K % 32 == 0, N % 4 == 0, 16-byte aligned rows of A and C
bool eq_gemm16_ok(const float* A, const eq_rowmap* am, const float* Cm, const eq_rowmap* cm, int N, int K) {
    if (K % GK != 0 || (N & 3)) return false;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(Cm) & 15)) return false;
    if ((am->outer & 3) || (am->inner & 3) || (cm->outer & 3) || (cm->inner & 3)) return false;
    return true;
}

int32_t eq_launch_gemm16(const float* A, const eq_rowmap* am, const float* rscale, const adf_w16* W, const float* bias,
                         float* Cm, const eq_rowmap* cm, long long M, int N, int K, int act, bool accumulate,
                         hipStream_t s, float* out_mag, int rs_div) {
    if (M <= 0 || N <= 0) return ADF_OK;
    static int big = -1;
    if (big < 0) { const char* e = getenv("ADF_EQV2_GEMM_TILE"); big = (e && atoi(e) == 128) ? 0 : 1; }
    if (big && N >= 256 && M >= 8192) {
        const int tiles_n = (N + 255) / 256;
        const long long tiles_m8 = ((M + 255) / 256 + 7) / 8 * 8;
        const long long nb = tiles_m8 * tiles_n;
        if (nb > 0x7fffffffLL) { adf_set_error("eq_gemm16: grid too large"); return ADF_EINVAL; }
#define EQ_L256(ACT_, ACC_)                                                                                          \
    hipLaunchKernelGGL((eq_gemm16_256_kernel<ACT_, ACC_>), dim3((unsigned)nb), dim3(512), 0, s, A, *am, rscale,      \
                       (const _Float16*)W->hi, (const _Float16*)W->lo, W->inv_scale, bias, Cm, *cm, M, N, K, tiles_n, \
                       reinterpret_cast<unsigned int*>(out_mag), rs_div)
        if (act == 2) { if (accumulate) EQ_L256(2, true); else EQ_L256(2, false); }
        else { if (accumulate) EQ_L256(0, true); else EQ_L256(0, false); }
#undef EQ_L256
        ADF_HIP_CHECK(hipGetLastError());
        return ADF_OK;
    }
    const int NJ = N <= 128 ? 2 : 4;
    const int TN = 64 * NJ;
    const int tiles_n = (N + TN - 1) / TN;
    const long long tiles_m = (M + 127) / 128;
    const long long tiles_m8 = (tiles_m + 7) / 8 * 8;
    const long long nblocks = tiles_m8 * tiles_n;
    if (nblocks > 0x7fffffffLL) { adf_set_error("eq_gemm16: grid too large"); return ADF_EINVAL; }
    dim3 grid((unsigned)nblocks);
#define EQ_L16(ACT_, NJ_, ACC_)                                                                                       \
    hipLaunchKernelGGL((eq_gemm16_kernel<ACT_, NJ_, ACC_>), grid, dim3(256), 0, s, A, *am, rscale,                    \
                       (const _Float16*)W->hi, (const _Float16*)W->lo, W->inv_scale, bias, Cm, *cm, M, N, K, tiles_n, \
                       reinterpret_cast<unsigned int*>(out_mag), rs_div)
    if (NJ == 2) {
        if (act == 2) { if (accumulate) EQ_L16(2, 2, true); else EQ_L16(2, 2, false); }
        else { if (accumulate) EQ_L16(0, 2, true); else EQ_L16(0, 2, false); }
    } else {
        if (act == 2) { if (accumulate) EQ_L16(2, 4, true); else EQ_L16(2, 4, false); }
        else { if (accumulate) EQ_L16(0, 4, true); else EQ_L16(0, 4, false); }
    }
#undef EQ_L16
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
