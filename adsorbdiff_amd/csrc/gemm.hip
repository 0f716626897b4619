// Dense per-node / per-row linear blocks on the f32 matrix cores.
//
//   C[M,N] = act( A[M,K] . W[N,K]^T + bias[N] )          (torch.nn.Linear layout)
//
// Used for every nn.Linear of the PaiNN denoiser that acts on node rows
// (reference: painn_denoising.py:531 x_proj, :603 vec_proj, :609 xvec_proj,
// :689-693 gated-equivariant blocks).  Exact f32: v_mfma_f32_32x32x2_f32 is a
// k-ordered fmaf chain (cdna guide §3), so parity with the CPU reference is at
// f32 rounding level, no reduced-precision split.
//
// Tiling (gfx950): 128x128 output tile per 256-thread workgroup, 4 waves as
// 2(M) x 2(N), each wave 64x64 = 2x2 MFMA 32x32 accumulators (64 acc VGPRs).
// K is consumed in 32-deep tiles staged through LDS k-major ([k][row], row
// stride 129 floats) so that an MFMA operand fetch is one conflict-free
// ds_read_b32 (lanes 0-31 -> 32 consecutive rows of k, lanes 32-63 -> k+1).
// Global loads are 16 B/lane, 8 lanes per 128-B row segment, prefetched into
// registers one K-tile ahead.  M is the long dimension (atoms x 1 or x 3), so
// the block->tile map keeps all N-tiles of one M-panel on one XCD (blocks b and
// b+8 share an XCD): the A panel is fetched from HBM once and re-read from that
// XCD's L2; W (<= 3 MB) lives in every L2.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define BK 32
#define LDT 129

__device__ __forceinline__ float ssilu_f(float x) {
    // ScaledSiLU: silu(x) * (1/0.6)   (gemnet_oc/layers/base_layers.py:65-72)
    float s = x / (1.0f + expf(-x));
    return s * 1.6666666666666667f;
}

template <int ACT>
__global__ __launch_bounds__(256) void adf_gemm_kernel(const float* __restrict__ A, int lda,
                                                        const float* __restrict__ W, int ldw,
                                                        const float* __restrict__ bias, float* __restrict__ C,
                                                        int ldc, int M, int N, int K, int tiles_n) {
    __shared__ float As[BK * LDT];
    __shared__ float Ws[BK * LDT];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave >> 1) * 64;
    const int wn = (wave & 1) * 64;

    // XCD-aware tile map: blocks with equal (id % 8) share an XCD.
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int q = id >> 3;
    const int tile_m = (q / tiles_n) * 8 + xcd;
    const int tile_n = q % tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    if (m0 >= M) return;

    // staging map: 4 float4 per operand per thread; 8 consecutive lanes cover one 128-B row segment
    const float* a_ptr[4];
    const float* w_ptr[4];
    int st_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int f = tid + 256 * i;
        int row = f >> 3, kq = f & 7;
        int ar = min(m0 + row, M - 1);
        int wr = min(n0 + row, N - 1);
        a_ptr[i] = A + (size_t)ar * lda + kq * 4;
        w_ptr[i] = W + (size_t)wr * ldw + kq * 4;
        st_off[i] = (kq * 4) * LDT + row;
    }
    float4 ra[4], rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const float4*>(a_ptr[i]);
        rw[i] = *reinterpret_cast<const float4*>(w_ptr[i]);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    const int frag = (lane >> 5) * LDT + (lane & 31);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* pa = As + st_off[i];
            pa[0] = ra[i].x; pa[LDT] = ra[i].y; pa[2 * LDT] = ra[i].z; pa[3 * LDT] = ra[i].w;
            float* pw = Ws + st_off[i];
            pw[0] = rw[i].x; pw[LDT] = rw[i].y; pw[2 * LDT] = rw[i].z; pw[3 * LDT] = rw[i].w;
        }
        __syncthreads();
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (size_t)(kt + 1) * BK);
                rw[i] = *reinterpret_cast<const float4*>(w_ptr[i] + (size_t)(kt + 1) * BK);
            }
        }
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const float* pa = As + (2 * s) * LDT + frag + wm;
            const float* pw = Ws + (2 * s) * LDT + frag + wn;
            float a0 = pa[0], a1 = pa[32];
            float b0 = pw[0], b1 = pw[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // epilogue: lane owns column (lane&31) of each 32x32 block, 16 rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn + 32 * j + (lane & 31);
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M) {
                    float v = acc[i][j][r] + bv;
                    if (ACT) v = ssilu_f(v);
                    C[(size_t)row * ldc + col] = v;
                }
            }
        }
    }
}

int32_t adf_launch_gemm(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                        int M, int N, int K, int act_ssilu, hipStream_t s) {
    if (M <= 0) return ADF_OK;
    if (K % BK != 0 || (lda & 3) || (ldw & 3)) {
        adf_set_error("gemm: K=%d must be a multiple of %d and lda/ldw multiples of 4", K, BK);
        return ADF_EINVAL;
    }
    const int tiles_n = (N + BN - 1) / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int tiles_m8 = (tiles_m + 7) / 8 * 8;
    dim3 grid((unsigned)(tiles_m8 * tiles_n));
    if (act_ssilu)
        hipLaunchKernelGGL(adf_gemm_kernel<1>, grid, dim3(256), 0, s, A, lda, W, ldw, bias, C, ldc, M, N, K, tiles_n);
    else
        hipLaunchKernelGGL(adf_gemm_kernel<0>, grid, dim3(256), 0, s, A, lda, W, ldw, bias, C, ldc, M, N, K, tiles_n);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
