// Training step of the PaiNN denoiser (score matching, SURVEY.md 8f-1 / BASELINE config 5): the device ops that
// adsorbdiff_amd/train_step.py strings together into forward-with-saved-activations, loss and backward.
// Reference: adsorbdiff/trainers/sde_denoising_trainer.py:675-728 (_compute_loss), base_trainer.py:787-820
// (_backward = autograd + clip + AdamW + EMA) and the model of models/painn/painn_denoising.py.
//
// Arithmetic: exact f32 throughout (v_mfma_f32_32x32x2_f32 in the GEMMs), the reference trains in fp32
// (models/painn/README.md:12).  Correctness-first layout: the radial projection rbfh [E,3H] is materialised
// (SURVEY 8d's HBM-bound variant) so that its weight gradient is one GEMM over the edges.
//
// Backward of the message block without atomics: the graph is symmetric (every edge j->i has its reverse i->j with the
// same distance, graph.hip), so the gradient that flows to a SOURCE atom j is a sum over j's own CSR segment: for
// each incoming edge e' = (i -> j) the reverse edge (j -> i) carries the same rbfh row (a function of the distance
// only) and the negated unit vector.  The per-edge gradient of rbfh is written at row e' instead of the reverse
// edge's row - the sum over edges of drbfh[e] (x) rbf[e] that makes dW is invariant under that relabelling.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TR_CHECK_LAUNCH() ADF_HIP_CHECK(hipGetLastError())

static inline unsigned tr_grid(long long total, int per_block = 256) {
    long long b = (total + per_block - 1) / per_block;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

__device__ __forceinline__ float tr_wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ScaledSiLU and its derivative (gemnet_oc/layers/base_layers.py:65-72): f(x) = x sigmoid(x) / 0.6
__device__ __forceinline__ float tr_ssilu(float x) { return x / (1.0f + expf(-x)) * 1.6666666666666667f; }
__device__ __forceinline__ float tr_dssilu(float x) {
    const float s = 1.0f / (1.0f + expf(-x));
    return s * (1.0f + x * (1.0f - s)) * 1.6666666666666667f;
}

// ------------------------------------------------------------------------------------------------ linear layers
// C = A W^T (+ bias): the f16x3 split GEMM of the sampling path (gemm16.hip) with the same per-row power-of-two lifts
// (every A row is measured by adf_launch_rowmag and lifted before the split: activations of any magnitude, and the
// ~1e-6 gradient rows of the data-gradient products, keep both fp16 terms) and the weight split per call - in training
// the weights change every step; the hi/lo image and the row magnitudes live in per-device scratch that consecutive
// calls on one stream reuse in order.  ADF_TRAIN_GEMM=f32 selects the exact-f32 MFMA GEMM (gemm.hip) everywhere.
static int tr_gemm_mode() {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("ADF_TRAIN_GEMM"); mode = (e && strcmp(e, "f32") == 0) ? 0 : 1; }
    return mode;
}
// the f16x3 path takes this product (A [M, K] with row stride lda)
static bool tr_gemm16_ok(int lda, long long M, int K) {
    return tr_gemm_mode() == 1 && K % 32 == 0 && (lda & 3) == 0 && M * (long long)lda * 4 < (1ll << 32);
}
static int32_t tr_gemm(const float* A, int lda, const float* W, const float* bias, float* C, int ldc, long long M, int N,
                       int K, hipStream_t s, int accumulate = 0) {
    const int mode = tr_gemm_mode();
    const bool ok16 = mode == 1 && K % 32 == 0 && (lda & 3) == 0 && M * (long long)lda * 4 < (1ll << 32);
    if (accumulate && !ok16) { adf_set_error("internal: accumulating product needs the f16x3 path"); return ADF_EINVAL; }
    if (!ok16) return adf_launch_gemm(A, lda, W, K, bias, C, ldc, (int)M, N, K, 0, s);
    static unsigned char* buf[16] = {};
    static size_t cap[16] = {};
    int dev = 0;
    ADF_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) {
        if (accumulate) { adf_set_error("internal: accumulating product needs the f16x3 path"); return ADF_EINVAL; }
        return adf_launch_gemm(A, lda, W, K, bias, C, ldc, (int)M, N, K, 0, s);
    }
    // (+ the fragment-ordered image of the split weight: the streamed-weights kernel of gemm16.hip, round 6)
    const bool fragok = N % 32 == 0 && K % 16 == 0;
    const size_t n = (size_t)N * K, need = n * 4 + 256 + (fragok ? n * 4 : 0);
    if (need > cap[dev]) {
        ADF_HIP_CHECK(hipDeviceSynchronize());
        if (buf[dev]) (void)hipFree(buf[dev]);
        buf[dev] = nullptr; cap[dev] = 0;
        const size_t want = need + need / 4;
        if (hipMalloc(reinterpret_cast<void**>(&buf[dev]), want) != hipSuccess) {
            (void)hipGetLastError();
            if (accumulate) { adf_set_error("out of memory (weight image of a product)"); return ADF_EOOM; }
            return adf_launch_gemm(A, lda, W, K, bias, C, ldc, (int)M, N, K, 0, s);
        }
        cap[dev] = want;
    }
    static float* mag[16] = {};
    static long long mag_cap[16] = {};
    if (M > mag_cap[dev]) {
        ADF_HIP_CHECK(hipDeviceSynchronize());
        if (mag[dev]) (void)hipFree(mag[dev]);
        mag[dev] = nullptr; mag_cap[dev] = 0;
        const long long want = M + M / 4 + 1024;
        if (hipMalloc(reinterpret_cast<void**>(&mag[dev]), sizeof(float) * (size_t)want) != hipSuccess) {
            (void)hipGetLastError();
            if (accumulate) { adf_set_error("out of memory (weight image of a product)"); return ADF_EOOM; }
            return adf_launch_gemm(A, lda, W, K, bias, C, ldc, (int)M, N, K, 0, s);
        }
        mag_cap[dev] = want;
    }
    adf_w16 w16 = {};
    w16.hi = buf[dev]; w16.lo = buf[dev] + n * 2;
    w16.inv_scale = reinterpret_cast<float*>(buf[dev] + n * 4);
    w16.bias_perm = nullptr;
    ADF_TRY(adf_split_weight(W, (long long)n, &w16, reinterpret_cast<unsigned int*>(buf[dev] + n * 4 + 16), s));
    if (fragok) {
        w16.frag = buf[dev] + n * 4 + 256;
        ADF_TRY(adf_pack_frag(&w16, N, K, w16.frag, s));
    }
    const adf_lift lf = {mag[dev], mag_cap[dev]};
    return adf_launch_gemm16(A, lda, &w16, bias, C, ldc, (int)M, N, K, 0, s, nullptr, 0, &lf, nullptr, nullptr, nullptr, accumulate);
}

__global__ void tr_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int rr = by + r, cc = bx + threadIdx.x;
        t[r][threadIdx.x] = (rr < R && cc < C) ? src[(size_t)rr * C + cc] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int cc = bx + r, rr = by + threadIdx.x;
        if (cc < C && rr < R) dst[(size_t)cc * R + rr] = t[threadIdx.x][r];
    }
}

// dW[n,k] partial = sum over this block's rows m of dC[m,n] * A[m,k]: 64 x 64 output tile per 256-thread block,
// 4 waves x (32 x 32), rows in chunks of 32 staged row-major in LDS - both MFMA operands are read with the lane index
// along the contiguous dimension (A operand: lane (i = n, kk = m mod 2) <- dC[m + kk][n0 + i]).
// bpart != null: the k-tile-0 workgroups also form the column sums of their dC tile (bias gradient partials
// [splits][N]) from the rows they stage anyway.
__global__ __launch_bounds__(256) void tr_wgrad_kernel(const float* __restrict__ dC, int ldc, const float* __restrict__ A,
                                                        int lda, float* __restrict__ part, int M, int N, int K,
                                                        int rows_per_split, int tiles_k, float* __restrict__ bpart) {
    __shared__ float Cs[32][64];
    __shared__ float As[32][64];
    const int tile = blockIdx.x, split = blockIdx.y;
    const int n0 = (tile / tiles_k) * 64, k0 = (tile % tiles_k) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
    const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool colsum = bpart && k0 == 0 && tid < 64;
    float bs = 0.f;
    for (int m0 = mbeg; m0 < mend; m0 += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * 64; i += 256) {
            const int r = i >> 6, c = i & 63;
            const int m = m0 + r;
            Cs[r][c] = (m < mend && n0 + c < N) ? dC[(size_t)m * ldc + n0 + c] : 0.f;
            As[r][c] = (m < mend && k0 + c < K) ? A[(size_t)m * lda + k0 + c] : 0.f;
        }
        __syncthreads();
        if (colsum) {
#pragma unroll
            for (int r = 0; r < 32; ++r) bs += Cs[r][tid];  // rows in order: reproducible
        }
#pragma unroll
        for (int mm = 0; mm < 32; mm += 2) {
            const float a = Cs[mm + (lane >> 5)][wn + (lane & 31)];
            const float b = As[mm + (lane >> 5)][wk + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    float* out = part + (size_t)split * N * K;
    const int col = k0 + wk + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < N && col < K) out[(size_t)row * K + col] = acc[r];
    }
    if (colsum && n0 + tid < N) bpart[(size_t)split * N + n0 + tid] = bs;
}

// Same contraction on 128 x 128 tiles (N % 128 == 0, K % 128 == 0): each wave owns 64 x 64 = 2 x 2 MFMA blocks, so an LDS
// value feeds two MFMAs and a staged row serves twice as many; rows are staged with 16-byte loads.  A 32-column block of
// the A chunk that is entirely zero is skipped (exactly the same sums): the radial basis rows of rbf_proj's weight
// gradient - 5 M edge rows, the largest product of the step - are zero outside +-14 centres of the edge's distance, and
// consecutive edges of a target are sorted by distance.  (Measured: walking the edges in GLOBAL order of their length - a
// radix sort per step, rows gathered through an index - narrows the window of a chunk to one row's, but every workgroup
// then reads its 512-byte slice of scattered 6 KB rows: 835 vs 837 graphs/s, no gain; not kept.)
__global__ __launch_bounds__(256) void tr_wgrad128_kernel(const float* __restrict__ dC, int ldc, const float* __restrict__ A,
                                                           int lda, float* __restrict__ part, int M, int N, int K,
                                                           int rows_per_split, int tiles_k, float* __restrict__ bpart) {
    __shared__ __attribute__((aligned(16))) float Cs[32][128];
    __shared__ __attribute__((aligned(16))) float As[32][128];
    __shared__ unsigned int nzblk[2][4];  // [chunk parity][32-column block of As]: some element is non-zero
    if (threadIdx.x < 8) nzblk[threadIdx.x >> 2][threadIdx.x & 3] = 0u;
    const int tile = blockIdx.x, split = blockIdx.y;
    const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the two waves along k take the 32-column blocks {0, 2} and {1, 3} of the A chunk: a window of adjacent non-zero blocks
    // (see the header) is then shared between them
    const int wn = (wave >> 1) * 64, kb = wave & 1;
    const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool colsum = bpart && k0 == 0 && tid < 128;
    float bs = 0.f;
    const int lr = tid >> 5, lc = (tid & 31) * 4;  // staging: 8 rows x 32 float4 per pass, 4 passes
    int par = 0;
    // The rows of chunk i + 1 are requested right after chunk i's rows have been converted into the LDS image, so they are in
    // flight while chunk i's MFMAs run (round 5; requested at the top of their own iteration they cost every 32-row chunk
    // a full memory round trip in front of its barrier).
    float4 c4[4], a4[4];
    auto load_chunk = [&](int m0) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = m0 + lr + 8 * ps;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            c4[ps] = m < mend ? *reinterpret_cast<const float4*>(dC + (size_t)m * ldc + n0 + lc) : z;
            a4[ps] = m < mend ? *reinterpret_cast<const float4*>(A + (size_t)m * lda + k0 + lc) : z;
        }
    };
    load_chunk(mbeg);
    for (int m0 = mbeg; m0 < mend; m0 += 32, par ^= 1) {
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            *reinterpret_cast<float4*>(&Cs[lr + 8 * ps][lc]) = c4[ps];
            *reinterpret_cast<float4*>(&As[lr + 8 * ps][lc]) = a4[ps];
        }
        {   // this thread's 16 values lie in column block lc / 32
            bool nz = false;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) nz = nz || a4[ps].x != 0.f || a4[ps].y != 0.f || a4[ps].z != 0.f || a4[ps].w != 0.f;
            if (nz) nzblk[par][lc >> 5] = 1u;  // benign race: every writer stores 1
            if (tid < 4) nzblk[par ^ 1][tid] = 0u;  // the other parity's flags were last read before the barrier above
        }
        if (m0 + 32 < mend) load_chunk(m0 + 32);   // the next chunk's rows: in flight behind this chunk's MFMAs
        __syncthreads();
        if (colsum) {
#pragma unroll
            for (int r = 0; r < 32; ++r) bs += Cs[r][tid];  // rows in order: reproducible
        }
        const bool do0 = nzblk[par][kb] != 0u, do1 = nzblk[par][kb + 2] != 0u;  // wave-uniform
        if (do0 || do1) {
#pragma unroll
            for (int mm = 0; mm < 32; mm += 2) {
                const int rr = mm + (lane >> 5), cc = lane & 31;
                const float a0 = Cs[rr][wn + cc], a1 = Cs[rr][wn + 32 + cc];
                if (do0) {
                    const float b0 = As[rr][32 * kb + cc];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                }
                if (do1) {
                    const float b1 = As[rr][32 * (kb + 2) + cc];
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
        }
    }
    float* out = part + (size_t)split * N * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = k0 + 32 * (kb + 2 * j) + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(size_t)row * K + col] = acc[i][j][r];
            }
        }
    if (colsum) bpart[(size_t)split * N + n0 + tid] = bs;
}

// The same 128 x 128 contraction on the f16-rate matrix cores, fp32-class accuracy without any scaling pass: both operands
// are split into THREE bf16 terms (a = a1 + a2 + a3: 3 x 8 = 24 significant bits, the fp32 exponent range - weight-gradient
// operands span ~1e-9 ... 1e2 and are contracted along their ROWS, so the per-row lifts of the forward's f16x3 split do not
// apply and per-column lifts would cost one more pass over the 30 GB of d(rbfh)) and the six products of order <= 2^-16 are
// issued: a3 b1, a2 b2, a1 b3, a2 b1, a1 b2, a1 b1 (dropped: <= 2^-24 relative).  v_mfma_f32_32x32x16_bf16 runs at 16x the
// rate of v_mfma_f32_32x32x2_f32, so the six products cost 3/8 of the exact-f32 kernel's matrix time.  The chunk's rows
// are staged ROW-major as bf16 (what the coalesced loads give) and read with ds_read_b64_tr_b16 (gfx950's transposing LDS
// read: 4 rows x 16 columns per 16-lane group, delivered column-major = the K-contiguous operand layout of the 32x32x16
// instruction); image (b) of the CDNA4 guide's T10 (256-byte rows, 16-byte chunks XOR-swizzled) keeps both the 8-byte
// stores and the transposed reads conflict-free.  Zero-block skipping and the bias column sums as in tr_wgrad128_kernel.
typedef __fp16 tr_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __bf16 tr_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tr_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 tr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tr_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned int tr_swz(int row) { return (unsigned int)(((row & 3) << 2) | ((row >> 2) & 3)); }

__global__ __launch_bounds__(256) void tr_wgrad_bf16x6_kernel(const float* __restrict__ dC, int ldc, const float* __restrict__ A,
                                                              int lda, float* __restrict__ part, int M, int N, int K,
                                                              int rows_per_split, int tiles_k, float* __restrict__ bpart) {
    // [operand C / A][term 1..3][32 rows][256 B]
    __shared__ __attribute__((aligned(16))) unsigned char img[2][3][32 * 256];
    __shared__ float csum[8][128];
    __shared__ unsigned int nzblk[2][4];
    if (threadIdx.x < 8) nzblk[threadIdx.x >> 2][threadIdx.x & 3] = 0u;
    int tile = blockIdx.x, split = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        // all column tiles of a row split on ONE XCD (workgroups go to the XCDs round-robin in dispatch order): its L2 then
        // serves the split's A rows to the N / 128 tiles once - the rbf_proj call read them 12 x from HBM (29 of 31 GB)
        const int L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, j = L >> 3;
        split = xcd + 8 * (j / (int)gridDim.x);
        tile = j % (int)gridDim.x;
    }
    const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = (wave >> 1) * 64, kb = wave & 1;
    const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool colsum = bpart && k0 == 0;
    float bs = 0.f;
    const int lr = tid >> 5, lc = (tid & 31) * 4;  // staging: 8 rows x 32 float4 per pass, 4 passes
    // transposed-read addressing of this lane (see the header): 16-lane group g, lane 4q + p of the group
    const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int kg = g >> 1;            // k half of the MFMA operand (lanes 32..63)
    const int cgrp = g & 1;           // 16-column half of the 32-column block
    int par = 0;
    // The rows of chunk i + 1 are requested right after chunk i's rows have been converted into the LDS image, so they are in
    // flight while chunk i's MFMAs run (round 5; requested at the top of their own iteration they cost every 32-row chunk
    // a full memory round trip in front of its barrier).
    float4 c4[4], a4[4];
    auto load_chunk = [&](int m0) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = m0 + lr + 8 * ps;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            c4[ps] = m < mend ? *reinterpret_cast<const float4*>(dC + (size_t)m * ldc + n0 + lc) : z;
            a4[ps] = m < mend ? *reinterpret_cast<const float4*>(A + (size_t)m * lda + k0 + lc) : z;
        }
    };
    load_chunk(mbeg);
    for (int m0 = mbeg; m0 < mend; m0 += 32, par ^= 1) {
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = lr + 8 * ps;
            const unsigned int off = 256u * row + 16u * ((unsigned int)(lc >> 3) ^ tr_swz(row)) + 8u * ((lc >> 2) & 1);
#pragma unroll
            for (int op = 0; op < 2; ++op) {
                const float4 v = op == 0 ? c4[ps] : a4[ps];
                float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    // one packed conversion per pair, reused for the exact residuals (low half << 16, high half masked):
                    // 5.5 vector instructions per value instead of 7.5 - the staging is a third of this kernel's issue time
                    union { tr_bf16x4 b; unsigned int u[2]; } o;
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        union { tr_bf16x2 b; unsigned int u; } pk;
                        pk.b = __builtin_convertvector((tr_f32x2){r[e], r[e + 1]}, tr_bf16x2);
                        o.u[e >> 1] = pk.u;
                        if (t < 2) {
                            r[e] -= __uint_as_float(pk.u << 16);
                            r[e + 1] -= __uint_as_float(pk.u & 0xffff0000u);
                        }
                    }
                    *reinterpret_cast<tr_bf16x4*>(&img[op][t][off]) = o.b;
                }
            }
        }
        {
            bool nz = false;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) nz = nz || a4[ps].x != 0.f || a4[ps].y != 0.f || a4[ps].z != 0.f || a4[ps].w != 0.f;
            if (nz) nzblk[par][lc >> 5] = 1u;
            if (tid < 4) nzblk[par ^ 1][tid] = 0u;
        }
        if (colsum) {   // this thread's four rows of its four columns, rows in order; the 8 row groups are summed in order below
            float4 sum = c4[0];
#pragma unroll
            for (int ps = 1; ps < 4; ++ps) { sum.x += c4[ps].x; sum.y += c4[ps].y; sum.z += c4[ps].z; sum.w += c4[ps].w; }
            *reinterpret_cast<float4*>(&csum[lr][lc]) = sum;
        }
        if (m0 + 32 < mend) load_chunk(m0 + 32);   // the next chunk's rows: in flight behind this chunk's MFMAs
        __syncthreads();
        if (colsum && tid < 128) {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) bs += csum[r8][tid];
        }
        const bool do0 = nzblk[par][kb] != 0u, do1 = nzblk[par][kb + 2] != 0u;  // wave-uniform
        if (do0 || do1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // operand fragment: rows 16 ks + 8 kg + (0..7) of the 16 columns c16 .. c16 + 15 that hold this lane's column
                auto frag = [&](int op, int t, int c32) -> tr_bf16x8 {
                    const int col = c32 + 16 * cgrp + 4 * tp;           // first of the 4 columns this lane addresses
                    union { tr_fp16x4 h[2]; tr_bf16x8 b; } u;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int row = 16 * ks + 8 * kg + 4 * hh + tq;
                        const unsigned int off = 256u * row + 16u * ((unsigned int)(col >> 3) ^ tr_swz(row)) + 8u * ((col >> 2) & 1);
                        u.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) tr_fp16x4*)(&img[op][t][off]));
                    }
                    return u.b;
                };
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    tr_bf16x8 a[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) a[t] = frag(0, t, wn + 32 * i);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (j == 0 ? !do0 : !do1) continue;
                        tr_bf16x8 b[3];
#pragma unroll
                        for (int t = 0; t < 3; ++t) b[t] = frag(1, t, 32 * (kb + 2 * j));
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
                }
            }
        }
    }
    float* out = part + (size_t)split * N * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = k0 + 32 * (kb + 2 * j) + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(size_t)row * K + col] = acc[i][j][r];
            }
        }
    if (colsum && tid < 128) bpart[(size_t)split * N + n0 + tid] = bs;
}

// dst[i] (+)= sum_s part[s][i] in a fixed order (run-to-run reproducible gradients)
__global__ void tr_reduce_splits_kernel(const float* __restrict__ part, long long stride, float* __restrict__ dst,
                                        long long n, int splits, int accumulate) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part[(size_t)k * stride + i];
        dst[i] = accumulate ? dst[i] + s : s;
    }
}

// column sums of dC [M,N] (bias gradient): stage 1 per block of rows, stage 2 = tr_reduce_splits_kernel
__global__ void tr_colsum_kernel(const float* __restrict__ dC, int ldc, float* __restrict__ part, int M, int N,
                                 int rows_per_split) {
    const int split = blockIdx.y;
    const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < N; c += gridDim.x * blockDim.x) {
        // eight rows in flight (one dependent load per row was latency-bound: 0.8 TB/s on the [N, 3H] bias rows of the
        // fused message backward); the partial sums are combined in a fixed order
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int m = mbeg;
        for (; m + 8 <= mend; m += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += dC[(size_t)(m + k) * ldc + c];
        }
        for (; m < mend; ++m) s[0] += dC[(size_t)m * ldc + c];
        part[(size_t)split * N + c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
}

// dA[m,k] (+)= sum_n dC[m,n] W[n,k] for layers with a tiny output width (N <= 4: the last gated block)
__global__ void tr_dgrad_small_kernel(const float* __restrict__ dC, int ldc, const float* __restrict__ W,
                                      float* __restrict__ dA, int lda, long long M, int N, int K, int accumulate) {
    const long long total = M * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / K;
        const int k = (int)(i - m * K);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dC[(size_t)m * ldc + n] * W[(size_t)n * K + k];
        float* o = dA + (size_t)m * lda + k;
        *o = accumulate ? *o + s : s;
    }
}

__global__ void tr_add_rows_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, long long M,
                                   int C) {
    const long long total = M * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        dst[(size_t)m * ldd + c] += src[(size_t)m * lds_ + c];
    }
}

// y = A W^T + b for any K (zero-padded staging is not needed: the f32 GEMM wants K % 32 == 0; every layer of the model
// satisfies it) and any N.  Exported for the orchestrator.
extern "C" int32_t adf_op_linear_fwd(const float* A, int32_t lda, const float* W, const float* bias, float* C, int32_t ldc,
                                     int64_t M, int32_t N, int32_t K, void* stream) {
    if (!A || !W || !C || M < 0 || N <= 0 || K <= 0) { adf_set_error("linear_fwd: bad argument"); return ADF_EINVAL; }
    return tr_gemm(A, lda, W, bias, C, ldc, M, N, K, (hipStream_t)stream);
}

// Backward of y = A W^T + b.  dA (optional, [M,K] with row stride ldda) = dC W, written or accumulated;
// dW [N,K] and db [N] (optional) written or accumulated.  scratch: at least adf_op_linear_bwd_scratch(M,N,K) floats.
extern "C" int64_t adf_op_linear_bwd_scratch(int64_t M, int32_t N, int32_t K) {
    const int64_t splits = 64;
    return (int64_t)N * K + splits * ((int64_t)N * K + N) + (int64_t)M * K + 1024;
}

extern "C" int32_t adf_op_linear_bwd(const float* A, int32_t lda, const float* W, const float* dC, int32_t ldc, float* dA,
                                     int32_t ldda, int32_t acc_dA, float* dW, float* db, int32_t acc_dW, int64_t M,
                                     int32_t N, int32_t K, float* scratch, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!W || !dC || !scratch || M <= 0 || N <= 0 || K <= 0) { adf_set_error("linear_bwd: bad argument"); return ADF_EINVAL; }
    float* Wt = scratch;                                  // [K,N]
    float* part = scratch + (size_t)N * K;                // [splits][N*K + N]
    int splits = (int)((M + 1023) / 1024);
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    int rows = (int)((M + splits - 1) / splits);
    rows = (rows + 31) / 32 * 32;
    splits = (int)((M + rows - 1) / rows);
    if (splits >= 8) {   // a multiple of 8 row splits (tr_wgrad_bf16x6_kernel keeps a split's column tiles on one XCD)
        splits &= ~7;
        rows = (int)((M + splits - 1) / splits);
        rows = (rows + 31) / 32 * 32;   // trailing splits may be empty: they contribute zeros
    }
    if (dA) {
        if (N % 32 == 0) {
            hipLaunchKernelGGL(tr_transpose_kernel, dim3((K + 31) / 32, (N + 31) / 32), dim3(32, 8), 0, s, W, Wt, N, K);
            if (!acc_dA) {
                // dA = dC Wt^T through the same lifted f16x3 product (rows of dC are ~1e-6: the lift keeps both fp16 terms)
                ADF_TRY(tr_gemm(dC, ldc, Wt, nullptr, dA, ldda, M, K, N, s));
            } else if (tr_gemm16_ok(ldc, M, N) && (ldda & 3) == 0 && (K & 3) == 0 && (reinterpret_cast<uintptr_t>(dA) & 15) == 0) {
                // dA += dC Wt^T in the product's own epilogue (until round 5: a temporary [M, K] + an add pass, 2 ms per step)
                ADF_TRY(tr_gemm(dC, ldc, Wt, nullptr, dA, ldda, M, K, N, s, 1));
            } else {
                float* tmp = part + (size_t)64 * ((size_t)N * K + N);  // [M,K]
                ADF_TRY(tr_gemm(dC, ldc, Wt, nullptr, tmp, K, M, K, N, s));
                hipLaunchKernelGGL(tr_add_rows_kernel, dim3(tr_grid(M * K)), dim3(256), 0, s, tmp, K, dA, ldda, (long long)M, K);
            }
        } else if (N <= 4) {
            hipLaunchKernelGGL(tr_dgrad_small_kernel, dim3(tr_grid(M * K)), dim3(256), 0, s, dC, ldc, W, dA, ldda,
                               (long long)M, N, K, acc_dA);
        } else {
            adf_set_error("linear_bwd: output width %d must be a multiple of 32 or <= 4", N);
            return ADF_EINVAL;
        }
    }
    if (dW) {
        if (!A) { adf_set_error("linear_bwd: dW needs A"); return ADF_EINVAL; }
        float* bp = db ? part + (size_t)splits * N * K : (float*)nullptr;
        if (N % 128 == 0 && K % 128 == 0 && (lda & 3) == 0 && (ldc & 3) == 0 &&
            ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(dC)) & 15) == 0) {
            // ADF_WGRAD=f32: the exact-f32 matrix-core kernel (v_mfma_f32_32x32x2_f32); default: three-term bf16 split
            static const bool wgrad_f32 = [] { const char* e = getenv("ADF_WGRAD"); return e && strcmp(e, "f32") == 0; }();
            if (wgrad_f32)
                hipLaunchKernelGGL(tr_wgrad128_kernel, dim3((N / 128) * (K / 128), splits), dim3(256), 0, s, dC, ldc, A, lda, part,
                                   (int)M, N, K, rows, K / 128, bp);
            else
                hipLaunchKernelGGL(tr_wgrad_bf16x6_kernel, dim3((N / 128) * (K / 128), splits), dim3(256), 0, s, dC, ldc, A, lda,
                                   part, (int)M, N, K, rows, K / 128, bp);
        } else {
            const int tiles_n = (N + 63) / 64, tiles_k = (K + 63) / 64;
            hipLaunchKernelGGL(tr_wgrad_kernel, dim3(tiles_n * tiles_k, splits), dim3(256), 0, s, dC, ldc, A, lda, part, (int)M,
                               N, K, rows, tiles_k, bp);
        }
        hipLaunchKernelGGL(tr_reduce_splits_kernel, dim3(tr_grid((long long)N * K)), dim3(256), 0, s, part, (long long)N * K,
                           dW, (long long)N * K, splits, acc_dW);
    }
    if (db) {
        float* bpart = part + (size_t)splits * N * K;
        if (!dW)  // no weight-gradient pass to ride on
            hipLaunchKernelGGL(tr_colsum_kernel, dim3((N + 255) / 256, splits), dim3(256), 0, s, dC, ldc, bpart, (int)M, N, rows);
        hipLaunchKernelGGL(tr_reduce_splits_kernel, dim3(tr_grid(N)), dim3(256), 0, s, bpart, (long long)N, db, (long long)N,
                           splits, acc_dW);
    }
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ elementwise
__global__ void tr_ssilu_fwd_kernel(const float* __restrict__ h, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = tr_ssilu(h[i]);
}
__global__ void tr_ssilu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dy, float* __restrict__ dh,
                                    long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dh[i] = dy[i] * tr_dssilu(h[i]);
}
extern "C" int32_t adf_op_ssilu_fwd(const float* h, float* y, int64_t n, void* stream) {
    hipLaunchKernelGGL(tr_ssilu_fwd_kernel, dim3(tr_grid(n)), dim3(256), 0, (hipStream_t)stream, h, y, (long long)n);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
extern "C" int32_t adf_op_ssilu_bwd(const float* h, const float* dy, float* dh, int64_t n, void* stream) {
    hipLaunchKernelGGL(tr_ssilu_bwd_kernel, dim3(tr_grid(n)), dim3(256), 0, (hipStream_t)stream, h, dy, dh, (long long)n);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// LayerNorm (eps 1e-5, biased variance), one wave per row; stats[row] = (mean, rstd)
__global__ __launch_bounds__(256) void tr_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y,
                                                         float2* __restrict__ stats, int N, int H) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* xr = x + (size_t)row * H;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s += xr[c];
    const float mean = tr_wsum(s) / (float)H;
    float q = 0.f;
    for (int c = lane; c < H; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(tr_wsum(q) / (float)H + 1e-5f);
    for (int c = lane; c < H; c += 64) y[(size_t)row * H + c] = (xr[c] - mean) * rstd * w[c] + b[c];
    if (lane == 0) stats[row] = make_float2(mean, rstd);
}
// dx += rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w;  per-row partial sums of dw, db go to part[blocks][2H]
__global__ __launch_bounds__(256) void tr_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float2* __restrict__ stats, const float* __restrict__ dy,
                                                         float* __restrict__ dx, float* __restrict__ part, int N, int H,
                                                         int rows_per_block) {
    extern __shared__ float sh[];  // [4 waves][2H]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = sh + wave * 2 * H;
    for (int c = lane; c < 2 * H; c += 64) mine[c] = 0.f;
    const int r0 = blockIdx.x * rows_per_block;
    for (int row = r0 + wave; row < min(N, r0 + rows_per_block); row += 4) {
        const float2 st = stats[row];
        const float* xr = x + (size_t)row * H;
        const float* gr = dy + (size_t)row * H;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < H; c += 64) {
            const float xh = (xr[c] - st.x) * st.y, g = gr[c] * w[c];
            s1 += g; s2 += g * xh;
            mine[c] += gr[c] * xh;   // dw
            mine[H + c] += gr[c];    // db
        }
        s1 = tr_wsum(s1) / (float)H; s2 = tr_wsum(s2) / (float)H;
        for (int c = lane; c < H; c += 64) {
            const float xh = (xr[c] - st.x) * st.y, g = gr[c] * w[c];
            dx[(size_t)row * H + c] += st.y * (g - s1 - xh * s2);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 256)
        part[(size_t)blockIdx.x * 2 * H + c] = sh[c] + sh[2 * H + c] + sh[4 * H + c] + sh[6 * H + c];
}
extern "C" int32_t adf_op_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* stats, int32_t N,
                                        int32_t H, void* stream) {
    hipLaunchKernelGGL(tr_ln_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, b, y,
                       reinterpret_cast<float2*>(stats), N, H);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
// dx is ACCUMULATED; dw, db [H] written.  scratch: 512 * 2H floats (one [dw | db] partial row per workgroup; 64 workgroups
// - a quarter of the chip - made this kernel 1.26 ms at N = 51 200, H = 512: 4.6 % of a training step).
extern "C" int32_t adf_op_layernorm_bwd(const float* x, const float* w, const float* stats, const float* dy, float* dx,
                                        float* dw, float* db, int32_t N, int32_t H, float* scratch, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int blocks = (N + 63) / 64;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    const int rows = (N + blocks - 1) / blocks;
    hipLaunchKernelGGL(tr_ln_bwd_kernel, dim3(blocks), dim3(256), sizeof(float) * 8 * H, s, x, w,
                       reinterpret_cast<const float2*>(stats), dy, dx, scratch, N, H, rows);
    // scratch rows are [dw | db]; reduce over blocks
    hipLaunchKernelGGL(tr_reduce_splits_kernel, dim3(tr_grid(H)), dim3(256), 0, s, scratch, (long long)2 * H, dw, (long long)H,
                       blocks, 0);
    hipLaunchKernelGGL(tr_reduce_splits_kernel, dim3(tr_grid(H)), dim3(256), 0, s, scratch + H, (long long)2 * H, db,
                       (long long)H, blocks, 0);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// embedding: x[n] = emb[Z[n]-1];  backward: demb[Z-1] += dx[n] (float atomics: 83 rows, many atoms per row)
__global__ void tr_embed_bwd_kernel(const float* __restrict__ dx, const int32_t* __restrict__ Z, float* __restrict__ demb,
                                    long long N, int H) {
    const long long total = N * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / H;
        const int c = (int)(i - n * H);
        atomicAdd(&demb[(size_t)(Z[n] - 1) * H + c], dx[i]);
    }
}
extern "C" int32_t adf_op_embed_fwd(adf_painn_t h, const int32_t* Z, int32_t N, float* x, void* stream) {
    if (!h || !h->weights_set) { adf_set_error("embed: weights not set"); return ADF_EINVAL; }
    return adf_nodewise_embed(h, Z, N, x, (hipStream_t)stream);
}
extern "C" int32_t adf_op_embed_bwd(const float* dx, const int32_t* Z, float* demb, int32_t N, int32_t H, void* stream) {
    hipLaunchKernelGGL(tr_embed_bwd_kernel, dim3(tr_grid((long long)N * H)), dim3(256), 0, (hipStream_t)stream, dx, Z, demb,
                       (long long)N, H);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ message block
// rbf[e,k] = env(d/rc) exp(-(d/rc - mu_k)^2 / (2 sigma^2))   (radial_basis.py:18-43,64-82)
__global__ void tr_rbf_kernel(const float4* __restrict__ e_geom, const int32_t* __restrict__ nptr, int N,
                              const float* __restrict__ mu, int R, float inv_cutoff, float coeff, float env_a, float env_b,
                              float env_c, int env_pi, float* __restrict__ rbf) {
    const long long E = nptr[N];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < E * R; t += (long long)gridDim.x * blockDim.x) {
        const long long e = t / R;
        const int k = (int)(t - e * R);
        const float xs = e_geom[e].w * inv_cutoff;
        float xp = xs;
        for (int i = 1; i < env_pi; ++i) xp *= xs;
        float env = 1.0f + env_a * xp + env_b * (xp * xs) + env_c * (xp * xs * xs);
        env = xs < 1.0f ? env : 0.0f;
        const float dm = xs - mu[k];
        rbf[t] = env * expf(coeff * dm * dm);
    }
}
extern "C" int32_t adf_op_rbf(adf_painn_t h, float* rbf, void* stream) {
    if (!h || h->lastN <= 0 || !h->weights_set) { adf_set_error("rbf: no graph / weights"); return ADF_EINVAL; }
    const int R = h->hp.num_rbf;
    const double step = 1.0 / (R - 1), pe = (double)h->hp.envelope_exponent;
    hipLaunchKernelGGL(tr_rbf_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, h->e_geom, h->nptr, (int)h->lastN,
                       h->rbf_offset, R, 1.0f / h->hp.cutoff, (float)(-0.5 / (step * step)),
                       (float)(-(pe + 1) * (pe + 2) / 2), (float)(pe * (pe + 2)), (float)(-pe * (pe + 1) / 2),
                       h->hp.envelope_exponent, rbf);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// forward: one 256-thread block per target atom i, channels strided over the threads
//   dx[i,c] = sum_e xa[j,c] ra[e,c];  dvec[i,k,c] = sum_e (vec[j,k,c] xb[j,c] rb[e,c] / sqrt3 + xc[j,c] rc[e,c] u_k) / sqrtH
//   x1 = (x + dx)/sqrt2;  vec1 = vec + dvec          (painn_denoising.py:443-445, 549-566)
__global__ __launch_bounds__(256) void tr_msg_fwd_kernel(const int32_t* __restrict__ nptr, const int32_t* __restrict__ e_src,
                                                          const float4* __restrict__ e_geom, const float* __restrict__ xh,
                                                          const float* __restrict__ vec, const float* __restrict__ rbfh,
                                                          const float* __restrict__ x, float* __restrict__ x1,
                                                          float* __restrict__ vec1, int H, int vec_is_zero) {
    const int i = blockIdx.x;
    const int e0 = nptr[i], e1 = nptr[i + 1];
    const float is3 = 0.57735026918962576f, ish = 1.0f / sqrtf((float)H), is2 = 0.70710678118654752f;
    for (int c = threadIdx.x; c < H; c += 256) {
        float sx = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f;
        for (int e = e0; e < e1; ++e) {
            const int j = e_src[e];
            const float4 g = e_geom[e];
            const float* xr = xh + (size_t)j * 3 * H;
            const float* rr = rbfh + (size_t)e * 3 * H;
            sx += xr[c] * rr[c];
            const float cb = xr[H + c] * rr[H + c] * is3;
            const float cc = xr[2 * H + c] * rr[2 * H + c];
            if (!vec_is_zero) {
                const float* vr = vec + (size_t)j * 3 * H;
                v0 += vr[c] * cb; v1 += vr[H + c] * cb; v2 += vr[2 * H + c] * cb;
            }
            v0 += cc * g.x; v1 += cc * g.y; v2 += cc * g.z;
        }
        x1[(size_t)i * H + c] = (x[(size_t)i * H + c] + sx) * is2;
        const size_t vo = (size_t)i * 3 * H + c;
        const float b0 = vec_is_zero ? 0.f : vec[vo], b1 = vec_is_zero ? 0.f : vec[vo + H], b2 = vec_is_zero ? 0.f : vec[vo + 2 * H];
        vec1[vo] = b0 + v0 * ish; vec1[vo + H] = b1 + v1 * ish; vec1[vo + 2 * H] = b2 + v2 * ish;
    }
}
// backward: one block per SOURCE atom j over its own CSR segment (see the header).  gx1, gv1 = gradients of x1, vec1.
//   writes dxh[j] [3H], drbfh[e'] [3H] for e' in CSR(j); dvec[j] = gv1[j] + sum ... (written; vec_is_zero: skipped)
//   and dx[j] = gx1[j] / sqrt2 (the residual path), written.
__global__ __launch_bounds__(256) void tr_msg_bwd_kernel(const int32_t* __restrict__ nptr, const int32_t* __restrict__ e_src,
                                                          const float4* __restrict__ e_geom, const float* __restrict__ xh,
                                                          const float* __restrict__ vec, const float* __restrict__ rbfh,
                                                          const float* __restrict__ gx1, const float* __restrict__ gv1,
                                                          float* __restrict__ dxh, float* __restrict__ drbfh,
                                                          float* __restrict__ dvec, float* __restrict__ dx, int H,
                                                          int vec_is_zero) {
    const int j = blockIdx.x;
    const int e0 = nptr[j], e1 = nptr[j + 1];
    const float is3 = 0.57735026918962576f, ish = 1.0f / sqrtf((float)H), is2 = 0.70710678118654752f;
    for (int c = threadIdx.x; c < H; c += 256) {
        const float* xr = xh + (size_t)j * 3 * H;
        const float xa = xr[c], xb = xr[H + c], xc = xr[2 * H + c];
        float w0 = 0.f, w1 = 0.f, w2 = 0.f;
        if (!vec_is_zero) { const float* vr = vec + (size_t)j * 3 * H; w0 = vr[c]; w1 = vr[H + c]; w2 = vr[2 * H + c]; }
        float dxa = 0.f, dxb = 0.f, dxc = 0.f, dv0 = 0.f, dv1 = 0.f, dv2 = 0.f;
        for (int e = e0; e < e1; ++e) {
            const int i = e_src[e];            // the neighbour: target of the reverse edge (j -> i)
            const float4 g = e_geom[e];        // unit vector j -> i; the reverse edge's is its negative
            const float* rr = rbfh + (size_t)e * 3 * H;
            const float ra = rr[c], rb = rr[H + c], rc = rr[2 * H + c];
            const float gx = gx1[(size_t)i * H + c] * is2;
            const size_t vo = (size_t)i * 3 * H + c;
            const float g0 = gv1[vo] * ish, g1 = gv1[vo + H] * ish, g2 = gv1[vo + 2 * H] * ish;
            const float S = (g0 * w0 + g1 * w1 + g2 * w2) * is3;
            const float T = -(g0 * g.x + g1 * g.y + g2 * g.z);
            dxa += gx * ra; dxb += S * rb; dxc += T * rc;
            const float f = xb * rb * is3;
            dv0 += g0 * f; dv1 += g1 * f; dv2 += g2 * f;
            float* dr = drbfh + (size_t)e * 3 * H;
            dr[c] = gx * xa; dr[H + c] = S * xb; dr[2 * H + c] = T * xc;
        }
        float* dh = dxh + (size_t)j * 3 * H;
        dh[c] = dxa; dh[H + c] = dxb; dh[2 * H + c] = dxc;
        dx[(size_t)j * H + c] = gx1[(size_t)j * H + c] * is2;
        if (!vec_is_zero) {
            const size_t vo = (size_t)j * 3 * H + c;
            dvec[vo] = gv1[vo] + dv0; dvec[vo + H] = gv1[vo + H] + dv1; dvec[vo + 2 * H] = gv1[vo + 2 * H] + dv2;
        }
    }
}
extern "C" int32_t adf_op_message_fwd(adf_painn_t h, const float* xh, const float* vec, const float* rbfh, const float* x,
                                      float* x1, float* vec1, int32_t vec_is_zero, void* stream) {
    if (!h || h->lastN <= 0) { adf_set_error("message_fwd: no graph"); return ADF_EINVAL; }
    hipLaunchKernelGGL(tr_msg_fwd_kernel, dim3((unsigned)h->lastN), dim3(256), 0, (hipStream_t)stream, h->nptr, h->e_src,
                       h->e_geom, xh, vec, rbfh, x, x1, vec1, h->hp.hidden_channels, vec_is_zero);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
// The same forward through the SAMPLER's fused kernel (message.hip: radial-basis projection on the matrix cores inside the
// kernel, no [E,3H] read): the layer's rbf_proj images are rebuilt from the current parameter values first.  The backward
// still reads the materialised rbfh (adf_op_linear_fwd of the training step writes it for that purpose).
extern "C" int32_t adf_op_message_fwd_fused(adf_painn_t h, int32_t layer, const float* xh, const float* vec, const float* x,
                                            float* x1, float* vec1, int32_t vec_is_zero, void* stream) {
    if (!h || h->lastN <= 0 || !h->weights_set || layer < 0 || layer >= h->hp.num_layers || !xh || !x || !x1 || !vec1 ||
        (!vec && !vec_is_zero)) {
        adf_set_error("message_fwd_fused: bad argument or no graph");
        return ADF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const int N = (int)h->lastN, H = h->hp.hidden_channels;
    const size_t row = (size_t)(H / 32) * 160;
    ADF_TRY(adf_pack_rbf_layer(h, layer, s));
    ADF_HIP_CHECK(hipMemsetAsync(h->rec + (size_t)N * row, 0, sizeof(float) * row, s));   // padded edge rows gather record N
    ADF_TRY(adf_pack_records(h, N, xh, vec, vec_is_zero != 0, s));
    return adf_message_impl(h, layer, N, x, xh, vec, x1, vec1, vec_is_zero != 0, s);
}
extern "C" int32_t adf_op_message_bwd(adf_painn_t h, const float* xh, const float* vec, const float* rbfh, const float* gx1,
                                      const float* gv1, float* dxh, float* drbfh, float* dvec, float* dx,
                                      int32_t vec_is_zero, void* stream) {
    if (!h || h->lastN <= 0) { adf_set_error("message_bwd: no graph"); return ADF_EINVAL; }
    hipLaunchKernelGGL(tr_msg_bwd_kernel, dim3((unsigned)h->lastN), dim3(256), 0, (hipStream_t)stream, h->nptr, h->e_src,
                       h->e_geom, xh, vec, rbfh, gx1, gv1, dxh, drbfh, dvec, dx, h->hp.hidden_channels, vec_is_zero);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ update block
// vv [N,3,2C] = (v1 | v2);  dot = sum_k v1 v2 / sqrtC;  nrm = sqrt(sum_k v2^2 + eps)   (painn_denoising.py:604-611)
// nrm is written with row stride ldn (straight into the right half of the [x | nrm] MLP input)
__global__ void tr_vdot_fwd_kernel(const float* __restrict__ vv, float* __restrict__ dot, float* __restrict__ nrm, int ldn,
                                   long long N, int C, float eps) {
    const float isc = 1.0f / sqrtf((float)C);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N * C; t += (long long)gridDim.x * blockDim.x) {
        const long long n = t / C;
        const int c = (int)(t - n * C);
        float d = 0.f, q = 0.f;
        for (int k = 0; k < 3; ++k) {
            const float a = vv[((size_t)n * 3 + k) * 2 * C + c], b = vv[((size_t)n * 3 + k) * 2 * C + C + c];
            d += a * b; q += b * b;
        }
        dot[t] = d * isc;
        nrm[(size_t)n * ldn + c] = sqrtf(q + eps);
    }
}
// dvv from (ddot, dnrm [stride ldn]) plus an optional direct gradient of v1 (dv1, [N,3,C], may be null)
__global__ void tr_vdot_bwd_kernel(const float* __restrict__ vv, const float* __restrict__ nrm, int ldn,
                                   const float* __restrict__ ddot, const float* __restrict__ dnrm, int lddn,
                                   const float* __restrict__ dv1, float* __restrict__ dvv, long long N, int C) {
    const float isc = 1.0f / sqrtf((float)C);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N * C; t += (long long)gridDim.x * blockDim.x) {
        const long long n = t / C;
        const int c = (int)(t - n * C);
        const float dd = ddot ? ddot[t] * isc : 0.f;
        const float dn = dnrm[(size_t)n * lddn + c] / nrm[(size_t)n * ldn + c];
        for (int k = 0; k < 3; ++k) {
            const size_t o = ((size_t)n * 3 + k) * 2 * C + c;
            const float a = vv[o], b = vv[o + C];
            dvv[o] = dd * b + (dv1 ? dv1[((size_t)n * 3 + k) * C + c] : 0.f);
            dvv[o + C] = dd * a + dn * b;
        }
    }
}
extern "C" int32_t adf_op_vdot_fwd(const float* vv, float* dot, float* nrm, int32_t ldn, int64_t N, int32_t C, float eps,
                                   void* stream) {
    hipLaunchKernelGGL(tr_vdot_fwd_kernel, dim3(tr_grid(N * C)), dim3(256), 0, (hipStream_t)stream, vv, dot, nrm, ldn,
                       (long long)N, C, eps);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
extern "C" int32_t adf_op_vdot_bwd(const float* vv, const float* nrm, int32_t ldn, const float* ddot, const float* dnrm,
                                   int32_t lddn, const float* dv1, float* dvv, int64_t N, int32_t C, void* stream) {
    hipLaunchKernelGGL(tr_vdot_bwd_kernel, dim3(tr_grid(N * C)), dim3(256), 0, (hipStream_t)stream, vv, nrm, ldn, ddot, dnrm,
                       lddn, dv1, dvv, (long long)N, C);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// x2 = (x1 + (a1 + a2 dot)/sqrt2) s ;  vec2 = vec1 + a3 (x) v1     (painn_denoising.py:614-623, 449-451)
__global__ void tr_upd_out_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ vec1,
                                      const float* __restrict__ a, const float* __restrict__ dot, const float* __restrict__ vv,
                                      float s, float* __restrict__ x2, float* __restrict__ vec2, long long N, int H) {
    const float is2 = 0.70710678118654752f;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N * H; t += (long long)gridDim.x * blockDim.x) {
        const long long n = t / H;
        const int c = (int)(t - n * H);
        const float* ar = a + (size_t)n * 3 * H;
        x2[t] = (x1[t] + (ar[c] + ar[H + c] * dot[t]) * is2) * s;
        for (int k = 0; k < 3; ++k) {
            const size_t o = ((size_t)n * 3 + k) * H + c;
            vec2[o] = vec1[o] + ar[2 * H + c] * vv[((size_t)n * 3 + k) * 2 * H + c];
        }
    }
}
// given dx2, dvec2: da [N,3H], ddot [N,H], dv1 [N,3,H] written; dx1 = dx2 s and dvec1 = dvec2 written
__global__ void tr_upd_out_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dot, const float* __restrict__ vv,
                                      float s, const float* __restrict__ dx2, const float* __restrict__ dvec2,
                                      float* __restrict__ da, float* __restrict__ ddot, float* __restrict__ dv1,
                                      float* __restrict__ dx1, float* __restrict__ dvec1, long long N, int H) {
    const float is2 = 0.70710678118654752f;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N * H; t += (long long)gridDim.x * blockDim.x) {
        const long long n = t / H;
        const int c = (int)(t - n * H);
        const float* ar = a + (size_t)n * 3 * H;
        const float g = dx2[t] * s;
        float* dr = da + (size_t)n * 3 * H;
        dr[c] = g * is2;
        dr[H + c] = g * dot[t] * is2;
        ddot[t] = g * ar[H + c] * is2;
        dx1[t] = g;
        float d3 = 0.f;
        for (int k = 0; k < 3; ++k) {
            const size_t o = ((size_t)n * 3 + k) * H + c;
            const float gv = dvec2[o];
            d3 += gv * vv[((size_t)n * 3 + k) * 2 * H + c];
            dv1[o] = gv * ar[2 * H + c];
            dvec1[o] = gv;
        }
        dr[2 * H + c] = d3;
    }
}
extern "C" int32_t adf_op_update_out_fwd(const float* x1, const float* vec1, const float* a, const float* dot, const float* vv,
                                         float s, float* x2, float* vec2, int64_t N, int32_t H, void* stream) {
    hipLaunchKernelGGL(tr_upd_out_fwd_kernel, dim3(tr_grid(N * H)), dim3(256), 0, (hipStream_t)stream, x1, vec1, a, dot, vv, s,
                       x2, vec2, (long long)N, H);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
extern "C" int32_t adf_op_update_out_bwd(const float* a, const float* dot, const float* vv, float s, const float* dx2,
                                         const float* dvec2, float* da, float* ddot, float* dv1, float* dx1, float* dvec1,
                                         int64_t N, int32_t H, void* stream) {
    hipLaunchKernelGGL(tr_upd_out_bwd_kernel, dim3(tr_grid(N * H)), dim3(256), 0, (hipStream_t)stream, a, dot, vv, s, dx2,
                       dvec2, da, ddot, dv1, dx1, dvec1, (long long)N, H);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ gated equivariant block
// nrm[n,c] = sqrt(sum_k t1[n,k,c]^2)  (torch.norm, no eps; painn_denoising.py:690), written with row stride ldn
__global__ void tr_vnorm_fwd_kernel(const float* __restrict__ t1, float* __restrict__ nrm, int ldn, long long N, int C) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N * C; t += (long long)gridDim.x * blockDim.x) {
        const long long n = t / C;
        const int c = (int)(t - n * C);
        float q = 0.f;
        for (int k = 0; k < 3; ++k) { const float a = t1[((size_t)n * 3 + k) * C + c]; q += a * a; }
        nrm[(size_t)n * ldn + c] = sqrtf(q);
    }
}
__global__ void tr_vnorm_bwd_kernel(const float* __restrict__ t1, const float* __restrict__ nrm, int ldn,
                                    const float* __restrict__ dnrm, int lddn, float* __restrict__ dt1, long long N, int C) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N * C; t += (long long)gridDim.x * blockDim.x) {
        const long long n = t / C;
        const int c = (int)(t - n * C);
        const float nv = nrm[(size_t)n * ldn + c];
        const float f = nv > 0.f ? dnrm[(size_t)n * lddn + c] / nv : 0.f;  // torch: subgradient 0 at the origin
        for (int k = 0; k < 3; ++k) { const size_t o = ((size_t)n * 3 + k) * C + c; dt1[o] = f * t1[o]; }
    }
}
// o [N,2C] = (xo | gate);  xs = ssilu(xo) [stride ldx];  vout[n,k,c] = gate[n,c] t2[n,k,c]    (painn_denoising.py:693-696)
__global__ void tr_gate_fwd_kernel(const float* __restrict__ o, const float* __restrict__ t2, float* __restrict__ xs, int ldx,
                                   float* __restrict__ vout, long long N, int C) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N * C; t += (long long)gridDim.x * blockDim.x) {
        const long long n = t / C;
        const int c = (int)(t - n * C);
        const float xo = o[(size_t)n * 2 * C + c], g = o[(size_t)n * 2 * C + C + c];
        if (xs) xs[(size_t)n * ldx + c] = tr_ssilu(xo);
        for (int k = 0; k < 3; ++k) { const size_t q = ((size_t)n * 3 + k) * C + c; vout[q] = g * t2[q]; }
    }
}
// do [N,2C] and dt2 [N,3,C] from dxs (stride lddx, may be null) and dvout
__global__ void tr_gate_bwd_kernel(const float* __restrict__ o, const float* __restrict__ t2, const float* __restrict__ dxs,
                                   int lddx, const float* __restrict__ dvout, float* __restrict__ d_o, float* __restrict__ dt2,
                                   long long N, int C) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < N * C; t += (long long)gridDim.x * blockDim.x) {
        const long long n = t / C;
        const int c = (int)(t - n * C);
        const float xo = o[(size_t)n * 2 * C + c], g = o[(size_t)n * 2 * C + C + c];
        float dg = 0.f;
        for (int k = 0; k < 3; ++k) {
            const size_t q = ((size_t)n * 3 + k) * C + c;
            dg += dvout[q] * t2[q];
            dt2[q] = dvout[q] * g;
        }
        d_o[(size_t)n * 2 * C + c] = dxs ? dxs[(size_t)n * lddx + c] * tr_dssilu(xo) : 0.f;
        d_o[(size_t)n * 2 * C + C + c] = dg;
    }
}
extern "C" int32_t adf_op_vnorm_fwd(const float* t1, float* nrm, int32_t ldn, int64_t N, int32_t C, void* stream) {
    hipLaunchKernelGGL(tr_vnorm_fwd_kernel, dim3(tr_grid(N * C)), dim3(256), 0, (hipStream_t)stream, t1, nrm, ldn, (long long)N, C);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
extern "C" int32_t adf_op_vnorm_bwd(const float* t1, const float* nrm, int32_t ldn, const float* dnrm, int32_t lddn,
                                    float* dt1, int64_t N, int32_t C, void* stream) {
    hipLaunchKernelGGL(tr_vnorm_bwd_kernel, dim3(tr_grid(N * C)), dim3(256), 0, (hipStream_t)stream, t1, nrm, ldn, dnrm, lddn,
                       dt1, (long long)N, C);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
extern "C" int32_t adf_op_gate_fwd(const float* o, const float* t2, float* xs, int32_t ldx, float* vout, int64_t N, int32_t C,
                                   void* stream) {
    hipLaunchKernelGGL(tr_gate_fwd_kernel, dim3(tr_grid(N * C)), dim3(256), 0, (hipStream_t)stream, o, t2, xs, ldx, vout,
                       (long long)N, C);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
extern "C" int32_t adf_op_gate_bwd(const float* o, const float* t2, const float* dxs, int32_t lddx, const float* dvout,
                                   float* d_o, float* dt2, int64_t N, int32_t C, void* stream) {
    hipLaunchKernelGGL(tr_gate_bwd_kernel, dim3(tr_grid(N * C)), dim3(256), 0, (hipStream_t)stream, o, t2, dxs, lddx, dvout,
                       d_o, dt2, (long long)N, C);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// strided copy / accumulate of a [M,C] block (the [x | norm] MLP inputs and their gradients)
__global__ void tr_copy_rows_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, long long M,
                                    int C) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M * C; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        dst[(size_t)m * ldd + c] = src[(size_t)m * lds_ + c];
    }
}
extern "C" int32_t adf_op_copy_rows(const float* src, int32_t lds_, float* dst, int32_t ldd, int64_t M, int32_t C,
                                    int32_t accumulate, void* stream) {
    if (accumulate)
        hipLaunchKernelGGL(tr_add_rows_kernel, dim3(tr_grid(M * C)), dim3(256), 0, (hipStream_t)stream, src, lds_, dst, ldd,
                           (long long)M, C);
    else
        hipLaunchKernelGGL(tr_copy_rows_kernel, dim3(tr_grid(M * C)), dim3(256), 0, (hipStream_t)stream, src, lds_, dst, ldd,
                           (long long)M, C);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ loss
// DenoisingTrainer._compute_loss with so3_denoising (sde_denoising_trainer.py:675-728), one wave per system:
//   p = mean_ads(f1) / sigma_tr, p_z = 0;   L_tr  = mean_{b,k} (p - s_tr)^2 sigma_tr^2
//   r = mean_ads(f2) / sigma_rot;           L_rot = mean_{b,k} ((r - s_rot) / norm)^2
// loss_part[b] = this system's share of (L_tr, L_rot); df1, df2 [N,3] written (zero off the adsorbate).
__global__ __launch_bounds__(64) void tr_loss_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                     const int32_t* __restrict__ tags, const int32_t* __restrict__ atom_offset,
                                                     const float* __restrict__ tr_sigma, const float* __restrict__ rot_sigma,
                                                     const float* __restrict__ tr_score, const float* __restrict__ rot_score,
                                                     const float* __restrict__ rot_norm, float* __restrict__ loss_part,
                                                     float* __restrict__ df1, float* __restrict__ df2, int B) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int a0 = atom_offset[b], a1 = atom_offset[b + 1];
    float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = a0 + lane; a < a1; a += 64)
        if (tags[a] == 2) {
            for (int k = 0; k < 3; ++k) { s[k] += f1[3 * a + k]; s[3 + k] += f2[3 * a + k]; }
            s[6] += 1.f;
        }
    for (int i = 0; i < 7; ++i) s[i] = tr_wsum(s[i]);
    const float cnt = fmaxf(s[6], 1.f), st = tr_sigma[b], sr = rot_sigma[b], nr = rot_norm[b];
    const float inv = 1.0f / (3.0f * (float)B);
    float lt = 0.f, lr = 0.f, g1[3], g2[3];
    for (int k = 0; k < 3; ++k) {
        const float p = k < 2 ? s[k] / cnt / st : 0.f;
        const float d = p - tr_score[3 * b + k];
        lt += d * d * st * st;
        g1[k] = k < 2 ? 2.f * d * st * st * inv / (st * cnt) : 0.f;
        const float r = s[3 + k] / cnt / sr;
        const float q = (r - rot_score[3 * b + k]) / nr;
        lr += q * q;
        g2[k] = 2.f * q * inv / (nr * sr * cnt);
    }
    if (lane == 0) { loss_part[2 * b] = lt * inv; loss_part[2 * b + 1] = lr * inv; }
    for (int a = a0 + lane; a < a1; a += 64) {
        const bool ads = tags[a] == 2;
        for (int k = 0; k < 3; ++k) { df1[3 * a + k] = ads ? g1[k] : 0.f; df2[3 * a + k] = ads ? g2[k] : 0.f; }
    }
}
__global__ void tr_loss_sum_kernel(const float* __restrict__ part, float* __restrict__ loss, int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float lt = 0.f, lr = 0.f;
        for (int b = 0; b < B; ++b) { lt += part[2 * b]; lr += part[2 * b + 1]; }
        loss[0] = lt + lr; loss[1] = lt; loss[2] = lr;
    }
}
// loss [3] = (total, translation term, rotation term); scratch: 2B floats
extern "C" int32_t adf_op_score_loss(const float* f1, const float* f2, const int32_t* tags, const int32_t* atom_offset,
                                     const float* tr_sigma, const float* rot_sigma, const float* tr_score,
                                     const float* rot_score, const float* rot_norm, float* loss, float* df1, float* df2,
                                     int32_t B, float* scratch, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tr_loss_kernel, dim3(B), dim3(64), 0, s, f1, f2, tags, atom_offset, tr_sigma, rot_sigma, tr_score,
                       rot_score, rot_norm, scratch, df1, df2, B);
    hipLaunchKernelGGL(tr_loss_sum_kernel, dim3(1), dim3(64), 0, s, scratch, loss, B);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ optimizer
// One fused pass per parameter tensor: global-norm clip factor (from a device scalar), AdamW (torch.optim.AdamW
// semantics: decoupled weight decay, bias-corrected moments) and the EMA shadow update (base_trainer.py:803-820,
// modules/exponential_moving_average.py:71-97).
__global__ void tr_sqnorm_kernel(const float* __restrict__ g, long long n, float* __restrict__ out) {
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        s += g[i] * g[i];
    s = tr_wsum(s);
    // one atomic per workgroup (one per wave from up to 4096 workgroups serialised on the single address: 44 us per call)
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (ws[0] + ws[1]) + (ws[2] + ws[3]));
}
extern "C" int32_t adf_op_sqnorm_accumulate(const float* g, int64_t n, float* out, void* stream) {
    unsigned blocks = tr_grid(n, 2048);   // 8 elements per thread
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(tr_sqnorm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, (long long)n, out);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
__global__ void tr_adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                float* __restrict__ ema, long long n, const float* __restrict__ sqnorm, float max_norm,
                                float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                float ema_decay) {
    // A non-finite global gradient norm (NaN / Inf anywhere in the gradients, e.g. an fp16-range overflow in the f16x3
    // forward) makes the whole update a no-op: parameters, moments and the EMA shadow keep their values.  The host also
    // skips the step on a non-finite loss like the reference (sde_denoising_trainer.py:425-431); this is the guard that
    // needs no host round trip and that holds on every rank after the gradient all-reduce.
    const float sq = *sqnorm;
    if (!(sq == sq) || sq > 3.0e38f) return;
    float clip = 1.0f;
    if (max_norm > 0.f) {  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
        const float c = max_norm / (sqrtf(*sqnorm) + 1e-6f);
        clip = c < 1.0f ? c : 1.0f;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        float pi = p[i];
        pi *= 1.0f - lr * wd;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        pi -= lr / bc1 * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
        p[i] = pi;
        if (ema) ema[i] += (1.0f - ema_decay) * (pi - ema[i]);
    }
}
extern "C" int32_t adf_op_adamw_step(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* sqnorm,
                                     float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                     int32_t step, float ema_decay, void* stream) {
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(tr_adamw_kernel, dim3(tr_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, (long long)n,
                       sqnorm, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2, ema_decay);
    TR_CHECK_LAUNCH();
    return ADF_OK;
}
