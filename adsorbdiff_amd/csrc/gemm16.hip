// Dense per-node linear blocks on the f16 matrix cores with a 3-product split ("f16x3"):
//
//   a = a_hi + a_lo,  w·s = w_hi + w_lo   (a_hi = fp16(a), a_lo = fp16(a - a_hi), same for w·s)
//   a·w  ~=  (a_hi·w_hi + a_hi·w_lo + a_lo·w_hi) / s          accumulated in fp32 by the MFMA
//
// Each operand is represented to ~2^-22 relative, the dropped a_lo·w_lo term is 2^-22 relative, so a
// product carries ~3·2^-22 = 7e-7 relative error (fp32: 6e-8) — two orders below the 1e-4 parity
// budget, and far tighter than the fp16 autocast the reference's own launch line uses (run.py:48-55
// `--amp`).  v_mfma_f32_32x32x16_f16 moves 16 k per 32 cycles against 2 k per 64 cycles of the exact
// f32 MFMA: 16x the rate, 5.3x after the 3 products.  `s` is a per-matrix power of two that lifts the
// weights (|w| ~ 0.05) to ~2^9 so that w_lo stays a normal fp16 number; activations are O(1) and are
// not scaled.  Weights are split once in adf_painn_set_weights; activations are split on the fly
// while they are staged into LDS.  The exact-f32 kernel (gemm.hip) stays selectable
// (ADF_GEMM=f32) and is what this kernel is tested against.
//
// Tiling (default): 128(M) x 256(N) x 32(K) per 256-thread workgroup, 4 waves as 2(M) x 2(N), each wave 64 x 128 =
// 2 x 4 MFMA 32x32 accumulators, two workgroups per CU.  LDS holds A_hi, A_lo [128][32+8] and W_hi, W_lo [256][32+8]
// halves (row stride 80 B: the 16 lanes of a ds_read_b128 lane group land on 16 distinct 16-B bank quads).
// Global loads are 16 B per lane, prefetched one K-tile ahead in registers; XCD-aware tile map as in
// gemm.hip (all N-tiles of an M-panel on one XCD's L2).  Every epilogue transposes the accumulators through the
// idle staging LDS so that global accesses are 16 B per lane, and issues all its loads before its first store.
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#ifndef EPI_ABL
#define EPI_ABL 0   // timing experiments only (wrong results): bit 0 = EPI 1 without its epilogue, bit 1 = EPI 2 without,
                    // bit 2 = EPI 1/2 without their global loads, bit 3 = without their global stores
#endif
#ifndef G16_ABL
#define G16_ABL 0   // timing experiments only (wrong results; harness builds): 1 = vec_proj epilogue without its global stores,
                    // 2 = A tile staged without the fp32 -> hi/lo conversion, 4 = A rows all read from row 0 (cache-hot),
                    // 8 = no products (staging + epilogue only)
#endif
#ifndef G16_WDEEP
#define G16_WDEEP 1   // WR kernels request their weight fragments two k-steps ahead (four register sets); 0: one k-step (node products 1699 -> 1681 ms per pass with 1)
#endif
#define HK 32
// Row strides (floats) of the epilogues' transposition buffers.  Unpadded on purpose: with the lane groups of
// ds_read_b128 ({0-3,12-15,20-27}, ...) a stride of 96 (= 32 mod 64 banks) resp. 64 puts the four rows a group
// touches on disjoint 16-bank ranges; the +4 paddings one would add by habit (100, 68) made 2-way conflicts.
#define TLD3 96
#define TLD2 64
#define HLD 40  // halves per LDS row (32 + 8 pad)

// 2^e with mx 2^e in [2^14, 2^15); 1 for mx == 0 or non-finite
__device__ __forceinline__ float adf_pow2_lift(float mx) {
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(mx, &e);
    e = 15 - e;
    e = e > 120 ? 120 : (e < -120 ? -120 : e);
    return ldexpf(1.0f, e);
}

__device__ __forceinline__ float ssilu16(float x) {
    float s = x / (1.0f + expf(-x));
    return s * 1.6666666666666667f;
}

// max over the 16 lanes of a DPP row (non-negative values), left in every lane: four v_max_f32 with row rotations
__device__ __forceinline__ float adf_row16_max(float v) {
#define ADF_ROR(n_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n_), 0xf, 0xf, false))
    v = fmaxf(v, ADF_ROR(8));
    v = fmaxf(v, ADF_ROR(4));
    v = fmaxf(v, ADF_ROR(2));
    v = fmaxf(v, ADF_ROR(1));
#undef ADF_ROR
    return v;
}

// Workgroup = 4 waves as 2(M) x 2(N); wave tile = (32*MI) rows x (32*NJ) columns of MFMA 32x32 blocks.
//   MI=2,NJ=4 (default): 128 x 256 tile, 2 workgroups per CU.
//   NJ=3 with EPI != 0: the three column blocks of a wave are the three H-wide parts of a 3H-wide
//   linear layer for the SAME 32 channels (weights row-permuted at set_weights:
//   column g*96 + part*32 + q  <->  original row part*H + 32g + q), so the consumer's elementwise
//   work runs on the accumulators and the 3H-wide intermediate never goes to HBM:
//     EPI 1  x_proj.2 -> gather records of the message kernel ((P0,P1,P2,xa) + xc, P_i = vec_i * xb; message.hip)
//     EPI 2  xvec_proj.2 -> PaiNNUpdate gating + residuals + ScaleFactor (painn_denoising.py:614-623,449-451)
//   MI=3, NJ=2 with EPI 3: vec_proj of PaiNNUpdate (painn_denoising.py:602-611).  The A rows of vec [N,3,H] are
//   staged component-major (LDS row = component*32 + atom), so accumulator block i of a wave is component i of
//   its 32 atoms; the two column blocks are v1 and v2 of the same 32 channels (weights row-permuted: column
//   g*64 + part*32 + q  <->  row part*H + 32g + q).  dot = sum_xyz v1*v2 / sqrt(H) and |v2| ([N,H] each) are formed
//   on the accumulators: v2 (1.2 GB per layer at N = 200k) never goes to HBM and the separate reduction pass is gone.
//   WR (round 6; every product of the PaiNN sampler): the weights are NOT staged through LDS - every wave loads the MFMA B
//   fragments of its own columns straight from a fragment-ordered image (adf_pack_frag, mlp16.hip; `Whi` then points at that
//   image) into a ring of register sets, two k-steps ahead; LDS holds two A buffers, there is one barrier per K tile, and the
//   lift / split of the next A tile is dealt between the MFMAs of the current one (see the main loop).  Same products in the
//   same order as the LDS-staged form (kept below, WR = false): same bits.
//   NWN (WR only): waves along N.  2 (default): 4 waves, tile columns 64 NJ.  4: 8 waves as 2(M) x 4(N), 512 threads, tile
//   columns 128 NJ - the A tile (staged, lifted and split once per workgroup) then serves twice the columns: half the L2 reads
//   and half the conversions of the A panel (vec_proj: the panel was re-staged by 8 column tiles; staging + epilogue alone
//   took 1.14 of the kernel's 2.2 ms in the stand-alone harness).
template <int ACT, int MI, int NJ, int EPI, bool WR = false, int NWN = 2>
__global__ __launch_bounds__(128 * NWN, (MI == 4 || NWN == 4 ? 1 : 2)) void adf_gemm_f16x3_kernel(
    const float* __restrict__ A, int lda, const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
    const float* __restrict__ inv_scale, const float* __restrict__ bias, float* __restrict__ C, int ldc, int Mh, int N,
    int K, int tiles_n, adf_epi ep) {
    const int M = ep.m_dev ? min(Mh, (int)*ep.m_dev) : Mh;  // rows: the host's bound, or fewer by a device-side count
    static_assert(NWN == 2 || WR, "eight waves only with the register-streamed weights");
    constexpr int NT = 128 * NWN;        // threads
    constexpr int TM = 64 * MI;          // rows per workgroup
    constexpr int TN = 32 * NJ * NWN;    // columns per workgroup
    constexpr int NA = TM * 8 / NT;      // float4 A loads per thread
    static_assert(TM * 8 % NT == 0, "A staging: whole float4s per thread");
    // (WR: no W tiles; two A buffers, reused as the epilogues' scratch of NT / 64 waves x 3200 floats)
    constexpr int LDS_WR = 4 * TM * HLD > NT / 64 * 6400 ? 4 * TM * HLD : NT / 64 * 6400;   // two A buffers (hi + lo) | the scratch
    __shared__ __attribute__((aligned(16))) _Float16 lds[WR ? LDS_WR : (2 * TM + 2 * TN) * HLD];
    __shared__ float rinv[TM];  // 1 / lift of every staged A row
    _Float16* Ahi = lds;
    _Float16* Alo = Ahi + TM * HLD;
    _Float16* Bhi = Alo + TM * HLD;
    _Float16* Blo = Bhi + TN * HLD;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave / NWN) * (32 * MI);
    const int wn = (wave % NWN) * (32 * NJ);

    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int qd = id >> 3;
    const int tile_m = (qd / tiles_n) * 8 + xcd;
    const int tile_n = qd % tiles_n;
    const int m0 = tile_m * (EPI >= 3 ? TM / 3 : TM);  // EPI 3, 4: M counts atoms, a tile holds TM/3 of them
    const int n0 = tile_n * TN;
    if (m0 >= M) return;

    // A staging: NA float4 per thread (8 lanes per 128-B row segment, 32 rows per pass).  Addresses are a
    // wave-uniform base + a 32-bit byte offset per lane (the launchers check rows*lda*4 < 2^32), which also lets
    // the K range be split over two sources of equal row stride: columns [0,K1) from A, [K1,K) from ep.A2
    // (the [x | norm] inputs of the update / head MLPs are never concatenated in memory).
    unsigned int a_goff[NA];
    int a_off[NA];
    float a_rs[NA];
    auto global_row = [&](int row) -> unsigned int {
        if constexpr (EPI >= 3) {  // LDS row = (wave row)*96 + component*32 + atom
            const int atom = m0 + (row / 96) * 32 + (row & 31), ax = (row % 96) >> 5;
            return (unsigned int)min(atom, M - 1) * 3u + ax;
        } else {
            return (unsigned int)min(m0 + row, M - 1);
        }
    };
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int f = tid + NT * i;
        const int row = f >> 3, kq = f & 7;
        const unsigned int grow = global_row(row);
        a_goff[i] = (grow * (unsigned int)lda + kq * 4) * 4u;
        a_off[i] = row * HLD + kq * 4;
        a_rs[i] = ep.rmag ? adf_pow2_lift(ep.rmag[grow]) : 1.0f;
    }
    if (tid < TM) rinv[tid] = ep.rmag ? 1.0f / adf_pow2_lift(ep.rmag[global_row(tid)]) : 1.0f;
    const char* const A1b = reinterpret_cast<const char*>(A);
    const char* const A2b = reinterpret_cast<const char*>(ep.A2) - (size_t)ep.K1 * 4;
    // W staging: NJ pieces of 16 B (8 halves) of hi and of lo per thread (4 lanes per 64-B row)
    int w_src[NJ];
    int w_off[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int f = tid + 256 * i;
        const int row = f >> 2, part = f & 3;
        w_src[i] = min(n0 + row, N - 1) * K + part * 8;
        w_off[i] = row * HLD + part * 8;
    }

    float4 ra[NA];
    half8 rwh[NJ], rwl[NJ];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4*>(A1b + a_goff[i]);
    if constexpr (!WR) {
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            rwh[i] = *reinterpret_cast<const half8*>(Whi + w_src[i]);
            rwl[i] = *reinterpret_cast<const half8*>(Wlo + w_src[i]);
        }
    }
    // WR: B fragments of one 16-deep k-step for this wave's NJ column blocks: [column block][hi | lo]; two register sets
    half8 wfa[NJ][2], wfb[NJ][2];
    const half8* const wfr = reinterpret_cast<const half8*>(Whi) + (size_t)((n0 + wn) / 32) * (K / 16) * 128 + lane;
    auto load_wf = [&](int s_, half8 (&wf)[NJ][2]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const size_t base = ((size_t)j * (K / 16) + s_) * 128;
            wf[j][0] = wfr[base];
            wf[j][1] = wfr[base + 64];
        }
    };
    if constexpr (WR) load_wf(0, wfa);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / HK;
    const int fa = (wm + (lane & 31)) * HLD + (lane >> 5) * 8;
    const int fb = (wn + (lane & 31)) * HLD + (lane >> 5) * 8;
    auto stage_a = [&](int boff = 0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (G16_ABL & 2) {   // move the bytes, skip the split
                *reinterpret_cast<float2*>(Ahi + boff + a_off[i]) = make_float2(ra[i].x, ra[i].y);
                *reinterpret_cast<float2*>(Alo + boff + a_off[i]) = make_float2(ra[i].z, ra[i].w);
                continue;
            }
            const float sx = ra[i].x * a_rs[i], sy = ra[i].y * a_rs[i], sz = ra[i].z * a_rs[i], sw = ra[i].w * a_rs[i];
            half4 h, l;
            h[0] = (_Float16)sx; h[1] = (_Float16)sy; h[2] = (_Float16)sz; h[3] = (_Float16)sw;
            l[0] = (_Float16)(sx - (float)h[0]); l[1] = (_Float16)(sy - (float)h[1]);
            l[2] = (_Float16)(sz - (float)h[2]); l[3] = (_Float16)(sw - (float)h[3]);
            *reinterpret_cast<half4*>(Ahi + boff + a_off[i]) = h;
            *reinterpret_cast<half4*>(Alo + boff + a_off[i]) = l;
        }
    };
    auto request_a = [&](int kt1) {
        const int k1 = kt1 * HK;
        const char* ab = ((ep.K1 > 0 && k1 >= ep.K1) ? A2b : A1b) + (size_t)k1 * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4*>(ab + ((G16_ABL & 4) ? (a_goff[i] & 127u) : a_goff[i]));
    };
    if constexpr (WR) {
        // The B fragments of a k-step are requested one or two k-steps ahead (G16_WDEEP), in front of the products of the k-step
        // before (sched_barrier(0) between the request group and the product group: left alone, hipcc sinks the requests to
        // their first use and every k-step waits out an L2 round trip - mlp16.hip's first build).
        auto kstep = [&](int boff, int ks, const half8 (&cur)[NJ][2]) {
            if (G16_ABL & 8) { asm volatile("" :: "v"(cur[0][0]), "v"(cur[0][1]), "v"(cur[NJ - 1][0]), "v"(cur[NJ - 1][1])); return; }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const half8 ah = *reinterpret_cast<const half8*>(Ahi + boff + fa + i * 32 * HLD + ks * 16);
                const half8 al = *reinterpret_cast<const half8*>(Alo + boff + fa + i * 32 * HLD + ks * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, cur[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[j][0], acc[i][j], 0, 0, 0);
                }
            }
        };
        // (First form of the pipeline, measured and dropped: the conversion of tile kt + 1 BEHIND tile kt's products instead of
        // between them - node products 1829-1833 ms per pass against 1812 with two barriers per tile and no second buffer.)
        {
            // Software-pipelined staging: two A buffers, one barrier per K tile, and the conversion of tile kt + 1 is dealt
            // BETWEEN the products of tile kt's second k-step (sched_group_barrier: 1 MFMA, 4 VALU), where the matrix pipe hides
            // it; its rows were requested two tiles ahead (two register sets).  With eight waves (the only workgroup of its
            // CU) nothing else could fill the staging time; with four it adds to the overlap between the CU's two workgroups.
            constexpr int BUF = 2 * TM * HLD;
            float4 rb[NA];
            auto request_b = [&](int kt1) {
                const int k1 = kt1 * HK;
                const char* ab = ((ep.K1 > 0 && k1 >= ep.K1) ? A2b : A1b) + (size_t)k1 * 4;
#pragma unroll
                for (int i = 0; i < NA; ++i) rb[i] = *reinterpret_cast<const float4*>(ab + ((G16_ABL & 4) ? (a_goff[i] & 127u) : a_goff[i]));
            };
            auto stage_from = [&](const float4 (&src)[NA], int boff) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    if (G16_ABL & 2) {   // move the bytes, skip the split
                        *reinterpret_cast<float2*>(Ahi + boff + a_off[i]) = make_float2(src[i].x, src[i].y);
                        *reinterpret_cast<float2*>(Alo + boff + a_off[i]) = make_float2(src[i].z, src[i].w);
                        continue;
                    }
                    const float sx = src[i].x * a_rs[i], sy = src[i].y * a_rs[i], sz = src[i].z * a_rs[i], sw = src[i].w * a_rs[i];
                    half4 h, l;
                    h[0] = (_Float16)sx; h[1] = (_Float16)sy; h[2] = (_Float16)sz; h[3] = (_Float16)sw;
                    l[0] = (_Float16)(sx - (float)h[0]); l[1] = (_Float16)(sy - (float)h[1]);
                    l[2] = (_Float16)(sz - (float)h[2]); l[3] = (_Float16)(sw - (float)h[3]);
                    *reinterpret_cast<half4*>(Ahi + boff + a_off[i]) = h;
                    *reinterpret_cast<half4*>(Alo + boff + a_off[i]) = l;
                }
            };
            // tile kt: products on buffer `cur` with the registers `nxt_regs` (tile kt + 1) converted into the other buffer
#if G16_WDEEP
            // weight fragments TWO k-steps ahead: four register sets, a tile uses (w0, w1) and requests the next tile's into (w2, w3)
            half8 wfc[NJ][2], wfd[NJ][2];
            load_wf(1, wfb);
            auto tile = [&](int kt, int cur, const float4 (&nxt_regs)[NA], const half8 (&w0)[NJ][2], const half8 (&w1)[NJ][2],
                            half8 (&w2)[NJ][2], half8 (&w3)[NJ][2]) {
                load_wf(min(2 * kt + 2, 2 * nk - 1), w2);
                __builtin_amdgcn_sched_barrier(0);
                kstep(cur, 0, w0);
                __builtin_amdgcn_sched_barrier(0);
                load_wf(min(2 * kt + 3, 2 * nk - 1), w3);
                __builtin_amdgcn_sched_barrier(0);
                kstep(cur, 1, w1);
                stage_from(nxt_regs, BUF - cur);
#else
            auto tile = [&](int kt, int cur, const float4 (&nxt_regs)[NA]) {
                load_wf(2 * kt + 1, wfb);
                __builtin_amdgcn_sched_barrier(0);
                kstep(cur, 0, wfa);
                __builtin_amdgcn_sched_barrier(0);
                load_wf(min(2 * kt + 2, 2 * nk - 1), wfa);
                __builtin_amdgcn_sched_barrier(0);
                kstep(cur, 1, wfb);
                stage_from(nxt_regs, BUF - cur);   // (behind the last tile: a copy nobody reads)
#endif
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * MI, 0);
#pragma unroll
                for (int u = 0; u < 3 * MI * NJ; ++u) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
                    if (u % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // (Measured and dropped: FOUR A buffers and one barrier per PAIR of K tiles for the eight-wave kernels - identical
            // sites, node products 1662 -> 1714 ms per pass, heads 420 -> 442: the second tile staged in the prologue costs every
            // workgroup another exposed HBM round trip, the barriers were not what the tiles were waiting for.)
            __syncthreads();   // (rinv)
            stage_a(0);                       // tile 0 (requested in the prologue)
            request_a(min(1, nk - 1));        // tile 1 -> ra
            request_b(min(2, nk - 1));        // tile 2 -> rb
            __syncthreads();
            for (int kt = 0; kt < nk; kt += 2) {   // (nk is even: checked by the launcher)
#if G16_WDEEP
                tile(kt, 0, ra, wfa, wfb, wfc, wfd);
                request_a(min(kt + 3, nk - 1));
                __syncthreads();
                tile(kt + 1, BUF, rb, wfc, wfd, wfa, wfb);
#else
                tile(kt, 0, ra);
                request_a(min(kt + 3, nk - 1));
                __syncthreads();
                tile(kt + 1, BUF, rb);
#endif
                request_b(min(kt + 4, nk - 1));
                __syncthreads();
            }
        }
    } else {
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        stage_a();
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            *reinterpret_cast<half8*>(Bhi + w_off[i]) = rwh[i];
            *reinterpret_cast<half8*>(Blo + w_off[i]) = rwl[i];
        }
        __syncthreads();
        if (kt + 1 < nk) {
            const int k1 = (kt + 1) * HK;
            request_a(kt + 1);
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
                bh[j] = *reinterpret_cast<const half8*>(Bhi + fb + j * 32 * HLD + ks * 16);
                bl[j] = *reinterpret_cast<const half8*>(Blo + fb + j * 32 * HLD + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const half8 ah = *reinterpret_cast<const half8*>(Ahi + fa + i * 32 * HLD + ks * 16);
                const half8 al = *reinterpret_cast<const half8*>(Alo + fa + i * 32 * HLD + ks * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    // small terms first, then the leading product
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    }

    const float isc = *inv_scale;
    if constexpr (EPI == 0) {
        if (((N | ldc) & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && NJ % 2 == 0) {
            // Transpose 32 x 64 pieces of the accumulators through the (now idle) staging LDS so that each lane
            // stores 16 B: 4 rows x 256 B per wave instruction instead of 2 rows x 128 B with 4-B stores.
            const int q = lane & 31;
            __syncthreads();  // all waves are done reading the operand tiles
            float* T = reinterpret_cast<float*>(lds) + wave * (32 * TLD2);  // [32 rows][64] floats
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
                        if (ACT) { v0 = ssilu16(v0); v1 = ssilu16(v1); }
                        T[lr * TLD2 + q] = v0;
                        T[lr * TLD2 + 32 + q] = v1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int item = lane + 64 * it;
                        const int lr = item >> 4, c4 = item & 15;
                        const int row = m0 + wm + 32 * i + lr, col = cb + 4 * c4;
                        float4 v = *reinterpret_cast<const float4*>(T + lr * TLD2 + 4 * c4);
                        if (row < M && col < N) {
                            if (ep.gate) {   // heads: v' = g (x) vec2_proj(v) (painn_denoising.py:693-696): row = 3 atom + component
                                const float4 g4 = *reinterpret_cast<const float4*>(ep.gate + (size_t)(row / 3) * ep.gate_ld + col);
                                v.x = g4.x * v.x; v.y = g4.y * v.y; v.z = g4.z * v.z; v.w = g4.w * v.w;
                            }
                            if (ep.accumulate) {   // (training: data gradients summed into their destination, no temporary + add pass)
                                const float4 o = *reinterpret_cast<const float4*>(C + (size_t)row * ldc + col);
                                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                            }
                            *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = v;
                        }
                        if (ep.out_mag) {  // the 16 lanes of a DPP row hold one output row: its maximum by four rotations
                            float mg = col < N ? fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) : 0.f;
                            mg = adf_row16_max(mg);
                            if (c4 == 0 && row < M) atomicMax(ep.out_mag + row, __float_as_uint(mg));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();  // T is rewritten by the next piece
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < M) {
                        float v = acc[i][j][r] * (isc * rinv[row - m0]) + bv;
                        if (ACT) v = ssilu16(v);
                        if (ep.accumulate) v += C[(size_t)row * ldc + col];
                        C[(size_t)row * ldc + col] = v;
                        if (ep.out_mag) atomicMax(ep.out_mag + row, __float_as_uint(fabsf(v)));
                    }
                }
            }
        }
    } else if constexpr (EPI == 3) {
        static_assert(EPI != 3 || (NJ == 2 && MI == 3), "vec_proj epilogue: 3 components x (v1, v2)");
        const int q = lane & 31;
        const int g = (n0 + wn) / 64;
        const int H = ep.H;
        const float inv_sqrt_h = 1.0f / sqrtf((float)H);
        __syncthreads();  // all waves are done reading the operand tiles
        float* T = reinterpret_cast<float*>(lds) + wave * 3200;  // [32 atoms][TLD3] floats
        const int a0 = m0 + (wave / NWN) * 32;                   // first atom of this wave
        float dv[16], nv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float d = 0.f, qq = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float sc = isc * rinv[wm + 32 * i + lr];
                const float v1 = acc[i][0][r] * sc, v2 = acc[i][1][r] * sc;
                d += v1 * v2;
                qq += v2 * v2;
                T[lr * TLD3 + i * 32 + q] = v1;
            }
            dv[r] = d * inv_sqrt_h;
            nv[r] = sqrtf(qq + 1e-8f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = lane + 64 * it;
            const int lr = item >> 3, c4 = item & 7;
            const int n = a0 + lr;
            if (n < M) {
                float* vo = ep.v1 + (size_t)n * 3 * H + 32 * g + 4 * c4;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const float4 tv_ = *reinterpret_cast<const float4*>(T + lr * TLD3 + ax * 32 + 4 * c4);
                    if (G16_ABL & 1) asm volatile("" :: "v"(tv_.x), "v"(tv_.w)); else
                    *reinterpret_cast<float4*>(vo + ax * H) = tv_;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            T[lr * TLD3 + q] = dv[r];
            T[lr * TLD3 + 32 + q] = nv[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int lr = it * 8 + (lane >> 3), c4 = lane & 7;
            const int n = a0 + lr, c = 32 * g + 4 * c4;
            if (n < M) {
                const float4 dv_ = *reinterpret_cast<const float4*>(T + lr * TLD3 + 4 * c4);
                const float4 nv_ = *reinterpret_cast<const float4*>(T + lr * TLD3 + 32 + 4 * c4);
                if (G16_ABL & 1) { asm volatile("" :: "v"(dv_.x), "v"(nv_.w)); } else {
                *reinterpret_cast<float4*>(ep.dotw + (size_t)n * H + c) = dv_;
                *reinterpret_cast<float4*>(ep.cat + (size_t)n * H + c) = nv_;
                }
            }
        }
    } else if constexpr (EPI == 4) {
        // GatedEquivariantBlock.vec1_proj (painn_denoising.py:687-692): only ||W1 v||_xyz is used downstream, so the
        // three component blocks are reduced on the accumulators and the [3N, C] product never goes to HBM.
        // Output: norm [M, N]; the consumer GEMM reads [x | norm] from two sources (ep.A2).
        static_assert(EPI != 4 || (NJ == 2 && MI == 3), "norm epilogue: 3 components x 64 columns per wave");
        const int q = lane & 31;
        const int cb = n0 + wn;
        __syncthreads();  // all waves are done reading the operand tiles
        float* T = reinterpret_cast<float*>(lds) + wave * 3200;  // [32 atoms][TLD2] floats used
        const int a0 = m0 + (wave / NWN) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float a = acc[0][j][r] * (isc * rinv[wm + lr]), b = acc[1][j][r] * (isc * rinv[wm + 32 + lr]),
                            d = acc[2][j][r] * (isc * rinv[wm + 64 + lr]);
                T[lr * TLD2 + j * 32 + q] = sqrtf(a * a + b * b + d * d);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int lr = it * 4 + (lane >> 4), c4 = lane & 15;
            const int n = a0 + lr, c = cb + 4 * c4;
            if (n < M && c < N)
                *reinterpret_cast<float4*>(ep.cat + (size_t)n * N + c) = *reinterpret_cast<const float4*>(T + lr * TLD2 + 4 * c4);
        }
    } else {
        // Columns of this wave: parts 0,1,2 of the 32 channels of group g.  The accumulators (lane =
        // channel, 16 rows) are transposed through the wave's share of the now idle staging LDS so
        // that every lane then owns 4 consecutive channels of one row: 16-B global accesses, 8 rows x
        // 128 B per wave instruction instead of 2 rows x 128 B.
        static_assert(EPI >= 3 || (NJ == 3 && MI == 2), "fused epilogues are written for the 64 x 96 wave tile");
        if (((EPI_ABL & 1) && EPI == 1) || ((EPI_ABL & 2) && EPI == 2)) {   // keep the accumulators alive, store nothing
            float sacc = 0.f;
            for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
            if (ep.H < 0) ep.x[0] = sacc;
            return;
        }
        const int q = lane & 31;
        const int g = (n0 + wn) / 96;
        const int H = ep.H;
        const float b0 = bias[n0 + wn + q], b1 = bias[n0 + wn + 32 + q], b2 = bias[n0 + wn + 64 + q];
        __syncthreads();  // all waves are done reading the operand tiles
        float* T = reinterpret_cast<float*>(lds) + wave * 3200;  // [32 rows][TLD3] floats
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float sc = isc * rinv[wm + 32 * i + lr];
                T[lr * TLD3 + q] = acc[i][0][r] * sc + b0;
                T[lr * TLD3 + 32 + q] = acc[i][1][r] * sc + b1;
                T[lr * TLD3 + 64 + q] = acc[i][2][r] * sc + b2;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // The 4 items of a lane (row lr = it*8 + lane/8, channels 4*c4..) : every global load of the block is
            // issued before the first store (rows clamped, stores predicated) — written item by item the stores,
            // which may alias the next loads, turn the epilogue into 16 dependent HBM round trips per block.
            const int c4 = lane & 7;
            const int c = 32 * g + 4 * c4;
            if constexpr (EPI == 1) {
                float4 v0[4], v1[4], v2[4];
                if (EPI_ABL & 4) { for (int it = 0; it < 4; ++it) { v0[it] = make_float4(1.f, 2.f, 3.f, 4.f); v1[it] = v0[it]; v2[it] = v0[it]; } }
                else if (!ep.vec_is_zero) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int n = min(m0 + wm + 32 * i + it * 8 + (lane >> 3), M - 1);
                        const float* vr = ep.vec_in + (size_t)n * 3 * H + c;
                        v0[it] = *reinterpret_cast<const float4*>(vr);
                        v1[it] = *reinterpret_cast<const float4*>(vr + H);
                        v2[it] = *reinterpret_cast<const float4*>(vr + 2 * H);
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int lr = it * 8 + (lane >> 3);
                    const int n = m0 + wm + 32 * i + lr;
                    const float4 p0 = *reinterpret_cast<const float4*>(T + lr * TLD3 + 4 * c4);
                    const float4 p1 = *reinterpret_cast<const float4*>(T + lr * TLD3 + 32 + 4 * c4);
                    const float4 p2 = *reinterpret_cast<const float4*>(T + lr * TLD3 + 64 + 4 * c4);
                    if ((EPI_ABL & 8) && n >= 0) { if (ep.H < 0) ep.rec[0] = p0.x + p1.x + p2.x + v0[it].x + v1[it].y + v2[it].z; }
                    else if (n < M) {
                        // half-record of (atom n, group g): [32 x (P0, P1, P2, xa)] then [32 x xc]
                        float* rec = ep.rec + ((size_t)(ep.row_map ? ep.row_map[n] : n) * (H / 32) + g) * 160;
                        float4* ra_ = reinterpret_cast<float4*>(rec + 16 * c4);
                        if (!ep.vec_is_zero) {
                            ra_[0] = make_float4(v0[it].x * p1.x, v1[it].x * p1.x, v2[it].x * p1.x, p0.x);
                            ra_[1] = make_float4(v0[it].y * p1.y, v1[it].y * p1.y, v2[it].y * p1.y, p0.y);
                            ra_[2] = make_float4(v0[it].z * p1.z, v1[it].z * p1.z, v2[it].z * p1.z, p0.z);
                            ra_[3] = make_float4(v0[it].w * p1.w, v1[it].w * p1.w, v2[it].w * p1.w, p0.w);
                        } else {
                            ra_[0] = make_float4(0.f, 0.f, 0.f, p0.x);
                            ra_[1] = make_float4(0.f, 0.f, 0.f, p0.y);
                            ra_[2] = make_float4(0.f, 0.f, 0.f, p0.z);
                            ra_[3] = make_float4(0.f, 0.f, 0.f, p0.w);
                        }
                        *reinterpret_cast<float4*>(rec + 128 + 4 * c4) = p2;
                    }
                }
            } else {
                float4 d[4], xv[4], w1[4][3], tv[4][3];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int n = min(m0 + wm + 32 * i + it * 8 + (lane >> 3), M - 1);
                    const size_t xo = (size_t)n * H + c;
                    d[it] = *reinterpret_cast<const float4*>(ep.dot + xo);
                    xv[it] = *reinterpret_cast<const float4*>(ep.x + xo);
                    const float* vr = ep.vec + (size_t)n * 3 * H + c;
                    const float* v1p = ep.vv + (size_t)n * 3 * H + c;  // v1 [N,3,H] written by EPI 3
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        w1[it][ax] = *reinterpret_cast<const float4*>(v1p + ax * H);
                        tv[it][ax] = *reinterpret_cast<const float4*>(vr + ax * H);
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int lr = it * 8 + (lane >> 3);
                    const int n = m0 + wm + 32 * i + lr;
                    const float4 p0 = *reinterpret_cast<const float4*>(T + lr * TLD3 + 4 * c4);
                    const float4 p1 = *reinterpret_cast<const float4*>(T + lr * TLD3 + 32 + 4 * c4);
                    const float4 p2 = *reinterpret_cast<const float4*>(T + lr * TLD3 + 64 + 4 * c4);
                    const float k2 = 0.70710678118654752f, sc = ep.scale;
                    float4 xo4 = xv[it];
                    xo4.x = (xo4.x + (p0.x + p1.x * d[it].x) * k2) * sc;
                    xo4.y = (xo4.y + (p0.y + p1.y * d[it].y) * k2) * sc;
                    xo4.z = (xo4.z + (p0.z + p1.z * d[it].z) * k2) * sc;
                    xo4.w = (xo4.w + (p0.w + p1.w * d[it].w) * k2) * sc;
                    if (n < M) {
                        *reinterpret_cast<float4*>(ep.x + (size_t)n * H + c) = xo4;
                        float* vr = ep.vec + (size_t)n * 3 * H + c;
#pragma unroll
                        for (int ax = 0; ax < 3; ++ax) {
                            float4 t = tv[it][ax];
                            const float4 v1 = w1[it][ax];
                            t.x += p2.x * v1.x; t.y += p2.y * v1.y; t.z += p2.z * v1.z; t.w += p2.w * v1.w;
                            *reinterpret_cast<float4*>(vr + ax * H) = t;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();  // T is rewritten for the next 32-row block
        }
    }
}

// mag[r] = max over the K1 leading elements of A row r (and the K2 of A2 row r, same row stride): one wave per row
__global__ __launch_bounds__(256) void adf_rowmag_kernel(const float* __restrict__ A, int lda, int K1,
                                                         const float* __restrict__ A2, int K2, long long M,
                                                         float* __restrict__ mag, const int32_t* __restrict__ m_dev,
                                                         int m_mul) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= M || (m_dev && r >= (long long)*m_dev * m_mul)) return;
    float mx = 0.f;
    const float* a = A + (size_t)r * lda;
    for (int k = lane * 4; k < K1; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(a + k);
        mx = fmaxf(fmaxf(fmaxf(mx, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (A2) {
        const float* b = A2 + (size_t)r * lda;
        for (int k = lane * 4; k < K2; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(b + k);
            mx = fmaxf(fmaxf(fmaxf(mx, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) mag[r] = mx;
}

int32_t adf_launch_rowmag(const float* A, int lda, int K1, const float* A2, int K2, long long M, float* mag, hipStream_t s,
                          const int32_t* m_dev, int m_mul) {
    if (M <= 0) return ADF_OK;
    hipLaunchKernelGGL(adf_rowmag_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, A, lda, K1, A2, K2, M, mag, m_dev,
                       m_mul);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// row magnitudes for a launcher: the caller's (premag), or measured into lf->buf, or none.  (Measured alternative: every
// workgroup measuring its own A panel in a prologue - no separate pass, same bits: 231.2 vs 233.5 sites/s at 1000 systems,
// 473 vs 493 it/s at B = 1: the panel is re-read by each of its 2-8 column tiles and the prologue is serial; not kept.)
// (Round 6: ONE pass over the vec rows and the x rows of a layer + norm(v2)'s row maxima raised from vec_proj's epilogue by
// atomicMax, instead of the two passes (vec; [x | norm(v2)]): identical sites, node products 1672 / 1676 ms per pass with and
// 1673 / 1677 without - the epilogue's atomics cost what the second pass did.)
// (Round 6, again: the eight-wave vec_proj kernel measuring its own 192 rows in a prologue - one wave per row, eight rows in
// flight - instead of the adf_rowmag pass over `vec`: identical sites, node products 1647 -> 2014 ms per pass.  At one workgroup
// per CU nothing overlaps the prologue's three dependent HBM round trips.)
static int32_t lift_mags(const float* A, int lda, int K1, const float* A2, int K2, long long rows, const adf_lift* lf,
                         const float* premag, const float** out, hipStream_t s, const int32_t* m_dev = nullptr, int m_mul = 1) {
    *out = premag;
    if (!premag && lf && lf->buf) {
        if (rows > lf->cap) { adf_set_error("gemm16: lift scratch holds %lld rows, need %lld", lf->cap, rows); return ADF_EINVAL; }
        ADF_TRY(adf_launch_rowmag(A, lda, K1, A2, K2, rows, lf->buf, s, m_dev, m_mul));
        *out = lf->buf;
    }
    return ADF_OK;
}

// ---- weight preparation: per-matrix power-of-two scale, then hi/lo split
__global__ void adf_absmax_kernel(const float* __restrict__ w, long long n, unsigned int* out_bits) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));  // non-negative floats order like uints
}

// perm_H > 0: the matrix is [parts*perm_H, K]; output row g*32*parts + part*32 + q takes input row part*perm_H + 32g + q
__global__ void adf_split_kernel(const float* __restrict__ w, long long n, const unsigned int* absmax_bits,
                                 _Float16* __restrict__ hi, _Float16* __restrict__ lo, float* inv_scale, int perm_H,
                                 int K, const float* __restrict__ bias, float* __restrict__ bias_perm, int parts) {
    const float amax = __uint_as_float(*absmax_bits);
    // scale = 2^(9 - floor(log2(amax)))  ->  |w*scale| in [2^9, 2^10)
    int e = 0;
    if (amax > 0.f) (void)frexpf(amax, &e);  // amax = m * 2^e, m in [0.5,1)
    const float scale = ldexpf(1.0f, 10 - e);
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = 1.0f / scale;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long src = i;
        if (perm_H > 0) {
            const long long prow = i / K;
            const int k = (int)(i - prow * K);
            const int gw = 32 * parts;
            const int g = (int)(prow / gw), part = (int)((prow % gw) / 32), qq = (int)(prow % 32);
            const long long orow = (long long)part * perm_H + 32 * g + qq;
            src = orow * K + k;
            if (k == 0 && bias_perm) bias_perm[prow] = bias ? bias[orow] : 0.f;
        }
        const float v = w[src] * scale;
        const _Float16 h = (_Float16)v;
        hi[i] = h;
        lo[i] = (_Float16)(v - (float)h);
    }
}

int32_t adf_split_weight(const float* w, long long n, adf_w16* out, unsigned int* scratch_bits, hipStream_t s,
                         int perm_H, int K, const float* bias, int parts) {
    ADF_HIP_CHECK(hipMemsetAsync(scratch_bits, 0, sizeof(unsigned int), s));
    hipLaunchKernelGGL(adf_absmax_kernel, dim3(64), dim3(256), 0, s, w, n, scratch_bits);
    hipLaunchKernelGGL(adf_split_kernel, dim3(64), dim3(256), 0, s, w, n, scratch_bits, (_Float16*)out->hi,
                       (_Float16*)out->lo, out->inv_scale, perm_H, K, bias, out->bias_perm, parts);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// the kernels address A with 32-bit byte offsets
static int32_t check_a_span(long long rows, int lda) {
    if (rows * (long long)lda * 4 >= (1ll << 32)) {
        adf_set_error("gemm16: A operand of %lld rows x %d exceeds the 32-bit offset range, split the batch", rows, lda);
        return ADF_EOOM;  // surfaces as RuntimeError -> ml_diffuse splits the batch
    }
    return ADF_OK;
}

int32_t adf_launch_gemm16(const float* A, int lda, const adf_w16* W, const float* bias, float* C, int ldc, int M,
                          int N, int K, int act_ssilu, hipStream_t s, const float* A2, int K1, const adf_lift* lf,
                          const float* premag, float* out_mag, const int32_t* m_dev, int accumulate, const float* gate,
                          int gate_ld) {
    if (M <= 0) return ADF_OK;
    if (K % HK != 0 || (lda & 3) || (A2 && (K1 <= 0 || K1 % HK != 0 || K1 >= K))) {
        adf_set_error("gemm16: K=%d (K1=%d) must be multiples of %d and lda a multiple of 4", K, K1, HK);
        return ADF_EINVAL;
    }
    ADF_TRY(check_a_span(M, lda));
    static int mi = 0;
    if (!mi) { const char* e = getenv("ADF_GEMM16_MI"); mi = e ? atoi(e) : 2; if (mi != 4) mi = 2; }
    const int TM = 64 * mi, TN = 256;
    const int tiles_n = (N + TN - 1) / TN;
    const int tiles_m = (M + TM - 1) / TM;
    const int tiles_m8 = (tiles_m + 7) / 8 * 8;
    dim3 grid((unsigned)(tiles_m8 * tiles_n));
    adf_epi ep = {};
    ep.A2 = A2; ep.K1 = A2 ? K1 : 0; ep.m_dev = m_dev; ep.accumulate = accumulate;
    ep.gate = gate; ep.gate_ld = gate_ld;
    if (gate && ((N | ldc) & 3)) { adf_set_error("gemm16: the gate epilogue needs N and ldc multiples of 4"); return ADF_EINVAL; }
    ADF_TRY(lift_mags(A, lda, A2 ? K1 : K, A2, A2 ? K - K1 : 0, M, lf, premag, &ep.rmag, s, m_dev, 1));
    if (out_mag) {
        ADF_HIP_CHECK(hipMemsetAsync(out_mag, 0, sizeof(float) * (size_t)M, s));
        ep.out_mag = reinterpret_cast<unsigned int*>(out_mag);
    }
    // Round 6: weights streamed as fragments, eight waves per workgroup, 192 x 256 tile, software-pipelined staging (see the
    // kernel comment: WR, NWN = 4).  Needs the fragment image, N a multiple of 256 and an even number of K tiles.
    static int w8 = -1;     // ADF_GEMM_W8_PLAIN=0: the 128 x 256 tile with LDS-staged weights of rounds 1-5
    if (w8 < 0) { const char* e = getenv("ADF_GEMM_W8_PLAIN"); w8 = (e && atoi(e) == 0) ? 0 : 1; }
    // (the same pipeline with four waves - 192 x 128 tile, two workgroups per CU - measured 1.3 % slower on the node products
    // and 2.3 % on the heads; a 128 x 256 four-wave tile spills 36-46 registers)
    // One such workgroup per CU: with few, long tiles the last round matters.  A launch of fewer than four rounds whose last
    // round would fill under 30 % of the chip (25 000 rows: 131 x 2 = 262 workgroups on 256 CUs) keeps the 128 x 256 tile,
    // two workgroups per CU (392 on 512 slots).
    static int ncu = 0;
    if (!ncu) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256; }
    const long long wg8 = (long long)((M + 191) / 192) * (N / 256 > 0 ? N / 256 : 1);
    const bool ragged = wg8 < 4ll * ncu && wg8 % ncu != 0 && wg8 % ncu < (3 * ncu) / 10 && wg8 > ncu;
    // (256-row tiles, MI = 4 - every weight fragment reused by 128 rows - measured: no gain, 1672-1676 against 1670-1673 ms)
    if (w8 == 1 && W->frag && N % 256 == 0 && (K / HK) % 2 == 0 && mi == 2 && !ragged) {
        const int tn8 = N / 256, tmw8 = ((M + 191) / 192 + 7) / 8 * 8;
        dim3 g8((unsigned)(tmw8 * tn8));
        if (act_ssilu)
            hipLaunchKernelGGL((adf_gemm_f16x3_kernel<1, 3, 2, 0, true, 4>), g8, dim3(512), 0, s, A, lda, (const _Float16*)W->frag,
                               (const _Float16*)nullptr, W->inv_scale, bias, C, ldc, M, N, K, tn8, ep);
        else
            hipLaunchKernelGGL((adf_gemm_f16x3_kernel<0, 3, 2, 0, true, 4>), g8, dim3(512), 0, s, A, lda, (const _Float16*)W->frag,
                               (const _Float16*)nullptr, W->inv_scale, bias, C, ldc, M, N, K, tn8, ep);
        ADF_HIP_CHECK(hipGetLastError());
        return ADF_OK;
    }
#define LAUNCH16(ACT_, MI_)                                                                                  \
    hipLaunchKernelGGL((adf_gemm_f16x3_kernel<ACT_, MI_, 4, 0>), grid, dim3(256), 0, s, A, lda,              \
                       (const _Float16*)W->hi, (const _Float16*)W->lo, W->inv_scale, bias, C, ldc, M, N, K, \
                       tiles_n, ep)
    if (mi == 4) { if (act_ssilu) LAUNCH16(1, 4); else LAUNCH16(0, 4); }
    else { if (act_ssilu) LAUNCH16(1, 2); else LAUNCH16(0, 2); }
#undef LAUNCH16
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ||W v||_xyz of a [M,3,K] vector field -> nrm [M, N]  (EPI 4; N % 4 == 0)
int32_t adf_launch_gemm16_vecnorm(const float* A, int lda, const adf_w16* W, float* nrm, int M, int N, int K,
                                  hipStream_t s, const adf_lift* lf, const float* premag) {
    if (M <= 0) return ADF_OK;
    if (K % HK != 0 || (lda & 3) || (N & 3)) { adf_set_error("gemm16_vecnorm: bad shape"); return ADF_EINVAL; }
    ADF_TRY(check_a_span(3ll * M, lda));
    adf_epi ep = {};
    ep.cat = nrm;
    ADF_TRY(lift_mags(A, lda, K, nullptr, 0, 3ll * M, lf, premag, &ep.rmag, s));
    const int tn = (N + 127) / 128, tm8 = ((M + 63) / 64 + 7) / 8 * 8;
    static int w8 = -1;
    if (w8 < 0) { const char* e = getenv("ADF_GEMM_W8"); w8 = (e && atoi(e) == 0) ? 0 : 1; }
    if (w8 && W->frag && N % 256 == 0 && (K / HK) % 2 == 0) {   // weights streamed as fragments, eight waves (see the kernel comment)
        hipLaunchKernelGGL((adf_gemm_f16x3_kernel<0, 3, 2, 4, true, 4>), dim3((unsigned)(tm8 * (N / 256))), dim3(512), 0, s, A, lda,
                           (const _Float16*)W->frag, (const _Float16*)nullptr, W->inv_scale, (const float*)nullptr,
                           (float*)nullptr, 0, M, N, K, N / 256, ep);
        ADF_HIP_CHECK(hipGetLastError());
        return ADF_OK;
    }
    hipLaunchKernelGGL((adf_gemm_f16x3_kernel<0, 3, 2, 4>), dim3((unsigned)(tm8 * tn)), dim3(256), 0, s, A, lda,
                       (const _Float16*)W->hi, (const _Float16*)W->lo, W->inv_scale, (const float*)nullptr,
                       (float*)nullptr, 0, M, N, K, tn, ep);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// 3H-wide layer with row-permuted weights and a fused consumer epilogue (EPI 1 or 2, see kernel comment)
int32_t adf_launch_gemm16_fused(const float* A, int lda, const adf_w16* W, int M, int H, int K, int epi,
                                const adf_epi* ep_in, hipStream_t s, const adf_lift* lf) {
    adf_epi epv = *ep_in;
    const adf_epi* ep = &epv;
    if (M <= 0) return ADF_OK;
    if (K % HK != 0 || (lda & 3) || H % 64 != 0 || (epi != 3 && !W->bias_perm)) {
        adf_set_error("gemm16_fused: bad shape");
        return ADF_EINVAL;
    }
    ADF_TRY(check_a_span(epi == 3 ? 3ll * M : (long long)M, lda));
    ADF_TRY(lift_mags(A, lda, K, nullptr, 0, epi == 3 ? 3ll * M : (long long)M, lf, ep_in->rmag, &epv.rmag, s, ep_in->m_dev,
                      epi == 3 ? 3 : 1));
    const int N = 3 * H, TM = 128, TN = 192;
    const int tiles_n = N / TN;  // H % 64 == 0
    const int tiles_m = (M + TM - 1) / TM;
    const int tiles_m8 = (tiles_m + 7) / 8 * 8;
    dim3 grid((unsigned)(tiles_m8 * tiles_n));
    if (epi == 3) {  // vec_proj: A = vec [N,3,H], weights [2H, K] permuted in (v1, v2) pairs; M = atoms
        const int tn = 2 * H / 128, tm8 = ((M + 63) / 64 + 7) / 8 * 8;
        static int wreg = -1;   // ADF_GEMM_WREG=0: weights staged through LDS as in rounds 1-5
        if (wreg < 0) { const char* e = getenv("ADF_GEMM_WREG"); wreg = (e && atoi(e) == 0) ? 0 : 1; }
        static int w8 = -1;     // ADF_GEMM_W8=0: four waves per workgroup (128 columns) instead of eight (256)
        if (w8 < 0) { const char* e = getenv("ADF_GEMM_W8"); w8 = (e && atoi(e) == 0) ? 0 : 1; }
        // (four waves, two workgroups per CU, same pipeline: node products 1748 ms per pass against 1679)
        if (wreg && W->frag && w8 == 1 && (2 * H) % 256 == 0 && (K / HK) % 2 == 0)
            hipLaunchKernelGGL((adf_gemm_f16x3_kernel<0, 3, 2, 3, true, 4>), dim3((unsigned)(tm8 * (2 * H / 256))), dim3(512), 0, s,
                               A, lda, (const _Float16*)W->frag, (const _Float16*)nullptr, W->inv_scale, (const float*)nullptr,
                               (float*)nullptr, 0, M, 2 * H, K, 2 * H / 256, *ep);
        else
        hipLaunchKernelGGL((adf_gemm_f16x3_kernel<0, 3, 2, 3>), dim3((unsigned)(tm8 * tn)), dim3(256), 0, s, A, lda,
                           (const _Float16*)W->hi, (const _Float16*)W->lo, W->inv_scale, (const float*)nullptr,
                           (float*)nullptr, 0, M, 2 * H, K, tn, *ep);
        ADF_HIP_CHECK(hipGetLastError());
        return ADF_OK;
    }
    static int wf = -1;   // ADF_GEMM_WR_FUSED: 0 = LDS-staged weights (rounds 1-5), 2 / 4 = streamed fragments with 4 / 8 waves
    if (wf < 0) { const char* e = getenv("ADF_GEMM_WR_FUSED"); wf = e ? atoi(e) : 2; }
    // (measured: 4 waves 1656 ms of node products per pass, LDS-staged weights 1702, 8 waves 1741 - the record / gating
    // epilogues are HBM-heavy and want the CU's second workgroup beside them)
    if (wf && W->frag && (K / HK) % 2 == 0 && (wf == 2 || N % 384 == 0)) {
        const int tnw = wf == 4 ? N / 384 : tiles_n;
        dim3 gw((unsigned)(tiles_m8 * tnw));
#define LAUNCHWF(EPI_, NWN_)                                                                                                  \
        hipLaunchKernelGGL((adf_gemm_f16x3_kernel<0, 2, 3, EPI_, true, NWN_>), gw, dim3(128 * NWN_), 0, s, A, lda,            \
                           (const _Float16*)W->frag, (const _Float16*)nullptr, W->inv_scale, W->bias_perm, (float*)nullptr, 0, \
                           M, N, K, tnw, *ep)
        if (epi == 1) { if (wf == 4) LAUNCHWF(1, 4); else LAUNCHWF(1, 2); }
        else { if (wf == 4) LAUNCHWF(2, 4); else LAUNCHWF(2, 2); }
#undef LAUNCHWF
        ADF_HIP_CHECK(hipGetLastError());
        return ADF_OK;
    }
    if (epi == 1)
        hipLaunchKernelGGL((adf_gemm_f16x3_kernel<0, 2, 3, 1>), grid, dim3(256), 0, s, A, lda, (const _Float16*)W->hi,
                           (const _Float16*)W->lo, W->inv_scale, W->bias_perm, (float*)nullptr, 0, M, N, K, tiles_n, *ep);
    else
        hipLaunchKernelGGL((adf_gemm_f16x3_kernel<0, 2, 3, 2>), grid, dim3(256), 0, s, A, lda, (const _Float16*)W->hi,
                           (const _Float16*)W->lo, W->inv_scale, W->bias_perm, (float*)nullptr, 0, M, N, K, tiles_n, *ep);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
