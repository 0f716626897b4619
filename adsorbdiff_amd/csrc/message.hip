// PaiNN message block, fully fused: radial-basis projection (MFMA) + gather + gated equivariant
// message + per-target segmented sum + residual, one launch per layer.
//
// Reference (adsorbdiff/models/painn/painn_denoising.py):
//   :534      rbfh = rbf_proj(edge_rbf)                       [E,R] x [R,3H]
//   :549-555  (a,b,c) = split(xh[src] * rbfh);  m_x = a
//             m_v = (vec[src] * (b/sqrt3) + c (x) r_hat) / sqrt(H)
//   :565-566  dx[dst] += m_x ; dvec[dst] += m_v               (torch_scatter sum)
//   :443-445  x = (x + dx)/sqrt2 ; vec = vec + dvec
// and gemnet_oc/layers/radial_basis.py:18-43,64-82,235-244 for edge_rbf = env(d/rc)*gauss_k(d/rc).
//
// MI355X mapping.  rbfh ([E,3H] fp32, 6 KB per edge) is never materialised: a persistent
// 512-thread workgroup owns one 64-channel slice (3 x 64 = 192 columns of rbf_proj, 104 KB of LDS
// as fp16 hi/lo images, 96 KB k-major in exact-f32 mode).  Its 8 waves pull *target atoms* from an LDS work counter; one wave owns all
// incoming edges of its target (CSR over targets, edges pre-sorted by distance), in 32-row blocks:
//   1. the MFMA A operand is built in registers — lane (row, k) evaluates env(d_row)*exp(..) for its
//      own k — and the MFMA runs over 6 column blocks, but only over the k-window where
//      some row's Gaussian is non-negligible: terms with |k - (R-1) d/rc| >= 6 are dropped (each
//      < exp(-18) = 1.5e-8 of the leading term, below f32 rounding; the exact-f32 mode keeps |..| < 8).  Because a target's edges are sorted by
//      distance, a 32-row block spans a narrow band and the 128-deep contraction shrinks to ~35;
//   2. the 16 accumulator rows of each lane gather their source's packed record (xa, xc and
//      P_i = vec_i * xb for the lane's channels c0+q and c0+32+q; 40 B per lane and row as
//      dwordx4 + dword pieces, each contiguous per half-wave), served by the XCD's L2: slice = blockIdx % 8 = XCD under round-robin
//      dispatch, so one XCD only touches its own 64-channel columns of the record table;
//   3. the messages are summed in registers (wavefront segmented sum: the 16 rows of a lane, then
//      one cross-half shuffle), and x_out / vec_out rows are written once with the residual fused.
// No LDS atomics (ds_add_f32 measured 4x slower than the whole rest of the kernel), no barriers
// after the weight image is staged, no zero-initialised outputs.  HBM traffic per layer = record
// table + residual rows once + 20 B per edge (vs 6 KB per edge if rbfh were materialised).  Measured:
// matrix pipe ~30 % of nominal (0.42 of the on-box MFMA peak), gathers ~15 TB/s out of L2 at a 93 % hit
// rate; the contraction path and the gather path are about equally long and overlap.
//
// Two arithmetic modes for step 1, same structure otherwise:
//   F16 = true  (default): f16x3 split (see gemm16.hip) — A = a_hi + a_lo generated in registers,
//                W·s = w_hi + w_lo in LDS as [col][k] halves, v_mfma_f32_32x32x16_f16 x 3 products,
//                fp32 accumulate; the k-window is aligned to 8 and contracted 16 k at a time.
//   F16 = false (ADF_GEMM=f32): exact f32 v_mfma_f32_32x32x2_f32, window contracted 2 k at a time.
#include <stdlib.h>
#include <string.h>

#include "message.h"

#ifndef MSG_ABL
#define MSG_ABL 0    // timing experiments only (wrong results): 1 = no contraction loop, 2 = no gathers / sums
#endif
#ifndef MSG_SKIP
#define MSG_SKIP 1
#endif
#ifndef MSG_EXP_NOXC
#define MSG_EXP_NOXC 0   // timing experiments only (wrong results): 1 no xc gathers; 2 no record gathers at all
#endif

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Geometry of THIS kernel (message_bwd.hip keeps message.h's 8 waves): FWD_WAVES waves per workgroup share the slice's weight
// image; FWD_AHEAD = gather rows requested ahead of the row being consumed (4: 80 landing registers, 2: 40).
#ifndef FWD_WAVES
#define FWD_WAVES 8
#endif
#ifndef FWD_AHEAD
#define FWD_AHEAD 4
#endif
#define FWD_THREADS (64 * FWD_WAVES)
#define MSG_WAVES_PER_SIMD (FWD_WAVES / 4)
// VZ = true: vec is identically zero on entry (first layer, painn_denoising.py:426) — its gathers,
// the vec*b sums and the residual read are skipped.
// UNI: the Gaussian centres are equally spaced (GaussianSmearing's linspace, radial_basis.py:64-82; checked by the host):
// the 8 basis values of a lane and k-step then follow from two exp2 by a multiplicative recurrence (see below).
template <bool F16, bool VZ, bool UNI>
__global__ __launch_bounds__(FWD_THREADS, MSG_WAVES_PER_SIMD) void adf_message_kernel(MsgParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // carve: weight image | [192] bias | [128] mu | [8 waves][32][8] row meta | work counter
    const int wfloats = F16 ? (2 * MSG_COLS * MSG_LDK) / 2 : p.R * MSG_COLS;
    float* Wl = lds;
    _Float16* Wh = reinterpret_cast<_Float16*>(lds);       // [192][MSG_LDK] hi
    _Float16* Wlo = Wh + MSG_COLS * MSG_LDK;               // [192][MSG_LDK] lo
    float* Bl = lds + wfloats;
    float* Mu = Bl + MSG_COLS;
    float* Meta = Mu + 144;  // 16-entry tail: the operand generated one step ahead may read past the last centre
    int* Ctr = reinterpret_cast<int*>(Meta + FWD_WAVES * 32 * 8);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane & 31;
    const int hi = lane >> 5;
    const int slice = blockIdx.x % p.nslices;
    const int worker = blockIdx.x / p.nslices;
    const int nworkers = gridDim.x / p.nslices;
    const int items = p.items_dev ? min(p.items, (int)*p.items_dev) : p.items;
    const int ngroups = (items + ADF_GROUP_NODES - 1) / ADF_GROUP_NODES;
    if (worker >= ngroups) return;  // nothing for this workgroup (before the weight image is staged)
    const int H = p.H;
    const int c0 = slice * ADF_SLICE_CH;

    {   // stage this slice's rbf_proj image once
        if (F16) {
            const int R8 = p.R / 8;  // 16-B pieces per column row (R <= 128; the rest of the row is zero-filled,
                                     // the 16-deep contraction steps may reach past R)
            const half8* src = reinterpret_cast<const half8*>(p.wpack16 + (size_t)slice * 2 * MSG_COLS * p.R);
            const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
            // k slots 128/129 of the hi image (the row padding) hold the bias as an fp16 hi/lo pair: the
            // accumulators are initialised by one MFMA against a constant A instead of 96 v_mov per block
            const _Float16* b16 = reinterpret_cast<const _Float16*>(p.bpack) + (size_t)slice * MSG_COLS * 2;
            for (int i = tid; i < 2 * MSG_COLS * 17; i += FWD_THREADS) {
                const int row = i / 17, piece = i - row * 17;  // row in [0, 384): hi rows then lo rows
                half8 v = piece < R8 ? src[row * R8 + piece] : zero8;
                if (piece == 16 && row < MSG_COLS) { v[0] = b16[2 * row]; v[1] = b16[2 * row + 1]; }
                *reinterpret_cast<half8*>(Wh + (size_t)row * MSG_LDK + piece * 8) = v;
            }
        } else {
            const float4* src = reinterpret_cast<const float4*>(p.wpack + (size_t)slice * p.R * MSG_COLS);
            float4* dst = reinterpret_cast<float4*>(Wl);
            const int n4 = p.R * MSG_COLS / 4;
            for (int i = tid; i < n4; i += FWD_THREADS) dst[i] = src[i];
        }
        if (!F16 && tid < MSG_COLS) Bl[tid] = p.bpack[slice * MSG_COLS + tid];
        // f16 mode: centres pre-multiplied by sqrt(-coeff*log2 e), so that a Gaussian is exp2(-(xs' - mu')^2)
        if (tid < 144) Mu[tid] = (tid < p.R ? p.mu[tid] : 2.0f) * (F16 ? p.sarg : 1.0f);
        if (tid == 0) *Ctr = 0;
    }
    __syncthreads();
    float* meta_w = Meta + wave * 32 * 8;
    const float inv_sqrt3 = 0.57735026918962576f;
    const float inv_sqrt2 = 0.70710678118654752f;
    const float out_scale = F16 ? *p.inv_scale * (1.0f / 256.0f) : 1.0f;  // accumulators hold 256*scale*rbfh (f16)
    const float inv_sqrt_h = out_scale / sqrtf((float)H);
    const float umax_scale = (float)(p.R - 1);
    const float coeff2 = p.coeff * 1.44269504088896341f;  // exp(c z) = exp2(c log2e z)
    // one record row = H/32 half-records of 640 B: [32 x (P0, P1, P2, xa)] + [32 x xc] for 32 channels, read as one
    // dwordx4 + one dword per lane (12-B-per-lane loads gather at 0.7x the rate of this split: measured 20 vs 28 TB/s).
    // Lane q owns channels c0+q (half-record 2*slice) and c0+32+q (half-record 2*slice+1).
    const unsigned int row_bytes = (unsigned int)p.nslices * 1280u;
    // gathers: wave-uniform base (SGPR pair) + a 32-bit byte offset per lane -> global_load ... v_off, s[base] (one v_add_u32
    // per piece instead of 64-bit vector address arithmetic)
    const char* recS = reinterpret_cast<const char*>(p.rec) + (size_t)slice * 1280;
    const unsigned int qA = (unsigned int)q * 16u;
    const unsigned int qP = 512u + (unsigned int)q * 4u;

    unsigned int ksteps = 0;  // wave-uniform; one global atomic per wave at the very end (profiling)

    // Work items of this workgroup are target atoms (item t -> group worker + (t/32)*nworkers, atom
    // t%32: consecutive items are consecutive atoms, so the 8 waves write neighbouring output rows);
    // a wave walks its targets' 32-edge blocks as one flat, software-pipelined sequence: the NEXT
    // block's edge records (and, one target ahead, the next target's CSR bounds and residual rows)
    // are requested before the current block is processed, so a block exposes no dependent
    // HBM/L2 round trip of its own.
    // n_out = target atom (CSR row, residual row); o_out = output row (= n_out unless a target list is given)
    // CSR bounds and the target list are written by earlier launches and only read here: loading them through the constant
    // address space makes them scalar loads (s_load), off the vector-memory queue whose in-order vmcnt the gathers share.
    typedef const __attribute__((address_space(4))) int32_t* cint_ptr;
    const cint_ptr nptr_c = (cint_ptr)p.nptr;
    const cint_ptr tlist_c = (cint_ptr)p.tlist;
    auto fetch_target = [&](int& n_out, int& o_out) -> bool {
        while (true) {
            int t = 0;
            if (lane == 0) t = atomicAdd(Ctr, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            const int g = worker + (t >> 5) * nworkers;
            if (g >= ngroups) return false;
            const int e = g * ADF_GROUP_NODES + (t & 31);
            if (e < items) { o_out = e; n_out = p.tlist ? tlist_c[e] : e; return true; }
        }
    };
    auto load_block = [&](int eb, int e1, float4& geo, int& src, bool& valid) {
        const int e = eb + q;
        valid = e < e1;
        geo = make_float4(0.f, 0.f, 0.f, 0.f);
        src = 0;
        if (valid) { geo = p.e_geom[e]; src = p.e_src[e]; }
    };

    int n = 0, orow = 0, eb = 0, e1 = 0;  // current block
    bool have = fetch_target(n, orow);
    if (have) { eb = nptr_c[n]; e1 = nptr_c[n + 1]; }
    int nN = 0, oN = 0, e0N = 0, e1N = 0;  // next target (bounds requested one target ahead)
    bool haveN = have && fetch_target(nN, oN);
    if (haveN) { e0N = nptr_c[nN]; e1N = nptr_c[nN + 1]; }
    float4 geo = make_float4(0.f, 0.f, 0.f, 0.f); int src = 0; bool valid = false;
    if (have) load_block(eb, e1, geo, src, valid);
    // The loop-carried copies of the first block's edge rows must not be load destinations: hipcc's waitcnt pass merges the
    // "load pending" state of this path into the loop header and then waits vmcnt(0) at the top of EVERY block, right
    // behind the next block's streaming loads (a full HBM round trip per block and wave).  An empty asm that "rewrites" the
    // registers forces the wait here, once.
    asm volatile("" : "+v"(geo.x), "+v"(geo.y), "+v"(geo.z), "+v"(geo.w), "+v"(src));
    bool first = true;                   // current block is the first of its target
    // running sums over the current target's edges: sx = sum a ; s* = sum P*b ; r* = sum c*r_hat
    // (the 1/sqrt3 and 1/sqrtH factors of painn_denoising.py:550-553 are applied once at the end)
    float sx0 = 0.f, sx1 = 0.f, sa0 = 0.f, sa1 = 0.f, sb0 = 0.f, sb1 = 0.f, sc0 = 0.f, sc1 = 0.f;
    float ra0 = 0.f, ra1 = 0.f, rb0 = 0.f, rb1 = 0.f, rc0 = 0.f, rc1 = 0.f;
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;  // residual inputs of this lane's output rows

    while (have) {
        {
            // ---- request what the NEXT block needs
            const bool last = eb + 32 >= e1;
            float4 geoN = make_float4(0.f, 0.f, 0.f, 0.f); int srcN = 0; bool validN = false;
            if (!last) load_block(eb + 32, e1, geoN, srcN, validN);
            else if (haveN) load_block(e0N, e1N, geoN, srcN, validN);
            if (first) {  // residual rows of this target (painn_denoising.py:443-445), used at its last block
                const size_t xo = (size_t)n * H + c0 + q;
                const size_t vo = (size_t)n * 3 * H + c0 + q;
                if (hi == 0) {
                    res0 = p.x[xo]; res1 = p.x[xo + 32];
                    if (!VZ) { res2 = p.vec[vo]; res3 = p.vec[vo + 32]; }
                } else if (!VZ) {
                    res0 = p.vec[vo + H]; res1 = p.vec[vo + H + 32];
                    res2 = p.vec[vo + 2 * H]; res3 = p.vec[vo + 2 * H + 32];
                }
            }

            const float xs = geo.w * p.inv_cutoff;
            // polynomial envelope (radial_basis.py:36-43)
            float xp = xs;
            if (p.env_pi == 5) { const float x2 = xs * xs; xp = x2 * x2 * xs; }  // the shipped exponent, no loop
            else for (int i = 1; i < p.env_pi; ++i) xp *= xs;  // xs^p by repeated multiplication (p is a small int)
            float env = 1.0f + p.env_a * xp + p.env_b * (xp * xs) + p.env_c * (xp * xs * xs);
            env = (xs < 1.0f && valid) ? env : 0.0f;
            __builtin_amdgcn_wave_barrier();  // previous block's meta reads are done
            if (hi == 0) {
                float* m = meta_w + q * 8;
                // byte offset of the source's record row; padded rows gather the all-zero row N
                m[0] = __uint_as_float((unsigned int)(valid ? src : p.N) * row_bytes);  // row N: zero record
                m[1] = geo.x; m[2] = geo.y; m[3] = geo.z;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // k-window of this block
            // a target's edges are sorted by distance (graph.hip), so the block's first and last valid rows
            // bound its window
            const float u = xs * umax_scale;
            const int nvalid = __builtin_amdgcn_readfirstlane(min(32, e1 - eb));
            const float umin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), 0));
            const float umax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), max(nvalid, 1) - 1));
            int klo, khi;
            if (nvalid <= 0) {  // target without incoming edges: one dummy step on all-zero A (env = 0)
                klo = 0; khi = 16;
            } else if (F16) {
                // keep every k with |k - u| < 6 for every row of the block
                klo = max(0, (int)floorf(umin) - 5) & ~7;
                khi = min(p.R, (int)ceilf(umax) + 6);
                khi = klo + ((khi - klo + 15) & ~15);       // whole 16-deep MFMA steps
                if (khi > 128) { klo -= khi - 128; khi = 128; }  // keep the LDS reads inside the image
            } else {
                klo = max(0, (int)floorf(umin) - 7) & ~1;
                khi = min(p.R, ((int)ceilf(umax) + 8 + 1) & ~1);
            }
            klo = __builtin_amdgcn_readfirstlane(klo);
            khi = __builtin_amdgcn_readfirstlane(khi);
            ksteps += (khi - klo) * (VZ && F16 ? 4 : 6);  // contracted k length x 32-column blocks that run (profiling)

            // Gather addresses of the 16 accumulator rows of this lane (32-bit byte offsets).
#define ROW_OF(r) ((r & 3) + 8 * (r >> 2) + 4 * hi)
#define DECL(r)                                                                                 \
    const float* m##r = meta_w + ROW_OF(r) * 8;                                                 \
    float4 ga0##r, ga1##r;                                     /* P0 P1 P2 xa of channel j = 0, 1 */ \
    float gz0##r, gz1##r;                                      /* xc */
#define GATHER(r)                                                                               \
    {                                                                                           \
    const unsigned int o##r = __float_as_uint(m##r[0]);                                         \
    if (MSG_EXP_NOXC) { gz0##r = 1.0f; gz1##r = 2.0f; } else {                                 \
    gz0##r = *reinterpret_cast<const float*>(recS + (size_t)(o##r + qP));                       \
    gz1##r = *reinterpret_cast<const float*>(recS + (size_t)(o##r + qP) + 640); }               \
    if (MSG_EXP_NOXC == 2) { ga0##r = make_float4(1.f, 2.f, 3.f, 4.f); ga1##r = ga0##r; }      \
    else if (!VZ) {                                                                             \
        ga0##r = *reinterpret_cast<const float4*>(recS + (size_t)(o##r + qA));                  \
        ga1##r = *reinterpret_cast<const float4*>(recS + (size_t)(o##r + qA) + 640);            \
    } else {                                                                                    \
        ga0##r = make_float4(0.f, 0.f, 0.f, *reinterpret_cast<const float*>(recS + (size_t)(o##r + qA) + 12)); \
        ga1##r = make_float4(0.f, 0.f, 0.f, *reinterpret_cast<const float*>(recS + (size_t)(o##r + qA) + 652)); \
    }                                                                                           \
    }
#define CONSUME(r)                                                                              \
    {                                                                                           \
        const float ux = m##r[1], uy = m##r[2], uz = m##r[3];                                   \
        const float t3 = gz0##r * acc[4][r];                                                    \
        sx0 += ga0##r.w * acc[0][r];                                                            \
        if (!VZ) { sa0 += ga0##r.x * acc[2][r]; sb0 += ga0##r.y * acc[2][r]; sc0 += ga0##r.z * acc[2][r]; } \
        ra0 += t3 * ux; rb0 += t3 * uy; rc0 += t3 * uz;                                         \
        const float u3 = gz1##r * acc[5][r];                                                    \
        sx1 += ga1##r.w * acc[1][r];                                                            \
        if (!VZ) { sa1 += ga1##r.x * acc[3][r]; sb1 += ga1##r.y * acc[3][r]; sc1 += ga1##r.z * acc[3][r]; } \
        ra1 += u3 * ux; rb1 += u3 * uy; rc1 += u3 * uz;                                         \
    }
            DECL(0) DECL(1) DECL(2) DECL(3) DECL(4) DECL(5) DECL(6) DECL(7)
            DECL(8) DECL(9) DECL(10) DECL(11) DECL(12) DECL(13) DECL(14) DECL(15)
            // rows 0-3: issued before the MFMA loop, they land while the matrix pipe is busy
            GATHER(0) GATHER(1)
#if FWD_AHEAD == 4
            GATHER(2) GATHER(3)
#endif

            const float env256 = env * 256.0f;
            f32x16 acc[6];
            if (F16) {
                // acc = 256 * (b_hi + b_lo): A = 256 at k slots 0 and 1 (half-wave 0 only), B = image slots 128.. of each column
                half8 aone = {0, 0, 0, 0, 0, 0, 0, 0};
                if (hi == 0) { aone[0] = (_Float16)256.0f; aone[1] = (_Float16)256.0f; }
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    if (VZ && (b == 2 || b == 3)) continue;
                    const half8 bb = *reinterpret_cast<const half8*>(Wh + (size_t)(b * 32 + q) * MSG_LDK + 128);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aone, bb, zero16, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const float bv = Bl[b * 32 + q];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][r] = bv;
                }
            }
            if (F16) {
                const float xsq = xs * p.sarg;
                // A fragment of one 16-deep step: lane (row q, half hi) holds k = k0 + 8*hi + j, j = 0..7
                auto gen_a = [&](int k0, half8& ah, half8& al) {
                    if constexpr (UNI) {
                        // a_j = E exp2(-(t - j d)^2), t = xs' - mu'_{k0+8hi}:  a_{j+1} = a_j r_j,  r_j = exp2(2 d (t - j d) - d^2),
                        // r_{j+1} = r_j exp2(-2 d^2).  Two exp2 + 16 multiplications instead of 8 x (sub, mul, exp2, mul); the
                        // products carry <= 14 roundings (1e-6 relative, the size of the f16x3 split error).  If a_0
                        // underflows, every a_j of the group is below 2^-39 of the row's leading term (dropped like the
                        // terms outside the k-window).  hi/lo split: packed round-to-zero conversions (lo absorbs the
                        // truncation exactly) - 4 instructions per pair instead of 8.
                        const float t0 = xsq - Mu[k0 + 8 * hi];
                        float a = env256 * __builtin_amdgcn_exp2f(-(t0 * t0));
                        // (clamped: far right of the group r would overflow while a_0 has long underflowed to 0: 0 * inf)
                        float r = __builtin_amdgcn_exp2f(fminf(p.dmu2 * t0 - p.dmusq, 64.0f));
                        float av[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            av[j] = a;
                            a *= r;
                            r *= p.cstep;
                        }
#pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
                            const h2_t hh = __builtin_amdgcn_cvt_pkrtz(av[j], av[j + 1]);
                            const h2_t ll = __builtin_amdgcn_cvt_pkrtz(av[j] - (float)hh[0], av[j + 1] - (float)hh[1]);
                            ah[j] = (_Float16)hh[0]; ah[j + 1] = (_Float16)hh[1];
                            al[j] = (_Float16)ll[0]; al[j + 1] = (_Float16)ll[1];
                        }
                    } else {
                        const float4 mu0 = *reinterpret_cast<const float4*>(Mu + k0 + 8 * hi);
                        const float4 mu1 = *reinterpret_cast<const float4*>(Mu + k0 + 8 * hi + 4);
                        const float mus[8] = {mu0.x, mu0.y, mu0.z, mu0.w, mu1.x, mu1.y, mu1.z, mu1.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float dm = xsq - mus[j];
                            // a in [0,1] is lifted by 2^8 before the split so that a_lo is a normal fp16 number
                            // for every term that matters (the matrix core flushes fp16 subnormals)
                            const float a = env256 * __builtin_amdgcn_exp2f(-(dm * dm));
                            const _Float16 h = (_Float16)a;
                            ah[j] = h;
                            al[j] = (_Float16)(a - (float)h);
                        }
                    }
                };
                int k0 = klo;
                if (MSG_ABL != 1)
                do {  // at least one step: lets the accumulators live in place across the loop
                    // (generating the NEXT step's operand beside this step's MFMAs was measured: 2-3 % slower - the other wave
                    // of the SIMD already fills this wave's operand phase with its own MFMAs)
                    half8 ah, al;
                    gen_a(k0, ah, al);
                    const _Float16* wh = Wh + (size_t)q * MSG_LDK + k0 + 8 * hi;
                    const _Float16* wl = Wlo + (size_t)q * MSG_LDK + k0 + 8 * hi;
#pragma unroll
                    for (int b = 0; b < 6; ++b) {
                        if (VZ && (b == 2 || b == 3)) continue;   // vec == 0: the xb columns multiply P = vec * xb = 0
                        const half8 bh = *reinterpret_cast<const half8*>(wh + b * 32 * MSG_LDK);
                        const half8 bl = *reinterpret_cast<const half8*>(wl + b * 32 * MSG_LDK);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
                    }
                    // spread the next operand's ~45 vector instructions and this step's B-fragment reads evenly behind the
                    // MFMAs (left alone, hipcc packs them behind the last five)
                    k0 += 16;
                } while (k0 < khi);
            } else {
                int k2 = klo;
                do {
                    const int k = k2 + hi;
                    const float dm = xs - Mu[k];
                    // exp via v_exp_f32 (exp2): |arg| <= 24.5 inside the window, relative error <= ~2e-6
                    // on the smallest kept terms and ~1e-7 on the leading ones
                    const float a = env * __builtin_amdgcn_exp2f(coeff2 * (dm * dm));
                    const float* wrow = Wl + k * MSG_COLS + q;
#pragma unroll
                    for (int b = 0; b < 6; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wrow[b * 32], acc[b], 0, 0, 0);
                    k2 += 2;
                } while (k2 < khi);
            }
            // epilogue, software pipelined: next rows' gathers are in flight while rows are consumed
#if MSG_ABL == 2
#pragma unroll
            for (int b = 0; b < 6; ++b) asm volatile("" :: "v"(acc[b]));
            sx0 += m0[1];
#elif FWD_AHEAD == 2
            GATHER(2) GATHER(3)
            CONSUME(0) CONSUME(1)
            GATHER(4) GATHER(5)
            CONSUME(2) CONSUME(3)
            GATHER(6) GATHER(7)
            CONSUME(4) CONSUME(5)
            if (nvalid <= 16) {
                CONSUME(6) CONSUME(7)
            } else {
                GATHER(8) GATHER(9)
                CONSUME(6) CONSUME(7)
                GATHER(10) GATHER(11)
                CONSUME(8) CONSUME(9)
                GATHER(12) GATHER(13)
                CONSUME(10) CONSUME(11)
                GATHER(14) GATHER(15)
                CONSUME(12) CONSUME(13)
                CONSUME(14) CONSUME(15)
            }
#elif MSG_SKIP
            // accumulator rows 4g..4g+3 of a lane are block rows 8g..8g+7: a block with nvalid rows has nothing but padding
            // beyond group (nvalid-1)/8 - its gathers (of the zero record) and sums are skipped (wave-uniform branches; about
            // 22 % of all block rows are padding: 50 edges per target in 32-row blocks)
            GATHER(4) GATHER(5) GATHER(6) GATHER(7)
            CONSUME(0) CONSUME(1) CONSUME(2) CONSUME(3)
            if (nvalid <= 16) {
                CONSUME(4) CONSUME(5) CONSUME(6) CONSUME(7)
            } else {
                GATHER(8) GATHER(9) GATHER(10) GATHER(11)
                CONSUME(4) CONSUME(5) CONSUME(6) CONSUME(7)
#if MSG_SKIP >= 2
                // (round 6, measured, not the default) 17..24 valid rows - the usual tail of a target with ~50 in-edges: rows
                // 24..31 are padding.  250 registers, no spills; 6.95-7.00 ms against 7.00-7.05 per full launch: inside the
                // noise, like the <= 16 skip before it - the kernel's time does not follow its gather / VALU counts.
                if (nvalid <= 24) {
                    CONSUME(8) CONSUME(9) CONSUME(10) CONSUME(11)
                } else
#endif
                {
                    GATHER(12) GATHER(13) GATHER(14) GATHER(15)
                    CONSUME(8) CONSUME(9) CONSUME(10) CONSUME(11)
                    CONSUME(12) CONSUME(13) CONSUME(14) CONSUME(15)
                }
            }
#else
            GATHER(4) GATHER(5) GATHER(6) GATHER(7)
            CONSUME(0) CONSUME(1) CONSUME(2) CONSUME(3)
            GATHER(8) GATHER(9) GATHER(10) GATHER(11)
            CONSUME(4) CONSUME(5) CONSUME(6) CONSUME(7)
            GATHER(12) GATHER(13) GATHER(14) GATHER(15)
            CONSUME(8) CONSUME(9) CONSUME(10) CONSUME(11)
            CONSUME(12) CONSUME(13) CONSUME(14) CONSUME(15)
#endif
#undef GATHER
#undef DECL
#undef CONSUME
#undef ROW_OF
            if (last) {
                // ---- finish this target: scale, cross-half reduction, residuals, one write per row
                sx0 *= out_scale; sx1 *= out_scale;
                sa0 = (sa0 * inv_sqrt3 + ra0) * inv_sqrt_h; sa1 = (sa1 * inv_sqrt3 + ra1) * inv_sqrt_h;
                sb0 = (sb0 * inv_sqrt3 + rb0) * inv_sqrt_h; sb1 = (sb1 * inv_sqrt3 + rb1) * inv_sqrt_h;
                sc0 = (sc0 * inv_sqrt3 + rc0) * inv_sqrt_h; sc1 = (sc1 * inv_sqrt3 + rc1) * inv_sqrt_h;
                // the two half-waves hold disjoint rows of the same channels
                sx0 += __shfl_xor(sx0, 32); sx1 += __shfl_xor(sx1, 32);
                sa0 += __shfl_xor(sa0, 32); sa1 += __shfl_xor(sa1, 32);
                sb0 += __shfl_xor(sb0, 32); sb1 += __shfl_xor(sb1, 32);
                sc0 += __shfl_xor(sc0, 32); sc1 += __shfl_xor(sc1, 32);
                // lane q owns channels c0+q and c0+32+q; half-wave 0 writes x and vec_x, half-wave 1 vec_y, vec_z
                const size_t xo = (size_t)orow * H + c0 + q;
                const size_t vo = (size_t)orow * 3 * H + c0 + q;
                if (hi == 0) {
                    p.x_out[xo] = (res0 + sx0) * inv_sqrt2;
                    p.x_out[xo + 32] = (res1 + sx1) * inv_sqrt2;
                    p.vec_out[vo] = res2 + sa0;
                    p.vec_out[vo + 32] = res3 + sa1;
                } else {
                    p.vec_out[vo + H] = res0 + sb0;
                    p.vec_out[vo + H + 32] = res1 + sb1;
                    p.vec_out[vo + 2 * H] = res2 + sc0;
                    p.vec_out[vo + 2 * H + 32] = res3 + sc1;
                }
                sx0 = sx1 = sa0 = sa1 = sb0 = sb1 = sc0 = sc1 = 0.f;
                ra0 = ra1 = rb0 = rb1 = rc0 = rc1 = 0.f;
                res0 = res1 = res2 = res3 = 0.f;
                // ---- advance to the next target and request the bounds of the one after it
                have = haveN;
                n = nN; orow = oN; eb = e0N; e1 = e1N;
                first = true;
                if (have) {
                    haveN = fetch_target(nN, oN);
                    if (haveN) { e0N = nptr_c[nN]; e1N = nptr_c[nN + 1]; }
                }
            } else {
                eb += 32;
                first = false;
            }
            geo = geoN; src = srcN; valid = validN;
        }
    }
    if (p.kcount && lane == 0) atomicAdd(p.kcount, (unsigned long long)ksteps);
}

// rbf_proj -> [slice][k][part*64 + j*32 + q]  with channel = slice*64 + 32j + q
__global__ void adf_pack_rbf_kernel(const float* __restrict__ w, const float* __restrict__ b, float* wpack,
                                    float* bpack, int H, int R) {
    const int nslices = H / ADF_SLICE_CH;
    const int total = nslices * R * MSG_COLS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int col = i % MSG_COLS;
        const int k = (i / MSG_COLS) % R;
        const int slice = i / (MSG_COLS * R);
        const int part = col / 64, j = (col >> 5) & 1, qq = col & 31;
        const int ch = slice * ADF_SLICE_CH + 32 * j + qq;
        wpack[i] = w[(size_t)(part * H + ch) * R + k];
        if (k == 0) bpack[slice * MSG_COLS + col] = b[part * H + ch];
    }
}

// f16 image: [slice][hi|lo][col][k] halves of (w * scale), bias16 = bias * scale.
// scale = power of two with |w*scale| in [2^9, 2^10)  (absmax from adf_absmax_kernel, gemm16.hip)
__global__ void adf_pack_rbf16_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                      const unsigned int* absmax_bits, _Float16* wpack16, float* bpack16,
                                      float* inv_scale, int H, int R) {
    const float amax = __uint_as_float(*absmax_bits);
    int e = 0;
    if (amax > 0.f) (void)frexpf(amax, &e);
    const float scale = ldexpf(1.0f, 10 - e);
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale = 1.0f / scale;
    const int nslices = H / ADF_SLICE_CH;
    const int total = nslices * MSG_COLS * R;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k = i % R;
        const int col = (i / R) % MSG_COLS;
        const int slice = i / (R * MSG_COLS);
        const int part = col / 64, j = (col >> 5) & 1, qq = col & 31;
        const int ch = slice * ADF_SLICE_CH + 32 * j + qq;
        const float v = w[(size_t)(part * H + ch) * R + k] * scale;
        const _Float16 h = (_Float16)v;
        const size_t base = (size_t)slice * 2 * MSG_COLS * R;
        wpack16[base + (size_t)col * R + k] = h;
        wpack16[base + (size_t)(MSG_COLS + col) * R + k] = (_Float16)(v - (float)h);
        if (k == 0) {  // bias * scale as an fp16 hi/lo pair in the 4 bytes of the column's slot
            const float bv = b[part * H + ch] * scale;
            const _Float16 bh = (_Float16)bv;
            _Float16* o = reinterpret_cast<_Float16*>(bpack16 + slice * MSG_COLS + col);
            o[0] = bh; o[1] = (_Float16)(bv - (float)bh);
        }
    }
}

__global__ void adf_absmax_kernel(const float* __restrict__ w, long long n, unsigned int* out_bits);

// Gather records of the message kernel (stand-alone producer, used by the exact-f32 mode; the f16x3
// mode writes the same layout from the x_proj.2 GEMM epilogue, gemm16.hip EPI 1).  Per source atom n and
// group g of 32 channels, 640 B:  [32 lanes][P0(c), P1(c), P2(c), xa(c)]  then  [32 lanes][xc(c)],
// c = 32 g + lane, xa/xb/xc = the three H-wide parts of xh, P_i = vec_i * xb.  vec*xb is the only place
// vec[src] and xb[src] enter the message (painn_denoising.py:549-552), so the per-edge gather shrinks
// from 6 to 5 floats per channel and every piece a half-wave reads is one contiguous run of the source row.
__global__ void adf_pack_records_kernel(const float* __restrict__ xh, const float* __restrict__ vec,
                                        float* __restrict__ rec, int N, int H, int vec_is_zero) {
    const int ng = H / 32;
    const long long total = (long long)N * ng * 32;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int qq = (int)(t & 31);
        const int g = (int)((t >> 5) % ng);
        const long long n = t / (32LL * ng);
        const int c = 32 * g + qq;
        const float* xr = xh + (size_t)n * 3 * H;
        const float xa = xr[c], xb = xr[H + c], xc = xr[2 * H + c];
        float* out = rec + ((size_t)n * ng + g) * 160;
        if (!vec_is_zero) {
            const float* vr = vec + (size_t)n * 3 * H;
            reinterpret_cast<float4*>(out)[qq] = make_float4(vr[c] * xb, vr[H + c] * xb, vr[2 * H + c] * xb, xa);
        } else {
            reinterpret_cast<float4*>(out)[qq] = make_float4(0.f, 0.f, 0.f, xa);
        }
        out[128 + qq] = xc;
    }
}

int32_t adf_pack_records(adf_painn* h, int N, const float* xh, const float* vec, bool vec_is_zero, hipStream_t s,
                         float* rec) {
    const int H = h->hp.hidden_channels;
    long long blocks = ((long long)N * H + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(adf_pack_records_kernel, dim3((unsigned)blocks), dim3(256), 0, s, xh, vec, rec ? rec : h->rec, N, H,
                       vec_is_zero ? 1 : 0);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

static size_t msg_lds_bytes(int R, bool f16) {
    const size_t w = f16 ? (size_t)2 * MSG_COLS * MSG_LDK * 2 : sizeof(float) * (size_t)R * MSG_COLS;
    return w + sizeof(float) * (MSG_COLS + 144 + FWD_WAVES * 32 * 8) + 16;
}

// The message kernel's images of ONE layer's rbf_proj (f32 image, fp16 hi/lo image, bias images, scale) from the weight
// tensors the handle is bound to.  adf_pack_rbf calls it for every layer at adf_painn_set_weights; the training step calls
// it per layer and step (the optimizer updates the bound tensors in place).
int32_t adf_pack_rbf_layer(adf_painn* h, int l, hipStream_t s) {
    const int H = h->hp.hidden_channels, R = h->hp.num_rbf;
    const size_t per_layer = (size_t)(H / ADF_SLICE_CH) * R * MSG_COLS;
    const size_t per_layer_b = (size_t)(H / ADF_SLICE_CH) * MSG_COLS;
    hipLaunchKernelGGL(adf_pack_rbf_kernel, dim3(256), dim3(256), 0, s, h->layer[l].rbf_w, h->layer[l].rbf_b,
                       h->rbf_pack + l * per_layer, h->rbf_bias_pack + l * per_layer_b, H, R);
    ADF_HIP_CHECK(hipMemsetAsync(h->w16_scratch, 0, sizeof(unsigned int), s));
    hipLaunchKernelGGL(adf_absmax_kernel, dim3(64), dim3(256), 0, s, h->layer[l].rbf_w, (long long)3 * H * R,
                       h->w16_scratch);
    // the bias shares the scale (it enters the same accumulators through the matrix core)
    hipLaunchKernelGGL(adf_absmax_kernel, dim3(8), dim3(256), 0, s, h->layer[l].rbf_b, (long long)3 * H,
                       h->w16_scratch);
    hipLaunchKernelGGL(adf_pack_rbf16_kernel, dim3(256), dim3(256), 0, s, h->layer[l].rbf_w, h->layer[l].rbf_b,
                       h->w16_scratch, reinterpret_cast<_Float16*>(h->rbf_pack16) + l * 2 * per_layer,
                       h->rbf_bias_pack16 + l * per_layer_b, h->rbf_scales + l, H, R);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t adf_pack_rbf(adf_painn* h, hipStream_t s) {
    const int H = h->hp.hidden_channels, R = h->hp.num_rbf;
    {   // are the Gaussian centres the linspace(0, 1, R) of GaussianSmearing?  (a buffer, but a checkpoint may carry another)
        float mu[128];
        ADF_HIP_CHECK(hipMemcpyAsync(mu, h->rbf_offset, sizeof(float) * R, hipMemcpyDeviceToHost, s));
        ADF_HIP_CHECK(hipStreamSynchronize(s));
        bool uni = true;
        for (int k = 0; k < R; ++k) uni = uni && fabsf(mu[k] - (float)k / (float)(R - 1)) <= 2e-7f;
        const char* e = getenv("ADF_MSG_RBF");
        h->rbf_uniform = uni && !(e && strcmp(e, "direct") == 0);
    }
    for (int l = 0; l < h->hp.num_layers; ++l) ADF_TRY(adf_pack_rbf_layer(h, l, s));
    ADF_HIP_CHECK(hipGetLastError());
    {   // per device (a function attribute is per device; set_weights is rare, so no caching)
#define SET_LDS(F16_, VZ_, UNI_)                                                                              \
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(adf_message_kernel<F16_, VZ_, UNI_>),     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)msg_lds_bytes(128, F16_)))
        SET_LDS(false, false, false); SET_LDS(false, true, false); SET_LDS(true, false, false); SET_LDS(true, true, false);
        SET_LDS(true, false, true); SET_LDS(true, true, true);
#undef SET_LDS
    }
    return ADF_OK;
}

int32_t adf_message_impl(adf_painn* h, int layer, int N, const float* x, const float* xh, const float* vec,
                         float* x_out, float* vec_out, bool vec_is_zero, hipStream_t s, const int32_t* tlist,
                         int n_targets, const float* rec, const int32_t* n_targets_dev) {
    const int H = h->hp.hidden_channels, R = h->hp.num_rbf;
    if ((unsigned long long)(N + 1) * 5ull * H * sizeof(float) >= (1ull << 32)) {
        adf_set_error("message kernel uses 32-bit byte offsets into the node tables: N=%d is too large, split the batch", N);
        return ADF_EOOM;  // surfaces as RuntimeError -> ml_diffuse splits the batch
    }
    MsgParams p;
    p.rec = rec ? rec : h->rec; p.vec = vec; p.x = x; p.x_out = x_out; p.vec_out = vec_out;
    p.nptr = h->nptr; p.e_src = h->e_src; p.e_geom = h->e_geom;
    p.nslices = H / ADF_SLICE_CH;
    const bool f16 = !h->msg_f32;
    p.wpack = h->rbf_pack + (size_t)layer * p.nslices * R * MSG_COLS;
    p.wpack16 = reinterpret_cast<const _Float16*>(h->rbf_pack16) + (size_t)layer * 2 * p.nslices * R * MSG_COLS;
    p.bpack = (f16 ? h->rbf_bias_pack16 : h->rbf_bias_pack) + (size_t)layer * p.nslices * MSG_COLS;
    p.inv_scale = h->rbf_scales + layer;
    p.mu = h->rbf_offset;
    p.N = N; p.H = H; p.R = R;
    p.tlist = tlist; p.items = tlist ? n_targets : N; p.items_dev = tlist ? n_targets_dev : nullptr;
    if (p.items <= 0) return ADF_OK;
    p.G = (p.items + ADF_GROUP_NODES - 1) / ADF_GROUP_NODES;
    p.inv_cutoff = 1.0f / h->hp.cutoff;
    const double step = 1.0 / (R - 1);
    p.coeff = (float)(-0.5 / (step * step));
    p.sarg = (float)sqrt(0.5 / (step * step) * 1.4426950408889634);  // exp(coeff z^2) = exp2(-(sarg z)^2)
    const double pe = (double)h->hp.envelope_exponent;
    p.env_pi = h->hp.envelope_exponent;
    p.env_a = (float)(-(pe + 1) * (pe + 2) / 2);
    p.env_b = (float)(pe * (pe + 2));
    p.env_c = (float)(-pe * (pe + 1) / 2);
    p.kcount = h->prof_on ? h->kcount : nullptr;
    int workers = h->num_cus / p.nslices;
    if (workers < 1) workers = 1;
    if (workers > p.G) workers = p.G;
    dim3 grid((unsigned)(workers * p.nslices));
#define LAUNCH_MSG(F16_, VZ_, UNI_)                                                                        \
    hipLaunchKernelGGL((adf_message_kernel<F16_, VZ_, UNI_>), grid, dim3(FWD_THREADS), msg_lds_bytes(R, F16_), s, p)
    {   // scaled spacing of the (equally spaced) centres: mu_k = k / (R - 1)
        const double d = sqrt(0.5 / (step * step) * 1.4426950408889634) * step;
        p.dmu2 = (float)(2.0 * d); p.dmusq = (float)(d * d); p.cstep = (float)exp2(-2.0 * d * d);
    }
    if (f16 && h->rbf_uniform) { if (vec_is_zero) LAUNCH_MSG(true, true, true); else LAUNCH_MSG(true, false, true); }
    else if (f16) { if (vec_is_zero) LAUNCH_MSG(true, true, false); else LAUNCH_MSG(true, false, false); }
    else { if (vec_is_zero) LAUNCH_MSG(false, true, false); else LAUNCH_MSG(false, false, false); }
#undef LAUNCH_MSG
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
