// PaiNN message block, round-4 kernel: the same fused computation as message.hip (radial-basis projection on the
// matrix cores + gather + gated equivariant message + per-target segmented sum + residual; reference
// painn_denoising.py:530-567, :443-445, gemnet_oc/layers/radial_basis.py:18-43,64-82), re-organised so that every wave
// issues its matrix-core instructions INTERLEAVED with the vector work of the neighbouring accumulators.
//
// Why.  message.hip alternates, per 32-edge block, an MFMA phase (k-window contraction into 6 accumulators) with a VALU
// phase (16 gathered rows x 8 FMA per channel pair) and relies on the SIMD's second wave being in the other phase.  On
// gfx950 that does not overlap (profiles/r02/r02_issue_rates.txt: "split roles" 118 cycles per MFMA against 44 for the same
// mix issued from ONE stream), so its launch time is MFMA time + VALU time (MfmaUtil 38 %, VALUBusy 47 %, adding up).
// What does overlap is a wave's own stream with ~6 vector instructions between two MFMAs.  Hence, per block t:
//     P1: MFMA chains a0, b0 of t   ||  consume c0, c1 of t-1   (xc pieces, r_hat)      + W stage of t+1, R stage of t+2,
//                                                                                          A operand of k-steps 2, 3 of t
//     P2: MFMA chains a1, b1 of t   ||  consume a0, b0 of t     ((P0,P1,P2,xa) of j=0)
//     P3: MFMA chains c0, c1 of t   ||  consume a1, b1 of t     ((P0,P1,P2,xa) of j=1)  + A operand of k-steps 0, 1 of t+1
//                                                                                          (in place, behind their last use)
// (j = the lane's channel c0+q / c0+32+q; a/b/c = the three H-wide parts of rbf_proj.)  The A operand of all k-steps of a
// block is held in registers, an accumulator pair is consumed while the next pair is contracted (4 accumulators live
// instead of 6), and a gathered record piece is needed by ONE phase only, so it is requested a few rows ahead of its
// consumption into a short register ring (no 160-register landing zone) - which is what lets two such waves share a SIMD
// (256 registers each) and cover each other's LDS / L2 round trips.
// What was measured on the way (round-4 notes in DESIGN.md): (i) one wave per SIMD with 512 registers does not help - only
// 256 of them are addressable by vector instructions, accumulators in AGPRs cost one v_accvgpr_read per consumed value;
// (ii) narrower pieces (dword xa + dwordx3 P) to shrink the landing zone run 1.6x SLOWER than message.hip: strided narrow
// loads touch every 64-B sector of the 16-B-stride piece again (1.8x the L1 traffic), and the kernel's floor is the
// 64 B/clk/CU of L1 fill bandwidth (2560 B per row and slice); (iii) three bodies specialised on the k-step count cost
// ~250 register moves per block at the loop's back edge, and wave-uniform branches around the MFMAs of k-steps 2, 3 spill
// (both versions of an accumulator stay live) - ONE body that always runs 4 k-steps, the missing ones on a zero A operand.
// Arithmetic: message.hip's f16x3 mode with equally spaced Gaussian centres (same A-operand recurrence, same MFMA order per
// accumulator, same order of the in-lane row sums); outputs agree to ~2e-7 relative.  Blocks are cut short so that their
// k-window never exceeds 64 (4 k-steps; +0.2 % MFMAs on the benchmark graph).  The exact-f32 and non-uniform-centre modes stay on
// message.hip.
#include <stdlib.h>
#include <string.h>

#include "message.h"

#define M3_THREADS 512
#define M3_WAVES 8
#define M3_ULIM 44.0f  // a block takes rows while u <= u_first + 44 (u = (R-1) d / rc): window <= 64 after alignment
#ifndef M3_D
#define M3_D 4         // (P0,P1,P2,xa) pieces: rows requested ahead of the consumed one
#endif
#ifndef M3_DC
#define M3_DC 8        // xc pieces (2 registers per row): deeper - the first 8 rows of P1 then only wait for loads that are
                       // OLDER than the block's streaming loads (edge rows of block t+2, residual rows), see body()
#endif
#ifndef M3_EXP
#define M3_EXP 0       // timing experiments only (wrong results): 1 no edge-row / residual loads in the loop; 2 also no gathers;
#endif                 // 3 every gather reads the all-zero record (L1 hits)

typedef __fp16 m3_h2 __attribute__((ext_vector_type(2)));

// End of one issue slot ("tick": one MFMA + its share of the phase's vector work).  Nothing but LDS reads and scalar
// instructions may be moved across it: the MFMA / VALU / vector-memory interleave is the one written in the source
// (hipcc otherwise issues the dependent MFMAs of a chain back to back and the vector work behind them - no overlap).
#ifndef M3_TICK_MASK
#define M3_TICK_MASK 0x104
#endif
#define M3_TICK_END() __builtin_amdgcn_sched_barrier(M3_TICK_MASK)

struct M3Block {  // wave-uniform
    int n, orow, eb, e1, nrows, last, klo, nk, have;
};

template <bool VZ>
__global__ __launch_bounds__(M3_THREADS, 2) void adf_message3_kernel(MsgParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // carve: weight image (hi rows, lo rows) | [128] mu | [8 waves][3 buffers][32 rows][4] row meta
    _Float16* Wh = reinterpret_cast<_Float16*>(lds);
    float* Mu = lds + (2 * MSG_COLS * MSG_LDK) / 2;
    float* Meta = Mu + 128;

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
    if (worker >= ngroups) return;
    const int H = p.H;
    const int c0 = slice * ADF_SLICE_CH;

    {   // stage this slice's rbf_proj image once (same image as message.hip: bias as an fp16 hi/lo pair in k slots 128/129)
        const int R8 = p.R / 8;
        const half8* src = reinterpret_cast<const half8*>(p.wpack16 + (size_t)slice * 2 * MSG_COLS * p.R);
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const _Float16* b16 = reinterpret_cast<const _Float16*>(p.bpack) + (size_t)slice * MSG_COLS * 2;
        for (int i = tid; i < 2 * MSG_COLS * 17; i += M3_THREADS) {
            const int row = i / 17, piece = i - row * 17;
            half8 v = piece < R8 ? src[row * R8 + piece] : zero8;
            if (piece == 16 && row < MSG_COLS) { v[0] = b16[2 * row]; v[1] = b16[2 * row + 1]; }
            *reinterpret_cast<half8*>(Wh + (size_t)row * MSG_LDK + piece * 8) = v;
        }
        if (tid < 128) Mu[tid] = (tid < p.R ? p.mu[tid] : 2.0f) * p.sarg;
    }
    __syncthreads();

    const float inv_sqrt3 = 0.57735026918962576f;
    const float inv_sqrt2 = 0.70710678118654752f;
    const float out_scale = *p.inv_scale * (1.0f / 256.0f);
    const float inv_sqrt_h = out_scale / sqrtf((float)H);
    const float umax_scale = (float)(p.R - 1);
    const unsigned int row_bytes = (unsigned int)p.nslices * 1280u;
    const char* rec_base = reinterpret_cast<const char*>(p.rec) + (size_t)slice * 1280;  // uniform
    const unsigned int lo4 = (unsigned int)q * 16u;         // lane offset of the (P0,P1,P2,xa) piece
    const unsigned int lo1 = 512u + (unsigned int)q * 4u;   // lane offset of the xc piece
    // LDS byte addresses (per lane, constant): B fragments, bias fragments, row meta of this wave
    const _Float16* whq = Wh + (size_t)q * MSG_LDK + 8 * hi;            // + klo + cb*32*LDK + 16 ks
    const _Float16* wlq = whq + (size_t)MSG_COLS * MSG_LDK;
    const _Float16* wbias = Wh + (size_t)q * MSG_LDK + 128;             // + cb*32*LDK
    float* meta_w = Meta + wave * 3 * 32 * 4;                           // [buffer][row][4]: blocks t-1, t, t+1
    const float* mu_h = Mu + 8 * hi;

    // ---- static work assignment: item it = wave, wave + 4, ... of this workgroup's sequence
    //      item -> group worker + (it / 32) * nworkers, atom it % 32  (consecutive items: consecutive atoms)
    auto item_target = [&](int it, int& n_out, int& o_out, int& have_out) {
        const int g = worker + (it >> 5) * nworkers;
        const int e = g * ADF_GROUP_NODES + (it & 31);
        have_out = g < ngroups ? 1 : 0;
        const int ok = (have_out && e < items) ? 1 : 0;    // a hole in the last group: a target without edges and without output
        const int ec = ok ? e : 0;
        o_out = ok ? e : -1;
        n_out = p.tlist ? __builtin_amdgcn_readfirstlane(p.tlist[ec]) : ec;
    };
    unsigned int ksteps = 0;

    // ---- per-lane state
    f32x16 acc[6];
    half8 Ah[4], Al[4];
    float4 g4[16];             // ring of gathered (P0,P1,P2,xa) pieces: row r lives from its request (M3_D rows ahead) to its use
    float gza[16], gzb[16];    // ring of gathered xc pieces (j = 0, 1)
    float sx0 = 0.f, sx1 = 0.f, sa0 = 0.f, sa1 = 0.f, sb0 = 0.f, sb1 = 0.f, sc0 = 0.f, sc1 = 0.f;
    float ra0 = 0.f, ra1 = 0.f, rb0 = 0.f, rb1 = 0.f, rc0 = 0.f, rc1 = 0.f;
    float resP[4] = {0.f, 0.f, 0.f, 0.f}, resC[4] = {0.f, 0.f, 0.f, 0.f};   // residual rows of the previous / current block's target
    float xsqC = 0.f, envC = 0.f, xsqN = 0.f, envN = 0.f;
    float4 geoR = make_float4(0.f, 0.f, 0.f, 0.f);   // rows of the block requested last (R stage)
    int srcR = 0;

#define M3_ROW(r) (((r) & 3) + 8 * ((r) >> 2))   // + 4 * hi: accumulator row r of this lane

    // ---- pipeline stages (lambdas; everything is inlined)
    // residual rows of target n (painn_denoising.py:443-445): half-wave 0 holds x, vec_x; half-wave 1 vec_y, vec_z
    auto load_res = [&](int n, float* r4) {
        const unsigned int xo = ((unsigned int)n * H + c0 + q) * 4u;            // byte offsets (N * 3H * 4 < 2^32, checked by the host)
        const unsigned int vo = ((unsigned int)n * 3 * H + c0 + q) * 4u;
        const char* xb = reinterpret_cast<const char*>(p.x);
        const char* vb = reinterpret_cast<const char*>(p.vec);
        if (VZ) {
            r4[0] = *reinterpret_cast<const float*>(xb + xo); r4[1] = *reinterpret_cast<const float*>(xb + xo + 128);
            r4[2] = 0.f; r4[3] = 0.f;   // half-wave 1's copy is not used
        } else {
            const char* a = hi ? vb + (size_t)H * 4 : xb;
            const unsigned int ao = hi ? vo : xo;
            r4[0] = *reinterpret_cast<const float*>(a + ao); r4[1] = *reinterpret_cast<const float*>(a + ao + 128);
            const unsigned int bo = vo + (hi ? 2u * H * 4u : 0u);
            r4[2] = *reinterpret_cast<const float*>(vb + bo); r4[3] = *reinterpret_cast<const float*>(vb + bo + 128);
        }
    };
    // R stage: request the 32 edge rows starting at eb (clamped: rows beyond e1 re-read the last one and are masked later)
    auto request_rows = [&](int eb, int e1) {
        int e = min(eb + q, e1 - 1);
        e = max(e, 0);
        geoR = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.e_geom) + (size_t)((unsigned int)e * 16u));
        srcR = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(p.e_src) + (size_t)((unsigned int)e * 4u));
    };
    // W stage: the requested rows have arrived -> rows taken, k-window, per-lane A-operand inputs, row meta in LDS
    auto window_stage = [&](M3Block& d, int buf, float& xsq_out, float& env_out) {
        const float xs = geoR.w * p.inv_cutoff;
        const float u = xs * umax_scale;
        const int navail = min(32, d.e1 - d.eb);
        const float u0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(u)));
        const bool take = q < navail && u <= u0 + M3_ULIM;      // rows are sorted by distance: a prefix
        const unsigned long long bal = __ballot(take);
        const int nrows = __builtin_popcount((unsigned int)bal);
        d.nrows = __builtin_amdgcn_readfirstlane(nrows);
        d.last = d.eb + d.nrows >= d.e1 ? 1 : 0;
        const bool valid = q < nrows;
        const float ulast = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), max(nrows, 1) - 1));
        int klo = max(0, (int)floorf(u0) - 5) & ~7;
        int khi = min(p.R, (int)ceilf(ulast) + 6);
        khi = klo + max(32, (khi - klo + 15) & ~15);             // whole 16-deep MFMA steps, at least two
        if (khi > 128) { klo -= khi - 128; khi = 128; }
        if (nrows <= 0) { klo = 0; khi = 32; }
        d.klo = __builtin_amdgcn_readfirstlane(klo);          // wave-uniform: keep the block descriptors in SGPRs
        d.nk = __builtin_amdgcn_readfirstlane((khi - klo) >> 4);
        float xp = xs;
        for (int i = 1; i < p.env_pi; ++i) xp *= xs;
        float env = 1.0f + p.env_a * xp + p.env_b * (xp * xs) + p.env_c * (xp * xs * xs);
        env = (xs < 1.0f && valid) ? env : 0.0f;
        env_out = env * 256.0f;
        xsq_out = xs * p.sarg;
        float4 m;
        m.x = __uint_as_float((unsigned int)(valid ? srcR : p.N) * row_bytes);   // row N: the all-zero record
        m.y = geoR.x; m.z = geoR.y; m.w = geoR.z;
        *reinterpret_cast<float4*>(meta_w + buf * 128 + q * 4) = m;               // both half-waves hold (and write) row q
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    // A operand of k-step ks (k = klo + 16 ks + 8 hi + j): same recurrence and hi/lo split as message.hip (UNI), cut into
    // five work items (part 0: the two exp2; parts 1-4: two values each; the last part moves the operand to the AGPRs)
    float ga_a = 0.f, ga_r = 0.f;
    half8 ga_h, ga_l;
    auto gen_a_part = [&](int part, int k0, float xsq, float env256, half8& ah, half8& al) {   // k0 = klo + 16 ks
        if (part == 0) {
            const float t0 = xsq - mu_h[k0];
            ga_a = env256 * __builtin_amdgcn_exp2f(-(t0 * t0));
            ga_r = __builtin_amdgcn_exp2f(fminf(p.dmu2 * t0 - p.dmusq, 64.0f));
        } else {
            const int j = 2 * (part - 1);
            const float v0 = ga_a;
            ga_a *= ga_r; ga_r *= p.cstep;
            const float v1 = ga_a;
            ga_a *= ga_r; ga_r *= p.cstep;
            const m3_h2 hh = __builtin_amdgcn_cvt_pkrtz(v0, v1);
            const m3_h2 ll = __builtin_amdgcn_cvt_pkrtz(v0 - (float)hh[0], v1 - (float)hh[1]);
            ga_h[j] = (_Float16)hh[0]; ga_h[j + 1] = (_Float16)hh[1];
            ga_l[j] = (_Float16)ll[0]; ga_l[j + 1] = (_Float16)ll[1];
            if (part == 4) { ah = ga_h; al = ga_l; }
        }
    };
    auto gen_a = [&](int k0, float xsq, float env256, half8& ah, half8& al) {
#pragma unroll
        for (int part = 0; part < 5; ++part) gen_a_part(part, k0, xsq, env256, ah, al);
    };
    // row meta (record offset, r_hat) of accumulator row r of this lane, from buffer buf
    auto meta_row = [&](int buf, int r) -> float4 {
        return *reinterpret_cast<const float4*>(meta_w + buf * 128 + (M3_ROW(r) + 4 * hi) * 4);
    };
    auto meta_roff = [&](int buf, int r) -> unsigned int {
        return __float_as_uint(meta_w[buf * 128 + (M3_ROW(r) + 4 * hi) * 4]);
    };
    // requests of one row's record pieces.  The row's record offset was read from LDS one row-slot earlier (ro_next):
    // "read offset, wait, add, load" in one slot exposed an LDS round trip per request.
    unsigned int ro_next = 0;
    auto request_x4 = [&](int r, unsigned int jofs) {   // jofs = 0 (j = 0) or 640 (j = 1)
        const unsigned int o = (M3_EXP == 3 ? (unsigned int)p.N * row_bytes : ro_next) + lo4;
        (void)r;
        if (M3_EXP == 2) { g4[r] = make_float4(1.f, 2.f, 3.f, 4.f); return; }
        if (!VZ) g4[r] = *reinterpret_cast<const float4*>(rec_base + (size_t)o + jofs);
        else g4[r] = make_float4(0.f, 0.f, 0.f, *reinterpret_cast<const float*>(rec_base + (size_t)o + jofs + 12));
    };
    auto request_xc = [&](int r) {
        const unsigned int o = (M3_EXP == 3 ? (unsigned int)p.N * row_bytes : ro_next) + lo1;
        if (M3_EXP == 2) { gza[r] = 1.f; gzb[r] = 2.f; return; }
        gza[r] = *reinterpret_cast<const float*>(rec_base + (size_t)o);
        gzb[r] = *reinterpret_cast<const float*>(rec_base + (size_t)o + 640);
    };
    // The block's 78 MFMA slots (3 phases x 13 ticks x 2 chains).  Slot g: phase g / 26, tick i = (g % 26) / 2 of chain
    // c = g % 2; column blocks per phase: (a0, b0) = (0, 2), (a1, b1) = (1, 3), (c0, c1) = (4, 5).  Tick 0 = the bias, then per
    // k-step a_lo.w_hi, a_hi.w_lo, a_hi.w_hi (message.hip's order per accumulator).  All four k-steps always run (k-steps the
    // block does not have multiply a ZERO A operand: a branch around an MFMA costs more than the MFMA - both versions of the
    // accumulator stay live and its B fragment cannot be read ahead); their fragments are read from a clamped, valid k offset.
    // The B fragment of slot g is read from LDS at slot g - 2 (three rotating registers): with the read placed right in front
    // of its MFMA every slot exposed a full LDS round trip (measured: 130 such waits per block = the whole kernel time).
    half8 fr[3];
    auto slot_cb = [&](int g) -> int {
        const int ph = g / 26, c = g & 1;
        return ph == 0 ? (c ? 2 : 0) : (ph == 1 ? (c ? 3 : 1) : (c ? 5 : 4));
    };
    auto slot_live = [&](int g) -> bool { return !(VZ && g < 52 && (g & 1)); };   // vec == 0: no b chains
    auto frag_load = [&](int g, int klo) {   // g may run past the block (78, 79 = bias slots of the next block: klo not used)
        const int gg = g % 78;
        if (!slot_live(gg)) return;
        const int i = (gg % 26) >> 1, cb = slot_cb(gg);
        const _Float16* bp;
        if (i == 0) {
            bp = wbias + cb * 32 * MSG_LDK;
        } else {
            const int ks = (i - 1) / 3, j = (i - 1) % 3;
            const int k0 = ks < 2 ? klo + 16 * ks : min(klo + 16 * ks, 112);
            bp = (j == 1 ? wlq : whq) + k0 + cb * 32 * MSG_LDK;
        }
        fr[g % 3] = *reinterpret_cast<const half8*>(bp);
    };
    auto mfma_slot = [&](int g, int klo) {
        frag_load(g + 2, klo);
        if (!slot_live(g)) return;
        const int i = (g % 26) >> 1, cb = slot_cb(g);
        if (i == 0) {
            half8 aone = {0, 0, 0, 0, 0, 0, 0, 0};
            if (hi == 0) { aone[0] = (_Float16)256.0f; aone[1] = (_Float16)256.0f; }
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aone, fr[g % 3], zero16, 0, 0, 0);
        } else {
            const int ks = (i - 1) / 3, j = (i - 1) % 3;
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(j == 0 ? Al[ks] : Ah[ks], fr[g % 3], acc[cb], 0, 0, 0);
        }
    };
    auto finish_target = [&](int orow, const float* res) {
        sx0 *= out_scale; sx1 *= out_scale;
        sa0 = (sa0 * inv_sqrt3 + ra0) * inv_sqrt_h; sa1 = (sa1 * inv_sqrt3 + ra1) * inv_sqrt_h;
        sb0 = (sb0 * inv_sqrt3 + rb0) * inv_sqrt_h; sb1 = (sb1 * inv_sqrt3 + rb1) * inv_sqrt_h;
        sc0 = (sc0 * inv_sqrt3 + rc0) * inv_sqrt_h; sc1 = (sc1 * inv_sqrt3 + rc1) * inv_sqrt_h;
        sx0 += __shfl_xor(sx0, 32); sx1 += __shfl_xor(sx1, 32);
        sa0 += __shfl_xor(sa0, 32); sa1 += __shfl_xor(sa1, 32);
        sb0 += __shfl_xor(sb0, 32); sb1 += __shfl_xor(sb1, 32);
        sc0 += __shfl_xor(sc0, 32); sc1 += __shfl_xor(sc1, 32);
        if (orow >= 0) {
            const size_t xo = (size_t)orow * H + c0 + q;
            const size_t vo = (size_t)orow * 3 * H + c0 + q;
            if (hi == 0) {
                p.x_out[xo] = (res[0] + sx0) * inv_sqrt2;
                p.x_out[xo + 32] = (res[1] + sx1) * inv_sqrt2;
                p.vec_out[vo] = res[2] + sa0;
                p.vec_out[vo + 32] = res[3] + sa1;
            } else if (VZ) {
                p.vec_out[vo + H] = sb0;
                p.vec_out[vo + H + 32] = sb1;
                p.vec_out[vo + 2 * H] = sc0;
                p.vec_out[vo + 2 * H + 32] = sc1;
            } else {
                p.vec_out[vo + H] = res[0] + sb0;
                p.vec_out[vo + H + 32] = res[1] + sb1;
                p.vec_out[vo + 2 * H] = res[2] + sc0;
                p.vec_out[vo + 2 * H + 32] = res[3] + sc1;
            }
        }
        sx0 = sx1 = sa0 = sa1 = sb0 = sb1 = sc0 = sc1 = 0.f;
        ra0 = ra1 = rb0 = rb1 = rc0 = rc1 = 0.f;
    };

    // ---- the stream of blocks.  tB = the target after the one of the block requested last (bounds already requested)
    int it = wave;
    M3Block cur, nxt, prv;
    int tB_n, tB_o, tB_have, tB_e0 = 0, tB_e1 = 0;
    {
        int n0, o0, h0;
        item_target(it, n0, o0, h0);
        it += M3_WAVES;
        if (!h0) return;
        cur.n = n0; cur.orow = o0; cur.have = 1;
        cur.eb = o0 >= 0 ? __builtin_amdgcn_readfirstlane(p.nptr[n0]) : 0;
        cur.e1 = o0 >= 0 ? __builtin_amdgcn_readfirstlane(p.nptr[n0 + 1]) : 0;
        cur.nrows = 0; cur.last = 0; cur.klo = 0; cur.nk = 2;
        item_target(it, tB_n, tB_o, tB_have);
        it += M3_WAVES;
        if (tB_have && tB_o >= 0) {
            tB_e0 = __builtin_amdgcn_readfirstlane(p.nptr[tB_n]);
            tB_e1 = __builtin_amdgcn_readfirstlane(p.nptr[tB_n + 1]);
        }
    }
    // advance the request cursor past block d (whose W stage is done): the next block of the same target or the first of tB
    auto next_request = [&](const M3Block& d, M3Block& o) {
        const int adv = d.last;
        o.n = adv ? tB_n : d.n;
        o.orow = adv ? tB_o : d.orow;
        o.eb = adv ? tB_e0 : d.eb + d.nrows;
        o.e1 = adv ? tB_e1 : d.e1;
        o.have = adv ? tB_have : d.have;
        o.nrows = 0; o.last = 0; o.klo = 0; o.nk = 2;
        if (adv) {   // wave-uniform
            int n2, o2, h2;
            item_target(it, n2, o2, h2);
            it += M3_WAVES;
            tB_n = n2; tB_o = o2; tB_have = h2;
            const int nn = (h2 && o2 >= 0) ? n2 : 0;
            const int b0 = __builtin_amdgcn_readfirstlane(p.nptr[nn]), b1 = __builtin_amdgcn_readfirstlane(p.nptr[nn + 1]);
            tB_e0 = (h2 && o2 >= 0) ? b0 : 0;
            tB_e1 = (h2 && o2 >= 0) ? b1 : 0;
        }
    };

    // prologue: block 0 through its W stage, its first A operands and requests; block 1 requested
    {   // meta buffer 2 = the (non-existent) block before the first one: every row points at the all-zero record
        float4 m;
        m.x = __uint_as_float((unsigned int)p.N * row_bytes); m.y = 0.f; m.z = 0.f; m.w = 0.f;
        *reinterpret_cast<float4*>(meta_w + 2 * 128 + q * 4) = m;
    }
    request_rows(cur.eb, cur.e1);
    window_stage(cur, 0, xsqC, envC);
    next_request(cur, nxt);
    request_rows(nxt.eb, nxt.e1);
    gen_a(cur.klo, xsqC, envC, Ah[0], Al[0]);
    gen_a(cur.klo + 16, xsqC, envC, Ah[1], Al[1]);
#pragma unroll
    for (int r = 0; r < 16; ++r) { gza[r] = 0.f; gzb[r] = 0.f; }   // the first block's "previous block" contributes nothing
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[4][i] = 0.f; acc[5][i] = 0.f; }
    prv = cur; prv.have = 0; prv.last = 0;
    int par = 0;   // meta buffer of the current block (previous: (par + 2) % 3, next: (par + 1) % 3)
    frag_load(0, cur.klo);
    frag_load(1, cur.klo);

    // One block t (see the table in the header).  A phase runs two MFMA chains tick by tick (13 slots each) with the phase's W
    // work items dealt evenly over the 26 slots.
    auto body = [&]() {
        const int nk = cur.nk;
        const int klo = cur.klo;
        const int pprev = par == 0 ? 2 : par - 1, pnext = par == 2 ? 0 : par + 1;
        // Request sequence of a block t (each request uses the offset read one step before it):
        //   P1 rows r:      xc(t-1) row r + DC  [meta pprev]        ... last D rows: x4a(t) rows 0..D-1 [meta par]
        //   P2 rows r:      x4a(t) row r + D                          ... last D rows: x4b(t) rows 0..D-1
        //   P3 rows r:      x4b(t) row r + D                          ... last DC rows: xc(t) rows 0..DC-1
        // The block's STREAMING loads (edge rows of block t+2, residual rows of this target: first touches, served by
        // HBM / Infinity Cache) are issued at the start of P1: vmcnt retires in order, so every younger gather waits for them -
        // the xc pieces of the first DC rows of P1 are older (requested in P3 of the previous block), the younger requests are
        // needed half a phase later.  (Issued in P2, between the x4 ring's requests, they cost 13 % of the launch.)
        // ---- P1: a0, b0 of t  ||  W(t+1), R(t+2), residual rows; consume c0, c1 of t-1; A operand of k-steps 2, 3 of t
        M3Block req;
        {
            constexpr int W = 3 + 10 + 16;
            float4 mnext = meta_row(pprev, 0);
            ro_next = meta_roff(pprev, M3_DC);
#pragma unroll
            for (int s = 0; s < 26; ++s) {
                mfma_slot(s, klo);
#pragma unroll
                for (int w = s * W / 26; w < (s + 1) * W / 26; ++w) {
                    if (w == 0) window_stage(nxt, pnext, xsqN, envN);
                    else if (w == 1) { next_request(nxt, req); if (M3_EXP == 0 || M3_EXP == 3) request_rows(req.eb, req.e1); }
                    else if (w == 2) { if (M3_EXP == 0 || M3_EXP == 3) load_res(cur.n, resC); }
                    else if (w < 13) {   // the A operand: k-step 2 is contracted from tick 7 on
                        const int v = w - 3;
                        if (nk > 2 + v / 5) {
                            gen_a_part(v % 5, klo + 32 + 16 * (v / 5), xsqC, envC, Ah[2 + v / 5], Al[2 + v / 5]);
                        } else if (v % 5 == 4) {
                            const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                            Ah[2 + v / 5] = z8; Al[2 + v / 5] = z8;
                        }
                    } else {
                        const int r = w - 13;
                        if (r + M3_DC < 16) {
                            request_xc(r + M3_DC);
                            ro_next = meta_roff(pprev, r + M3_DC + 1 < 16 ? r + M3_DC + 1 : 15);
                        }
                        if (r == 16 - M3_D - 1) ro_next = meta_roff(par, 0);
                        if (r >= 16 - M3_D) {
                            request_x4(r - (16 - M3_D), 0);   // rows 0.. of this block's j = 0 pieces
                            ro_next = meta_roff(par, r - (16 - M3_D) + 1);
                        }
                        const float4 m = mnext;
                        if (r + 1 < 16) mnext = meta_row(pprev, r + 1);
                        const float t3 = gza[r] * acc[4][r];
                        ra0 += t3 * m.y; rb0 += t3 * m.z; rc0 += t3 * m.w;
                        const float u3 = gzb[r] * acc[5][r];
                        ra1 += u3 * m.y; rb1 += u3 * m.z; rc1 += u3 * m.w;
                    }
                }
                M3_TICK_END();
            }
        }
        if (prv.have && prv.last) finish_target(prv.orow, resP);
        // ---- P2: a1, b1 of t  ||  consume a0, b0 of t                                        [ro_next = offset of row D]
        {
#pragma unroll
            for (int s = 0; s < 26; ++s) {
                mfma_slot(26 + s, klo);
#pragma unroll
                for (int r = s * 16 / 26; r < (s + 1) * 16 / 26; ++r) {
                    if (r + M3_D < 16) {
                        request_x4(r + M3_D, 0);
                        ro_next = r + M3_D + 1 < 16 ? meta_roff(par, r + M3_D + 1) : meta_roff(par, 0);
                    } else {
                        request_x4(r + M3_D - 16, 640);   // rows 0.. of the j = 1 pieces: their registers were consumed D rows ago
                        ro_next = meta_roff(par, r + M3_D - 16 + 1);
                    }
                    sx0 += g4[r].w * acc[0][r];
                    if (!VZ) { sa0 += g4[r].x * acc[2][r]; sb0 += g4[r].y * acc[2][r]; sc0 += g4[r].z * acc[2][r]; }
                }
                M3_TICK_END();
            }
        }
        // ---- P3: c0, c1 of t  ||  consume a1, b1 of t; A operand of k-steps 0, 1 of t+1       [ro_next = offset of row D]
        {
            // The A operand of k-steps 0 and 1 of block t+1 is generated IN PLACE: the chains are past k-step 0 after slot 7 and
            // past k-step 1 after slot 13, so its registers are free from there on (no second operand set).
#pragma unroll
            for (int s = 0; s < 26; ++s) {
                mfma_slot(52 + s, s < 24 ? klo : nxt.klo);
                if (s >= 8 && s < 13) gen_a_part(s - 8, nxt.klo, xsqN, envN, Ah[0], Al[0]);
                if (s >= 14 && s < 19) gen_a_part(s - 14, nxt.klo + 16, xsqN, envN, Ah[1], Al[1]);
#pragma unroll
                for (int r = s * 16 / 26; r < (s + 1) * 16 / 26; ++r) {
                    if (r + M3_D < 16) {
                        request_x4(r + M3_D, 640);
                        ro_next = meta_roff(par, r + M3_D + 1 < 16 ? r + M3_D + 1 : 15);
                    }
                    if (r >= 16 - M3_DC) {   // rows 0.. of this block's xc pieces (consumed in the next P1); their offset is read here
                        ro_next = meta_roff(par, r - (16 - M3_DC));
                        request_xc(r - (16 - M3_DC));
                        if (r + M3_D + 1 < 16) ro_next = meta_roff(par, r + M3_D + 1);
                    }
                    sx1 += g4[r].w * acc[1][r];
                    if (!VZ) { sa1 += g4[r].x * acc[3][r]; sb1 += g4[r].y * acc[3][r]; sc1 += g4[r].z * acc[3][r]; }
                }
                M3_TICK_END();
            }
        }
        ksteps += 64 * (VZ ? 4 : 6);   // what the matrix cores execute: 4 k-steps x the column blocks that run
        // rotate
        prv = cur; cur = nxt; nxt = req;
        xsqC = xsqN; envC = envN;
#pragma unroll
        for (int i = 0; i < 4; ++i) resP[i] = resC[i];
        par = pnext;
    };

    while (cur.have) body();
    // drain: column blocks c0, c1 of the last block, then its target
    {
        const int pprev = par == 0 ? 2 : par - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r + M3_DC < 16) { ro_next = meta_roff(pprev, r + M3_DC); request_xc(r + M3_DC); }
            const float4 m = meta_row(pprev, r);
            const float t3 = gza[r] * acc[4][r];
            ra0 += t3 * m.y; rb0 += t3 * m.z; rc0 += t3 * m.w;
            const float u3 = gzb[r] * acc[5][r];
            ra1 += u3 * m.y; rb1 += u3 * m.z; rc1 += u3 * m.w;
        }
    }
    if (prv.have) finish_target(prv.orow, resP);
    if (p.kcount && lane == 0) atomicAdd(p.kcount, (unsigned long long)ksteps);
#undef M3_ROW
}

static size_t m3_lds_bytes() {
    return (size_t)2 * MSG_COLS * MSG_LDK * 2 + sizeof(float) * (128 + M3_WAVES * 3 * 32 * 4);
}

int32_t adf_message3_prepare() {
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(adf_message3_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)m3_lds_bytes()));
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(adf_message3_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)m3_lds_bytes()));
    return ADF_OK;
}

// p: filled by adf_message_impl (message.hip); f16x3 mode with equally spaced centres only
int32_t adf_message3_launch(const MsgParams& p, int num_cus, bool vec_is_zero, hipStream_t s) {
    int workers = num_cus / p.nslices;
    if (workers < 1) workers = 1;
    if (workers > p.G) workers = p.G;
    dim3 grid((unsigned)(workers * p.nslices));
    if (vec_is_zero) hipLaunchKernelGGL(adf_message3_kernel<true>, grid, dim3(M3_THREADS), m3_lds_bytes(), s, p);
    else hipLaunchKernelGGL(adf_message3_kernel<false>, grid, dim3(M3_THREADS), m3_lds_bytes(), s, p);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
