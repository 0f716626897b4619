// RETIRED EXPERIMENT (round 5; not part of the build).  Measured 7.76 ms per full launch against message.hip's 6.92 (1.72 vs
// 1.52 ms at 200 systems; 8.0 ms before the B fragments were requested one slot ahead and the unit vectors carried in
// registers).  Placing the gathers and sums between the MFMAs couples the matrix pipe to the gathers' latency - an in-order
// wave that waits for a record cannot issue its next MFMA - which costs more than the in-wave overlap returns (the overlap
// microbenchmark hides register-resident FMAs only).  Its outputs agree with message.hip's to 1e-6, not bit for bit (not
// debugged).  To rebuild: copy into adsorbdiff_amd/csrc/, add to build.SOURCES, declare adf_message_il_prepare /
// adf_message_il_launch in message.h and call them from message.hip (git history of round 5).
//
// PaiNN message block, the kernel of message.hip with the sums of one channel half placed BETWEEN the MFMAs of the other
// half inside each wave (f16x3 arithmetic, equally spaced Gaussian centres: the default mode; every other mode stays on
// message.hip).  Reference: adsorbdiff/models/painn/painn_denoising.py:530-567 (see message.hip for the mapping).
//
// Why.  On gfx950 the matrix pipe and the vector issue of a SIMD do not overlap ACROSS the two waves of a SIMD: a wave's
// MFMA phase and its partner's VALU phase take the sum of their times (profiles/r05_mfma_valu_overlap.txt); only vector
// instructions between the MFMAs of the SAME wave hide, about half of their cost.  message.hip runs a block as
// [18 MFMAs per k-step] then [~290 vector instructions of gathers and sums]: nothing of the second part hides.  Here a
// block's 192 columns are contracted in two halves - X = (a, b, c) of channels c0 + q, Y = the same of channels
// c0 + 32 + q (column blocks 0, 2, 4 and 1, 3, 5 of the slice's weight image) - and while the MFMAs of one half issue,
// the wave gathers and sums the rows of the OTHER half, whose accumulators are complete:
//     block i, phase X:  MFMAs X(i)   ||  sums of Y(i-1)
//     block i, phase Y:  MFMAs Y(i)   ||  sums of X(i)
// The A operand of the first three 16-deep k-steps is generated once per block and kept in registers (both phases use it);
// further steps of unusually wide windows regenerate theirs.  Every accumulator sees the same MFMAs in the same order and
// every running sum the same rows in the same order as in message.hip: the outputs are bit-identical
// (tests/test_gpu_parity.py::test_interleaved_message_kernel_is_bit_identical).
#include <stdlib.h>
#include <string.h>

#include "message.h"

#define IL_THREADS 512
#define IL_WAVES 8
#define IL_MU 176   // centres + tail: the operand of a cached step past the window reads behind the last centre

template <bool VZ>
__global__ __launch_bounds__(IL_THREADS, 2) void adf_message_il_kernel(MsgParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // carve: weight image (hi | lo, [192][MSG_LDK] halves each) | [IL_MU] mu | [8 waves][32][8] row meta | work counter
    _Float16* Wh = reinterpret_cast<_Float16*>(lds);
    _Float16* Wlo = Wh + MSG_COLS * MSG_LDK;
    float* Mu = lds + (2 * MSG_COLS * MSG_LDK) / 2;
    float* Meta = Mu + IL_MU;
    int* Ctr = reinterpret_cast<int*>(Meta + IL_WAVES * 32 * 8);

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

    {   // stage this slice's rbf_proj image once (as message.hip: bias as an fp16 hi/lo pair in k slots 128/129)
        const int R8 = p.R / 8;
        const half8* src = reinterpret_cast<const half8*>(p.wpack16 + (size_t)slice * 2 * MSG_COLS * p.R);
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const _Float16* b16 = reinterpret_cast<const _Float16*>(p.bpack) + (size_t)slice * MSG_COLS * 2;
        for (int i = tid; i < 2 * MSG_COLS * 17; i += IL_THREADS) {
            const int row = i / 17, piece = i - row * 17;
            half8 v = piece < R8 ? src[row * R8 + piece] : zero8;
            if (piece == 16 && row < MSG_COLS) { v[0] = b16[2 * row]; v[1] = b16[2 * row + 1]; }
            *reinterpret_cast<half8*>(Wh + (size_t)row * MSG_LDK + piece * 8) = v;
        }
        if (tid < IL_MU) Mu[tid] = (tid < p.R ? p.mu[tid] : 2.0f) * p.sarg;
        if (tid == 0) *Ctr = 0;
    }
    __syncthreads();
    float* meta_w = Meta + wave * 32 * 8;
    const float inv_sqrt3 = 0.57735026918962576f;
    const float inv_sqrt2 = 0.70710678118654752f;
    const float out_scale = *p.inv_scale * (1.0f / 256.0f);  // accumulators hold 256 * scale * rbfh
    const float inv_sqrt_h = out_scale / sqrtf((float)H);
    const float umax_scale = (float)(p.R - 1);
    const unsigned int row_bytes = (unsigned int)p.nslices * 1280u;
    const char* recS = reinterpret_cast<const char*>(p.rec) + (size_t)slice * 1280;
    const unsigned int qA = (unsigned int)q * 16u;
    const unsigned int qP = 512u + (unsigned int)q * 4u;
    unsigned int ksteps = 0;

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
    // row meta of a block: per row 8 floats = two 4-float slots (record byte offset, unit vector) used alternately by
    // consecutive blocks of this wave (a block's Y rows are summed while the NEXT block's MFMAs run)
    auto write_meta = [&](int par_, const float4& g, int s_, bool v) {
        if (hi == 0) {
            float* m = meta_w + q * 8 + 4 * par_;
            m[0] = __uint_as_float((unsigned int)(v ? s_ : p.N) * row_bytes);  // padded rows gather the all-zero row N
            m[1] = g.x; m[2] = g.y; m[3] = g.z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    int n = 0, orow = 0, eb = 0, e1 = 0;
    bool have = fetch_target(n, orow);
    if (have) { eb = nptr_c[n]; e1 = nptr_c[n + 1]; }
    int nN = 0, oN = 0, e0N = 0, e1N = 0;
    bool haveN = have && fetch_target(nN, oN);
    if (haveN) { e0N = nptr_c[nN]; e1N = nptr_c[nN + 1]; }
    float4 geo = make_float4(0.f, 0.f, 0.f, 0.f); int src = 0; bool valid = false;
    if (have) load_block(eb, e1, geo, src, valid);
    asm volatile("" : "+v"(geo.x), "+v"(geo.y), "+v"(geo.z), "+v"(geo.w), "+v"(src));  // see message.hip
    int par = 0;
    // "previous block" of the first block: all rows padding, accumulators zero - its Y sums add nothing
    write_meta(1, make_float4(0.f, 0.f, 0.f, 0.f), 0, false);
    if (have) write_meta(0, geo, src, valid);
    bool first = true;
    bool lastP = false;   // the previous block was the last of its target: its Y outputs are written after this block's X phase
    int orowP = 0;
    float sx0 = 0.f, sx1 = 0.f, sa0 = 0.f, sa1 = 0.f, sb0 = 0.f, sb1 = 0.f, sc0 = 0.f, sc1 = 0.f;
    float ra0 = 0.f, ra1 = 0.f, rb0 = 0.f, rb1 = 0.f, rc0 = 0.f, rc1 = 0.f;
    float res0 = 0.f, res1 = 0.f, res2 = 0.f, res3 = 0.f;   // residual inputs of the targets whose sums are running
    float nr0 = 0.f, nr1 = 0.f, nr2 = 0.f, nr3 = 0.f;       // ... of the target that has just started
    f32x16 acc[6];
#pragma unroll
    for (int b = 0; b < 6; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

#define ROW_OF(r) ((r & 3) + 8 * (r >> 2) + 4 * hi)
    // one half-row of gathered record: (P0, P1, P2, xa) + xc of channel c0 + 32 j + q
#define DECLH(h, r) float4 g##h##a##r; float g##h##z##r; float u##h##x##r, u##h##y##r, u##h##z##r;
    // gather of one half-row: ONE 16-byte LDS read brings the record offset and the unit vector of the row (kept in
    // registers until the row is summed: an LDS round trip in front of the sums would sit in an MFMA's shadow and hold the
    // next MFMA back)
#define GATHERH(h, J, MP, r)                                                                              \
    {                                                                                                     \
        const float4 mm_ = *reinterpret_cast<const float4*>((MP) + ROW_OF(r) * 8);                        \
        const unsigned int o_ = __float_as_uint(mm_.x);                                                   \
        u##h##x##r = mm_.y; u##h##y##r = mm_.z; u##h##z##r = mm_.w;                                       \
        g##h##z##r = *reinterpret_cast<const float*>(recS + (size_t)(o_ + qP) + 640 * (J));              \
        if (!VZ) g##h##a##r = *reinterpret_cast<const float4*>(recS + (size_t)(o_ + qA) + 640 * (J));    \
        else g##h##a##r = make_float4(0.f, 0.f, 0.f, *reinterpret_cast<const float*>(recS + (size_t)(o_ + qA) + 640 * (J) + 12)); \
    }
#define CONSUMEX(MP, r)                                                                                   \
    {                                                                                                     \
        const float t3 = gxz##r * acc[4][r];                                                              \
        sx0 += gxa##r.w * acc[0][r];                                                                      \
        if (!VZ) { sa0 += gxa##r.x * acc[2][r]; sb0 += gxa##r.y * acc[2][r]; sc0 += gxa##r.z * acc[2][r]; } \
        ra0 += t3 * uxx##r; rb0 += t3 * uxy##r; rc0 += t3 * uxz##r;                                       \
    }
#define CONSUMEY(MP, r)                                                                                   \
    {                                                                                                     \
        const float u3 = gyz##r * acc[5][r];                                                              \
        sx1 += gya##r.w * acc[1][r];                                                                      \
        if (!VZ) { sa1 += gya##r.x * acc[3][r]; sb1 += gya##r.y * acc[3][r]; sc1 += gya##r.z * acc[3][r]; } \
        ra1 += u3 * uyx##r; rb1 += u3 * uyy##r; rc1 += u3 * uyz##r;                                       \
    }
    DECLH(x, 0) DECLH(x, 1) DECLH(x, 2) DECLH(x, 3) DECLH(x, 4) DECLH(x, 5) DECLH(x, 6) DECLH(x, 7)
    DECLH(x, 8) DECLH(x, 9) DECLH(x, 10) DECLH(x, 11) DECLH(x, 12) DECLH(x, 13) DECLH(x, 14) DECLH(x, 15)
    DECLH(y, 0) DECLH(y, 1) DECLH(y, 2) DECLH(y, 3) DECLH(y, 4) DECLH(y, 5) DECLH(y, 6) DECLH(y, 7)
    DECLH(y, 8) DECLH(y, 9) DECLH(y, 10) DECLH(y, 11) DECLH(y, 12) DECLH(y, 13) DECLH(y, 14) DECLH(y, 15)
    // head start of the null previous block's Y rows
    {
        const float* mp0 = meta_w + 4;
        GATHERH(y, 1, mp0, 0) GATHERH(y, 1, mp0, 1)
    }

    while (have) {
        const bool last = eb + 32 >= e1;
        // ---- request what the NEXT block needs
        float4 geoN = make_float4(0.f, 0.f, 0.f, 0.f); int srcN = 0; bool validN = false;
        if (!last) load_block(eb + 32, e1, geoN, srcN, validN);
        else if (haveN) load_block(e0N, e1N, geoN, srcN, validN);
        if (first) {  // residual rows of this target (painn_denoising.py:443-445)
            const size_t xo = (size_t)n * H + c0 + q;
            const size_t vo = (size_t)n * 3 * H + c0 + q;
            nr0 = nr1 = nr2 = nr3 = 0.f;
            if (hi == 0) {
                nr0 = p.x[xo]; nr1 = p.x[xo + 32];
                if (!VZ) { nr2 = p.vec[vo]; nr3 = p.vec[vo + 32]; }
            } else if (!VZ) {
                nr0 = p.vec[vo + H]; nr1 = p.vec[vo + H + 32];
                nr2 = p.vec[vo + 2 * H]; nr3 = p.vec[vo + 2 * H + 32];
            }
        }
        const float xs = geo.w * p.inv_cutoff;
        float xp = xs;
        if (p.env_pi == 5) { const float x2 = xs * xs; xp = x2 * x2 * xs; }
        else for (int i = 1; i < p.env_pi; ++i) xp *= xs;
        float env = 1.0f + p.env_a * xp + p.env_b * (xp * xs) + p.env_c * (xp * xs * xs);
        env = (xs < 1.0f && valid) ? env : 0.0f;
        // k-window of this block (edges sorted by distance: first and last valid row bound it)
        const float u = xs * umax_scale;
        const int nvalid = __builtin_amdgcn_readfirstlane(min(32, e1 - eb));
        const float umin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), 0));
        const float umax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), max(nvalid, 1) - 1));
        int klo, khi;
        if (nvalid <= 0) {
            klo = 0; khi = 16;
        } else {
            klo = max(0, (int)floorf(umin) - 5) & ~7;
            khi = min(p.R, (int)ceilf(umax) + 6);
            khi = klo + ((khi - klo + 15) & ~15);
            if (khi > 128) { klo -= khi - 128; khi = 128; }
        }
        klo = __builtin_amdgcn_readfirstlane(klo);
        khi = __builtin_amdgcn_readfirstlane(khi);
        const int nsteps = (khi - klo) >> 4;
        ksteps += (khi - klo) * (VZ ? 4 : 6);

        const float env256 = env * 256.0f;
        const float xsq = xs * p.sarg;
        // A fragment of one 16-deep step (message.hip: recurrence over the 8 centres of a lane, packed hi/lo split)
        auto gen_a = [&](int k0, half8& ah, half8& al) {
            const float t0 = xsq - Mu[k0 + 8 * hi];
            float a = env256 * __builtin_amdgcn_exp2f(-(t0 * t0));
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
        };
        half8 ah0, al0, ah1, al1, ah2, al2;
        gen_a(klo, ah0, al0);
        gen_a(klo + 16, ah1, al1);
        const bool three = nsteps >= 3;      // which of the two schedules below runs (wave-uniform)
        if (three) gen_a(klo + 32, ah2, al2);
        else { ah2 = ah1; al2 = al1; }
        // a one-step window runs the two-step schedule with a zero operand on the first step's weights again: adds exact zeros
        const half8 zero8h = {0, 0, 0, 0, 0, 0, 0, 0};
        if (nsteps < 2) { ah1 = zero8h; al1 = zero8h; }
        const int soff1 = nsteps < 2 ? 0 : 16;

        const float* meta_c = meta_w + 4 * par;         // this block's rows
        const float* meta_p = meta_w + 4 * (par ^ 1);   // the previous block's rows
        const _Float16* whq = Wh + (size_t)q * MSG_LDK + klo + 8 * hi;
        const _Float16* wlq = Wlo + (size_t)q * MSG_LDK + klo + 8 * hi;
        // A slot = the three products of column block B_ at halves offset KO_ of the window, with gathers (G_) and the sums
        // of up to three rows (C0_, C1_, C2_) of the other half between them.  No branch inside a phase: hipcc's waitcnt
        // pass counts the outstanding gathers exactly only in straight-line code.
#define LOADB(BH_, BL_, B_, KO_)                                                                          \
        BH_ = *reinterpret_cast<const half8*>(whq + (B_) * 32 * MSG_LDK + (KO_));                         \
        BL_ = *reinterpret_cast<const half8*>(wlq + (B_) * 32 * MSG_LDK + (KO_));
        // BH_/BL_: this slot's fragments (requested by the previous slot); NB_: statement that requests the next slot's
#define SLOT(B_, AH_, AL_, BH_, BL_, NB_, G_, C0_, C1_, C2_)                                              \
        {                                                                                                  \
            NB_                                                                                            \
            acc[B_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL_, BH_, acc[B_], 0, 0, 0);                  \
            G_ __builtin_amdgcn_sched_barrier(0);                                                          \
            acc[B_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH_, BL_, acc[B_], 0, 0, 0);                  \
            C0_ __builtin_amdgcn_sched_barrier(0);                                                         \
            acc[B_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH_, BH_, acc[B_], 0, 0, 0);                  \
            C1_ C2_ __builtin_amdgcn_sched_barrier(0);                                                     \
        }
        // the same slot without MFMAs (vec == 0: the xb columns multiply P = 0)
#define NOSLOT(G_, C0_, C1_, C2_) { G_ C0_ C1_ C2_ }
#define SLOTB(B_, AH_, AL_, BH_, BL_, NB_, G_, C0_, C1_, C2_)                                             \
        if (!VZ) SLOT(B_, AH_, AL_, BH_, BL_, NB_, G_, C0_, C1_, C2_) else { NB_ NOSLOT(G_, C0_, C1_, C2_) }
        half8 aone = zero8h;
        if (hi == 0) { aone[0] = (_Float16)256.0f; aone[1] = (_Float16)256.0f; }
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define BIAS_INIT(B_) acc[B_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aone, *reinterpret_cast<const half8*>(Wh + (size_t)((B_) * 32 + q) * MSG_LDK + 128), zero16, 0, 0, 0);
#define NOP_
#define GY(r) GATHERH(y, 1, meta_p, r)
#define CY(r) CONSUMEY(meta_p, r)
#define GX(r) GATHERH(x, 0, meta_c, r)
#define CX(r) CONSUMEX(meta_c, r)

        // ================= phase X: MFMAs of (a, b, c)(channel c0 + q) || sums of the previous block's Y half
        GX(0) GX(1)     // head start of this block's X rows (summed in phase Y)
        half8 bhA, blA, bhB, blB;        // B fragments of the current / next slot (requested one slot ahead)
        LOADB(bhA, blA, 0, 0)
        BIAS_INIT(0) if (!VZ) { BIAS_INIT(2) } BIAS_INIT(4)
        __builtin_amdgcn_sched_barrier(0);
        SLOT(0, ah0, al0, bhA, blA, LOADB(bhB, blB, 2, 0), GY(2) GY(3), CY(0), CY(1), NOP_)
        SLOTB(2, ah0, al0, bhB, blB, LOADB(bhA, blA, 4, 0), GY(4) GY(5), CY(2), CY(3), NOP_)
        SLOT(4, ah0, al0, bhA, blA, LOADB(bhB, blB, 0, soff1), GY(6) GY(7), CY(4), CY(5), NOP_)
        SLOT(0, ah1, al1, bhB, blB, LOADB(bhA, blA, 2, soff1), GY(8) GY(9), CY(6), CY(7), NOP_)
        SLOTB(2, ah1, al1, bhA, blA, LOADB(bhB, blB, 4, soff1), GY(10) GY(11), CY(8), CY(9), NOP_)
        SLOT(4, ah1, al1, bhB, blB, LOADB(bhA, blA, 0, 32), GY(12) GY(13), CY(10), CY(11), NOP_)
        if (three) {   // (the empty asm keeps hipcc from hoisting the arms' common gathers / sums in front of the branch)
            asm volatile("" ::: "memory");
            SLOT(0, ah2, al2, bhA, blA, LOADB(bhB, blB, 2, 32), GY(14) GY(15), CY(12), CY(13), NOP_)
            SLOTB(2, ah2, al2, bhB, blB, LOADB(bhA, blA, 4, 32), NOP_, CY(14), CY(15), NOP_)
            SLOT(4, ah2, al2, bhA, blA, NOP_, NOP_, NOP_, NOP_, NOP_)
            for (int s = 3; s < nsteps; ++s) {   // unusually wide window: further steps regenerate their operand
                half8 ahs, als;
                gen_a(klo + 16 * s, ahs, als);
#pragma unroll
                for (int b = 0; b < 6; b += 2) {
                    if (VZ && b == 2) continue;
                    const half8 bh = *reinterpret_cast<const half8*>(whq + b * 32 * MSG_LDK + 16 * s);
                    const half8 bl = *reinterpret_cast<const half8*>(wlq + b * 32 * MSG_LDK + 16 * s);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(als, bh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahs, bl, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahs, bh, acc[b], 0, 0, 0);
                }
            }
        } else {
            asm volatile("" ::: "memory");
            GY(14) GY(15) CY(12) CY(13) CY(14) CY(15)
        }
        if (lastP) {
            // ---- the previous block ended its target: channel c0 + 32 + q of that target is complete
            sx1 *= out_scale;
            sa1 = (sa1 * inv_sqrt3 + ra1) * inv_sqrt_h;
            sb1 = (sb1 * inv_sqrt3 + rb1) * inv_sqrt_h;
            sc1 = (sc1 * inv_sqrt3 + rc1) * inv_sqrt_h;
            sx1 += __shfl_xor(sx1, 32); sa1 += __shfl_xor(sa1, 32); sb1 += __shfl_xor(sb1, 32); sc1 += __shfl_xor(sc1, 32);
            const size_t xo = (size_t)orowP * H + c0 + q + 32;
            const size_t vo = (size_t)orowP * 3 * H + c0 + q + 32;
            if (hi == 0) {
                p.x_out[xo] = (res1 + sx1) * inv_sqrt2;
                p.vec_out[vo] = res3 + sa1;
            } else {
                p.vec_out[vo + H] = res1 + sb1;
                p.vec_out[vo + 2 * H] = res3 + sc1;
            }
            sx1 = sa1 = sb1 = sc1 = ra1 = rb1 = rc1 = 0.f;
        }
        if (first) { res0 = nr0; res1 = nr1; res2 = nr2; res3 = nr3; }   // (requested a whole phase ago)

        // ================= phase Y: MFMAs of (a, b, c)(channel c0 + 32 + q) || sums of this block's X half
        GATHERH(y, 1, meta_c, 0) GATHERH(y, 1, meta_c, 1)   // head start of this block's Y rows (summed in the next block)
        LOADB(bhA, blA, 1, 0)
        BIAS_INIT(1) if (!VZ) { BIAS_INIT(3) } BIAS_INIT(5)
        __builtin_amdgcn_sched_barrier(0);
        SLOT(1, ah0, al0, bhA, blA, LOADB(bhB, blB, 3, 0), GX(2) GX(3), CX(0), CX(1), NOP_)
        SLOTB(3, ah0, al0, bhB, blB, LOADB(bhA, blA, 5, 0), GX(4) GX(5), CX(2), CX(3), NOP_)
        SLOT(5, ah0, al0, bhA, blA, LOADB(bhB, blB, 1, soff1), GX(6) GX(7), CX(4), CX(5), NOP_)
        SLOT(1, ah1, al1, bhB, blB, LOADB(bhA, blA, 3, soff1), GX(8) GX(9), CX(6), CX(7), NOP_)
        SLOTB(3, ah1, al1, bhA, blA, LOADB(bhB, blB, 5, soff1), GX(10) GX(11), CX(8), CX(9), NOP_)
        SLOT(5, ah1, al1, bhB, blB, LOADB(bhA, blA, 1, 32), GX(12) GX(13), CX(10), CX(11), NOP_)
        if (three) {   // (the empty asm keeps hipcc from hoisting the arms' common gathers / sums in front of the branch)
            asm volatile("" ::: "memory");
            SLOT(1, ah2, al2, bhA, blA, LOADB(bhB, blB, 3, 32), GX(14) GX(15), CX(12), CX(13), NOP_)
            SLOTB(3, ah2, al2, bhB, blB, LOADB(bhA, blA, 5, 32), NOP_, CX(14), CX(15), NOP_)
            SLOT(5, ah2, al2, bhA, blA, NOP_, NOP_, NOP_, NOP_, NOP_)
            for (int s = 3; s < nsteps; ++s) {   // unusually wide window: further steps regenerate their operand
                half8 ahs, als;
                gen_a(klo + 16 * s, ahs, als);
#pragma unroll
                for (int b = 1; b < 6; b += 2) {
                    if (VZ && b == 3) continue;
                    const half8 bh = *reinterpret_cast<const half8*>(whq + b * 32 * MSG_LDK + 16 * s);
                    const half8 bl = *reinterpret_cast<const half8*>(wlq + b * 32 * MSG_LDK + 16 * s);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(als, bh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahs, bl, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahs, bh, acc[b], 0, 0, 0);
                }
            }
        } else {
            asm volatile("" ::: "memory");
            GX(14) GX(15) CX(12) CX(13) CX(14) CX(15)
        }
        if (last) {
            // ---- channel c0 + q of this target is complete
            sx0 *= out_scale;
            sa0 = (sa0 * inv_sqrt3 + ra0) * inv_sqrt_h;
            sb0 = (sb0 * inv_sqrt3 + rb0) * inv_sqrt_h;
            sc0 = (sc0 * inv_sqrt3 + rc0) * inv_sqrt_h;
            sx0 += __shfl_xor(sx0, 32); sa0 += __shfl_xor(sa0, 32); sb0 += __shfl_xor(sb0, 32); sc0 += __shfl_xor(sc0, 32);
            const size_t xo = (size_t)orow * H + c0 + q;
            const size_t vo = (size_t)orow * 3 * H + c0 + q;
            if (hi == 0) {
                p.x_out[xo] = (res0 + sx0) * inv_sqrt2;
                p.vec_out[vo] = res2 + sa0;
            } else {
                p.vec_out[vo + H] = res0 + sb0;
                p.vec_out[vo + 2 * H] = res2 + sc0;
            }
            sx0 = sa0 = sb0 = sc0 = ra0 = rb0 = rc0 = 0.f;
        }
        // ---- advance
        lastP = last; orowP = orow;
        if (last) {
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
        par ^= 1;
        write_meta(par, geo, src, valid);   // the next block's rows; the slot just left keeps this block's for its Y sums
    }
    // ================= drain: the Y half of the last block
    {
        const float* meta_p = meta_w + 4 * (par ^ 1);
        GATHERH(y, 1, meta_p, 2) GATHERH(y, 1, meta_p, 3) CONSUMEY(meta_p, 0) CONSUMEY(meta_p, 1)
        GATHERH(y, 1, meta_p, 4) GATHERH(y, 1, meta_p, 5) CONSUMEY(meta_p, 2) CONSUMEY(meta_p, 3)
        GATHERH(y, 1, meta_p, 6) GATHERH(y, 1, meta_p, 7) CONSUMEY(meta_p, 4) CONSUMEY(meta_p, 5)
        GATHERH(y, 1, meta_p, 8) GATHERH(y, 1, meta_p, 9) CONSUMEY(meta_p, 6) CONSUMEY(meta_p, 7)
        GATHERH(y, 1, meta_p, 10) GATHERH(y, 1, meta_p, 11) CONSUMEY(meta_p, 8) CONSUMEY(meta_p, 9)
        GATHERH(y, 1, meta_p, 12) GATHERH(y, 1, meta_p, 13) CONSUMEY(meta_p, 10) CONSUMEY(meta_p, 11)
        GATHERH(y, 1, meta_p, 14) GATHERH(y, 1, meta_p, 15) CONSUMEY(meta_p, 12) CONSUMEY(meta_p, 13)
        CONSUMEY(meta_p, 14) CONSUMEY(meta_p, 15)
        if (lastP) {
            sx1 *= out_scale;
            sa1 = (sa1 * inv_sqrt3 + ra1) * inv_sqrt_h;
            sb1 = (sb1 * inv_sqrt3 + rb1) * inv_sqrt_h;
            sc1 = (sc1 * inv_sqrt3 + rc1) * inv_sqrt_h;
            sx1 += __shfl_xor(sx1, 32); sa1 += __shfl_xor(sa1, 32); sb1 += __shfl_xor(sb1, 32); sc1 += __shfl_xor(sc1, 32);
            const size_t xo = (size_t)orowP * H + c0 + q + 32;
            const size_t vo = (size_t)orowP * 3 * H + c0 + q + 32;
            if (hi == 0) {
                p.x_out[xo] = (res1 + sx1) * inv_sqrt2;
                p.vec_out[vo] = res3 + sa1;
            } else {
                p.vec_out[vo + H] = res1 + sb1;
                p.vec_out[vo + 2 * H] = res3 + sc1;
            }
        }
    }
    if (p.kcount && lane == 0) atomicAdd(p.kcount, (unsigned long long)ksteps);
#undef ROW_OF
#undef DECLH
#undef GATHERH
#undef CONSUMEX
#undef CONSUMEY
#undef SLOT
#undef LOADB
#undef NOSLOT
#undef SLOTB
#undef GX
#undef GY
#undef CX
#undef CY
#undef BIAS_INIT
#undef NOP_
}

static size_t il_lds_bytes() {
    return (size_t)2 * MSG_COLS * MSG_LDK * 2 + sizeof(float) * (IL_MU + IL_WAVES * 32 * 8) + 16;
}

int32_t adf_message_il_prepare() {
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(adf_message_il_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)il_lds_bytes()));
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(adf_message_il_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)il_lds_bytes()));
    return ADF_OK;
}

int32_t adf_message_il_launch(const MsgParams& p, int num_cus, bool vec_is_zero, hipStream_t s) {
    int workers = num_cus / p.nslices;
    if (workers < 1) workers = 1;
    if (workers > p.G) workers = p.G;
    dim3 grid((unsigned)(workers * p.nslices));
    if (vec_is_zero) hipLaunchKernelGGL(adf_message_il_kernel<true>, grid, dim3(IL_THREADS), il_lds_bytes(), s, p);
    else hipLaunchKernelGGL(adf_message_il_kernel<false>, grid, dim3(IL_THREADS), il_lds_bytes(), s, p);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
