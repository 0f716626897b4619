// PaiNN message block, the kernel of message.hip at FOUR waves per SIMD (opt-in: ADF_MSG_KERNEL=v4; f16x3 arithmetic with
// equally spaced Gaussian centres only).  Reference: adsorbdiff/models/painn/painn_denoising.py:530-567, 443-445 (see message.hip).
//
// Why: at two waves per SIMD the vector pipe issues one wave64 instruction per ~4.7 cycles, at four per ~3.3
// (profiles/r02/r02_issue_rates.txt), and message.hip is bound by vector issue beside its MFMAs.  Its 96
// accumulator registers per wave (a 64-channel slice: 6 column blocks) hold it at two waves.  Here a workgroup has 16 waves
// on the same 64-channel weight image (104 KB of LDS, as before); waves 2p and 2p + 1 walk the SAME targets and each owns
// one 32-channel half of the slice: 3 column blocks = 48 accumulator registers, one gathered channel per lane and row, <= 128
// registers.  The price: both waves of a pair build the row metadata and the A operand (the Gaussian recurrence) of the
// same edges - ~165 of a block's ~350 vector instructions per wave are duplicated work.
// MEASURED (MI355X, 200 systems, one launch): 2.30 ms against 1.75 ms for message.hip - SLOWER; outputs bit-identical (the same
// operations in the same order per channel).  The duplicated A-operand / metadata work outweighs the better issue rate, and
// 128 registers are tight (a few spills remain).  Kept opt-in as the record of the experiment.
// Targets are dealt to the pairs round-robin (pair p takes items p, p + 8, ... of the workgroup's groups): the two waves of
// a pair need no communication.
#include <stdlib.h>
#include <string.h>

#include "message.h"

#define M4_THREADS 1024
#define M4_WAVES 16
#ifndef M4_AHEAD
#define M4_AHEAD 2   // gather rows requested ahead of the row being consumed (2 or 4; 4 spills at 128 registers: 2.9 vs 2.3 ms)
#endif

template <bool VZ>
__global__ __launch_bounds__(M4_THREADS, 1) void adf_message4_kernel(MsgParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wfloats = (2 * MSG_COLS * MSG_LDK) / 2;
    _Float16* Wh = reinterpret_cast<_Float16*>(lds);       // [192][MSG_LDK] hi
    _Float16* Wlo = Wh + MSG_COLS * MSG_LDK;               // [192][MSG_LDK] lo
    float* Mu = lds + wfloats;
    float* Meta = Mu + 128;                                // [16 waves][32][8]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int pair = wave >> 1, jsel = wave & 1;
    const int q = lane & 31;
    const int hi = lane >> 5;
    const int slice = blockIdx.x % p.nslices;
    const int worker = blockIdx.x / p.nslices;
    const int nworkers = gridDim.x / p.nslices;
    const int items = p.items_dev ? min(p.items, (int)*p.items_dev) : p.items;
    const int ngroups = (items + ADF_GROUP_NODES - 1) / ADF_GROUP_NODES;
    if (worker >= ngroups) return;
    const int H = p.H;
    const int c0 = slice * ADF_SLICE_CH + 32 * jsel;       // this wave's 32 channels

    {   // stage this slice's rbf_proj image once (message.hip)
        const int R8 = p.R / 8;
        const half8* src = reinterpret_cast<const half8*>(p.wpack16 + (size_t)slice * 2 * MSG_COLS * p.R);
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const _Float16* b16 = reinterpret_cast<const _Float16*>(p.bpack) + (size_t)slice * MSG_COLS * 2;
        for (int i = tid; i < 2 * MSG_COLS * 17; i += M4_THREADS) {
            const int row = i / 17, piece = i - row * 17;
            half8 v = piece < R8 ? src[row * R8 + piece] : zero8;
            if (piece == 16 && row < MSG_COLS) { v[0] = b16[2 * row]; v[1] = b16[2 * row + 1]; }
            *reinterpret_cast<half8*>(Wh + (size_t)row * MSG_LDK + piece * 8) = v;
        }
        if (tid < 128) Mu[tid] = (tid < p.R ? p.mu[tid] : 2.0f) * p.sarg;
    }
    __syncthreads();
    float* meta_w = Meta + wave * 32 * 8;
    const float inv_sqrt3 = 0.57735026918962576f;
    const float inv_sqrt2 = 0.70710678118654752f;
    const float out_scale = *p.inv_scale * (1.0f / 256.0f);  // accumulators hold 256*scale*rbfh
    const float inv_sqrt_h = out_scale / sqrtf((float)H);
    const float umax_scale = (float)(p.R - 1);
    const unsigned int row_bytes = (unsigned int)p.nslices * 1280u;
    // half-record 2*slice + jsel of a record row: [32 x (P0, P1, P2, xa)] + [32 x xc]
    const char* recA = reinterpret_cast<const char*>(p.rec) + (size_t)slice * 1280 + (size_t)jsel * 640 + (size_t)q * 16;
    const char* recP = reinterpret_cast<const char*>(p.rec) + (size_t)slice * 1280 + (size_t)jsel * 640 + 512 + (size_t)q * 4;
    // this wave's three column blocks of the image: parts x, a, b of its 32 channels
    const int wrow = 32 * jsel + q;

    unsigned int ksteps = 0;
    int tnext = pair;   // static deal: pair p takes items p, p + 8, ... (both waves of a pair walk the same sequence)
    auto fetch_target = [&](int& n_out, int& o_out) -> bool {
        while (true) {
            const int t = tnext;
            tnext += M4_WAVES / 2;
            const int g = worker + (t >> 5) * nworkers;
            if (g >= ngroups) return false;
            const int e = g * ADF_GROUP_NODES + (t & 31);
            if (e < items) { o_out = e; n_out = p.tlist ? p.tlist[e] : e; return true; }
        }
    };
    auto load_block = [&](int eb, int e1, float4& geo, int& src, bool& valid) {
        const int e = eb + q;
        valid = e < e1;
        geo = make_float4(0.f, 0.f, 0.f, 0.f);
        src = 0;
        if (valid) { geo = p.e_geom[e]; src = p.e_src[e]; }
    };

    int n = 0, orow = 0, eb = 0, e1 = 0;
    bool have = fetch_target(n, orow);
    if (have) { eb = p.nptr[n]; e1 = p.nptr[n + 1]; }
    int nN = 0, oN = 0, e0N = 0, e1N = 0;
    bool haveN = have && fetch_target(nN, oN);
    if (haveN) { e0N = p.nptr[nN]; e1N = p.nptr[nN + 1]; }
    float4 geo; int src; bool valid;
    if (have) load_block(eb, e1, geo, src, valid);
    bool first = true;
    float sx = 0.f, sa = 0.f, sb = 0.f, sc = 0.f, ra = 0.f, rb = 0.f, rc = 0.f;
    float res0 = 0.f, res1 = 0.f;  // residual inputs of this lane's two output rows

    while (have) {
        {
            const bool last = eb + 32 >= e1;
            float4 geoN = make_float4(0.f, 0.f, 0.f, 0.f); int srcN = 0; bool validN = false;
            if (!last) load_block(eb + 32, e1, geoN, srcN, validN);
            else if (haveN) load_block(e0N, e1N, geoN, srcN, validN);
            if (first) {  // residual rows of this target (painn_denoising.py:443-445)
                const size_t xo = (size_t)n * H + c0 + q;
                const size_t vo = (size_t)n * 3 * H + c0 + q;
                if (hi == 0) {
                    res0 = p.x[xo];
                    if (!VZ) res1 = p.vec[vo];
                } else if (!VZ) {
                    res0 = p.vec[vo + H];
                    res1 = p.vec[vo + 2 * H];
                }
            }
            const float xs = geo.w * p.inv_cutoff;
            float xp = xs;
            for (int i = 1; i < p.env_pi; ++i) xp *= xs;
            float env = 1.0f + p.env_a * xp + p.env_b * (xp * xs) + p.env_c * (xp * xs * xs);
            env = (xs < 1.0f && valid) ? env : 0.0f;
            __builtin_amdgcn_wave_barrier();
            if (hi == 0) {
                float* m = meta_w + q * 8;
                m[0] = __uint_as_float((unsigned int)(valid ? src : p.N) * row_bytes);
                m[1] = geo.x; m[2] = geo.y; m[3] = geo.z;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
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
            if (jsel == 0) ksteps += (khi - klo) * (VZ ? 4 : 6);  // per 64-channel block, as message.hip counts

#define ROW_OF(r) ((r & 3) + 8 * (r >> 2) + 4 * hi)
#define GATHER(r)                                                                               \
    const float* m##r = meta_w + ROW_OF(r) * 8;                                                 \
    const unsigned int o##r = __float_as_uint(m##r[0]);                                         \
    float4 ga##r;                                                                               \
    const float gz##r = *reinterpret_cast<const float*>(recP + o##r);                           \
    if (!VZ) ga##r = *reinterpret_cast<const float4*>(recA + o##r);                             \
    else ga##r = make_float4(0.f, 0.f, 0.f, *reinterpret_cast<const float*>(recA + o##r + 12));
#define CONSUME(r)                                                                              \
    {                                                                                           \
        const float ux = m##r[1], uy = m##r[2], uz = m##r[3];                                   \
        const float t3 = gz##r * acc[2][r];                                                     \
        sx += ga##r.w * acc[0][r];                                                              \
        if (!VZ) { sa += ga##r.x * acc[1][r]; sb += ga##r.y * acc[1][r]; sc += ga##r.z * acc[1][r]; } \
        ra += t3 * ux; rb += t3 * uy; rc += t3 * uz;                                            \
    }
            GATHER(0) GATHER(1) GATHER(2) GATHER(3)

            const float env256 = env * 256.0f;
            f32x16 acc[3];
            {
                half8 aone = {0, 0, 0, 0, 0, 0, 0, 0};
                if (hi == 0) { aone[0] = (_Float16)256.0f; aone[1] = (_Float16)256.0f; }
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (VZ && b == 1) continue;
                    const half8 bb = *reinterpret_cast<const half8*>(Wh + (size_t)(b * 64 + wrow) * MSG_LDK + 128);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aone, bb, zero16, 0, 0, 0);
                }
            }
            {
                const float xsq = xs * p.sarg;
                int k0 = klo;
                do {
                    half8 ah, al;
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
                    const _Float16* wh = Wh + (size_t)wrow * MSG_LDK + k0 + 8 * hi;
                    const _Float16* wl = Wlo + (size_t)wrow * MSG_LDK + k0 + 8 * hi;
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        if (VZ && b == 1) continue;   // vec == 0: the xb columns multiply P = vec * xb = 0
                        const half8 bh = *reinterpret_cast<const half8*>(wh + b * 64 * MSG_LDK);
                        const half8 bl = *reinterpret_cast<const half8*>(wl + b * 64 * MSG_LDK);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
                    }
                    k0 += 16;
                } while (k0 < khi);
            }
#if M4_AHEAD == 4
            GATHER(4) GATHER(5) GATHER(6) GATHER(7)
            CONSUME(0) CONSUME(1) CONSUME(2) CONSUME(3)
            GATHER(8) GATHER(9) GATHER(10) GATHER(11)
            CONSUME(4) CONSUME(5) CONSUME(6) CONSUME(7)
            GATHER(12) GATHER(13) GATHER(14) GATHER(15)
            CONSUME(8) CONSUME(9) CONSUME(10) CONSUME(11)
            CONSUME(12) CONSUME(13) CONSUME(14) CONSUME(15)
#else
            GATHER(4) GATHER(5)
            CONSUME(0) CONSUME(1)
            GATHER(6) GATHER(7)
            CONSUME(2) CONSUME(3)
            GATHER(8) GATHER(9)
            CONSUME(4) CONSUME(5)
            GATHER(10) GATHER(11)
            CONSUME(6) CONSUME(7)
            GATHER(12) GATHER(13)
            CONSUME(8) CONSUME(9)
            GATHER(14) GATHER(15)
            CONSUME(10) CONSUME(11)
            CONSUME(12) CONSUME(13) CONSUME(14) CONSUME(15)
#endif
#undef GATHER
#undef CONSUME
#undef ROW_OF
            if (last) {
                // ---- finish this target: scale, cross-half reduction, residuals, one write per row
                sx *= out_scale;
                sa = (sa * inv_sqrt3 + ra) * inv_sqrt_h;
                sb = (sb * inv_sqrt3 + rb) * inv_sqrt_h;
                sc = (sc * inv_sqrt3 + rc) * inv_sqrt_h;
                sx += __shfl_xor(sx, 32);
                sa += __shfl_xor(sa, 32);
                sb += __shfl_xor(sb, 32);
                sc += __shfl_xor(sc, 32);
                const size_t xo = (size_t)orow * H + c0 + q;
                const size_t vo = (size_t)orow * 3 * H + c0 + q;
                if (hi == 0) {
                    p.x_out[xo] = (res0 + sx) * inv_sqrt2;
                    p.vec_out[vo] = res1 + sa;
                } else {
                    p.vec_out[vo + H] = res0 + sb;
                    p.vec_out[vo + 2 * H] = res1 + sc;
                }
                sx = sa = sb = sc = 0.f;
                ra = rb = rc = 0.f;
                res0 = res1 = 0.f;
                have = haveN;
                n = nN; orow = oN; eb = e0N; e1 = e1N;
                first = true;
                if (have) {
                    haveN = fetch_target(nN, oN);
                    if (haveN) { e0N = p.nptr[nN]; e1N = p.nptr[nN + 1]; }
                }
            } else {
                eb += 32;
                first = false;
            }
            geo = geoN; src = srcN; valid = validN;
        }
    }
    if (p.kcount && lane == 0 && jsel == 0) atomicAdd(p.kcount, (unsigned long long)ksteps);
}

static size_t m4_lds_bytes() {
    return (size_t)2 * MSG_COLS * MSG_LDK * 2 + sizeof(float) * (128 + M4_WAVES * 32 * 8) + 16;
}

int32_t adf_message4_prepare() {
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(adf_message4_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)m4_lds_bytes()));
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(adf_message4_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)m4_lds_bytes()));
    return ADF_OK;
}

int32_t adf_message4_launch(const MsgParams& p, int num_cus, bool vec_is_zero, hipStream_t s) {
    int workers = num_cus / p.nslices;
    if (workers < 1) workers = 1;
    if (workers > p.G) workers = p.G;
    dim3 grid((unsigned)(workers * p.nslices));
    if (vec_is_zero) hipLaunchKernelGGL(adf_message4_kernel<true>, grid, dim3(M4_THREADS), m4_lds_bytes(), s, p);
    else hipLaunchKernelGGL(adf_message4_kernel<false>, grid, dim3(M4_THREADS), m4_lds_bytes(), s, p);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
