// PaiNN message block, second-generation kernel (f16x3 arithmetic): same mathematics and the same results layout
// as message.hip (reference: adsorbdiff/models/painn/painn_denoising.py:530-567 + residual :443-445 and
// gemnet_oc/layers/radial_basis.py:18-43,64-82,235-244), restructured around three measurements on MI355X
// (scratch/ubench/issue.hip, profiles/r02_issue_rates.txt):
//   * a wave64 VALU instruction costs the SIMD ~4.8 cycles however many waves share it, so the ~650 VALU
//     instructions per 32-edge x 64-channel block of message.hip alone cost more than its 60 MFMAs (1920 cycles);
//   * MFMA and VALU overlap almost perfectly when they alternate inside ONE wave's instruction stream
//     (max(32, 4.8 V) cycles per MFMA + V VALU), but poorly across two waves of a SIMD (a wave in its matrix phase
//     beside a wave in its vector phase ran slower than the two phases back to back);
//   * the gathers are capped by the CU's vector-memory path (~28 TB/s chip-wide for this pattern).
// So: (1) the radial-basis MFMA operand is no longer evaluated in the kernel (8 x {sub, mul, exp2, mul, 2 cvt, sub, cvt}
// per lane and k-step, redone by every channel slice and layer): adf_atab_kernel (below) writes it once per graph
// build as fp16 hi/lo fragments per edge and the kernel loads them (2 x 16 B per lane and k-step, L2-resident because
// all channel slices of a target chunk run on the same XCD); (2) a wave owns 32 edge rows x 32 channels and walks
// its three column tiles (b, a, c parts of rbf_proj) one after the other: while the matrix core fills one 32x32
// accumulator tile the wave's vector instructions consume the previous one, so only two tiles (32 registers) are
// live and MFMA / FMA / gather issue share one instruction stream; (3) gathered records live in a rolling register
// window that is refilled for the next block row by row as it is consumed (one block period of latency tolerance).
//
// Gather record of (source atom, 32 channels), 640 B: [32 x (P0, P1, P2, xa)] + [32 x xc], P_i = vec_i * xb
// (gemm16.hip EPI 1).  Tile order b -> a -> c: tile b consumes quad.xyz, tile a quad.w (after which the quad of the
// NEXT block's same row is requested into the same registers), tile c the xc dword (same rolling refill).
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define M32_THREADS 256
#define M32_WAVES 4
#define M32_COLS 96
#define M32_LDK 136   // halves per column row of the weight image (128 + 8: bias pair at 128/129, conflict-free b128 reads)
#define M32_KSP 4     // k-steps whose A fragments are prefetched one block ahead (more steps: loaded on demand)

struct Msg32Params {
    const float* rec;
    const float* vec;
    const float* x;
    float* x_out;
    float* vec_out;
    const int32_t* tlist;
    int items;
    const int32_t* nptr;
    const int32_t* e_src;
    const float4* e_geom;
    const unsigned char* atab;       // [E][3 groups][hi 8 halves | lo 8 halves]
    const unsigned char* atab_zero;  // 32 zero bytes
    const _Float16* wpack16;         // this layer: [slice64][hi|lo][192][R]
    const float* bpack;              // this layer: [slice64][192] bias * scale as (hi, lo) half pairs
    const float* inv_scale;
    int N, H, R, G, nslices, wpx, Gx;
    float inv_cutoff, umax_scale;
    unsigned long long* kcount;
};

__device__ __forceinline__ int atab_kbase(float d, float inv_cutoff, float umax_scale) {
    // first k (multiple of 8) of the three 8-wide groups stored for an edge: covers every k with |k - u| < 6
    const float u = __fmul_rn(__fmul_rn(d, inv_cutoff), umax_scale);
    return max(0, (int)floorf(u) - 5) & ~7;
}

// ---- A table: 256 * env(d/rc) * exp(-(d/rc - mu_k)^2 / (2 sigma^2)) as fp16 hi/lo, 24 k per edge -------------------------
// One thread per (edge, group of 8 k).  Same arithmetic as message.hip's in-register evaluation: the 2^8 lift keeps
// a_lo a normal fp16 number for every term that matters (the matrix core flushes fp16 subnormals).
__global__ void adf_atab_kernel(const float4* __restrict__ e_geom, const int32_t* __restrict__ nptr, int N,
                                const float* __restrict__ mu, int R, float inv_cutoff, float umax_scale, float sarg,
                                float env_a, float env_b, float env_c, int env_pi, unsigned char* __restrict__ atab) {
    const long long E = nptr[N];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < 3 * E; t += (long long)gridDim.x * blockDim.x) {
        const long long e = t / 3;
        const int grp = (int)(t - 3 * e);
        const float d = e_geom[e].w;
        const int k0 = atab_kbase(d, inv_cutoff, umax_scale) + 8 * grp;
        const float xs = d * inv_cutoff;
        float xp = xs;
        for (int i = 1; i < env_pi; ++i) xp *= xs;
        float env = 1.0f + env_a * xp + env_b * (xp * xs) + env_c * (xp * xs * xs);
        env = xs < 1.0f ? env : 0.0f;
        const float env256 = env * 256.0f;
        const float xsq = xs * sarg;
        half8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            float a = 0.f;
            if (k < R) {
                const float dm = xsq - mu[k] * sarg;
                a = env256 * __builtin_amdgcn_exp2f(-(dm * dm));
            }
            const _Float16 hh = (_Float16)a;
            h[j] = hh;
            l[j] = (_Float16)(a - (float)hh);
        }
        half8* out = reinterpret_cast<half8*>(atab + (size_t)t * 32);
        out[0] = h;
        out[1] = l;
    }
}

int32_t adf_build_atab(adf_painn* h, int N, hipStream_t s) {
    const int R = h->hp.num_rbf;
    const double step = 1.0 / (R - 1);
    const float sarg = (float)sqrt(0.5 / (step * step) * 1.4426950408889634);
    const double pe = (double)h->hp.envelope_exponent;
    hipLaunchKernelGGL(adf_atab_kernel, dim3(256 * 16), dim3(256), 0, s, h->e_geom, h->nptr, N, h->rbf_offset, R,
                       1.0f / h->hp.cutoff, (float)(R - 1), sarg, (float)(-(pe + 1) * (pe + 2) / 2),
                       (float)(pe * (pe + 2)), (float)(-pe * (pe + 1) / 2), h->hp.envelope_exponent, h->atab + 64);
    ADF_HIP_CHECK(hipGetLastError());
    h->atab_valid = true;
    return ADF_OK;
}

#define ROW_OF(r) (((r) & 3) + 8 * ((r) >> 2) + 4 * hi)

// uniform base + 32-bit byte offset: the scalar-base addressing form (no per-lane 64-bit pointers to keep alive)
template <typename T>
__device__ __forceinline__ T ld32(const void* base, unsigned int off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off);
}
#ifndef M32_ABL
#define M32_ABL 0  // development only: bit 0 = A fragments not reloaded, bit 1 = records not gathered (wrong results)
#endif
// gather of one record piece; with the ablation bit the register keeps its value (the asm keeps the address alive)
#define M32_GATHER(dst, T, off)                                                    \
    if (M32_ABL & 2) { asm volatile("" ::"v"(off)); } else { dst = ld32<T>(recB, off); }
// CSR bounds of a target by a SCALAR load.  hipcc proves the address uniform and wants the result in SGPRs; written
// as a plain load it emits global_load + s_waitcnt vmcnt(0) + v_readfirstlane on the spot - a full drain of the
// gather pipeline per target (measured: 1300 cycles per block).  The scalar load counts on lgkmcnt instead.  Load
// and wait sit in ONE asm statement: the compiler does not know that an asm output arrives later and may copy or
// spill the SGPR pair in between (seen: wild CSR bounds -> memory fault), so the ~200-cycle scalar-cache round trip
// per target is paid in place.
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i32x2 sload2(const int32_t* ptr) {
    i32x2 v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
    return v;
}

template <typename T>
__device__ __forceinline__ void st32(void* base, unsigned int off, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off) = v;
}

// VZ: vec is identically zero on entry (layer 0): tile b (vec * b) is not computed at all, no quad.xyz use.
template <bool VZ>
__global__ __launch_bounds__(M32_THREADS, 2) void adf_message32_kernel(Msg32Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    _Float16* Wh = reinterpret_cast<_Float16*>(lds);                 // [96][LDK] hi
    _Float16* Wlo = Wh + M32_COLS * M32_LDK;                         // [96][LDK] lo
    float* Meta = reinterpret_cast<float*>(Wlo + M32_COLS * M32_LDK);  // [4 waves][2 buffers][off | ux | uy | uz][32 rows]
    int* Ctr = reinterpret_cast<int*>(Meta + M32_WAVES * 2 * 128);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane & 31;
    const int hi = lane >> 5;
    // blockIdx -> (xcd, slice, worker): every slice of a target chunk runs on the chunk's XCD (A table and records
    // are then shared in that XCD's L2)
    const int xcd = blockIdx.x & 7;
    const int local = blockIdx.x >> 3;
    const int slice = local % p.nslices;   // 32-channel slice
    const int worker = local / p.nslices;  // < wpx
    const int H = p.H;
    const int c0 = slice * 32;

    {   // stage this slice's rbf_proj image: columns part*64 + 32*(slice&1) + q of the 64-channel pack -> part*32 + q
        const int s64 = slice >> 1, jh = slice & 1;
        const int R8 = p.R / 8;
        const half8* src = reinterpret_cast<const half8*>(p.wpack16 + (size_t)s64 * 2 * 192 * p.R);
        const _Float16* b16 = reinterpret_cast<const _Float16*>(p.bpack) + (size_t)s64 * 192 * 2;
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < 2 * M32_COLS * 17; i += M32_THREADS) {
            const int row = i / 17, piece = i - row * 17;      // row in [0, 192): hi rows then lo rows
            const int hl = row / M32_COLS, col = row - hl * M32_COLS;
            const int part = col >> 5, qq = col & 31;
            const int scol = part * 64 + 32 * jh + qq;
            half8 v = piece < R8 ? src[(size_t)(hl * 192 + scol) * R8 + piece] : zero8;
            if (piece == 16 && hl == 0) { v[0] = b16[2 * scol]; v[1] = b16[2 * scol + 1]; }
            *reinterpret_cast<half8*>(Wh + (size_t)row * M32_LDK + piece * 8) = v;
        }
        // meta buffers start zeroed: the first block's pipeline consumes a non-existent previous tile as 0 * (unit
        // vector read from the still unwritten buffer) - leftover LDS bits may be NaN / inf
        for (int i = tid; i < M32_WAVES * 256; i += M32_THREADS) Meta[i] = 0.f;
        if (tid == 0) *Ctr = 0;
    }
    __syncthreads();

    float* meta_w = Meta + wave * 256;
    const int mrow = 4 * hi;  // + 8 g: first of the 4 consecutive rows (accumulator registers 4g..4g+3) of this lane
    const float inv_sqrt3 = 0.57735026918962576f;
    const float inv_sqrt2 = 0.70710678118654752f;
    const float out_scale = *p.inv_scale * (1.0f / 256.0f);  // accumulators hold 256 * scale * rbfh
    const float inv_sqrt_h = out_scale / sqrtf((float)H);
    const unsigned int row_bytes = (unsigned int)(H / 32) * 640u;
    // record addresses = uniform base + 32-bit offset (record-row offset from the meta + this lane's part): the loads
    // use the scalar-base addressing form, one v_add_u32 per gather (the host checks that the table is < 4 GiB)
    const char* recB = reinterpret_cast<const char*>(p.rec);
    const unsigned int laneQ = (unsigned int)slice * 640u + (unsigned int)q * 16u;
    const unsigned int laneX = (unsigned int)slice * 640u + 512u + (unsigned int)q * 4u;
    const _Float16* wbase = Wh + (size_t)q * M32_LDK + 8 * hi;  // + part*32*LDK + k0 ; lo image at + 96*LDK
    unsigned int ksteps = 0;
#ifdef M32_STAMP  // development: wave cycles per segment of the block body, summed into p.kcount[1..7]
    unsigned long long st_[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
#define STAMP(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; }
#else
#define STAMP(i)
#endif
    half8 aone = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hi == 0) { aone[0] = (_Float16)256.0f; aone[1] = (_Float16)256.0f; }

    // ---- target / block sequence -------------------------------------------------------------------------------
    // items = target atoms in groups of 32; this workgroup (xcd, worker) owns groups xcd*Gx + worker + t*wpx
    int static_t = wave; (void)static_t;
    auto fetch_target = [&](int& n_out, int& o_out) -> bool {
        while (true) {
#ifdef M32_STATIC
            int t = static_t; static_t += M32_WAVES;
#else
            int t = 0;
            if (lane == 0) t = atomicAdd(Ctr, 1);
            t = __builtin_amdgcn_readfirstlane(t);
#endif
            const int gl = worker + (t >> 5) * p.wpx;
            const int g = xcd * p.Gx + gl;
            if (gl >= p.Gx || g >= p.G) return false;
            const int e = g * 32 + (t & 31);
            if (e < p.items) { o_out = e; n_out = p.tlist ? __builtin_amdgcn_readfirstlane(p.tlist[e]) : e; return true; }
        }
    };
    // generator state: current target (edges gen_e .. gen_e1) and the one after it (bounds requested one target ahead)
    int gen_n = 0, gen_o = 0, gen_e = 0, gen_e1 = 0;
    bool gen_have = fetch_target(gen_n, gen_o);
    if (gen_have) { const i32x2 b_ = sload2(p.nptr + gen_n); gen_e = b_[0]; gen_e1 = b_[1]; }
    int nxt_n = 0, nxt_o = 0;
    i32x2 nxt_b = {0, 0};  // bounds of the next target
    bool nxt_have = gen_have && fetch_target(nxt_n, nxt_o);
    if (nxt_have) nxt_b = sload2(p.nptr + nxt_n);

    struct Blk { int eb, nv, n, orow; bool last, valid; };
    auto gen_next = [&]() -> Blk {
        Blk b;
        b.valid = gen_have; b.eb = gen_e; b.n = gen_n; b.orow = gen_o;
        b.nv = gen_have ? max(0, min(32, gen_e1 - gen_e)) : 0;
        b.last = gen_have && (gen_e + 32 >= gen_e1);
        if (b.last) {
            gen_have = nxt_have; gen_n = nxt_n; gen_o = nxt_o;
            if (gen_have) {
                gen_e = nxt_b[0]; gen_e1 = nxt_b[1];
                nxt_have = fetch_target(nxt_n, nxt_o);
                if (nxt_have) nxt_b = sload2(p.nptr + nxt_n);
            }
        } else {
            gen_e += 32;
        }
        return b;
    };
    auto load_geo = [&](const Blk& b, float4& geo, int& src) {
        geo = make_float4(0.f, 0.f, 0.f, 0.f);
        src = 0;
        if (q < b.nv) {
            geo = ld32<float4>(p.e_geom, (unsigned int)(b.eb + q) * 16u);
            src = ld32<int>(p.e_src, (unsigned int)(b.eb + q) * 4u);
        }
    };
    // window of a block from its rows' group bases (rows are sorted by distance): [klo, klo + 16 ks)
    auto window = [&](const Blk& b, const float4& geo, int& kb, int& klo, int& ks) {
        kb = atab_kbase(geo.w, p.inv_cutoff, p.umax_scale);
        if (b.nv <= 0) { klo = 0; ks = 1; return; }
        const float u = __fmul_rn(__fmul_rn(geo.w, p.inv_cutoff), p.umax_scale);
        klo = __builtin_amdgcn_readlane(kb, 0);
        const float ulast = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), b.nv - 1));
        int khi = min(min((int)floorf(ulast) + 7, p.R), 128);
        ks = max(1, (khi - klo + 15) >> 4);
        if (klo + 16 * ks > 136) ks = (136 - klo) >> 4;  // never read past the image row (klo <= 120)
    };
    // address of this lane's A fragment (row q, k = klo + 16 s + 8 hi ..+8) in the table, or the zero block
    // (byte offset from p.atab_zero: the 64 zero bytes sit in front of the table)
    auto atab_addr = [&](const Blk& b, int kb, int klo, int s) -> const half8* {
        const int idx = ((klo >> 3) + 2 * s + hi) - (kb >> 3);
        const bool ok = q < b.nv && idx >= 0 && idx < 3;
        const unsigned int off = ok ? 64u + ((unsigned int)(b.eb + q) * 3u + (unsigned int)idx) * 32u : 0u;
        return reinterpret_cast<const half8*>(p.atab_zero + off);
    };
    auto write_meta = [&](int buf, const Blk& b, const float4& geo, int src) {
        if (hi == 0) {
            const unsigned int off = (unsigned int)(q < b.nv ? src : p.N) * row_bytes;  // row N: all-zero record
            float* m = meta_w + buf * 128 + q;
            m[0] = __uint_as_float(off); m[32] = geo.x; m[64] = geo.y; m[96] = geo.z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- pipeline state ------------------------------------------------------------------------------------------
    f32x16 T0, T1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { T0[r] = 0.f; T1[r] = 0.f; }
    float4 gq[16];  // (P0, P1, P2, xa) of the block whose tiles b/a are consumed next
    float gx[16];   // xc of the block whose tile c is consumed next
#pragma unroll
    for (int r = 0; r < 16; ++r) { gq[r] = make_float4(0.f, 0.f, 0.f, 0.f); gx[r] = 0.f; }
    half8 Ah[M32_KSP], Al[M32_KSP];  // this block's fragments; refilled for the next block step by step in tile c
    float sx = 0.f, sa = 0.f, sb = 0.f, sc = 0.f, ra = 0.f, rb = 0.f, rc = 0.f;
    // residual inputs of the target rows: (x | vec_y, vec_x | vec_z) by half-wave; N = requested for the current block,
    // P = of the previous block (consumed when its target is finished)
    float resPx = 0.f, resPy = 0.f, resPz = 0.f, resNx = 0.f, resNy = 0.f, resNz = 0.f;

    Blk bP; bP.valid = false; bP.last = false; bP.eb = bP.nv = bP.n = bP.orow = 0;
    Blk b0 = gen_next();
    Blk b1 = gen_next();
    float4 geo0, geo1; int src0, src1;
    load_geo(b0, geo0, src0);
    load_geo(b1, geo1, src1);
    int kb0, klo0, ks0;
    window(b0, geo0, kb0, klo0, ks0);
    // prologue: meta, A fragments and gathers of block 0 (not overlapped with anything)
    write_meta(0, b0, geo0, src0);
#pragma unroll
    for (int s = 0; s < M32_KSP; ++s) {
        const half8* a = atab_addr(b0, kb0, klo0, s < ks0 ? s : 1000);
        Ah[s] = a[0]; Al[s] = a[1];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned int off = __float_as_uint(meta_w[ROW_OF(r)]);
        if (!VZ) gq[r] = ld32<float4>(recB, off + laneQ);
        else gq[r].w = ld32<float>(recB, off + laneQ + 12u);
    }
    // Requested for EVERY block, branch-free and straight into the loop-carried registers: a conditional load (only
    // for a target's last block) ends in a register copy of the fresh value at the loop edge, i.e. a vmcnt(0) drain
    // of the whole gather pipeline per block.  rx: x (half-wave 1 reads a valid dummy), ry: vec_y, rz: vec_x | vec_z.
    auto load_res = [&](const Blk& b, float& rx, float& ry, float& rz) {
        const unsigned int xo = ((unsigned int)b.n * H + c0 + q) * 4u;
        const unsigned int vo = ((unsigned int)b.n * 3u * H + c0 + q) * 4u;
        rx = ld32<float>(p.x, xo);
        if (!VZ) { ry = ld32<float>(p.vec, vo + H * 4u); rz = ld32<float>(p.vec, vo + (unsigned int)hi * (H * 8u)); }
    };
    load_res(b0, resNx, resNy, resNz);

    // one accumulator tile: bias MFMA + 3 products per k-step of the window; `work(i)` is called between MFMAs
    // with i = 0..15 (consume / gather work of one accumulator row each), in program order
    auto finish_target = [&](const Blk& b) {
        float fx = __fmul_rn(sx, out_scale);
        float fa = __fmul_rn(__fmaf_rn(sa, inv_sqrt3, ra), inv_sqrt_h);
        float fb = __fmul_rn(__fmaf_rn(sb, inv_sqrt3, rb), inv_sqrt_h);
        float fc = __fmul_rn(__fmaf_rn(sc, inv_sqrt3, rc), inv_sqrt_h);
        fx += __shfl_xor(fx, 32); fa += __shfl_xor(fa, 32); fb += __shfl_xor(fb, 32); fc += __shfl_xor(fc, 32);
        const unsigned int xo = ((unsigned int)b.orow * H + c0 + q) * 4u;
        const unsigned int vo = ((unsigned int)b.orow * 3u * H + c0 + q) * 4u;
        if (hi == 0) {
            st32<float>(p.x_out, xo, __fmul_rn(__fadd_rn(resPx, fx), inv_sqrt2));
            st32<float>(p.vec_out, vo, resPz + fa);
        } else {
            st32<float>(p.vec_out, vo + H * 4u, resPy + fb);
            st32<float>(p.vec_out, vo + H * 8u, resPz + fc);
        }
        sx = sa = sb = sc = ra = rb = rc = 0.f;
    };

#define SB_ __builtin_amdgcn_sched_barrier(0);
#define M32_LOADB(BH, BL, PART, S)                                                                            \
    {                                                                                                         \
        const _Float16* w_ = wbase + (PART) * 32 * M32_LDK + min(klo0 + 16 * (S), 120);                       \
        BH = *reinterpret_cast<const half8*>(w_);                                                             \
        BL = *reinterpret_cast<const half8*>(w_ + M32_COLS * M32_LDK);                                        \
    }
#define M32_MFMA3(T, AH, AL, BH, BL)                                                                          \
    T = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, T, 0, 0, 0);                                           \
    T = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, T, 0, 0, 0);                                           \
    T = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, T, 0, 0, 0);
// One accumulator tile = bias + window contraction.  A wave issues in order, so MFMAs placed back to back would
// hold up the vector work behind them (the second one waits 32 cycles for the pipe): every MFMA is followed by one
// row of vector work WORK(r) (~6 instructions ~ one MFMA time), the pattern that overlapped fully in the
// measurement.  Weight fragments of a k-step are read from LDS one step ahead of their use; POST(s) follows the
// MFMAs of step s.  Steps 0..2 always run (beyond the block's window on zero fragments: branch-free), step 3 only
// for wide windows, further steps in the loop at the end (fragments loaded on demand: rare).
#define M32_M1(T, A_, B_) T = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, T, 0, 0, 0);
#define M32_TILE(T, PART, AH, AL, WORK, POST)                                                                 \
    {                                                                                                         \
        half8 bb_ = *reinterpret_cast<const half8*>(Wh + (size_t)((PART) * 32 + q) * M32_LDK + 128);          \
        half8 bh0_, bl0_, bh1_, bl1_;                                                                         \
        M32_LOADB(bh0_, bl0_, PART, 0)                                                                        \
        SB_ WORK(0) SB_                                                                                       \
        {   /* T = 256 * (b_hi + b_lo): A = 256 at k slots 0, 1 of half-wave 0, B = image slots 128.. */      \
            const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
            T = __builtin_amdgcn_mfma_f32_32x32x16_f16(aone, bb_, z16, 0, 0, 0);                              \
        }                                                                                                     \
        SB_ M32_LOADB(bh1_, bl1_, PART, 1) WORK(1) SB_                                                        \
        M32_M1(T, AL[0], bh0_) SB_ WORK(2) SB_                                                                \
        M32_M1(T, AH[0], bl0_) SB_ WORK(3) SB_                                                                \
        M32_M1(T, AH[0], bh0_) SB_ POST(0) M32_LOADB(bh0_, bl0_, PART, 2) WORK(4) SB_                         \
        M32_M1(T, AL[1], bh1_) SB_ WORK(5) SB_                                                                \
        M32_M1(T, AH[1], bl1_) SB_ WORK(6) SB_                                                                \
        M32_M1(T, AH[1], bh1_) SB_ POST(1) if (ks0 > 3) M32_LOADB(bh1_, bl1_, PART, 3) WORK(7) SB_            \
        M32_M1(T, AL[2], bh0_) SB_ WORK(8) SB_                                                                \
        M32_M1(T, AH[2], bl0_) SB_ WORK(9) SB_                                                                \
        M32_M1(T, AH[2], bh0_) SB_ POST(2) WORK(10) SB_                                                       \
        if (ks0 > 3) {                                                                                        \
            M32_M1(T, AL[3], bh1_) SB_ WORK(11) SB_                                                           \
            M32_M1(T, AH[3], bl1_) SB_ WORK(12) SB_                                                           \
            M32_M1(T, AH[3], bh1_) SB_                                                                        \
        } else {                                                                                              \
            WORK(11) SB_ WORK(12) SB_                                                                         \
        }                                                                                                     \
        POST(3)                                                                                               \
        WORK(13) SB_ WORK(14) SB_ WORK(15) SB_                                                                \
        for (int s_ = M32_KSP; s_ < ks0; ++s_) { /* very wide windows: fragments loaded on demand */          \
            const half8* a_ = atab_addr(b0, kb0, klo0, s_);                                                   \
            const half8 ah_ = a_[0], al_ = a_[1];                                                             \
            M32_LOADB(bh0_, bl0_, PART, s_)                                                                   \
            M32_MFMA3(T, ah_, al_, bh0_, bl0_)                                                                \
        }                                                                                                     \
    }

    // The rows' meta words (unit vectors, record offsets) come from LDS in batches of 4 rows = the 4 consecutive rows
    // behind accumulator registers 4g..4g+3, one 16-B read per array, requested one batch (4 rows of work) ahead of
    // their use: a dependent LDS round trip per row would stall the in-order wave 16 times per tile.
    // Explicit fma / mul: the three instances of this code (two tile parities + drain) must round identically.
#define PRE_C(g)                                                                                              \
    {                                                                                                         \
        const float* mp_ = meta_w + pbuf * 128 + 8 * (g) + mrow;                                              \
        UX[(g) & 1] = *reinterpret_cast<const f32x4*>(mp_ + 32);                                              \
        UY[(g) & 1] = *reinterpret_cast<const f32x4*>(mp_ + 64);                                              \
        UZ[(g) & 1] = *reinterpret_cast<const f32x4*>(mp_ + 96);                                              \
        OC[(g) & 1] = *reinterpret_cast<const u32x4*>(meta_w + cbuf * 128 + 8 * (g) + mrow);                  \
    }
    // consume tile c of the previous block (pbuf = its unit vectors), then request xc of the current block
#define WORK_C(r)                                                                                             \
    {                                                                                                         \
        if (((r) & 3) == 0 && (r) < 12) PRE_C(((r) >> 2) + 1)                                                 \
        const float t_ = __fmul_rn(gx[r], TB[r]);                                                             \
        ra = __fmaf_rn(t_, UX[((r) >> 2) & 1][(r) & 3], ra);                                                  \
        rb = __fmaf_rn(t_, UY[((r) >> 2) & 1][(r) & 3], rb);                                                  \
        rc = __fmaf_rn(t_, UZ[((r) >> 2) & 1][(r) & 3], rc);                                                  \
        asm volatile("" : "+v"(ra), "+v"(rb), "+v"(rc));  /* keep the FMAs here (not sunk behind the MFMAs) */    \
        M32_GATHER(gx[r], float, OC[((r) >> 2) & 1][(r) & 3] + laneX)                          \
    }
#define WORK_B(r)                                                                                             \
    {                                                                                                         \
        sa = __fmaf_rn(gq[r].x, TA[r], sa); sb = __fmaf_rn(gq[r].y, TA[r], sb); sc = __fmaf_rn(gq[r].z, TA[r], sc); \
        asm volatile("" : "+v"(sa), "+v"(sb), "+v"(sc));                                                      \
    }
#define PRE_A(g) OA[(g) & 1] = *reinterpret_cast<const u32x4*>(meta_w + pbuf * 128 + 8 * (g) + mrow);
    // consume tile a of the current block, then request the next block's quad into the same registers
#define WORK_A(r)                                                                                             \
    {                                                                                                         \
        if (((r) & 3) == 0 && (r) < 12) PRE_A(((r) >> 2) + 1)                                                 \
        sx = __fmaf_rn(gq[r].w, TB[r], sx);                                                                   \
        asm volatile("" : "+v"(sx));                                                                          \
        if (!VZ) { M32_GATHER(gq[r], float4, OA[((r) >> 2) & 1][(r) & 3] + laneQ) }                           \
        else { M32_GATHER(gq[r].w, float, OA[((r) >> 2) & 1][(r) & 3] + laneQ + 12u) }                        \
    }
#define WORK_NONE(r)
#define POST_NONE(s)
    // after tile c's step s: this block's fragment s is dead, request the next block's into the same registers
#define POST_A(s)                                                                                             \
    {                                                                                                         \
        const half8* a_ = atab_addr(b1, kb1, klo1, (s) < ks1 ? (s) : 1000);                                   \
        if (M32_ABL & 1) { asm volatile("" ::"v"(a_)); } else { Ah[s] = a_[0]; Al[s] = a_[1]; }               \
    }

    // One block.  TA / TB alternate between calls (3 tiles per block).
    // cbuf = meta buffer of this block; pbuf = the other one (previous block's until it is rewritten for the next).
    auto body = [&](f32x16& TA, f32x16& TB, const int cbuf) __attribute__((always_inline)) {
        const int pbuf = cbuf ^ 1;
        STAMP(6)
        // -- S0: look two blocks ahead; window and A fragments of the next block; residual rows
        const Blk b2 = gen_next();
        float4 geo2; int src2;
        load_geo(b2, geo2, src2);
        int kb1, klo1, ks1;
        window(b1, geo1, kb1, klo1, ks1);
        ksteps += ks0;
        f32x4 UX[2], UY[2], UZ[2];
        u32x4 OC[2], OA[2];
        PRE_C(0)
        STAMP(0)
        // -- S1: tile b of this block (not in the vec == 0 layer) while tile c of the previous block is consumed
        if (!VZ) {
            M32_TILE(TA, 1, Ah, Al, WORK_C, POST_NONE)
        } else {
            WORK_C(0) WORK_C(1) WORK_C(2) WORK_C(3) WORK_C(4) WORK_C(5) WORK_C(6) WORK_C(7)
            WORK_C(8) WORK_C(9) WORK_C(10) WORK_C(11) WORK_C(12) WORK_C(13) WORK_C(14) WORK_C(15)
        }
        STAMP(1)
        if (bP.valid && bP.last) finish_target(bP);
        STAMP(2)
        // -- S2: the previous block's meta is dead now: write the next block's into its buffer
        write_meta(pbuf, b1, geo1, src1);
        PRE_A(0)
        STAMP(3)
        // -- S3: tile a while tile b is consumed
        if (!VZ) {
            M32_TILE(TB, 0, Ah, Al, WORK_B, POST_NONE)
        } else {
            M32_TILE(TB, 0, Ah, Al, WORK_NONE, POST_NONE)
        }
        STAMP(4)
        // -- S4: tile c while tile a is consumed and the next block's quads are requested
        M32_TILE(TA, 2, Ah, Al, WORK_A, POST_A)
        STAMP(5)
        // -- rotate
        bP = b0; b0 = b1; b1 = b2;
        geo1 = geo2; src1 = src2;
        kb0 = kb1; klo0 = klo1; ks0 = ks1;
        resPx = resNx; resPy = resNy; resPz = resNz;
        load_res(b0, resNx, resNy, resNz);
    };

    // resN of block 0 was loaded in the prologue; body() loads the residual rows of the block that becomes current
    bool parity = false;
    while (b0.valid) {
        body(T0, T1, 0);
        parity = true;
        if (!b0.valid) break;
        body(T1, T0, 1);
        parity = false;
    }
    // drain: tile c of the last block
    {
        const int pbuf = parity ? 0 : 1;  // meta buffer of the last block (cbuf of the last body call)
        const int cbuf = pbuf;
#define DRAIN                                                                                                 \
    WORK_C(0) WORK_C(1) WORK_C(2) WORK_C(3) WORK_C(4) WORK_C(5) WORK_C(6) WORK_C(7)                          \
    WORK_C(8) WORK_C(9) WORK_C(10) WORK_C(11) WORK_C(12) WORK_C(13) WORK_C(14) WORK_C(15)
        if (bP.valid) {
            f32x4 UX[2], UY[2], UZ[2];
            u32x4 OC[2];
            PRE_C(0)
            if (parity) { f32x16& TB = T0; DRAIN } else { f32x16& TB = T1; DRAIN }
            if (bP.last) finish_target(bP);
        }
    }
    if (p.kcount && lane == 0) atomicAdd(p.kcount, (unsigned long long)ksteps * 8ull);
#ifdef M32_STAMP
    if (p.kcount && lane == 0)
        for (int i = 0; i < 7; ++i) atomicAdd(p.kcount + 1 + i, st_[i]);
#endif
}

static size_t m32_lds_bytes() {
    static long pad = -1;  // development: ADF_M32_PAD=<bytes> of extra LDS limits the workgroups per CU
    if (pad < 0) { const char* e = getenv("ADF_M32_PAD"); pad = e ? atol(e) : 0; }
    return (size_t)2 * M32_COLS * M32_LDK * 2 + sizeof(float4) * M32_WAVES * 2 * 32 + 16 + (size_t)pad;
}

int32_t adf_message32_prepare() {
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(adf_message32_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)m32_lds_bytes()));
    ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(adf_message32_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)m32_lds_bytes()));
    return ADF_OK;
}

int32_t adf_message32_impl(adf_painn* h, int layer, int N, const float* x, const float* vec, float* x_out,
                           float* vec_out, bool vec_is_zero, hipStream_t s, const int32_t* tlist, int n_targets,
                           const float* rec) {
    const int H = h->hp.hidden_channels, R = h->hp.num_rbf;
    if ((unsigned long long)(N + 1) * 5ull * H * sizeof(float) >= (1ull << 32)) {
        adf_set_error("message kernel uses 32-bit byte offsets into the node tables: N=%d is too large, split the batch", N);
        return ADF_EOOM;
    }
    if ((unsigned long long)h->capE * 96ull + 64ull >= (1ull << 32)) {
        adf_set_error("message kernel uses 32-bit byte offsets into the edge tables: %lld edge slots are too many, split the batch",
                      (long long)h->capE);
        return ADF_EOOM;
    }
    if (!h->atab_valid) ADF_TRY(adf_build_atab(h, N, s));
    Msg32Params p;
    p.rec = rec ? rec : h->rec; p.vec = vec; p.x = x; p.x_out = x_out; p.vec_out = vec_out;
    p.nptr = h->nptr; p.e_src = h->e_src; p.e_geom = h->e_geom;
    p.atab = h->atab + 64; p.atab_zero = h->atab;
    const int ns64 = H / ADF_SLICE_CH;
    p.wpack16 = reinterpret_cast<const _Float16*>(h->rbf_pack16) + (size_t)layer * 2 * ns64 * R * 192;
    p.bpack = h->rbf_bias_pack16 + (size_t)layer * ns64 * 192;
    p.inv_scale = h->rbf_scales + layer;
    p.N = N; p.H = H; p.R = R;
    p.tlist = tlist; p.items = tlist ? n_targets : N;
    if (p.items <= 0) return ADF_OK;
    p.G = (p.items + 31) / 32;
    p.nslices = H / 32;
    p.Gx = (p.G + 7) / 8;
    int wpx = (2 * h->num_cus / 8) / p.nslices;
    if (wpx < 1) wpx = 1;
    if (wpx > p.Gx) wpx = p.Gx;
    p.wpx = wpx;
    p.inv_cutoff = 1.0f / h->hp.cutoff;
    p.umax_scale = (float)(R - 1);
    p.kcount = h->prof_on ? h->kcount : nullptr;
    dim3 grid((unsigned)(8 * p.nslices * wpx));
    if (vec_is_zero)
        hipLaunchKernelGGL((adf_message32_kernel<true>), grid, dim3(M32_THREADS), m32_lds_bytes(), s, p);
    else
        hipLaunchKernelGGL((adf_message32_kernel<false>), grid, dim3(M32_THREADS), m32_lds_bytes(), s, p);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
