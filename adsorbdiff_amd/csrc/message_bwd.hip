// Backward of the PaiNN message block for the training step (BASELINE config 5), fused like the forward
// (message.hip): the radial-basis projection rbfh = rbf_proj(edge_rbf) is REGENERATED on the matrix cores inside the kernel
// instead of being kept from the forward ([E,3H] fp32 = 6 KB per edge and layer) or recomputed by a dense product.
//
// Reference: torch.autograd through adsorbdiff/models/painn/painn_denoising.py:530-567 (PaiNNMessage.forward / .message):
//   x_ji = xh[j] * rbfh(e);  (a, b, c) = split(x_ji);  dx[i] += a;  dvec[i] += (vec[j] * b / sqrt3 + c (x) r_hat) / sqrtH
// The graph is symmetric (symmetrize_edges, gemnet_oc utils): the CSR segment of atom j lists its neighbours i, and the
// reverse edge j -> i has the same distance (same rbfh row) and the opposite unit vector.  One wave owns all edges of its
// atom j in 32-row blocks exactly as in the forward; per edge and channel, with gx = d(x1)[i] / sqrt2, g = d(vec1)[i] / sqrtH
// (gathered from packed gradient records), w = vec[j] / sqrt3, u = unit vector, (ra, rb, rc) = the regenerated rbfh row:
//   S = g . w;  T = -(g . u)
//   dxh[j] += (gx ra, S rb, T rc);   dvec[j] += g (xb[j] rb / sqrt3);   drbfh[e] = (gx xa[j], S xb[j], T xc[j])
// drbfh does not depend on rbfh; it is written in the order this kernel holds it - per edge row [slice][6][32]: the a, b, c
// parts of channels 64 slice + q, then of channels 64 slice + 32 + q, q = 0..31 - so that every store instruction writes whole
// 128-byte lines (one per half-wave and value).  The weight-gradient product of rbf_proj runs on that column order and
// train_step.py permutes its 3H x R result back (adf_op_message_bwd_perm gives the map).  (Measured: 24 contiguous bytes per
// lane - [slice][q][6], one dwordx4 + one dwordx2 per row instead of six dwords - is 2.5 % of a training step SLOWER: partial
// lines.  Without the stores the step is 23 ms of 167 shorter: the kernel is bound by the 15 GB it writes.)
//
// f16x3 split arithmetic with equally spaced Gaussian centres only (the default; otherwise the training step keeps the
// unfused backward, train.hip tr_msg_bwd_kernel).
#include <stdlib.h>
#include <string.h>

#include "message.h"

struct MsgBwdParams {
    MsgParams m;          // rec = gradient records [(N+1)][H/32][160]: [32 x (g0, g1, g2, gx)] + [32 unused]; row N zero
    const float* xh;      // [N, 3H]
    const float* vec;     // [N, 3, H] or null (first layer: vec == 0)
    const float* gv1;     // [N, 3, H] gradient of vec1 (the residual path of dvec)
    float* dxh;           // [N, 3H]
    float* dvec;          // [N, 3, H] or null
    float* drbfh;         // [E + 1, 3H] in kernel order (row E: spare, written by padded edge rows); unused when !ST
    float* dbias_rows;    // !ST: [N, 3H] per-atom column sums of drbfh over the atom's edge rows ([a | b | c] order)
    int E;
};

#define MSGB_WAVES_PER_SIMD 2
#ifndef MSGB_STORE
#define MSGB_STORE 1   // layout of a drbfh row: 0 = [slice][q][6] (24 B per lane: dwordx4 + dwordx2), 1 = [slice][6][q] (six dwords)
#endif
#ifndef MSGB_AHEAD
#define MSGB_AHEAD 2   // gather rows requested ahead of the row being consumed (4: as the forward kernel)
#endif

// ST: drbfh is stored (the materialised path: tr_wgrad_bf16x6_kernel reads it back).  !ST: nothing per edge leaves the
// kernel - rbf_wgrad.hip forms drbfh again while it stages its product - and the bias gradient's column sums are handed
// over per atom (deterministic: an atom's rows are summed by one wave in row order, whichever wave pulls the atom).
template <bool VZ, bool ST>
__global__ __launch_bounds__(MSG_THREADS, MSGB_WAVES_PER_SIMD) void adf_message_bwd_kernel(MsgBwdParams pb) {
    const MsgParams& p = pb.m;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wfloats = (2 * MSG_COLS * MSG_LDK) / 2;
    _Float16* Wh = reinterpret_cast<_Float16*>(lds);       // [192][MSG_LDK] hi
    _Float16* Wlo = Wh + MSG_COLS * MSG_LDK;               // [192][MSG_LDK] lo
    float* Bl = lds + wfloats;
    float* Mu = Bl + MSG_COLS;
    float* Meta = Mu + 128;
    int* Ctr = reinterpret_cast<int*>(Meta + MSG_WAVES * 32 * 8);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = lane & 31;
    const int hi = lane >> 5;
    const int slice = blockIdx.x % p.nslices;
    const int worker = blockIdx.x / p.nslices;
    const int nworkers = gridDim.x / p.nslices;
    const int items = p.items;
    const int ngroups = (items + ADF_GROUP_NODES - 1) / ADF_GROUP_NODES;
    if (worker >= ngroups) return;
    const int H = p.H;
    const int c0 = slice * ADF_SLICE_CH;

    {   // stage this slice's rbf_proj image once (same image as the forward kernel)
        const int R8 = p.R / 8;
        const half8* src = reinterpret_cast<const half8*>(p.wpack16 + (size_t)slice * 2 * MSG_COLS * p.R);
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const _Float16* b16 = reinterpret_cast<const _Float16*>(p.bpack) + (size_t)slice * MSG_COLS * 2;
        for (int i = tid; i < 2 * MSG_COLS * 17; i += MSG_THREADS) {
            const int row = i / 17, piece = i - row * 17;
            half8 v = piece < R8 ? src[row * R8 + piece] : zero8;
            if (piece == 16 && row < MSG_COLS) { v[0] = b16[2 * row]; v[1] = b16[2 * row + 1]; }
            *reinterpret_cast<half8*>(Wh + (size_t)row * MSG_LDK + piece * 8) = v;
        }
        if (tid < 128) Mu[tid] = (tid < p.R ? p.mu[tid] : 2.0f) * p.sarg;
        if (tid == 0) *Ctr = 0;
    }
    __syncthreads();
    float* meta_w = Meta + wave * 32 * 8;
    const float inv_sqrt3 = 0.57735026918962576f;
    const float out_scale = *p.inv_scale * (1.0f / 256.0f);  // accumulators hold 256*scale*rbfh
    const float umax_scale = (float)(p.R - 1);
    const unsigned int row_bytes = (unsigned int)p.nslices * 1280u;
    // gathers: wave-uniform base + 32-bit lane offset (global_load ... v_off, s[base]), as in message.hip
    const char* recS = reinterpret_cast<const char*>(p.rec) + (size_t)slice * 1280;
    const unsigned int qA = (unsigned int)q * 16u;
    // kernel order of drbfh: row e, then [slice][6][32] (MSGB_STORE 1) or [slice][q][6] (0)
    const size_t drow_bytes = (size_t)3 * H * sizeof(float);
    char* dlane = reinterpret_cast<char*>(pb.drbfh) + (size_t)slice * 768 + (size_t)q * 24;
    char* dline = reinterpret_cast<char*>(pb.drbfh) + (size_t)slice * 768 + (size_t)q * 4;   // MSGB_STORE 1: [slice][6][32]

    auto fetch_target = [&](int& n_out) -> bool {
        while (true) {
            int t = 0;
            if (lane == 0) t = atomicAdd(Ctr, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            const int g = worker + (t >> 5) * nworkers;
            if (g >= ngroups) return false;
            const int e = g * ADF_GROUP_NODES + (t & 31);
            if (e < items) { n_out = e; return true; }
        }
    };
    auto load_block = [&](int eb, int e1, float4& geo, int& src, bool& valid) {
        const int e = eb + q;
        valid = e < e1;
        geo = make_float4(0.f, 0.f, 0.f, 0.f);
        src = 0;
        if (valid) { geo = p.e_geom[e]; src = p.e_src[e]; }
    };

    // CSR bounds through the constant address space (scalar loads), first block's rows laundered: see message.hip
    typedef const __attribute__((address_space(4))) int32_t* cint_ptr;
    const cint_ptr nptr_c = (cint_ptr)p.nptr;
    int n = 0, eb = 0, e1 = 0;
    bool have = fetch_target(n);
    if (have) { eb = nptr_c[n]; e1 = nptr_c[n + 1]; }
    int nN = 0, e0N = 0, e1N = 0;
    bool haveN = have && fetch_target(nN);
    if (haveN) { e0N = nptr_c[nN]; e1N = nptr_c[nN + 1]; }
    float4 geo = make_float4(0.f, 0.f, 0.f, 0.f); int src = 0; bool valid = false;
    if (have) load_block(eb, e1, geo, src, valid);
    asm volatile("" : "+v"(geo.x), "+v"(geo.y), "+v"(geo.z), "+v"(geo.w), "+v"(src));
    bool first = true;
    // per-atom constants of this lane's two channels (c0 + q, c0 + 32 + q): xh parts and vec / sqrt3
    float xa0 = 0.f, xb0 = 0.f, xc0 = 0.f, xa1 = 0.f, xb1 = 0.f, xc1 = 0.f;
    float wx0 = 0.f, wy0 = 0.f, wz0 = 0.f, wx1 = 0.f, wy1 = 0.f, wz1 = 0.f;
    // running sums over the atom's edges
    float da0 = 0.f, da1 = 0.f, db0 = 0.f, db1 = 0.f, dc0 = 0.f, dc1 = 0.f;
    float vx0 = 0.f, vy0 = 0.f, vz0 = 0.f, vx1 = 0.f, vy1 = 0.f, vz1 = 0.f;
    float ba0 = 0.f, ba1 = 0.f, bb0 = 0.f, bb1 = 0.f, bc0 = 0.f, bc1 = 0.f;   // !ST: column sums of the atom's drbfh rows

    while (have) {
        {
            const bool last = eb + 32 >= e1;
            float4 geoN = make_float4(0.f, 0.f, 0.f, 0.f); int srcN = 0; bool validN = false;
            if (!last) load_block(eb + 32, e1, geoN, srcN, validN);
            else if (haveN) load_block(e0N, e1N, geoN, srcN, validN);
            if (first) {
                const float* xr = pb.xh + (size_t)n * 3 * H + c0 + q;
                xa0 = xr[0]; xa1 = xr[32]; xb0 = xr[H]; xb1 = xr[H + 32]; xc0 = xr[2 * H]; xc1 = xr[2 * H + 32];
                if (!VZ) {
                    const float* vr = pb.vec + (size_t)n * 3 * H + c0 + q;
                    // (scaled by 1/sqrt3 behind the contraction loop: a use right here would wait for the loads)
                    wx0 = vr[0]; wx1 = vr[32]; wy0 = vr[H]; wy1 = vr[H + 32]; wz0 = vr[2 * H]; wz1 = vr[2 * H + 32];
                }
            }
            const float xs = geo.w * p.inv_cutoff;
            float xp = xs;
            if (p.env_pi == 5) { const float x2 = xs * xs; xp = x2 * x2 * xs; }
            else for (int i = 1; i < p.env_pi; ++i) xp *= xs;
            float env = 1.0f + p.env_a * xp + p.env_b * (xp * xs) + p.env_c * (xp * xs * xs);
            env = (xs < 1.0f && valid) ? env : 0.0f;
            __builtin_amdgcn_wave_barrier();
            if (hi == 0) {
                float* m = meta_w + q * 8;
                m[0] = __uint_as_float((unsigned int)(valid ? src : p.N) * row_bytes);
                m[1] = geo.x; m[2] = geo.y; m[3] = geo.z;
                // row of drbfh this edge row writes; padded rows write the spare row E (no branch around the stores)
                m[4] = __int_as_float(valid ? eb + q : pb.E);
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

#define ROW_OF(r) ((r & 3) + 8 * (r >> 2) + 4 * hi)
#define GATHER(r)                                                                               \
    const float* m##r = meta_w + ROW_OF(r) * 8;                                                 \
    const unsigned int o##r = __float_as_uint(m##r[0]);                                         \
    const float4 ga0##r = *reinterpret_cast<const float4*>(recS + (size_t)(o##r + qA));         \
    const float4 ga1##r = *reinterpret_cast<const float4*>(recS + (size_t)(o##r + qA) + 640);
#define CONSUME(r)                                                                              \
    {                                                                                           \
        const float ux = m##r[1], uy = m##r[2], uz = m##r[3];                                   \
        float S0 = 0.f, S1 = 0.f;                                                               \
        if (!VZ) {                                                                              \
            S0 = ga0##r.x * wx0 + ga0##r.y * wy0 + ga0##r.z * wz0;                              \
            S1 = ga1##r.x * wx1 + ga1##r.y * wy1 + ga1##r.z * wz1;                              \
        }                                                                                       \
        const float T0 = -(ga0##r.x * ux + ga0##r.y * uy + ga0##r.z * uz);                      \
        const float T1 = -(ga1##r.x * ux + ga1##r.y * uy + ga1##r.z * uz);                      \
        da0 += ga0##r.w * acc[0][r]; da1 += ga1##r.w * acc[1][r];                               \
        dc0 += T0 * acc[4][r]; dc1 += T1 * acc[5][r];                                           \
        if (!VZ) {                                                                              \
            db0 += S0 * acc[2][r]; db1 += S1 * acc[3][r];                                       \
            const float f0 = xb0 * acc[2][r], f1 = xb1 * acc[3][r];   /* x 1/sqrt3 at the end */  \
            vx0 += ga0##r.x * f0; vy0 += ga0##r.y * f0; vz0 += ga0##r.z * f0;                   \
            vx1 += ga1##r.x * f1; vy1 += ga1##r.y * f1; vz1 += ga1##r.z * f1;                   \
        }                                                                                       \
        if (!ST) {                                                                              \
            ba0 += ga0##r.w * xa0; ba1 += ga1##r.w * xa1; bc0 += T0 * xc0; bc1 += T1 * xc1;     \
            if (!VZ) { bb0 += S0 * xb0; bb1 += S1 * xb1; }                                      \
        } else if (MSGB_STORE == 0) {                                                           \
            char* d = dlane + (size_t)(unsigned int)__float_as_int(m##r[4]) * drow_bytes;       \
            *reinterpret_cast<float4*>(d) = make_float4(ga0##r.w * xa0, S0 * xb0, T0 * xc0, ga1##r.w * xa1); \
            *reinterpret_cast<float2*>(d + 16) = make_float2(S1 * xb1, T1 * xc1);               \
        } else {                                                                                \
            float* d = reinterpret_cast<float*>(dline + (size_t)(unsigned int)__float_as_int(m##r[4]) * drow_bytes); \
            d[0] = ga0##r.w * xa0; d[32] = S0 * xb0; d[64] = T0 * xc0;                          \
            d[96] = ga1##r.w * xa1; d[128] = S1 * xb1; d[160] = T1 * xc1;                       \
        }                                                                                       \
    }
            GATHER(0) GATHER(1) GATHER(2) GATHER(3)

            const float env256 = env * 256.0f;
            f32x16 acc[6];
            {
                half8 aone = {0, 0, 0, 0, 0, 0, 0, 0};
                if (hi == 0) { aone[0] = (_Float16)256.0f; aone[1] = (_Float16)256.0f; }
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    if (VZ && (b == 2 || b == 3)) continue;
                    const half8 bb = *reinterpret_cast<const half8*>(Wh + (size_t)(b * 32 + q) * MSG_LDK + 128);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aone, bb, zero16, 0, 0, 0);
                }
            }
            {
                const float xsq = xs * p.sarg;
                int k0 = klo;
                do {
                    // A fragment as in the forward kernel (message.hip): Gaussian recurrence, packed round-to-zero hi/lo split
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
                    const _Float16* wh = Wh + (size_t)q * MSG_LDK + k0 + 8 * hi;
                    const _Float16* wl = Wlo + (size_t)q * MSG_LDK + k0 + 8 * hi;
#pragma unroll
                    for (int b = 0; b < 6; ++b) {
                        if (VZ && (b == 2 || b == 3)) continue;   // vec == 0: S = 0 and dvec is not needed
                        const half8 bh = *reinterpret_cast<const half8*>(wh + b * 32 * MSG_LDK);
                        const half8 bl = *reinterpret_cast<const half8*>(wl + b * 32 * MSG_LDK);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
                    }
                    k0 += 16;
                } while (k0 < khi);
            }
            if (!VZ && first) {
                wx0 *= inv_sqrt3; wx1 *= inv_sqrt3; wy0 *= inv_sqrt3; wy1 *= inv_sqrt3; wz0 *= inv_sqrt3; wz1 *= inv_sqrt3;
            }
#if MSGB_AHEAD == 4
            GATHER(4) GATHER(5) GATHER(6) GATHER(7)
            CONSUME(0) CONSUME(1) CONSUME(2) CONSUME(3)
            GATHER(8) GATHER(9) GATHER(10) GATHER(11)
            CONSUME(4) CONSUME(5) CONSUME(6) CONSUME(7)
            GATHER(12) GATHER(13) GATHER(14) GATHER(15)
            CONSUME(8) CONSUME(9) CONSUME(10) CONSUME(11)
            CONSUME(12) CONSUME(13) CONSUME(14) CONSUME(15)
#else
            // (two rows ahead: the backward holds 24 per-atom constants and sums beside the 96 accumulators)
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
                // ---- finish this atom: scale, cross-half reduction, one write per row
                da0 *= out_scale; da1 *= out_scale; db0 *= out_scale; db1 *= out_scale; dc0 *= out_scale; dc1 *= out_scale;
                da0 += __shfl_xor(da0, 32); da1 += __shfl_xor(da1, 32);
                db0 += __shfl_xor(db0, 32); db1 += __shfl_xor(db1, 32);
                dc0 += __shfl_xor(dc0, 32); dc1 += __shfl_xor(dc1, 32);
                if (!VZ) {
                    const float vs = out_scale * inv_sqrt3;
                    vx0 *= vs; vx1 *= vs; vy0 *= vs; vy1 *= vs; vz0 *= vs; vz1 *= vs;
                    vx0 += __shfl_xor(vx0, 32); vx1 += __shfl_xor(vx1, 32);
                    vy0 += __shfl_xor(vy0, 32); vy1 += __shfl_xor(vy1, 32);
                    vz0 += __shfl_xor(vz0, 32); vz1 += __shfl_xor(vz1, 32);
                }
                const size_t ro = (size_t)n * 3 * H + c0 + q;
                if (hi == 0) {
                    pb.dxh[ro] = da0; pb.dxh[ro + 32] = da1;
                    pb.dxh[ro + H] = db0; pb.dxh[ro + H + 32] = db1;
                    pb.dxh[ro + 2 * H] = dc0; pb.dxh[ro + 2 * H + 32] = dc1;
                } else if (!VZ) {
                    pb.dvec[ro] = pb.gv1[ro] + vx0; pb.dvec[ro + 32] = pb.gv1[ro + 32] + vx1;
                    pb.dvec[ro + H] = pb.gv1[ro + H] + vy0; pb.dvec[ro + H + 32] = pb.gv1[ro + H + 32] + vy1;
                    pb.dvec[ro + 2 * H] = pb.gv1[ro + 2 * H] + vz0; pb.dvec[ro + 2 * H + 32] = pb.gv1[ro + 2 * H + 32] + vz1;
                }
                if (!ST) {
                    ba0 += __shfl_xor(ba0, 32); ba1 += __shfl_xor(ba1, 32);
                    bb0 += __shfl_xor(bb0, 32); bb1 += __shfl_xor(bb1, 32);
                    bc0 += __shfl_xor(bc0, 32); bc1 += __shfl_xor(bc1, 32);
                    float* br = pb.dbias_rows + ro + 32 * hi;
                    br[0] = hi ? ba1 : ba0; br[H] = hi ? bb1 : bb0; br[2 * H] = hi ? bc1 : bc0;
                    ba0 = ba1 = bb0 = bb1 = bc0 = bc1 = 0.f;
                }
                da0 = da1 = db0 = db1 = dc0 = dc1 = 0.f;
                vx0 = vx1 = vy0 = vy1 = vz0 = vz1 = 0.f;
                have = haveN;
                n = nN; eb = e0N; e1 = e1N;
                first = true;
                if (have) {
                    haveN = fetch_target(nN);
                    if (haveN) { e0N = nptr_c[nN]; e1N = nptr_c[nN + 1]; }
                }
            } else {
                eb += 32;
                first = false;
            }
            geo = geoN; src = srcN; valid = validN;
        }
    }
}

// Gradient records in the forward's record layout (message.hip adf_pack_records_kernel): per atom and group of 32 channels
// [32 x (g0, g1, g2, gx)], g = d(vec1) / sqrtH, gx = d(x1) / sqrt2; also dx = d(x1) / sqrt2, the residual path of x.
__global__ void adf_pack_grad_records_kernel(const float* __restrict__ gx1, const float* __restrict__ gv1,
                                             float* __restrict__ rec, float* __restrict__ dx, int N, int H, float is2,
                                             float ish) {
    const int ng = H / 32;
    const long long total = (long long)N * ng * 32;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int qq = (int)(t & 31);
        const int g = (int)((t >> 5) % ng);
        const long long n = t / (32LL * ng);
        const int c = 32 * g + qq;
        const float* vr = gv1 + (size_t)n * 3 * H;
        const float gx = gx1[(size_t)n * H + c] * is2;
        float* out = rec + ((size_t)n * ng + g) * 160;
        reinterpret_cast<float4*>(out)[qq] = make_float4(vr[c] * ish, vr[H + c] * ish, vr[2 * H + c] * ish, gx);
        if (dx) dx[(size_t)n * H + c] = gx;
    }
}

static size_t msgb_lds_bytes() {
    return (size_t)2 * MSG_COLS * MSG_LDK * 2 + sizeof(float) * (MSG_COLS + 128 + MSG_WAVES * 32 * 8) + 16;
}

extern "C" int32_t adf_op_message_bwd_fused_supported(adf_painn_t h) {
    return h && h->weights_set && !h->msg_f32 && h->rbf_uniform && h->hp.num_rbf <= 128 && (h->hp.num_rbf % 8) == 0 ? 1 : 0;
}

// column c' of the kernel-ordered drbfh row <-> column perm[c'] of the [a | b | c] layout of rbf_proj's output
extern "C" int32_t adf_op_message_bwd_perm(adf_painn_t h, int32_t* perm_host, int32_t n) {
    if (!h || !perm_host || n != 3 * h->hp.hidden_channels) { adf_set_error("message_bwd_perm: bad argument"); return ADF_EINVAL; }
    const int H = h->hp.hidden_channels;
    for (int c = 0; c < n; ++c) {
        const int slice = c / 192, r = c % 192, qq = MSGB_STORE == 0 ? r / 6 : r % 32, v = MSGB_STORE == 0 ? r % 6 : r / 32;
        perm_host[c] = (v % 3) * H + slice * ADF_SLICE_CH + (v / 3) * 32 + qq;
    }
    return ADF_OK;
}

extern "C" int32_t adf_op_message_bwd_fused(adf_painn_t h, int32_t layer, const float* xh, const float* vec, const float* gx1,
                                            const float* gv1, float* dxh, float* drbfh_lane_order, int64_t num_edges,
                                            float* dvec, float* dx, int32_t vec_is_zero, float* dbias_rows, void* stream) {
    if (!h || h->lastN <= 0 || layer < 0 || layer >= h->hp.num_layers || !xh || !gx1 || !gv1 || !dxh ||
        (!drbfh_lane_order && !dbias_rows) || !dx || (!vec_is_zero && (!vec || !dvec))) {
        adf_set_error("message_bwd_fused: bad argument or no graph");
        return ADF_EINVAL;
    }
    if (!adf_op_message_bwd_fused_supported(h)) {
        adf_set_error("message_bwd_fused: needs the f16x3 arithmetic and equally spaced Gaussian centres");
        return ADF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const int N = (int)h->lastN, H = h->hp.hidden_channels, R = h->hp.num_rbf;
    if (num_edges <= 0) { adf_set_error("message_bwd_fused: num_edges must be the graph's edge count"); return ADF_EINVAL; }
    const int E = (int)num_edges;
    if ((unsigned long long)(N + 1) * 5ull * H * sizeof(float) >= (1ull << 32)) {
        adf_set_error("message kernel uses 32-bit byte offsets into the node tables: N=%d is too large, split the batch", N);
        return ADF_EOOM;
    }
    static bool attr_set = false;  // per process and device: training runs on one device per process
    if (!attr_set) {
        const void* kernels[4] = {reinterpret_cast<const void*>(adf_message_bwd_kernel<false, false>),
                                  reinterpret_cast<const void*>(adf_message_bwd_kernel<false, true>),
                                  reinterpret_cast<const void*>(adf_message_bwd_kernel<true, false>),
                                  reinterpret_cast<const void*>(adf_message_bwd_kernel<true, true>)};
        for (const void* k : kernels)
            ADF_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)msgb_lds_bytes()));
        attr_set = true;
    }
    const size_t row = (size_t)(H / 32) * 160;
    ADF_HIP_CHECK(hipMemsetAsync(h->rec + (size_t)N * row, 0, sizeof(float) * row, s));   // padded edge rows gather record N
    {
        long long blocks = ((long long)N * H + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(adf_pack_grad_records_kernel, dim3((unsigned)blocks), dim3(256), 0, s, gx1, gv1, h->rec, dx, N, H,
                           0.70710678118654752f, 1.0f / sqrtf((float)H));
    }
    MsgBwdParams pb;
    MsgParams& p = pb.m;
    memset(&pb, 0, sizeof(pb));
    p.rec = h->rec; p.nptr = h->nptr; p.e_src = h->e_src; p.e_geom = h->e_geom;
    p.nslices = H / ADF_SLICE_CH;
    p.wpack16 = reinterpret_cast<const _Float16*>(h->rbf_pack16) + (size_t)layer * 2 * p.nslices * R * MSG_COLS;
    p.bpack = h->rbf_bias_pack16 + (size_t)layer * p.nslices * MSG_COLS;
    p.inv_scale = h->rbf_scales + layer;
    p.mu = h->rbf_offset;
    p.N = N; p.H = H; p.R = R; p.items = N;
    p.G = (N + ADF_GROUP_NODES - 1) / ADF_GROUP_NODES;
    p.inv_cutoff = 1.0f / h->hp.cutoff;
    const double step = 1.0 / (R - 1);
    p.sarg = (float)sqrt(0.5 / (step * step) * 1.4426950408889634);
    const double pe = (double)h->hp.envelope_exponent;
    p.env_pi = h->hp.envelope_exponent;
    p.env_a = (float)(-(pe + 1) * (pe + 2) / 2);
    p.env_b = (float)(pe * (pe + 2));
    p.env_c = (float)(-pe * (pe + 1) / 2);
    {
        const double d = sqrt(0.5 / (step * step) * 1.4426950408889634) * step;
        p.dmu2 = (float)(2.0 * d); p.dmusq = (float)(d * d); p.cstep = (float)exp2(-2.0 * d * d);
    }
    pb.E = E; pb.xh = xh; pb.vec = vec; pb.gv1 = gv1; pb.dxh = dxh; pb.dvec = dvec; pb.drbfh = drbfh_lane_order;
    int workers = h->num_cus / p.nslices;
    if (workers < 1) workers = 1;
    if (workers > p.G) workers = p.G;
    dim3 grid((unsigned)(workers * p.nslices));
    pb.dbias_rows = dbias_rows;
    if (drbfh_lane_order) {
        if (vec_is_zero) hipLaunchKernelGGL((adf_message_bwd_kernel<true, true>), grid, dim3(MSG_THREADS), msgb_lds_bytes(), s, pb);
        else hipLaunchKernelGGL((adf_message_bwd_kernel<false, true>), grid, dim3(MSG_THREADS), msgb_lds_bytes(), s, pb);
    } else {
        if (vec_is_zero) hipLaunchKernelGGL((adf_message_bwd_kernel<true, false>), grid, dim3(MSG_THREADS), msgb_lds_bytes(), s, pb);
        else hipLaunchKernelGGL((adf_message_bwd_kernel<false, false>), grid, dim3(MSG_THREADS), msgb_lds_bytes(), s, pb);
    }
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
