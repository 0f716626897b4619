// Weight (and bias) gradient of a message block's rbf_proj for the training step (BASELINE config 5) WITHOUT
// the per-edge gradient d(rbfh) [E, 3H] in memory.
//
// Reference: torch.autograd through models/painn/painn_denoising.py:530-567 - rbfh = rbf_proj(edge_rbf) multiplies the
// gathered xh rows, so dW = d(rbfh)^T edge_rbf is a product contracted over the EDGES (K = 2.6 M at 256 graphs) and d(rbfh)
// is a 16 GB tensor per layer.  Until round 4 message_bwd.hip wrote it (the kernel was bound by those stores) and
// tr_wgrad_bf16x6_kernel (train.hip) read it back: 36 of the ~70 GB a layer's backward moved through HBM.
//
// d(rbfh) does not depend on rbfh (message_bwd.hip's header): for edge row e of atom j's CSR segment, neighbour i = e_src[e],
//   d(rbfh)[e] = (gx[i] xa[j],  (g[i] . vec[j] / sqrt3) xb[j],  -(g[i] . u_e) xc[j])        per channel
// with (g, gx) the packed gradient records of i (adf_pack_grad_records_kernel) and (xa, xb, xc) = xh[j].  This kernel
// forms those values while it stages a 32-edge chunk - the same 16-B record gathers as the forward message kernel, served
// by the XCD's L2 (slice = workgroup index mod 8: one XCD only touches its own 64 channels of the record table) - and runs
// the product on the f16-rate matrix cores with all-zero 32-column blocks of the radial basis skipped.  Arithmetic (RW_F16, the
// default): both operands as TWO fp16 terms (hi + lo, 22 bits), three products - the arithmetic of the step's other
// products; the contraction runs along the edges, so a power-of-two lift per COLUMN of d(rbfh) factors out of the sum, and
// the columns' maxima are bounded from the node tables (rw_colmax_kernel) instead of from the tensor that is never written;
// the basis is scaled by 2^14.  RW_F16=0: tr_wgrad_bf16x6_kernel's three bf16 terms, six products, no lifts (5.3 ms per
// launch at 256 graphs against 3.6).  A workgroup owns the 192 gradient rows of its slice (a, b, c parts of 64 channels) x
// all R <= 128 basis functions for its share of the edge rows; the per-workgroup partial results are summed in a fixed
// order (run-to-run reproducible), directly into the reference's row order.  The values staged are computed by the same
// expressions as message_bwd.hip's stores were (the 1/sqrt3 of the b part multiplies xb instead of the three vec components).
// Entry points: adf_op_edge_owner, adf_op_rbf_image (once per step), adf_op_rbf_wgrad_fused (per layer, after
// adf_op_message_bwd_fused with drbfh = NULL).  Experiment log: profiles/NOTES.md (round 5, training).
#include <stdlib.h>
#include <string.h>

#include "message.h"
#if defined(RW_PROF) && RW_PROF
#include <stdio.h>
#include <vector>
#endif

typedef __fp16 rw_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __bf16 rw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rw_bf16x4 __attribute__((ext_vector_type(4)));

struct RbfWgradParams {
    const float* rec;        // gradient records [(N+1)][H/32][160]: [32 x (g0, g1, g2, gx)] + [32 unused]; row N zero
    const float* xh;         // [N, 3H]
    const float* vec;        // [N, 3, H] or null (first layer)
    const unsigned char* img;   // radial basis as the product's RW_NT terms in the kernel's LDS layout (rw_basis_image_kernel), + masks
    const int32_t* e_src;    // neighbour i of edge row e
    const int32_t* owner;    // atom j whose CSR segment holds row e
    const float4* e_geom;    // (unit vector, distance) of row e
    float* part;             // [splits][3H x R] partial gradients, reference row order
    const float* lift;          // RW_F16: [3][H] power-of-two lift of a d(rbfh) column (part b: x 1/sqrt3); null otherwise
    const float* unlift;        // RW_F16: [3][H] 1 / (lift x 2^14): what an accumulator row is multiplied by
    unsigned long long* prof;   // RW_PROF builds only: [workgroup][8] cycle sums
    int E, N, H, R, workers, nslices;   // workers: workgroups per slice (= edge-range partial results)
};

// LDS images, one buffer per chunk in flight.  d(rbfh) ROW-major as its producer threads hold it (one edge row x 8 channels):
// [term][32 edge rows][512 B] bf16 with the 16-byte pieces XOR-swizzled (tr_wgrad_bf16x6_kernel's image; 192 of the row's 256
// columns are used: the swizzle spans 16 pieces), read with the transposing ds_read_b64_tr_b16.  The radial basis
// COLUMN-major - [term][column][32 edge rows] bf16, 80 bytes per column (64 + 16 pad: the 16-byte accesses of 16 consecutive
// columns fall into 16 different bank groups) - which is the K-contiguous layout the 32x32x16 instruction wants: a fragment
// is one ds_read_b128; it arrives in that layout from rw_basis_image_kernel.
#define RW_CROW 512
#define RW_COLB 80
#ifndef RW_F16
#define RW_F16 1   // 1: both operands as TWO fp16 terms (hi + lo), three products, d(rbfh) lifted per column by a power of two, the
                   //    basis by 2^14 (the arithmetic of every other product of the step); 0: three bf16 terms, six products, no lifts
#endif
#define RW_NT (RW_F16 ? 2 : 3)                  // terms per operand
#define RW_CIMG (32 * RW_CROW)                  // one term of d(rbfh)
#define RW_AIMG (128 * RW_COLB)                 // one term of the radial basis
#define RW_IMG (RW_NT * (RW_CIMG + RW_AIMG))
#define RW_BUF (RW_IMG + 64)                    // + the chunk's mask of non-zero 32-column basis blocks
#define RW_IMG_CHUNK (RW_NT * 4 * 128 * 16)     // global image of the basis: [chunk][term][row octet][column][8 rows] 16-bit
#define RW_THREADS 512
#define RW_MINI 8      // chunks per mini-range of the sweep
#ifndef RW_PROF
#define RW_PROF 0  // 1: cycle counters per phase (lane 0 of waves 0 and 4), printed by the launcher
#endif
#if RW_PROF
#define RW_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define RW_ACC(k, a, b) pf[k] += (b) - (a)
#else
#define RW_T(var)
#define RW_ACC(k, a, b)
#endif
#ifndef RW_ABL
#define RW_ABL 0   // timing experiments: 1 no gathers, 2 no products, 4 no d(rbfh) conversion / stores, 8 no basis copy, 128 owner rows of atom 0, 256 records of atom 0, 512 the 8 lanes of a row read the same 16 bytes of the owner rows (wrong results); 64 no zero-block skipping (right results)
#endif

// v[0..N-1] -> RW_NT 16-bit terms, packed pairs o[term][pair].
//  RW_F16: hi = round-to-zero fp16 (one packed conversion per pair), lo = fp16(v - hi) (mixed-precision subtraction): 2 vector
//          instructions per value, 22 significant bits while v stays inside fp16's normal range (the lifts see to that);
//  else:   three bf16 terms t0 + t1 + t2 = v exactly (24 bits), round-to-nearest; the packed conversion of a pair is reused for
//          the residuals (low half << 16, high half masked): 5.5 vector instructions per value.
typedef float rw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rw_bf16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 rw_h2 __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void rw_split(float* v, unsigned int (*o)[N / 2]) {
#if RW_F16
#pragma unroll
    for (int e = 0; e < N; e += 2) {
        union { rw_h2 h; unsigned int u; } hi, lo;
        hi.h = __builtin_amdgcn_cvt_pkrtz(v[e], v[e + 1]);
        lo.h = __builtin_amdgcn_cvt_pkrtz(v[e] - (float)hi.h[0], v[e + 1] - (float)hi.h[1]);
        o[0][e >> 1] = hi.u;
        o[1][e >> 1] = lo.u;
    }
#else
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < N; e += 2) {
            union { rw_bf16x2 b; unsigned int u; } pk;
            pk.b = __builtin_convertvector((rw_f32x2){v[e], v[e + 1]}, rw_bf16x2);
            o[k][e >> 1] = pk.u;
            if (k < 2) {
                v[e] -= __uint_as_float(pk.u << 16);
                v[e + 1] -= __uint_as_float(pk.u & 0xffff0000u);
            }
        }
#endif
}

struct rw_yes { static const bool value = true; };
struct rw_no { static const bool value = false; };
typedef const __attribute__((address_space(4))) int32_t* rw_cint_ptr;
typedef const __attribute__((address_space(4))) float* rw_cflt_ptr;

__device__ __forceinline__ unsigned int rw_swz(int row) { return (unsigned int)(((row & 3) << 2) | ((row >> 2) & 3)); }

// what a producer thread keeps in flight for one chunk: one edge row x 8 channels
struct rw_gather_set {
    float4 g[8];                                   // the neighbour's packed gradients (g0, g1, g2, gx) per channel
    float4 xa[2], xb[2], xc[2], wx[2], wy[2], wz[2];   // the owner's xh parts and vec, 2 x 4 channels
    float ux, uy, uz;
};
typedef unsigned int rw_u32x4 __attribute__((ext_vector_type(4)));
struct rw_basis_set { rw_u32x4 q[2][RW_NT]; int mask; };

// Waves 4-7 PRODUCE: producer thread (edge row sr, 8 channels) gathers the neighbour's packed gradients and the owner's xh / vec
// rows, forms the three parts of d(rbfh), lifts and splits them (RW_NT terms) and writes them into LDS buffer h & 1; it also
// copies the chunk's radial-basis image (already split, rw_basis_image_kernel) into that buffer.  Waves 0-3 CONSUME the other
// buffer: 96 accumulator registers each, nothing else to do.  One barrier per chunk.  Every request is TWO chunks ahead of
// its use in two alternating register sets (the producers hold no accumulators), the edge rows' (neighbour, owner, unit
// vector) one chunk before that.
//
// What bounds it (profiles/r05_rbf_wgrad_*): on a SIMD the producer's vector instructions (4 cycles each) and the consumer's
// MFMAs (32 cycles each) ADD - per 32-edge chunk 307 vector instructions + 28 MFMAs (of 36: all-zero basis blocks are
// skipped) = 2 100 cycles - and the CU's vector-memory path moves ~100 KB per chunk (32 KB of records, 48 KB of owner rows,
// 16 KB of basis image: ~1 600 cycles at 64 B per clock), which is why the requests are spread over the staging.  3 200
// cycles per chunk measured.
//
// Measured on the way here (256 graphs, per launch; the kernel that read d(rbfh) from memory took 5.2 ms and its producer,
// message_bwd, 3.0 ms longer): one role per wave and requests at the top of the chunk 6.7 ms (two dependent round trips);
// requests one chunk ahead beside the accumulators: 268 B of scratch per lane, whose reloads drain the vector-memory counter
// and with it the prefetch; producer / consumer waves, requests one chunk ahead 8.5 ms (a request issued at the end of a
// chunk has only the barrier wait to land); two chunks ahead 6.5 ms; wave = row octet and lane = channel with wave-uniform
// rows (coalesced gathers, column-major d(rbfh), owner rows once per owner: scratch/experiments/rbf_wgrad_octet.hip)
// 8.1-9.2 ms - the scalar bookkeeping and lane<->scalar moves cost more vector-issue slots than the coalescing saved; back to
// thread = edge row x 8 channels with the basis image copied, not converted, and a 5.5-instruction bf16 split 5.3 ms; two
// fp16 terms with column lifts 4.3 ms; each half's requests right behind the staging of the same registers 3.6 ms (the
// basis copy in front of the staging instead of behind it: 4.1 ms; records and owner rows requested apart: 3.9 ms).
template <bool VZ>
__global__ __launch_bounds__(RW_THREADS, 1) void rw_rbf_wgrad_kernel(RbfWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rw_lds[];
    const int slice = blockIdx.x % p.nslices, split = blockIdx.x / p.nslices;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, R = p.R;
    const int c0 = slice * ADF_SLICE_CH;
    // The workgroups of a slice (one XCD) sweep the edge rows TOGETHER: worker w takes the mini-ranges w, w + W, w + 2 W, ... of
    // RW_MINI chunks each, so at any time the XCD gathers from ~W * RW_MINI * 32 consecutive edge rows (one or two systems).
    const int total_chunks = (p.E + 31) / 32, minis = (total_chunks + RW_MINI - 1) / RW_MINI;
    const int my_minis = split < minis ? (minis - split + p.workers - 1) / p.workers : 0;
    const int nchunks = my_minis * RW_MINI;
    auto row0 = [&](int t) { return 32 * (((t / RW_MINI) * p.workers + split) * RW_MINI + (t % RW_MINI)); };   // local chunk t -> first edge row

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pt = tid - 256, pw = wave - 4;
        // thread = (edge row sr, channels c0 + 4 sg .. + 3 and c0 + 32 + 4 sg .. + 3): the 8 lanes of a row read 512 contiguous
        // bytes of each 32-channel record group and one 128-byte line of each xh / vec row per instruction
        const int sr = pt >> 3, sg = pt & 7;
        // uniform bases + 32-bit byte offsets (global_load ... v_off, s[base]): the node tables are < 4 GB (checked by the launcher)
        const unsigned int rec_row_b = (unsigned int)(H / 32) * 640u, row_b = 12u * (unsigned int)H;
        const char* rec_b = reinterpret_cast<const char*>(p.rec) + (size_t)(2 * slice) * 640;
        const unsigned int rec_lane = (unsigned int)sg * 64u;
        const char* xh_b = reinterpret_cast<const char*>(p.xh) + (size_t)c0 * 4;
        const char* vec_b = VZ ? nullptr : reinterpret_cast<const char*>(p.vec) + (size_t)c0 * 4;
        const unsigned int ch_lane = (unsigned int)sg * 16u;
        const unsigned int Hb = 4u * (unsigned int)H;
        const float inv_sqrt3 = 0.57735026918962576f;
        float la[8], lb[8], lc[8];   // column lifts of this thread's 8 channels (part b: x 1/sqrt3)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ch = c0 + 32 * (c >> 2) + 4 * sg + (c & 3);
            la[c] = RW_F16 ? p.lift[ch] : 1.0f;
            lb[c] = RW_F16 ? p.lift[H + ch] : inv_sqrt3;
            lc[c] = RW_F16 ? p.lift[2 * H + ch] : 1.0f;
        }
        int srcN = p.N, ownN = 0;
        float4 geoN = make_float4(0.f, 0.f, 0.f, 0.f);
        auto load_meta = [&](int t) {   // rows past the end gather the zero record (row N) and a valid owner
            const int e = row0(t) + sr;
            const unsigned int ec = (unsigned int)min(e, p.E - 1);
            const int sv = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.e_src) + 4u * ec);
            srcN = e < p.E ? sv : p.N;
            ownN = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p.owner) + 4u * ec);
            geoN = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.e_geom) + 16u * ec);
        };
        // gathers of the chunk whose (neighbour, owner, unit vector) are in srcN / ownN / geoN, one half of the thread's channels at
        // a time: each half is requested right behind the staging of the same registers' previous contents, so the 26 loads of
        // a chunk are spread over the staging instead of arriving at the texture addresser in one burst
        auto request_half = [&](rw_gather_set& S, int hh) {
            if (hh == 0) { S.ux = geoN.x; S.uy = geoN.y; S.uz = geoN.z; }
            if (RW_ABL & 1) return;
            const unsigned int ro = (unsigned int)((RW_ABL & 256) ? 0 : srcN) * rec_row_b + rec_lane;   // (256: every row gathers atom 0's record)
#pragma unroll
            for (int k = 0; k < 4; ++k)   // group hh (640 bytes apart), channel 4 sg + k
                S.g[4 * hh + k] = *reinterpret_cast<const float4*>(rec_b + (ro + (unsigned int)(hh * 640 + k * 16)));
            const unsigned int oo = (unsigned int)((RW_ABL & 128) ? 0 : ownN) * row_b + ((RW_ABL & 512) ? 0u : ch_lane);   // (128: every row reads atom 0's rows; 512: the 8 lanes of a row read the same 16 bytes)
            S.xa[hh] = *reinterpret_cast<const float4*>(xh_b + (oo + 128u * hh));
            S.xc[hh] = *reinterpret_cast<const float4*>(xh_b + (oo + 2u * Hb + 128u * hh));
            if (!VZ) {
                S.xb[hh] = *reinterpret_cast<const float4*>(xh_b + (oo + Hb + 128u * hh));
                S.wx[hh] = *reinterpret_cast<const float4*>(vec_b + (oo + 128u * hh));
                S.wy[hh] = *reinterpret_cast<const float4*>(vec_b + (oo + Hb + 128u * hh));
                S.wz[hh] = *reinterpret_cast<const float4*>(vec_b + (oo + 2u * Hb + 128u * hh));
            }
        };
        // d(rbfh) of this thread's row and 4 of its 8 channels -> the terms' images in buffer b (u: the row's unit vector, read
        // before the first half's request overwrites it)
        auto stage_half = [&](rw_gather_set& S, int b, int hh, float ux, float uy, float uz) {
            if (RW_ABL & 4) { if (S.g[0].x == 123.f && S.xa[0].x == 1.f) rw_lds[tid] = 1; return; }
            unsigned char* imgC = rw_lds + (size_t)b * RW_BUF;
            {
                float va[4], vb[4], vc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 gg = S.g[4 * hh + k];
                    const float xac = k == 0 ? S.xa[hh].x : k == 1 ? S.xa[hh].y : k == 2 ? S.xa[hh].z : S.xa[hh].w;
                    const float xcc = k == 0 ? S.xc[hh].x : k == 1 ? S.xc[hh].y : k == 2 ? S.xc[hh].z : S.xc[hh].w;
                    va[k] = gg.w * (RW_F16 ? xac * la[4 * hh + k] : xac);
                    const float T = -(gg.x * ux + gg.y * uy + gg.z * uz);
                    vc[k] = T * (RW_F16 ? xcc * lc[4 * hh + k] : xcc);
                    vb[k] = 0.f;
                    if (!VZ) {   // (g . vec / sqrt3) xb as (g . vec)(xb / sqrt3): one scaling per channel instead of three
                        const float xbc = (k == 0 ? S.xb[hh].x : k == 1 ? S.xb[hh].y : k == 2 ? S.xb[hh].z : S.xb[hh].w) * lb[4 * hh + k];
                        const float wxc = k == 0 ? S.wx[hh].x : k == 1 ? S.wx[hh].y : k == 2 ? S.wx[hh].z : S.wx[hh].w;
                        const float wyc = k == 0 ? S.wy[hh].x : k == 1 ? S.wy[hh].y : k == 2 ? S.wy[hh].z : S.wy[hh].w;
                        const float wzc = k == 0 ? S.wz[hh].x : k == 1 ? S.wz[hh].y : k == 2 ? S.wz[hh].z : S.wz[hh].w;
                        const float Sd = gg.x * wxc + gg.y * wyc + gg.z * wzc;
                        vb[k] = Sd * xbc;
                    }
                }
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    if (VZ && part == 1) continue;   // the b columns are never read on the first layer
                    float* v = part == 0 ? va : part == 1 ? vb : vc;
                    // image column 64 part + 32 hh + 4 sg: 16-byte piece 8 part + 4 hh + sg / 2, half sg % 2
                    const unsigned int off = RW_CROW * sr + 16u * ((unsigned int)(8 * part + 4 * hh + (sg >> 1)) ^ rw_swz(sr)) + 8u * (sg & 1);
                    unsigned int tt[RW_NT][2];
                    rw_split<4>(v, tt);
#pragma unroll
                    for (int t = 0; t < RW_NT; ++t) *reinterpret_cast<uint2*>(imgC + (size_t)t * RW_CIMG + off) = make_uint2(tt[t][0], tt[t][1]);
                }
            }
        };
        // radial basis: already split into its terms, in the LDS layout (rw_basis_image_kernel, once per step): a producer thread moves
        // two (column, row octet) units of 16 bytes per term (the consumers' registers are the accumulators)
        const int bcol = 64 * (pw & 1) + lane, boct = pw >> 1;   // unit k: column bcol, octet boct + 2 k
        const rw_cint_ptr mask_c = (rw_cint_ptr) reinterpret_cast<const int32_t*>(p.img + (size_t)total_chunks * RW_IMG_CHUNK);
        auto request_basis = [&](rw_basis_set& B, int t) {
            const int gc = min(row0(t) >> 5, total_chunks - 1);   // (past the end: any chunk - its d(rbfh) rows are all zero)
            const unsigned char* src = p.img + (size_t)gc * RW_IMG_CHUNK + (size_t)bcol * 16;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int tm = 0; tm < RW_NT; ++tm)
                    B.q[k][tm] = *reinterpret_cast<const rw_u32x4*>(src + (size_t)tm * 8192 + (size_t)(boct + 2 * k) * 2048);
            B.mask = mask_c[gc];
        };
        auto stage_basis = [&](rw_basis_set& B, int b) {
            if (RW_ABL & 8) { if (B.q[0][0][0] == 123u) rw_lds[tid] = 1; return; }
            unsigned char* imgA = rw_lds + (size_t)b * RW_BUF + RW_NT * RW_CIMG;
            unsigned int* nzf = reinterpret_cast<unsigned int*>(rw_lds + (size_t)b * RW_BUF + RW_IMG);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int tm = 0; tm < RW_NT; ++tm)
                    *reinterpret_cast<rw_u32x4*>(imgA + (size_t)tm * RW_AIMG + (size_t)bcol * RW_COLB + 16 * (boct + 2 * k)) = B.q[k][tm];
            if (tid == 256) nzf[0] = (unsigned int)B.mask;
        };
        // Half-step h: stage chunk h - 2 from set h & 1 into buffer h & 1, request chunk h into that set, read the edge rows of
        // chunk h + 1; a barrier behind every staged chunk.  The loop starts with nothing in flight (no peeled prologue: with
        // one, the compiler's wait-count state at the loop header was the conservative merge of two different histories and
        // one of the two stages waited for every outstanding load).  No branch around a load anywhere in here (the compiler
        // drains the vector-memory counter at the join): past-the-end rows read clamped addresses.
        rw_gather_set S0, S1;
        rw_basis_set B0, B1;
        load_meta(0);
        unsigned long long pf[4] = {0, 0, 0, 0};
        (void)pf;
        for (int h = 0;; h += 2) {
            if (h - 2 >= nchunks) break;
            RW_T(t0);
            {
                const float ux = S0.ux, uy = S0.uy, uz = S0.uz;
                if (h >= 2) stage_half(S0, 0, 0, ux, uy, uz);
                request_half(S0, 0);
                if (h >= 2) stage_half(S0, 0, 1, ux, uy, uz);
                request_half(S0, 1);
                if (h >= 2) stage_basis(B0, 0);
            }
            RW_T(t1);
            request_basis(B0, h);
            load_meta(h + 1);
            RW_T(t2);
            if (h >= 2) __syncthreads();
            RW_T(t3);
            RW_ACC(0, t0, t1); RW_ACC(1, t1, t2); RW_ACC(2, t2, t3);
            if (h - 1 >= nchunks) break;
            {
                const float ux = S1.ux, uy = S1.uy, uz = S1.uz;
                if (h >= 2) stage_half(S1, 1, 0, ux, uy, uz);
                request_half(S1, 0);
                if (h >= 2) stage_half(S1, 1, 1, ux, uy, uz);
                request_half(S1, 1);
                if (h >= 2) stage_basis(B1, 1);
            }
            RW_T(t4);
            request_basis(B1, h + 1);
            load_meta(h + 2);
            RW_T(t5);
            if (h >= 2) __syncthreads();
            RW_T(t6);
            RW_ACC(0, t3, t4); RW_ACC(1, t4, t5); RW_ACC(2, t5, t6);
        }
        __syncthreads();   // the consumers' barrier behind the last chunk's products
#if RW_PROF
        if (tid == 256 && p.prof) for (int k = 0; k < 3; ++k) p.prof[(size_t)blockIdx.x * 8 + 4 + k] = pf[k];
#endif
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumers
    const int kb = wave & 1, cgp = wave >> 1;
    const int fm = lane & 31, fkg = lane >> 5;   // basis fragments: column fm of a 32-column block, k half fkg
    // d(rbfh) fragments, transposed-read addressing (tr_wgrad_bf16x6_kernel): 16-lane group g16, lane 4 tq + tp of the group
    const int g16 = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int kg = g16 >> 1, cgrp = g16 & 1;
    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // the products of one (32-column block of d(rbfh)) x (32 basis functions) pair over one 16-row step: lo x hi, hi x lo,
    // hi x hi (fp16 terms) or the six products of order <= 2^-16 of the three-term bf16 split
    auto multiply = [&](f32x16& c, const rw_u32x4* a, const rw_u32x4* bq) __attribute__((always_inline)) {
#if RW_F16
        typedef _Float16 rw_half8 __attribute__((ext_vector_type(8)));
        const rw_half8 ah = __builtin_bit_cast(rw_half8, a[0]), al = __builtin_bit_cast(rw_half8, a[1]);
        const rw_half8 bh = __builtin_bit_cast(rw_half8, bq[0]), bl = __builtin_bit_cast(rw_half8, bq[1]);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
#else
        rw_bf16x8 x[3], y[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) { x[t] = __builtin_bit_cast(rw_bf16x8, a[t]); y[t] = __builtin_bit_cast(rw_bf16x8, bq[t]); }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[0], c, 0, 0, 0);
#endif
    };
    // D0 / D1: this wave's basis blocks kb / kb + 2 hold a non-zero in this chunk (straight-line code per case, so that the
    // fragment reads of the next pair are issued behind the products of the current one)
    auto products_case = [&](const unsigned char* imgC, const unsigned char* imgA, auto D0, auto D1) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {   // (unrolled: as a loop the compiler copied all 96 accumulator registers around every step)
            const unsigned int koff = 32u * ks + 16u * fkg;
            rw_u32x4 b0[RW_NT], b1[RW_NT];
#pragma unroll
            for (int t = 0; t < RW_NT; ++t) {
                if (D0.value) b0[t] = *reinterpret_cast<const rw_u32x4*>(imgA + (size_t)t * RW_AIMG + (size_t)(32 * kb + fm) * RW_COLB + koff);
                if (D1.value) b1[t] = *reinterpret_cast<const rw_u32x4*>(imgA + (size_t)t * RW_AIMG + (size_t)(32 * (kb + 2) + fm) * RW_COLB + koff);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int bi = 3 * cgp + i;           // 32-column block of the slice's 192: part = bi / 2
                if (VZ && (bi >> 1) == 1) continue;
                rw_u32x4 a[RW_NT];
#pragma unroll
                for (int t = 0; t < RW_NT; ++t) {   // rows 16 ks + 8 kg + (0..7) of the 16 columns that hold this lane's column
                    const int col = 32 * bi + 16 * cgrp + 4 * tp;
                    union { rw_fp16x4 h[2]; rw_u32x4 b; } u;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int row = 16 * ks + 8 * kg + 4 * hh + tq;
                        const unsigned int off = RW_CROW * row + 16u * ((unsigned int)(col >> 3) ^ rw_swz(row)) + 8u * ((col >> 2) & 1);
                        u.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                            (__attribute__((address_space(3))) rw_fp16x4*)(imgC + (size_t)t * RW_CIMG + off));
                    }
                    a[t] = u.b;
                }
                if (D0.value) multiply(acc[i][0], a, b0);
                if (D1.value) multiply(acc[i][1], a, b1);
            }
        }
    };
    auto products = [&](int b) __attribute__((always_inline)) {
        const unsigned char* imgC = rw_lds + (size_t)b * RW_BUF;
        const unsigned char* imgA = imgC + RW_NT * RW_CIMG;
        const unsigned int* nzf = reinterpret_cast<const unsigned int*>(imgC + RW_IMG);
        const unsigned int nzmask = nzf[0];                                   // bit k: basis block k of this chunk holds a non-zero
        const bool do0 = ((nzmask >> kb) & 1u) != 0u, do1 = ((nzmask >> (kb + 2)) & 1u) != 0u;   // wave-uniform
        if (RW_ABL & 2) return;
        if (RW_ABL & 64) { products_case(imgC, imgA, rw_yes(), rw_yes()); return; }   // no skipping
        if (do0 && do1) products_case(imgC, imgA, rw_yes(), rw_yes());
        else if (do0) products_case(imgC, imgA, rw_yes(), rw_no());
        else if (do1) products_case(imgC, imgA, rw_no(), rw_yes());
    };
    // Half-step h: the products of chunk h - 3 from buffer (h - 3) & 1; a barrier per half-step from h = 2 on (n + 1 in all,
    // as the producers).
    unsigned long long pf[4] = {0, 0, 0, 0};
    (void)pf;
    for (int h = 0;; h += 2) {
        if (h - 3 >= nchunks) break;
        RW_T(t0);
        RW_T(t1);
        if (h >= 3) products(1);          // chunk h - 3 (odd): buffer 1
        RW_T(t2);
        if (h >= 2) __syncthreads();
        RW_T(t3);
        RW_ACC(0, t0, t1); RW_ACC(1, t1, t2); RW_ACC(2, t2, t3);
        if (h - 2 >= nchunks) break;
        RW_T(t4);
        if (h >= 2) products(0);          // chunk h - 2 (even): buffer 0
        RW_T(t5);
        if (h >= 2) __syncthreads();
        RW_T(t6);
        RW_ACC(0, t3, t4); RW_ACC(1, t4, t5); RW_ACC(2, t5, t6);
    }
#if RW_PROF
    if (tid == 0 && p.prof) { for (int k = 0; k < 3; ++k) p.prof[(size_t)blockIdx.x * 8 + k] = pf[k]; p.prof[(size_t)blockIdx.x * 8 + 3] = (unsigned long long)nchunks; }
#endif
    // ---- partial gradient of this edge range, written in the reference's row order ([a | b | c] x H)
    float* out = p.part + (size_t)split * 3 * H * R;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int bi = 3 * cgp + i;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = 32 * (kb + 2 * j) + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cc = 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int row = (cc >> 6) * H + c0 + (cc & 63);
                if (col < R) out[(size_t)row * R + col] = RW_F16 ? acc[i][j][r] * p.unlift[row] : acc[i][j][r];
            }
        }
    }
}

// dst[i] += sum_s part[s][i] in a fixed order
__global__ void rw_reduce_kernel(const float* __restrict__ part, long long stride, float* __restrict__ dst, long long n,
                                 int splits) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part[(size_t)k * stride + i];
        dst[i] += s;
    }
}

// The radial basis [E, R] as the RW_NT 16-bit terms of the product (RW_F16: x 2^14, hi / lo fp16), in the layout the producers
// copy into LDS: per 32-row chunk [term][row octet][column 0..127][8 rows] (16 KB; 24 KB with three bf16 terms), followed for all chunks by one mask word each (bit k: 32-column block k of the
// chunk holds a non-zero).  Written once per training step - the basis does not depend on the layer - instead of being split
// by every slice's workgroup of every layer's launch (8 x 6 times, ~120 vector instructions per thread and chunk).
__global__ __launch_bounds__(256) void rw_basis_image_kernel(const float* __restrict__ rbf, int E, int R, unsigned char* __restrict__ img,
                                                             int total_chunks) {
    __shared__ unsigned int blk[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    if (tid < 4) blk[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 256 * k, col = u & 127, oct = u >> 7;
        float v[8];
        bool nz = false;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = 32 * c + 8 * oct + r;
            v[r] = (row < E && col < R) ? rbf[(size_t)row * R + col] : 0.f;
            nz = nz || v[r] != 0.f;
        }
        if (nz) blk[col >> 5] = 1u;
        if (RW_F16) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] *= 16384.0f;   // exact; basis values down to 2^-28 keep an fp16-normal hi term
        }
        unsigned int tt[RW_NT][4];
        rw_split<8>(v, tt);
#pragma unroll
        for (int t = 0; t < RW_NT; ++t)
            *reinterpret_cast<uint4*>(img + (size_t)c * RW_IMG_CHUNK + (size_t)t * 8192 + (size_t)oct * 2048 + (size_t)col * 16) =
                make_uint4(tt[t][0], tt[t][1], tt[t][2], tt[t][3]);
    }
    __syncthreads();
    if (tid == 0)
        reinterpret_cast<unsigned int*>(img + (size_t)total_chunks * RW_IMG_CHUNK)[c] = blk[0] | (blk[1] << 1) | (blk[2] << 2) | (blk[3] << 3);
}

extern "C" int64_t adf_op_rbf_image_bytes(int64_t num_edges) {
    const int64_t chunks = (num_edges + 31) / 32;
    return chunks * (RW_IMG_CHUNK + 4) + 64;
}

extern "C" int32_t adf_op_rbf_image(adf_painn_t h, const float* rbf, int64_t num_edges, void* image, void* stream) {
    if (!h || !rbf || !image || num_edges <= 0 || h->hp.num_rbf > 128) { adf_set_error("rbf_image: bad argument"); return ADF_EINVAL; }
    const int chunks = (int)((num_edges + 31) / 32);
    hipLaunchKernelGGL(rw_basis_image_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, rbf, (int)num_edges,
                       h->hp.num_rbf, reinterpret_cast<unsigned char*>(image), chunks);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// Column lifts of d(rbfh) (RW_F16).  The contraction runs along the EDGES, so a power of two per COLUMN factors out of the sum.
// The columns' maxima are bounded from the node tables instead of from the 16 GB the kernel never writes:
//   |d_a[e,c]| <= max_i |gx[i,c]| max_j |xa[j,c]|,   |d_b| <= sum_d max|g_d| max|vec_d| / sqrt3 max|xb|,   |d_c| <= sum_d max|g_d| max|xc|
// (|u_d| <= 1).  rw_colmax_kernel: maxima of the 10 node-table columns per channel (|.| as unsigned bits, atomicMax);
// rw_lifts_kernel: lift = 2^(13 - floor(log2 bound)) so that every lifted value is below 2^14, unlift = 1 / (lift x 2^14)
// (2^14: the basis' own scale).  A value 2^-24 of its column's bound still has an fp16-normal hi term.
__global__ __launch_bounds__(256) void rw_colmax_kernel(const float* __restrict__ rec, const float* __restrict__ xh,
                                                        const float* __restrict__ vec, int N, int H, int rows_per_block,
                                                        unsigned int* __restrict__ mx) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    const int n0 = blockIdx.y * rows_per_block, n1 = min(N, n0 + rows_per_block);
    const size_t rec_row = (size_t)(H / 32) * 160, rec_off = (size_t)(c >> 5) * 160 + (size_t)(c & 31) * 4;
    float m[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int n = n0; n < n1; ++n) {
        const float4 g = *reinterpret_cast<const float4*>(rec + (size_t)n * rec_row + rec_off);
        const float* xr = xh + (size_t)n * 3 * H + c;
        m[0] = fmaxf(m[0], fabsf(g.x)); m[1] = fmaxf(m[1], fabsf(g.y)); m[2] = fmaxf(m[2], fabsf(g.z)); m[3] = fmaxf(m[3], fabsf(g.w));
        m[4] = fmaxf(m[4], fabsf(xr[0])); m[5] = fmaxf(m[5], fabsf(xr[H])); m[6] = fmaxf(m[6], fabsf(xr[2 * H]));
        if (vec) {
            const float* vr = vec + (size_t)n * 3 * H + c;
            m[7] = fmaxf(m[7], fabsf(vr[0])); m[8] = fmaxf(m[8], fabsf(vr[H])); m[9] = fmaxf(m[9], fabsf(vr[2 * H]));
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) atomicMax(&mx[(size_t)k * H + c], __float_as_uint(m[k]));
}

__global__ void rw_lifts_kernel(const unsigned int* __restrict__ mx, int H, float* __restrict__ lift, float* __restrict__ unlift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    float m[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) m[k] = __uint_as_float(mx[(size_t)k * H + c]);
    const float inv_sqrt3 = 0.57735026918962576f;
    const float bound[3] = {m[3] * m[4], (m[0] * m[7] + m[1] * m[8] + m[2] * m[9]) * inv_sqrt3 * m[5], (m[0] + m[1] + m[2]) * m[6]};
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const float b = bound[part];
        int e = (b > 0.f && b < 3.0e38f) ? 13 - ilogbf(b) : 0;      // b x 2^e in [2^13, 2^14)
        e = max(-100, min(100, e));
        const float l = ldexpf(1.0f, e);
        lift[part * H + c] = part == 1 ? l * inv_sqrt3 : l;            // part b: the 1/sqrt3 of the value rides on the lift
        unlift[part * H + c] = ldexpf(1.0f, -e - 14);
    }
}

// owner[e] = the atom whose CSR segment holds edge row e (binary search over nptr)
__global__ void rw_owner_kernel(const int32_t* __restrict__ nptr, int N, int32_t* __restrict__ owner, int E) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        int lo = 0, hi = N;   // invariant: nptr[lo] <= e < nptr[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (nptr[mid] <= e) lo = mid; else hi = mid;
        }
        owner[e] = lo;
    }
}

extern "C" int32_t adf_op_edge_owner(adf_painn_t h, int32_t* owner, int64_t num_edges, void* stream) {
    if (!h || h->lastN <= 0 || !owner || num_edges <= 0) { adf_set_error("edge_owner: bad argument or no graph"); return ADF_EINVAL; }
    long long blocks = (num_edges + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(rw_owner_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, h->nptr, (int)h->lastN, owner,
                       (int)num_edges);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

static size_t rw_lds_bytes() { return (size_t)2 * RW_BUF; }

static int rw_splits(const adf_painn* h, int nslices) {   // workgroups per slice: one per compute unit of the slice's XCD
    int s = h->num_cus / nslices;
    if (s < 1) s = 1;
    return s;
}

extern "C" int64_t adf_op_rbf_wgrad_fused_scratch(adf_painn_t h) {
    if (!h) return 0;
    const int H = h->hp.hidden_channels, R = h->hp.num_rbf;
    return (int64_t)rw_splits(h, H / ADF_SLICE_CH) * ((int64_t)3 * H * R) + 16 * (int64_t)H + 64;   // partial gradients + maxima, lifts
}

// dW [3H, R] of the layer's rbf_proj is ACCUMULATED (the bias gradient: message_bwd.hip's per-atom column sums).  Uses the gradient records the preceding
// adf_op_message_bwd_fused call of the same layer left in the handle (same gx1 / gv1), the graph of the handle, and
// `edge_owner` from adf_op_edge_owner, the basis image of adf_op_rbf_image.
extern "C" int32_t adf_op_rbf_wgrad_fused(adf_painn_t h, const float* xh, const float* vec, const void* rbf_image,
                                          const int32_t* edge_owner, int64_t num_edges, float* dW, float* scratch,
                                          int32_t vec_is_zero, void* stream) {
    if (!h || h->lastN <= 0 || !xh || !rbf_image || !edge_owner || !dW || !scratch || num_edges <= 0 || (!vec_is_zero && !vec)) {
        adf_set_error("rbf_wgrad_fused: bad argument or no graph");
        return ADF_EINVAL;
    }
    const int N = (int)h->lastN, H = h->hp.hidden_channels, R = h->hp.num_rbf;
    if (H % ADF_SLICE_CH != 0 || R > 128 || (R % 4) != 0) {
        adf_set_error("rbf_wgrad_fused: needs hidden_channels %% 64 == 0 and num_rbf <= 128, a multiple of 4");
        return ADF_EINVAL;
    }

    if ((unsigned long long)(N + 1) * 5ull * H * sizeof(float) >= (1ull << 32) || (unsigned long long)num_edges * 16ull >= (1ull << 32)) {
        adf_set_error("rbf_wgrad_fused: 32-bit byte offsets into the node / edge tables: N=%d, E=%lld is too large, split the batch", N,
                      (long long)num_edges);
        return ADF_EOOM;
    }
    hipStream_t s = (hipStream_t)stream;
    RbfWgradParams p;
    memset(&p, 0, sizeof(p));
    p.rec = h->rec; p.xh = xh; p.vec = vec; p.img = reinterpret_cast<const unsigned char*>(rbf_image); p.e_src = h->e_src; p.owner = edge_owner; p.e_geom = h->e_geom;
    p.E = (int)num_edges; p.N = N; p.H = H; p.R = R; p.nslices = H / ADF_SLICE_CH;
    int splits = rw_splits(h, p.nslices);
    p.workers = splits;
    p.part = scratch;
#if RW_F16
    {   // column lifts of this layer's d(rbfh): maxima of the node tables -> powers of two
        float* lifts = scratch + (size_t)splits * 3 * H * R;
        unsigned int* mx = reinterpret_cast<unsigned int*>(lifts + 6 * (size_t)H);
        ADF_HIP_CHECK(hipMemsetAsync(mx, 0, sizeof(unsigned int) * 10 * (size_t)H, s));
        const int row_blocks = 2 * h->num_cus, rows_per_block = (N + row_blocks - 1) / row_blocks;
        hipLaunchKernelGGL(rw_colmax_kernel, dim3((unsigned)((H + 255) / 256), (unsigned)row_blocks), dim3(256), 0, s, h->rec, xh,
                           vec_is_zero ? (const float*)nullptr : vec, N, H, rows_per_block, mx);
        hipLaunchKernelGGL(rw_lifts_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, s, mx, H, lifts, lifts + 3 * (size_t)H);
        p.lift = lifts;
        p.unlift = lifts + 3 * (size_t)H;
    }
#endif
    const dim3 grid((unsigned)(splits * p.nslices));
    static bool attr_set = false;  // per process and device: training runs on one device per process
    if (!attr_set) {
        ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rw_rbf_wgrad_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)rw_lds_bytes()));
        ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rw_rbf_wgrad_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)rw_lds_bytes()));
        attr_set = true;
    }
#if RW_PROF
    unsigned long long* prof_dev = nullptr;
    ADF_HIP_CHECK(hipMalloc(&prof_dev, sizeof(unsigned long long) * 8 * grid.x));
    ADF_HIP_CHECK(hipMemsetAsync(prof_dev, 0, sizeof(unsigned long long) * 8 * grid.x, s));
    p.prof = prof_dev;
#endif
    if (vec_is_zero) hipLaunchKernelGGL(rw_rbf_wgrad_kernel<true>, grid, dim3(RW_THREADS), rw_lds_bytes(), s, p);
    else hipLaunchKernelGGL(rw_rbf_wgrad_kernel<false>, grid, dim3(RW_THREADS), rw_lds_bytes(), s, p);
#if RW_PROF
    {
        std::vector<unsigned long long> hp((size_t)8 * grid.x);
        ADF_HIP_CHECK(hipStreamSynchronize(s));
        ADF_HIP_CHECK(hipMemcpy(hp.data(), prof_dev, sizeof(unsigned long long) * hp.size(), hipMemcpyDeviceToHost));
        (void)hipFree(prof_dev);
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < hp.size(); ++i) sum[i & 7] += (double)hp[i];
        const double nc = sum[3] > 0 ? sum[3] : 1.0;
        fprintf(stderr, "rw_prof (cycles per chunk): consumer basis %.0f products %.0f barrier %.0f | producer stage %.0f request %.0f barrier %.0f | chunks/wg %.0f\n",
                sum[0] / nc, sum[1] / nc, sum[2] / nc, sum[4] / nc, sum[5] / nc, sum[6] / nc, nc / grid.x);
    }
#endif
    const long long nW = (long long)3 * H * R;
    long long blocks = (nW + 255) / 256;
    hipLaunchKernelGGL(rw_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p.part, nW, dW, nW, splits);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
