// RETIRED EXPERIMENT (round 5), kept for the record; not built.  The "octet" layout of csrc/rbf_wgrad.hip: a producer wave owns
// the edge rows 8 w .. 8 w + 7 of every chunk and lane = channel, so that a row's (neighbour, owner, unit vector) are
// wave-uniform (scalar loads), every gather instruction reads two full 512-byte record groups, the owner's xh / vec rows are
// read once per OWNER (an octet has one or two; rows of a third one are zeroed, their chunk flagged and replayed behind the
// main loop), d(rbfh) is staged column-major (8 rows of a column = one ds_write_b128; fragments by plain ds_read_b128).
// Measured 8.1-9.2 ms per launch at 256 graphs against 6.5 ms of the thread = (edge row, 8 channels) layout it was meant to
// beat (5.3 ms in its final form): SQ_INSTS_VALU 865 per SIMD and chunk against 570 - 350 scalar instructions of bookkeeping
// per chunk, the lane <-> scalar moves of spilled SGPRs (v_readlane / v_writelane are vector-issue slots) and the per-row
// owner selection cost more vector-issue time than the coalescing saved, and the kernel is bound by vector issue + MFMA, not
// by memory.  See profiles/NOTES.md (round 5, training).
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
// the product exactly as tr_wgrad_bf16x6_kernel does: both operands split into three bf16 terms, six products, rows staged
// row-major in LDS and read with the transposing ds_read_b64_tr_b16, all-zero 32-column blocks of the radial basis skipped.
// A workgroup owns the 192 gradient rows of its slice (a, b, c parts of 64 channels) x all R <= 128 basis functions for one
// range of edges; the per-range partial results are summed in a fixed order (run-to-run reproducible), directly into the
// reference's row order.  The values staged are computed by the same expressions as message_bwd.hip's stores were.
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
    const unsigned char* img;   // radial basis as three bf16 terms in the kernel's LDS layout (rw_basis_image_kernel), + masks
    const int32_t* e_src;    // neighbour i of edge row e
    const int32_t* owner;    // atom j whose CSR segment holds row e
    const float4* e_geom;    // (unit vector, distance) of row e
    float* part;             // [splits][3H x R] partial gradients, reference row order
    unsigned long long* prof;   // RW_PROF builds only: [workgroup][8] cycle sums
    int E, N, H, R, workers, nslices;   // workers: workgroups per slice (= edge-range partial results)
};

// LDS images, one buffer per chunk in flight: both operands COLUMN-major - [term][column][32 edge rows] bf16, 80 bytes per
// column (64 + 16 pad: the 16-byte accesses of 16 consecutive columns fall into 16 different bank groups) - which is the
// K-contiguous layout the 32x32x16 instruction wants: a fragment is one ds_read_b128, no transposing read.
#define RW_COLB 80
#define RW_CIMG (192 * RW_COLB)                 // one term of d(rbfh)
#define RW_AIMG (128 * RW_COLB)                 // one term of the radial basis
#define RW_IMG (3 * (RW_CIMG + RW_AIMG))
#define RW_BUF (RW_IMG + 64)                    // + the chunk's mask of non-zero 32-column basis blocks
#define RW_IMG_CHUNK (3 * 4 * 128 * 16)         // global image of the basis: [chunk][term][row octet][column][8 rows] bf16
#define RW_MASK_WORDS 1024                      // one bit per chunk of a workgroup: the chunk has an octet with a third owner
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
#define RW_ABL 0   // timing experiments (wrong results): 1 no gathers, 2 no products, 4 no d(rbfh) conversion / stores, 8 no basis staging
#endif

// v[0..7] -> three bf16 terms t[0] + t[1] + t[2] = v exactly (24 significant bits), round-to-nearest terms.  The packed conversion
// of a pair is reused for the residuals (low half << 16, high half masked): 5.5 vector instructions per value.
typedef float rw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rw_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rw_split3(float* v, rw_bf16x8* t) {
    union { rw_bf16x8 b; unsigned int u[4]; } o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            union { rw_bf16x2 b; unsigned int u; } pk;
            pk.b = __builtin_convertvector((rw_f32x2){v[e], v[e + 1]}, rw_bf16x2);
            o[k].u[e >> 1] = pk.u;
            if (k < 2) {
                v[e] -= __uint_as_float(pk.u << 16);
                v[e + 1] -= __uint_as_float(pk.u & 0xffff0000u);
            }
        }
    t[0] = o[0].b; t[1] = o[1].b; t[2] = o[2].b;
}

struct rw_yes { static const bool value = true; };
struct rw_no { static const bool value = false; };
typedef const __attribute__((address_space(4))) int32_t* rw_cint_ptr;
typedef const __attribute__((address_space(4))) float* rw_cflt_ptr;

// (neighbour, owner, unit vector) of the 8 edge rows of a wave's octet: wave-uniform, read through the scalar cache
struct rw_meta {
    int src[8], own[8];
    float u[8][3];
};
// what a producer wave keeps in flight for one chunk: lane = channel c0 + lane, rows = the wave's octet
struct rw_gather_set {
    float4 g[8];          // the neighbours' packed gradients (g0, g1, g2, gx) of this channel
    float cA[6], cB[6];   // (xa, xb, xc, wx, wy, wz) of the octet's first owner and of its second one
    float u[8][3];        // uniform
    int own[8];           // uniform
    int oA, oB, slow;     // uniform; slow: a third owner in the octet
};

// Waves 4-7 PRODUCE d(rbfh): producer wave w owns the edge rows 8 w .. 8 w + 7 of every chunk, lane = channel.  Per row one
// 16-byte gather per lane (the 64 lanes read the slice's two 512-byte record groups of the neighbour: fully coalesced), per
// OWNER - an octet of consecutive CSR rows has one or two - six coalesced dword loads (xh, vec); a row of a third owner
// (atoms with fewer than four edges) reads its owner's rows when it is staged.  (Thread = one edge row x 8 channels: 48 KB of
// owner rows per chunk through the vector-memory path beside 32 KB of records - the request phase alone took 2400 cycles
// per chunk of back-pressure.)  The three parts are formed, split into three bf16 terms and written to LDS buffer
// (i + 1) & 1 - 8 rows of one column = one 16-byte store - while waves 0-3 CONSUME chunk i from buffer i & 1 (96
// accumulator registers each) and convert chunk i + 1's radial-basis rows.  One barrier per chunk.  Every request is TWO
// chunks ahead of its use in two alternating register sets (the producers hold no accumulators).  Measured on the way here
// (256 graphs, per launch): one role per wave and requests at the top of the chunk 6.7 ms (two dependent round trips,
// 5.2 us per chunk); requests one chunk ahead beside the accumulators: 268 B of scratch per lane, whose reloads drain the
// vector-memory counter and with it the prefetch; producer / consumer waves with requests one chunk ahead 8.5 ms (a request
// issued at the end of a chunk has only the barrier wait to land); two chunks ahead, row-major images 6.5 ms.
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
    unsigned int* slowmask = reinterpret_cast<unsigned int*>(rw_lds + (size_t)2 * RW_BUF);
    for (int i = tid; i < RW_MASK_WORDS; i += RW_THREADS) slowmask[i] = 0u;
    __syncthreads();

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pw = wave - 4;
        const float inv_sqrt3 = 0.57735026918962576f;
        const rw_cint_ptr esrc_c = (rw_cint_ptr)p.e_src, own_c = (rw_cint_ptr)p.owner;
        const rw_cflt_ptr geom_c = (rw_cflt_ptr) reinterpret_cast<const float*>(p.e_geom);
        const size_t rec_row_b = (size_t)(H / 32) * 160 * sizeof(float);
        const char* rec_lane = reinterpret_cast<const char*>(p.rec) + (size_t)(2 * slice + (lane >> 5)) * 640 + (size_t)(lane & 31) * 16;
        const float* xh_lane = p.xh + c0 + lane;
        const float* vec_lane = VZ ? nullptr : p.vec + c0 + lane;
        auto load_meta = [&](rw_meta& M, int t) {   // rows past the end: the zero record (row N), the last row's owner
            const int rb = row0(t) + 8 * pw;
#pragma unroll
            for (int r = 0; r < 8; ++r) {   // (unconditional loads at a clamped row; arithmetic, not a select: no branch around a load)
                const int e = rb + r, ec = min(e, p.E - 1);
                const int sv = esrc_c[ec], past = e >= p.E ? 1 : 0;
                M.src[r] = sv + (p.N - sv) * past;
                M.own[r] = own_c[ec];
                M.u[r][0] = geom_c[4 * (size_t)ec]; M.u[r][1] = geom_c[4 * (size_t)ec + 1]; M.u[r][2] = geom_c[4 * (size_t)ec + 2];
            }
        };
        auto load_owner = [&](float* c, int o) {
            const float* xr = xh_lane + (size_t)o * 3 * H;
            c[0] = xr[0]; c[2] = xr[2 * H];
            if (!VZ) {
                const float* vr = vec_lane + (size_t)o * 3 * H;
                c[1] = xr[H]; c[3] = vr[0]; c[4] = vr[H]; c[5] = vr[2 * H];
            } else {
                c[1] = c[3] = c[4] = c[5] = 0.f;
            }
        };
        auto request = [&](rw_gather_set& S, const rw_meta& M) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (!(RW_ABL & 1)) S.g[r] = *reinterpret_cast<const float4*>(rec_lane + (size_t)M.src[r] * rec_row_b);
                S.own[r] = M.own[r];
                S.u[r][0] = M.u[r][0]; S.u[r][1] = M.u[r][1]; S.u[r][2] = M.u[r][2];
            }
            S.oA = M.own[0];
            int oB = M.own[0];
#pragma unroll
            for (int r = 7; r >= 1; --r) oB = M.own[r] != M.own[0] ? M.own[r] : oB;   // ends at the FIRST row of another owner
            S.oB = oB;
            int slow = 0;
#pragma unroll
            for (int r = 1; r < 8; ++r) slow |= (M.own[r] != M.own[0] && M.own[r] != oB) ? 1 : 0;
            S.slow = slow;
            if (RW_ABL & 1) return;
            load_owner(S.cA, S.oA);
            if (!(RW_ABL & 16)) load_owner(S.cB, S.oB);   // (also when oB == oA: the same lines again, and no branch around the loads)
        };
        auto emit = [&](float* va, float* vb, float* vc, int b) {   // three parts x three bf16 terms; 8 rows of a column = 16 bytes
            unsigned char* imgC = rw_lds + (size_t)b * RW_BUF;
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                if (VZ && part == 1) continue;   // the b columns are never read on the first layer
                float* v = part == 0 ? va : part == 1 ? vb : vc;
                unsigned char* dst = imgC + (size_t)(64 * part + lane) * RW_COLB + 16 * pw;
                rw_bf16x8 tt[3];
                rw_split3(v, tt);
#pragma unroll
                for (int t = 0; t < 3; ++t) *reinterpret_cast<rw_bf16x8*>(dst + (size_t)t * RW_CIMG) = tt[t];
            }
        };
        auto row_values = [&](const float4 gg, const float* c, const float* u, float& a, float& bv, float& cv) {
            a = gg.w * c[0];
            const float T = -(gg.x * u[0] + gg.y * u[1] + gg.z * u[2]);
            cv = T * c[2];
            bv = 0.f;
            if (!VZ) {
                const float Sd = gg.x * c[3] + gg.y * c[4] + gg.z * c[5];
                bv = Sd * c[1];
            }
        };
        // d(rbfh) of this lane's channel, 8 rows -> buffer b.  No load in here (a branch around a load makes the compiler drain
        // the vector-memory counter at the join - and with it the other set's requests): a row of a third owner is staged as
        // zeros and its chunk is flagged; the flagged chunks are replayed behind the main loop for just those rows.
        auto stage = [&](rw_gather_set& S, int b, int t) {
            if (RW_ABL & 4) { if (S.g[0].x == 123.f && S.cA[0] == 1.f) rw_lds[tid] = 1; return; }
            float va[8], vb[8], vc[8];
            float cA[6], cB[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { cA[k] = S.cA[k]; cB[k] = S.cB[k]; }
#pragma unroll
            for (int k = 3; k < 6; ++k) { cA[k] *= inv_sqrt3; cB[k] *= inv_sqrt3; }
#pragma unroll
            for (int r = 0; r < 8; ++r) {   // (uniform branches: the owner is the wave's, not the lane's; six selects per row otherwise)
                if (S.own[r] == S.oA) row_values(S.g[r], cA, S.u[r], va[r], vb[r], vc[r]);
                else if (S.own[r] == S.oB) row_values(S.g[r], cB, S.u[r], va[r], vb[r], vc[r]);
                else { va[r] = 0.f; vb[r] = 0.f; vc[r] = 0.f; }
            }
            emit(va, vb, vc, b);
            if (S.slow && lane == 0) atomicOr(&slowmask[t >> 5], 1u << (t & 31));
        };
        // replay of a flagged chunk into buffer 0: only the rows of a third owner, every other row zero
        auto replay = [&](int t) {
            rw_meta M;
            load_meta(M, t);
            int oB = M.own[0];
#pragma unroll
            for (int r = 7; r >= 1; --r) oB = M.own[r] != M.own[0] ? M.own[r] : oB;
            float va[8], vb[8], vc[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                va[r] = vb[r] = vc[r] = 0.f;
                if (M.own[r] != M.own[0] && M.own[r] != oB) {   // uniform
                    const float4 gg = *reinterpret_cast<const float4*>(rec_lane + (size_t)M.src[r] * rec_row_b);
                    float c[6];
                    load_owner(c, M.own[r]);
#pragma unroll
                    for (int k = 3; k < 6; ++k) c[k] *= inv_sqrt3;
                    row_values(gg, c, M.u[r], va[r], vb[r], vc[r]);
                }
            }
            emit(va, vb, vc, 0);
        };
        // radial basis: already three bf16 terms in the LDS layout (rw_basis_image_kernel, once per step): a producer thread moves
        // two (column, row octet) units of 16 bytes per term (the consumers' registers are the accumulators)
        const int bcol = 64 * (pw & 1) + lane, boct = pw >> 1;   // unit k: column bcol, octet boct + 2 k
        const rw_cint_ptr mask_c = (rw_cint_ptr) reinterpret_cast<const int32_t*>(p.img + (size_t)total_chunks * RW_IMG_CHUNK);
        typedef unsigned int rw_u32x4 __attribute__((ext_vector_type(4)));
        struct basis_set { rw_u32x4 q[2][3]; int mask; };
        auto request_basis = [&](basis_set& B, int t) {
            const int gc = min(row0(t) >> 5, total_chunks - 1);   // (past the end: any chunk - its d(rbfh) rows are all zero)
            const unsigned char* src = p.img + (size_t)gc * RW_IMG_CHUNK + (size_t)bcol * 16;
    #pragma unroll
            for (int k = 0; k < 2; ++k)
    #pragma unroll
                for (int tm = 0; tm < 3; ++tm)
                    B.q[k][tm] = *reinterpret_cast<const rw_u32x4*>(src + (size_t)tm * 8192 + (size_t)(boct + 2 * k) * 2048);
            B.mask = mask_c[gc];
        };
        auto stage_basis = [&](basis_set& B, int b) {
            if (RW_ABL & 8) { if (B.q[0][0][0] == 123u) rw_lds[tid] = 1; return; }
            unsigned char* imgA = rw_lds + (size_t)b * RW_BUF + 3 * RW_CIMG;
            unsigned int* nzf = reinterpret_cast<unsigned int*>(rw_lds + (size_t)b * RW_BUF + RW_IMG);
    #pragma unroll
            for (int k = 0; k < 2; ++k)
    #pragma unroll
                for (int tm = 0; tm < 3; ++tm)
                    *reinterpret_cast<rw_u32x4*>(imgA + (size_t)tm * RW_AIMG + (size_t)bcol * RW_COLB + 16 * (boct + 2 * k)) = B.q[k][tm];
            if (tid == 256) nzf[0] = (unsigned int)B.mask;
        };
        // Half-step h: stage chunk h - 2 from set h & 1 into buffer h & 1, request chunk h into that set, read the edge rows of
        // chunk h + 1; a barrier behind every staged chunk.  The loop starts with nothing in flight (no peeled prologue: with
        // one, the compiler's wait-count state at the loop header was the conservative merge of two different histories and
        // one of the two stages waited for every outstanding load).
        rw_gather_set S0, S1;
        basis_set B0, B1;
        rw_meta M;
        load_meta(M, 0);
        unsigned long long pf[4] = {0, 0, 0, 0};
        (void)pf;
        for (int h = 0;; h += 2) {
            if (h - 2 >= nchunks) break;
            RW_T(t0);
            if (h >= 2) { stage(S0, 0, h - 2); stage_basis(B0, 0); }
            RW_T(t1);
            request(S0, M);
            request_basis(B0, h);
            load_meta(M, h + 1);
            RW_T(t2);
            if (h >= 2) __syncthreads();
            RW_T(t3);
            RW_ACC(0, t0, t1); RW_ACC(1, t1, t2); RW_ACC(2, t2, t3);
            if (h - 1 >= nchunks) break;
            if (h >= 2) { stage(S1, 1, h - 1); stage_basis(B1, 1); }
            RW_T(t4);
            request(S1, M);
            request_basis(B1, h + 1);
            load_meta(M, h + 2);
            RW_T(t5);
            if (h >= 2) __syncthreads();
            RW_T(t6);
            RW_ACC(0, t3, t4); RW_ACC(1, t4, t5); RW_ACC(2, t5, t6);
        }
        __syncthreads();   // the consumers' barrier behind the last chunk's products
#if RW_PROF
        if (tid == 256 && p.prof) for (int k = 0; k < 3; ++k) p.prof[(size_t)blockIdx.x * 8 + 4 + k] = pf[k];
#endif
        for (int t = 0; t < nchunks; ++t) {
            if (!((slowmask[t >> 5] >> (t & 31)) & 1u)) continue;
            replay(t);
            request_basis(B0, t);
            stage_basis(B0, 0);
            __syncthreads();
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumers
    const int kb = wave & 1, cgp = wave >> 1;
    const int fm = lane & 31, fkg = lane >> 5;   // fragment addressing: column fm of a 32-column block, k half fkg
    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // the six products of one (32-column block of d(rbfh)) x (32 basis functions) pair over one 16-row step
    auto six = [&](f32x16& c, const rw_bf16x8* a, const rw_bf16x8* bq) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bq[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[0], c, 0, 0, 0);
    };
    // D0 / D1: this wave's basis blocks kb / kb + 2 hold a non-zero in this chunk (straight-line code per case, so that the
    // fragment reads of the next pair are issued behind the products of the current one)
    auto products_case = [&](const unsigned char* imgC, const unsigned char* imgA, auto D0, auto D1) {
#pragma unroll 1
        for (int ks = 0; ks < 2; ++ks) {   // (not unrolled: both steps' fragments in flight at once spill beside the accumulators)
            const unsigned int koff = 32u * ks + 16u * fkg;
            rw_bf16x8 b0[3], b1[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (D0.value) b0[t] = *reinterpret_cast<const rw_bf16x8*>(imgA + (size_t)t * RW_AIMG + (size_t)(32 * kb + fm) * RW_COLB + koff);
                if (D1.value) b1[t] = *reinterpret_cast<const rw_bf16x8*>(imgA + (size_t)t * RW_AIMG + (size_t)(32 * (kb + 2) + fm) * RW_COLB + koff);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int bi = 3 * cgp + i;           // 32-column block of the slice's 192: part = bi / 2
                if (VZ && (bi >> 1) == 1) continue;
                rw_bf16x8 a[3];
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    a[t] = *reinterpret_cast<const rw_bf16x8*>(imgC + (size_t)t * RW_CIMG + (size_t)(32 * bi + fm) * RW_COLB + koff);
                if (D0.value) six(acc[i][0], a, b0);
                if (D1.value) six(acc[i][1], a, b1);
            }
        }
    };
    auto products = [&](int b) {
        const unsigned char* imgC = rw_lds + (size_t)b * RW_BUF;
        const unsigned char* imgA = imgC + 3 * RW_CIMG;
        const unsigned int* nzf = reinterpret_cast<const unsigned int*>(imgC + RW_IMG);
        const unsigned int nzmask = nzf[0];                                   // bit k: basis block k of this chunk holds a non-zero
        const bool do0 = ((nzmask >> kb) & 1u) != 0u, do1 = ((nzmask >> (kb + 2)) & 1u) != 0u;   // wave-uniform
        if (RW_ABL & 2) return;
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
    for (int t = 0; t < nchunks; ++t) {   // chunks with rows of a third owner: see the producers
        if (!((slowmask[t >> 5] >> (t & 31)) & 1u)) continue;
        __syncthreads();
        products(0);
        __syncthreads();
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
                if (col < R) out[(size_t)row * R + col] = acc[i][j][r];
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

// The radial basis [E, R] as the three bf16 terms of the product, in the layout the consumers copy into LDS: per 32-row chunk
// [term][row octet][column 0..127][8 rows] (24 KB), followed for all chunks by one mask word each (bit k: 32-column block k of the
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
        rw_bf16x8 tt[3];
        rw_split3(v, tt);
#pragma unroll
        for (int t = 0; t < 3; ++t)
            *reinterpret_cast<rw_bf16x8*>(img + (size_t)c * RW_IMG_CHUNK + (size_t)t * 8192 + (size_t)oct * 2048 + (size_t)col * 16) = tt[t];
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

static size_t rw_lds_bytes() { return (size_t)2 * RW_BUF + RW_MASK_WORDS * sizeof(unsigned int); }

static int rw_splits(const adf_painn* h, int nslices) {   // workgroups per slice: one per compute unit of the slice's XCD
    int s = h->num_cus / nslices;
    if (s < 1) s = 1;
    return s;
}

extern "C" int64_t adf_op_rbf_wgrad_fused_scratch(adf_painn_t h) {
    if (!h) return 0;
    const int H = h->hp.hidden_channels, R = h->hp.num_rbf;
    return (int64_t)rw_splits(h, H / ADF_SLICE_CH) * ((int64_t)3 * H * R);
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
    if ((num_edges + 31) / 32 > (int64_t)32 * RW_MASK_WORDS * rw_splits(h, H / ADF_SLICE_CH)) {
        adf_set_error("rbf_wgrad_fused: %lld edges exceed the kernel's chunk mask; split the batch", (long long)num_edges);
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
