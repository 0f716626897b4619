// Periodic radius graph + strict top-K + symmetrisation, built on the device every forward.
//
// Reference semantics (all in adsorbdiff/):
//   utils/utils.py:556-730   radius_graph_pbc      candidates (centre i, neighbour j, lattice shift)
//                                                   with 1e-4 < d^2 <= rc^2, ordered by (i, j, shift)
//   utils/utils.py:733-853   get_max_neighbors_mask keep the K nearest per centre (strict)
//   utils/utils.py:513-553   get_pbc_distances      v = pos[j]-pos[i] + shift.cell, d = |v|
//   models/painn/painn_denoising.py:262-327         keep j<i (or same atom & lexicographically
//                                                   negative shift), append reversed copies
//
// MI355X design: the graph is integer/HBM-latency work, not GEMM work.
//   1. adf_topk_kernel: one 256-thread workgroup per centre atom.  All n x n_shift candidates
//      of the centre's own system are evaluated from L2-resident positions; in-cutoff ones are
//      compacted into LDS, then the K smallest are selected by rank counting on the key
//      (d^2, candidate index) — i.e. a *stable* sort order.  (The reference's torch.sort is not
//      stable, so for exact d^2 ties at the K-th place its pick is arbitrary; ours is the
//      lowest candidate index.  Everything else is bit-identical: d^2 is evaluated with the
//      reference's fp32 operation order, no FMA contraction.)  Survivors are written in
//      candidate order, which is the reference's edge order before symmetrisation.
//   2. count / scan / fill: the symmetrised edge list is produced directly in the layout the
//      message kernel consumes — edges grouped by ADF_GROUP_NODES consecutive *target* atoms
//      (a CSR over node groups), each record = (source, target-in-group) + (unit vector, d).
#include "common.h"

struct GraphParams {
    const float* pos;
    const float* cell;
    const int32_t* batch;
    const int32_t* atom_offset;
    int r0, r1, r2;
    float rc2;
    int K;
    int N;
    int32_t* nbr_cnt;
    int32_t* nbr_src;
    int32_t* nbr_shift;
    int32_t* img_cnt;
    int32_t* flags;
};

__device__ __forceinline__ void decode_shift(int c, int r0, int r1, int r2, float& sa, float& sb, float& sc) {
    const int n2 = 2 * r2 + 1, n1 = 2 * r1 + 1;
    const int ia = c / (n1 * n2);
    const int rem = c - ia * (n1 * n2);
    const int ib = rem / n2;
    const int ic = rem - ib * n2;
    sa = (float)(ia - r0);
    sb = (float)(ib - r1);
    sc = (float)(ic - r2);
}

__global__ __launch_bounds__(256) void adf_topk_kernel(GraphParams p) {
    __shared__ float s_d2[ADF_MAX_CAND];
    __shared__ int32_t s_id[ADF_MAX_CAND];
    __shared__ int32_t s_kept[ADF_MAX_K];
    __shared__ int32_t s_count, s_nkept;
    const int i = blockIdx.x;
    const int tid = threadIdx.x;
    const int b = p.batch[i];
    const int a0 = p.atom_offset[b];
    const int n = p.atom_offset[b + 1] - a0;
    const int C = (2 * p.r0 + 1) * (2 * p.r1 + 1) * (2 * p.r2 + 1);
    const int ncand = n * C;
    if (tid == 0) { s_count = 0; s_nkept = 0; }
    const float* cl = p.cell + 9 * b;
    const float c00 = cl[0], c01 = cl[1], c02 = cl[2];
    const float c10 = cl[3], c11 = cl[4], c12 = cl[5];
    const float c20 = cl[6], c21 = cl[7], c22 = cl[8];
    const float pix = p.pos[3 * i], piy = p.pos[3 * i + 1], piz = p.pos[3 * i + 2];
    __syncthreads();
    for (int cid = tid; cid < ncand; cid += 256) {
        const int j = cid / C;
        const int c = cid - j * C;
        float sa, sb, sc;
        decode_shift(c, p.r0, p.r1, p.r2, sa, sb, sc);
        // offset = cell^T . shift, summed in k order without FMA (utils.py:680-681)
        const float ox = __fadd_rn(__fadd_rn(__fmul_rn(c00, sa), __fmul_rn(c10, sb)), __fmul_rn(c20, sc));
        const float oy = __fadd_rn(__fadd_rn(__fmul_rn(c01, sa), __fmul_rn(c11, sb)), __fmul_rn(c21, sc));
        const float oz = __fadd_rn(__fadd_rn(__fmul_rn(c02, sa), __fmul_rn(c12, sb)), __fmul_rn(c22, sc));
        const float* pj = p.pos + 3 * (size_t)(a0 + j);
        const float dx = __fsub_rn(pix, __fadd_rn(pj[0], ox));
        const float dy = __fsub_rn(piy, __fadd_rn(pj[1], oy));
        const float dz = __fsub_rn(piz, __fadd_rn(pj[2], oz));
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        if (d2 <= p.rc2 && d2 > 0.0001f) {
            const int slot = atomicAdd(&s_count, 1);
            if (slot < ADF_MAX_CAND) { s_d2[slot] = d2; s_id[slot] = cid; }
        }
    }
    __syncthreads();
    int M = s_count;
    if (M > ADF_MAX_CAND) {
        if (tid == 0) atomicExch(&p.flags[0], 1);
        M = ADF_MAX_CAND;
    }
    const int K = p.K;
    // rank selection on (d2, cid)
    for (int e = tid; e < M; e += 256) {
        bool keep = true;
        if (M > K) {
            const float d = s_d2[e];
            const int id = s_id[e];
            int rank = 0;
            for (int f = 0; f < M; ++f) {
                const float df = s_d2[f];
                rank += (df < d) || (df == d && s_id[f] < id);
            }
            keep = rank < K;
        }
        if (keep) {
            const int slot = atomicAdd(&s_nkept, 1);
            s_kept[slot] = s_id[e];
        }
    }
    __syncthreads();
    const int nk = s_nkept;
    if (tid < nk) {
        const int id = s_kept[tid];
        int posn = 0;
        for (int f = 0; f < nk; ++f) posn += s_kept[f] < id;
        const int j = id / C;
        p.nbr_src[(size_t)i * K + posn] = a0 + j;
        p.nbr_shift[(size_t)i * K + posn] = id - j * C;
    }
    if (tid == 0) {
        p.nbr_cnt[i] = nk;
        if (nk) atomicAdd(&p.img_cnt[b], nk);
    }
}

__device__ __forceinline__ bool edge_kept(int j, int i, int c, int r0, int r1, int r2) {
    if (j < i) return true;
    if (j != i) return false;
    float sa, sb, sc;
    decode_shift(c, r0, r1, r2, sa, sb, sc);
    return (sa < 0.f) || (sa == 0.f && sb < 0.f) || (sa == 0.f && sb == 0.f && sc < 0.f);
}

__global__ void adf_count_kernel(GraphParams p, int32_t* gcount) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(t / p.K);
    const int k = (int)(t - (long long)i * p.K);
    if (i >= p.N || k >= p.nbr_cnt[i]) return;
    const int j = p.nbr_src[(size_t)i * p.K + k];
    const int c = p.nbr_shift[(size_t)i * p.K + k];
    if (!edge_kept(j, i, c, p.r0, p.r1, p.r2)) return;
    atomicAdd(&gcount[i / ADF_GROUP_NODES], 1);
    atomicAdd(&gcount[j / ADF_GROUP_NODES], 1);
}

// single-block exclusive scan over the group counts; also validates capacities / empty images
__global__ __launch_bounds__(1024) void adf_scan_kernel(const int32_t* gcount, int32_t* gptr, int32_t* gcursor,
                                                         int G, long long capE, const int32_t* img_cnt, int B,
                                                         int32_t* flags) {
    __shared__ int32_t s_part[1024];
    __shared__ int32_t s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < G; base += 1024) {
        const int idx = base + tid;
        const int v = idx < G ? gcount[idx] : 0;
        s_part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int add = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += add;
            __syncthreads();
        }
        const int incl = s_part[tid] + s_carry;
        if (idx < G) { gptr[idx] = incl - v; gcursor[idx] = incl - v; }
        __syncthreads();
        if (tid == 1023) s_carry = incl;
        __syncthreads();
    }
    if (tid == 0) {
        gptr[G] = s_carry;
        if ((long long)s_carry > capE) atomicExch(&flags[2], 1);
    }
    for (int b = tid; b < B; b += 1024)
        if (img_cnt[b] == 0) atomicExch(&flags[1], 1);
}

__global__ void adf_fill_kernel(GraphParams p, int32_t* gcursor, adf_edge_meta* e_meta, float4* e_geom,
                                long long capE) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(t / p.K);
    const int k = (int)(t - (long long)i * p.K);
    if (i >= p.N || k >= p.nbr_cnt[i]) return;
    const int j = p.nbr_src[(size_t)i * p.K + k];
    const int c = p.nbr_shift[(size_t)i * p.K + k];
    if (!edge_kept(j, i, c, p.r0, p.r1, p.r2)) return;
    float sa, sb, sc;
    decode_shift(c, p.r0, p.r1, p.r2, sa, sb, sc);
    const float* cl = p.cell + 9 * p.batch[i];
    // shift (row vector) . cell   (utils.py:529)
    const float ox = __fadd_rn(__fadd_rn(__fmul_rn(sa, cl[0]), __fmul_rn(sb, cl[3])), __fmul_rn(sc, cl[6]));
    const float oy = __fadd_rn(__fadd_rn(__fmul_rn(sa, cl[1]), __fmul_rn(sb, cl[4])), __fmul_rn(sc, cl[7]));
    const float oz = __fadd_rn(__fadd_rn(__fmul_rn(sa, cl[2]), __fmul_rn(sb, cl[5])), __fmul_rn(sc, cl[8]));
    const float vx = __fadd_rn(__fsub_rn(p.pos[3 * j], p.pos[3 * i]), ox);
    const float vy = __fadd_rn(__fsub_rn(p.pos[3 * j + 1], p.pos[3 * i + 1]), oy);
    const float vz = __fadd_rn(__fsub_rn(p.pos[3 * j + 2], p.pos[3 * i + 2]), oz);
    float d = sqrtf(fmaf(vz, vz, fmaf(vy, vy, vx * vx)));
    if (d == 0.f) return;  // utils.py:536-540 (unreachable after the d^2 > 1e-4 filter; counted edges stay padded)
    if (fabsf(d) <= 1.0e-3f) d = 1.0e-3f;  // painn_denoising.py:366-367
    const float ux = vx / d, uy = vy / d, uz = vz / d;
    // edge j -> i, stored in the group of its target i
    int slot = atomicAdd(&gcursor[i / ADF_GROUP_NODES], 1);
    if (slot < capE) {
        e_meta[slot] = adf_edge_meta{j, i % ADF_GROUP_NODES};
        e_geom[slot] = make_float4(ux, uy, uz, d);
    }
    // reversed copy i -> j, same distance, negated unit vector (painn_denoising.py:171-181,322-327)
    slot = atomicAdd(&gcursor[j / ADF_GROUP_NODES], 1);
    if (slot < capE) {
        e_meta[slot] = adf_edge_meta{i, j % ADF_GROUP_NODES};
        e_geom[slot] = make_float4(-ux, -uy, -uz, d);
    }
}

int32_t adf_graph_build_impl(adf_painn* h, const adf_batch* b, hipStream_t s) {
    const int N = b->num_atoms, B = b->num_systems, K = h->hp.max_neighbors;
    const int G = (N + ADF_GROUP_NODES - 1) / ADF_GROUP_NODES;
    GraphParams p;
    p.pos = b->pos; p.cell = b->cell; p.batch = b->batch; p.atom_offset = b->atom_offset;
    p.r0 = b->reps[0]; p.r1 = b->reps[1]; p.r2 = b->reps[2];
    p.rc2 = h->hp.cutoff * h->hp.cutoff;
    p.K = K; p.N = N;
    p.nbr_cnt = h->nbr_cnt; p.nbr_src = h->nbr_src; p.nbr_shift = h->nbr_shift;
    p.img_cnt = h->gcursor + (G + 1);  // [B] scratch behind the cursors
    p.flags = h->flags;
    ADF_HIP_CHECK(hipMemsetAsync(h->gcount, 0, sizeof(int32_t) * (G + 1), s));
    ADF_HIP_CHECK(hipMemsetAsync(p.img_cnt, 0, sizeof(int32_t) * B, s));
    ADF_HIP_CHECK(hipMemsetAsync(h->flags, 0, sizeof(int32_t) * 4, s));
    hipLaunchKernelGGL(adf_topk_kernel, dim3(N), dim3(256), 0, s, p);
    const long long slots = (long long)N * K;
    const unsigned nb = (unsigned)((slots + 255) / 256);
    hipLaunchKernelGGL(adf_count_kernel, dim3(nb), dim3(256), 0, s, p, h->gcount);
    hipLaunchKernelGGL(adf_scan_kernel, dim3(1), dim3(1024), 0, s, h->gcount, h->gptr, h->gcursor, G,
                       (long long)h->capE, p.img_cnt, B, h->flags);
    hipLaunchKernelGGL(adf_fill_kernel, dim3(nb), dim3(256), 0, s, p, h->gcursor, h->e_meta, h->e_geom,
                       (long long)h->capE);
    ADF_HIP_CHECK(hipGetLastError());
    h->lastN = N; h->lastB = B;
    h->last_reps[0] = p.r0; h->last_reps[1] = p.r1; h->last_reps[2] = p.r2;
    return ADF_OK;
}
