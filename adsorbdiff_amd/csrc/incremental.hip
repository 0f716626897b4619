// Incremental layers: bookkeeping kernels of api.hip's forward_incremental.
//
// In the sampling loop (denoising_torch.py:235-356) only the adsorbate moves, so from one step to the next most node
// rows of the early layers see bit-identical inputs: layer 0 changes only where an atom's in-edge list or geometry
// changed, layer 1 where a layer-0 row or one of its in-neighbours' rows changed, and so on — the receptive field of
// the moving atoms grows by one neighbour shell per layer.  The handle keeps x / vec / gather records of every layer
// across the forwards of one static-atom promise and recomputes a row only when one of its inputs changed since the
// row was computed.  Change is DETECTED, not predicted: the CSR of this build is compared bit for bit with the previous
// build's, and flags are propagated along the current edges.  Per-row arithmetic is the same code on the same inputs,
// so every output is bit-identical to a full forward (tests/test_gpu_parity.py::test_incremental_*).
//
//   c0[i]        in-edge list of i (sources, unit vectors, distances) differs from the previous build
//   chg_l[i]     x / vec entering layer l at row i differ from the previous forward's   (chg_0 = 0: x0 = emb(Z))
//   out_l[i]   = c0[i] | chg_l[i] | OR_{j -> i} chg_l[j]            (true value of layer l's output row changed)
//   pend_l[i]    the kept row has unapplied changes;  need_l[i]: the row is wanted (all rows unless the caller only
//                wants outputs on a subset: then need_{L-1} = subset, need_{l-1} = need_l + its in-neighbours)
//   tf_l[i]    = need_l[i] & (pend_l[i] | out_l[i])   -> recompute list of layer l;   pend_l = (pend_l | out_l) & ~tf_l
#include <hipcub/hipcub.hpp>

#include "common.h"

__global__ __launch_bounds__(256) void adf_inc_compare_kernel(const int32_t* __restrict__ nptr, const int32_t* __restrict__ src,
                                                               const float4* __restrict__ geo, const int32_t* __restrict__ pnptr,
                                                               const int32_t* __restrict__ psrc, const float4* __restrict__ pgeo,
                                                               int N, unsigned char* __restrict__ c0) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const int e0 = nptr[n], d = nptr[n + 1] - e0, p0 = pnptr[n], pd = pnptr[n + 1] - p0;
    int diff = d != pd;
    if (!diff) {
        for (int t = lane; t < d; t += 64) {
            const float4 a = geo[e0 + t], b = pgeo[p0 + t];
            diff |= (src[e0 + t] != psrc[p0 + t]) | (__float_as_uint(a.x) != __float_as_uint(b.x)) |
                    (__float_as_uint(a.y) != __float_as_uint(b.y)) | (__float_as_uint(a.z) != __float_as_uint(b.z)) |
                    (__float_as_uint(a.w) != __float_as_uint(b.w));
        }
    }
    const bool any = __ballot(diff) != 0ull;
    if (lane == 0) c0[n] = any ? 1 : 0;
}

// one wave per 64 targets (thread = target): flags of one layer + in-edge total of the listed rows
__global__ __launch_bounds__(256) void adf_inc_flags_kernel(const int32_t* __restrict__ nptr, const int32_t* __restrict__ src,
                                                             int N, const unsigned char* __restrict__ c0,
                                                             const unsigned char* __restrict__ chg,
                                                             const unsigned char* __restrict__ need,
                                                             unsigned char* __restrict__ pend,
                                                             unsigned char* __restrict__ chg_next,
                                                             unsigned char* __restrict__ tf, int32_t* __restrict__ edge_total,
                                                             int first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int deg = 0, t = 0;
    if (i < N) {
        const int e0 = nptr[i], e1 = nptr[i + 1];
        int out = first | c0[i];
        if (chg && !out) {
            out = chg[i];
            for (int e = e0; e < e1 && !out; ++e) out = chg[src[e]];
        }
        const int p = first | pend[i] | out;
        t = (need ? need[i] : 1) & p;
        pend[i] = (unsigned char)(p & !t);
        chg_next[i] = (unsigned char)out;
        tf[i] = (unsigned char)t;
        deg = t ? e1 - e0 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) deg += __shfl_xor(deg, o);
    if ((threadIdx.x & 63) == 0 && deg) atomicAdd(edge_total, deg);
}

__global__ void adf_inc_mark_kernel(const int32_t* __restrict__ idx, int n, unsigned char* __restrict__ need) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) need[idx[i]] = 1;
}

// need_prev = need + in-neighbours of need   (need_prev pre-zeroed; all writers store 1)
__global__ void adf_inc_need_kernel(const int32_t* __restrict__ nptr, const int32_t* __restrict__ src, int N,
                                    const unsigned char* __restrict__ need, unsigned char* __restrict__ need_prev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || !need[i]) return;
    need_prev[i] = 1;
    for (int e = nptr[i]; e < nptr[i + 1]; ++e) need_prev[src[e]] = 1;
}

__global__ void adf_inc_scatter_kernel(const float4* __restrict__ src, const int32_t* __restrict__ idx, long long total,
                                       int w4, float4* __restrict__ dst, const int32_t* __restrict__ n_dev) {
    if (n_dev) total = min(total, (long long)*n_dev * w4);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / w4;
        const int c = (int)(t - r * w4);
        dst[(long long)idx[r] * w4 + c] = src[t];
    }
}

__global__ void adf_inc_gather_kernel(const float4* __restrict__ src, const int32_t* __restrict__ idx, long long total,
                                      int w4, float4* __restrict__ dst) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / w4;
        const int c = (int)(t - r * w4);
        dst[t] = src[(long long)idx[r] * w4 + c];
    }
}

size_t adf_inc_temp_bytes(int64_t n) {
    size_t bytes = 0;
    hipcub::CountingInputIterator<int32_t> it(0);
    (void)hipcub::DeviceSelect::Flagged(nullptr, bytes, it, (const unsigned char*)nullptr, (int32_t*)nullptr,
                                        (int32_t*)nullptr, (int)n);
    return bytes;
}

int32_t adf_inc_compare(adf_painn* h, int N, hipStream_t s) {
    hipLaunchKernelGGL(adf_inc_compare_kernel, dim3((N + 3) / 4), dim3(256), 0, s, h->nptr, h->e_src, h->e_geom,
                       h->prev_nptr, h->prev_src, h->prev_geom, N, h->inc_c0);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t adf_inc_need_from_list(adf_painn* h, int N, int L, const int32_t* out_idx, int n_out, hipStream_t s) {
    const size_t cap = (size_t)h->inc_capN;
    ADF_HIP_CHECK(hipMemsetAsync(h->inc_need, 0, cap * L, s));
    if (n_out > 0)
        hipLaunchKernelGGL(adf_inc_mark_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, out_idx, n_out,
                           h->inc_need + cap * (L - 1));
    for (int l = L - 1; l > 0; --l)
        hipLaunchKernelGGL(adf_inc_need_kernel, dim3((N + 255) / 256), dim3(256), 0, s, h->nptr, h->e_src, N,
                           h->inc_need + cap * l, h->inc_need + cap * (l - 1));
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// flags + compacted list of layer l.  chg ping-pong: layer l reads half (l & 1), writes half ((l + 1) & 1)
int32_t adf_inc_plan_layer(adf_painn* h, int l, int N, bool first, bool have_need, hipStream_t s) {
    const size_t cap = (size_t)h->inc_capN;
    const int L = h->inc_layers;
    unsigned char* tf = h->inc_tf + cap * l;
    hipLaunchKernelGGL(adf_inc_flags_kernel, dim3((N + 255) / 256), dim3(256), 0, s, h->nptr, h->e_src, N, h->inc_c0,
                       l == 0 ? (const unsigned char*)nullptr : h->inc_chg + cap * (l & 1),
                       have_need ? h->inc_need + cap * l : (const unsigned char*)nullptr, h->inc_pend + cap * l,
                       h->inc_chg + cap * ((l + 1) & 1), tf, h->inc_cnt + L + l, first ? 1 : 0);
    ADF_HIP_CHECK(hipGetLastError());
    hipcub::CountingInputIterator<int32_t> it(0);
    size_t bytes = h->inc_tmp_bytes;
    ADF_HIP_CHECK(hipcub::DeviceSelect::Flagged(h->inc_tmp, bytes, it, tf, h->inc_list + cap * l, h->inc_cnt + l, N, s));
    return ADF_OK;
}

static int rows_grid(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 65536 ? 65536 : (g < 1 ? 1 : g));
}

int32_t adf_inc_scatter_rows(const float* src, const int32_t* idx, int n, int width, float* dst, hipStream_t s,
                             const int32_t* n_dev) {
    if (n <= 0) return ADF_OK;
    const long long total = (long long)n * (width / 4);
    hipLaunchKernelGGL(adf_inc_scatter_kernel, dim3(rows_grid(total)), dim3(256), 0, s, (const float4*)src, idx, total,
                       width / 4, (float4*)dst, n_dev);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t adf_inc_gather_rows(const float* src, const int32_t* idx, int n, int width, float* dst, hipStream_t s) {
    if (n <= 0) return ADF_OK;
    const long long total = (long long)n * (width / 4);
    hipLaunchKernelGGL(adf_inc_gather_kernel, dim3(rows_grid(total)), dim3(256), 0, s, (const float4*)src, idx, total,
                       width / 4, (float4*)dst);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
