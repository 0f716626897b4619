// Edge- and node-side kernels of the EquiformerV2 denoiser (everything that is not a plain dense product).
//
// Reference (adsorbdiff/models/equiformer_v2/): edge frames edge_rot_mat.py:6-63, Wigner rows so3.py:493-531 +
// wigner.py:16-40, rotate in / out so3.py:493-506, SO(2) convolution bookkeeping so2_ops.py:158-262, separable S2
// activation activation.py:155-202, attention weights transformer_block.py:312-345, norm layer_norm.py:129-250,
// edge-degree embedding input_block.py:84-138, grid MLP transformer_block.py:497-531.
//
// Layouts.  Node features [N, S, C] degree-major like the reference (S = (L+1)^2).  Edges are grouped by TARGET atom
// (CSR eptr[N+1]); an edge keeps, for every degree l, only the rows |m'| <= min(l, M) of its Wigner matrix D_l (the
// rows an SO(2) convolution reads and the columns the rotation back needs): DR floats per edge, degree l at d_off[l],
// row-major [2 min(l,M)+1][2l+1].  In an edge's frame the |m| <= M coefficients are kept order-major ("m-major"):
// m = 0 for l = 0..L, then per m the +m entries for l = m..L and the -m entries; the operands of the per-order dense
// maps are separate buffers: order 0 [E, (L+1) c], order m >= 1 [2E, (L-m+1) c] with row 2e = +m, row 2e+1 = -m.
// A chunk of targets [n0, n1) is processed at a time; its edges are [eptr[n0], eptr[n1]) and every edge buffer is
// indexed by e - eptr[n0].
#include <hipcub/hipcub.hpp>

#include <stdlib.h>

#include "eqv2.h"
#include "graph.h"

#define EQ_FOR_L(L_, MACRO) \
    switch (L_) {           \
        case 1: MACRO(1); break; \
        case 2: MACRO(2); break; \
        case 3: MACRO(3); break; \
        case 4: MACRO(4); break; \
        case 5: MACRO(5); break; \
        case 6: MACRO(6); break; \
        default: adf_set_error("eqv2: lmax %d not instantiated (1..6)", L_); return ADF_EINVAL; \
    }

__device__ __forceinline__ float eq_silu(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ float eq_pow2_lift(float mx) {
    // 2^e such that mx 2^e in [2^14, 2^15); 1 for mx == 0 or non-finite
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(mx, &e);
    e = 15 - e;
    e = e > 120 ? 120 : (e < -120 ? -120 : e);
    return ldexpf(1.0f, e);
}

__device__ __forceinline__ float eq_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// sum over a workgroup of up to 1024 threads; every thread gets the result
__device__ __forceinline__ float eq_block_sum(float v, float* red) {
    v = eq_wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// ------------------------------------------------------------------------------------------------ graph -> edges
__global__ void eq_edges_kernel(GraphParams p, const int32_t* __restrict__ eptr, int32_t* __restrict__ e_src,
                                int32_t* __restrict__ e_dst, float* __restrict__ e_vec, long long capE, int32_t* flags) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(t / p.K);
    const int k = (int)(t - (long long)i * p.K);
    if (i >= p.N || k >= p.nbr_cnt[i]) return;
    const int j = p.nbr_src[(size_t)i * p.K + k];
    const int c = p.nbr_shift[(size_t)i * p.K + k];
    float sa, sb, sc;
    decode_shift(c, p.r0, p.r1, p.r2, sa, sb, sc);
    const float* cl = p.cell + 9 * p.batch[i];
    // shift (row vector) . cell, then pos[j] - pos[i] + offset   (utils.py:529-533)
    const float ox = __fadd_rn(__fadd_rn(__fmul_rn(sa, cl[0]), __fmul_rn(sb, cl[3])), __fmul_rn(sc, cl[6]));
    const float oy = __fadd_rn(__fadd_rn(__fmul_rn(sa, cl[1]), __fmul_rn(sb, cl[4])), __fmul_rn(sc, cl[7]));
    const float oz = __fadd_rn(__fadd_rn(__fmul_rn(sa, cl[2]), __fmul_rn(sb, cl[5])), __fmul_rn(sc, cl[8]));
    const long long e = (long long)eptr[i] + k;
    if (e >= capE) { atomicExch(&flags[2], 1); return; }
    e_src[e] = j;
    e_dst[e] = i;
    e_vec[3 * e] = __fadd_rn(__fsub_rn(p.pos[3 * j], p.pos[3 * i]), ox);
    e_vec[3 * e + 1] = __fadd_rn(__fsub_rn(p.pos[3 * j + 1], p.pos[3 * i + 1]), oy);
    e_vec[3 * e + 2] = __fadd_rn(__fsub_rn(p.pos[3 * j + 2], p.pos[3 * i + 2]), oz);
}

__global__ void eq_flag_images_kernel(const int32_t* img_cnt, int B, int32_t* flags) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B && img_cnt[t] == 0) atomicExch(&flags[1], 1);
}

int32_t eq_launch_edges_from_topk(adf_eqv2* h, const adf_batch* b, hipStream_t s) {
    const int N = b->num_atoms, B = b->num_systems, K = h->hp.max_neighbors;
    GraphParams p;
    p.pos = b->pos; p.cell = b->cell; p.batch = b->batch; p.atom_offset = b->atom_offset;
    p.r0 = b->reps[0]; p.r1 = b->reps[1]; p.r2 = b->reps[2];
    p.rc2 = h->hp.max_radius * h->hp.max_radius;
    p.K = K; p.N = N;
    p.nbr_cnt = h->nbr_cnt; p.nbr_src = h->nbr_src; p.nbr_shift = h->nbr_shift;
    p.img_cnt = h->img_cnt; p.flags = h->flags;
    p.moving = h->moving; p.mov_idx = h->mov_idx; p.mov_off = h->mov_off;
    p.cache_d2 = h->cache_d2; p.cache_cid = h->cache_cid; p.cache_cnt = h->cache_cnt;
    ADF_HIP_CHECK(hipMemsetAsync(h->img_cnt, 0, sizeof(int32_t) * B, s));
    int mode = 0;
    if (h->moving) { mode = h->cache_valid ? 2 : 1; h->cache_valid = true; }
    ADF_TRY(adf_topk_launch(p, mode, s));
    size_t tmp = h->scan_tmp_bytes;
    // eptr[i] = sum_{i' < i} nbr_cnt[i'], eptr[N] = E   (nbr_cnt[N] is kept 0)
    ADF_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(h->scan_tmp, tmp, h->nbr_cnt, h->eptr, N + 1, s));
    hipLaunchKernelGGL(eq_flag_images_kernel, dim3((B + 255) / 256), dim3(256), 0, s, h->img_cnt, B, h->flags);
    const long long slots = (long long)N * K;
    hipLaunchKernelGGL(eq_edges_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, p, h->eptr, h->e_src,
                       h->e_dst, h->e_vec, (long long)h->capE, h->flags);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// atomic numbers must index the embedding tables [0, max_num_elements) and the radius table [0, 100]
__global__ void eq_check_z_kernel(const int32_t* __restrict__ Z, int N, int max_elem, int32_t* flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && (Z[i] < 0 || Z[i] >= max_elem || Z[i] > 100)) atomicExch(&flags[4], 1);
}

int32_t eq_launch_check_z(const adf_eqv2* h, const int32_t* Z, int N, hipStream_t s) {
    hipLaunchKernelGGL(eq_check_z_kernel, dim3((N + 255) / 256), dim3(256), 0, s, Z, N, h->hp.max_num_elements, h->flags);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// eptr from a target-sorted edge list: eptr[i] = first edge with dst >= i
__global__ void eq_eptr_kernel(const int32_t* __restrict__ dst, long long E, int N, int32_t* __restrict__ eptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    long long lo = 0, hi = E;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (dst[mid] < i) lo = mid + 1; else hi = mid;
    }
    eptr[i] = (int32_t)lo;
}

int32_t eq_launch_eptr_from_dst(adf_eqv2* h, int N, int64_t E, hipStream_t s) {
    hipLaunchKernelGGL(eq_eptr_kernel, dim3((N + 256) / 256), dim3(256), 0, s, h->e_dst, (long long)E, N, h->eptr);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ Wigner rows
// One wave per edge.  Frame: R = Rx(b) Ry(g) with Ry(g) turning the edge direction n into the y-z plane and Rx(b)
// turning it onto +y, i.e. (cos g, sin g) = (n_z, -n_x) / rho, (cos b, sin b) = (n_y, -rho), rho = |(n_x, n_z)| — one
// of the frames "second row = edge direction" of edge_rot_mat.py:51-63 (the reference rolls it at random about the
// edge; the outputs do not depend on the roll).  D_l(R) = J_l Z_l(b) J_l Z_l(g) (wigner.py:16-40 with alpha = 0), Z the
// rotation about y: cos(f_i t) on the diagonal, sin(f_i t) on the anti-diagonal, f_i = l - i.  cos / sin of the
// multiples come from the Chebyshev recurrence on (cos, sin) of the frame itself: no inverse trigonometric function.
__global__ __launch_bounds__(256) void eq_wigner_kernel(const float* __restrict__ e_vec, const int32_t* __restrict__ eptr,
                                                        int N, const float* __restrict__ jd, eq_dims d,
                                                        float* __restrict__ wig) {
    __shared__ float Jl[1024];
    __shared__ float trig[4][4][EQ_MAX_L + 2];
    extern __shared__ float Pdyn[];  // [4][(2M+1) * (2L+1)]: rows l-ml..l+ml of J Z(b) J
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nJ = d.j_off[d.L + 1];
    for (int t = threadIdx.x; t < nJ; t += 256) Jl[t] = jd[t];
    __syncthreads();
    const long long e = (long long)blockIdx.x * 4 + w;
    const long long E = eptr[N];
    if (e >= E) return;
    float* P = Pdyn + w * ((2 * d.M + 1) * (2 * d.L + 1));
    float nx = e_vec[3 * e], ny = e_vec[3 * e + 1], nz = e_vec[3 * e + 2];
    const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
    nx *= inv; ny *= inv; nz *= inv;
    const float rho = sqrtf(nx * nx + nz * nz);
    float cg = 1.f, sg = 0.f;
    if (rho > 1e-20f) { cg = nz / rho; sg = -nx / rho; }
    const float cb = ny, sb = -rho;
    if (lane <= d.L) {
        float c0 = 1.f, s0 = 0.f, c1 = cb, s1 = sb, g0 = 1.f, h0 = 0.f, g1 = cg, h1 = sg;
        for (int k = 1; k < lane; ++k) {
            const float c2 = 2.f * cb * c1 - c0, s2 = 2.f * cb * s1 - s0;
            c0 = c1; s0 = s1; c1 = c2; s1 = s2;
            const float g2 = 2.f * cg * g1 - g0, h2 = 2.f * cg * h1 - h0;
            g0 = g1; h0 = h1; g1 = g2; h1 = h2;
        }
        trig[w][0][lane] = lane == 0 ? 1.f : c1;
        trig[w][1][lane] = lane == 0 ? 0.f : s1;
        trig[w][2][lane] = lane == 0 ? 1.f : g1;
        trig[w][3][lane] = lane == 0 ? 0.f : h1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* out = wig + (size_t)e * d.DR;
    for (int l = 0; l <= d.L; ++l) {
        const int n = 2 * l + 1, ml = min(l, d.M), nr = 2 * ml + 1;
        const float* J = Jl + d.j_off[l];
        for (int t = lane; t < nr * n; t += 64) {
            const int ri = t / n, dc = t - ri * n;
            const int i = l - ml + ri;
            float acc = 0.f;
            for (int b = 0; b < n; ++b) {
                const int f = l - b, af = f < 0 ? -f : f;
                const float cbf = trig[w][0][af], sbf = f < 0 ? -trig[w][1][af] : trig[w][1][af];
                acc += J[i * n + b] * (cbf * J[b * n + dc] + sbf * J[(n - 1 - b) * n + dc]);
            }
            P[t] = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < nr * n; t += 64) {
            const int ri = t / n, j = t - ri * n;
            const int f = l - j, af = f < 0 ? -f : f;
            const float cgf = trig[w][2][af], sgf = f < 0 ? -trig[w][3][af] : trig[w][3][af];
            out[d.d_off[l] + t] = P[ri * n + j] * cgf - P[ri * n + (n - 1 - j)] * sgf;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static inline long long eq_edge_bound(const adf_eqv2* h, long long nodes) {
    return h->ext_graph ? nodes * (long long)h->maxdeg : nodes * (long long)h->hp.max_neighbors;
}

int32_t eq_launch_wigner(adf_eqv2* h, int N, hipStream_t s) {
    const long long Eub = h->ext_graph ? h->E_ext : (long long)N * h->hp.max_neighbors;
    if (Eub <= 0) return ADF_OK;
    const eq_dims& d = h->d;
    const size_t dyn = sizeof(float) * 4 * (2 * d.M + 1) * (2 * d.L + 1);
    hipLaunchKernelGGL(eq_wigner_kernel, dim3((unsigned)((Eub + 3) / 4)), dim3(256), dyn, s, h->e_vec, h->eptr, N, h->jd,
                       d, h->wig);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ norm
// layer_norm.py:129-250: LayerNorm over the channels of l = 0; for l > 0 one scale per node from the degree-balanced
// mean square (every degree weighs 1/L, every order of a degree 1/(2l+1)), times a per-degree, per-channel weight.
__global__ void eq_norm_kernel(const float* __restrict__ x, const float* __restrict__ aff, const float* __restrict__ w0,
                               const float* __restrict__ b0, float* __restrict__ y, int N, eq_dims d, float eps) {
    __shared__ float red[16];
    const int n = blockIdx.x, c = threadIdx.x;
    const bool on = c < d.C;
    const float* xr = x + (size_t)n * d.S * d.C;
    float* yr = y + (size_t)n * d.S * d.C;
    const float v0 = on ? xr[c] : 0.f;
    const float mean = eq_block_sum(v0, red) / d.C;
    const float dv = on ? v0 - mean : 0.f;
    const float var = eq_block_sum(dv * dv, red) / d.C;
    if (on) yr[c] = dv * rsqrtf(var + eps) * w0[c] + b0[c];
    float q = 0.f;
    if (on) {
        for (int l = 1; l <= d.L; ++l) {
            const float wl = 1.0f / (float)(2 * l + 1) / (float)d.L;
            float a = 0.f;
            for (int s = l * l; s < (l + 1) * (l + 1); ++s) { const float t = xr[(size_t)s * d.C + c]; a += t * t; }
            q += a * wl;
        }
    }
    const float ms = eq_block_sum(q, red) / d.C;
    const float sc = 1.0f / sqrtf(ms + eps);
    if (on) {
        for (int l = 1; l <= d.L; ++l) {
            const float f = sc * aff[(size_t)(l - 1) * d.C + c];
            for (int s = l * l; s < (l + 1) * (l + 1); ++s) yr[(size_t)s * d.C + c] = xr[(size_t)s * d.C + c] * f;
        }
    }
}

int32_t eq_launch_norm(const adf_eqv2* h, const eq_norm* nm, const float* x, float* y, int N, hipStream_t s) {
    const int bd = (h->d.C + 63) / 64 * 64;
    hipLaunchKernelGGL(eq_norm_kernel, dim3(N), dim3(bd), 0, s, x, nm->affine, nm->l0_w, nm->l0_b, y, N, h->d, 1e-5f);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ radial MLP, first layer
// RadialFunction.net[0] on [Gaussian basis of (d - r[Z_src] - r[Z_dst]) | source embedding | target embedding]
// (transformer_block.py:247-262, equiformer_v2_denoising.py:208-213, 267).  Only basis functions that do not underflow
// to zero are visited (GaussianSmearing: exp(coeff (x - mu_k)^2), coeff = -0.5 / (2 delta)^2, equiformer_v2_oc20.py:
// 41-62); with the reference's radii (tabulated in pm, subtracted from Angstrom) the window is empty on every edge.
// w0t: [NB + 2 EC, EC] (transposed at set_weights so that the lanes of a wave read consecutive outputs).
__global__ void eq_radial_pre_kernel(const float* __restrict__ e_vec, const int32_t* __restrict__ e_src,
                                     const int32_t* __restrict__ e_dst, const int32_t* __restrict__ eptr,
                                     const int32_t* __restrict__ Z, const float* __restrict__ radii,
                                     const float* __restrict__ w0t, const float* __restrict__ b0,
                                     const float* __restrict__ semb, const float* __restrict__ temb, int n0, int n1,
                                     int EC, int NB, float rc, int max_elem, float* __restrict__ out, int32_t* flags) {
    const long long e0 = eptr[n0];
    const long long e = e0 + blockIdx.x;
    if (e >= eptr[n1]) return;
    const int j = threadIdx.x;
    const int zs = Z[e_src[e]], zt = Z[e_dst[e]];
    if (zs < 0 || zs >= max_elem || zt < 0 || zt >= max_elem || zs > 100 || zt > 100) {
        if (j == 0) atomicExch(&flags[4], 1);
        return;
    }
    const float vx = e_vec[3 * e], vy = e_vec[3 * e + 1], vz = e_vec[3 * e + 2];
    const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
    const float dd = dist - radii[zs] - radii[zt];
    if (j >= EC) return;
    float acc = b0[j];
    const float delta = rc / (float)(NB - 1);
    const float coeff = -0.5f / ((2.0f * delta) * (2.0f * delta));
    if (!(dd == dd)) {
        acc = dd;  // NaN radius (elements without a tabulated value): the reference's output is NaN too
    } else {
        const float tmax = sqrtf(104.0f / -coeff);
        int k0 = (int)ceilf((dd - tmax) / delta), k1 = (int)floorf((dd + tmax) / delta);
        k0 = max(k0, 0); k1 = min(k1, NB - 1);
        for (int k = k0; k <= k1; ++k) {
            const float t = dd - delta * (float)k;
            acc += expf(coeff * t * t) * w0t[(size_t)k * EC + j];
        }
    }
    const float* se = semb + (size_t)zs * EC;
    const float* te = temb + (size_t)zt * EC;
    const float* ws = w0t + (size_t)NB * EC;
    const float* wt = ws + (size_t)EC * EC;
    for (int q = 0; q < EC; ++q) acc += se[q] * ws[(size_t)q * EC + j];
    for (int q = 0; q < EC; ++q) acc += te[q] * wt[(size_t)q * EC + j];
    out[(size_t)(e - e0) * EC + j] = acc;
}

// The same first layer on a TABLE of element pairs instead of edges, for models whose Gaussian window is empty for
// every pair whatever the distance (decided from the radii table alone at set_weights: with the reference's pm-valued
// radii it always is): the radial MLP is then a function of (Z_src, Z_tgt) only and is evaluated once per weight
// binding; row = Z_src * NE + Z_tgt.  A pair with a non-finite radius gives NaN, as the per-edge evaluation does.
__global__ void eq_radial_pre_pairs_kernel(const float* __restrict__ radii, const float* __restrict__ w0t,
                                           const float* __restrict__ b0, const float* __restrict__ semb,
                                           const float* __restrict__ temb, int EC, int NB, int NE, float* __restrict__ out) {
    const int pair = blockIdx.x, j = threadIdx.x;
    const int zs = pair / NE, zt = pair - zs * NE;
    if (j >= EC) return;
    const float rr = (zs <= 100 ? radii[zs] : 0.f) + (zt <= 100 ? radii[zt] : 0.f);
    float acc = b0[j];
    if (!(rr == rr)) acc = rr;
    const float* se = semb + (size_t)zs * EC;
    const float* te = temb + (size_t)zt * EC;
    const float* ws = w0t + (size_t)NB * EC;
    const float* wt = ws + (size_t)EC * EC;
    for (int q = 0; q < EC; ++q) acc += se[q] * ws[(size_t)q * EC + j];
    for (int q = 0; q < EC; ++q) acc += te[q] * wt[(size_t)q * EC + j];
    out[(size_t)pair * EC + j] = acc;
}

int32_t eq_launch_radial_pre_pairs(const adf_eqv2* h, const eq_radial* r, const float* src_emb, const float* dst_emb,
                                   float* out, hipStream_t s) {
    const int NE = h->hp.max_num_elements, EC = h->d.EC;
    hipLaunchKernelGGL(eq_radial_pre_pairs_kernel, dim3(NE * NE), dim3((EC + 63) / 64 * 64), 0, s, h->atom_radii, r->w0t,
                       r->l0.b, src_emb, dst_emb, EC, h->d.NB, NE, out);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t eq_launch_radial_pre(const adf_eqv2* h, const eq_radial* r, const float* src_emb, const float* dst_emb,
                             const int32_t* Z, int n0, int n1, float* out, int N, hipStream_t s) {
    const long long Eub = eq_edge_bound(h, n1 - n0);
    if (Eub <= 0) return ADF_OK;
    const int EC = h->d.EC;
    hipLaunchKernelGGL(eq_radial_pre_kernel, dim3((unsigned)Eub), dim3((EC + 63) / 64 * 64), 0, s, h->e_vec, h->e_src,
                       h->e_dst, h->eptr, Z, h->atom_radii, r->w0t, r->l0.b, src_emb, dst_emb, n0, n1, EC, h->d.NB,
                       h->hp.max_radius, h->hp.max_num_elements, out, h->flags);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// LayerNorm (biased variance, eps 1e-5) followed by SiLU on rows of `width` floats, in place; one wave per row
__global__ __launch_bounds__(256) void eq_ln_silu_kernel(float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, long long rows, int width) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float* xr = x + (size_t)r * width;
    float sum = 0.f;
    for (int c = lane; c < width; c += 64) sum += xr[c];
    const float mean = eq_wave_sum(sum) / width;
    float q = 0.f;
    for (int c = lane; c < width; c += 64) { const float t = xr[c] - mean; q += t * t; }
    const float rstd = rsqrtf(eq_wave_sum(q) / width + 1e-5f);
    for (int c = lane; c < width; c += 64) xr[c] = eq_silu((xr[c] - mean) * rstd * w[c] + b[c]);
}

int32_t eq_launch_ln_silu(float* x, const float* w, const float* b, long long rows, int width, hipStream_t s) {
    if (rows <= 0) return ADF_OK;
    hipLaunchKernelGGL(eq_ln_silu_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, w, b, rows, width);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ edge-degree embedding
// input_block.py:84-138: the radial MLP gives the m = 0 coefficients (l = 0..L) of every edge in its own frame; they are
// rotated back (row m' = 0 of D_l, with the m-truncation rescale), summed per target, divided by the average degree, and
// added to the element embedding on l = 0.  One workgroup per target, thread = channel.
template <int LT>
__global__ void eq_edge_degree_kernel(const float* __restrict__ m0, const float* __restrict__ wig,
                                      const int32_t* __restrict__ eptr, const int32_t* __restrict__ Z,
                                      const float* __restrict__ emb, int n0, int n1, eq_dims d, float inv_avg,
                                      int max_elem, float* __restrict__ x, int32_t* flags,
                                      const int32_t* __restrict__ e_src, int pair_ne) {
    const int n = n0 + blockIdx.x, c = threadIdx.x;
    if (n >= n1 || c >= d.C) return;
    const long long ebase = eptr[n0];
    float acc[(LT + 1) * (LT + 1)];
#pragma unroll
    for (int s = 0; s < (LT + 1) * (LT + 1); ++s) acc[s] = 0.f;
    for (long long e = eptr[n]; e < eptr[n + 1]; ++e) {
        const float* D = wig + (size_t)e * d.DR;
        long long mrow = e - ebase;
        if (pair_ne > 0) mrow = (long long)min(max(Z[e_src[e]], 0), pair_ne - 1) * pair_ne + min(max(Z[n], 0), pair_ne - 1);
        const float* mr = m0 + (size_t)mrow * ((LT + 1) * d.C);
#pragma unroll
        for (int l = 0; l <= LT; ++l) {
            const int ml = l < d.M ? l : d.M;
            const float v = mr[l * d.C + c] * d.resc[l];
            const float* Dr = D + d.d_off[l] + ml * (2 * l + 1);  // row m' = 0
#pragma unroll
            for (int m = 0; m < 2 * l + 1; ++m) acc[l * l + m] += Dr[m] * v;
        }
    }
    const int z = Z[n];
    if (z < 0 || z >= max_elem) { if (c == 0) atomicExch(&flags[4], 1); return; }
    float* xr = x + (size_t)n * d.S * d.C;
#pragma unroll
    for (int s = 0; s < (LT + 1) * (LT + 1); ++s) xr[(size_t)s * d.C + c] = acc[s] * inv_avg + (s == 0 ? emb[(size_t)z * d.C + c] : 0.f);
}

int32_t eq_launch_edge_degree(const adf_eqv2* h, const float* m0, const int32_t* Z, int pair_ne, int n0, int n1, float* x,
                              hipStream_t s) {
    if (n1 <= n0) return ADF_OK;
    const int bd = (h->d.C + 63) / 64 * 64;
#define EQ_ED(LT_) hipLaunchKernelGGL(eq_edge_degree_kernel<LT_>, dim3(n1 - n0), dim3(bd), 0, s, m0, h->wig, h->eptr, Z, \
                                      h->sphere_emb, n0, n1, h->d, 1.0f / h->hp.avg_degree, h->hp.max_num_elements, x, h->flags, \
                                      h->e_src, pair_ne)
    EQ_FOR_L(h->d.L, EQ_ED)
#undef EQ_ED
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ rotate in
// so3.py:493-499 + so2_ops.py:188-213: [x_src | x_tgt] rotated into the edge's frame (rows |m'| <= M of D_l), multiplied
// by the radial weights, written as the per-order operands of the first SO(2) convolution.  One workgroup per edge,
// thread = one of the 2C input channels.
struct eq_ptrs { float* p[EQ_MAX_M + 1]; };

// MT > 0: the order cut-off M as a compile-time constant (= d.M): which rows of D_l are kept is then known statically and
// every load of the edge - the 49 coefficients of the thread's channel and the 29 radial weights - is requested BEFORE the
// first product (round 5; the runtime-M form loads degree by degree and weight by weight: 7 + dependent round trips per edge
// in a kernel that lives for one edge).  MT = 0: M read from d.
template <int LT, bool PRESPLIT, int MT>
__global__ __launch_bounds__(256) void eq_rotate_in_kernel(const float* __restrict__ y, const float* __restrict__ rad, const float* __restrict__ wig,
                                    const int32_t* __restrict__ eptr, const int32_t* __restrict__ e_src,
                                    const int32_t* __restrict__ e_dst, int n0, int n1, eq_dims d, eq_ptrs mb, eq_ptrs rs,
                                    const int32_t* __restrict__ Z, int pair_ne) {
    // (Measured alternative: one workgroup per TARGET looping over its edges with the target's own features loaded once —
    // slower, 63 -> 82 ms per forward at 256 k edges: fewer, longer workgroups with two barriers per edge.)
    // (Also measured, no change either way: two adjacent channels per thread - 8-byte gathers, 4-byte fp16 stores, half the
    // memory instructions - 63.0 vs 63.6 ms; runs of 256...4096 consecutive edges per XCD for L2 locality of the gathered
    // features - 59.3...60.8 vs 59.8 ms.  PMC: memory unit stalled 42 % of the time, VALU busy 46 %, 3.9 TB/s: the kernel
    // waits on its chain of dependent loads (edge -> node index -> feature rows) at 4 workgroups per CU.)
    // PRESPLIT: the operand rows are written as fp16 hi / lo images lifted by the row's own power of two (what
    // eq_gemm16p_kernel stages by plain copies): mb.p[m] holds [rows][nm 2C] halves of hi followed by the same of lo
    __shared__ unsigned int smax[2 * EQ_MAX_M + 1];  // |.| maxima of the edge's operand rows (bit patterns order like floats)
    const long long ebase = eptr[n0];
    const long long e = ebase + blockIdx.x;
    if (e >= eptr[n1]) return;
    const int c = threadIdx.x, C2 = 2 * d.C;
    const bool on = c < C2;
    if (threadIdx.x < 2 * EQ_MAX_M + 1) smax[threadIdx.x] = 0u;
    __syncthreads();
    const long long el = e - ebase;
    const long long Erows = eptr[n1] - ebase;
    float rmx[2 * LT + 1];
#pragma unroll
    for (int t = 0; t < 2 * LT + 1; ++t) rmx[t] = 0.f;
    float vals[PRESPLIT ? (LT + 1) * (LT + 1) : 1];
    if (on) {
        const int node = c < d.C ? e_src[e] : e_dst[e];
        const float* yr = y + (size_t)node * d.S * d.C + (c < d.C ? c : c - d.C);
        const float* D = wig + (size_t)e * d.DR;
        long long rrow = el;
        if (pair_ne > 0) {  // atomic numbers are range-checked by eq_check_z_kernel; clamped here against stray reads
            const int zs = min(max(Z[e_src[e]], 0), pair_ne - 1), zt = min(max(Z[e_dst[e]], 0), pair_ne - 1);
            rrow = (long long)zs * pair_ne + zt;
        }
        const float* rr = rad + (size_t)rrow * d.RW * C2;
        constexpr int SS = (LT + 1) * (LT + 1);
        float vin[MT > 0 ? SS : 1], rw[MT > 0 ? SS : 1];
        if constexpr (MT > 0) {
#pragma unroll
            for (int t = 0; t < SS; ++t) vin[t] = yr[(size_t)t * d.C];
#pragma unroll
            for (int l = 0; l <= LT; ++l) {
#pragma unroll
                for (int mp = -l; mp <= l; ++mp) {
                    const int am = mp < 0 ? -mp : mp;
                    if (am > (l < MT ? l : MT)) continue;
                    rw[l * l + l + mp] = rr[(size_t)(d.rad_off[am] + (l - am)) * C2 + c];
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // every load of the edge is requested before the first product
        }
#pragma unroll
        for (int l = 0; l <= LT; ++l) {
            float v[2 * LT + 1];
#pragma unroll
            for (int m = 0; m < 2 * l + 1; ++m) v[m] = MT > 0 ? vin[l * l + m] : yr[(size_t)(l * l + m) * d.C];
            const int ml = MT > 0 ? (l < MT ? l : MT) : (l < d.M ? l : d.M);
            const float* Dl = D + d.d_off[l];
#pragma unroll
            for (int mp = -l; mp <= l; ++mp) {  // compile-time order: the row-maximum slot is a constant
                const int am = mp < 0 ? -mp : mp;
                if (am > ml) continue;
                const int ri = mp + ml;
                float a = 0.f;
#pragma unroll
                for (int m = 0; m < 2 * l + 1; ++m) a += Dl[ri * (2 * l + 1) + m] * v[m];
                const int nm = LT - am + 1;
                a *= MT > 0 ? rw[l * l + l + mp] : rr[(size_t)(d.rad_off[am] + (l - am)) * C2 + c];
                if (PRESPLIT) {
                    vals[l * l + l + mp] = a;
                } else {
                    const long long row = am == 0 ? el : 2 * el + (mp < 0 ? 1 : 0);
                    mb.p[am][(size_t)row * nm * C2 + (size_t)(l - am) * C2 + c] = a;
                }
                rmx[am == 0 ? 0 : 2 * am - 1 + (mp < 0 ? 1 : 0)] = fmaxf(rmx[am == 0 ? 0 : 2 * am - 1 + (mp < 0 ? 1 : 0)], fabsf(a));
            }
        }
    }
    if (rs.p[0]) {
        // row maxima over the workgroup: four DPP row rotations inside every 16 lanes, then one LDS atomic per row of lanes
        // (was: a 64-lane butterfly of six ds_bpermute round trips per operand row, one after the other behind the `break`s
        // of the run-time order cut-off - the five chains of M = 2 did not overlap)
        constexpr int NT = MT > 0 ? 2 * MT + 1 : 2 * LT + 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (MT == 0 && t >= 2 * d.M + 1) break;
            float v = rmx[t];
#define EQ_ROR(n_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n_), 0xf, 0xf, false))
            v = fmaxf(v, EQ_ROR(8));
            v = fmaxf(v, EQ_ROR(4));
            v = fmaxf(v, EQ_ROR(2));
            v = fmaxf(v, EQ_ROR(1));
#undef EQ_ROR
            if ((threadIdx.x & 15) == 0) atomicMax(&smax[t], __float_as_uint(v));
        }
    }
    if (!rs.p[0]) return;
    __syncthreads();
    if (threadIdx.x < 2 * d.M + 1) {
        const int t = threadIdx.x, m = (t + 1) >> 1;
        const long long row = m == 0 ? el : 2 * el + ((t + 1) & 1);
        rs.p[m][row] = __uint_as_float(smax[t]);
    }
    if (PRESPLIT && on) {
        (void)Erows;
#pragma unroll
        for (int l = 0; l <= LT; ++l) {
            const int ml = l < d.M ? l : d.M;
#pragma unroll
            for (int mp = -l; mp <= l; ++mp) {
                const int am = mp < 0 ? -mp : mp;
                if (am > ml) continue;
                const int nm = LT - am + 1, sg = mp < 0 ? 1 : 0;
                const float lift = eq_pow2_lift(__uint_as_float(smax[am == 0 ? 0 : 2 * am - 1 + sg]));
                const float sv = vals[l * l + l + mp] * lift;
                const _Float16 hh = (_Float16)sv;
                const long long row = am == 0 ? el : 2 * el + sg;
                const size_t rows_total = (size_t)(am == 0 ? 1 : 2) * (size_t)d.presplit_rows;
                _Float16* hi = reinterpret_cast<_Float16*>(mb.p[am]);
                const size_t off = (size_t)row * nm * C2 + (size_t)(l - am) * C2 + c;
                hi[off] = hh;
                hi[rows_total * nm * C2 + off] = (_Float16)(sv - (float)hh);
            }
        }
    }
}

int32_t eq_launch_rotate_in(const adf_eqv2* h, const float* y, const float* rad, const int32_t* Z, int pair_ne, int n0, int n1,
                            float* const* mbuf, float* const* rsp, bool presplit, hipStream_t s) {
    const long long Eub = eq_edge_bound(h, n1 - n0);
    if (Eub <= 0) return ADF_OK;
    if (presplit && !rsp) { adf_set_error("rotate_in: pre-split output needs the row magnitudes"); return ADF_EINVAL; }
    eq_dims dd = h->d;
    dd.presplit_rows = (int)Eub;  // rows of the order-0 operand buffer: the lo image starts after rows * width halves
    eq_ptrs mb, rs;
    for (int m = 0; m <= EQ_MAX_M; ++m) { mb.p[m] = m <= h->d.M ? mbuf[m] : nullptr; rs.p[m] = (rsp && m <= h->d.M) ? rsp[m] : nullptr; }
    const int bd = (2 * h->d.C + 63) / 64 * 64;
    if (bd > 256) { adf_set_error("rotate_in: more than 128 sphere channels (one thread per input channel, 256-thread workgroups; INTEGRATION.md 5)"); return ADF_EINVAL; }
    static const bool generic_ri = getenv("ADF_EQV2_ROTIN_GENERIC") != nullptr;   // read once, not per launch
    const bool m2 = h->d.M == 2 && !generic_ri;   // compile-time-M fast path (M = 2: the shipped models)
#define EQ_RI_(LT_, MT_)                                                                                               \
    if (presplit)                                                                                                      \
        hipLaunchKernelGGL((eq_rotate_in_kernel<LT_, true, MT_>), dim3((unsigned)Eub), dim3(bd), 0, s, y, rad, h->wig, h->eptr, \
                           h->e_src, h->e_dst, n0, n1, dd, mb, rs, Z, pair_ne);                                       \
    else                                                                                                               \
        hipLaunchKernelGGL((eq_rotate_in_kernel<LT_, false, MT_>), dim3((unsigned)Eub), dim3(bd), 0, s, y, rad, h->wig, h->eptr, \
                           h->e_src, h->e_dst, n0, n1, dd, mb, rs, Z, pair_ne)
#define EQ_RI(LT_) if (m2) { EQ_RI_(LT_, 2); } else { EQ_RI_(LT_, 0); }
    EQ_FOR_L(h->d.L, EQ_RI)
#undef EQ_RI
#undef EQ_RI_
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// value of the m-major reduced coefficient r of an SO(2) convolution's output (so2_ops.py:52-66: the +m / -m pair mixes
// like a complex product): y0 [E, ld0] (order 0 at column off0), ym[m] [2E, 2 nm cw]
__device__ __forceinline__ float eq_conv_out(const float* __restrict__ y0, int ld0, int off0, const eq_ptrs& ym,
                                             const eq_dims& d, long long el, int r, int cw, int c) {
    const int m = d.r_m[r], l = d.r_l[r];
    if (m == 0) return y0[(size_t)el * ld0 + off0 + l * cw + c];
    const int nm = d.L - m + 1, half = nm * cw, W = 2 * half, q = (l - m) * cw + c;
    const float* a = ym.p[m] + (size_t)(2 * el) * W;
    if (d.r_sgn[r] == 0) return a[q] - a[W + half + q];
    return a[W + q] + a[half + q];
}

// ------------------------------------------------------------------------------------------------ separable S2 activation
// activation.py:176-202 on the hidden message of an attention block: the l = 0 output is SiLU of the extra scalar gate,
// the l > 0 outputs go through the S2 grid: to_grid (res^2 x S_r), point-wise SiLU, from_grid.  First version: VALU,
// thread = (edge, hidden channel), coefficients in registers, both matrices in LDS.
template <int SRT>
__global__ __launch_bounds__(256) void eq_s2act_kernel(const float* __restrict__ y0, int ld0, int off0, int gate_off,
                                                       eq_ptrs ym, const int32_t* __restrict__ eptr, int n0, int n1,
                                                       eq_dims d, const float* __restrict__ tg, const float* __restrict__ fg,
                                                       eq_ptrs mb, int epb) {
    extern __shared__ float lds[];  // T [G][SRT], F [G][SRT]
    float* T = lds;
    float* Fm = lds + (size_t)d.G * SRT;
    for (int t = threadIdx.x; t < d.G * SRT; t += blockDim.x) {
        const int p = t / SRT, r = t - p * SRT;
        T[t] = r < d.Sr ? tg[(size_t)p * d.Sr + r] : 0.f;
        Fm[t] = r < d.Sr ? fg[(size_t)p * d.Sr + r] : 0.f;
    }
    __syncthreads();
    const long long ebase = eptr[n0];
    const int sub = threadIdx.x / d.Hd, c = threadIdx.x - sub * d.Hd;
    const long long el = (long long)blockIdx.x * epb + sub;
    if (sub >= epb || ebase + el >= eptr[n1]) return;
    float in[SRT], out[SRT];
#pragma unroll
    for (int r = 0; r < SRT; ++r) {
        in[r] = r < d.Sr ? eq_conv_out(y0, ld0, off0, ym, d, el, r, d.Hd, c) : 0.f;
        out[r] = 0.f;
    }
    for (int p = 0; p < d.G; ++p) {
        const float* tp = T + p * SRT;
        float g = 0.f;
#pragma unroll
        for (int r = 0; r < SRT; ++r) g += tp[r] * in[r];
        const float sv = eq_silu(g);
        const float* fp = Fm + p * SRT;
#pragma unroll
        for (int r = 0; r < SRT; ++r) out[r] += fp[r] * sv;
    }
    out[0] = eq_silu(y0[(size_t)el * ld0 + gate_off + c]);
#pragma unroll
    for (int r = 0; r < SRT; ++r) {
        if (r < d.Sr) {
            const int m = d.r_m[r], l = d.r_l[r], nm = d.L - m + 1;
            const long long row = m == 0 ? el : 2 * el + d.r_sgn[r];
            mb.p[m][(size_t)row * nm * d.Hd + (size_t)(l - m) * d.Hd + c] = out[r];
        }
    }
}

// ---- matrix-core version.  Per edge (one wave): G = T . in (M = grid point, K = coefficient, N = hidden channel),
// SiLU on the accumulators, out = F^T . silu(G) (M = coefficient, K = grid point, N = channel) with G's accumulator tile
// taken as the B operand of the second product without leaving the registers (its k order is the accumulator's row
// order, the constant operand is laid out to match).  f16x3 split products; the edge's inputs are lifted by a power
// of two chosen from the edge's own maximum (row-local: results do not depend on the batch), the SiLU outputs by
// that lift over a bound of the grid transform's gain.  Constant operands (fp16 hi/lo, fragment order) live in LDS.
typedef float eqf32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 eqhalf8 __attribute__((ext_vector_type(8)));


// Global-memory accesses through pointers the compiler cannot trace to a kernel argument (here: selected per lane) would be
// FLAT instructions, which count on the LDS counter as well: every wait for an LDS read then drains the loads in flight.
__device__ __forceinline__ float eq_ldg(const float* p) { return *(const __attribute__((address_space(1))) float*)p; }
__device__ __forceinline__ void eq_stg(float* p, float v) { *(__attribute__((address_space(1))) float*)p = v; }

// Round 6: the item's 32 input loads are issued back to back.  Before, every coefficient row was fetched behind a branch
// through pointers read from LDS: flat loads, each followed by `s_waitcnt vmcnt(0) lgkmcnt(0)` - 16 serial memory round
// trips per item (3.9 ms per launch at 256 k edges; the kernel had been filed as VALU-bound).  Now: descriptors are packed
// words in LDS (four per ds_read_b128), the order's row base is selected per lane among wave-uniform bases derived from the
// kernel arguments, rows past Sr read a valid address and are zeroed by a select, the stores go the same way.
//   input descriptor of coefficient r:  bits 0-2 m, 3-4 sign code (1: +1, 2: -1, else 0), 5-17 o1, 18-30 o2, 31 valid;
//     value = base_m[o1] + sign * base_m[o2], base_m = the edge's first row in the order-m buffer (m = 0: y0 + el * ld0)
//   output descriptor of coefficient r': bits 0-2 m, 3 valid, 4 row sign, 5-31 float offset from the edge's first output row
// MT: compile-time bound of the order cut-off M (length of the select chains).
// max of a non-negative value over the 64 lanes, left in every lane: four DPP row rotations inside each 16 lanes, then the four
// rows' results through scalar registers (a 64-lane `__shfl_xor` butterfly is six dependent ds_bpermute round trips)
__device__ __forceinline__ float eq_wave_max_nonneg(float v) {
#define EQ_ROR(n_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n_), 0xf, 0xf, false))
    v = fmaxf(v, EQ_ROR(8));
    v = fmaxf(v, EQ_ROR(4));
    v = fmaxf(v, EQ_ROR(2));
    v = fmaxf(v, EQ_ROR(1));
#undef EQ_ROR
    const unsigned int b = __float_as_uint(v);   // non-negative floats order like their bit patterns
    const unsigned int m01 = max((unsigned int)__builtin_amdgcn_readlane((int)b, 0), (unsigned int)__builtin_amdgcn_readlane((int)b, 16));
    const unsigned int m23 = max((unsigned int)__builtin_amdgcn_readlane((int)b, 32), (unsigned int)__builtin_amdgcn_readlane((int)b, 48));
    return __uint_as_float(max(m01, m23));
}

// fp16 hi / lo terms of p0 s and p1 s (s a power of two: the products are exact), packed in the MFMA operand order:
// hi = f16(p s), lo = f16(p s - hi), one v_fma_mix per term (the multiplication and the conversion in one instruction, the
// subtrahend read as the fp16 half it is).  hipcc's own code for `sv = p * s; hh = (_Float16)sv; ll = (_Float16)(sv - (float)hh)`
// forms hi twice (v_cvt_pk_f16_f32 for the operand, v_fma_mixlo for the residual) beside the multiply: 3.5 instructions per value.
__device__ __forceinline__ void eq_split_pair(float p0, float p1, float s, unsigned int& hi, unsigned int& lo) {
    unsigned int h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(p0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(p1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(p0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(p1), "v"(s), "v"(h));
    hi = h;
    lo = l;
}
// the same for the PRODUCTS u0 r0, u1 r1 (two different factors per value): hi = f16(u r), lo = f16(u r - hi) with the product
// formed inside the fused instruction
__device__ __forceinline__ void eq_split_pair_prod(float u0, float r0, float u1, float r1, unsigned int& hi, unsigned int& lo) {
    unsigned int h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(u0), "v"(r0));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(u1), "v"(r1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(u0), "v"(r0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(u1), "v"(r1), "v"(h));
    hi = h;
    lo = l;
}
typedef unsigned int equint4v __attribute__((ext_vector_type(4)));

template <int NBK, int MT>
__global__ __launch_bounds__(1024, 4) void eq_s2act_mfma_kernel(const float* __restrict__ y0, int ld0, int off0, int gate_off,
                                                               eq_ptrs ym, const int32_t* __restrict__ eptr, int n0, int n1,
                                                               const eq_dims* __restrict__ dg, int Sr, int L, int M, int Hd,
                                                               const eqhalf8* __restrict__ tabs, int npb, float inv_sT,
                                                               float inv_sF, float gain_shift, eq_ptrs mb, eq_ptrs rs) {
    extern __shared__ eqhalf8 tab[];  // TA [npb][2 ks][hi|lo][64], then FA likewise
    __shared__ __attribute__((aligned(16))) unsigned int idesc[32];   // input side, by coefficient r
    __shared__ __attribute__((aligned(16))) unsigned int odesc[32];   // output side, by coefficient r'
    __shared__ unsigned int wmag[16][2 * EQ_MAX_M + 2];  // per wave: magnitudes of the item's destination rows
    const int ntab = npb * 2 * 2 * 64;
    for (int t = threadIdx.x; t < 2 * ntab; t += 1024) tab[t] = tabs[t];
    if (threadIdx.x < 32) {
        const int r = threadIdx.x;
        unsigned int q = ((unsigned int)off0 << 5) | ((unsigned int)off0 << 18);   // not a coefficient: a valid address, zeroed
        unsigned int o = 0u;
        if (r < Sr) {
            const int m = dg->r_m[r], l = dg->r_l[r], sg = dg->r_sgn[r];
            const int nm = L - m + 1, half = nm * Hd, W = 2 * half, c0 = (l - m) * Hd;
            int o1, o2, code;
            if (m == 0) { o1 = off0 + l * Hd; o2 = o1; code = 0; }
            else if (sg == 0) { o1 = c0; o2 = W + half + c0; code = 2; }
            else { o1 = W + c0; o2 = half + c0; code = 1; }
            q = (unsigned int)m | ((unsigned int)code << 3) | ((unsigned int)o1 << 5) | ((unsigned int)o2 << 18) | 0x80000000u;
            o = (unsigned int)m | 8u | ((unsigned int)sg << 4) | ((unsigned int)((m == 0 ? 0 : sg * half) + c0) << 5);
        }
        idesc[r] = q;
        odesc[r] = o;
    }
    __syncthreads();
    const eqhalf8* TA = tab;
    const eqhalf8* FA = tab + ntab;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: the item's addresses stay scalar
    const int cl = lane & 31, kh = lane >> 5;
    const long long ebase = eptr[n0];
    const long long Ec = eptr[n1] - ebase;
    const bool emit = rs.p[0] != nullptr;
    // one wave per (edge, block of 32 hidden channels)
    // (Measured alternative: writing the outputs as the second convolution's pre-split operand rows - fp16 hi / lo lifted per
    // row, the two waves of an edge exchanging their row maxima through LDS - made that product 8 ms per forward faster at
    // 256 k edges and this kernel 9 ms slower: 600 more VALU instructions per item.  Not kept.)
    const int nblk = Hd >> 5;
    for (long long item = (long long)blockIdx.x * 16 + wave; item < Ec * nblk; item += (long long)gridDim.x * 16) {
        const long long el = item / nblk;
        const int cb = (int)(item - el * nblk) * 32;
        // the edge's first row per order, input and output side (wave-uniform)
        const float* bin[MT + 1];
        float* bout[MT + 1];
        bin[0] = y0 + (size_t)el * ld0;
        bout[0] = mb.p[0] + (size_t)el * (L + 1) * Hd;
#pragma unroll
        for (int m = 1; m <= MT; ++m) {
            const int nm = L - m + 1;
            bin[m] = ym.p[m <= M ? m : 0] + (size_t)(2 * el) * (2 * nm * Hd);
            bout[m] = mb.p[m <= M ? m : 0] + (size_t)(2 * el) * (nm * Hd);
        }
        eqhalf8 b1h[NBK][2], b1l[NBK][2];
        float lift;
        float gate[NBK];
        {
            uint4 dq[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) dq[t] = *reinterpret_cast<const uint4*>(&idesc[16 * (t >> 1) + 8 * kh + 4 * (t & 1)]);
            float va[NBK][2][8], vb[NBK][2][8];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint4 d4 = dq[2 * ks + (j >> 2)];
                    const unsigned int d = (j & 3) == 0 ? d4.x : ((j & 3) == 1 ? d4.y : ((j & 3) == 2 ? d4.z : d4.w));
                    const int m = (int)(d & 7u);
                    const float* bp = bin[0];
#pragma unroll
                    for (int mm = 1; mm <= MT; ++mm) bp = m == mm ? bin[mm] : bp;
                    const int o1 = (int)((d >> 5) & 0x1fffu), o2 = (int)((d >> 18) & 0x1fffu);
#pragma unroll
                    for (int nb = 0; nb < NBK; ++nb) {
                        const int c = cb + 32 * nb + cl;
                        va[nb][ks][j] = eq_ldg(bp + o1 + c);
                        vb[nb][ks][j] = eq_ldg(bp + o2 + c);
                    }
                }
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) gate[nb] = eq_ldg(bin[0] + gate_off + cb + 32 * nb + cl);
            float mx = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint4 d4 = dq[2 * ks + (j >> 2)];
                    const unsigned int d = (j & 3) == 0 ? d4.x : ((j & 3) == 1 ? d4.y : ((j & 3) == 2 ? d4.z : d4.w));
                    const unsigned int code = (d >> 3) & 3u;
                    const float sg = code == 1u ? 1.f : (code == 2u ? -1.f : 0.f);
#pragma unroll
                    for (int nb = 0; nb < NBK; ++nb) {
                        float v = va[nb][ks][j] + sg * vb[nb][ks][j];
                        v = (d & 0x80000000u) ? v : 0.f;
                        va[nb][ks][j] = v;
                        mx = fmaxf(mx, fabsf(v));
                    }
                }
            mx = eq_wave_max_nonneg(mx);
            lift = eq_pow2_lift(mx);
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    equint4v h4, l4;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        unsigned int hw, lw;
                        eq_split_pair(va[nb][ks][j], va[nb][ks][j + 1], lift, hw, lw);
                        h4[j >> 1] = hw;
                        l4[j >> 1] = lw;
                    }
                    b1h[nb][ks] = __builtin_bit_cast(eqhalf8, h4);
                    b1l[nb][ks] = __builtin_bit_cast(eqhalf8, l4);
                }
        }
        const float lift2 = lift * gain_shift;  // SiLU outputs: |silu(g)| <= |g| <= gain * max|in|
        eqf32x16 acc2[NBK];
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;
        const float sc1 = inv_sT / lift;
        const float c1e = -1.4426950408889634f * sc1, c2s = sc1 * lift2;   // (lift2 is a power of two: c2s is exact)
        for (int pb = 0; pb < npb; ++pb) {
            eqf32x16 acc1[NBK];
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[nb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const eqhalf8 ah = TA[((pb * 2 + ks) * 2 + 0) * 64 + lane];
                const eqhalf8 al = TA[((pb * 2 + ks) * 2 + 1) * 64 + lane];
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb) {
                    acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h[nb][ks], acc1[nb], 0, 0, 0);
                    acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l[nb][ks], acc1[nb], 0, 0, 0);
                    acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h[nb][ks], acc1[nb], 0, 0, 0);
                }
            }
            eqhalf8 b2h[NBK][2], b2l[NBK][2];
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    equint4v h4, l4;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        // silu(g) lift2 with g = acc sc1:  (acc sc1 lift2) / (1 + exp2(acc (-sc1 log2 e))) - both factors straight
                        // from the accumulator (7 vector instructions per value: 2 multiplies, exp2, add, rcp, 2 conversions)
                        const float a0 = acc1[nb][8 * ks + j], a1 = acc1[nb][8 * ks + j + 1];
                        const float r0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a0 * c1e));
                        const float r1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a1 * c1e));
                        unsigned int hw, lw;
                        eq_split_pair_prod(a0 * c2s, r0, a1 * c2s, r1, hw, lw);
                        h4[j >> 1] = hw;
                        l4[j >> 1] = lw;
                    }
                    b2h[nb][ks] = __builtin_bit_cast(eqhalf8, h4);
                    b2l[nb][ks] = __builtin_bit_cast(eqhalf8, l4);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const eqhalf8 fh = FA[((pb * 2 + ks) * 2 + 0) * 64 + lane];
                const eqhalf8 fl = FA[((pb * 2 + ks) * 2 + 1) * 64 + lane];
#pragma unroll
                for (int nb = 0; nb < NBK; ++nb) {
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, b2h[nb][ks], acc2[nb], 0, 0, 0);
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, b2l[nb][ks], acc2[nb], 0, 0, 0);
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, b2h[nb][ks], acc2[nb], 0, 0, 0);
                }
            }
        }
        const float sc2 = inv_sF / lift2;
        // outputs (row r' = (reg & 3) + 8 (reg >> 2) + 4 kh), the l = 0 row replaced by SiLU of the scalar gate
        if (emit && lane < 2 * EQ_MAX_M + 2) wmag[wave][lane] = 0u;
        uint4 oq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) oq[t] = *reinterpret_cast<const uint4*>(&odesc[8 * t + 4 * kh]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint4 o4 = oq[r >> 2];
            const unsigned int o = (r & 3) == 0 ? o4.x : ((r & 3) == 1 ? o4.y : ((r & 3) == 2 ? o4.z : o4.w));
            const int m = (int)(o & 7u);
            float* dp = bout[0];
#pragma unroll
            for (int mm = 1; mm <= MT; ++mm) dp = m == mm ? bout[mm] : dp;
            float* dst = dp + (o >> 5);
            float mg = 0.f;
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
                const int c = cb + 32 * nb + cl;
                float v = acc2[nb][r] * sc2;
                if (r == 0 && kh == 0) v = eq_silu(gate[nb]);
                if (o & 8u) eq_stg(dst + c, v);
                mg = fmaxf(mg, fabsf(v));
            }
            // magnitude of the destination row (zeroed by the launcher): the next product's power-of-two lift comes from it
            if (emit) {
#pragma unroll
                for (int sh = 16; sh > 0; sh >>= 1) mg = fmaxf(mg, __shfl_xor(mg, sh));
                if (cl == 0 && (o & 8u)) atomicMax(&wmag[wave][m == 0 ? 0 : 2 * m - 1 + (int)((o >> 4) & 1u)], __float_as_uint(mg));
            }
        }
        if (emit) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < 2 * M + 1) {
                const int m = (lane + 1) >> 1;
                const long long row = m == 0 ? el : 2 * el + ((lane + 1) & 1);
                float* mp = rs.p[0];
#pragma unroll
                for (int mm = 1; mm <= MT; ++mm) mp = m == mm ? rs.p[mm <= M ? mm : 0] : mp;
                atomicMax(reinterpret_cast<unsigned int*>(mp) + row, wmag[wave][lane]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

int32_t eq_launch_s2act(const adf_eqv2* h, const float* y0, float* const* ym, int extra, int gate_off, int n0, int n1,
                        float* const* mbp, float* const* rsp, bool* rs_written, hipStream_t s) {
    const long long Eub = eq_edge_bound(h, n1 - n0);
    if (Eub <= 0) return ADF_OK;
    const eq_dims& d = h->d;
    if (d.Hd > 256) { adf_set_error("eqv2: attn_hidden_channels > 256"); return ADF_EINVAL; }
    eq_ptrs a, b;
    for (int m = 0; m <= d.M; ++m) { a.p[m] = ym[m]; b.p[m] = mbp[m]; }
    const int ld0 = extra + (d.L + 1) * d.Hd;
    // (the packed descriptors of the matrix-core kernel hold 13-bit column offsets)
    if (!h->exact_f32 && h->s2tab && d.Sr <= 32 && d.Hd % 32 == 0 && ld0 <= 8192 && 4 * (d.L + 1) * d.Hd <= 8192) {
        eq_ptrs r;
        for (int m = 0; m <= EQ_MAX_M; ++m) r.p[m] = (rsp && m <= d.M) ? rsp[m] : nullptr;
        if (rsp)
            for (int m = 0; m <= d.M; ++m)
                ADF_HIP_CHECK(hipMemsetAsync(rsp[m], 0, sizeof(float) * (size_t)(m == 0 ? Eub : 2 * Eub), s));
        const size_t dyn = (size_t)h->s2_npb * 2 * 2 * 64 * 16 * 2;
#define EQ_S2(MT_)                                                                                                          \
    do {                                                                                                                    \
        ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&eq_s2act_mfma_kernel<1, MT_>),                     \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));                           \
        hipLaunchKernelGGL((eq_s2act_mfma_kernel<1, MT_>), dim3(h->num_cus), dim3(1024), dyn, s, y0, ld0, extra, gate_off, a, \
                           h->eptr, n0, n1, h->d_dev, d.Sr, d.L, d.M, d.Hd, (const eqhalf8*)h->s2tab, h->s2_npb,            \
                           h->s2_inv_sT, h->s2_inv_sF, h->s2_gain_shift, b, r);                                             \
    } while (0)
        if (d.M <= 2) EQ_S2(2); else EQ_S2(EQ_MAX_M);
#undef EQ_S2
        ADF_HIP_CHECK(hipGetLastError());
        *rs_written = rsp != nullptr;
        return ADF_OK;
    }
    *rs_written = false;
    const int epb = 256 / d.Hd;
    const unsigned grid = (unsigned)((Eub + epb - 1) / epb);
    if (d.Sr <= 32) {
        const size_t dyn = sizeof(float) * 2 * d.G * 32;
        ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&eq_s2act_kernel<32>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(eq_s2act_kernel<32>, dim3(grid), dim3(256), dyn, s, y0, ld0, extra, gate_off, a, h->eptr, n0, n1,
                           d, h->to_red, h->from_red, b, epb);
    } else if (d.Sr <= 49) {
        const size_t dyn = sizeof(float) * 2 * d.G * 49;
        ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&eq_s2act_kernel<49>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(eq_s2act_kernel<49>, dim3(grid), dim3(256), dyn, s, y0, ld0, extra, gate_off, a, h->eptr, n0, n1,
                           d, h->to_red, h->from_red, b, epb);
    } else {
        adf_set_error("eqv2: more than 49 reduced coefficients");
        return ADF_EINVAL;
    }
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ attention weights
// transformer_block.py:312-345: per head LayerNorm over the alpha channels, smooth leaky ReLU, dot with alpha_dot,
// softmax over the edges that share a target (torch_geometric.utils.softmax: exp(a - max) / (sum + 1e-16)).
// One workgroup per target; thread = (edge of the segment, head).
// Two kernels: (1) one wave per edge: lanes over the alpha channels of one head at a time (coalesced reads of the
// convolution's extra scalars), wave reductions for mean / variance / dot -> logits [E, NH]; (2) one thread per
// (target, head): softmax over the target's CSR segment, edges in order (run-to-run identical).
__global__ __launch_bounds__(256) void eq_alpha_logit_kernel(const float* __restrict__ y0, int ldy,
                                                             const int32_t* __restrict__ eptr, int n0, int n1,
                                                             const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                             const float* __restrict__ adot, int NH, int A,
                                                             float* __restrict__ logit) {
    const long long ebase = eptr[n0];
    const long long el = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ebase + el >= eptr[n1]) return;
    const float* row = y0 + (size_t)el * ldy;
    for (int hd = 0; hd < NH; ++hd) {
        const float* v = row + hd * A;
        float sum = 0.f;
        for (int a = lane; a < A; a += 64) sum += v[a];
        const float mean = eq_wave_sum(sum) / A;
        float q = 0.f;
        for (int a = lane; a < A; a += 64) { const float u = v[a] - mean; q += u * u; }
        const float rstd = rsqrtf(eq_wave_sum(q) / A + 1e-5f);
        float acc = 0.f;
        for (int a = lane; a < A; a += 64) {
            const float u = (v[a] - mean) * rstd * lnw[a] + lnb[a];
            // SmoothLeakyReLU(0.2): (1 + a)/2 x + (1 - a)/2 x (2 sigmoid(x) - 1)   (activation.py:30-45)
            const float sl = 0.6f * u + 0.4f * u * (2.0f / (1.0f + expf(-u)) - 1.0f);
            acc += sl * adot[hd * A + a];
        }
        acc = eq_wave_sum(acc);
        if (lane == 0) logit[(size_t)el * NH + hd] = acc;
    }
}

// A = 64 alpha channels (every shipped configuration): one wave per edge, FOUR heads at a time - lane = 16 (head & 3) + j, j owns
// channels 4 j .. 4 j + 3 of its head (one 16-byte load) - and the three sums per head (mean, variance, dot) are DPP row rotations
// inside the head's 16 lanes.  The kernel above runs the heads one after the other with three dependent 64-lane butterflies of
// ds_bpermute each: 144 LDS-crossbar round trips per edge in a chain (0.54 ms per launch at 256 k edges, 2.4 % of the pass).
__device__ __forceinline__ float eq_row16_sum(float v) {
#define EQ_ROR(n_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n_), 0xf, 0xf, false))
    v += EQ_ROR(8);
    v += EQ_ROR(4);
    v += EQ_ROR(2);
    v += EQ_ROR(1);
#undef EQ_ROR
    return v;
}

__global__ __launch_bounds__(256) void eq_alpha_logit64_kernel(const float* __restrict__ y0, int ldy,
                                                               const int32_t* __restrict__ eptr, int n0, int n1,
                                                               const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                               const float* __restrict__ adot, int NH,
                                                               float* __restrict__ logit) {
    const long long ebase = eptr[n0];
    const long long el = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 15, hq = lane >> 4;
    if (ebase + el >= eptr[n1]) return;
    const float* row = y0 + (size_t)el * ldy;
    const float4 w4 = *reinterpret_cast<const float4*>(lnw + 4 * j);
    const float4 b4 = *reinterpret_cast<const float4*>(lnb + 4 * j);
    for (int h0 = 0; h0 < NH; h0 += 4) {
        const int hd = h0 + hq, hc = hd < NH ? hd : NH - 1;   // (a head past NH repeats the last one and is not stored)
        const float4 v = *reinterpret_cast<const float4*>(row + hc * 64 + 4 * j);
        const float4 d4 = *reinterpret_cast<const float4*>(adot + hc * 64 + 4 * j);
        const float mean = eq_row16_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 64.0f);
        const float ux = v.x - mean, uy = v.y - mean, uz = v.z - mean, uw = v.w - mean;
        const float rstd = rsqrtf(eq_row16_sum((ux * ux + uy * uy) + (uz * uz + uw * uw)) * (1.0f / 64.0f) + 1e-5f);
        // SmoothLeakyReLU(0.2): (1 + a)/2 x + (1 - a)/2 x (2 sigmoid(x) - 1)   (activation.py:30-45)
        auto slr = [](float u) { return 0.6f * u + 0.4f * u * (2.0f / (1.0f + expf(-u)) - 1.0f); };
        const float tx = slr(ux * rstd * w4.x + b4.x), ty = slr(uy * rstd * w4.y + b4.y);
        const float tz = slr(uz * rstd * w4.z + b4.z), tw = slr(uw * rstd * w4.w + b4.w);
        const float acc = eq_row16_sum((tx * d4.x + ty * d4.y) + (tz * d4.z + tw * d4.w));
        if (j == 0 && hd < NH) logit[(size_t)el * NH + hd] = acc;
    }
}

__global__ void eq_alpha_softmax_kernel(float* __restrict__ alpha, const int32_t* __restrict__ eptr, int n0, int n1, int NH) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = n0 + (int)(t / NH), hd = (int)(t % NH);
    if (n >= n1) return;
    const long long ebase = eptr[n0], e0 = eptr[n] - ebase, e1 = eptr[n + 1] - ebase;
    float mx = -3.0e38f;
    for (long long e = e0; e < e1; ++e) mx = fmaxf(mx, alpha[(size_t)e * NH + hd]);
    float den = 0.f;
    for (long long e = e0; e < e1; ++e) den += expf(alpha[(size_t)e * NH + hd] - mx);
    const float inv = 1.0f / (den + 1e-16f);
    for (long long e = e0; e < e1; ++e) alpha[(size_t)e * NH + hd] = expf(alpha[(size_t)e * NH + hd] - mx) * inv;
}

int32_t eq_launch_alpha(const adf_eqv2* h, const eq_attn* at, const float* y0, int ldy, int n0, int n1, float* alpha,
                        hipStream_t s) {
    if (n1 <= n0) return ADF_OK;
    if (h->d.NH > 16) { adf_set_error("eqv2: more than 16 heads"); return ADF_EINVAL; }
    const long long Eub = eq_edge_bound(h, n1 - n0);
    const bool a64 = !h->alpha_generic && h->d.A == 64 && (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y0) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(at->alpha_ln_w) & 15) == 0 && (reinterpret_cast<uintptr_t>(at->alpha_ln_b) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(at->alpha_dot) & 15) == 0;
    if (a64)
        hipLaunchKernelGGL(eq_alpha_logit64_kernel, dim3((unsigned)((Eub + 3) / 4)), dim3(256), 0, s, y0, ldy, h->eptr, n0, n1,
                           at->alpha_ln_w, at->alpha_ln_b, at->alpha_dot, h->d.NH, alpha);
    else
        hipLaunchKernelGGL(eq_alpha_logit_kernel, dim3((unsigned)((Eub + 3) / 4)), dim3(256), 0, s, y0, ldy, h->eptr, n0, n1,
                           at->alpha_ln_w, at->alpha_ln_b, at->alpha_dot, h->d.NH, h->d.A, alpha);
    const long long nt = (long long)(n1 - n0) * h->d.NH;
    hipLaunchKernelGGL(eq_alpha_softmax_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, alpha, h->eptr, n0, n1,
                       h->d.NH);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ rotate out + aggregate
// transformer_block.py:347-366 + so3.py:501-506: the value message of every edge, weighted per head, is rotated back
// (D_l^T on the kept rows, with the m-truncation rescale) and summed over the edges of its target — one workgroup per
// target, thread = value channel, the target's S coefficients in registers, edges in CSR order (run-to-run identical).
// ONLY1: only the l = 1 coefficients (all a force block's projection reads), written as [N, 3, HV].
// (Measured alternative: several thread groups per target, each taking every k-th edge, partial sums added through LDS —
// slower, 61 -> 80 ms per forward at 256 k edges: the edge's Wigner rows stop being wave-uniform scalar loads.  Dealing the
// DEGREES of a target to four thread groups instead - whole waves, scalar Wigner loads kept, bit-identical sums, four times
// the waves per target - was slower too: 64.4 vs 60.9 ms.  Unrolling the edge loop 2x / 4x: no change.)
// MT > 0: the order cut-off M as a compile-time constant (= d.M; the shipped models use M = 2): the loop over the kept rows of
// D_l unrolls, the choice between the m = 0 row and the +-m pair resolves at compile time, and the 29 values an edge
// contributes are requested together instead of one dependent load per row (round 5: the runtime-M loop waited for every
// load before issuing the next - 580 memory round trips per target).  MT = 0: M read from d (any other cut-off).
template <int LT, bool ONLY1, int MT>
__global__ __launch_bounds__(256) void eq_rotate_out_kernel(const float* __restrict__ z0, eq_ptrs zm, const float* __restrict__ alpha,
                                     const float* __restrict__ wig, const int32_t* __restrict__ eptr, int n0, int n1,
                                     eq_dims d, int compact, float* __restrict__ agg) {
    // compact (ONLY1): the convolution wrote only the l = 1 columns - z0 [E, HV], z1 [2E, 2 HV] (real | imaginary part)
    const int n = n0 + blockIdx.x, c = threadIdx.x;
    if (n >= n1 || c >= d.HV) return;
    const long long ebase = eptr[n0];
    constexpr int NS = ONLY1 ? 3 : (LT + 1) * (LT + 1);
    float acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.f;
    const int hd = c / d.V;
    const int ld0 = compact ? d.HV : (d.L + 1) * d.HV;
    const int Mrt = MT > 0 ? MT : d.M;
    for (long long e = eptr[n]; e < eptr[n + 1]; ++e) {
        const long long el = e - ebase;
        const float a = alpha[(size_t)el * d.NH + hd];
        const float* D = wig + (size_t)e * d.DR;
        if constexpr (MT > 0) {
            // gather the edge's values first (static indices), then the products
            constexpr int LA = ONLY1 ? 1 : 0, LB = ONLY1 ? 1 : LT;
            float va[LB - LA + 1][2 * MT + 1], vb[LB - LA + 1][2 * MT + 1];   // the two loads of a +-m row (vb unused for m = 0)
#pragma unroll
            for (int l = LA; l <= LB; ++l) {
                const int ml = l < MT ? l : MT;
#pragma unroll
                for (int ri = 0; ri < 2 * MT + 1; ++ri) {
                    if (ri >= 2 * ml + 1) continue;
                    const int mp = ri - ml, am = mp < 0 ? -mp : mp;
                    if (am == 0) {
                        va[l - LA][ri] = z0[(size_t)el * ld0 + (compact ? 0 : l * d.HV) + c];
                        vb[l - LA][ri] = 0.f;
                    } else {
                        const int nm = compact ? 1 : d.L - am + 1, half = nm * d.HV, W = 2 * half, q = (compact ? 0 : (l - am) * d.HV) + c;
                        const float* zr = zm.p[am] + (size_t)(2 * el) * W;
                        va[l - LA][ri] = mp > 0 ? zr[q] : zr[W + q];
                        vb[l - LA][ri] = mp > 0 ? zr[W + half + q] : zr[half + q];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // every load of the edge is requested before the first one is used
            float vals[LB - LA + 1][2 * MT + 1];
#pragma unroll
            for (int l = LA; l <= LB; ++l) {
                const int ml = l < MT ? l : MT;
#pragma unroll
                for (int ri = 0; ri < 2 * MT + 1; ++ri) {
                    if (ri >= 2 * ml + 1) continue;
                    const int mp = ri - ml;
                    vals[l - LA][ri] = mp > 0 ? va[l - LA][ri] - vb[l - LA][ri] : va[l - LA][ri] + vb[l - LA][ri];
                }
            }
#pragma unroll
            for (int l = LA; l <= LB; ++l) {
                const int ml = l < MT ? l : MT;
                const float* Dl = D + d.d_off[l];
                const float sc = a * d.resc[l];
#pragma unroll
                for (int ri = 0; ri < 2 * MT + 1; ++ri) {
                    if (ri >= 2 * ml + 1) continue;
                    const float v = vals[l - LA][ri] * sc;
#pragma unroll
                    for (int m = 0; m < 2 * l + 1; ++m) acc[(ONLY1 ? 0 : l * l) + m] += Dl[ri * (2 * l + 1) + m] * v;
                }
            }
        } else {
#pragma unroll
        for (int l = (ONLY1 ? 1 : 0); l <= (ONLY1 ? 1 : LT); ++l) {
            const int ml = l < Mrt ? l : Mrt;
            const float* Dl = D + d.d_off[l];
            const float sc = a * d.resc[l];
            for (int ri = 0; ri < 2 * ml + 1; ++ri) {
                const int mp = ri - ml, am = mp < 0 ? -mp : mp;
                float v;
                if (am == 0) {
                    v = z0[(size_t)el * ld0 + (compact ? 0 : l * d.HV) + c];
                } else {
                    const int nm = compact ? 1 : d.L - am + 1, half = nm * d.HV, W = 2 * half, q = (compact ? 0 : (l - am) * d.HV) + c;
                    const float* zr = zm.p[am] + (size_t)(2 * el) * W;
                    v = mp > 0 ? zr[q] - zr[W + half + q] : zr[W + q] + zr[half + q];
                }
                v *= sc;
#pragma unroll
                for (int m = 0; m < 2 * l + 1; ++m) acc[(ONLY1 ? 0 : l * l) + m] += Dl[ri * (2 * l + 1) + m] * v;
            }
        }
        }
    }
    float* out = agg + (size_t)n * NS * d.HV + c;
#pragma unroll
    for (int s = 0; s < NS; ++s) out[(size_t)s * d.HV] = acc[s];
}

int32_t eq_launch_rotate_out(const adf_eqv2* h, float* const* z, const float* alpha, int n0, int n1, float* agg,
                             bool only_l1, hipStream_t s, bool compact) {
    if (compact && !only_l1) { adf_set_error("internal: compact rotate-out is for the l = 1 mode"); return ADF_EINVAL; }
    if (n1 <= n0) return ADF_OK;
    eq_ptrs zm;
    for (int m = 0; m <= h->d.M; ++m) zm.p[m] = z[m];
    const int bd = (h->d.HV + 63) / 64 * 64;
    if (bd > 256) { adf_set_error("rotate_out: num_heads * attn_value_channels > 256 (one thread per value channel; INTEGRATION.md 5)"); return ADF_EINVAL; }
    static const bool generic_ro = getenv("ADF_EQV2_ROTOUT_GENERIC") != nullptr;   // read once, not per launch
    const bool m2 = h->d.M == 2 && !generic_ro;   // the compile-time-M fast path (M = 2: the shipped models)
#define EQ_RO_(LT_, MT_)                                                                                               \
    if (only_l1)                                                                                                       \
        hipLaunchKernelGGL((eq_rotate_out_kernel<LT_, true, MT_>), dim3(n1 - n0), dim3(bd), 0, s, z[0], zm, alpha, h->wig, \
                           h->eptr, n0, n1, h->d, compact ? 1 : 0, agg);                                               \
    else                                                                                                               \
        hipLaunchKernelGGL((eq_rotate_out_kernel<LT_, false, MT_>), dim3(n1 - n0), dim3(bd), 0, s, z[0], zm, alpha, h->wig, \
                           h->eptr, n0, n1, h->d, 0, agg)
#define EQ_RO(LT_) if (m2) { EQ_RO_(LT_, 2); } else { EQ_RO_(LT_, 0); }
    EQ_FOR_L(h->d.L, EQ_RO)
#undef EQ_RO
#undef EQ_RO_
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// l = 1 rows of a force block's SO3_LinearV2 with one output channel (so3.py:694-745; no bias above l = 0):
// f[n, m] = sum_i agg3[n, m, i] * W[1, 0, i]; one wave per atom
__global__ __launch_bounds__(256) void eq_force_out_kernel(const float* __restrict__ agg3, const float* __restrict__ w1,
                                                           int N, int HV, float* __restrict__ f) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float* r = agg3 + (size_t)n * 3 * HV;
    for (int i = lane; i < HV; i += 64) {
        const float w = w1[i];
        a0 += r[i] * w; a1 += r[HV + i] * w; a2 += r[2 * HV + i] * w;
    }
    a0 = eq_wave_sum(a0); a1 = eq_wave_sum(a1); a2 = eq_wave_sum(a2);
    if (lane == 0) { f[3 * n] = a0; f[3 * n + 1] = a1; f[3 * n + 2] = a2; }
}

int32_t eq_launch_force_out(const adf_eqv2* h, const eq_attn* at, const float* agg3, int N, float* f, hipStream_t s) {
    hipLaunchKernelGGL(eq_force_out_kernel, dim3((N + 3) / 4), dim3(256), 0, s, agg3, at->proj_w + (size_t)1 * 1 * h->d.HV,
                       N, h->d.HV, f);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ target subsets
__global__ void eq_subset_count_kernel(const int32_t* __restrict__ eptr, const int32_t* __restrict__ out_idx, int n_out,
                                       int N, int32_t* __restrict__ cnt, int32_t* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_out) return;
    if (i == n_out) { cnt[i] = 0; return; }
    const int n = out_idx[i];
    if (n < 0 || n >= N) { cnt[i] = 0; atomicOr(&flags[5], 1); return; }
    cnt[i] = eptr[n + 1] - eptr[n];
}

// one workgroup per listed target: its incoming edges (source, target, vector, Wigner rows) to the compact arrays
__global__ void eq_subset_copy_kernel(const int32_t* __restrict__ eptr, const int32_t* __restrict__ out_idx, int N,
                                      const int32_t* __restrict__ sub_eptr, const int32_t* __restrict__ e_src,
                                      const int32_t* __restrict__ e_dst, const float* __restrict__ e_vec,
                                      const float* __restrict__ wig, int DR, int32_t* __restrict__ s_src,
                                      int32_t* __restrict__ s_dst, float* __restrict__ s_vec, float* __restrict__ s_wig) {
    const int i = blockIdx.x, n = out_idx[i];
    if (n < 0 || n >= N) return;
    const long long e0 = eptr[n], o0 = sub_eptr[i];
    const int cnt = eptr[n + 1] - eptr[n];
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
        s_src[o0 + t] = e_src[e0 + t];
        s_dst[o0 + t] = e_dst[e0 + t];
    }
    for (int t = threadIdx.x; t < 3 * cnt; t += blockDim.x) s_vec[3 * o0 + t] = e_vec[3 * e0 + t];
    for (long long t = threadIdx.x; t < (long long)cnt * DR; t += blockDim.x) s_wig[o0 * DR + t] = wig[e0 * DR + t];
}

int32_t eq_launch_subset_graph(adf_eqv2* h, const int32_t* out_idx, int n_out, hipStream_t s) {
    if (n_out <= 0) return ADF_OK;
    int32_t* cnt = h->sub_eptr + (h->sub_cap + 2);  // the counts the CSR is scanned from live behind it
    hipLaunchKernelGGL(eq_subset_count_kernel, dim3((n_out + 256) / 256), dim3(256), 0, s, h->eptr, out_idx, n_out, (int)h->lastN,
                       cnt, h->flags);
    size_t tmp = h->scan_tmp_bytes;
    ADF_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(h->scan_tmp, tmp, cnt, h->sub_eptr, n_out + 1, s));
    hipLaunchKernelGGL(eq_subset_copy_kernel, dim3(n_out), dim3(256), 0, s, h->eptr, out_idx, (int)h->lastN, h->sub_eptr, h->e_src,
                       h->e_dst, h->e_vec, h->wig, h->d.DR, h->sub_src, h->sub_dst, h->sub_vec, h->sub_wig);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

__global__ void eq_scatter_rows3_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int n,
                                        float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * n) dst[(size_t)idx[i / 3] * 3 + i % 3] = src[i];
}

int32_t eq_launch_scatter_rows3(const float* f_sub, const int32_t* out_idx, int n_out, float* f, hipStream_t s) {
    if (n_out <= 0) return ADF_OK;
    hipLaunchKernelGGL(eq_scatter_rows3_kernel, dim3((3 * n_out + 255) / 256), dim3(256), 0, s, f_sub, out_idx, n_out, f);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ incremental blocks
// Which rows a block has to recompute (adf_eqv2_set_incremental).  The embedding of atom t reads t's incoming edges (who,
// where); block i reads, for target t, row t and the rows of t's sources in the block below plus t's edge geometry
// (transformer_block.py:650-700).  With G = the targets whose in-edge list differs bit for bit from the previous forward's,
// block i therefore recomputes P^(i+1)(G), P(D) = D + the targets with a source in D.  Everything else is unchanged since
// the forward that last computed it.
__global__ void eq_inc_compare_kernel(const int32_t* __restrict__ eptr, const int32_t* __restrict__ e_src,
                                      const float* __restrict__ e_vec, const int32_t* __restrict__ peptr,
                                      const int32_t* __restrict__ psrc, const float* __restrict__ pvec, int N,
                                      unsigned char* __restrict__ dirty) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const int c0 = eptr[t], c1 = eptr[t + 1], p0 = peptr[t], p1 = peptr[t + 1];
    bool d = (c1 - c0) != (p1 - p0);
    for (int k = 0; !d && k < c1 - c0; ++k) {
        d = e_src[c0 + k] != psrc[p0 + k];
        for (int c = 0; c < 3; ++c)
            d = d || __float_as_uint(e_vec[3 * (size_t)(c0 + k) + c]) != __float_as_uint(pvec[3 * (size_t)(p0 + k) + c]);
    }
    dirty[t] = d ? 1 : 0;
}

__global__ void eq_inc_propagate_kernel(const int32_t* __restrict__ eptr, const int32_t* __restrict__ e_src, int N,
                                        const unsigned char* __restrict__ in, unsigned char* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    bool d = in[t] != 0;
    for (int e = eptr[t]; !d && e < eptr[t + 1]; ++e) d = in[e_src[e]] != 0;
    out[t] = d ? 1 : 0;
}

size_t eq_inc_select_bytes(int N) {
    size_t bytes = 0;
    (void)hipcub::DeviceSelect::Flagged(nullptr, bytes, hipcub::CountingInputIterator<int32_t>(0), (unsigned char*)nullptr,
                                        (int32_t*)nullptr, (int32_t*)nullptr, N);
    return bytes;
}

// lists[l] (ascending rows, stride capN) and counts[l] for blocks l = 0..nl-1, against the graph kept by eq_launch_inc_keep
int32_t eq_launch_inc_lists(adf_eqv2* h, int N, int nl, hipStream_t s) {
    const dim3 grid((N + 255) / 256), block(256);
    unsigned char* f0 = h->inc_dirty;
    unsigned char* f1 = h->inc_dirty + h->inc_capN;
    hipLaunchKernelGGL(eq_inc_compare_kernel, grid, block, 0, s, h->eptr, h->e_src, h->e_vec, h->inc_peptr, h->inc_psrc,
                       h->inc_pvec, N, f0);
    for (int l = 0; l < nl; ++l) {
        hipLaunchKernelGGL(eq_inc_propagate_kernel, grid, block, 0, s, h->eptr, h->e_src, N, f0, f1);
        size_t bytes = h->inc_sel_bytes;
        ADF_HIP_CHECK(hipcub::DeviceSelect::Flagged(h->inc_sel_tmp, bytes, hipcub::CountingInputIterator<int32_t>(0), f1,
                                                    h->inc_idx + (size_t)l * h->inc_capN, h->inc_cnt + l, N, s));
        unsigned char* t = f0; f0 = f1; f1 = t;
    }
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// keep this forward's graph for the next comparison
int32_t eq_launch_inc_keep(adf_eqv2* h, int N, hipStream_t s) {
    int64_t E = (int64_t)N * (h->ext_graph ? h->maxdeg : h->hp.max_neighbors);
    if (h->ext_graph && h->E_ext < E) E = h->E_ext;
    if (E > h->inc_capE) E = h->inc_capE;
    ADF_HIP_CHECK(hipMemcpyAsync(h->inc_peptr, h->eptr, sizeof(int32_t) * ((size_t)N + 1), hipMemcpyDeviceToDevice, s));
    ADF_HIP_CHECK(hipMemcpyAsync(h->inc_psrc, h->e_src, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToDevice, s));
    ADF_HIP_CHECK(hipMemcpyAsync(h->inc_pvec, h->e_vec, sizeof(float) * 3 * (size_t)E, hipMemcpyDeviceToDevice, s));
    return ADF_OK;
}

// rows idx[i] of a [*, row4 float4] array <-> a compact [n, row4] array
__global__ void eq_gather_rows_kernel(const float4* __restrict__ src, const int32_t* __restrict__ idx, int row4,
                                      float4* __restrict__ dst) {
    const float4* a = src + (size_t)idx[blockIdx.x] * row4;
    float4* b = dst + (size_t)blockIdx.x * row4;
    for (int t = threadIdx.x; t < row4; t += blockDim.x) b[t] = a[t];
}

__global__ void eq_scatter_rows_kernel(const float4* __restrict__ src, const int32_t* __restrict__ idx, int row4,
                                       float4* __restrict__ dst) {
    const float4* a = src + (size_t)blockIdx.x * row4;
    float4* b = dst + (size_t)idx[blockIdx.x] * row4;
    for (int t = threadIdx.x; t < row4; t += blockDim.x) b[t] = a[t];
}

int32_t eq_launch_gather_rows(const float* src, const int32_t* idx, int n, int row_floats, float* dst, hipStream_t s) {
    if (n <= 0) return ADF_OK;
    if (row_floats & 3) { adf_set_error("eqv2: row length must be a multiple of 4"); return ADF_EINVAL; }
    hipLaunchKernelGGL(eq_gather_rows_kernel, dim3(n), dim3(256), 0, s, reinterpret_cast<const float4*>(src), idx, row_floats / 4,
                       reinterpret_cast<float4*>(dst));
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t eq_launch_scatter_rows(const float* src, const int32_t* idx, int n, int row_floats, float* dst, hipStream_t s) {
    if (n <= 0) return ADF_OK;
    if (row_floats & 3) { adf_set_error("eqv2: row length must be a multiple of 4"); return ADF_EINVAL; }
    hipLaunchKernelGGL(eq_scatter_rows_kernel, dim3(n), dim3(256), 0, s, reinterpret_cast<const float4*>(src), idx, row_floats / 4,
                       reinterpret_cast<float4*>(dst));
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ S2 grid of the feed-forward
// transformer_block.py:497-531: SO3 features -> res^2 grid points (to_full), point-wise MLP (dense products elsewhere),
// grid -> SO3 (from_full), l = 0 replaced by the scalar gate.  One workgroup per node, thread = hidden channel.
template <int LT>
__global__ void eq_to_grid_kernel(const float* __restrict__ h1, const float* __restrict__ tg, int n0, int n1, eq_dims d,
                                  int silu, float* __restrict__ g) {
    extern __shared__ float T[];  // [G][S]
    constexpr int S = (LT + 1) * (LT + 1);
    for (int t = threadIdx.x; t < d.G * S; t += blockDim.x) T[t] = tg[t];
    __syncthreads();
    const int n = n0 + blockIdx.x, f = threadIdx.x;
    if (n >= n1 || f >= d.F) return;
    float in[S];
    const float* hr = h1 + (size_t)n * S * d.F + f;
#pragma unroll
    for (int s = 0; s < S; ++s) in[s] = hr[(size_t)s * d.F];
    float* gr = g + (size_t)(n - n0) * d.G * d.F + f;
    for (int p = 0; p < d.G; ++p) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) a += T[p * S + s] * in[s];
        gr[(size_t)p * d.F] = silu ? eq_silu(a) : a;
    }
}

template <int LT>
__global__ void eq_from_grid_kernel(const float* __restrict__ g, const float* __restrict__ fg, const float* __restrict__ gate,
                                    int n0, int n1, eq_dims d, float* __restrict__ h2) {
    extern __shared__ float Fm[];  // [G][S]
    constexpr int S = (LT + 1) * (LT + 1);
    for (int t = threadIdx.x; t < d.G * S; t += blockDim.x) Fm[t] = fg[t];
    __syncthreads();
    const int n = n0 + blockIdx.x, f = threadIdx.x;
    if (n >= n1 || f >= d.F) return;
    float acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = 0.f;
    const float* gr = g + (size_t)(n - n0) * d.G * d.F + f;
    for (int p = 0; p < d.G; ++p) {
        const float v = gr[(size_t)p * d.F];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] += Fm[p * S + s] * v;
    }
    float* hr = h2 + (size_t)n * S * d.F + f;
    hr[0] = gate[(size_t)n * d.F + f];
#pragma unroll
    for (int s = 1; s < S; ++s) hr[(size_t)s * d.F] = acc[s];
}

// ---- matrix-core versions (f16x3 split, constant operands as fp16 hi/lo fragment images in LDS, one wave per
// (node, block of 32 hidden channels), power-of-two lift from that item's own maximum).
//   to grid:   G[p, f] = sum_s T[p, s] h1[s, f]      A = T (M = grid point, K = coefficient, 4 k-steps), B from memory
//   from grid: out[s, f] = sum_p F[p, s] g[p, f]     A = F^T (M = coefficient, 2 blocks; K = grid point), B from memory
__global__ __launch_bounds__(512, 2) void eq_to_grid_mfma_kernel(const float* __restrict__ h1, const eqhalf8* __restrict__ tabs,
                                                                  int npb, float inv_sT, int n0, int n1, int S, int F, int G,
                                                                  int silu, float* __restrict__ g,
                                                                  unsigned int* __restrict__ node_mag) {
    extern __shared__ eqhalf8 tab[];  // [npb][4 ks][hi|lo][64]
    const int ntab = npb * 4 * 2 * 64;
    for (int t = threadIdx.x; t < ntab; t += 512) tab[t] = tabs[t];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cl = lane & 31, kh = lane >> 5;
    const int nblk = F >> 5;
    const long long items = (long long)(n1 - n0) * nblk;
    for (long long item = (long long)blockIdx.x * 8 + wave; item < items; item += (long long)gridDim.x * 8) {
        const int n = n0 + (int)(item / nblk), f = (int)(item % nblk) * 32 + cl;
        const float* hr = h1 + (size_t)n * S * F + f;
        float vin[4][8];
        float mx = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sidx = 16 * ks + 8 * kh + j;
                const float v = sidx < S ? hr[(size_t)sidx * F] : 0.f;
                vin[ks][j] = v;
                mx = fmaxf(mx, fabsf(v));
            }
        mx = eq_wave_max_nonneg(mx);
        const float lift = eq_pow2_lift(mx);
        eqhalf8 bh[4], bl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            equint4v h4, l4;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                unsigned int hw, lw;
                eq_split_pair(vin[ks][j], vin[ks][j + 1], lift, hw, lw);
                h4[j >> 1] = hw;
                l4[j >> 1] = lw;
            }
            bh[ks] = __builtin_bit_cast(eqhalf8, h4);
            bl[ks] = __builtin_bit_cast(eqhalf8, l4);
        }
        const float sc = inv_sT / lift;
        float* gr = g + (size_t)(n - n0) * G * F + f;
        float omx = 0.f;
        for (int pb = 0; pb < npb; ++pb) {
            eqf32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const eqhalf8 ah = tab[((pb * 4 + ks) * 2 + 0) * 64 + lane];
                const eqhalf8 al = tab[((pb * 4 + ks) * 2 + 1) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = 32 * pb + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float v = acc[r] * sc, w = silu ? eq_silu(v) : v;
                if (pp < G) { gr[(size_t)pp * F] = w; omx = fmaxf(omx, fabsf(w)); }
            }
        }
        if (node_mag) {  // float bits of a non-negative value order like unsigned integers
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) omx = fmaxf(omx, __shfl_xor(omx, o));
            if (lane == 0) atomicMax(node_mag + (n - n0), __float_as_uint(omx));
        }
    }
}

// One pass over memory: the item's [G, 32] tile is taken in chunks of CH k-steps (16 grid points each); every chunk has
// its loads in flight together, its own power-of-two lift PER CHANNEL (a lane and its partner 32 lanes on hold one column
// of the B operand, and every accumulator register of a lane belongs to that column, so the lift is a per-lane scalar) and
// its own accumulators, folded into the running sums in fp32.  FT > 0: the channel count as a compile-time constant
// (row offsets become instruction immediates).
template <int CH, int FT>
__global__ __launch_bounds__(512, 1) void eq_from_grid_mfma_kernel(const float* __restrict__ g, const eqhalf8* __restrict__ tabs,
                                                                    int nkst, float inv_sF, const float* __restrict__ gate,
                                                                    int n0, int n1, int S, int Frt, int G, float* __restrict__ h2) {
    extern __shared__ eqhalf8 tab[];  // [nkst][2 sb][hi|lo][64]
    const int F = FT > 0 ? FT : Frt;
    const int ntab = nkst * 2 * 2 * 64;
    for (int t = threadIdx.x; t < ntab; t += 512) tab[t] = tabs[t];
    __syncthreads();
    const int lane = threadIdx.x & 63, cl = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: item, node and base pointers stay scalar
    const int nblk = F >> 5;
    const long long items = (long long)(n1 - n0) * nblk;
    for (long long item = (long long)blockIdx.x * 8 + wave; item < items; item += (long long)gridDim.x * 8) {
        const int n = n0 + (int)(item / nblk), fb = (int)(item % nblk) * 32, f = fb + cl;
        const float* gl = g + (size_t)(n - n0) * G * F + fb + 8 * kh * F + cl;
        eqf32x16 tot[2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[sb][r] = 0.f;
        for (int k0 = 0; k0 < nkst; k0 += CH) {
            float v[CH][8];
            float mx = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int pp = 16 * (k0 + c) + 8 * kh + j;
                    v[c][j] = pp < G ? gl[(size_t)(16 * (k0 + c) + j) * F] : 0.f;
                }
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(v[c][j]));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float lift = eq_pow2_lift(mx);
            eqf32x16 acc[2];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[sb][r] = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (k0 + c < nkst) {
                    equint4v h4, l4;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        unsigned int hw, lw;
                        eq_split_pair(v[c][j], v[c][j + 1], lift, hw, lw);
                        h4[j >> 1] = hw;
                        l4[j >> 1] = lw;
                    }
                    const eqhalf8 bh = __builtin_bit_cast(eqhalf8, h4), bl = __builtin_bit_cast(eqhalf8, l4);
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        const eqhalf8 ah = tab[(((k0 + c) * 2 + sb) * 2 + 0) * 64 + lane];
                        const eqhalf8 al = tab[(((k0 + c) * 2 + sb) * 2 + 1) * 64 + lane];
                        acc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[sb], 0, 0, 0);
                        acc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[sb], 0, 0, 0);
                        acc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[sb], 0, 0, 0);
                    }
                }
            }
            const float il = 1.0f / lift;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[sb][r] += acc[sb][r] * il;
        }
        float* hr = h2 + (size_t)n * S * F + f;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sidx = 32 * sb + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (sidx < S) hr[(size_t)sidx * F] = sidx == 0 ? gate[(size_t)n * F + f] : tot[sb][r] * inv_sF;
            }
    }
}

// out[o, i] = sum_k A[o, k] B[k, i] (row-major, double accumulation): folded weights, once per weight binding
__global__ void eq_fold_kernel(const float* __restrict__ A, const float* __restrict__ B, int O, int K, int I,
                               float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)O * I) return;
    const int o = (int)(t / I), i = (int)(t % I);
    double a = 0.0;
    for (int k = 0; k < K; ++k) a += (double)A[(size_t)o * K + k] * (double)B[(size_t)k * I + i];
    out[t] = (float)a;
}

int32_t eq_launch_fold(const float* A, const float* B, int O, int K, int I, float* out, hipStream_t s) {
    const long long n = (long long)O * I;
    hipLaunchKernelGGL(eq_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A, B, O, K, I, out);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t eq_launch_to_grid(const adf_eqv2* h, const float* h1, int n0, int n1, float* g, bool silu, hipStream_t s, float* node_mag,
                          bool* emitted) {
    if (emitted) *emitted = false;
    if (n1 <= n0) return ADF_OK;
    if (!h->exact_f32 && h->gtab_to && h->d.S <= 64 && h->d.F % 32 == 0) {
        const size_t dyn = (size_t)h->g_npb * 4 * 2 * 64 * 16;
        ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&eq_to_grid_mfma_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        if (node_mag) ADF_HIP_CHECK(hipMemsetAsync(node_mag, 0, sizeof(float) * (size_t)(n1 - n0), s));
        hipLaunchKernelGGL(eq_to_grid_mfma_kernel, dim3(h->num_cus), dim3(512), dyn, s, h1, (const eqhalf8*)h->gtab_to, h->g_npb,
                           h->g_inv_sT, n0, n1, h->d.S, h->d.F, h->d.G, silu ? 1 : 0, g, reinterpret_cast<unsigned int*>(node_mag));
        if (emitted) *emitted = node_mag != nullptr;
        ADF_HIP_CHECK(hipGetLastError());
        return ADF_OK;
    }
    const int bd = (h->d.F + 63) / 64 * 64;
    const size_t dyn = sizeof(float) * h->d.G * h->d.S;
#define EQ_TG(LT_) hipLaunchKernelGGL(eq_to_grid_kernel<LT_>, dim3(n1 - n0), dim3(bd), dyn, s, h1, h->to_full, n0, n1, h->d, silu ? 1 : 0, g)
    EQ_FOR_L(h->d.L, EQ_TG)
#undef EQ_TG
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t eq_launch_from_grid(const adf_eqv2* h, const float* g, const float* gate, int n0, int n1, float* h2, hipStream_t s) {
    if (n1 <= n0) return ADF_OK;
    if (!h->exact_f32 && h->gtab_from && h->d.S <= 64 && h->d.F % 32 == 0) {
        const size_t dyn = (size_t)h->g_nkst * 2 * 2 * 64 * 16;
#define EQ_FGM(CH_, FT_)                                                                                                     \
    do {                                                                                                                     \
        ADF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&eq_from_grid_mfma_kernel<CH_, FT_>),                \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));                            \
        hipLaunchKernelGGL((eq_from_grid_mfma_kernel<CH_, FT_>), dim3(h->num_cus), dim3(512), dyn, s, g,                     \
                           (const eqhalf8*)h->gtab_from, h->g_nkst, h->g_inv_sF, gate, n0, n1, h->d.S, h->d.F, h->d.G, h2);  \
    } while (0)
        if (h->d.F == 128) EQ_FGM(7, 128); else EQ_FGM(7, 0);
#undef EQ_FGM
        ADF_HIP_CHECK(hipGetLastError());
        return ADF_OK;
    }
    const int bd = (h->d.F + 63) / 64 * 64;
    const size_t dyn = sizeof(float) * h->d.G * h->d.S;
#define EQ_FG(LT_) hipLaunchKernelGGL(eq_from_grid_kernel<LT_>, dim3(n1 - n0), dim3(bd), dyn, s, g, h->from_full, gate, n0, n1, h->d, h2)
    EQ_FOR_L(h->d.L, EQ_FG)
#undef EQ_FG
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// ------------------------------------------------------------------------------------------------ dense products (any shape)
// C (+)= act(A . W^T + bias), exact f32 on the vector ALU: 64 x 64 tile per 256-thread workgroup, 4 x 4 per thread,
// K in steps of 16 through LDS.  Rows of A and C may be strided in two levels (eq_rowmap) so that the per-degree maps of an
// SO3_LinearV2 read and write [N, S, C] tensors in place.  The f16x3 matrix-core kernel (gemm16.hip) takes over for
// aligned shapes (eqv2_api.hip).
__global__ __launch_bounds__(256) void eq_gemm_kernel(const float* __restrict__ A, int lda, eq_rowmap am,
                                                      const float* __restrict__ W, const float* __restrict__ bias,
                                                      float* __restrict__ Cm, int ldc, eq_rowmap cm, long long M, int N,
                                                      int K, int act, int accumulate) {
    __shared__ float As[16][68];
    __shared__ float Ws[16][68];
    const int tid = threadIdx.x;
    const long long m0 = (long long)blockIdx.x * 64;
    const int nb = blockIdx.y * 64;
    const int tr = tid >> 4, tc = tid & 15;  // thread tile: rows 4 tr.., columns 4 tc..
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // loader: thread -> (row = tid / 4, k quad = tid % 4)
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const long long arow = m0 + lr;
    const float* ap = nullptr;
    if (arow < M) ap = A + (arow / am.period) * am.outer + (arow % am.period) * (long long)am.inner;
    const int wrow = nb + lr;
    const float* wp = wrow < N ? W + (size_t)wrow * K : nullptr;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + lk + u;
            As[lk + u][lr] = (ap && k < K) ? ap[k] : 0.f;
            Ws[lk + u][lr] = (wp && k < K) ? wp[k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[k][4 * tr + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Ws[k][4 * tc + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long row = m0 + 4 * tr + i;
        if (row >= M) continue;
        float* cp = Cm + (row / cm.period) * cm.outer + (row % cm.period) * (long long)cm.inner;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = nb + 4 * tc + j;
            if (col >= N) continue;
            float v = acc[i][j] + (bias ? bias[col] : 0.f);
            if (act == 2) v = eq_silu(v);
            if (accumulate) v += cp[col];
            cp[col] = v;
        }
    }
}

int32_t eq_gemm_f32(const float* A, int lda, const eq_rowmap* amap, const float* W, const float* bias, float* Cm, int ldc,
                    const eq_rowmap* cmap, long long M, int N, int K, int act, bool accumulate, hipStream_t s) {
    if (M <= 0 || N <= 0) return ADF_OK;
    const eq_rowmap a1 = {lda, 1, 0}, c1 = {ldc, 1, 0};  // plain rows: r * ld
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64));
    hipLaunchKernelGGL(eq_gemm_kernel, grid, dim3(256), 0, s, A, lda, amap ? *amap : a1, W, bias, Cm, ldc,
                       cmap ? *cmap : c1, M, N, K, act, accumulate ? 1 : 0);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// [rows, cols] -> [cols, rows]
__global__ void eq_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)rows * cols) return;
    const int r = (int)(t / cols), c = (int)(t - (long long)r * cols);
    out[(size_t)c * rows + r] = in[t];
}

int32_t eq_launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t s) {
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(eq_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, rows, cols);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
