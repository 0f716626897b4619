// Per-node elementwise / reduction kernels of the PaiNN denoiser that sit between the MFMA
// GEMMs: embedding lookup, LayerNorm, the PaiNNUpdate gating and the gated-equivariant heads.
// All are HBM-bound streaming kernels: one float4 (16 B) per lane, rows contiguous.
//
// Reference: adsorbdiff/models/painn/painn_denoising.py
//   :425-426 atom_emb / vec = 0      :531 x_layernorm      :601-623 PaiNNUpdate
//   :647-650,688-697 PaiNNOutput / GatedEquivariantBlock
#include "common.h"

__device__ __forceinline__ float ssilu_d(float x) {
    float s = x / (1.0f + expf(-x));
    return s * 1.6666666666666667f;
}

// x[n,:] = emb[Z[n]-1,:]   (gemnet_oc/layers/embedding_block.py:42).  vec = 0 (painn_denoising.py:426) is
// not materialised: the first message layer runs in its vec-is-zero mode.
// Z outside [1, num_elements] (torch's nn.Embedding raises IndexError there): flagged, row read clamped.
__global__ void adf_embed_kernel(const float* __restrict__ emb, const int32_t* __restrict__ Z, float* __restrict__ x,
                                 int N, int H, int num_elements, int32_t* flags) {
    const int h4 = H / 4;
    const long long total = (long long)N * h4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / h4), c = (int)(i - (long long)n * h4);
        int z = Z[n] - 1;
        if (z < 0 || z >= num_elements) { if (c == 0) atomicExch(&flags[4], 1); z = min(max(z, 0), num_elements - 1); }
        const float4 v = reinterpret_cast<const float4*>(emb + (size_t)z * H)[c];
        reinterpret_cast<float4*>(x + (size_t)n * H)[c] = v;
    }
}

// torch.nn.LayerNorm(H), eps = 1e-5, biased variance.  One wave per row, 16 B per lane and access.
__global__ __launch_bounds__(256) void adf_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ y,
                                                             int N, int H, float* __restrict__ out_mag,
                                                             const int32_t* __restrict__ n_dev) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N || (n_dev && row >= *n_dev)) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * H);
    const int h4 = H / 4;  // H <= 1024: at most 4 float4 per lane
    float4 vals[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        vals[i] = c < h4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (vals[i].x + vals[i].y) + (vals[i].z + vals[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < h4) {
            const float dx = vals[i].x - mean, dy = vals[i].y - mean, dz = vals[i].z - mean, dw = vals[i].w - mean;
            sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)H + 1e-5f);
    float4* yr = reinterpret_cast<float4*>(y + (size_t)row * H);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float mg = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < h4) {
            const float4 ww = w4[c], bb = b4[c];
            const float4 o = make_float4((vals[i].x - mean) * rstd * ww.x + bb.x, (vals[i].y - mean) * rstd * ww.y + bb.y,
                                         (vals[i].z - mean) * rstd * ww.z + bb.z, (vals[i].w - mean) * rstd * ww.w + bb.w);
            yr[c] = o;
            mg = fmaxf(fmaxf(fmaxf(mg, fabsf(o.x)), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (out_mag) {  // the row's magnitude for the f16x3 product that reads it (gemm16.hip row lifts)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mg = fmaxf(mg, __shfl_xor(mg, o));
        if (lane == 0) out_mag[row] = mg;
    }
}

// PaiNNUpdate, first half (painn_denoising.py:602-613): vv = vec_proj(vec) as [N,3,2H] (v1|v2)
//   dot = sum_xyz(v1*v2)/sqrt(H) ; cat = [x, sqrt(sum_xyz v2^2 + 1e-8)]
__global__ void adf_update_prep_kernel(const float* __restrict__ vv, const float* __restrict__ x,
                                       float* __restrict__ cat, float* __restrict__ dot, int N, int H) {
    const int h4 = H / 4;
    const long long total = (long long)N * h4;
    const float inv_sqrt_h = 1.0f / sqrtf((float)H);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / h4), c = (int)(i - (long long)n * h4);
        const float4* base = reinterpret_cast<const float4*>(vv + (size_t)n * 6 * H);
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f), q = d;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float4 v1 = base[a * 2 * h4 + c];
            const float4 v2 = base[a * 2 * h4 + h4 + c];
            d.x += v1.x * v2.x; d.y += v1.y * v2.y; d.z += v1.z * v2.z; d.w += v1.w * v2.w;
            q.x += v2.x * v2.x; q.y += v2.y * v2.y; q.z += v2.z * v2.z; q.w += v2.w * v2.w;
        }
        reinterpret_cast<float4*>(dot + (size_t)n * H)[c] =
            make_float4(d.x * inv_sqrt_h, d.y * inv_sqrt_h, d.z * inv_sqrt_h, d.w * inv_sqrt_h);
        float4* cr = reinterpret_cast<float4*>(cat + (size_t)n * 2 * H);
        cr[c] = reinterpret_cast<const float4*>(x + (size_t)n * H)[c];
        cr[h4 + c] = make_float4(sqrtf(q.x + 1e-8f), sqrtf(q.y + 1e-8f), sqrtf(q.z + 1e-8f), sqrtf(q.w + 1e-8f));
    }
}

// PaiNNUpdate, second half + residuals + ScaleFactor (painn_denoising.py:614-623, 449-451):
//   x = (x + (h1 + h2*dot)/sqrt2) * s ; vec += h3 (x) v1
__global__ void adf_update_apply_kernel(const float* __restrict__ h3, const float* __restrict__ dot,
                                        const float* __restrict__ vv, float* __restrict__ x, float* __restrict__ vec,
                                        float scale, int N, int H) {
    const int h4 = H / 4;
    const long long total = (long long)N * h4;
    const float inv_sqrt2 = 0.70710678118654752f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / h4), c = (int)(i - (long long)n * h4);
        const float4* hr = reinterpret_cast<const float4*>(h3 + (size_t)n * 3 * H);
        const float4 a = hr[c], b = hr[h4 + c], g = hr[2 * h4 + c];
        const float4 d = reinterpret_cast<const float4*>(dot + (size_t)n * H)[c];
        float4* xr = reinterpret_cast<float4*>(x + (size_t)n * H) + c;
        float4 xv = *xr;
        xv.x = (xv.x + (a.x + b.x * d.x) * inv_sqrt2) * scale;
        xv.y = (xv.y + (a.y + b.y * d.y) * inv_sqrt2) * scale;
        xv.z = (xv.z + (a.z + b.z * d.z) * inv_sqrt2) * scale;
        xv.w = (xv.w + (a.w + b.w * d.w) * inv_sqrt2) * scale;
        *xr = xv;
        const float4* vb = reinterpret_cast<const float4*>(vv + (size_t)n * 6 * H);
        float4* vr = reinterpret_cast<float4*>(vec + (size_t)n * 3 * H);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float4 v1 = vb[ax * 2 * h4 + c];
            float4 t = vr[ax * h4 + c];
            t.x += g.x * v1.x; t.y += g.y * v1.y; t.z += g.z * v1.z; t.w += g.w * v1.w;
            vr[ax * h4 + c] = t;
        }
    }
}

// GatedEquivariantBlock: cat = [x, ||t1||_xyz]  (torch.norm, no eps; painn_denoising.py:689,692)
__global__ void adf_head_norm_cat_kernel(const float* __restrict__ xin, const float* __restrict__ t1,
                                         float* __restrict__ cat, int N, int C) {
    const int c4 = C / 4;
    const long long total = (long long)N * c4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / c4), c = (int)(i - (long long)n * c4);
        const float4* tb = reinterpret_cast<const float4*>(t1 + (size_t)n * 3 * C);
        const float4 a = tb[c], b = tb[c4 + c], d = tb[2 * c4 + c];
        float4* cr = reinterpret_cast<float4*>(cat + (size_t)n * 2 * C);
        cr[c] = reinterpret_cast<const float4*>(xin + (size_t)n * C)[c];
        cr[c4 + c] = make_float4(sqrtf(a.x * a.x + b.x * b.x + d.x * d.x), sqrtf(a.y * a.y + b.y * b.y + d.y * d.y),
                                 sqrtf(a.z * a.z + b.z * b.z + d.z * d.z), sqrtf(a.w * a.w + b.w * b.w + d.w * d.w));
    }
}

// (x', g) = split(o) ; v' = g (x) t2 ; x' = ssilu(x')   (painn_denoising.py:693-696), Cout % 4 == 0
__global__ void adf_head_gate_kernel(const float* __restrict__ o, const float* __restrict__ t2,
                                     float* __restrict__ xout, float* __restrict__ vout, int N, int Cout) {
    const int c4 = Cout / 4;
    const long long total = (long long)N * c4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / c4), c = (int)(i - (long long)n * c4);
        const float4* orow = reinterpret_cast<const float4*>(o + (size_t)n * 2 * Cout);
        const float4 xs = orow[c], g = orow[c4 + c];
        reinterpret_cast<float4*>(xout + (size_t)n * Cout)[c] =
            make_float4(ssilu_d(xs.x), ssilu_d(xs.y), ssilu_d(xs.z), ssilu_d(xs.w));
        const float4* tb = reinterpret_cast<const float4*>(t2 + (size_t)n * 3 * Cout);
        float4* vb = reinterpret_cast<float4*>(vout + (size_t)n * 3 * Cout);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float4 t = tb[ax * c4 + c];
            vb[ax * c4 + c] = make_float4(g.x * t.x, g.y * t.y, g.z * t.z, g.w * t.w);
        }
    }
}

// x' = ssilu(o[:, :Cout]) only (the gate half of o multiplies vec2_proj's output in that product's epilogue, gemm16.hip)
__global__ void adf_head_xact_kernel(const float* __restrict__ o, float* __restrict__ xout, int N, int Cout) {
    const int c4 = Cout / 4;
    const long long total = (long long)N * c4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / c4), c = (int)(i - (long long)n * c4);
        const float4 xs = reinterpret_cast<const float4*>(o + (size_t)n * 2 * Cout)[c];
        reinterpret_cast<float4*>(xout + (size_t)n * Cout)[c] =
            make_float4(ssilu_d(xs.x), ssilu_d(xs.y), ssilu_d(xs.z), ssilu_d(xs.w));
    }
}

// Last gated block (C -> 1), one wave per atom:
//   vec2 = vec2_proj(v)  [3] ;  g = update_net.2(u)[1] ; out = g * vec2       (painn_denoising.py:690-694,650)
__global__ __launch_bounds__(256) void adf_head_final_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                              const float* __restrict__ w_vec2,
                                                              const float* __restrict__ un2_w,
                                                              const float* __restrict__ un2_b, float* __restrict__ out,
                                                              int N, int C, int32_t* flags) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float g = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float w2 = w_vec2[c];
        g += u[(size_t)n * C + c] * un2_w[C + c];
        t0 += v[((size_t)n * 3 + 0) * C + c] * w2;
        t1 += v[((size_t)n * 3 + 1) * C + c] * w2;
        t2 += v[((size_t)n * 3 + 2) * C + c] * w2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        g += __shfl_xor(g, o); t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o);
    }
    if (lane == 0) {
        g += un2_b[1];
        out[(size_t)n * 3 + 0] = g * t0;
        out[(size_t)n * 3 + 1] = g * t1;
        out[(size_t)n * 3 + 2] = g * t2;
        // Non-finite output: inf / nan inputs or weights - or, in the f16x3 arithmetic, an activation beyond the fp16
        // range (|a| > 65504 -> a_hi = inf), which always ends up here (LayerNorm and the residual streams spread
        // it).  Reported by adf_check_flags as ADF_ENUMERIC; the host re-runs in exact f32 (engine.py).
        const float s3 = g * t0 + g * t1 + g * t2;
        if (!(fabsf(s3) <= 3.0e38f)) atomicExch(&flags[5], 1);
    }
}

static inline unsigned ew_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 256 * 8) b = 256 * 8;  // grid-stride the rest (cdna guide G11)
    if (b < 1) b = 1;
    return (unsigned)b;
}

int32_t adf_nodewise_embed(adf_painn* h, const int32_t* Z, int N, float* x, hipStream_t s) {
    const int H = h->hp.hidden_channels;
    hipLaunchKernelGGL(adf_embed_kernel, dim3(ew_grid((long long)N * H / 4)), dim3(256), 0, s, h->emb, Z, x, N, H,
                       h->hp.num_elements, h->flags);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t adf_nodewise_layernorm(const float* x, const float* w, const float* b, float* y, int N, int H, hipStream_t s,
                               float* out_mag, const int32_t* n_dev) {
    hipLaunchKernelGGL(adf_layernorm_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, w, b, y, N, H, out_mag, n_dev);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t adf_nodewise_update_prep(const float* vv, const float* x, float* cat, float* dot, int N, int H, hipStream_t s) {
    hipLaunchKernelGGL(adf_update_prep_kernel, dim3(ew_grid((long long)N * H / 4)), dim3(256), 0, s, vv, x, cat, dot, N, H);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t adf_nodewise_update_apply(const float* h3, const float* dot, const float* vv, float* x, float* vec,
                                  float scale, int N, int H, hipStream_t s) {
    hipLaunchKernelGGL(adf_update_apply_kernel, dim3(ew_grid((long long)N * H / 4)), dim3(256), 0, s, h3, dot, vv, x,
                       vec, scale, N, H);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// PaiNNOutput for one head.  Scratch reuses the layer buffers (all free after the last layer):
//   vv  = [ t1 (3N x H) | t2 (3N x H/2) | v1 (3N x H/2) ]     cat = [N, 2H]     y = un0 out / u
//   xh  = [ o (N x H) | x1 (N x H/2) | t1' (3N x H/2) ]        dot = cat' [N, H]
int32_t adf_head_forward(adf_painn* h, int head, int N, const float* x, const float* vec, float* out, hipStream_t s) {
    const int H = h->hp.hidden_channels, H2 = H / 2;
    const adf_block_weights& b0 = h->head[head][0];
    const adf_block_weights& b1 = h->head[head][1];
    float* t1 = h->vv;
    float* t2 = h->vv + (size_t)3 * N * H;
    float* v1 = t2 + (size_t)3 * N * H2;
    float* o = h->xh;
    float* x1 = h->xh + (size_t)N * H;
    float* t1b = x1 + (size_t)N * H2;
    float* cat1 = h->dot;
    // row lifts (gemm16.hip): the vec rows feed vec1_proj and vec2_proj of BOTH heads: measured once, by head 0
    const bool lift = h->lift_on && !h->gemm_f32;
    const adf_lift* lf = lift ? &h->lift : nullptr;
    const float* vmag = nullptr;
    if (lift) {
        if (head == 0 || !h->mag_v3_valid) ADF_TRY(adf_launch_rowmag(vec, H, H, nullptr, 0, 3ll * N, h->mag_v3, s));
        h->mag_v3_valid = head == 0;
        vmag = h->mag_v3;
    }
    // block 0: H -> H/2
    if (h->gemm_f32) {
        ADF_TRY(adf_linear(h, vec, H, b0.vec1_w, &b0.vec1_16, nullptr, t1, H, 3 * N, H, H, 0, s));
        hipLaunchKernelGGL(adf_head_norm_cat_kernel, dim3(ew_grid((long long)N * H / 4)), dim3(256), 0, s, x, t1, h->cat, N, H);
    } else {  // ||vec1_proj(vec)|| straight from the accumulators into cat [N,H]
        ADF_TRY(adf_launch_gemm16_vecnorm(vec, H, &b0.vec1_16, h->cat, N, H, H, s, lf, vmag));
    }
    // f16x3 path (round 6): vec2_proj runs AFTER the update net and multiplies its rows by the gate half of `o` in its own
    // epilogue (same single fp32 multiplication per element as adf_head_gate_kernel: same bits); t2 never reaches HBM
    static int fuse_gate = -1;
    if (fuse_gate < 0) { const char* e = getenv("ADF_HEAD_GATE_FUSED"); fuse_gate = (e && atoi(e) == 0) ? 0 : 1; }
    const bool fg = fuse_gate && !h->gemm_f32 && (H2 & 3) == 0;
    if (!fg) ADF_TRY(adf_linear(h, vec, H, b0.vec2_w, &b0.vec2_16, nullptr, t2, H2, 3 * N, H2, H, 0, s, vmag));
    if (h->gemm_f32) ADF_TRY(adf_linear(h, h->cat, 2 * H, b0.un0_w, &b0.un0_16, b0.un0_b, h->y, H, N, H, 2 * H, 1, s));
    else ADF_TRY(adf_launch_gemm16(x, H, &b0.un0_16, b0.un0_b, h->y, H, N, H, 2 * H, 1, s, h->cat, H,  // [x | norm]
                                   h->lift_on ? &h->lift : nullptr, nullptr, h->lift_on ? h->mag_b : nullptr));
    ADF_TRY(adf_linear(h, h->y, H, b0.un2_w, &b0.un2_16, b0.un2_b, o, H, N, H, H, 0, s, (h->lift_on && !h->gemm_f32) ? h->mag_b : nullptr));
    if (fg) {
        // (its epilogue also emits the magnitudes of the gated rows v1: block 1's vector-norm product needs no measuring pass)
        ADF_TRY(adf_launch_gemm16(vec, H, &b0.vec2_16, nullptr, v1, H2, 3 * N, H2, H, 0, s, nullptr, 0, lf, vmag,
                                  lift ? h->lift.buf : nullptr, nullptr, 0, o + H2, H));
        hipLaunchKernelGGL(adf_head_xact_kernel, dim3(ew_grid((long long)N * H2 / 4)), dim3(256), 0, s, o, x1, N, H2);
    } else
    hipLaunchKernelGGL(adf_head_gate_kernel, dim3(ew_grid((long long)N * H2 / 4)), dim3(256), 0, s, o, t2, x1, v1, N, H2);
    // block 1: H/2 -> 1
    if (h->gemm_f32) {
        ADF_TRY(adf_linear(h, v1, H2, b1.vec1_w, &b1.vec1_16, nullptr, t1b, H2, 3 * N, H2, H2, 0, s));
        hipLaunchKernelGGL(adf_head_norm_cat_kernel, dim3(ew_grid((long long)N * H2 / 4)), dim3(256), 0, s, x1, t1b, cat1, N, H2);
        ADF_TRY(adf_linear(h, cat1, H, b1.un0_w, &b1.un0_16, b1.un0_b, h->y, H2, N, H2, H, 1, s));
    } else {
        ADF_TRY(adf_launch_gemm16_vecnorm(v1, H2, &b1.vec1_16, cat1, N, H2, H2, s, h->lift_on ? &h->lift : nullptr,
                                          (fg && lift) ? h->lift.buf : nullptr));
        ADF_TRY(adf_launch_gemm16(x1, H2, &b1.un0_16, b1.un0_b, h->y, H2, N, H2, H, 1, s, cat1, H2, h->lift_on ? &h->lift : nullptr));
    }
    hipLaunchKernelGGL(adf_head_final_kernel, dim3((N + 3) / 4), dim3(256), 0, s, h->y, v1, b1.vec2_w, b1.un2_w,
                       b1.un2_b, out, N, H2, h->flags);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
