// EquiformerV2 denoiser: handle, weight binding, forward orchestration and the C ABI (include/adsorbdiff_hip.h,
// "EquiformerV2 denoiser").  Reference: models/equiformer_v2/equiformer_v2_denoising.py:185-318 (forward),
// transformer_block.py:226-372 (SO2EquivariantGraphAttention.forward), :473-531 (FeedForwardNetwork.forward),
// :650-728 (TransBlockV2.forward), input_block.py:84-138 (EdgeDegreeEmbedding.forward).
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "eqv2.h"

int32_t eq_gemm_f32(const float* A, int lda, const eq_rowmap* amap, const float* W, const float* bias, float* Cm, int ldc,
                    const eq_rowmap* cmap, long long M, int N, int K, int act, bool accumulate, hipStream_t s);
int32_t eq_launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t s);
bool eq_gemm16_ok(const float* A, const eq_rowmap* am, const float* Cm, const eq_rowmap* cm, int N, int K);
int32_t eq_launch_rowscale(const float* A, const eq_rowmap* am, long long M, int K, float* rs, hipStream_t s);
int32_t eq_launch_gemm16(const float* A, const eq_rowmap* am, const float* rscale, const adf_w16* W, const float* bias,
                         float* Cm, const eq_rowmap* cm, long long M, int N, int K, int act, bool accumulate,
                         hipStream_t s, float* out_mag, int rs_div = 1);

template <typename T>
static int32_t eq_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        adf_set_error("eqv2: device allocation of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
        return ADF_EOOM;
    }
    return ADF_OK;
}
template <typename T>
static void eq_free(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

// ---------------------------------------------------------------------------------------------- profiling
static void eq_prof_begin(adf_eqv2* h, int cat, hipStream_t s) {
    if (!h->prof_on) return;
    if (h->prof_used + 2 > h->prof_ev->size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            h->prof_ev->push_back(e);
        }
    }
    h->prof_cat->push_back(cat);
    (void)hipEventRecord((*h->prof_ev)[h->prof_used], s);
}
static void eq_prof_end(adf_eqv2* h, hipStream_t s) {
    if (!h->prof_on || h->prof_used + 2 > h->prof_ev->size()) return;
    (void)hipEventRecord((*h->prof_ev)[h->prof_used + 1], s);
    h->prof_used += 2;
}
struct eq_prof_scope {
    adf_eqv2* h; hipStream_t s;
    eq_prof_scope(adf_eqv2* h_, int cat, hipStream_t s_) : h(h_), s(s_) { eq_prof_begin(h, cat, s); }
    ~eq_prof_scope() { eq_prof_end(h, s); }
};

// ---------------------------------------------------------------------------------------------- create / destroy
static void eq_fill_dims(adf_eqv2* h) {
    const adf_eqv2_hparams& hp = h->hp;
    eq_dims& d = h->d;
    memset(&d, 0, sizeof(d));
    d.L = hp.lmax; d.M = hp.mmax; d.C = hp.sphere_channels; d.S = (d.L + 1) * (d.L + 1);
    d.G = hp.grid_resolution * hp.grid_resolution;
    d.Hd = hp.attn_hidden_channels; d.NH = hp.num_heads; d.A = hp.attn_alpha_channels; d.V = hp.attn_value_channels;
    d.HV = d.NH * d.V; d.F = hp.ffn_hidden_channels; d.EC = hp.edge_channels; d.NB = hp.num_distance_basis;
    int off = 0, joff = 0;
    for (int l = 0; l <= d.L; ++l) {
        const int ml = l < d.M ? l : d.M;
        d.d_off[l] = off;
        d.nrow[l] = 2 * ml + 1;
        off += d.nrow[l] * (2 * l + 1);
        d.j_off[l] = joff;
        joff += (2 * l + 1) * (2 * l + 1);
        d.resc[l] = l > d.M ? sqrtf((float)(2 * l + 1) / (float)(2 * d.M + 1)) : 1.0f;
    }
    d.d_off[d.L + 1] = off; d.DR = off; d.j_off[d.L + 1] = joff;
    int r = 0, ro = 0;
    for (int m = 0; m <= d.M; ++m) {
        d.rbase[m] = r;
        d.rad_off[m] = ro;
        const int nm = d.L - m + 1;
        ro += nm;
        for (int sg = 0; sg < (m == 0 ? 1 : 2); ++sg)
            for (int l = m; l <= d.L; ++l) { d.r_m[r] = (short)m; d.r_sgn[r] = (short)sg; d.r_l[r] = (short)l; ++r; }
    }
    d.Sr = r; d.RW = ro;
}

extern "C" int32_t adf_eqv2_create(const adf_eqv2_hparams* hp, adf_eqv2_t* out) {
    if (!hp || !out) { adf_set_error("eqv2_create: null argument"); return ADF_EINVAL; }
    if (hp->lmax < 1 || hp->lmax > EQ_MAX_L || hp->mmax < 0 || hp->mmax > hp->lmax || hp->num_layers < 0 ||
        hp->num_layers > EQ_MAX_LAYERS || hp->sphere_channels < 1 || hp->sphere_channels > 512 || hp->num_heads < 1 ||
        hp->num_heads > 16 || hp->attn_hidden_channels < 1 || hp->attn_hidden_channels > 256 ||
        hp->attn_alpha_channels < 1 || hp->attn_value_channels < 1 || hp->num_heads * hp->attn_value_channels > 1024 ||
        hp->ffn_hidden_channels < 1 || hp->ffn_hidden_channels > 1024 || hp->edge_channels < 1 || hp->edge_channels > 1024 ||
        hp->grid_resolution < 2 || (hp->grid_resolution & 1) || hp->num_distance_basis < 2 || hp->max_neighbors < 1 ||
        hp->max_neighbors > ADF_MAX_K || !(hp->max_radius > 0.f) || !(hp->avg_degree > 0.f) || hp->max_num_elements < 1) {
        adf_set_error("eqv2_create: unsupported hyper-parameters");
        return ADF_EINVAL;
    }
    adf_eqv2* h = new (std::nothrow) adf_eqv2();
    if (!h) { adf_set_error("eqv2_create: host allocation failed"); return ADF_EOOM; }
    memset(static_cast<void*>(h), 0, sizeof(*h));
    h->hp = *hp;
    eq_fill_dims(h);
    if ((2 * h->d.M + 1) * (2 * h->d.L + 1) > 13 * 13 || h->d.Sr > 49) { delete h; adf_set_error("eqv2_create: lmax / mmax too large"); return ADF_EINVAL; }
    ADF_HIP_CHECK(hipGetDevice(&h->device));
    hipDeviceProp_t prop;
    ADF_HIP_CHECK(hipGetDeviceProperties(&prop, h->device));
    h->num_cus = prop.multiProcessorCount;
    const char* env = getenv("ADF_GEMM");
    h->exact_f32 = env && !strcmp(env, "f32");
    // the S2 activation can hand the magnitudes of its output rows to the second convolution (atomics in its epilogue)
    // instead of a separate pass over them: measured slower on MI355X (+8 ms vs -3 ms per forward at 256 k edges): off
    { const char* e2 = getenv("ADF_EQV2_S2_EMIT"); h->s2_emit_mag = e2 && atoi(e2) != 0; }
    { const char* e3 = getenv("ADF_EQV2_PRESPLIT"); h->presplit = !(e3 && atoi(e3) == 0); }
    { const char* e4 = getenv("ADF_EQV2_CONV1_WR"); h->conv1_wr = !(e4 && atoi(e4) == 0); }
    { const char* e5 = getenv("ADF_EQV2_CONV2_WR"); h->conv2_wr = !(e5 && atoi(e5) == 0); }
    { const char* e6 = getenv("ADF_EQV2_ALPHA_GENERIC"); h->alpha_generic = e6 && atoi(e6) != 0; }
    { const char* e4 = getenv("ADF_EQV2_FOLD"); h->fold_on = !(e4 && atoi(e4) == 0); }
    { const char* e5 = getenv("ADF_EQV2_COMPACT"); h->no_compact = e5 && atoi(e5) == 0; }
    h->prof_ev = new std::vector<hipEvent_t>();
    h->prof_cat = new std::vector<int>();
    int32_t st = eq_alloc(&h->flags, EQ_NFLAGS);
    if (st == ADF_OK) st = eq_alloc(&h->d_dev, 1);
    if (st == ADF_OK && hipMemcpy(h->d_dev, &h->d, sizeof(eq_dims), hipMemcpyHostToDevice) != hipSuccess) st = ADF_EHIP;
    if (st == ADF_OK) { hipError_t e = hipMemset(h->flags, 0, sizeof(int32_t) * EQ_NFLAGS); if (e != hipSuccess) st = ADF_EHIP; }
    if (st != ADF_OK) { adf_eqv2_destroy(h); return st; }
    *out = h;
    return ADF_OK;
}

static void eq_inc_free(adf_eqv2* h) {
    for (int i = 0; i <= EQ_MAX_LAYERS; ++i) eq_free(h->inc_x[i]);
    eq_free(h->inc_xg); eq_free(h->inc_peptr); eq_free(h->inc_psrc); eq_free(h->inc_pvec); eq_free(h->inc_dirty);
    eq_free(h->inc_idx); eq_free(h->inc_cnt);
    if (h->inc_sel_tmp) { (void)hipFree(h->inc_sel_tmp); h->inc_sel_tmp = nullptr; }
    h->inc_capN = h->inc_capE = 0; h->inc_nl = 0; h->inc_sel_bytes = 0;
    h->inc_valid = false;
}

static void eq_free_workspaces(adf_eqv2* h) {
    eq_inc_free(h);
    eq_free(h->nbr_cnt); eq_free(h->nbr_src); eq_free(h->nbr_shift); eq_free(h->img_cnt); eq_free(h->eptr);
    eq_free(h->e_src); eq_free(h->e_dst); eq_free(h->e_vec); eq_free(h->wig);
    eq_free(h->sub_eptr); eq_free(h->sub_src); eq_free(h->sub_dst); eq_free(h->sub_vec); eq_free(h->sub_wig); eq_free(h->sub_f);
    h->sub_cap = 0;
    if (h->scan_tmp) { (void)hipFree(h->scan_tmp); h->scan_tmp = nullptr; }
    eq_free(h->cache_d2); eq_free(h->cache_cid); eq_free(h->cache_cnt);
    eq_free(h->x); eq_free(h->y); eq_free(h->agg); eq_free(h->gate); eq_free(h->h1); eq_free(h->h2);
    eq_free(h->arena); eq_free(h->garena); eq_free(h->sys); eq_free(h->rs);
    h->rs_cap = 0;
    h->capN = h->capB = h->capE = 0;
    h->arena_floats = h->garena_floats = 0;
}

extern "C" int32_t adf_eqv2_destroy(adf_eqv2_t h) {
    if (!h) return ADF_OK;
    (void)hipDeviceSynchronize();
    eq_free_workspaces(h);
    eq_free(h->flags); eq_free(h->d_dev);
    eq_free(h->xe_src); eq_free(h->xe_dst); eq_free(h->xe_vec);
    { unsigned char* t = (unsigned char*)h->s2tab; eq_free(t); h->s2tab = nullptr; }
    { unsigned char* t = (unsigned char*)h->gtab_to; eq_free(t); h->gtab_to = nullptr; }
    { unsigned char* t = (unsigned char*)h->gtab_from; eq_free(t); h->gtab_from = nullptr; }
    eq_free(h->jd); eq_free(h->to_red); eq_free(h->from_red); eq_free(h->to_full); eq_free(h->from_full);
    eq_free(h->w16_arena); eq_free(h->wfrag_arena); eq_free(h->w16_scales); eq_free(h->w16_scratch); eq_free(h->wt_arena); eq_free(h->rtab_arena); eq_free(h->fold_arena);
    if (h->prof_ev) { for (hipEvent_t e : *h->prof_ev) (void)hipEventDestroy(e); delete h->prof_ev; }
    delete h->prof_cat;
    delete h;
    return ADF_OK;
}

extern "C" int32_t adf_eqv2_set_constants(adf_eqv2_t h, const float* jd, const float* to_red, const float* from_red,
                                          const float* to_full, const float* from_full) {
    if (!h || !jd || !to_red || !from_red || !to_full || !from_full) { adf_set_error("eqv2_set_constants: null argument"); return ADF_EINVAL; }
    const eq_dims& d = h->d;
    const size_t nj = d.j_off[d.L + 1], nr = (size_t)d.G * d.Sr, nf = (size_t)d.G * d.S;
    eq_free(h->jd); eq_free(h->to_red); eq_free(h->from_red); eq_free(h->to_full); eq_free(h->from_full);
    { unsigned char* t = (unsigned char*)h->s2tab; eq_free(t); h->s2tab = nullptr; }
    { unsigned char* t = (unsigned char*)h->gtab_to; eq_free(t); h->gtab_to = nullptr; }
    { unsigned char* t = (unsigned char*)h->gtab_from; eq_free(t); h->gtab_from = nullptr; }
    ADF_TRY(eq_alloc(&h->jd, nj)); ADF_TRY(eq_alloc(&h->to_red, nr)); ADF_TRY(eq_alloc(&h->from_red, nr));
    ADF_TRY(eq_alloc(&h->to_full, nf)); ADF_TRY(eq_alloc(&h->from_full, nf));
    ADF_HIP_CHECK(hipMemcpy(h->jd, jd, nj * 4, hipMemcpyHostToDevice));
    ADF_HIP_CHECK(hipMemcpy(h->to_red, to_red, nr * 4, hipMemcpyHostToDevice));
    ADF_HIP_CHECK(hipMemcpy(h->from_red, from_red, nr * 4, hipMemcpyHostToDevice));
    ADF_HIP_CHECK(hipMemcpy(h->to_full, to_full, nf * 4, hipMemcpyHostToDevice));
    ADF_HIP_CHECK(hipMemcpy(h->from_full, from_full, nf * 4, hipMemcpyHostToDevice));
    // fragment-order fp16 hi/lo images of to_red / from_red for the matrix-core S2 activation (eqv2_kernels.hip)
    if (d.Sr <= 32) {
        const int npb = (d.G + 31) / 32;
        float tmax = 0.f, fmax_ = 0.f, gain = 0.f;
        for (int p = 0; p < d.G; ++p) {
            float row = 0.f;
            for (int r = 0; r < d.Sr; ++r) {
                const float t = fabsf(to_red[(size_t)p * d.Sr + r]);
                row += t;
                tmax = t > tmax ? t : tmax;
                const float f = fabsf(from_red[(size_t)p * d.Sr + r]);
                fmax_ = f > fmax_ ? f : fmax_;
            }
            gain = row > gain ? row : gain;
        }
        auto pow2_for = [](float amax) { int e = 0; if (amax > 0.f) (void)frexpf(amax, &e); return ldexpf(1.0f, 10 - e); };
        const float sT = pow2_for(tmax), sF = pow2_for(fmax_);
        int ge = 0;
        if (gain > 1.0f) { (void)frexpf(gain, &ge); if (ldexpf(1.0f, ge - 1) >= gain) --ge; }
        const size_t nh8 = (size_t)npb * 2 * 2 * 64;  // half8 entries per table
        std::vector<_Float16> img(2 * nh8 * 8);
        for (int pb = 0; pb < npb; ++pb)
            for (int ks = 0; ks < 2; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int kh = lane >> 5, q = lane & 31;
                        // first product, A = to_grid: row = grid point 32 pb + q, k = coefficient 16 ks + 8 kh + j
                        const int p1 = 32 * pb + q, r1 = 16 * ks + 8 * kh + j;
                        const float tv = (p1 < d.G && r1 < d.Sr) ? to_red[(size_t)p1 * d.Sr + r1] * sT : 0.f;
                        // second product, A = from_grid^T: row = coefficient q, k = grid point in accumulator-row order
                        const int p2 = 32 * pb + 16 * ks + 8 * (j >> 2) + 4 * kh + (j & 3), r2 = q;
                        const float fv = (p2 < d.G && r2 < d.Sr) ? from_red[(size_t)p2 * d.Sr + r2] * sF : 0.f;
                        const size_t base = ((size_t)(pb * 2 + ks) * 2) * 64;
                        const _Float16 th = (_Float16)tv, fh = (_Float16)fv;
                        img[((base + lane) * 8) + j] = th;
                        img[((base + 64 + lane) * 8) + j] = (_Float16)(tv - (float)th);
                        img[((nh8 + base + lane) * 8) + j] = fh;
                        img[((nh8 + base + 64 + lane) * 8) + j] = (_Float16)(fv - (float)fh);
                    }
        unsigned char* dev = nullptr;
        ADF_TRY(eq_alloc(&dev, img.size() * 2));
        ADF_HIP_CHECK(hipMemcpy(dev, img.data(), img.size() * 2, hipMemcpyHostToDevice));
        h->s2tab = dev; h->s2_npb = npb; h->s2_inv_sT = 1.0f / sT; h->s2_inv_sF = 1.0f / sF;
        h->s2_gain_shift = ldexpf(1.0f, -ge);
    }
    // and of to_full / from_full for the feed-forward grid transforms
    if (d.S <= 64) {
        const int npb = (d.G + 31) / 32, nkst = (d.G + 15) / 16;
        float tmax = 0.f, fmx = 0.f;
        for (size_t i = 0; i < nf; ++i) { tmax = fmaxf(tmax, fabsf(to_full[i])); fmx = fmaxf(fmx, fabsf(from_full[i])); }
        auto pow2_for = [](float amax) { int e = 0; if (amax > 0.f) (void)frexpf(amax, &e); return ldexpf(1.0f, 10 - e); };
        const float sT = pow2_for(tmax), sF = pow2_for(fmx);
        std::vector<_Float16> ti((size_t)npb * 4 * 2 * 64 * 8), fi((size_t)nkst * 2 * 2 * 64 * 8);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int kh = lane >> 5, q = lane & 31;
                for (int pb = 0; pb < npb; ++pb)
                    for (int ks = 0; ks < 4; ++ks) {
                        const int pp = 32 * pb + q, sidx = 16 * ks + 8 * kh + j;
                        const float v = (pp < d.G && sidx < d.S) ? to_full[(size_t)pp * d.S + sidx] * sT : 0.f;
                        const size_t base = ((size_t)(pb * 4 + ks) * 2) * 64;
                        const _Float16 hh = (_Float16)v;
                        ti[(base + lane) * 8 + j] = hh;
                        ti[(base + 64 + lane) * 8 + j] = (_Float16)(v - (float)hh);
                    }
                for (int kst = 0; kst < nkst; ++kst)
                    for (int sb = 0; sb < 2; ++sb) {
                        const int pp = 16 * kst + 8 * kh + j, sidx = 32 * sb + q;
                        const float v = (pp < d.G && sidx < d.S) ? from_full[(size_t)pp * d.S + sidx] * sF : 0.f;
                        const size_t base = ((size_t)(kst * 2 + sb) * 2) * 64;
                        const _Float16 hh = (_Float16)v;
                        fi[(base + lane) * 8 + j] = hh;
                        fi[(base + 64 + lane) * 8 + j] = (_Float16)(v - (float)hh);
                    }
            }
        unsigned char *dt = nullptr, *df = nullptr;
        ADF_TRY(eq_alloc(&dt, ti.size() * 2)); ADF_TRY(eq_alloc(&df, fi.size() * 2));
        ADF_HIP_CHECK(hipMemcpy(dt, ti.data(), ti.size() * 2, hipMemcpyHostToDevice));
        ADF_HIP_CHECK(hipMemcpy(df, fi.data(), fi.size() * 2, hipMemcpyHostToDevice));
        h->gtab_to = dt; h->gtab_from = df; h->g_npb = npb; h->g_nkst = nkst; h->g_inv_sT = 1.0f / sT; h->g_inv_sF = 1.0f / sF;
    }
    h->consts_set = true;
    return ADF_OK;
}

// ---------------------------------------------------------------------------------------------- weights
struct eq_wcursor {
    const void* const* w; int n, i;
    const float* next() { return i < n ? static_cast<const float*>(w[i++]) : (i++, nullptr); }
};
static eq_lin mk_lin(const float* w, const float* b, int out, int in) {
    eq_lin l; memset(&l, 0, sizeof(l));
    l.w = w; l.b = b; l.out = out; l.in = in;
    return l;
}
static void bind_radial(eq_wcursor& c, eq_radial* r, int in0, int ec, int out) {
    const float* w0 = c.next(); const float* b0 = c.next();
    r->ln1_w = c.next(); r->ln1_b = c.next();
    const float* w3 = c.next(); const float* b3 = c.next();
    r->ln4_w = c.next(); r->ln4_b = c.next();
    const float* w6 = c.next(); const float* b6 = c.next();
    r->l0 = mk_lin(w0, b0, ec, in0); r->l3 = mk_lin(w3, b3, ec, ec); r->l6 = mk_lin(w6, b6, out, ec);
}
static void bind_norm(eq_wcursor& c, eq_norm* n) { n->affine = c.next(); n->l0_w = c.next(); n->l0_b = c.next(); }
static void bind_attn(eq_wcursor& c, eq_attn* a, const eq_dims& d, int out_channels) {
    a->out_channels = out_channels;
    a->alpha_dot = c.next(); a->src_emb = c.next(); a->dst_emb = c.next();
    const int extra = d.NH * d.A + d.Hd;
    const float* w = c.next(); const float* b = c.next();
    a->c1_m0 = mk_lin(w, b, extra + (d.L + 1) * d.Hd, (d.L + 1) * 2 * d.C);
    for (int m = 1; m <= d.M; ++m) { const int nm = d.L - m + 1; a->c1_m[m - 1] = mk_lin(c.next(), nullptr, 2 * nm * d.Hd, nm * 2 * d.C); }
    bind_radial(c, &a->rad, d.NB + 2 * d.EC, d.EC, d.RW * 2 * d.C);
    a->alpha_ln_w = c.next(); a->alpha_ln_b = c.next();
    w = c.next(); b = c.next();
    a->c2_m0 = mk_lin(w, b, (d.L + 1) * d.HV, (d.L + 1) * d.Hd);
    for (int m = 1; m <= d.M; ++m) { const int nm = d.L - m + 1; a->c2_m[m - 1] = mk_lin(c.next(), nullptr, 2 * nm * d.HV, nm * d.Hd); }
    a->proj_w = c.next(); a->proj_b = c.next();
    for (int l = 0; l <= d.L; ++l)
        a->proj_l[l] = mk_lin(a->proj_w ? a->proj_w + (size_t)l * out_channels * d.HV : nullptr, l == 0 ? a->proj_b : nullptr, out_channels, d.HV);
}

static int eq_expected_weights(const eq_dims& d, int layers) {
    const int attn = 21 + 2 * d.M;
    return 14 + layers * (3 + attn + 3 + 9) + 3 + 2 * attn;
}

// fp16 hi/lo images (per-matrix power-of-two scale, gemm16.hip) of every weight a dense product reads
static void eq_collect_lins(adf_eqv2* h, std::vector<eq_lin*>& v) {
    const eq_dims& d = h->d;
    auto rad = [&](eq_radial* r) { v.push_back(&r->l3); v.push_back(&r->l6); };
    auto attn = [&](eq_attn* a, bool with_proj) {
        v.push_back(&a->c1_m0); v.push_back(&a->c2_m0);
        for (int m = 1; m <= d.M; ++m) { v.push_back(&a->c1_m[m - 1]); v.push_back(&a->c2_m[m - 1]); }
        rad(&a->rad);
        if (with_proj) for (int l = 0; l <= d.L; ++l) v.push_back(&a->proj_l[l]);
    };
    rad(&h->ed_rad);
    for (int i = 0; i < h->hp.num_layers; ++i) {
        eq_block& b = h->blk[i];
        attn(&b.ga, true);
        v.push_back(&b.ffn.scalar); v.push_back(&b.ffn.g0); v.push_back(&b.ffn.g2); v.push_back(&b.ffn.g4);
        for (int l = 0; l <= d.L; ++l) { v.push_back(&b.ffn.l1[l]); v.push_back(&b.ffn.l2[l]); }
        if (h->folded)
            for (int l = 0; l <= d.L; ++l) { v.push_back(&b.ffn.l1f[l]); if (l) v.push_back(&b.ffn.l2f[l]); }
    }
    attn(&h->force[0], false); attn(&h->force[1], false);
}

static int32_t eq_split_weights(adf_eqv2* h, hipStream_t s) {
    std::vector<eq_lin*> v;
    eq_collect_lins(h, v);
    size_t halves = 0, nmat = 0;
    for (eq_lin* l : v) {
        l->has16 = false;
        if (l->in % 32 != 0 || (l->out & 3)) continue;
        halves += 2 * (size_t)l->out * l->in;
        ++nmat;
    }
    if (h->w16_bytes < halves * 2 + 64) {
        ADF_HIP_CHECK(hipDeviceSynchronize());
        eq_free(h->w16_arena); eq_free(h->wfrag_arena);
        ADF_TRY(eq_alloc(&h->w16_arena, halves * 2 + 64));
        ADF_TRY(eq_alloc(&h->wfrag_arena, halves * 2 + 64));
        h->w16_bytes = halves * 2 + 64;
    }
    eq_free(h->w16_scales);
    ADF_TRY(eq_alloc(&h->w16_scales, nmat + 1));
    if (!h->w16_scratch) ADF_TRY(eq_alloc(&h->w16_scratch, 4));
    unsigned char* p = h->w16_arena;
    unsigned char* pf = h->wfrag_arena;
    size_t k = 0;
    for (eq_lin* l : v) {
        if (l->in % 32 != 0 || (l->out & 3)) continue;
        const size_t n = (size_t)l->out * l->in;
        l->w16.hi = p; l->w16.lo = p + n * 2; l->w16.inv_scale = h->w16_scales + k; l->w16.bias_perm = nullptr;
        l->w16.frag = nullptr;
        ADF_TRY(adf_split_weight(l->w, (long long)n, &l->w16, h->w16_scratch, s));
        if (l->out % 32 == 0) {   // the MFMA B operands in fragment order (eq_gemm16pw_kernel)
            ADF_TRY(adf_pack_frag(&l->w16, l->out, l->in, pf, s));
            l->w16.frag = pf;
        }
        p += n * 4; pf += n * 4; ++k;
        l->has16 = true;
    }
    return ADF_OK;
}

// Static-radial mode.  The Gaussian basis of an edge is exp(coeff (d - r_s - r_t - mu_k)^2) with d in (0.01, rc]; if for
// EVERY pair of elements with finite radii the window of non-vanishing terms (|d - r_s - r_t - mu_k| < tmax,
// eq_radial_pre_kernel) is empty for every possible d, the radial MLP of an edge depends on (Z_s, Z_t) only and is
// tabulated once per weight binding (NE^2 rows per radial function).  With the reference's radii (pm subtracted from
// Angstrom, equiformer_v2_denoising.py:165-213) this always holds; with radii in Angstrom it would not and every edge
// is evaluated.  ADF_EQV2_RADIAL=edge forces the per-edge evaluation (parity tests compare both).
static int32_t eq_radial_static(adf_eqv2* h, eq_radial** rads, int nrad, hipStream_t s) {
    const eq_dims& d = h->d;
    const int NE = h->hp.max_num_elements;
    h->rad_static = false;
    for (int i = 0; i < nrad; ++i) rads[i]->table = nullptr;
    const char* env = getenv("ADF_EQV2_RADIAL");
    if (env && !strcmp(env, "edge")) return ADF_OK;
    float radii[101];
    ADF_HIP_CHECK(hipMemcpyAsync(radii, h->atom_radii, sizeof(radii), hipMemcpyDeviceToHost, s));
    ADF_HIP_CHECK(hipStreamSynchronize(s));
    const float rc = h->hp.max_radius, delta = rc / (float)(d.NB - 1);
    const float coeff = -0.5f / ((2.0f * delta) * (2.0f * delta)), tmax = sqrtf(104.0f / -coeff);
    for (int a = 0; a < NE && a <= 100; ++a)
        for (int b = a; b < NE && b <= 100; ++b) {
            const float rr = radii[a] + radii[b];
            if (!(rr == rr)) continue;  // NaN output either way
            const float lo = 0.0f - rr, hi = rc - rr;  // range of dd = d - r_a - r_b
            if (hi + tmax >= 0.f && lo - tmax <= rc) return ADF_OK;  // some d reaches a basis function: per-edge mode
        }
    size_t total = 0;
    for (int i = 0; i < nrad; ++i) total += (size_t)NE * NE * rads[i]->l6.out;
    if (h->rtab_floats < total) {
        ADF_HIP_CHECK(hipDeviceSynchronize());
        eq_free(h->rtab_arena);
        if (eq_alloc(&h->rtab_arena, total) != ADF_OK) { h->rtab_floats = 0; return ADF_OK; }  // no room: per-edge mode
        h->rtab_floats = total;
    }
    float *t1 = nullptr, *t2 = nullptr, *rs = nullptr;
    const long long rows = (long long)NE * NE;
    if (eq_alloc(&t1, (size_t)rows * d.EC) != ADF_OK || eq_alloc(&t2, (size_t)rows * d.EC) != ADF_OK ||
        eq_alloc(&rs, (size_t)rows) != ADF_OK) {  // no room for the scratch rows: per-edge mode
        eq_free(t1); eq_free(t2); eq_free(rs);
        (void)hipGetLastError();
        return ADF_OK;
    }
    float* keep_rs = h->rs; const int64_t keep_cap = h->rs_cap;
    h->rs = rs; h->rs_cap = rows;
    // The tables are built ONCE per weight binding and are then read by both arithmetics (adf_eqv2_set_arithmetic only flips
    // a flag), so they are always evaluated in exact f32: an exact-f32 forward must not multiply by f16x3-evaluated tables
    // (NE^2 rows: the cost is negligible).
    const bool keep_exact = h->exact_f32;
    h->exact_f32 = true;
    float* p = h->rtab_arena;
    int32_t st = ADF_OK;
    for (int i = 0; i < nrad && st == ADF_OK; ++i) {
        eq_radial* r = rads[i];
        const float* semb = i == 0 ? h->ed_src_emb : (i <= h->hp.num_layers ? h->blk[i - 1].ga.src_emb : h->force[i - 1 - h->hp.num_layers].src_emb);
        const float* temb = i == 0 ? h->ed_dst_emb : (i <= h->hp.num_layers ? h->blk[i - 1].ga.dst_emb : h->force[i - 1 - h->hp.num_layers].dst_emb);
        st = eq_launch_radial_pre_pairs(h, r, semb, temb, t1, s);
        if (st == ADF_OK) st = eq_launch_ln_silu(t1, r->ln1_w, r->ln1_b, rows, d.EC, s);
        if (st == ADF_OK) st = eq_gemm(h, t1, d.EC, nullptr, &r->l3, true, t2, d.EC, nullptr, rows, 0, false, s);
        if (st == ADF_OK) st = eq_launch_ln_silu(t2, r->ln4_w, r->ln4_b, rows, d.EC, s);
        if (st == ADF_OK) st = eq_gemm(h, t2, d.EC, nullptr, &r->l6, true, p, r->l6.out, nullptr, rows, 0, false, s);
        r->table = p;
        p += (size_t)rows * r->l6.out;
    }
    (void)hipStreamSynchronize(s);
    h->exact_f32 = keep_exact;
    h->rs = keep_rs; h->rs_cap = keep_cap;
    eq_free(t1); eq_free(t2); eq_free(rs);
    if (st != ADF_OK) { for (int i = 0; i < nrad; ++i) rads[i]->table = nullptr; return st; }
    h->rad_static = true;
    return ADF_OK;
}

// Folded feed-forward weights (eq_ffn): l1f[l] = g0 . l1[l] (+ bias g0 . b on l = 0), l2f[l] = l2[l] . g4 for l >= 1.
static int32_t eq_fold_ffn(adf_eqv2* h, hipStream_t s) {
    const eq_dims& d = h->d;
    h->folded = false;
    if (!h->fold_on) return ADF_OK;
    const size_t per = (size_t)(d.L + 1) * d.F * d.C + d.F + (size_t)d.L * d.C * d.F;
    const size_t total = per * h->hp.num_layers;
    if (h->fold_floats < total) {
        ADF_HIP_CHECK(hipDeviceSynchronize());
        eq_free(h->fold_arena);
        h->fold_floats = 0;
        ADF_TRY(eq_alloc(&h->fold_arena, total));
        h->fold_floats = total;
    }
    float* p = h->fold_arena;
    for (int i = 0; i < h->hp.num_layers; ++i) {
        eq_ffn& f = h->blk[i].ffn;
        float* b0 = p + (size_t)(d.L + 1) * d.F * d.C;
        for (int l = 0; l <= d.L; ++l) {
            float* w = p + (size_t)l * d.F * d.C;
            ADF_TRY(eq_launch_fold(f.g0.w, f.l1[l].w, d.F, d.F, d.C, w, s));
            f.l1f[l] = mk_lin(w, l == 0 ? b0 : nullptr, d.F, d.C);
        }
        ADF_TRY(eq_launch_fold(f.g0.w, f.l1_b, d.F, d.F, 1, b0, s));
        float* q = b0 + d.F;
        for (int l = 1; l <= d.L; ++l) {
            float* w = q + (size_t)(l - 1) * d.C * d.F;
            ADF_TRY(eq_launch_fold(f.l2[l].w, f.g4.w, d.C, d.F, d.F, w, s));
            f.l2f[l] = mk_lin(w, nullptr, d.C, d.F);
        }
        p += per;
    }
    h->folded = true;
    return ADF_OK;
}

extern "C" int32_t adf_eqv2_set_weights(adf_eqv2_t h, int32_t n_weights, const void* const* weights, void* stream) {
    if (!h || !weights) { adf_set_error("eqv2_set_weights: null argument"); return ADF_EINVAL; }
    const eq_dims& d = h->d;
    const int want = eq_expected_weights(d, h->hp.num_layers);
    if (n_weights != want) { adf_set_error("eqv2_set_weights: expected %d tensors, got %d", want, n_weights); return ADF_EINVAL; }
    for (int i = 0; i < n_weights; ++i)
        if (!weights[i]) { adf_set_error("eqv2_set_weights: tensor %d is null", i); return ADF_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    eq_wcursor c{weights, n_weights, 0};
    h->atom_radii = c.next();
    h->sphere_emb = c.next();
    h->ed_src_emb = c.next(); h->ed_dst_emb = c.next();
    bind_radial(c, &h->ed_rad, d.NB + 2 * d.EC, d.EC, (d.L + 1) * d.C);
    for (int i = 0; i < h->hp.num_layers; ++i) {
        eq_block& b = h->blk[i];
        bind_norm(c, &b.n1);
        bind_attn(c, &b.ga, d, d.C);
        bind_norm(c, &b.n2);
        eq_ffn& f = b.ffn;
        f.l1_w = c.next(); f.l1_b = c.next();
        const float* sw = c.next(); const float* sb = c.next();
        f.scalar = mk_lin(sw, sb, d.F, d.C);
        f.g0 = mk_lin(c.next(), nullptr, d.F, d.F); f.g2 = mk_lin(c.next(), nullptr, d.F, d.F); f.g4 = mk_lin(c.next(), nullptr, d.F, d.F);
        f.l2_w = c.next(); f.l2_b = c.next();
        for (int l = 0; l <= d.L; ++l) {
            f.l1[l] = mk_lin(f.l1_w + (size_t)l * d.F * d.C, l == 0 ? f.l1_b : nullptr, d.F, d.C);
            f.l2[l] = mk_lin(f.l2_w + (size_t)l * d.C * d.F, l == 0 ? f.l2_b : nullptr, d.C, d.F);
        }
    }
    bind_norm(c, &h->final_norm);
    bind_attn(c, &h->force[0], d, 1);
    bind_attn(c, &h->force[1], d, 1);
    // transposed first radial layers (one arena)
    const int nrad = 1 + h->hp.num_layers + 2;
    const size_t per = (size_t)(d.NB + 2 * d.EC) * d.EC;
    if (h->wt_bytes < per * nrad * 4) {
        eq_free(h->wt_arena);
        ADF_TRY(eq_alloc(&h->wt_arena, per * nrad));
        h->wt_bytes = per * nrad * 4;
    }
    eq_radial* rads[EQ_MAX_LAYERS + 3];
    int k = 0;
    rads[k++] = &h->ed_rad;
    for (int i = 0; i < h->hp.num_layers; ++i) rads[k++] = &h->blk[i].ga.rad;
    rads[k++] = &h->force[0].rad; rads[k++] = &h->force[1].rad;
    for (int i = 0; i < k; ++i) {
        rads[i]->w0t = h->wt_arena + per * i;
        ADF_TRY(eq_launch_transpose(rads[i]->l0.w, rads[i]->w0t, d.EC, d.NB + 2 * d.EC, s));
    }
    ADF_TRY(eq_fold_ffn(h, s));
    ADF_TRY(eq_split_weights(h, s));
    if (h->folded)
        for (int i = 0; i < h->hp.num_layers; ++i) h->blk[i].ffn.l2f[0] = h->blk[i].ffn.l2[0];  // l = 0 comes from the gate
    ADF_TRY(eq_radial_static(h, rads, k, s));
    h->weights_set = true;
    h->inc_valid = false;
    return ADF_OK;
}

extern "C" int32_t adf_eqv2_set_arithmetic(adf_eqv2_t h, int32_t exact_f32) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    h->exact_f32 = exact_f32 != 0;
    h->inc_valid = false;
    return ADF_OK;
}

// ---------------------------------------------------------------------------------------------- capacity
static size_t eq_arena_floats_per_edge(const eq_dims& d) {
    const size_t extra = (size_t)d.NH * d.A + d.Hd;
    size_t y = extra + (size_t)(d.L + 1) * d.Hd, z = (size_t)(d.L + 1) * d.HV;
    for (int m = 1; m <= d.M; ++m) { const size_t nm = d.L - m + 1; y += 4 * nm * d.Hd; z += 4 * nm * d.HV; }
    return 2 * (size_t)d.EC + (size_t)d.RW * 2 * d.C + (size_t)d.Sr * 2 * d.C + y + d.NH + (size_t)d.Sr * d.Hd + z + 2 * (size_t)d.M + 1 + 32;
}

static int32_t eq_ensure_capacity(adf_eqv2* h, int64_t N, int64_t B, int64_t Eneed) {
    const eq_dims& d = h->d;
    const int K = h->hp.max_neighbors;
    int64_t kk = K;
    if (h->ext_graph && h->maxdeg > kk) kk = h->maxdeg;
    if (N > h->capN || B > h->capB || Eneed > h->capE || kk > h->arena_kk) {
        ADF_HIP_CHECK(hipDeviceSynchronize());
        const int64_t cN = N > h->capN ? N : h->capN, cB = B > h->capB ? B : h->capB;
        int64_t cE = cN * K;
        if (Eneed > cE) cE = Eneed;
        if (h->capE > cE) cE = h->capE;
        eq_free_workspaces(h);
        ADF_TRY(eq_alloc(&h->nbr_cnt, (size_t)cN + 1));
        ADF_HIP_CHECK(hipMemset(h->nbr_cnt, 0, sizeof(int32_t) * ((size_t)cN + 1)));
        ADF_TRY(eq_alloc(&h->nbr_src, (size_t)cN * K)); ADF_TRY(eq_alloc(&h->nbr_shift, (size_t)cN * K));
        ADF_TRY(eq_alloc(&h->img_cnt, (size_t)cB)); ADF_TRY(eq_alloc(&h->eptr, (size_t)cN + 2));
        ADF_TRY(eq_alloc(&h->e_src, (size_t)cE)); ADF_TRY(eq_alloc(&h->e_dst, (size_t)cE));
        ADF_TRY(eq_alloc(&h->e_vec, (size_t)cE * 3)); ADF_TRY(eq_alloc(&h->wig, (size_t)cE * d.DR));
        size_t bytes = 0;
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, (int)(cN + 1));
        h->scan_tmp_bytes = bytes;
        ADF_HIP_CHECK(hipMalloc(&h->scan_tmp, bytes ? bytes : 16));
        ADF_TRY(eq_alloc(&h->cache_d2, (size_t)cN * K)); ADF_TRY(eq_alloc(&h->cache_cid, (size_t)cN * K));
        ADF_TRY(eq_alloc(&h->cache_cnt, (size_t)cN));
        h->cache_valid = false;
        const size_t ns = (size_t)cN * d.S;
        const size_t wmax = (size_t)(d.C > d.F ? d.C : d.F);
        ADF_TRY(eq_alloc(&h->x, ns * d.C)); ADF_TRY(eq_alloc(&h->y, ns * d.C));
        ADF_TRY(eq_alloc(&h->agg, ns * (d.HV > (int)wmax ? d.HV : wmax)));
        ADF_TRY(eq_alloc(&h->gate, (size_t)cN * d.F));
        ADF_TRY(eq_alloc(&h->h1, ns * d.F)); ADF_TRY(eq_alloc(&h->h2, ns * d.F));
        ADF_TRY(eq_alloc(&h->sys, (size_t)cB * 16));
        // chunk size: bounded edge-arena (ADF_EQV2_CHUNK_EDGES, default 2^19 edges)
        int64_t chunk_edges = 1 << 19;
        if (const char* e = getenv("ADF_EQV2_CHUNK_EDGES")) { const long long v = atoll(e); if (v > 0) chunk_edges = v; }
        if (h->arena_kk > kk) kk = h->arena_kk;
        h->arena_kk = kk;
        int64_t cn = chunk_edges / kk;
        if (cn < 1) cn = 1;
        if (cn > cN) cn = cN;
        h->chunk_nodes = cn;
        h->arena_floats = eq_arena_floats_per_edge(d) * (size_t)(cn * kk);
        ADF_TRY(eq_alloc(&h->arena, h->arena_floats));
        h->garena_floats = 2 * (size_t)cn * d.G * d.F;
        ADF_TRY(eq_alloc(&h->garena, h->garena_floats));
        int64_t rc = 2 * cn * kk;
        if (3 * cn * d.G > rc) rc = 3 * cn * d.G;
        if (cN * (2 * d.L + 1) > rc) rc = cN * (2 * d.L + 1);
        ADF_TRY(eq_alloc(&h->rs, (size_t)rc));
        h->rs_cap = rc;
        h->capN = cN; h->capB = cB; h->capE = cE;
    }
    return ADF_OK;
}

extern "C" int32_t adf_eqv2_set_edges(adf_eqv2_t h, int64_t num_edges, const int32_t* src, const int32_t* dst,
                                      const float* vec, int32_t max_in_degree, void* stream) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    h->inc_valid = false;
    if (num_edges <= 0) { h->ext_graph = false; h->E_ext = 0; return ADF_OK; }
    if (!src || !dst || !vec || max_in_degree < 1 || max_in_degree > 128) { adf_set_error("eqv2_set_edges: bad argument"); return ADF_EINVAL; }
    // private copy; the forward moves it into the (possibly re-allocated) graph workspaces
    if (num_edges > h->xe_cap) {
        ADF_HIP_CHECK(hipDeviceSynchronize());
        eq_free(h->xe_src); eq_free(h->xe_dst); eq_free(h->xe_vec);
        ADF_TRY(eq_alloc(&h->xe_src, (size_t)num_edges)); ADF_TRY(eq_alloc(&h->xe_dst, (size_t)num_edges));
        ADF_TRY(eq_alloc(&h->xe_vec, (size_t)num_edges * 3));
        h->xe_cap = num_edges;
    }
    ADF_HIP_CHECK(hipMemcpyAsync(h->xe_src, src, sizeof(int32_t) * num_edges, hipMemcpyDeviceToDevice, s));
    ADF_HIP_CHECK(hipMemcpyAsync(h->xe_dst, dst, sizeof(int32_t) * num_edges, hipMemcpyDeviceToDevice, s));
    ADF_HIP_CHECK(hipMemcpyAsync(h->xe_vec, vec, sizeof(float) * 3 * num_edges, hipMemcpyDeviceToDevice, s));
    h->ext_graph = true; h->E_ext = num_edges; h->maxdeg = max_in_degree;
    return ADF_OK;
}

extern "C" int32_t adf_eqv2_set_moving(adf_eqv2_t h, const int32_t* moving, const int32_t* mov_idx, const int32_t* mov_off) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    if (moving && (!mov_idx || !mov_off)) { adf_set_error("eqv2_set_moving: need mov_idx and mov_off"); return ADF_EINVAL; }
    h->moving = moving; h->mov_idx = moving ? mov_idx : nullptr; h->mov_off = moving ? mov_off : nullptr;
    h->cache_valid = false;
    h->inc_valid = false;
    return ADF_OK;
}

// ---------------------------------------------------------------------------------------------- incremental blocks
extern "C" int32_t adf_eqv2_set_incremental(adf_eqv2_t h, int32_t on) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    h->inc_on = on != 0;
    h->inc_valid = false;
    h->inc_rows = h->inc_rows_full = 0;
    if (!h->inc_on && h->inc_capN) { ADF_HIP_CHECK(hipDeviceSynchronize()); eq_inc_free(h); }
    return ADF_OK;
}

// kept state for the handle's current capacity; an allocation failure switches the mode off instead of failing the forward
static bool eq_inc_ensure(adf_eqv2* h) {
    const eq_dims& d = h->d;
    const int nl = h->hp.num_layers;
    if (h->inc_capN == h->capN && h->inc_capE == h->capE && h->inc_nl == nl) return true;
    (void)hipDeviceSynchronize();
    eq_inc_free(h);
    const size_t row = (size_t)d.S * d.C, cN = (size_t)h->capN;
    int32_t st = ADF_OK;
    for (int i = 0; i <= nl && st == ADF_OK; ++i) st = eq_alloc(&h->inc_x[i], cN * row);
    if (st == ADF_OK) st = eq_alloc(&h->inc_xg, cN * row);
    if (st == ADF_OK) st = eq_alloc(&h->inc_peptr, cN + 2);
    if (st == ADF_OK) st = eq_alloc(&h->inc_psrc, (size_t)h->capE);
    if (st == ADF_OK) st = eq_alloc(&h->inc_pvec, (size_t)h->capE * 3);
    if (st == ADF_OK) st = eq_alloc(&h->inc_dirty, 2 * cN);
    if (st == ADF_OK) st = eq_alloc(&h->inc_idx, (size_t)nl * cN);
    if (st == ADF_OK) st = eq_alloc(&h->inc_cnt, (size_t)nl + 1);
    if (st == ADF_OK) {
        h->inc_sel_bytes = eq_inc_select_bytes((int)cN);
        if (hipMalloc(&h->inc_sel_tmp, h->inc_sel_bytes ? h->inc_sel_bytes : 16) != hipSuccess) st = ADF_EOOM;
    }
    if (st != ADF_OK) {
        (void)hipGetLastError();
        eq_inc_free(h);
        h->inc_on = false;
        return false;
    }
    h->inc_capN = h->capN; h->inc_capE = h->capE; h->inc_nl = nl;
    return true;
}

// ---------------------------------------------------------------------------------------------- dense product dispatch
// rows [row0, row0 + n) of a weight matrix as a map of their own (the fp16 images are plain [out, in] arrays with one
// power-of-two scale per matrix, so a block of rows is a pointer offset)
static eq_lin eq_lin_rows(const eq_lin& p, int row0, int n) {
    eq_lin r = p;
    r.w = p.w + (size_t)row0 * p.in;
    r.b = p.b ? p.b + row0 : nullptr;
    r.out = n;
    if (p.has16) {
        r.w16.hi = static_cast<unsigned char*>(p.w16.hi) + (size_t)row0 * p.in * 2;
        r.w16.lo = static_cast<unsigned char*>(p.w16.lo) + (size_t)row0 * p.in * 2;
        r.w16.frag = nullptr;   // (the fragment image is ordered by 32-row blocks of the whole matrix)
    }
    return r;
}

int32_t eq_gemm(const adf_eqv2* h, const float* A, int lda, const eq_rowmap* amap, const eq_lin* W, bool use_bias,
                float* Cm, int ldc, const eq_rowmap* cmap, long long M, int act, bool accumulate, hipStream_t s,
                const float* rs_pre, float* out_mag, int rs_div) {
    const eq_rowmap a1 = {lda, 1, 0}, c1 = {ldc, 1, 0};
    const eq_rowmap* am = amap ? amap : &a1;
    const eq_rowmap* cm = cmap ? cmap : &c1;
    if (!h->exact_f32 && W->has16 && (rs_pre || M <= h->rs_cap) && eq_gemm16_ok(A, am, Cm, cm, W->out, W->in)) {
        // per-row power-of-two lift of A (eqv2_gemm16.hip; rs_pre: already written by the producer of A), then the product
        if (!rs_pre) ADF_TRY(eq_launch_rowscale(A, am, M, W->in, h->rs, s));
        // Plain dense rows on both sides, whole 256-column tiles, an even number of K tiles: the sampler's streamed-fragment
        // product kernel of gemm16.hip (eight waves, 192 x 256 tile, software-pipelined conversion; same lifts, same products
        // in the same order: same bits) - the second SO(2) convolution's orders m >= 1 (N = 1536 / 1280 at config 4); edge-level
        // launches only (node-level products measured no better there)
        if (h->conv2_wr && !amap && !cmap && !accumulate && act == 0 && !out_mag && (!rs_pre || rs_div == 1) && W->w16.frag &&
            W->out % 256 == 0 && (W->in / 32) % 2 == 0 && lda == W->in && M >= 65536 && M * (long long)lda * 4 < (1ll << 32) && M < (1ll << 31))
            return adf_launch_gemm16(A, lda, &W->w16, use_bias ? W->b : nullptr, Cm, ldc, (int)M, W->out, W->in, 0, s, nullptr, 0,
                                     nullptr, rs_pre ? rs_pre : h->rs, nullptr, nullptr, 0, nullptr, 0);
        if (out_mag) ADF_HIP_CHECK(hipMemsetAsync(out_mag, 0, sizeof(float) * (size_t)M, s));
        return eq_launch_gemm16(A, am, rs_pre ? rs_pre : h->rs, &W->w16, use_bias ? W->b : nullptr, Cm, cm, M, W->out, W->in,
                                act, accumulate, s, out_mag, rs_pre ? rs_div : 1);
    }
    // exact-f32 product: no lifts needed downstream either (out_mag stays untouched; callers only pass it on when the
    // matrix-core path runs, see eq_uses_mfma)
    return eq_gemm_f32(A, lda, amap, W->w, use_bias ? W->b : nullptr, Cm, ldc, cmap, M, W->out, W->in, act, accumulate, s);
}

// whether eq_gemm will take the f16x3 kernel for this product (then row magnitudes can be handed from producer to consumer)
static bool eq_uses_mfma(const adf_eqv2* h, const float* A, int lda, const eq_lin* W, const float* Cm, int ldc) {
    const eq_rowmap a1 = {lda, 1, 0}, c1 = {ldc, 1, 0};
    return !h->exact_f32 && W->has16 && eq_gemm16_ok(A, &a1, Cm, &c1, W->out, W->in);
}

// ---------------------------------------------------------------------------------------------- forward
struct eq_chunk_bufs {
    float *radh, *radh2, *rad, *alpha, *y0, *z0;
    float* m[EQ_MAX_M + 1];
    float* y[EQ_MAX_M + 1];
    float* mb[EQ_MAX_M + 1];
    float* z[EQ_MAX_M + 1];
    float* rsb[EQ_MAX_M + 1];  // power-of-two lifts of the rows of mb[m] (written by the S2 activation)
};

static void eq_carve(const adf_eqv2* h, long long Eub, eq_chunk_bufs* b) {
    const eq_dims& d = h->d;
    float* p = h->arena;
    auto take = [&](size_t n) { float* q = p; p += (n + 3) / 4 * 4; return q; };
    const size_t extra = (size_t)d.NH * d.A + d.Hd;
    b->radh = take((size_t)Eub * d.EC); b->radh2 = take((size_t)Eub * d.EC);
    b->rad = take((size_t)Eub * d.RW * 2 * d.C);
    b->alpha = take((size_t)Eub * d.NH);
    for (int m = 0; m <= d.M; ++m) {
        const size_t nm = d.L - m + 1, rows = m == 0 ? Eub : 2 * Eub;
        b->m[m] = take(rows * nm * 2 * d.C);
        b->y[m] = take(rows * (m == 0 ? extra + nm * d.Hd : 2 * nm * d.Hd));
        b->mb[m] = take(rows * nm * d.Hd);
        b->z[m] = take(rows * (m == 0 ? nm * d.HV : 2 * nm * d.HV));
        b->rsb[m] = take(rows);
    }
    b->y0 = b->y[0]; b->z0 = b->z[0];
}

// RadialFunction on the edges of targets [n0, n1): out [Eub, r->l6.out]
static int32_t eq_radial(adf_eqv2* h, const eq_radial* r, const float* semb, const float* temb, const int32_t* Z, int n0,
                         int n1, long long Eub, eq_chunk_bufs* b, float* out, int N, hipStream_t s) {
    const eq_dims& d = h->d;
    eq_prof_scope ps(h, EQ_PROF_RADIAL, s);
    ADF_TRY(eq_launch_radial_pre(h, r, semb, temb, Z, n0, n1, b->radh, N, s));
    ADF_TRY(eq_launch_ln_silu(b->radh, r->ln1_w, r->ln1_b, Eub, d.EC, s));
    ADF_TRY(eq_gemm(h, b->radh, d.EC, nullptr, &r->l3, true, b->radh2, d.EC, nullptr, Eub, 0, false, s));
    ADF_TRY(eq_launch_ln_silu(b->radh2, r->ln4_w, r->ln4_b, Eub, d.EC, s));
    ADF_TRY(eq_gemm(h, b->radh2, d.EC, nullptr, &r->l6, true, out, r->l6.out, nullptr, Eub, 0, false, s));
    return ADF_OK;
}

// SO2EquivariantGraphAttention up to (not including) the final SO3 linear: agg [N, S, HV] (or [N, 3, HV] when only_l1)
static int32_t eq_attention(adf_eqv2* h, const eq_attn* at, const float* y, const int32_t* Z, int N, float* agg, bool only_l1,
                            hipStream_t s) {
    const eq_dims& d = h->d;
    const int extra = d.NH * d.A + d.Hd;
    for (int n0 = 0; n0 < N; n0 += (int)h->chunk_nodes) {
        const int n1 = (int)((n0 + h->chunk_nodes < N) ? n0 + h->chunk_nodes : N);
        const long long Eub = (long long)(n1 - n0) * (h->ext_graph ? h->maxdeg : h->hp.max_neighbors);
        eq_chunk_bufs b;
        eq_carve(h, Eub, &b);
        const bool tab = h->rad_static && at->rad.table;
        if (!tab) ADF_TRY(eq_radial(h, &at->rad, at->src_emb, at->dst_emb, Z, n0, n1, Eub, &b, b.rad, N, s));
        const bool lifts = !h->exact_f32;
        // first convolution on pre-split operands (rotate-in writes lifted fp16 hi / lo rows, the product kernel copies
        // them): every order's weight must have its fp16 image and a contraction length that is a multiple of 32
        bool pre = lifts && h->presplit && at->c1_m0.has16 && at->c1_m0.in % 32 == 0 && (at->c1_m0.out & 3) == 0;
        for (int m = 1; m <= d.M; ++m) pre = pre && at->c1_m[m - 1].has16 && at->c1_m[m - 1].in % 32 == 0 && (at->c1_m[m - 1].out & 3) == 0;
        {
            eq_prof_scope ps(h, EQ_PROF_ROTATE, s);
            ADF_TRY(eq_launch_rotate_in(h, y, tab ? at->rad.table : b.rad, Z, tab ? h->hp.max_num_elements : 0, n0, n1, b.m,
                                        lifts ? b.rsb : nullptr, pre, s));
        }
        {
            eq_prof_scope ps(h, EQ_PROF_CONV, s);
            if (pre) {
                for (int m = 0; m <= d.M; ++m) {
                    const eq_lin* W = m == 0 ? &at->c1_m0 : &at->c1_m[m - 1];
                    const long long rows = m == 0 ? Eub : 2 * Eub;
                    const _Float16* hi = reinterpret_cast<const _Float16*>(b.m[m]);
                    if (h->conv1_wr && eq_gemm16pw_ok(&W->w16, W->out, W->in))
                        ADF_TRY(eq_launch_gemm16pw(hi, hi + (size_t)rows * W->in, b.rsb[m], &W->w16, m == 0 ? W->b : nullptr,
                                                   b.y[m], W->out, rows, W->out, W->in, 0, s));
                    else
                        ADF_TRY(eq_launch_gemm16p(hi, hi + (size_t)rows * W->in, b.rsb[m], &W->w16, m == 0 ? W->b : nullptr, b.y[m],
                                                  W->out, rows, W->out, W->in, 0, s));
                }
            } else {
                ADF_TRY(eq_gemm(h, b.m[0], at->c1_m0.in, nullptr, &at->c1_m0, true, b.y[0], at->c1_m0.out, nullptr, Eub, 0, false, s,
                                lifts ? b.rsb[0] : nullptr));
                for (int m = 1; m <= d.M; ++m)
                    ADF_TRY(eq_gemm(h, b.m[m], at->c1_m[m - 1].in, nullptr, &at->c1_m[m - 1], false, b.y[m], at->c1_m[m - 1].out,
                                    nullptr, 2 * Eub, 0, false, s, lifts ? b.rsb[m] : nullptr));
            }
        }
        { eq_prof_scope ps(h, EQ_PROF_ATTN, s); ADF_TRY(eq_launch_alpha(h, at, b.y[0], at->c1_m0.out, n0, n1, b.alpha, s)); }
        bool rs_ok = false;
        { eq_prof_scope ps(h, EQ_PROF_S2ACT, s); ADF_TRY(eq_launch_s2act(h, b.y[0], b.y, extra, d.NH * d.A, n0, n1, b.mb, h->s2_emit_mag ? b.rsb : nullptr, &rs_ok, s)); }
        // a force block reads only the l = 1 rows of the rotated-back message: its second convolution keeps only the
        // output columns that reach them (order 0: l = 1; order 1: real and imaginary part of l = 1; order 2: none)
        const bool compact = only_l1 && d.M >= 1 && d.L >= 1 && !h->no_compact;
        if (compact) {
            eq_prof_scope ps(h, EQ_PROF_CONV, s);
            const eq_lin w0 = eq_lin_rows(at->c2_m0, d.HV, d.HV);
            const eq_lin wr = eq_lin_rows(at->c2_m[0], 0, d.HV), wi = eq_lin_rows(at->c2_m[0], d.L * d.HV, d.HV);
            ADF_TRY(eq_gemm(h, b.mb[0], w0.in, nullptr, &w0, true, b.z[0], d.HV, nullptr, Eub, 0, false, s, rs_ok ? b.rsb[0] : nullptr));
            const float* rs1 = rs_ok ? b.rsb[1] : nullptr;
            const bool mf = eq_uses_mfma(h, b.mb[1], wr.in, &wr, b.z[1], 2 * d.HV) && 2 * Eub <= h->rs_cap;
            if (!rs1 && mf) { const eq_rowmap am1 = {wr.in, 1, 0}; ADF_TRY(eq_launch_rowscale(b.mb[1], &am1, 2 * Eub, wr.in, h->rs, s)); rs1 = h->rs; }
            ADF_TRY(eq_gemm(h, b.mb[1], wr.in, nullptr, &wr, false, b.z[1], 2 * d.HV, nullptr, 2 * Eub, 0, false, s, rs1));
            ADF_TRY(eq_gemm(h, b.mb[1], wi.in, nullptr, &wi, false, b.z[1] + d.HV, 2 * d.HV, nullptr, 2 * Eub, 0, false, s, rs1));
        } else {
            eq_prof_scope ps(h, EQ_PROF_CONV, s);
            ADF_TRY(eq_gemm(h, b.mb[0], at->c2_m0.in, nullptr, &at->c2_m0, true, b.z[0], at->c2_m0.out, nullptr, Eub, 0, false, s,
                            rs_ok ? b.rsb[0] : nullptr));
            for (int m = 1; m <= d.M; ++m)
                ADF_TRY(eq_gemm(h, b.mb[m], at->c2_m[m - 1].in, nullptr, &at->c2_m[m - 1], false, b.z[m], at->c2_m[m - 1].out,
                                nullptr, 2 * Eub, 0, false, s, rs_ok ? b.rsb[m] : nullptr));
        }
        { eq_prof_scope ps(h, EQ_PROF_ROTATE, s); ADF_TRY(eq_launch_rotate_out(h, b.z, b.alpha, n0, n1, agg, only_l1, s, compact)); }
    }
    return ADF_OK;
}

// SO3_LinearV2 (so3.py:694-745): one weight matrix per degree, bias on l = 0; in [N, S, cin] -> out [N, S, cout]
static int32_t eq_so3_linear(adf_eqv2* h, const eq_lin* per_l, const float* in, int cin, float* out, int cout, int N,
                             bool accumulate, hipStream_t s) {
    const eq_dims& d = h->d;
    for (int l = 0; l <= d.L; ++l) {
        const int P = 2 * l + 1;
        eq_rowmap am = {(long long)d.S * cin, P, cin}, cm = {(long long)d.S * cout, P, cout};
        ADF_TRY(eq_gemm(h, in + (size_t)l * l * cin, cin, &am, &per_l[l], l == 0, out + (size_t)l * l * cout, cout, &cm,
                        (long long)N * P, 0, accumulate, s));
    }
    return ADF_OK;
}

static int32_t eq_check_batch(const adf_eqv2* h, const adf_batch* b) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    if (!b || b->num_atoms <= 0 || b->num_systems <= 0 || !b->pos || !b->cell || !b->atomic_numbers || !b->batch || !b->atom_offset) {
        adf_set_error("eqv2: incomplete batch descriptor");
        return ADF_EINVAL;
    }
    if (!h->weights_set || !h->consts_set) { adf_set_error("eqv2: set_constants / set_weights first"); return ADF_EINVAL; }
    return ADF_OK;
}

// compact edge arrays for n_out listed targets (at most kk incoming edges each)
static int32_t eq_ensure_subset(adf_eqv2* h, int64_t n_out) {
    if (n_out <= h->sub_cap) return ADF_OK;
    ADF_HIP_CHECK(hipDeviceSynchronize());
    eq_free(h->sub_eptr); eq_free(h->sub_src); eq_free(h->sub_dst); eq_free(h->sub_vec); eq_free(h->sub_wig); eq_free(h->sub_f);
    h->sub_cap = 0;
    const int64_t cap = n_out + n_out / 4 + 64;
    const int64_t kk = h->arena_kk > 0 ? h->arena_kk : h->hp.max_neighbors;
    ADF_TRY(eq_alloc(&h->sub_eptr, (size_t)2 * (cap + 2)));  // CSR, then the counts it is scanned from
    ADF_TRY(eq_alloc(&h->sub_src, (size_t)cap * kk)); ADF_TRY(eq_alloc(&h->sub_dst, (size_t)cap * kk));
    ADF_TRY(eq_alloc(&h->sub_vec, (size_t)cap * kk * 3)); ADF_TRY(eq_alloc(&h->sub_wig, (size_t)cap * kk * h->d.DR));
    ADF_TRY(eq_alloc(&h->sub_f, (size_t)cap * 3));
    h->sub_cap = cap;
    return ADF_OK;
}

// the graph the attention kernels read (through the handle): the full one or the compact copy of a target subset
struct eq_graph_view { int32_t *eptr, *src, *dst; float *vec, *wig; };
static eq_graph_view eq_graph_use_subset(adf_eqv2* h) {
    const eq_graph_view full = {h->eptr, h->e_src, h->e_dst, h->e_vec, h->wig};
    h->eptr = h->sub_eptr; h->e_src = h->sub_src; h->e_dst = h->sub_dst; h->e_vec = h->sub_vec; h->wig = h->sub_wig;
    return full;
}
static void eq_graph_restore(adf_eqv2* h, const eq_graph_view& v) {
    h->eptr = v.eptr; h->e_src = v.src; h->e_dst = v.dst; h->e_vec = v.vec; h->wig = v.wig;
}

// node side of a block on rows [0, n) of X (h->agg holds their aggregated messages):
// X += proj(agg); X += ffn(norm_2(X))   (transformer_block.py:650-700, 375-531)
static int32_t eq_block_nodes(adf_eqv2* h, const eq_block& bk, float* X, int n, hipStream_t s) {
    const eq_dims& d = h->d;
    { eq_prof_scope ps(h, EQ_PROF_NODE, s); ADF_TRY(eq_so3_linear(h, bk.ga.proj_l, h->agg, d.HV, X, d.C, n, true, s)); }
    { eq_prof_scope ps(h, EQ_PROF_NODE, s); ADF_TRY(eq_launch_norm(h, &bk.n2, X, h->y, n, s)); }
    eq_prof_scope ps(h, EQ_PROF_FFN, s);
    const eq_ffn& f = bk.ffn;
    // scalar gate from the l = 0 row of every node (row stride S*C)
    ADF_TRY(eq_gemm(h, h->y, d.S * d.C, nullptr, &f.scalar, true, h->gate, d.F, nullptr, n, 2, false, s));
    const bool fold = h->folded && !h->exact_f32;
    ADF_TRY(eq_so3_linear(h, fold ? f.l1f : f.l1, h->y, d.C, h->h1, d.F, n, false, s));
    for (int n0 = 0; n0 < n; n0 += (int)h->chunk_nodes) {
        const int n1 = (int)((n0 + h->chunk_nodes < n) ? n0 + h->chunk_nodes : n);
        const long long rows = (long long)(n1 - n0) * d.G;
        float* ga = h->garena;
        float* gb = h->garena + (size_t)h->chunk_nodes * d.G * d.F;
        if (fold) {  // silu(to_grid(.)) -> one product + SiLU -> from_grid
            // the product's row lifts: one per NODE (the largest grid value of its tile, left by to_grid) - from_grid
            // sums a node's rows with comparable weights, so a row far below the node's largest carries no weight
            bool em = false;
            const bool want = eq_uses_mfma(h, ga, d.F, &f.g2, gb, d.F) && (n1 - n0) <= h->rs_cap;
            ADF_TRY(eq_launch_to_grid(h, h->h1, n0, n1, ga, true, s, want ? h->rs : nullptr, &em));
            ADF_TRY(eq_gemm(h, ga, d.F, nullptr, &f.g2, false, gb, d.F, nullptr, rows, 2, false, s, em ? h->rs : nullptr,
                            nullptr, d.G));
            ADF_TRY(eq_launch_from_grid(h, gb, h->gate, n0, n1, h->h2, s));
            continue;
        }
        ADF_TRY(eq_launch_to_grid(h, h->h1, n0, n1, ga, false, s));
        // the first product measures its operand rows itself; the next two get them from the producer's epilogue
        const bool chain = eq_uses_mfma(h, ga, d.F, &f.g0, gb, d.F) && eq_uses_mfma(h, gb, d.F, &f.g2, ga, d.F) &&
                           eq_uses_mfma(h, ga, d.F, &f.g4, gb, d.F) && 3 * rows <= h->rs_cap;
        float* m1 = chain ? h->rs + rows : nullptr;
        float* m2 = chain ? h->rs + 2 * rows : nullptr;
        ADF_TRY(eq_gemm(h, ga, d.F, nullptr, &f.g0, false, gb, d.F, nullptr, rows, 2, false, s, nullptr, m1));
        ADF_TRY(eq_gemm(h, gb, d.F, nullptr, &f.g2, false, ga, d.F, nullptr, rows, 2, false, s, m1, m2));
        ADF_TRY(eq_gemm(h, ga, d.F, nullptr, &f.g4, false, gb, d.F, nullptr, rows, 0, false, s, m2, nullptr));
        ADF_TRY(eq_launch_from_grid(h, gb, h->gate, n0, n1, h->h2, s));
    }
    ADF_TRY(eq_so3_linear(h, fold ? f.l2f : f.l2, h->h2, d.F, X, d.C, n, true, s));
    return ADF_OK;
}

// out_idx != null: the two force blocks (all that reads the last embedding) run for the listed target atoms only, on a
// compacted copy of their incoming edges; rows out_idx[*] of f1 / f2 are written, bit-identical to the full forward's
// (every row of every product, the softmax of a target and its aggregation depend on that target's edges alone).
static int32_t eq_forward_impl(adf_eqv2* h, const adf_batch* b, float* f1, float* f2, float* x_blocks, hipStream_t s,
                               const int32_t* out_idx = nullptr, int32_t n_out = 0) {
    const eq_dims& d = h->d;
    const int N = b->num_atoms, B = b->num_systems;
    ADF_TRY(eq_ensure_capacity(h, N, B, h->ext_graph ? h->E_ext : 0));
    const int32_t* Z = b->atomic_numbers;
    {
        eq_prof_scope ps(h, EQ_PROF_GRAPH, s);
        if (h->ext_graph) {
            ADF_HIP_CHECK(hipMemcpyAsync(h->e_src, h->xe_src, sizeof(int32_t) * h->E_ext, hipMemcpyDeviceToDevice, s));
            ADF_HIP_CHECK(hipMemcpyAsync(h->e_dst, h->xe_dst, sizeof(int32_t) * h->E_ext, hipMemcpyDeviceToDevice, s));
            ADF_HIP_CHECK(hipMemcpyAsync(h->e_vec, h->xe_vec, sizeof(float) * 3 * h->E_ext, hipMemcpyDeviceToDevice, s));
            ADF_TRY(eq_launch_eptr_from_dst(h, N, h->E_ext, s));
        } else {
            ADF_TRY(eq_launch_edges_from_topk(h, b, s));
        }
        ADF_TRY(eq_launch_wigner(h, N, s));
        ADF_TRY(eq_launch_check_z(h, Z, N, s));
    }
    h->lastN = N;
    const size_t xs = (size_t)N * d.S * d.C;
    const int nl = h->hp.num_layers;
    // incremental blocks: only while the static-atom promise is in force (the same batch, only flagged atoms move)
    const bool inc = h->inc_on && h->moving && nl > 0 && eq_inc_ensure(h);
    bool lists = false;
    int32_t cnt[EQ_MAX_LAYERS];
    if (inc) {
        eq_prof_scope ps(h, EQ_PROF_GRAPH, s);
        if (h->inc_valid && h->inc_N == N) {
            ADF_TRY(eq_launch_inc_lists(h, N, nl, s));
            // the list lengths size the launches of this forward: one read-back (the forward is ~10^5 times longer)
            ADF_HIP_CHECK(hipMemcpyAsync(cnt, h->inc_cnt, sizeof(int32_t) * nl, hipMemcpyDeviceToHost, s));
            ADF_HIP_CHECK(hipStreamSynchronize(s));
            lists = true;
        }
        ADF_TRY(eq_launch_inc_keep(h, N, s));
        h->inc_valid = false;  // until this forward has filled every kept array
        // a block's listed targets can be nearly all of them: size the compact edge arrays once
        ADF_TRY(eq_ensure_subset(h, N));
    }
    float* X = inc ? h->inc_x[0] : h->x;
    // node embedding + edge-degree embedding (equiformer_v2_denoising.py:232-285)
    for (int n0 = 0; n0 < N; n0 += (int)h->chunk_nodes) {
        const int n1 = (int)((n0 + h->chunk_nodes < N) ? n0 + h->chunk_nodes : N);
        const long long Eub = (long long)(n1 - n0) * (h->ext_graph ? h->maxdeg : h->hp.max_neighbors);
        eq_chunk_bufs cb;
        eq_carve(h, Eub, &cb);
        const bool tab = h->rad_static && h->ed_rad.table;
        if (!tab) ADF_TRY(eq_radial(h, &h->ed_rad, h->ed_src_emb, h->ed_dst_emb, Z, n0, n1, Eub, &cb, cb.rad, N, s));
        eq_prof_scope ps(h, EQ_PROF_ROTATE, s);
        ADF_TRY(eq_launch_edge_degree(h, tab ? h->ed_rad.table : cb.rad, Z, tab ? h->hp.max_num_elements : 0, n0, n1, X, s));
    }
    if (x_blocks) ADF_HIP_CHECK(hipMemcpyAsync(x_blocks, X, xs * 4, hipMemcpyDeviceToDevice, s));
    h->last_block_rows = 0;
    for (int i = 0; i < nl; ++i) {
        const eq_block& bk = h->blk[i];
        int n = N;             // rows this block computes
        bool listed = false;   // ... as a compact list (the rest keep the rows of the forward that last computed them)
        if (inc) {
            float* Xout = h->inc_x[i + 1];
            if (lists) {
                n = cnt[i] < 0 ? 0 : (cnt[i] > N ? N : cnt[i]);
                listed = (long long)n * 10 <= (long long)N * 9;  // above 90 % the all-rows form is as fast
                if (!listed) n = N;
            }
            if (!listed) ADF_HIP_CHECK(hipMemcpyAsync(Xout, X, xs * 4, hipMemcpyDeviceToDevice, s));
            if (listed && n > 0) {
                const int32_t* idx = h->inc_idx + (size_t)i * h->inc_capN;
                // x = x + ga(norm_1(x)) + ffn(norm_2(.)) on the listed rows: sources are read from every row of the block below
                { eq_prof_scope ps(h, EQ_PROF_NODE, s); ADF_TRY(eq_launch_norm(h, &bk.n1, X, h->y, N, s)); }
                { eq_prof_scope ps(h, EQ_PROF_GRAPH, s); ADF_TRY(eq_launch_subset_graph(h, idx, n, s)); }
                const eq_graph_view full = eq_graph_use_subset(h);
                int32_t st = eq_attention(h, &bk.ga, h->y, Z, n, h->agg, false, s);
                eq_graph_restore(h, full);
                ADF_TRY(st);
                { eq_prof_scope ps(h, EQ_PROF_NODE, s); ADF_TRY(eq_launch_gather_rows(X, idx, n, d.S * d.C, h->inc_xg, s)); }
                ADF_TRY(eq_block_nodes(h, bk, h->inc_xg, n, s));
                { eq_prof_scope ps(h, EQ_PROF_NODE, s); ADF_TRY(eq_launch_scatter_rows(h->inc_xg, idx, n, d.S * d.C, Xout, s)); }
            }
            X = Xout;
        }
        if (!listed) {
            // x = x + ga(norm_1(x))   (transformer_block.py:650-700; drop path / dropout are identity in eval mode)
            { eq_prof_scope ps(h, EQ_PROF_NODE, s); ADF_TRY(eq_launch_norm(h, &bk.n1, X, h->y, N, s)); }
            ADF_TRY(eq_attention(h, &bk.ga, h->y, Z, N, h->agg, false, s));
            ADF_TRY(eq_block_nodes(h, bk, X, N, s));
        }
        h->last_block_rows += n;
        if (x_blocks) ADF_HIP_CHECK(hipMemcpyAsync(x_blocks + (size_t)(i + 1) * xs, X, xs * 4, hipMemcpyDeviceToDevice, s));
    }
    h->prof_block_rows += h->last_block_rows; h->prof_forwards += 1;
    if (inc) {
        h->inc_valid = true; h->inc_N = N;
        h->inc_rows += h->last_block_rows; h->inc_rows_full += (int64_t)nl * N;
    }
    { eq_prof_scope ps(h, EQ_PROF_NODE, s); ADF_TRY(eq_launch_norm(h, &h->final_norm, X, h->y, N, s)); }
    h->last_subset = out_idx ? n_out : -1;
    if (out_idx) {
        if (n_out <= 0) return ADF_OK;
        if (n_out > N) { adf_set_error("eqv2_forward_subset: more indices than atoms"); return ADF_EINVAL; }
        ADF_TRY(eq_ensure_subset(h, n_out));
        { eq_prof_scope ps(h, EQ_PROF_GRAPH, s); ADF_TRY(eq_launch_subset_graph(h, out_idx, n_out, s)); }
        // the attention kernels read the graph through the handle: point it at the compact arrays for the force blocks
        const eq_graph_view full = eq_graph_use_subset(h);
        int32_t st = ADF_OK;
        for (int k = 0; k < 2 && st == ADF_OK; ++k) {
            float* f = k == 0 ? f1 : f2;
            if (!f) continue;
            st = eq_attention(h, &h->force[k], h->y, Z, n_out, h->agg, true, s);
            eq_prof_scope ps(h, EQ_PROF_NODE, s);
            if (st == ADF_OK) st = eq_launch_force_out(h, &h->force[k], h->agg, n_out, h->sub_f, s);
            if (st == ADF_OK) st = eq_launch_scatter_rows3(h->sub_f, out_idx, n_out, f, s);
        }
        eq_graph_restore(h, full);
        return st;
    }
    for (int k = 0; k < 2; ++k) {
        float* f = k == 0 ? f1 : f2;
        if (!f) continue;
        ADF_TRY(eq_attention(h, &h->force[k], h->y, Z, N, h->agg, true, s));
        eq_prof_scope ps(h, EQ_PROF_NODE, s);
        ADF_TRY(eq_launch_force_out(h, &h->force[k], h->agg, N, f, s));
    }
    return ADF_OK;
}

extern "C" int32_t adf_eqv2_forward(adf_eqv2_t h, const adf_batch* b, float* f1, float* f2, float* x_blocks, void* stream) {
    ADF_TRY(eq_check_batch(h, b));
    if (!f1) { adf_set_error("eqv2_forward: f1 is null"); return ADF_EINVAL; }
    return eq_forward_impl(h, b, f1, f2, x_blocks, (hipStream_t)stream);
}

extern "C" int32_t adf_eqv2_forward_subset(adf_eqv2_t h, const adf_batch* b, const int32_t* out_idx, int32_t n_out, float* f1,
                                          float* f2, void* stream) {
    ADF_TRY(eq_check_batch(h, b));
    if (!f1 || !out_idx || n_out < 0) { adf_set_error("eqv2_forward_subset: null argument"); return ADF_EINVAL; }
    return eq_forward_impl(h, b, f1, f2, nullptr, (hipStream_t)stream, out_idx, n_out);
}

extern "C" int32_t adf_eqv2_check_flags(adf_eqv2_t h, void* stream) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    int32_t fl[EQ_NFLAGS];
    ADF_HIP_CHECK(hipMemcpyAsync(fl, h->flags, sizeof(fl), hipMemcpyDeviceToHost, s));
    ADF_HIP_CHECK(hipStreamSynchronize(s));
    bool any = false;
    for (int i = 0; i < EQ_NFLAGS; ++i) any = any || fl[i];
    if (any) ADF_HIP_CHECK(hipMemsetAsync(h->flags, 0, sizeof(fl), s));
    if (fl[0]) { adf_set_error("eqv2 graph: more in-cutoff candidates around one atom than the search holds"); return ADF_EOVERFLOW; }
    if (fl[2]) { adf_set_error("eqv2 graph: edge capacity exceeded"); return ADF_EOVERFLOW; }
    if (fl[3]) { adf_set_error("eqv2: more than 128 incoming edges on one atom"); return ADF_EOVERFLOW; }
    if (fl[4]) { adf_set_error("eqv2: atomic number outside the embedding / radius tables"); return ADF_EINVAL; }
    if (fl[5]) { adf_set_error("eqv2_forward_subset: atom index outside the batch"); return ADF_EINVAL; }
    if (fl[1]) { adf_set_error("An image has no neighbors"); return ADF_ENONEIGHBOR; }
    return ADF_OK;
}

// ---------------------------------------------------------------------------------------------- stepper on this handle
extern "C" int32_t adf_eqv2_init_placement(adf_eqv2_t h, const adf_batch* b, float* pos, const int32_t* tags,
                                           const float* noise, void* stream) {
    ADF_TRY(eq_check_batch(h, b));
    if (!pos || !tags || !noise) { adf_set_error("null argument"); return ADF_EINVAL; }
    return adf_stepper_init(b, pos, tags, noise, (hipStream_t)stream);
}

extern "C" int32_t adf_eqv2_sde_step(adf_eqv2_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                                     const float* f1, const float* f2, const adf_step_coef* coef,
                                     const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr, const float* z_rot,
                                     int32_t early_stop_count, int32_t* state, float* dcom, float* drot, void* stream) {
    ADF_TRY(eq_check_batch(h, b));
    if (!pos || !tags || !f1 || !f2 || (!coef && !coefs_dev) || !state) { adf_set_error("null argument"); return ADF_EINVAL; }
    if (!coef && num_steps <= 0) { adf_set_error("num_steps must be positive"); return ADF_EINVAL; }
    ADF_TRY(eq_ensure_capacity(h, b->num_atoms, b->num_systems, h->ext_graph ? h->E_ext : 0));
    eq_prof_scope ps(h, EQ_PROF_STEPPER, (hipStream_t)stream);
    return adf_stepper_step(h->sys, b, pos, tags, fixed, f1, f2, coef, coef ? nullptr : coefs_dev, num_steps, z_tr, z_rot,
                            early_stop_count, state, dcom, drot, (hipStream_t)stream);
}

struct adf_frames;
int32_t adf_frames_push_impl(adf_frames* f, const float* src, hipStream_t s);

static int32_t eq_sample_impl(adf_eqv2_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                              const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr_all,
                              const float* z_rot_all, int32_t early_stop_count, int32_t poll_every, int32_t* state,
                              const int32_t* out_idx, int32_t n_out, float* f1, float* f2, adf_frames* sink,
                              int32_t frame_every, void* stream) {
    ADF_TRY(eq_check_batch(h, b));
    if (num_steps <= 0 || !f1 || !f2 || !state || !coefs_dev || !pos || !tags || (out_idx && n_out < 0)) {
        adf_set_error("eqv2_sample: bad argument");
        return ADF_EINVAL;
    }
    if ((z_tr_all == nullptr) != (z_rot_all == nullptr)) { adf_set_error("eqv2_sample: need both noise tables or none"); return ADF_EINVAL; }
    if (sink && frame_every <= 0) { adf_set_error("eqv2_sample: frame_every must be positive"); return ADF_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const size_t zs = (size_t)b->num_systems * 3;
    for (int t = 0; t < num_steps; ++t) {
        ADF_TRY(eq_forward_impl(h, b, f1, f2, nullptr, s, out_idx, n_out));
        ADF_TRY(adf_eqv2_sde_step(h, b, pos, tags, fixed, f1, f2, nullptr, coefs_dev, num_steps,
                                  z_tr_all ? z_tr_all + t * zs : nullptr, z_rot_all ? z_rot_all + t * zs : nullptr,
                                  early_stop_count, state, nullptr, nullptr, stream));
        if (sink && ((t + 1) % frame_every == 0 || t + 1 == num_steps)) ADF_TRY(adf_frames_push_impl(sink, pos, s));
        if (early_stop_count > 0 && poll_every > 0 && (t % poll_every) == poll_every - 1 && t + 1 < num_steps) {
            int32_t frozen = 0;
            ADF_HIP_CHECK(hipMemcpyAsync(&frozen, state + 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            ADF_HIP_CHECK(hipStreamSynchronize(s));
            if (frozen) break;
        }
    }
    return ADF_OK;
}

extern "C" int32_t adf_eqv2_sample(adf_eqv2_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                                   const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr_all,
                                   const float* z_rot_all, int32_t early_stop_count, int32_t poll_every, int32_t* state,
                                   const int32_t* out_idx, int32_t n_out, float* f1, float* f2, void* stream) {
    return eq_sample_impl(h, b, pos, tags, fixed, coefs_dev, num_steps, z_tr_all, z_rot_all, early_stop_count, poll_every,
                          state, out_idx, n_out, f1, f2, nullptr, 0, stream);
}

extern "C" int32_t adf_eqv2_sample_traj(adf_eqv2_t h, const adf_batch* b, float* pos, const int32_t* tags,
                                        const int32_t* fixed, const adf_step_coef* coefs_dev, int32_t num_steps,
                                        const float* z_tr_all, const float* z_rot_all, int32_t early_stop_count,
                                        int32_t poll_every, int32_t* state, const int32_t* out_idx, int32_t n_out, float* f1,
                                        float* f2, adf_frames_t sink, int32_t frame_every, void* stream) {
    if (!sink) { adf_set_error("eqv2_sample_traj: null sink"); return ADF_EINVAL; }
    return eq_sample_impl(h, b, pos, tags, fixed, coefs_dev, num_steps, z_tr_all, z_rot_all, early_stop_count, poll_every,
                          state, out_idx, n_out, f1, f2, sink, frame_every, stream);
}

// Stand-alone C = act(A . W^T + b) through the dense-product kernels of this path (unit tests, micro-benchmarks).
// mode 0: exact f32 (any shape); 1: f16x3 with per-row lifts, fp32 A staged and split in the kernel; 2: f16x3 on
// pre-split fp16 hi / lo rows (eq_gemm16p_kernel).  `repeat` > 1 re-runs the product kernel alone (timing).
int32_t eq_launch_presplit(const float* A, const float* mag, long long M, int K, void* hi, void* lo, hipStream_t s);

extern "C" int32_t adf_eqv2_linear_forward(const float* A, const float* W, const float* bias, float* Cm, int64_t M, int32_t N,
                                           int32_t K, int32_t act, int32_t mode, int32_t repeat, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!A || !W || !Cm || M <= 0 || N <= 0 || K <= 0) { adf_set_error("bad argument"); return ADF_EINVAL; }
    if (mode == 0) return eq_gemm_f32(A, K, nullptr, W, bias, Cm, N, nullptr, M, N, K, act, false, s);
    const eq_rowmap am = {K, 1, 0}, cm = {N, 1, 0};
    if (!eq_gemm16_ok(A, &am, Cm, &cm, N, K)) { adf_set_error("shape not taken by the f16x3 kernels"); return ADF_EINVAL; }
    unsigned char* buf = nullptr;
    const size_t n = (size_t)N * K, ma = (size_t)M * K;
    if (mode == 3 && (N % 32 || K % 64 || N < 128)) { adf_set_error("mode 3 needs N %% 32 == 0, N >= 128 and K %% 64 == 0"); return ADF_EINVAL; }
    ADF_TRY(eq_alloc(&buf, n * 4 + 64 + (size_t)M * 4 + 64 + (mode >= 2 ? ma * 4 : 0) + 64 + (mode == 3 ? n * 4 : 0)));
    adf_w16 w16 = {};
    w16.hi = buf; w16.lo = buf + n * 2; w16.inv_scale = reinterpret_cast<float*>(buf + n * 4); w16.bias_perm = nullptr;
    unsigned int* scratch = reinterpret_cast<unsigned int*>(buf + n * 4 + 16);
    float* mag = reinterpret_cast<float*>(buf + n * 4 + 64);
    unsigned char* split = buf + n * 4 + 64 + (((size_t)M * 4 + 63) / 64) * 64;
    int32_t st = adf_split_weight(W, (long long)n, &w16, scratch, s);
    if (st == ADF_OK) st = eq_launch_rowscale(A, &am, M, K, mag, s);
    if (st == ADF_OK && mode >= 2) st = eq_launch_presplit(A, mag, M, K, split, split + ma * 2, s);
    if (st == ADF_OK && mode == 3) {
        w16.frag = split + ((ma * 4 + 63) / 64) * 64;
        st = adf_pack_frag(&w16, N, K, w16.frag, s);
    }
    for (int r = 0; r < (repeat > 0 ? repeat : 1) && st == ADF_OK; ++r) {
        if (mode == 3) st = eq_launch_gemm16pw(split, split + ma * 2, mag, &w16, bias, Cm, N, M, N, K, act, s);
        else if (mode == 2) st = eq_launch_gemm16p(split, split + ma * 2, mag, &w16, bias, Cm, N, M, N, K, act, s);
        else st = eq_launch_gemm16(A, &am, mag, &w16, bias, Cm, &cm, M, N, K, act, false, s, nullptr);
    }
    (void)hipStreamSynchronize(s);
    eq_free(buf);
    return st;
}

// ---------------------------------------------------------------------------------------------- counters / profile
extern "C" int32_t adf_eqv2_get_counters(adf_eqv2_t h, adf_eqv2_counters* out, void* stream) {
    if (!h || !out || h->lastN <= 0) { adf_set_error("eqv2: no forward has run"); return ADF_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const eq_dims& d = h->d;
    int32_t e = 0;
    ADF_HIP_CHECK(hipMemcpyAsync(&e, h->eptr + h->lastN, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ADF_HIP_CHECK(hipStreamSynchronize(s));
    const int64_t E = e, N = h->lastN;
    const int64_t extra = (int64_t)d.NH * d.A + d.Hd;
    int64_t conv1 = (int64_t)(d.L + 1) * 2 * d.C * (extra + (int64_t)(d.L + 1) * d.Hd);
    int64_t conv2 = (int64_t)(d.L + 1) * d.Hd * (int64_t)(d.L + 1) * d.HV;
    for (int m = 1; m <= d.M; ++m) {
        const int64_t nm = d.L - m + 1;
        conv1 += 2 * (nm * 2 * d.C) * (2 * nm * d.Hd);
        conv2 += 2 * (nm * d.Hd) * (2 * nm * d.HV);
    }
    const int64_t radial = (int64_t)d.EC * (2 * d.EC) + (int64_t)d.EC * d.EC + (int64_t)d.EC * d.RW * 2 * d.C;
    const int64_t s2 = 2ll * d.G * d.Sr * d.Hd;
    const int64_t rot = (int64_t)d.DR * (2 * d.C + d.HV);
    const int64_t per_edge_block = conv1 + conv2 + radial + s2 + rot;
    const int64_t per_node_block = (int64_t)d.S * d.HV * d.C + (int64_t)d.C * d.F + 2ll * d.S * d.C * d.F + 2ll * d.G * d.S * d.F +
                                   3ll * d.G * d.F * d.F;
    const int64_t blocks = h->hp.num_layers;
    const int64_t macs = E * per_edge_block * (blocks + 2) + N * per_node_block * blocks;
    out->num_edges = E; out->num_atoms = N;
    out->dense_flops = 2 * macs;  // the reference's algorithm (f32-equivalent), whatever shortcuts this path takes
    // what the SO(2)-convolution kernels of the last forward executed: the force blocks keep only the l = 1 columns of their
    // second convolution and, in a subset forward, only the edges of the listed targets (estimated as E n_out / N)
    const bool compact = d.M >= 1 && d.L >= 1 && !h->no_compact;
    const int64_t conv2f = compact ? (int64_t)(d.L + 1) * d.Hd * d.HV + 2 * ((int64_t)d.L * d.Hd) * (2 * d.HV) : conv2;
    const int64_t Ef = h->last_subset >= 0 ? E * h->last_subset / (N > 0 ? N : 1) : E;
    // incremental blocks: a block's convolutions run on the edges of its listed targets only (estimated as E rows / N)
    const int64_t Eb = N > 0 ? E * h->last_block_rows / N : E * blocks;
    out->conv_flops = 2 * (Eb * (conv1 + conv2) + Ef * (conv1 + conv2f) * 2);
    out->inc_rows = h->inc_rows; out->inc_rows_full = h->inc_rows_full;
    out->forwards_total = h->prof_forwards;
    out->conv_flops_total = 2 * ((N > 0 ? E * h->prof_block_rows / N : 0) * (conv1 + conv2) + Ef * (conv1 + conv2f) * 2 * h->prof_forwards);
    return ADF_OK;
}

extern "C" int32_t adf_eqv2_profile_enable(adf_eqv2_t h, int32_t on) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    h->prof_on = on != 0;
    h->prof_used = 0;
    h->prof_cat->clear();
    if (on) h->prof_block_rows = h->prof_forwards = 0;
    return ADF_OK;
}

extern "C" int32_t adf_eqv2_profile_read(adf_eqv2_t h, float* ms, int64_t* count, void* stream) {
    if (!h || !ms || !count) { adf_set_error("null argument"); return ADF_EINVAL; }
    ADF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < EQ_PROF_NCAT; ++i) { ms[i] = 0.f; count[i] = 0; }
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, (*h->prof_ev)[i], (*h->prof_ev)[i + 1]) == hipSuccess) {
            const int c = (*h->prof_cat)[i / 2];
            ms[c] += t; count[c] += 1;
        }
    }
    h->prof_used = 0;
    h->prof_cat->clear();
    return ADF_OK;
}
