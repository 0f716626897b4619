// C ABI of libadsorbdiff_hip.so: handle life-cycle, grow-only workspaces, and the launch
// sequence of one PaiNN denoiser forward (reference: painn_denoising.py:402-481).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "common.h"

static thread_local char g_err[512] = "";

void adf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* adf_last_error(void) { return g_err; }
extern "C" const char* adf_version(void) { return "adsorbdiff_hip 0.5.0 (gfx950)"; }

// ---- HIP-event profiling: pairs of events on the launch stream around kernel groups
void adf_prof_begin(adf_painn* h, int cat, hipStream_t s) {
    if (!h->prof_on) return;
    if (h->prof_used + 2 > h->prof_ev->size()) {
        for (int i = 0; i < 512; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            h->prof_ev->push_back(e);
        }
    }
    h->prof_cat->push_back(cat);
    (void)hipEventRecord((*h->prof_ev)[h->prof_used], s);
    h->prof_used += 1;
}
void adf_prof_end(adf_painn* h, hipStream_t s) {
    if (!h->prof_on || (h->prof_used & 1) == 0) return;
    (void)hipEventRecord((*h->prof_ev)[h->prof_used], s);
    h->prof_used += 1;
}

extern "C" int32_t adf_profile_enable(adf_painn_t h, int32_t on) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    h->prof_on = on != 0;
    h->prof_used = 0;
    h->prof_cat->clear();
    ADF_HIP_CHECK(hipMemset(h->kcount, 0, 8 * sizeof(unsigned long long)));
    return ADF_OK;
}

extern "C" int32_t adf_profile_read(adf_painn_t h, float* ms, int64_t* count, int64_t* message_ksteps, void* stream) {
    if (!h || !ms || !count) { adf_set_error("null argument"); return ADF_EINVAL; }
    ADF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    if (message_ksteps) {
        unsigned long long k[8];
        ADF_HIP_CHECK(hipMemcpy(k, h->kcount, sizeof(k), hipMemcpyDeviceToHost));
        ADF_HIP_CHECK(hipMemset(h->kcount, 0, sizeof(k)));
        *message_ksteps = (int64_t)k[0];
    }
    for (int c = 0; c < ADF_PROF_NCAT; ++c) { ms[c] = 0.f; count[c] = 0; }
    const size_t pairs = h->prof_used / 2;
    for (size_t i = 0; i < pairs; ++i) {
        float t = 0.f;
        ADF_HIP_CHECK(hipEventElapsedTime(&t, (*h->prof_ev)[2 * i], (*h->prof_ev)[2 * i + 1]));
        const int c = (*h->prof_cat)[i];
        ms[c] += t;
        count[c] += 1;
    }
    h->prof_used = 0;
    h->prof_cat->clear();
    return ADF_OK;
}

template <typename T>
static int32_t dev_alloc(T** p, size_t count) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (count == 0) return ADF_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        (void)hipGetLastError();
        adf_set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
        return ADF_EOOM;
    }
    return ADF_OK;
}

extern "C" int32_t adf_painn_create(const adf_painn_hparams* hp, adf_painn_t* out) {
    if (!hp || !out) { adf_set_error("null argument"); return ADF_EINVAL; }
    if (hp->hidden_channels % ADF_SLICE_CH != 0 || hp->hidden_channels < 128 || hp->hidden_channels > 1024) {
        adf_set_error("hidden_channels=%d must be a multiple of %d in [128,1024]", hp->hidden_channels, ADF_SLICE_CH);
        return ADF_EINVAL;
    }
    if (hp->num_rbf % 2 != 0 || hp->num_rbf > 128 || hp->num_rbf < 2) {
        adf_set_error("num_rbf=%d must be even and <= 128", hp->num_rbf);
        return ADF_EINVAL;
    }
    if (hp->max_neighbors < 1 || hp->max_neighbors > ADF_MAX_K) {
        adf_set_error("max_neighbors=%d must be in [1,%d]", hp->max_neighbors, ADF_MAX_K);
        return ADF_EINVAL;
    }
    if (hp->num_layers < 1 || hp->num_layers > 16 || hp->num_heads < 1 || hp->num_heads > 2) {
        adf_set_error("num_layers must be in [1,16], num_heads in [1,2]");
        return ADF_EINVAL;
    }
    adf_painn* h = new (std::nothrow) adf_painn();
    if (!h) { adf_set_error("host allocation failed"); return ADF_EOOM; }
    memset(h, 0, sizeof(*h));
    h->hp = *hp;
    h->prof_ev = new std::vector<hipEvent_t>();
    h->prof_cat = new std::vector<int>();
    hipDeviceProp_t prop;
    if (hipGetDevice(&h->device) != hipSuccess || hipGetDeviceProperties(&prop, h->device) != hipSuccess) {
        adf_set_error("no usable HIP device: %s", hipGetErrorString(hipGetLastError()));
        adf_painn_destroy(h);
        return ADF_EHIP;
    }
    h->num_cus = prop.multiProcessorCount;
    {   // incremental layers: on unless ADF_INCREMENTAL=0 (adf_painn_set_incremental overrides)
        const char* e = getenv("ADF_INCREMENTAL");
        h->inc_on = !(e && e[0] == '0');
        const char* e2 = getenv("ADF_INC_SYNC");
        h->inc_sync = e2 && e2[0] == '1';
    }
    const int H = hp->hidden_channels, R = hp->num_rbf, L = hp->num_layers;
    int32_t st = dev_alloc(&h->rbf_pack, (size_t)L * (H / ADF_SLICE_CH) * R * 192);
    if (st == ADF_OK) st = dev_alloc(&h->rbf_bias_pack, (size_t)L * (H / ADF_SLICE_CH) * 192);
    {
        uint16_t* p16 = nullptr;
        if (st == ADF_OK) st = dev_alloc(&p16, (size_t)L * (H / ADF_SLICE_CH) * R * 192 * 2);
        h->rbf_pack16 = p16;
    }
    if (st == ADF_OK) st = dev_alloc(&h->rbf_bias_pack16, (size_t)L * (H / ADF_SLICE_CH) * 192);
    if (st == ADF_OK) st = dev_alloc(&h->rbf_scales, 16);
    if (st == ADF_OK) st = dev_alloc(&h->flags, ADF_NFLAGS);
    if (st == ADF_OK && hipMemset(h->flags, 0, sizeof(int32_t) * ADF_NFLAGS) != hipSuccess) st = ADF_EHIP;
    {   // fp16 hi/lo arena: per layer 9 H^2 + 2 (H*2H) ... computed exactly below
        const size_t HH = (size_t)H * H;
        const size_t per_layer = HH + 3 * HH + 2 * HH + 2 * HH + 3 * HH;                 // xp0 xp2 vp xv0 xv2
        const size_t per_head = HH + HH / 2 + 2 * HH + HH + HH / 4 + HH / 2;              // b0: vec1 vec2 un0 un2 ; b1: vec1 un0
        const size_t elems = (size_t)L * per_layer + (size_t)hp->num_heads * per_head;
        h->w16_bytes = elems * 2 * sizeof(uint16_t) + 4096;
        if (st == ADF_OK) st = dev_alloc(&h->w16_arena, h->w16_bytes);
        if (st == ADF_OK) st = dev_alloc(&h->w16_scales, 256);
        if (st == ADF_OK) st = dev_alloc(&h->w16_bias_perm, (size_t)L * 2 * 3 * H);
        if (st == ADF_OK) st = dev_alloc(&h->w16_scratch, 1);
        if (st == ADF_OK) st = dev_alloc(&h->wfrag_arena, h->w16_bytes);
        {   // form of the x_proj / xvec_proj pairs: the two-kernel form unless asked otherwise (mlp16.hip: measured slower)
            const char* ef = getenv("ADF_FUSED_MLP");
            h->fused_mlp = ef ? atoi(ef) : 0;
            if (h->fused_mlp < 0 || h->fused_mlp > 2) h->fused_mlp = 0;
        }
        const char* e = getenv("ADF_GEMM");
        h->gemm_f32 = e && strcmp(e, "f32") == 0;
        const char* el = getenv("ADF_LIFT");
        h->lift_on = !(el && strcmp(el, "0") == 0);
        const char* e2 = getenv("ADF_MSG");
        h->msg_f32 = e2 ? strcmp(e2, "f32") == 0 : h->gemm_f32;
    }
    if (st == ADF_OK) st = dev_alloc(&h->kcount, 8);
    if (st == ADF_OK && hipMemset(h->kcount, 0, 8 * sizeof(unsigned long long)) != hipSuccess) st = ADF_EHIP;
    if (st != ADF_OK) { adf_painn_destroy(h); return st; }
    *out = h;
    return ADF_OK;
}

static void inc_free(adf_painn* h) {
    for (int l = 0; l <= ADF_MAX_LAYERS; ++l) {
        if (h->incX[l]) (void)hipFree(h->incX[l]);
        if (h->incV[l]) (void)hipFree(h->incV[l]);
        if (l < ADF_MAX_LAYERS && h->incR[l]) (void)hipFree(h->incR[l]);
        h->incX[l] = h->incV[l] = nullptr;
        if (l < ADF_MAX_LAYERS) h->incR[l] = nullptr;
    }
    void* ptrs[] = {h->inc_c0, h->inc_chg, h->inc_pend, h->inc_need, h->inc_tf, h->inc_list, h->inc_cnt, h->inc_tmp};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->inc_c0 = h->inc_chg = h->inc_pend = h->inc_need = h->inc_tf = nullptr;
    h->inc_list = h->inc_cnt = nullptr; h->inc_tmp = nullptr; h->inc_tmp_bytes = 0;
    h->inc_capN = 0; h->inc_valid = false;
}

static void free_workspaces(adf_painn* h) {
    void* ptrs[] = {h->nbr_cnt, h->nbr_src, h->nbr_shift, h->deg, h->nptr, h->cursor, h->img_cnt, h->sys_slow, h->scan_tmp,
                    h->e_src, h->e_geom, h->x, h->vecA, h->vecB, h->y, h->xh, h->vv, h->cat, h->dot, h->sys, h->rec, h->lift.buf, h->mag_a, h->mag_b, h->mag_v3,
                    h->cache_d2, h->cache_cid, h->cache_cnt, h->prev_nptr, h->prev_src, h->prev_geom};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->prev_nptr = h->prev_src = nullptr; h->prev_geom = nullptr; h->inc_valid = false;
    h->nbr_cnt = h->nbr_src = h->nbr_shift = h->deg = h->nptr = h->cursor = h->img_cnt = h->sys_slow = h->e_src = nullptr;
    h->scan_tmp = nullptr; h->scan_tmp_bytes = 0;
    h->e_geom = nullptr;
    h->x = h->vecA = h->vecB = h->y = h->xh = h->vv = h->cat = h->dot = h->sys = h->rec = nullptr;
    h->lift.buf = h->mag_a = h->mag_b = h->mag_v3 = nullptr; h->lift.cap = 0; h->mag_v3_valid = false;
    h->cache_d2 = nullptr; h->cache_cid = nullptr; h->cache_cnt = nullptr; h->cache_valid = false;
    h->capN = h->capB = h->capE = 0;
}

extern "C" int32_t adf_painn_destroy(adf_painn_t h) {
    if (!h) return ADF_OK;
    free_workspaces(h);
    if (h->rbf_pack) (void)hipFree(h->rbf_pack);
    if (h->rbf_bias_pack) (void)hipFree(h->rbf_bias_pack);
    if (h->rec0) (void)hipFree(h->rec0);
    inc_free(h);
    if (h->inc_cnt_host) (void)hipHostFree(h->inc_cnt_host);
    for (int i = 0; i < 2; ++i)
        if (h->inc_ev[i]) (void)hipEventDestroy((hipEvent_t)h->inc_ev[i]);
    if (h->sub_x) (void)hipFree(h->sub_x);
    if (h->sub_vec) (void)hipFree(h->sub_vec);
    if (h->sub_f) (void)hipFree(h->sub_f);
    if (h->rbf_pack16) (void)hipFree(h->rbf_pack16);
    if (h->rbf_bias_pack16) (void)hipFree(h->rbf_bias_pack16);
    if (h->rbf_scales) (void)hipFree(h->rbf_scales);
    if (h->flags) (void)hipFree(h->flags);
    if (h->kcount) (void)hipFree(h->kcount);
    if (h->w16_arena) (void)hipFree(h->w16_arena);
    if (h->wfrag_arena) (void)hipFree(h->wfrag_arena);
    if (h->w16_scales) (void)hipFree(h->w16_scales);
    if (h->w16_bias_perm) (void)hipFree(h->w16_bias_perm);
    if (h->w16_scratch) (void)hipFree(h->w16_scratch);
    if (h->prof_ev) { for (hipEvent_t e : *h->prof_ev) (void)hipEventDestroy(e); delete h->prof_ev; }
    delete h->prof_cat;
    delete h;
    return ADF_OK;
}

extern "C" int32_t adf_painn_set_weights(adf_painn_t h, int32_t n_weights, const void* const* w,
                                         const float* scale_factors, void* stream) {
    if (!h || !w || !scale_factors) { adf_set_error("null argument"); return ADF_EINVAL; }
    const int L = h->hp.num_layers;
    const int expect = 2 + ADF_WEIGHTS_PER_LAYER * L + ADF_WEIGHTS_PER_HEAD * h->hp.num_heads;
    if (n_weights != expect) {
        adf_set_error("expected %d weight tensors, got %d", expect, n_weights);
        return ADF_EINVAL;
    }
    for (int i = 0; i < n_weights; ++i)
        if (!w[i]) { adf_set_error("weight %d is null", i); return ADF_EINVAL; }
    auto f = [&](int i) { return reinterpret_cast<const float*>(w[i]); };
    h->emb = f(0);
    h->rbf_offset = f(1);
    int k = 2;
    for (int l = 0; l < L; ++l) {
        adf_layer_weights& lw = h->layer[l];
        lw.ln_w = f(k++); lw.ln_b = f(k++); lw.xp0_w = f(k++); lw.xp0_b = f(k++); lw.xp2_w = f(k++); lw.xp2_b = f(k++);
        lw.rbf_w = f(k++); lw.rbf_b = f(k++); lw.vp_w = f(k++); lw.xv0_w = f(k++); lw.xv0_b = f(k++);
        lw.xv2_w = f(k++); lw.xv2_b = f(k++);
        h->scale[l] = scale_factors[l];
    }
    for (int hd = 0; hd < h->hp.num_heads; ++hd)
        for (int b = 0; b < 2; ++b) {
            adf_block_weights& bw = h->head[hd][b];
            bw.vec1_w = f(k++); bw.vec2_w = f(k++); bw.un0_w = f(k++); bw.un0_b = f(k++); bw.un2_w = f(k++);
            bw.un2_b = f(k++);
        }
    ADF_TRY(adf_pack_rbf(h, (hipStream_t)stream));
    {   // split every GEMM weight into fp16 hi/lo (gemm16.hip)
        hipStream_t s = (hipStream_t)stream;
        const long long H = h->hp.hidden_channels, HH = H * H;
        unsigned char* cur = h->w16_arena;
        unsigned char* fcur = h->wfrag_arena;
        int nscale = 0;
        float* bperm = h->w16_bias_perm;
        // w [rows, K] -> hi / lo planes (+ the permutations of the fused layers) and their fragment-ordered image
        auto split = [&](const float* w, long long rows, long long K, adf_w16* out, const float* fused_bias = nullptr,
                         bool pair_perm = false) -> int32_t {
            const long long n = rows * K;
            out->hi = cur; cur += n * 2;
            out->lo = cur; cur += n * 2;
            out->inv_scale = h->w16_scales + nscale++;
            out->bias_perm = nullptr;
            out->frag = nullptr;
            if (fused_bias) {  // 3H-wide layer feeding a fused epilogue: rows permuted (gemm16.hip)
                out->bias_perm = bperm; bperm += 3 * H;
                ADF_TRY(adf_split_weight(w, n, out, h->w16_scratch, s, (int)H, (int)H, fused_bias));
            } else if (pair_perm) {  // vec_proj [2H, H]: (v1, v2) rows of the same channels side by side (gemm16.hip EPI 3)
                ADF_TRY(adf_split_weight(w, n, out, h->w16_scratch, s, (int)H, (int)H, nullptr, 2));
            } else {
                ADF_TRY(adf_split_weight(w, n, out, h->w16_scratch, s));
            }
            if (rows % 32 == 0 && K % 16 == 0) {
                out->frag = fcur; fcur += n * 4;
                ADF_TRY(adf_pack_frag(out, (int)rows, (int)K, out->frag, s));
            }
            return ADF_OK;
        };
        for (int l = 0; l < L; ++l) {
            adf_layer_weights& lw = h->layer[l];
            ADF_TRY(split(lw.xp0_w, H, H, &lw.xp0_16));
            ADF_TRY(split(lw.xp2_w, 3 * H, H, &lw.xp2_16, lw.xp2_b));
            ADF_TRY(split(lw.vp_w, 2 * H, H, &lw.vp_16, nullptr, true));
            ADF_TRY(split(lw.xv0_w, H, 2 * H, &lw.xv0_16));
            ADF_TRY(split(lw.xv2_w, 3 * H, H, &lw.xv2_16, lw.xv2_b));
        }
        for (int hd = 0; hd < h->hp.num_heads; ++hd) {
            adf_block_weights& b0 = h->head[hd][0];
            adf_block_weights& b1 = h->head[hd][1];
            ADF_TRY(split(b0.vec1_w, H, H, &b0.vec1_16));
            ADF_TRY(split(b0.vec2_w, H / 2, H, &b0.vec2_16));
            ADF_TRY(split(b0.un0_w, H, 2 * H, &b0.un0_16));
            ADF_TRY(split(b0.un2_w, H, H, &b0.un2_16));
            ADF_TRY(split(b1.vec1_w, H / 2, H / 2, &b1.vec1_16));
            ADF_TRY(split(b1.un0_w, H / 2, H, &b1.un0_16));
        }
        if ((size_t)(cur - h->w16_arena) > h->w16_bytes || (size_t)(fcur - h->wfrag_arena) > h->w16_bytes || nscale > 256) {
            adf_set_error("internal: fp16 weight arena overflow");
            return ADF_EINVAL;
        }
    }
    h->weights_set = true;
    h->rec0_valid = false;
    h->inc_valid = false;
    return ADF_OK;
}

// grow-only workspaces for N atoms in B systems
static int32_t ensure_capacity(adf_painn* h, int64_t N, int64_t B) {
    if (N <= h->capN && B <= h->capB) return ADF_OK;
    const int64_t capN = N > h->capN ? N : h->capN;
    const int64_t capB = B > h->capB ? B : h->capB;
    // synchronise before freeing buffers that enqueued work may still use
    ADF_HIP_CHECK(hipDeviceSynchronize());
    free_workspaces(h);
    const int64_t H = h->hp.hidden_channels, K = h->hp.max_neighbors;
    // symmetrised edges: every directed top-K entry (j -> i) survives at most once (j < i, or a
    // self image with a negative shift) and is then doubled: E <= 2*N*K.
    const int64_t capE = 2 * capN * K;
    int32_t st = ADF_OK;
#define ALLOC(field, count) if (st == ADF_OK) st = dev_alloc(&h->field, (size_t)(count))
    ALLOC(nbr_cnt, capN);
    ALLOC(nbr_src, capN * K);
    ALLOC(nbr_shift, capN * K);
    ALLOC(cache_d2, capN * K);
    ALLOC(cache_cid, capN * K);
    ALLOC(cache_cnt, capN);
    ALLOC(deg, capN + 1);
    ALLOC(nptr, capN + 1);
    ALLOC(cursor, capN);
    ALLOC(img_cnt, capB);
    ALLOC(sys_slow, capB);
    ALLOC(e_src, capE);
    ALLOC(e_geom, capE);
    if (h->inc_on) {  // previous build's CSR (incremental layers); swapped with the live one per build
        ALLOC(prev_nptr, capN + 1);
        ALLOC(prev_src, capE);
        ALLOC(prev_geom, capE);
    }
    if (st == ADF_OK) {
        h->scan_tmp_bytes = adf_scan_temp_bytes(capN + 1);
        unsigned char* tmp = nullptr;
        st = dev_alloc(&tmp, h->scan_tmp_bytes + 16);
        h->scan_tmp = tmp;
    }
    ALLOC(x, capN * H);
    ALLOC(vecA, capN * 3 * H);
    ALLOC(vecB, capN * 3 * H);
    ALLOC(rec, (capN + 1) * 5 * H);  // + one all-zero record row: gather target of padded edge rows
    ALLOC(y, capN * H);
    ALLOC(xh, capN * 3 * H);
    ALLOC(vv, capN * 6 * H);
    ALLOC(cat, capN * 2 * H);
    ALLOC(dot, capN * H);
    ALLOC(sys, capB * 16);
    ALLOC(lift.buf, capN * 3);
    ALLOC(mag_a, capN);
    ALLOC(mag_b, capN);
    ALLOC(mag_v3, capN * 3);
#undef ALLOC
    h->lift.cap = st == ADF_OK ? capN * 3 : 0;
    if (st != ADF_OK) { free_workspaces(h); return st; }
    h->capN = capN; h->capB = capB; h->capE = capE;
    return ADF_OK;
}

static int32_t check_batch(const adf_painn* h, const adf_batch* b) {
    if (!h || !b) { adf_set_error("null argument"); return ADF_EINVAL; }
    if (b->num_atoms <= 0 || b->num_systems <= 0) { adf_set_error("empty batch"); return ADF_EINVAL; }
    if (!b->pos || !b->cell || !b->batch || !b->atom_offset) { adf_set_error("null batch array"); return ADF_EINVAL; }
    for (int k = 0; k < 3; ++k)
        if (b->reps[k] < 0 || b->reps[k] > 16) { adf_set_error("reps[%d]=%d out of range", k, b->reps[k]); return ADF_EINVAL; }
    return ADF_OK;
}

// The flags are sticky on the device: kernels only ever set them, so a condition raised at any step of a sampling
// loop survives until it is read here (which clears them).
static int32_t read_flags(adf_painn* h, hipStream_t s) {
    int32_t f[ADF_NFLAGS];
    ADF_HIP_CHECK(hipMemcpyAsync(f, h->flags, sizeof(f), hipMemcpyDeviceToHost, s));
    ADF_HIP_CHECK(hipMemsetAsync(h->flags, 0, sizeof(f), s));
    ADF_HIP_CHECK(hipStreamSynchronize(s));
    if (f[4]) { adf_set_error("atomic number outside [1, %d] (rows of the embedding table)", h->hp.num_elements); return ADF_EINVAL; }
    if (f[0]) { adf_set_error("a centre atom has more than %d in-cutoff candidates", ADF_MAX_CAND); return ADF_EOVERFLOW; }
    if (f[2]) { adf_set_error("edge buffer overflow"); return ADF_EOVERFLOW; }
    if (f[3]) { adf_set_error("an atom has more than %d incoming edges", ADF_MAX_INDEG); return ADF_EOVERFLOW; }
    if (f[1]) { adf_set_error("An image has no neighbors"); return ADF_ENONEIGHBOR; }
    if (f[5]) {
        adf_set_error("non-finite model output%s", h->gemm_f32 ? " (exact-f32 arithmetic: the inputs or weights are not finite)"
                      : " in f16x3 arithmetic: an activation left the fp16 range or the inputs are not finite; "
                        "adf_painn_set_arithmetic(h, 1) selects exact f32");
        return ADF_ENUMERIC;
    }
    return ADF_OK;
}

extern "C" int32_t adf_graph_set_moving(adf_painn_t h, const int32_t* moving, const int32_t* mov_idx,
                                        const int32_t* mov_off) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    if (moving && (!mov_idx || !mov_off)) { adf_set_error("moving mask needs mov_idx and mov_off"); return ADF_EINVAL; }
    h->moving = moving; h->mov_idx = mov_idx; h->mov_off = mov_off;
    ADF_HIP_CHECK(hipMemset(h->flags, 0, sizeof(int32_t) * ADF_NFLAGS));  // a new promise starts with clean flags
    h->cache_valid = false;
    h->rec0_valid = false;
    h->inc_valid = false;
    return ADF_OK;
}

extern "C" int32_t adf_graph_build(adf_painn_t h, const adf_batch* b, void* stream, int64_t* num_edges) {
    ADF_TRY(check_batch(h, b));
    ADF_TRY(ensure_capacity(h, b->num_atoms, b->num_systems));
    hipStream_t s = (hipStream_t)stream;
    ADF_TRY(adf_graph_build_impl(h, b, s));
    if (num_edges) {
        int32_t e = 0;
        ADF_HIP_CHECK(hipMemcpyAsync(&e, h->nptr + b->num_atoms, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        ADF_TRY(read_flags(h, s));
        *num_edges = e;
    }
    return ADF_OK;
}

__global__ void adf_export_edges_kernel(const int32_t* nptr, const int32_t* e_src, const float4* eg, int N,
                                        int32_t* src, int32_t* dst, float* dist, float* vec, long long cap) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    for (int e = nptr[n] + (threadIdx.x & 63); e < nptr[n + 1]; e += 64) {
        if (e >= cap) continue;
        if (src) src[e] = e_src[e];
        if (dst) dst[e] = n;
        const float4 q = eg[e];
        if (dist) dist[e] = q.w;
        if (vec) { vec[3 * (size_t)e] = q.x; vec[3 * (size_t)e + 1] = q.y; vec[3 * (size_t)e + 2] = q.z; }
    }
}

__global__ void adf_export_shift_kernel(const int32_t* nbr_shift, int32_t* out, long long n, int r0, int r1, int r2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = nbr_shift[i];
    const int n2 = 2 * r2 + 1, n1 = 2 * r1 + 1;
    const int ia = c / (n1 * n2), rem = c - ia * (n1 * n2);
    out[3 * i] = ia - r0;
    out[3 * i + 1] = rem / n2 - r1;
    out[3 * i + 2] = rem % n2 - r2;
}

extern "C" int32_t adf_graph_export(adf_painn_t h, int32_t* nbr_count, int32_t* nbr_src, int32_t* nbr_shift,
                                    int64_t edge_capacity, int32_t* edge_src, int32_t* edge_dst, float* edge_dist,
                                    float* edge_vec, int64_t* num_edges, void* stream) {
    if (!h || h->lastN <= 0) { adf_set_error("no graph built"); return ADF_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const int64_t N = h->lastN, K = h->hp.max_neighbors;
    if (nbr_count) ADF_HIP_CHECK(hipMemcpyAsync(nbr_count, h->nbr_cnt, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, s));
    if (nbr_src) ADF_HIP_CHECK(hipMemcpyAsync(nbr_src, h->nbr_src, sizeof(int32_t) * N * K, hipMemcpyDeviceToDevice, s));
    if (nbr_shift)
        hipLaunchKernelGGL(adf_export_shift_kernel, dim3((unsigned)((N * K + 255) / 256)), dim3(256), 0, s, h->nbr_shift,
                           nbr_shift, (long long)(N * K), h->last_reps[0], h->last_reps[1], h->last_reps[2]);
    if (edge_src || edge_dst || edge_dist || edge_vec)
        hipLaunchKernelGGL(adf_export_edges_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, h->nptr, h->e_src,
                           h->e_geom, (int)N, edge_src, edge_dst, edge_dist, edge_vec, (long long)edge_capacity);
    ADF_HIP_CHECK(hipGetLastError());
    int32_t e = 0;
    ADF_HIP_CHECK(hipMemcpyAsync(&e, h->nptr + N, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ADF_HIP_CHECK(hipStreamSynchronize(s));
    if (num_edges) *num_edges = e;
    return ADF_OK;
}

// Record row N is the all-zero gather target of padded edge rows (message.hip); rows beyond the
// current N may hold data of an earlier, larger batch, so it is re-zeroed per forward.
static int32_t zero_pad_rows(adf_painn* h, int N, hipStream_t s) {
    const size_t row = (size_t)5 * h->hp.hidden_channels;
    ADF_HIP_CHECK(hipMemsetAsync(h->rec + (size_t)N * row, 0, sizeof(float) * row, s));
    return ADF_OK;
}

// Gather records of layer l for the n rows (x, vec): xh = x_proj(LayerNorm(x)) (painn_denoising.py:531), packed with
// vec for the message kernel.  row_map != null: row r of (x, vec) is atom row_map[r] of the record table.
// The fused two-layer kernel (mlp16.hip) or the two-kernel form of the x_proj / xvec_proj pairs?  Same bits either way.
// By size (mode 2): a 64-row tile per CU wants at least two rounds of tiles on the chip; below that the 128 x 192 tiles of
// the two-kernel form spread a small batch over more CUs (B = 1: 200 rows are 4 fused tiles against 16 workgroups).
static bool use_fused_mlp(const adf_painn* h, int rows) {
    if (h->hp.hidden_channels != 512 || h->gemm_f32 || h->fused_mlp == 0) return false;
    if (h->fused_mlp == 1) return true;
    return rows >= 2 * 64 * h->num_cus;
}

static int32_t make_records(adf_painn* h, int l, int n, const float* x, const float* vec, bool vec_is_zero, float* rec,
                            const int32_t* row_map, hipStream_t s) {
    const int H = h->hp.hidden_channels;
    const adf_layer_weights& w = h->layer[l];
    if (n <= 0) return ADF_OK;
    adf_prof_begin(h, ADF_PROF_NODE, s);
    const bool lift = h->lift_on && !h->gemm_f32;
    static int emit = -1;
    if (emit < 0) { const char* e = getenv("ADF_LIFT_EMIT"); emit = (e && atoi(e) == 0) ? 0 : 1; }
    const bool em = lift && emit;
    // row magnitudes travel with the rows: LayerNorm -> x_proj.0 -> (its epilogue) -> x_proj.2
    ADF_TRY(adf_nodewise_layernorm(x, w.ln_w, w.ln_b, h->y, n, H, s, em ? h->mag_a : nullptr, h->rows_dev));
    if (use_fused_mlp(h, n) && (em || !lift)) {   // x_proj.0 -> x_proj.2 -> records in one kernel (mlp16.hip), same bits
        adf_epi ep = {};
        ep.vec_in = vec; ep.rec = rec ? rec : h->rec; ep.H = H; ep.vec_is_zero = vec_is_zero ? 1 : 0;
        ep.row_map = row_map; ep.m_dev = h->rows_dev; ep.lift_y = lift ? 1 : 0;
        ep.rec_rows = row_map ? (long long)h->inc_capN : (long long)n;   // mapped rows index the whole kept table
        ADF_TRY(adf_launch_mlp16(h->y, nullptr, H, em ? h->mag_a : nullptr, w.xp0_16.frag, &w.xp0_16, w.xp0_b, w.xp2_16.frag, &w.xp2_16, n,
                                 H, 1, &ep, s));
        adf_prof_end(h, s);
        return ADF_OK;
    }
    ADF_TRY(adf_linear(h, h->y, H, w.xp0_w, &w.xp0_16, w.xp0_b, h->cat, H, n, H, H, 1, s, em ? h->mag_a : nullptr,
                       em ? h->mag_b : nullptr));
    if (h->gemm_f32) {
        if (row_map) { adf_set_error("internal: mapped records need the f16x3 path"); return ADF_EINVAL; }
        ADF_TRY(adf_launch_gemm(h->cat, H, w.xp2_w, H, w.xp2_b, h->xh, 3 * H, n, 3 * H, H, 0, s));
        ADF_TRY(adf_pack_records(h, n, h->xh, vec, vec_is_zero, s, rec));
    } else {  // x_proj.2 with the gather records written from the accumulators (xh never materialised)
        adf_epi ep = {};
        ep.vec_in = vec; ep.rec = rec ? rec : h->rec; ep.H = H; ep.vec_is_zero = vec_is_zero ? 1 : 0;
        ep.row_map = row_map;
        ep.rmag = em ? h->mag_b : nullptr;
        ep.m_dev = h->rows_dev;
        ADF_TRY(adf_launch_gemm16_fused(h->cat, H, &w.xp2_16, n, H, H, 1, &ep, s, lift ? &h->lift : nullptr));
    }
    adf_prof_end(h, s);
    return ADF_OK;
}

// tlist != null: only the listed targets are evaluated and x_out / vec_out are compact [n_targets, ...] rows
// rec != null: gather records go to / come from this buffer instead of h->rec; records_ready: they are already there
static int32_t message_layer(adf_painn* h, int l, int N, const float* x, const float* vec, float* x_out,
                             float* vec_out, bool vec_is_zero, hipStream_t s, const int32_t* tlist = nullptr,
                             int n_targets = 0, float* rec = nullptr, bool records_ready = false) {
    if (!records_ready) ADF_TRY(make_records(h, l, N, x, vec, vec_is_zero, rec, nullptr, s));
    adf_prof_begin(h, ADF_PROF_MESSAGE, s);
    const int32_t st = adf_message_impl(h, l, N, x, h->xh, vec, x_out, vec_out, vec_is_zero, s, tlist, n_targets, rec,
                                        tlist ? h->rows_dev : nullptr);
    adf_prof_end(h, s);
    return st;
}

static int32_t update_layer(adf_painn* h, int l, int N, float* x, float* vec, hipStream_t s) {
    const int H = h->hp.hidden_channels;
    const adf_layer_weights& w = h->layer[l];
    adf_prof_begin(h, ADF_PROF_NODE, s);
    if (h->gemm_f32) {
        ADF_TRY(adf_launch_gemm(vec, H, w.vp_w, H, nullptr, h->vv, 2 * H, 3 * N, 2 * H, H, 0, s));
        ADF_TRY(adf_nodewise_update_prep(h->vv, x, h->cat, h->dot, N, H, s));
        ADF_TRY(adf_linear(h, h->cat, 2 * H, w.xv0_w, &w.xv0_16, w.xv0_b, h->y, H, N, H, 2 * H, 1, s));
    } else {  // vec_proj with dot and |v2| formed on the accumulators (v1 -> vv [N,3,H], |v2| -> cat [N,H]);
              // xvec_proj.0 then reads its [x | |v2|] input from the two arrays
        adf_epi ep = {};
        ep.v1 = h->vv; ep.dotw = h->dot; ep.cat = h->cat; ep.H = H; ep.m_dev = h->rows_dev;
        const adf_lift* lf = h->lift_on ? &h->lift : nullptr;
        // vec rows and the [x | |v2|] rows are measured by a pass of their own; xvec_proj.0 hands its output rows' on.
        // (Measured alternative: the message kernel emitting the magnitudes of its vec_out rows - DPP maxima per
        // half-wave, 4 atomicMax per target and channel slice: measuring passes -11 ms, message kernel +9 ms per 10 full
        // steps of 1000 systems: a wash, not kept.)
        ADF_TRY(adf_launch_gemm16_fused(vec, H, &w.vp_16, N, H, H, 3, &ep, s, lf));
        if (use_fused_mlp(h, N)) {   // xvec_proj.0 -> xvec_proj.2 -> gating in one kernel (mlp16.hip), same bits
            const float* rm = nullptr;
            if (lf) {
                if (N > lf->cap) { adf_set_error("mlp16: lift scratch holds %lld rows, need %d", lf->cap, N); return ADF_EINVAL; }
                ADF_TRY(adf_launch_rowmag(x, H, H, h->cat, H, N, lf->buf, s, h->rows_dev, 1));
                rm = lf->buf;
            }
            adf_epi e2 = {};
            e2.x = x; e2.vec = vec; e2.dot = h->dot; e2.vv = h->vv; e2.scale = h->scale[l]; e2.H = H;
            e2.m_dev = h->rows_dev; e2.lift_y = h->lift_on ? 1 : 0;
            const int32_t stf = adf_launch_mlp16(x, h->cat, H, rm, w.xv0_16.frag, &w.xv0_16, w.xv0_b, w.xv2_16.frag, &w.xv2_16, N, H, 2, &e2, s);
            adf_prof_end(h, s);
            return stf;
        }
        ADF_TRY(adf_launch_gemm16(x, H, &w.xv0_16, w.xv0_b, h->y, H, N, H, 2 * H, 1, s, h->cat, H, lf, nullptr,
                                  h->lift_on ? h->mag_b : nullptr, h->rows_dev));
    }
    int32_t st;
    if (h->gemm_f32) {
        ADF_TRY(adf_launch_gemm(h->y, H, w.xv2_w, H, w.xv2_b, h->xh, 3 * H, N, 3 * H, H, 0, s));
        st = adf_nodewise_update_apply(h->xh, h->dot, h->vv, x, vec, h->scale[l], N, H, s);
    } else {  // xvec_proj.2 with gating + residuals + ScaleFactor applied on the accumulators
        adf_epi ep = {};
        ep.x = x; ep.vec = vec; ep.dot = h->dot; ep.vv = h->vv; ep.scale = h->scale[l]; ep.H = H;
        ep.rmag = h->lift_on ? h->mag_b : nullptr;
        ep.m_dev = h->rows_dev;
        st = adf_launch_gemm16_fused(h->y, H, &w.xv2_16, N, H, H, 2, &ep, s);
    }
    adf_prof_end(h, s);
    return st;
}

extern "C" int32_t adf_painn_message_layer(adf_painn_t h, int32_t layer, int32_t N, const float* x, const float* vec,
                                           float* x_out, float* vec_out, void* stream) {
    if (!h || !h->weights_set || layer < 0 || layer >= h->hp.num_layers || N != h->lastN) {
        adf_set_error("message_layer: bad handle/layer, or N differs from the built graph");
        return ADF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    ADF_TRY(zero_pad_rows(h, N, s));
    ADF_TRY(message_layer(h, layer, N, x, vec, x_out, vec_out, false, s));
    return ADF_OK;
}

extern "C" int32_t adf_painn_update_layer(adf_painn_t h, int32_t layer, int32_t N, float* x, float* vec, void* stream) {
    if (!h || !h->weights_set || layer < 0 || layer >= h->hp.num_layers) {
        adf_set_error("update_layer: bad handle/layer");
        return ADF_EINVAL;
    }
    ADF_TRY(ensure_capacity(h, N, 1));
    return update_layer(h, layer, N, x, vec, (hipStream_t)stream);
}

__global__ void adf_scatter_rows3_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int n,
                                         float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * n) dst[(size_t)idx[i / 3] * 3 + i % 3] = src[i];
}

static int32_t forward_impl(adf_painn_t h, const adf_batch* b, const int32_t* out_idx, int32_t n_out, float* f1,
                            float* f2, void* stream);

extern "C" int32_t adf_painn_forward(adf_painn_t h, const adf_batch* b, float* f1, float* f2, void* stream) {
    return forward_impl(h, b, nullptr, 0, f1, f2, stream);
}

extern "C" int32_t adf_painn_forward_subset(adf_painn_t h, const adf_batch* b, const int32_t* out_idx, int32_t n_out,
                                            float* f1, float* f2, void* stream) {
    if (!out_idx || n_out < 0) { adf_set_error("forward_subset: null index list"); return ADF_EINVAL; }
    return forward_impl(h, b, out_idx, n_out, f1, f2, stream);
}

// Incremental layers usable for this forward?  0 = no; 1 = yes and the kept state is current; 2 = yes, but every row has
// to be computed.  Allocates the kept state on first use (an allocation failure switches the feature off: the plain
// path needs none of it) and swaps the CSR buffers so that the coming graph build leaves the previous build's CSR in
// prev_*.  The state is marked invalid until forward_incremental has finished: an error on the way (graph overflow,
// launch failure) must not leave rows behind that a later forward would trust.
static int inc_prepare(adf_painn* h, int N) {
    const int L = h->hp.num_layers, H = h->hp.hidden_channels;
    if (!h->inc_on || !h->moving || h->gemm_f32 || h->msg_f32 || L > ADF_MAX_LAYERS || !h->prev_nptr)
        return 0;
    if (N > h->inc_capN) {
        (void)hipDeviceSynchronize();
        inc_free(h);
        const size_t cap = (size_t)N;
        int32_t st = ADF_OK;
        for (int l = 0; l <= L && st == ADF_OK; ++l) {
            st = dev_alloc(&h->incX[l], cap * H);
            if (st == ADF_OK && l >= 1) st = dev_alloc(&h->incV[l], cap * 3 * H);
            if (st == ADF_OK && l < L) st = dev_alloc(&h->incR[l], (cap + 1) * 5 * H);
        }
        if (st == ADF_OK) st = dev_alloc(&h->inc_c0, cap);
        if (st == ADF_OK) st = dev_alloc(&h->inc_chg, 2 * cap);
        if (st == ADF_OK) st = dev_alloc(&h->inc_pend, cap * L);
        if (st == ADF_OK) st = dev_alloc(&h->inc_need, cap * L);
        if (st == ADF_OK) st = dev_alloc(&h->inc_tf, cap * L);
        if (st == ADF_OK) st = dev_alloc(&h->inc_list, cap * L);
        if (st == ADF_OK) st = dev_alloc(&h->inc_cnt, (size_t)2 * ADF_MAX_LAYERS + 1);
        if (st == ADF_OK) {
            h->inc_tmp_bytes = adf_inc_temp_bytes((int64_t)cap);
            unsigned char* tmp = nullptr;
            st = dev_alloc(&tmp, h->inc_tmp_bytes + 16);
            h->inc_tmp = tmp;
        }
        if (st == ADF_OK && !h->inc_cnt_host &&
            hipHostMalloc(reinterpret_cast<void**>(&h->inc_cnt_host), sizeof(int32_t) * 2 * (2 * ADF_MAX_LAYERS + 1)) != hipSuccess)
            st = ADF_EOOM;
        for (int i = 0; i < 2 && st == ADF_OK; ++i)
            if (!h->inc_ev[i] && hipEventCreateWithFlags(reinterpret_cast<hipEvent_t*>(&h->inc_ev[i]), hipEventDisableTiming) != hipSuccess)
                st = ADF_EHIP;
        if (st != ADF_OK) {
            (void)hipGetLastError();
            inc_free(h);
            h->inc_on = false;
            fprintf(stderr, "adsorbdiff_hip: no memory for incremental layers (%d atoms), continuing without\n", N);
            return 0;
        }
        h->inc_capN = (int64_t)cap;
    }
    if (h->inc_N != N || h->inc_layers != L || h->build_serial != h->inc_serial) h->inc_valid = false;
    h->inc_N = N; h->inc_layers = L;
    int32_t* tn = h->nptr; h->nptr = h->prev_nptr; h->prev_nptr = tn;
    int32_t* ts = h->e_src; h->e_src = h->prev_src; h->prev_src = ts;
    float4* tg = h->e_geom; h->e_geom = h->prev_geom; h->prev_geom = tg;
    const int state = h->inc_valid ? 1 : 2;
    h->inc_valid = false;
    return state;
}

__global__ void adf_scatter_rows3_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int n,
                                         float* __restrict__ dst);

// Counts of an earlier forward that have reached the pinned buffer: remember them (they steer the next forwards' choice
// between the list and the all-rows form of a layer) and book that forward's statistics.  wait: block until they are there.
static int32_t inc_harvest(adf_painn* h, int slot, bool wait, bool peek = false) {
    // peek: only read the counts (they steer THIS forward, ADF_INC_SYNC=1); the statistics are booked by a later call, once
    // the forward has noted which form each layer took
    if (!h->inc_ev_live[slot]) return ADF_OK;
    hipEvent_t ev = (hipEvent_t)h->inc_ev[slot];
    if (wait) {
        ADF_HIP_CHECK(hipEventSynchronize(ev));
    } else {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); return ADF_OK; }
        ADF_HIP_CHECK(q);
    }
    const int L = h->inc_layers, N = h->inc_pend_N[slot];
    const int32_t* c = h->inc_cnt_host + (size_t)slot * (2 * ADF_MAX_LAYERS + 1);
    for (int i = 0; i < 2 * L + 1; ++i) h->inc_seen[i] = c[i];
    h->inc_seen_valid = true;
    h->inc_pend_Nseen = N;
    if (peek) return ADF_OK;
    h->inc_ev_live[slot] = false;
    for (int l = 0; l < L; ++l) {
        const bool whole = h->inc_pend_whole[slot][l] != 0;
        h->inc_rows += (unsigned long long)(whole ? N : c[l]);
        h->inc_rows_full += (unsigned long long)N;
        h->inc_edges += (unsigned long long)(whole ? c[2 * L] : c[L + l]);
        if (whole || c[l] > 0) ++h->inc_launches;
    }
    return ADF_OK;
}

// One forward on the kept per-layer state.  The graph of this step is built; prev_* hold the previous build's CSR.
static int32_t forward_incremental(adf_painn* h, int N, const int32_t* Z, const int32_t* out_idx, int32_t n_out,
                                   float* f1, float* f2, bool first, hipStream_t s) {
    const int L = h->hp.num_layers, H = h->hp.hidden_channels;
    const size_t cap = (size_t)h->inc_capN, row = (size_t)5 * H;
    if (out_idx && n_out == 0) return ADF_OK;  // nothing wanted; the state stays invalid (this build was not applied)
    adf_prof_begin(h, ADF_PROF_GRAPH, s);
    if (first) {
        ADF_TRY(adf_nodewise_embed(h, Z, N, h->incX[0], s));
        for (int l = 0; l < L; ++l) ADF_HIP_CHECK(hipMemsetAsync(h->incR[l] + (size_t)N * row, 0, sizeof(float) * row, s));
        ADF_HIP_CHECK(hipMemsetAsync(h->inc_pend, 0, cap * L, s));
    } else {
        ADF_TRY(adf_inc_compare(h, N, s));
    }
    ADF_HIP_CHECK(hipMemsetAsync(h->inc_cnt, 0, sizeof(int32_t) * (2 * ADF_MAX_LAYERS + 1), s));
    if (out_idx) ADF_TRY(adf_inc_need_from_list(h, N, L, out_idx, n_out, s));
    for (int l = 0; l < L; ++l) ADF_TRY(adf_inc_plan_layer(h, l, N, first, out_idx != nullptr, s));
    ADF_HIP_CHECK(hipMemcpyAsync(h->inc_cnt + 2 * L, h->nptr + N, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    // the counts go to the host behind an event; nobody waits for them (unless ADF_INC_SYNC=1)
    const int slot = h->inc_slot;
    h->inc_slot ^= 1;
    if (h->inc_ev_live[slot]) ADF_TRY(inc_harvest(h, slot, true));  // two forwards behind: cannot happen on one stream, but be safe
    int32_t* host = h->inc_cnt_host + (size_t)slot * (2 * ADF_MAX_LAYERS + 1);
    ADF_HIP_CHECK(hipMemcpyAsync(host, h->inc_cnt, sizeof(int32_t) * (2 * L + 1), hipMemcpyDeviceToHost, s));
    ADF_HIP_CHECK(hipEventRecord((hipEvent_t)h->inc_ev[slot], s));
    h->inc_ev_live[slot] = true;
    h->inc_pend_N[slot] = N;
    ADF_TRY(inc_harvest(h, slot ^ 1, false));   // the previous forward's counts, if they have arrived
    if (h->inc_sync) ADF_TRY(inc_harvest(h, slot, true, true));
    adf_prof_end(h, s);
    if (first) ADF_TRY(make_records(h, 0, N, h->incX[0], nullptr, true, h->incR[0], nullptr, s));
    for (int l = 0; l < L; ++l) {
        // all rows (straight into the next layer's tables) or the listed rows.  Recomputing a row that was not listed is
        // harmless: a row without pending changes has unchanged inputs and gets the same value again; a pending but
        // unwanted row keeps its flag and is recomputed before it is next read.  Above ~90 % the compaction costs more
        // than the rows it saves.  The choice rests on the latest counts the host has seen (this forward's with
        // ADF_INC_SYNC=1, else an earlier forward's: only the choice is late, never a result); none seen yet = all rows.
        const int n_seen = h->inc_seen_valid ? h->inc_seen[l] : N;
        const bool whole = first || !h->inc_seen_valid || h->inc_pend_Nseen != N || (long long)n_seen * 10 >= (long long)N * 9;
        h->inc_pend_whole[slot][l] = whole ? 1 : 0;
        if (h->inc_sync && !whole && n_seen == 0) continue;  // exact count: nothing to do
        const float* vin = l == 0 ? h->vecA : h->incV[l];  // layer 0: vec is zero, the pointer is not read
        if (whole) {
            ADF_TRY(message_layer(h, l, N, h->incX[l], vin, h->incX[l + 1], h->incV[l + 1], l == 0, s, nullptr, 0,
                                  h->incR[l], true));
            ADF_TRY(update_layer(h, l, N, h->incX[l + 1], h->incV[l + 1], s));
            if (l + 1 < L)
                ADF_TRY(make_records(h, l + 1, N, h->incX[l + 1], h->incV[l + 1], false, h->incR[l + 1], nullptr, s));
            continue;
        }
        // listed rows: launches sized for `cap` rows, the kernels stop at the list length they read from inc_cnt[l]
        const int32_t* list = h->inc_list + cap * l;
        const int ncap = h->inc_sync ? n_seen : N;
        h->rows_dev = h->inc_cnt + l;
        int32_t st = message_layer(h, l, N, h->incX[l], vin, h->x, h->vecB, l == 0, s, list, ncap, h->incR[l], true);
        if (st == ADF_OK) st = update_layer(h, l, ncap, h->x, h->vecB, s);
        if (st == ADF_OK) {
            adf_prof_begin(h, ADF_PROF_NODE, s);
            st = adf_inc_scatter_rows(h->x, list, ncap, H, h->incX[l + 1], s, h->rows_dev);
            if (st == ADF_OK) st = adf_inc_scatter_rows(h->vecB, list, ncap, 3 * H, h->incV[l + 1], s, h->rows_dev);
            adf_prof_end(h, s);
        }
        if (st == ADF_OK && l + 1 < L) st = make_records(h, l + 1, ncap, h->x, h->vecB, false, h->incR[l + 1], list, s);
        h->rows_dev = nullptr;
        ADF_TRY(st);
    }
    adf_prof_begin(h, ADF_PROF_HEADS, s);
    if (!out_idx) {
        ADF_TRY(adf_head_forward(h, 0, N, h->incX[L], h->incV[L], f1, s));
        if (h->hp.num_heads == 2) ADF_TRY(adf_head_forward(h, 1, N, h->incX[L], h->incV[L], f2, s));
    } else {
        if (n_out > h->capS) {
            const int64_t c = (int64_t)n_out + n_out / 4 + 64;
            if (h->sub_x) (void)hipFree(h->sub_x);
            if (h->sub_vec) (void)hipFree(h->sub_vec);
            if (h->sub_f) (void)hipFree(h->sub_f);
            h->sub_x = h->sub_vec = h->sub_f = nullptr; h->capS = 0;
            ADF_TRY(dev_alloc(&h->sub_x, (size_t)c * H));
            ADF_TRY(dev_alloc(&h->sub_vec, (size_t)c * 3 * H));
            ADF_TRY(dev_alloc(&h->sub_f, (size_t)c * 3));
            h->capS = c;
        }
        ADF_TRY(adf_inc_gather_rows(h->incX[L], out_idx, n_out, H, h->sub_x, s));
        ADF_TRY(adf_inc_gather_rows(h->incV[L], out_idx, n_out, 3 * H, h->sub_vec, s));
        for (int hd = 0; hd < h->hp.num_heads; ++hd) {
            ADF_TRY(adf_head_forward(h, hd, n_out, h->sub_x, h->sub_vec, h->sub_f, s));
            hipLaunchKernelGGL(adf_scatter_rows3_kernel, dim3((3 * n_out + 255) / 256), dim3(256), 0, s, h->sub_f, out_idx,
                               n_out, hd == 0 ? f1 : f2);
        }
        ADF_HIP_CHECK(hipGetLastError());
    }
    adf_prof_end(h, s);
    h->inc_valid = true;
    return ADF_OK;
}

static int32_t forward_impl(adf_painn_t h, const adf_batch* b, const int32_t* out_idx, int32_t n_out, float* f1,
                            float* f2, void* stream) {
    ADF_TRY(check_batch(h, b));
    if (!h->weights_set) { adf_set_error("weights not set"); return ADF_EINVAL; }
    if (!b->atomic_numbers || !f1 || (h->hp.num_heads == 2 && !f2)) { adf_set_error("null argument"); return ADF_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const int N = b->num_atoms;
    ADF_TRY(ensure_capacity(h, N, b->num_systems));
    const int inc = inc_prepare(h, N);
    adf_prof_begin(h, ADF_PROF_GRAPH, s);
    ADF_TRY(adf_graph_build_impl(h, b, s));
    adf_prof_end(h, s);
    if (inc) {
        h->inc_serial = h->build_serial;
        return forward_incremental(h, N, b->atomic_numbers, out_idx, n_out, f1, f2, inc == 2, s);
    }
    ADF_TRY(zero_pad_rows(h, N, s));
    ADF_TRY(adf_nodewise_embed(h, b->atomic_numbers, N, h->x, s));  // vec = 0 is implicit in layer 0
    float* vin = h->vecA;
    float* vout = h->vecB;
    const int L = h->hp.num_layers, H = h->hp.hidden_channels;
    // layer-0 records: invariant while the static-atom promise holds (same batch and weights)
    float* rec0 = nullptr;
    bool rec0_ready = false;
    if (h->moving) {
        const size_t row = (size_t)5 * H;
        if ((int64_t)N + 1 > h->rec0_cap) {
            if (h->rec0) { (void)hipFree(h->rec0); h->rec0 = nullptr; }
            h->rec0_cap = 0; h->rec0_valid = false;
            ADF_TRY(dev_alloc(&h->rec0, ((size_t)N + 1) * row));
            h->rec0_cap = (int64_t)N + 1;
        }
        rec0 = h->rec0;
        rec0_ready = h->rec0_valid && h->rec0_N == N;
        if (!rec0_ready) ADF_HIP_CHECK(hipMemsetAsync(rec0 + (size_t)N * row, 0, sizeof(float) * row, s));
        h->rec0_valid = true; h->rec0_N = N;
    }
    if (!out_idx) {
        for (int l = 0; l < L; ++l) {
            if (l == 0) ADF_TRY(message_layer(h, 0, N, h->x, vin, h->x, vout, true, s, nullptr, 0, rec0, rec0_ready));
            else
            ADF_TRY(message_layer(h, l, N, h->x, vin, h->x, vout, l == 0, s));
            ADF_TRY(update_layer(h, l, N, h->x, vout, s));
            float* t = vin; vin = vout; vout = t;
        }
        adf_prof_begin(h, ADF_PROF_HEADS, s);
        ADF_TRY(adf_head_forward(h, 0, N, h->x, vin, f1, s));
        if (h->hp.num_heads == 2) ADF_TRY(adf_head_forward(h, 1, N, h->x, vin, f2, s));
        adf_prof_end(h, s);
        return ADF_OK;
    }
    // Outputs wanted on a subset only.  Every layer but the last is needed in full (the listed atoms' messages
    // gather from all their neighbours); in the last layer only the listed atoms are message targets, and their
    // update and the heads run on compact rows.  Per-row arithmetic is unchanged, so the listed rows of f1 / f2
    // are bit-identical to adf_painn_forward's.
    if (n_out == 0) return ADF_OK;
    if (n_out > h->capS) {
        const int64_t cap = (int64_t)n_out + n_out / 4 + 64;
        if (h->sub_x) (void)hipFree(h->sub_x);
        if (h->sub_vec) (void)hipFree(h->sub_vec);
        if (h->sub_f) (void)hipFree(h->sub_f);
        h->sub_x = h->sub_vec = h->sub_f = nullptr; h->capS = 0;
        ADF_TRY(dev_alloc(&h->sub_x, (size_t)cap * H));
        ADF_TRY(dev_alloc(&h->sub_vec, (size_t)cap * 3 * H));
        ADF_TRY(dev_alloc(&h->sub_f, (size_t)cap * 3));
        h->capS = cap;
    }
    for (int l = 0; l + 1 < L; ++l) {
        if (l == 0) ADF_TRY(message_layer(h, 0, N, h->x, vin, h->x, vout, true, s, nullptr, 0, rec0, rec0_ready));
        else
        ADF_TRY(message_layer(h, l, N, h->x, vin, h->x, vout, l == 0, s));
        ADF_TRY(update_layer(h, l, N, h->x, vout, s));
        float* t = vin; vin = vout; vout = t;
    }
    ADF_TRY(message_layer(h, L - 1, N, h->x, vin, h->sub_x, h->sub_vec, L == 1, s, out_idx, n_out,
                          L == 1 ? rec0 : nullptr, L == 1 && rec0_ready));
    ADF_TRY(update_layer(h, L - 1, n_out, h->sub_x, h->sub_vec, s));
    adf_prof_begin(h, ADF_PROF_HEADS, s);
    for (int hd = 0; hd < h->hp.num_heads; ++hd) {
        ADF_TRY(adf_head_forward(h, hd, n_out, h->sub_x, h->sub_vec, h->sub_f, s));
        hipLaunchKernelGGL(adf_scatter_rows3_kernel, dim3((3 * n_out + 255) / 256), dim3(256), 0, s, h->sub_f, out_idx,
                           n_out, hd == 0 ? f1 : f2);
    }
    ADF_HIP_CHECK(hipGetLastError());
    adf_prof_end(h, s);
    return ADF_OK;
}

// Stand-alone nn.Linear (unit tests of the two GEMM kernels): C = act(A . W^T + b)
extern "C" int32_t adf_linear_forward(const float* A, const float* W, const float* bias, float* C, int32_t M,
                                      int32_t N, int32_t K, int32_t act_ssilu, int32_t mode, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) { adf_set_error("bad argument"); return ADF_EINVAL; }
    if (mode == 0) return adf_launch_gemm(A, K, W, K, bias, C, N, M, N, K, act_ssilu, s);
    unsigned char* buf = nullptr;
    const size_t n = (size_t)N * K;
    ADF_TRY(dev_alloc(&buf, n * 4 + 64));
    adf_w16 w16 = {};
    w16.hi = buf; w16.lo = buf + n * 2; w16.inv_scale = reinterpret_cast<float*>(buf + n * 4);
    unsigned int* scratch = reinterpret_cast<unsigned int*>(buf + n * 4 + 16);
    int32_t st = adf_split_weight(W, (long long)n, &w16, scratch, s);
    float* mags = nullptr;
    const char* el = getenv("ADF_LIFT");
    const bool lift = !(el && strcmp(el, "0") == 0);
    if (st == ADF_OK && lift) st = dev_alloc(&mags, (size_t)M);
    adf_lift lf = {mags, M};
    if (st == ADF_OK) st = adf_launch_gemm16(A, K, &w16, bias, C, N, M, N, K, act_ssilu, s, nullptr, 0, lift ? &lf : nullptr);
    (void)hipStreamSynchronize(s);
    (void)hipFree(buf);
    if (mags) (void)hipFree(mags);
    return st;
}

// 0 = f16x3 split products on the f16 matrix cores (default), 1 = exact f32 MFMA everywhere (what ADF_GEMM=f32 selects
// at creation).  Takes effect with the next forward; the weight images of both modes are kept by adf_painn_set_weights.
extern "C" int32_t adf_painn_set_arithmetic(adf_painn_t h, int32_t exact_f32) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    h->gemm_f32 = exact_f32 != 0;
    h->msg_f32 = exact_f32 != 0;
    h->rec0_valid = false;
    h->inc_valid = false;
    return ADF_OK;
}

// Incremental layers (see incremental.hip): 1 = keep per-layer node state across the forwards of a static-atom promise
// and recompute only rows whose inputs changed (default; bit-identical outputs), 0 = every forward computes every row.
// Resets the row counters reported by adf_get_counters.
extern "C" int32_t adf_painn_set_fused_mlp(adf_painn_t h, int32_t mode) {
    if (!h || mode < 0 || mode > 2) { adf_set_error("set_fused_mlp: mode must be 0 (never), 1 (always) or 2 (by size)"); return ADF_EINVAL; }
    h->fused_mlp = mode;
    return ADF_OK;
}

extern "C" int32_t adf_painn_set_incremental(adf_painn_t h, int32_t on) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    if ((on != 0) != h->inc_on) {
        ADF_HIP_CHECK(hipDeviceSynchronize());
        h->inc_on = on != 0;
        if (!h->inc_on) inc_free(h);
        if (h->inc_on && h->capN > 0 && !h->prev_nptr) {  // workspaces exist without the previous-CSR buffers
            int32_t st = dev_alloc(&h->prev_nptr, (size_t)h->capN + 1);
            if (st == ADF_OK) st = dev_alloc(&h->prev_src, (size_t)h->capE);
            if (st == ADF_OK) st = dev_alloc(&h->prev_geom, (size_t)h->capE);
            if (st != ADF_OK) { h->inc_on = false; return st; }
        }
    }
    h->inc_valid = false;
    (void)inc_harvest(h, h->inc_slot, true);
    (void)inc_harvest(h, h->inc_slot ^ 1, true);
    h->inc_seen_valid = false;
    h->inc_rows = h->inc_rows_full = h->inc_edges = h->inc_launches = 0;
    return ADF_OK;
}

extern "C" int32_t adf_check_flags(adf_painn_t h, void* stream) {
    if (!h) { adf_set_error("null handle"); return ADF_EINVAL; }
    return read_flags(h, (hipStream_t)stream);
}

extern "C" int32_t adf_sde_init_placement(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags,
                                          const float* noise, void* stream) {
    ADF_TRY(check_batch(h, b));
    if (!pos || !tags || !noise) { adf_set_error("null argument"); return ADF_EINVAL; }
    return adf_stepper_init(b, pos, tags, noise, (hipStream_t)stream);
}

static int32_t sde_step_common(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags,
                               const int32_t* fixed, const float* f1, const float* f2, const adf_step_coef* coef,
                               const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr, const float* z_rot,
                               int32_t early_stop_count, int32_t* state, float* dcom, float* drot, void* stream) {
    ADF_TRY(check_batch(h, b));
    if (!pos || !tags || !f1 || !f2 || (!coef && !coefs_dev) || !state) { adf_set_error("null argument"); return ADF_EINVAL; }
    ADF_TRY(ensure_capacity(h, b->num_atoms, b->num_systems));
    adf_prof_begin(h, ADF_PROF_STEPPER, (hipStream_t)stream);
    const int32_t st = adf_stepper_step(h->sys, b, pos, tags, fixed, f1, f2, coef, coefs_dev, num_steps, z_tr, z_rot,
                                        early_stop_count, state, dcom, drot, (hipStream_t)stream);
    adf_prof_end(h, (hipStream_t)stream);
    return st;
}

extern "C" int32_t adf_sde_step(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags,
                                const int32_t* fixed, const float* f1, const float* f2, const adf_step_coef* coef,
                                const float* z_tr, const float* z_rot, int32_t early_stop_count, int32_t* state,
                                float* dcom, float* drot, void* stream) {
    return sde_step_common(h, b, pos, tags, fixed, f1, f2, coef, nullptr, 0, z_tr, z_rot, early_stop_count, state, dcom,
                           drot, stream);
}

extern "C" int32_t adf_sde_step_scheduled(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags,
                                          const int32_t* fixed, const float* f1, const float* f2,
                                          const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr,
                                          const float* z_rot, int32_t early_stop_count, int32_t* state, float* dcom,
                                          float* drot, void* stream) {
    if (num_steps <= 0) { adf_set_error("num_steps must be positive"); return ADF_EINVAL; }
    return sde_step_common(h, b, pos, tags, fixed, f1, f2, nullptr, coefs_dev, num_steps, z_tr, z_rot, early_stop_count,
                           state, dcom, drot, stream);
}

// The whole reverse loop (denoising_torch.py:235-356) in one call: num_steps x (graph build + forward + step), all
// on `stream`.  Host round trips: with poll_every > 0 the frozen flag is read back every poll_every steps and the loop
// ends early, exactly like the reference's `break`; with incremental layers on, each forward reads its list lengths.
struct adf_frames;
int32_t adf_frames_push_impl(adf_frames* f, const float* src, hipStream_t s);

static int32_t sample_impl(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                           const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr_all,
                           const float* z_rot_all, int32_t early_stop_count, int32_t poll_every, int32_t* state,
                           const int32_t* out_idx, int32_t n_out, float* f1, float* f2, adf_frames* sink,
                           int32_t frame_every, void* stream) {
    ADF_TRY(check_batch(h, b));
    if (num_steps <= 0 || !f1 || !f2 || !state || !coefs_dev || !pos || !tags) { adf_set_error("sample: bad argument"); return ADF_EINVAL; }
    if ((z_tr_all == nullptr) != (z_rot_all == nullptr)) { adf_set_error("sample: need both noise tables or none"); return ADF_EINVAL; }
    if (sink && frame_every <= 0) { adf_set_error("sample: frame_every must be positive"); return ADF_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const size_t zs = (size_t)b->num_systems * 3;
    for (int t = 0; t < num_steps; ++t) {
        ADF_TRY(forward_impl(h, b, out_idx, out_idx ? n_out : 0, f1, f2, stream));
        ADF_TRY(sde_step_common(h, b, pos, tags, fixed, f1, f2, nullptr, coefs_dev, num_steps,
                                z_tr_all ? z_tr_all + t * zs : nullptr, z_rot_all ? z_rot_all + t * zs : nullptr,
                                early_stop_count, state, nullptr, nullptr, stream));
        // trajectory frame of this step: snapshot on this stream, copy-out on the sink's stream (frames.hip)
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

extern "C" int32_t adf_sample(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                              const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr_all,
                              const float* z_rot_all, int32_t early_stop_count, int32_t poll_every, int32_t* state,
                              const int32_t* out_idx, int32_t n_out, float* f1, float* f2, void* stream) {
    return sample_impl(h, b, pos, tags, fixed, coefs_dev, num_steps, z_tr_all, z_rot_all, early_stop_count, poll_every, state,
                       out_idx, n_out, f1, f2, nullptr, 0, stream);
}

extern "C" int32_t adf_sample_traj(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                                   const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr_all,
                                   const float* z_rot_all, int32_t early_stop_count, int32_t poll_every, int32_t* state,
                                   const int32_t* out_idx, int32_t n_out, float* f1, float* f2, adf_frames_t sink,
                                   int32_t frame_every, void* stream) {
    if (!sink) { adf_set_error("sample_traj: null sink"); return ADF_EINVAL; }
    return sample_impl(h, b, pos, tags, fixed, coefs_dev, num_steps, z_tr_all, z_rot_all, early_stop_count, poll_every, state,
                       out_idx, n_out, f1, f2, sink, frame_every, stream);
}

extern "C" int32_t adf_get_counters(adf_painn_t h, adf_counters* out, void* stream) {
    if (!h || !out || h->lastN <= 0) { adf_set_error("no forward has run"); return ADF_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const int64_t N = h->lastN, H = h->hp.hidden_channels, R = h->hp.num_rbf, L = h->hp.num_layers;
    int32_t e = 0;
    ADF_HIP_CHECK(hipMemcpyAsync(&e, h->nptr + N, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ADF_HIP_CHECK(hipStreamSynchronize(s));
    const int64_t E = e;
    out->num_edges = E;
    out->num_atoms = N;
    // E*(3H*4 + 12 + 8) + N*(3H*4 + 3H*4) + N*(H*4 + 3H*4)
    out->message_bytes_per_layer = E * (3 * H * 4 + 12 + 8) + N * (3 * H * 4 + 3 * H * 4) + N * (H * 4 + 3 * H * 4);
    // per layer: node side 30 H^2 N, edge side 2 R 3H E; heads: per head 2*(3N*H*H + 3N*H*H/2 + N*2H*H + N*H*H
    //            + 3N*(H/2)^2 + N*H*H/2)
    const int64_t head = 2 * (3 * N * H * H + 3 * N * H * H / 2 + N * 2 * H * H + N * H * H + 3 * N * (H / 2) * (H / 2) +
                              N * H * H / 2);
    out->dense_flops = L * (30 * H * H * N + 2 * R * 3 * H * E) + h->hp.num_heads * head;
    ADF_TRY(inc_harvest(h, h->inc_slot, true));      // the older of the two outstanding forwards first
    ADF_TRY(inc_harvest(h, h->inc_slot ^ 1, true));
    out->inc_rows = (int64_t)h->inc_rows; out->inc_rows_full = (int64_t)h->inc_rows_full;
    out->inc_msg_launches = (int64_t)h->inc_launches; out->inc_msg_edges = (int64_t)h->inc_edges;
    return ADF_OK;
}
