// Internal declarations of the EquiformerV2 denoiser path (BASELINE config 4).
// Reference: adsorbdiff/models/equiformer_v2/equiformer_v2_denoising.py:185-318 and the modules it calls.
#pragma once
#include "common.h"

#define EQ_MAX_L 6        // register-array kernels are instantiated for lmax = 1..6
#define EQ_MAX_M 6
#define EQ_MAX_LAYERS 32
#define EQ_NFLAGS 8

// nn.Linear weight [out, in] (+ bias) as bound by adf_eqv2_set_weights, plus the library's fp16 hi/lo image
struct eq_lin {
    const float* w;
    const float* b;
    int out, in;
    adf_w16 w16;
    bool has16;
};
struct eq_radial {      // RadialFunction: Linear, LayerNorm, SiLU, Linear, LayerNorm, SiLU, Linear (radial_function.py:11-32)
    eq_lin l0, l3, l6;
    const float *ln1_w, *ln1_b, *ln4_w, *ln4_b;
    float* w0t;         // library-owned transpose of l0.w: [in, out]
    float* table;       // [NE*NE, l6.out] radial weights per element pair (static-radial mode), else null
};
struct eq_norm {        // EquivariantLayerNormArraySphericalHarmonics (layer_norm.py:129-250)
    const float *affine, *l0_w, *l0_b;
};
struct eq_attn {        // SO2EquivariantGraphAttention (transformer_block.py:38-372)
    const float *alpha_dot, *src_emb, *dst_emb, *alpha_ln_w, *alpha_ln_b;
    eq_lin c1_m0, c1_m[EQ_MAX_M], c2_m0, c2_m[EQ_MAX_M];
    eq_radial rad;
    const float *proj_w, *proj_b;  // SO3_LinearV2 [L+1, out, HV], bias [out]
    eq_lin proj_l[EQ_MAX_L + 1];   // per-degree views of proj_w
    int out_channels;
};
struct eq_ffn {         // FeedForwardNetwork with grid MLP + separable S2 activation (transformer_block.py:375-531)
    const float *l1_w, *l1_b, *l2_w, *l2_b;
    eq_lin l1[EQ_MAX_L + 1], l2[EQ_MAX_L + 1];
    eq_lin scalar, g0, g2, g4;
    // The grid MLP's first and last maps act on channels only and have no bias, the grid transforms act on coefficients
    // only: grid_mlp[0] commutes with to_grid and grid_mlp[4] with from_grid.  l1f[l] = g0 . l1[l] and (l >= 1)
    // l2f[l] = l2[l] . g4 are formed once per weight binding, so that only ONE product runs on the 324-point grid rows.
    eq_lin l1f[EQ_MAX_L + 1], l2f[EQ_MAX_L + 1];
};
struct eq_block {
    eq_norm n1, n2;
    eq_attn ga;
    eq_ffn ffn;
};

// index bookkeeping shared by host and kernels (passed by value)
struct eq_dims {
    int L, M, C, S, Sr, DR, G;
    int Hd, NH, A, V, HV, F, EC, NB;
    int d_off[EQ_MAX_L + 2];   // offset of degree l inside an edge's Wigner rows
    int nrow[EQ_MAX_L + 1];    // kept rows of degree l: 2 min(l, M) + 1
    int rbase[EQ_MAX_M + 1];   // first m-major reduced index of order m (m = 0: 0)
    int rad_off[EQ_MAX_M + 1]; // column offset of order m in a radial row, in units of the input channel count
    int RW;                    // sum_m (L - m + 1)
    int presplit_rows;         // rotate-in with fp16 hi/lo output: rows of the order-0 operand buffer (set per launch)
    int j_off[EQ_MAX_L + 2];   // offset of J_l in the concatenated table
    short r_m[64], r_sgn[64], r_l[64];  // m-major reduced index -> (order m, 0: +m or m = 0 / 1: -m, degree l)
    float resc[EQ_MAX_L + 1];  // m-truncation rescale sqrt((2l+1)/(2M+1)) for l > M (so3.py:160-186)
};

// A operand / C result row addressing of the dense kernels: row r lives at base + (r / period) * outer + (r % period) * inner
struct eq_rowmap {
    long long outer;
    int period, inner;
};

struct adf_eqv2 {
    adf_eqv2_hparams hp;
    eq_dims d;
    eq_dims* d_dev;   // device copy (kernels that index its tables per lane)
    int device, num_cus;
    bool weights_set, consts_set, exact_f32, s2_emit_mag, presplit;
    bool no_compact;           // ADF_EQV2_COMPACT=0: force blocks evaluate every column of their second convolution
    bool fold_on, folded;      // grid-MLP end maps folded into the SO(3) linears (eq_ffn); ADF_EQV2_FOLD=0 switches it off
    float* fold_arena; size_t fold_floats;
    // constants (device)
    float *jd, *to_red, *from_red, *to_full, *from_full;
    // weights
    const float* atom_radii;
    const float* sphere_emb;
    const float *ed_src_emb, *ed_dst_emb;
    eq_radial ed_rad;
    eq_block blk[EQ_MAX_LAYERS];
    eq_norm final_norm;
    eq_attn force[2];
    unsigned char* w16_arena; size_t w16_bytes; float* w16_scales; unsigned int* w16_scratch;
    unsigned char* wfrag_arena;   // fragment images (adf_w16::frag) of the split weights, for eq_launch_gemm16pw
    bool conv1_wr;                // first convolution with the weights streamed as fragments (ADF_EQV2_CONV1_WR, default on)
    bool alpha_generic;           // attention logits by the one-head-at-a-time kernel for every width (ADF_EQV2_ALPHA_GENERIC=1)
    bool conv2_wr;                // plain products on whole 256-column tiles through gemm16.hip's streamed-fragment kernel (ADF_EQV2_CONV2_WR)
    float* wt_arena; size_t wt_bytes;   // transposed first radial layers
    float* rtab_arena; size_t rtab_floats; bool rad_static;   // per-element-pair radial tables (see eq_radial_static)
    // graph
    int64_t capN, capB, capE;
    int32_t *nbr_cnt, *nbr_src, *nbr_shift, *img_cnt, *eptr, *e_src, *e_dst, *flags;
    float* e_vec;     // [capE,3]
    float* wig;       // [capE, DR]
    void* scan_tmp; size_t scan_tmp_bytes;
    const int32_t *moving, *mov_idx, *mov_off;
    float* cache_d2; int32_t *cache_cid, *cache_cnt; bool cache_valid;
    bool ext_graph;   // edges were supplied by adf_eqv2_set_edges
    int64_t E_ext; int maxdeg;
    int32_t *xe_src, *xe_dst; float* xe_vec; int64_t xe_cap;   // the caller's edge list (private copy)
    // incoming edges of a listed subset of targets, compacted (the force blocks of adf_eqv2_forward_subset)
    int32_t *sub_eptr, *sub_src, *sub_dst; float *sub_vec, *sub_wig, *sub_f; int64_t sub_cap;
    int64_t last_subset;       // n_out of the last forward, -1 = all atoms
    int64_t arena_kk;   // edges per target the chunk arena was sized for
    // node buffers
    float *x, *y, *agg, *gate, *h1, *h2;
    // chunk arena
    int64_t chunk_nodes;          // targets per chunk
    float* arena; size_t arena_floats;
    float* garena; size_t garena_floats;   // grid MLP buffers of a node chunk
    float* sys;
    void* s2tab; int s2_npb; float s2_inv_sT, s2_inv_sF, s2_gain_shift;  // fragment-order fp16 hi/lo images of to_red / from_red
    void *gtab_to, *gtab_from; int g_npb, g_nkst; float g_inv_sT, g_inv_sF;  // likewise for to_full / from_full
    float* rs; int64_t rs_cap;   // per-row magnitudes max|a| of the A operand of an f16x3 product (-> power-of-two lift)
    int64_t lastN;
    // incremental blocks (adf_eqv2_set_incremental): the embedding and every block's output are kept across the forwards
    // of a static-atom run and a block recomputes only the rows whose inputs changed (eq_launch_inc_lists)
    bool inc_on, inc_valid;
    int64_t inc_N, inc_capN, inc_capE;
    int inc_nl;
    float* inc_x[EQ_MAX_LAYERS + 1];   // [capN, S, C] each: embedding, then the blocks' outputs
    float* inc_xg;                     // the rows being recomputed, compact
    int32_t *inc_peptr, *inc_psrc; float* inc_pvec;   // the previous forward's graph
    unsigned char* inc_dirty;          // [2][capN]
    int32_t *inc_idx, *inc_cnt;        // [nl][capN] ascending row lists, [nl] their lengths
    void* inc_sel_tmp; size_t inc_sel_bytes;
    int64_t inc_rows, inc_rows_full;   // block rows recomputed / block rows of full forwards since set_incremental
    int64_t last_block_rows;           // block rows the last forward computed (all blocks: layers x N)
    int64_t prof_block_rows, prof_forwards;   // the same summed over the forwards since adf_eqv2_profile_enable(1)
    // HIP-event timing per kernel group (bench.py roofline)
    bool prof_on;
    std::vector<hipEvent_t>* prof_ev;
    std::vector<int>* prof_cat;
    size_t prof_used;
};

enum { EQ_PROF_GRAPH = 0, EQ_PROF_RADIAL, EQ_PROF_ROTATE, EQ_PROF_CONV, EQ_PROF_S2ACT, EQ_PROF_ATTN, EQ_PROF_NODE,
       EQ_PROF_FFN, EQ_PROF_STEPPER, EQ_PROF_NCAT };

// ---- launchers (eqv2_kernels.hip)
int32_t eq_launch_edges_from_topk(adf_eqv2* h, const adf_batch* b, hipStream_t s);
int32_t eq_launch_eptr_from_dst(adf_eqv2* h, int N, int64_t E, hipStream_t s);
int32_t eq_launch_wigner(adf_eqv2* h, int N, hipStream_t s);
int32_t eq_launch_check_z(const adf_eqv2* h, const int32_t* Z, int N, hipStream_t s);
int32_t eq_launch_norm(const adf_eqv2* h, const eq_norm* nm, const float* x, float* y, int N, hipStream_t s);
int32_t eq_launch_radial_pre(const adf_eqv2* h, const eq_radial* r, const float* src_emb, const float* dst_emb,
                             const int32_t* Z, int n0, int n1, float* out, int N, hipStream_t s);
int32_t eq_launch_ln_silu(float* x, const float* w, const float* b, long long rows, int width, hipStream_t s);
// pair_ne > 0: m0 / rad are tables with one row per element pair (Z_src * pair_ne + Z_tgt) instead of one row per edge
int32_t eq_launch_edge_degree(const adf_eqv2* h, const float* m0, const int32_t* Z, int pair_ne, int n0, int n1, float* x,
                              hipStream_t s);
int32_t eq_launch_radial_pre_pairs(const adf_eqv2* h, const eq_radial* r, const float* src_emb, const float* dst_emb,
                                   float* out, hipStream_t s);
// rsp (optional): per-order arrays that receive the power-of-two lifts of the operand rows it writes
// presplit: write the operand rows as lifted fp16 hi / lo images (for eq_launch_gemm16p) instead of fp32
int32_t eq_launch_rotate_in(const adf_eqv2* h, const float* y, const float* rad, const int32_t* Z, int pair_ne, int n0, int n1,
                            float* const* mbuf, float* const* rsp, bool presplit, hipStream_t s);
int32_t eq_launch_gemm16p(const void* Ahi, const void* Alo, const float* mag, const adf_w16* W, const float* bias, float* Cm,
                          int ldc, long long M, int N, int K, int act, hipStream_t s);
// the same product with the weights streamed from their fragment image (W->frag); eq_gemm16pw_ok: shapes it takes
bool eq_gemm16pw_ok(const adf_w16* W, int N, int K);
int32_t eq_launch_gemm16pw(const void* Ahi, const void* Alo, const float* mag, const adf_w16* W, const float* bias, float* Cm,
                           int ldc, long long M, int N, int K, int act, hipStream_t s);
// rsp (optional): per-order arrays that receive the power-of-two lifts of the output rows (matrix-core version only;
// *rs_written tells whether they were filled)
int32_t eq_launch_s2act(const adf_eqv2* h, const float* y0, float* const* ym, int extra, int gate_off, int n0, int n1,
                        float* const* mb, float* const* rsp, bool* rs_written, hipStream_t s);
int32_t eq_launch_alpha(const adf_eqv2* h, const eq_attn* at, const float* y0, int ldy, int n0, int n1, float* alpha,
                        hipStream_t s);
int32_t eq_launch_rotate_out(const adf_eqv2* h, float* const* z, const float* alpha, int n0, int n1, float* agg,
                             bool only_l1, hipStream_t s, bool compact = false);
// silu: apply SiLU to the grid values (the folded feed-forward network, eq_ffn)
// node_mag != null: the matrix-core kernel also leaves max |g| of every node's tile in node_mag[n - n0] (zeroed here) and
// sets *emitted; the VALU kernel does not
int32_t eq_launch_to_grid(const adf_eqv2* h, const float* h1, int n0, int n1, float* g, bool silu, hipStream_t s,
                          float* node_mag = nullptr, bool* emitted = nullptr);
int32_t eq_launch_fold(const float* A, const float* B, int O, int K, int I, float* out, hipStream_t s);
int32_t eq_launch_from_grid(const adf_eqv2* h, const float* g, const float* gate, int n0, int n1, float* h2, hipStream_t s);
int32_t eq_launch_force_out(const adf_eqv2* h, const eq_attn* at, const float* agg3, int N, float* f, hipStream_t s);
// CSR + edge arrays of the listed targets only (sub_* of the handle); then rows of f_sub scattered to rows out_idx of f
int32_t eq_launch_subset_graph(adf_eqv2* h, const int32_t* out_idx, int n_out, hipStream_t s);
int32_t eq_launch_scatter_rows3(const float* f_sub, const int32_t* out_idx, int n_out, float* f, hipStream_t s);
// incremental blocks: recompute lists of blocks 0..nl-1 from a bit-for-bit comparison with the kept graph; keep the graph
size_t eq_inc_select_bytes(int N);
int32_t eq_launch_inc_lists(adf_eqv2* h, int N, int nl, hipStream_t s);
int32_t eq_launch_inc_keep(adf_eqv2* h, int N, hipStream_t s);
int32_t eq_launch_gather_rows(const float* src, const int32_t* idx, int n, int row_floats, float* dst, hipStream_t s);
int32_t eq_launch_scatter_rows(const float* src, const int32_t* idx, int n, int row_floats, float* dst, hipStream_t s);
int32_t eq_gemm_f32(const float* A, int lda, const eq_rowmap* amap, const float* W, const float* bias, float* Cm, int ldc,
                    const eq_rowmap* cmap, long long M, int N, int K, int act, bool accumulate, hipStream_t s);
int32_t eq_launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t s);
// C (+)= act(A . W^T + b): exact f32 for any shape; act: 0 none, 2 SiLU
int32_t eq_gemm(const adf_eqv2* h, const float* A, int lda, const eq_rowmap* amap, const eq_lin* W, bool use_bias,
                float* Cm, int ldc, const eq_rowmap* cmap, long long M, int act, bool accumulate, hipStream_t s,
                const float* rs_pre = nullptr, float* out_mag = nullptr, int rs_div = 1);  // rs_div: row r reads rs_pre[r / rs_div]
