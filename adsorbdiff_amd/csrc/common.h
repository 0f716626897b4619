// Internal declarations shared by the HIP translation units of libadsorbdiff_hip.so.
// gfx950 (MI355X) only: 64-wide wavefronts, f32 MFMA 32x32x2, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/adsorbdiff_hip.h"

#define ADF_GROUP_NODES 32      // target nodes per message-kernel work item
#define ADF_SLICE_CH 64         // channels per message-kernel slice (x3 parts = 192 MFMA columns)
#define ADF_MAX_CAND 1024       // in-cutoff candidates per centre held in LDS by the top-K kernel
#define ADF_MAX_K 128
#define ADF_NFLAGS 8
#define ADF_MAX_INDEG 1024
#define ADF_MAX_LAYERS 16      // incoming edges per target the per-target sorter handles (graph.hip)

void adf_set_error(const char* fmt, ...);

#define ADF_HIP_CHECK(expr)                                                             \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            adf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                          __FILE__, __LINE__);                                          \
            return (_e == hipErrorOutOfMemory) ? ADF_EOOM : ADF_EHIP;                   \
        }                                                                               \
    } while (0)

#define ADF_TRY(expr)                     \
    do {                                  \
        int32_t _s = (expr);              \
        if (_s != ADF_OK) return _s;      \
    } while (0)

// fp16 hi/lo split of one nn.Linear weight (gemm16.hip), library-owned
struct adf_w16 {
    void* hi;
    void* lo;
    float* inv_scale;  // device scalar: 1 / (power-of-two scale applied before the split)
    float* bias_perm;  // row-permuted bias of the fused 3H-wide layers (else null)
    void* frag;        // fragment-ordered image of hi / lo (adf_pack_frag, mlp16.hip): the B operands of v_mfma_f32_32x32x16_f16
                       // as the lanes load them, for the kernels that stream weights straight into registers (null: none)
};
// operands of the fused GEMM epilogues (gemm16.hip)
struct adf_epi {
    const float* vec_in;  // EPI 1: vec [N,3,H] (P_i = vec_i * xb)
    float* rec;           // EPI 1: gather records
    float* x;             // EPI 2
    float* vec;           // EPI 2
    const float* dot;     // EPI 2
    const float* vv;      // EPI 2: v1 [N,3,H] (written by EPI 3)
    float* v1;            // EPI 3: v1 out [N,3,H]
    float* dotw;          // EPI 3: dot out [N,H]
    float* cat;           // EPI 3: |v2| [N,H] ; EPI 4: norm [M,N]
    float scale;          // EPI 2: ScaleFactor
    int H;
    int vec_is_zero;      // EPI 1
    const float* A2;      // any EPI: second source of the A operand for columns [K1, K) (same row stride), K1 % 32 == 0
    int K1;               // 0 = single source
    const int32_t* row_map;  // EPI 1: record row of tile row n is row_map[n] (compact rows of an incremental layer); null = n
    // any EPI: per-row magnitudes max|a| of the A rows (indexed like the rows are loaded: atom*3 + component for EPI 3/4).
    // Each row is lifted by its own power of two before the fp16 hi/lo split and the lift is divided out in the epilogue
    // (an unlifted element below ~0.1 loses its a_lo term to the matrix core's subnormal flush).  null = no lift.
    const float* rmag;
    unsigned int* out_mag;   // EPI 0: receives max|c| of every output row (atomicMax on float bits; zeroed by the launcher)
    // any EPI: the row count lives on the device (an incremental layer's recompute list, incremental.hip): rows =
    // min(M, *m_dev); the launch is sized for M.  null = M.
    const int32_t* m_dev;
    int accumulate;          // EPI 0: C += A W^T (+ bias) instead of C = (the training step's accumulated data gradients)
    int lift_y;              // mlp16.hip: lift the intermediate rows by their own power of two (= the engine's lift_on)
    const float* gate;       // EPI 0 (heads): multiply output row r, column c by gate[(r / 3) * gate_ld + c]  (null: no gate)
    int gate_ld;
    long long rec_rows;      // mlp16.hip EPI 1: rows of the record table the (mapped) rows are written into (0 = the launch's rows)
};
// scratch for the row magnitudes a launcher measures itself (adf_launch_rowmag) when the caller has none to hand over
struct adf_lift {
    float* buf;
    long long cap;   // rows
};
struct adf_layer_weights {
    const float *ln_w, *ln_b, *xp0_w, *xp0_b, *xp2_w, *xp2_b, *rbf_w, *rbf_b;
    const float *vp_w, *xv0_w, *xv0_b, *xv2_w, *xv2_b;
    adf_w16 xp0_16, xp2_16, vp_16, xv0_16, xv2_16;
};
struct adf_block_weights {
    const float *vec1_w, *vec2_w, *un0_w, *un0_b, *un2_w, *un2_b;
    adf_w16 vec1_16, vec2_16, un0_16, un2_16;
};

struct adf_painn {
    adf_painn_hparams hp;
    int device;
    bool weights_set;
    const float* emb;
    const float* rbf_offset;
    adf_layer_weights layer[16];
    adf_block_weights head[2][2];
    float scale[16];
    // message-kernel image of rbf_proj: [layer][slice][R][192] and bias [layer][slice][192]
    float* rbf_pack;
    float* rbf_bias_pack;
    // f16 hi/lo image of (rbf_proj * scale): [layer][slice][hi|lo][192][R] halves; bias * scale; 1/scale per layer
    void* rbf_pack16;
    float* rbf_bias_pack16;
    float* rbf_scales;
    // fp16 hi/lo images of every GEMM weight (one arena) + their scales; gemm_f32 selects the exact path
    unsigned char* w16_arena;
    size_t w16_bytes;
    float* w16_scales;
    float* w16_bias_perm;  // [L][2][3H] row-permuted biases of x_proj.2 / xvec_proj.2
    unsigned int* w16_scratch;
    unsigned char* wfrag_arena;   // fragment images of every split weight (adf_w16::frag), same size as w16_arena
    int fused_mlp;                // 0 never (default), 1 always, 2 by size: adf_painn_set_fused_mlp / ADF_FUSED_MLP
    bool gemm_f32;
    bool msg_f32;
    // per-row power-of-two lifts of the f16x3 products' A operands (default on; ADF_LIFT=0 = the unlifted split of rounds 1-2)
    bool lift_on;
    adf_lift lift;       // [3 capN] magnitudes a launcher measures itself
    float *mag_a, *mag_b;  // [capN] magnitudes handed from a producer (LayerNorm, a product's epilogue) to the next product
    float* mag_v3;         // [3 capN] magnitudes of the vec rows entering the heads (measured once, used by both heads)
    bool mag_v3_valid;

    // ---- grow-only workspaces
    int64_t capN, capB, capE;
    int32_t* nbr_cnt;    // [N]
    int32_t* nbr_src;    // [N*K]
    int32_t* nbr_shift;  // [N*K]  index into the lexicographic shift table
    int32_t* deg;        // [N+1] in-degree per target atom (symmetrised graph)
    int32_t* nptr;       // [N+1] exclusive scan = CSR row pointer over targets
    int32_t* cursor;     // [N]   fill cursors
    int32_t* img_cnt;    // [B]   directed edges per image (empty-image check)
    int32_t* sys_slow;   // [B]   1 = the system does not fit the per-system LDS kernels of the CSR build (graph.hip)
    void* scan_tmp;      // hipcub scan workspace
    size_t scan_tmp_bytes;
    // static-atom cache of the top-K kernel (adf_graph_set_moving)
    const int32_t* moving;   // caller-owned [N] mask, null = every atom may move (no cache)
    const int32_t* mov_idx;  // caller-owned: indices of the moving atoms grouped by system
    const int32_t* mov_off;  // caller-owned [B+1]
    float* cache_d2;         // [capN*K]
    int32_t* cache_cid;      // [capN*K]
    int32_t* cache_cnt;      // [capN]
    bool cache_valid;
    int32_t* e_src;      // [capE] source atom of every edge, grouped by target, sorted by distance
    float4* e_geom;      // [capE] (ux,uy,uz,d): unit vector target->source, distance
    // device int32[8], STICKY (only adf_check_flags / adf_graph_build(num_edges) / adf_graph_set_moving clear them):
    // {0 candidate overflow, 1 empty image, 2 edge overflow, 3 in-degree beyond the sorter, 4 atomic number out of
    //  range, 5 non-finite or fp16-range-exceeding activation (gemm16.hip), 6-7 unused}
    int32_t* flags;
    float *x, *vecA, *vecB, *y, *xh, *vv, *cat, *dot;  // node buffers
    float* rec;          // [(N+1)][H/32][160] gather records of the message kernel (message.hip)
    bool rbf_uniform;    // Gaussian centres are k/(R-1): the message kernel may use its recurrence (ADF_MSG_RBF=direct: never)
    // Layer-0 gather records depend on the atomic numbers only (x0 = emb(Z), vec0 = 0).  While a static-atom
    // promise is in force (adf_graph_set_moving: same batch, only flagged atoms move) they are computed once.
    float* rec0;
    int64_t rec0_cap, rec0_N;
    bool rec0_valid;
    // ---- incremental layers (api.hip forward_incremental, incremental.hip): per-layer node state kept across the
    // forwards of one static-atom promise; a forward recomputes only rows whose inputs changed since they were computed
    bool inc_on;                       // adf_painn_set_incremental / ADF_INCREMENTAL (default on)
    bool inc_valid;                    // the kept state belongs to the current batch, weights and arithmetic
    int64_t inc_capN, inc_N;
    int inc_layers;
    float* incX[ADF_MAX_LAYERS + 1];   // x entering layer l (l = L: entering the heads)   [capN, H]
    float* incV[ADF_MAX_LAYERS + 1];   // vec likewise, l >= 1 (vec entering layer 0 is zero)  [capN, 3, H]
    float* incR[ADF_MAX_LAYERS];       // gather records of layer l  [(capN+1), H/32, 160]
    int32_t *prev_nptr, *prev_src;     // CSR of the previous build (swapped with nptr / e_src / e_geom per build)
    float4* prev_geom;
    unsigned char *inc_c0, *inc_chg;   // [capN] in-edges changed; [2][capN] layer input changed (ping-pong)
    unsigned char *inc_pend, *inc_need, *inc_tf;  // [L][capN] row has unapplied changes / is needed / is recomputed now
    int32_t* inc_list;                 // [L][capN] compacted recompute lists (ascending)
    int32_t* inc_cnt;                  // device [2L+1]: list lengths, in-edges of the listed rows, all edges
    // The list lengths never gate a launch: list-mode kernels are sized for all N rows and read their row count from
    // inc_cnt on the device (rows_dev below).  The host sees the counts one forward late, through a pinned double buffer
    // and an event it polls without waiting, and uses them only to choose between the list and the all-rows form of a
    // layer and for the statistics.  ADF_INC_SYNC=1: read them back synchronously instead (exact launch sizes).
    int32_t* inc_cnt_host;             // pinned [2][2L+1]
    void* inc_ev[2];                   // hipEvent_t after the copy into inc_cnt_host[slot]
    bool inc_ev_live[2];
    int inc_slot;
    bool inc_sync;
    int32_t inc_seen[2 * ADF_MAX_LAYERS + 1];  // the latest counts the host has seen
    bool inc_seen_valid;
    unsigned char inc_pend_whole[2][ADF_MAX_LAYERS];  // per slot: which layers of that forward ran in the all-rows form
    int32_t inc_pend_N[2];
    int32_t inc_pend_Nseen;            // the atom count inc_seen belongs to
    const int32_t* rows_dev;           // device row count of the launches being enqueued (a list-mode layer), else null
    void* inc_tmp; size_t inc_tmp_bytes;  // hipcub select workspace
    unsigned long long inc_rows, inc_rows_full, inc_edges, inc_launches;  // totals since adf_painn_set_incremental
    unsigned long long build_serial, inc_serial;  // graph builds made / the build the kept state belongs to
    float *sub_x, *sub_vec, *sub_f;  // compact rows of adf_painn_forward_subset: [capS,H], [capS,3,H], [capS,3]
    int64_t capS;
    float* sys;          // [B*16] per-system scratch of the stepper
    // last graph
    int64_t lastN, lastB;
    int32_t last_reps[3];
    int num_cus;
    // ---- optional HIP-event profiling of the forward (bench.py roofline); see api.hip
    bool prof_on;
    std::vector<hipEvent_t>* prof_ev;     // pool of events, used pairwise
    std::vector<int>* prof_cat;           // category of every recorded pair
    size_t prof_used;                     // events handed out
    unsigned long long* kcount;           // device counter: executed k-steps of the message kernel
};

enum { ADF_PROF_GRAPH = 0, ADF_PROF_MESSAGE = 1, ADF_PROF_NODE = 2, ADF_PROF_HEADS = 3, ADF_PROF_STEPPER = 4 };
void adf_prof_begin(adf_painn* h, int cat, hipStream_t s);
void adf_prof_end(adf_painn* h, hipStream_t s);

// ---- kernels' host launchers (each enqueues on `s`, returns ADF_*)
int32_t adf_launch_gemm(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                        int M, int N, int K, int act_ssilu, hipStream_t s);
// lf: measure the A rows' magnitudes into lf->buf first (row lifts, see adf_epi::rmag); premag: they are already known
// (written by the producer of A: LayerNorm, a previous product's out_mag); out_mag: emit the output rows' magnitudes
int32_t adf_launch_gemm16(const float* A, int lda, const adf_w16* W, const float* bias, float* C, int ldc, int M,
                          int N, int K, int act_ssilu, hipStream_t s, const float* A2 = nullptr, int K1 = 0,
                          const adf_lift* lf = nullptr, const float* premag = nullptr, float* out_mag = nullptr,
                          const int32_t* m_dev = nullptr, int accumulate = 0,   // m_dev: see adf_epi::m_dev
                          const float* gate = nullptr, int gate_ld = 0);        // gate: see adf_epi::gate
// m_dev / m_mul: rows = min(M, *m_dev * m_mul) when the count lives on the device (m_mul = 3: [N,3,K] vector rows)
int32_t adf_launch_rowmag(const float* A, int lda, int K1, const float* A2, int K2, long long M, float* mag, hipStream_t s,
                          const int32_t* m_dev = nullptr, int m_mul = 1);
int32_t adf_split_weight(const float* w, long long n, adf_w16* out, unsigned int* scratch_bits, hipStream_t s,
                         int perm_H = 0, int K = 0, const float* bias = nullptr, int parts = 3);
int32_t adf_launch_gemm16_vecnorm(const float* A, int lda, const adf_w16* W, float* nrm, int M, int N, int K,
                                  hipStream_t s, const adf_lift* lf = nullptr, const float* premag = nullptr);
int32_t adf_launch_gemm16_fused(const float* A, int lda, const adf_w16* W, int M, int H, int K, int epi,
                                const adf_epi* ep, hipStream_t s, const adf_lift* lf = nullptr);
// C = act(A . W^T + b): f16x3 split MFMA by default, exact-f32 MFMA when h->gemm_f32 (ADF_GEMM=f32)
static inline int32_t adf_linear(const adf_painn* h, const float* A, int lda, const float* W, const adf_w16* W16,
                                 const float* bias, float* C, int ldc, int M, int N, int K, int act, hipStream_t s,
                                 const float* premag = nullptr, float* out_mag = nullptr) {
    if (h->gemm_f32) return adf_launch_gemm(A, lda, W, K, bias, C, ldc, M, N, K, act, s);
    return adf_launch_gemm16(A, lda, W16, bias, C, ldc, M, N, K, act, s, nullptr, 0, h->lift_on ? &h->lift : nullptr,
                             h->lift_on ? premag : nullptr, h->lift_on ? out_mag : nullptr, h->rows_dev);
}
// mlp16.hip: fragment image of a split weight [N, K]; the fused two-layer product (see the kernel comment)
int32_t adf_pack_frag(const adf_w16* w, int N, int K, void* out, hipStream_t s);
int32_t adf_launch_mlp16(const float* A1, const float* A2, int lda, const float* rmag, const void* W0f, const adf_w16* W0,
                         const float* bias0, const void* W2f, const adf_w16* W2, int M, int H, int epi, const adf_epi* ep,
                         hipStream_t s);
int32_t adf_graph_build_impl(adf_painn* h, const adf_batch* b, hipStream_t s);
// incremental.hip
size_t adf_inc_temp_bytes(int64_t n);
int32_t adf_inc_compare(adf_painn* h, int N, hipStream_t s);  // inc_c0 from (nptr, e_src, e_geom) vs prev_*
int32_t adf_inc_need_from_list(adf_painn* h, int N, int L, const int32_t* out_idx, int n_out, hipStream_t s);
int32_t adf_inc_plan_layer(adf_painn* h, int l, int N, bool first, bool have_need, hipStream_t s);
// n_dev: rows = min(n, *n_dev) (the list length stays on the device)
int32_t adf_inc_scatter_rows(const float* src, const int32_t* idx, int n, int width, float* dst, hipStream_t s,
                             const int32_t* n_dev = nullptr);
int32_t adf_inc_gather_rows(const float* src, const int32_t* idx, int n, int width, float* dst, hipStream_t s);
size_t adf_scan_temp_bytes(int64_t n);
int32_t adf_message_impl(adf_painn* h, int layer, int N, const float* x, const float* xh, const float* vec,
                         float* x_out, float* vec_out, bool vec_is_zero, hipStream_t s,
                         const int32_t* tlist = nullptr, int n_targets = 0, const float* rec = nullptr,
                         const int32_t* n_targets_dev = nullptr);  // n_targets_dev: the list length lives on the device
int32_t adf_pack_rbf(adf_painn* h, hipStream_t s);
int32_t adf_pack_rbf_layer(adf_painn* h, int l, hipStream_t s);
int32_t adf_pack_records(adf_painn* h, int N, const float* xh, const float* vec, bool vec_is_zero, hipStream_t s,
                         float* rec = nullptr);
int32_t adf_nodewise_embed(adf_painn* h, const int32_t* Z, int N, float* x, hipStream_t s);
int32_t adf_nodewise_layernorm(const float* x, const float* w, const float* b, float* y, int N, int H, hipStream_t s,
                               float* out_mag = nullptr, const int32_t* n_dev = nullptr);  // out_mag: max|y| per row
int32_t adf_nodewise_update_prep(const float* vv, const float* x, float* cat, float* dot, int N, int H, hipStream_t s);
int32_t adf_nodewise_update_apply(const float* h3, const float* dot, const float* vv, float* x, float* vec,
                                  float scale, int N, int H, hipStream_t s);
int32_t adf_head_forward(adf_painn* h, int head, int N, const float* x, const float* vec, float* out, hipStream_t s);
int32_t adf_stepper_init(const adf_batch* b, float* pos, const int32_t* tags, const float* noise,
                         hipStream_t s);
// sys: [16 B] floats of per-system scratch
int32_t adf_stepper_step(float* sys, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                         const float* f1, const float* f2, const adf_step_coef* coef, const adf_step_coef* coefs_dev,
                         int num_steps, const float* z_tr, const float* z_rot, int32_t early_stop_count,
                         int32_t* state, float* dcom, float* drot, hipStream_t s);
