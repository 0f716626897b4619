/*
 * adsorbdiff_hip.h — C ABI of libadsorbdiff_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the denoising-diffusion sampling hot path of AdsorbDiff.
 * The reference has no FFI of its own (it is pure Python on top of PyTorch /
 * torch_scatter); each entry point below replaces the Python-level interface
 * named in its comment (paths relative to the reference tree).  The Python
 * binding that a maintainer adds is a ctypes stub — see INTEGRATION.md and
 * adsorbdiff_amd/lib.py.
 *
 * Conventions
 *   - every function returns an int32 status: ADF_OK, or an ADF_E* code; the
 *     message of the last failure on the calling thread is adf_last_error().
 *     ADF_EOOM is what the host shim turns into RuntimeError so that
 *     ml_diffuse's split-the-batch-and-retry contract keeps working
 *     (relaxation/ml_relaxation.py:146-165); ADF_ENONEIGHBOR becomes the
 *     ValueError of painn_denoising.py:370-375.
 *   - all array arguments are caller-owned DEVICE pointers, row-major, float32
 *     or int32 as named; `stream` is a hipStream_t passed as void*.
 *   - a handle is bound to the device that was current at adf_painn_create and
 *     is NOT thread-safe.  Work is enqueued on `stream`; nothing synchronises
 *     unless stated.
 */
#ifndef ADSORBDIFF_HIP_H
#define ADSORBDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADF_OK 0
#define ADF_EINVAL 1      /* invalid argument / unsupported hyper-parameter        */
#define ADF_EOOM 2        /* device allocation failed                             */
#define ADF_ENONEIGHBOR 3 /* an image has no neighbours (painn_denoising.py:370)  */
#define ADF_EHIP 4        /* HIP runtime error                                    */
#define ADF_EOVERFLOW 5   /* candidate list of one centre exceeded its capacity   */
#define ADF_ENUMERIC 6    /* non-finite model output (see adf_painn_set_arithmetic) */

typedef struct adf_painn* adf_painn_t;

/* Hyper-parameters of the PaiNN denoiser: the constructor arguments of
 * adsorbdiff.models.painn.painn_denoising.PaiNN (painn_denoising.py:57-81)
 * that reach the arithmetic. */
typedef struct {
    int32_t hidden_channels;   /* H, multiple of 64                              */
    int32_t num_layers;
    int32_t num_rbf;           /* R, even, <= 128                                */
    int32_t num_elements;      /* rows of the atom embedding table               */
    int32_t max_neighbors;     /* K of the strict top-K neighbour cap, <= 128    */
    int32_t envelope_exponent; /* p of the polynomial envelope                   */
    int32_t num_heads;         /* 2 when so3_denoising else 1                    */
    float cutoff;              /* Angstrom                                       */
} adf_painn_hparams;

/* Order of the weight table handed to adf_painn_set_weights (all float32,
 * torch.nn.Linear layout [out, in]; names are the reference state_dict keys):
 *   0  atom_emb.embeddings.weight              [num_elements, H]
 *   1  radial_basis.rbf.offset                 [R]
 *   then per layer l (13 entries, base 2 + 13*l):
 *    +0  message_layers.l.x_layernorm.weight   [H]
 *    +1  message_layers.l.x_layernorm.bias     [H]
 *    +2  message_layers.l.x_proj.0.weight      [H, H]
 *    +3  message_layers.l.x_proj.0.bias        [H]
 *    +4  message_layers.l.x_proj.2.weight      [3H, H]
 *    +5  message_layers.l.x_proj.2.bias        [3H]
 *    +6  message_layers.l.rbf_proj.weight      [3H, R]
 *    +7  message_layers.l.rbf_proj.bias        [3H]
 *    +8  update_layers.l.vec_proj.weight       [2H, H]
 *    +9  update_layers.l.xvec_proj.0.weight    [H, 2H]
 *    +10 update_layers.l.xvec_proj.0.bias      [H]
 *    +11 update_layers.l.xvec_proj.2.weight    [3H, H]
 *    +12 update_layers.l.xvec_proj.2.bias      [3H]
 *   then per head h in {out_forces, out_forces2}, per block b in {0,1} (6 entries):
 *    +0 output_network.b.vec1_proj.weight  +1 vec2_proj.weight
 *    +2 update_net.0.weight  +3 update_net.0.bias  +4 update_net.2.weight  +5 update_net.2.bias
 */
#define ADF_WEIGHTS_PER_LAYER 13
#define ADF_WEIGHTS_PER_HEAD 12

/* Replaces constructing the model (painn_denoising.py:57-148). */
int32_t adf_painn_create(const adf_painn_hparams* hp, adf_painn_t* out);
int32_t adf_painn_destroy(adf_painn_t h);

/* (Re)bind the weights.  Pointers are device pointers that stay owned by the
 * caller and must outlive their use; rbf_proj is re-packed into the library's
 * own layout.  scale_factors: HOST array [num_layers], the effective
 * upd_out_scalar_scale_l multipliers (1.0 when unfitted; scale_factor.py:166). */
int32_t adf_painn_set_weights(adf_painn_t h, int32_t n_weights, const void* const* weights,
                              const float* scale_factors, void* stream);

/* Describes one batch of independent adsorbate+slab systems (the fields of the
 * PyG Batch the reference's forward reads).  All device. */
typedef struct {
    int32_t num_systems;          /* B                                            */
    int32_t num_atoms;            /* N                                            */
    const float* pos;             /* [N,3]                                        */
    const float* cell;            /* [B,3,3]  rows = lattice vectors              */
    const int32_t* atomic_numbers;/* [N]                                          */
    const int32_t* batch;         /* [N] system index, non-decreasing             */
    const int32_t* atom_offset;   /* [B+1] prefix sum of natoms                   */
    int32_t reps[3];              /* periodic images per lattice direction (host) */
} adf_batch;

/* Replaces BaseModel.generate_graph + PaiNN.generate_graph_values
 * (models/base.py:33-123, painn_denoising.py:353-400): periodic radius graph,
 * strict top-K, symmetrisation.  The graph stays in the handle.  *num_edges
 * (HOST, may be NULL) forces a stream sync when non-NULL. */
int32_t adf_graph_build(adf_painn_t h, const adf_batch* b, void* stream, int64_t* num_edges);

/* Static-atom cache for repeated graph builds of the SAME batch in which only some atoms move
 * (sampling: the adsorbate; denoising_torch.py:353 only rewrites pos[tags == 2]).  moving: [N] int32
 * mask (1 = may move), mov_idx: indices of the moving atoms grouped by system, mov_off: [B+1] offsets
 * into mov_idx; all caller-owned device arrays that must stay valid until reset.  The next build
 * evaluates everything and caches each static centre's K nearest static candidates; later builds
 * only re-evaluate candidates that involve a moving atom — results are identical to a full build.
 * While the promise is in force adf_painn_forward also keeps the layer-0 gather records across calls: they depend
 * on the atomic numbers only (x0 = emb(Z), vec0 = 0), so the batch's atomic numbers and the weights must not change
 * either (adf_painn_set_weights invalidates).  Pass NULLs to switch the cache off (default).  Any call invalidates. */
int32_t adf_graph_set_moving(adf_painn_t h, const int32_t* moving, const int32_t* mov_idx, const int32_t* mov_off);

/* Arithmetic of the dense products: 0 = f16x3 (three fp16 matrix-core products per fp32 product, fp32 accumulate:
 * 2^-22 relative per product for activations inside the fp16 range, |a| <= 65504; default), 1 = exact f32 MFMA (the
 * reference's width, models/painn/README.md:12; ~2x slower).  A forward whose output is not finite reports
 * ADF_ENUMERIC through adf_check_flags; the host mirror then re-runs the forward / the sampling run in exact f32. */
int32_t adf_painn_set_arithmetic(adf_painn_t h, int32_t exact_f32);

/* Incremental layers (default on; ADF_INCREMENTAL=0 in the environment at creation = off).  While a static-atom promise
 * is in force (adf_graph_set_moving: same batch, only flagged atoms move) the handle keeps x / vec / gather records of
 * every layer and a forward recomputes a node row only if one of its inputs — its own row in the layer below, a
 * neighbour's, or its in-edge list / geometry, compared bit for bit with the previous build — changed since the row
 * was computed.  Every output is bit-identical to a full forward; the reference recomputes everything every step
 * (denoising_torch.py:498 -> painn_denoising.py:460-471).  Costs about 110 KB of HBM per atom (H=512, 6 layers).
 * The lengths of the recompute lists stay on the device: list-mode launches are sized for all N rows and every kernel
 * of a layer reads its row count from device memory; the host sees the counts one forward late (pinned double buffer
 * behind an event it polls without waiting) and uses them only to choose between the list and the all-rows form of a
 * layer - an incremental forward synchronises nothing (ADF_INC_SYNC=1 restores a synchronous 52-byte read-back).
 * Calling this (with either value) drops the kept state and zeroes the counters adf_get_counters reports. */
int32_t adf_painn_set_incremental(adf_painn_t h, int32_t on);

/* Form of the two node MLP pairs of a layer (x_proj: painn_denoising.py:531; xvec_proj: :614-623) at hidden width 512 in the
 * f16x3 arithmetic: mode 0 (default) = two kernels per pair (product + ScaledSiLU, then product + fused epilogue;
 * csrc/gemm16.hip), 1 = one kernel per pair with the [rows, 512] intermediate kept in LDS and the weights streamed as MFMA
 * fragments (csrc/mlp16.hip; round 6: bit-identical results, measured 5-10 % slower per pair on MI355X - its epilogues'
 * HBM traffic does not overlap its matrix phases at one workgroup per CU, profiles/NOTES.md), 2 = mode 1 from 2 x 64 rows per
 * CU on.  ADF_FUSED_MLP sets the initial mode. */
int32_t adf_painn_set_fused_mlp(adf_painn_t h, int32_t mode);

/* Read the device-side error flags of the last graph build (candidate overflow,
 * empty image).  Synchronises the stream.  adf_painn_forward does not check
 * them itself so that a sampling loop stays free of host round trips. */
int32_t adf_check_flags(adf_painn_t h, void* stream);

/* Copy the handle's current graph out (parity tests).  Directed top-K stage in
 * the reference's candidate order: nbr_count[N], nbr_src[N*K], nbr_shift[N*K*3].
 * Symmetrised stage (grouped by target, order within a group unspecified):
 * edge_src/edge_dst [cap], edge_dist [cap], edge_vec [cap*3]; returns E in
 * *num_edges.  Any pointer may be NULL.  Synchronises the stream. */
int32_t adf_graph_export(adf_painn_t h, int32_t* nbr_count, int32_t* nbr_src, int32_t* nbr_shift,
                         int64_t edge_capacity, int32_t* edge_src, int32_t* edge_dst, float* edge_dist,
                         float* edge_vec, int64_t* num_edges, void* stream);

/* Replaces PaiNN.forward(data) (painn_denoising.py:402-481): graph build + 6
 * message/update layers + the two gated-equivariant heads.  f1,f2: [N,3]
 * (f2 may be NULL when num_heads == 1). */
int32_t adf_painn_forward(adf_painn_t h, const adf_batch* b, float* f1, float* f2, void* stream);

/* The same forward when the caller reads the outputs of a subset of atoms only — the sampler: the per-system score
 * is the mean over the adsorbate atoms (denoising_torch.py:263-268, 460-467), the slab rows of f1 / f2 are never
 * read.  out_idx: n_out ascending atom indices (device int32).  All layers but the last run in full; the last
 * layer's message targets, its update and the heads are evaluated for the listed atoms only.  Rows out_idx[*] of
 * f1 / f2 are bit-identical to adf_painn_forward's; the other rows are NOT written. */
int32_t adf_painn_forward_subset(adf_painn_t h, const adf_batch* b, const int32_t* out_idx, int32_t n_out, float* f1,
                                 float* f2, void* stream);

/* Unit-testable pieces of the forward on the handle's current graph
 * (PaiNNMessage.forward, painn_denoising.py:530-567, fused with the residual
 * of :443-445):  x_out = (x + dx)/sqrt2 [N,H],  vec_out = vec + dvec [N,3,H]. */
int32_t adf_painn_message_layer(adf_painn_t h, int32_t layer, int32_t num_atoms, const float* x,
                                const float* vec, float* x_out, float* vec_out, void* stream);
/* PaiNNUpdate.forward + residual + ScaleFactor (painn_denoising.py:447-451,601-623), in place. */
int32_t adf_painn_update_layer(adf_painn_t h, int32_t layer, int32_t num_atoms, float* x, float* vec,
                               void* stream);

/* Stand-alone torch.nn.functional.linear (+ optional ScaledSiLU) on device pointers, A[M,K]
 * W[N,K] bias[N] C[M,N] contiguous; K % 32 == 0.  mode 0: exact-f32 MFMA (gemm.hip); mode 1:
 * f16x3 split MFMA (gemm16.hip).  Unit-test entry for the dense blocks of painn_denoising.py:531,
 * 603, 609, 689-693; synchronises in mode 1. */
int32_t adf_linear_forward(const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N,
                           int32_t K, int32_t act_ssilu, int32_t mode, void* stream);

/* Per-step schedule scalars, computed by the host with the reference's own
 * 0-dim tensor arithmetic (denoising_torch.py:237-293) so that the products
 * round exactly as there:
 *   dcom = coef_tr * s_tr                      (+ noise_tr  * z_tr   in SDE mode)
 *   drot = ((rot_pre * s_rot) * rot_dt) * rot_g2 (+ noise_rot * z_rot in SDE mode)
 * ODE: coef_tr = 0.5*g_tr^2*dt, rot_pre = 0.5.  SDE: coef_tr = g_tr^2*dt, rot_pre = 1,
 * noise_tr = g_tr*sqrt(dt), noise_rot = g_rot*sqrt(dt). */
typedef struct {
    float coef_tr;
    float rot_pre;
    float rot_dt;
    float rot_g2;
    float noise_tr;
    float noise_rot;
} adf_step_coef;

/* Replaces the initial random placement (denoising_torch.py:215-232).
 * noise: [B,3] uniform [0,1) drawn by the host from the CPU generator. */
int32_t adf_sde_init_placement(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags,
                               const float* noise, void* stream);

/* Replaces one iteration of Denoiser.reverse_sde_sampling_rot after the model
 * call (denoising_torch.py:263-353 incl. DiffTorchCalc :491-500): zero f2 on
 * fixed atoms, per-system adsorbate means, ODE/SDE update, COM wrap, rigid
 * rotation+translation of the adsorbate, cumulative early-stop counter.
 * z_tr,z_rot: [B,3] standard normals (SDE only, else NULL).  state: device
 * int32[8] = {cumulative converged-step count, frozen flag, all-converged flag
 * of the step in flight, steps applied, steps issued, -, -, -}; the caller
 * initialises it to {0,0,1,0,0,0,0,0}.  Once the count reaches `early_stop_count` (>0) that step and all
 * later ones leave pos untouched, which is the reference's `break`
 * (denoising_torch.py:312-320).  dcom,drot: optional [B,3] outputs. */
int32_t adf_sde_step(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags,
                     const int32_t* fixed, const float* f1, const float* f2, const adf_step_coef* coef,
                     const float* z_tr, const float* z_rot, int32_t early_stop_count, int32_t* state,
                     float* dcom, float* drot, void* stream);

/* Same step with the whole schedule in a DEVICE table coefs_dev[num_steps], indexed by state[4]
 * (steps issued).  Every step is then the identical launch sequence, so forward + step can be
 * captured once into a hipGraph and replayed (small batches are launch-bound). */
int32_t adf_sde_step_scheduled(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags,
                               const int32_t* fixed, const float* f1, const float* f2,
                               const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr,
                               const float* z_rot, int32_t early_stop_count, int32_t* state, float* dcom,
                               float* drot, void* stream);

/* Counters of the last adf_painn_forward: algorithmic bytes of the message
 * kernel, dense FLOPs, padded/real edge rows (bench.py roofline). */
/* The whole reverse loop of Denoiser.reverse_sde_sampling_rot after the initial placement (denoising_torch.py:
 * 235-356) in one call: num_steps x (adf_painn_forward[_subset] + adf_sde_step_scheduled), enqueued on `stream`.
 * z_tr_all, z_rot_all: [num_steps][B][3] standard normals (SDE) or both NULL (ODE).  poll_every > 0 (and
 * early_stop_count > 0): every poll_every steps the frozen flag state[1] is read back (one stream synchronisation)
 * and the loop ends once it is set — the reference's `break`; 0 = never synchronise for that (steps after the stop
 * are no-ops on pos; incremental layers never synchronise: their list lengths stay on the device, ADF_INC_SYNC=1 restores the
 * per-forward read-back).  out_idx / n_out: optional subset of atoms whose model outputs are evaluated (see
 * adf_painn_forward_subset), NULL = all.  f1, f2: [N,3] work arrays (last step's outputs on return). */
int32_t adf_sample(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                   const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr_all, const float* z_rot_all,
                   int32_t early_stop_count, int32_t poll_every, int32_t* state, const int32_t* out_idx,
                   int32_t n_out, float* f1, float* f2, void* stream);

typedef struct {
    int64_t num_edges;
    int64_t num_atoms;
    int64_t message_bytes_per_layer;  /* SURVEY.md §8d formula on the real E, N */
    int64_t dense_flops;              /* node + edge GEMM flops of one forward  */
    /* incremental layers, totals since adf_painn_set_incremental: node rows recomputed / rows a full forward would
     * have computed (layers x atoms), message-kernel launches made and the in-edges of the targets they evaluated */
    int64_t inc_rows, inc_rows_full, inc_msg_launches, inc_msg_edges;
} adf_counters;
int32_t adf_get_counters(adf_painn_t h, adf_counters* out, void* stream);

/* Optional HIP-event timing of the kernel groups of adf_painn_forward / adf_sde_step, recorded
 * on the launch stream.  Categories: 0 graph build, 1 message kernel (one launch per layer),
 * 2 node-side dense blocks (LayerNorm + GEMMs + update), 3 output heads, 4 stepper.
 * adf_profile_read synchronises, returns summed milliseconds and number of timed groups per
 * category (arrays of 5) and resets the log.  *message_ksteps (optional) = sum over all 32-edge
 * row blocks of all message launches of (k-window length actually contracted x 32-column blocks that ran: 6, or 4 in
 * the vec == 0 launches of the first layer); executed MFMA flops of the message kernel = message_ksteps * 32 * 32 * 2
 * (x 3 products in the f16x3 arithmetic). */
#define ADF_PROF_NCAT 5
int32_t adf_profile_enable(adf_painn_t h, int32_t on);
int32_t adf_profile_read(adf_painn_t h, float* ms, int64_t* count, int64_t* message_ksteps, void* stream);

/* On-box peaks for the roofline fractions: out_host3[0] = HBM stream copy GB/s (read + write bytes of a
 * 1 GiB -> 1 GiB copy), [1] = v_mfma_f32_32x32x16_f16 TFLOP/s, [2] = v_mfma_f32_32x32x2_f32 TFLOP/s, both from a
 * register-resident loop on every CU with non-zero operands.  Measurement aid of bench.py, not on the sampling path. */
int32_t adf_measure_peaks(float* out_host3, void* stream);

/* ---- per-step trajectory frames.  Replaces Denoiser.write's blocking per-step host copy
 * (relaxation/diffusers/denoising_torch.py:358-367, 469-477; relaxation/ase_utils.py:19-48): a frame = the [N,3] positions
 * after a reverse step.  adf_frames_push snapshots `src` into one of two device staging buffers on `stream` and copies it
 * into slot (index % slots) of a pinned host ring on the sink's own stream; frames are numbered in push order.  A host
 * writer thread takes them with adf_frames_wait (blocks up to timeout_ms; *host_ptr = NULL on time-out) and gives the slot
 * back with adf_frames_release; push blocks the enqueueing thread only while the slot it needs is still unreleased.
 * adf_sample_traj / adf_eqv2_sample_traj = adf_sample / adf_eqv2_sample that push a frame after every `frame_every`-th
 * applied step (and after the last one) without leaving the fused loop.  adf_frames_abort cancels a sink: a push that is
 * waiting for a slot (or any later push) returns ADF_EINVAL and every wait returns at once - the host writer calls it when
 * it stops early (I/O error), so that the sampler reports the failure instead of blocking on a full ring.  Not thread-safe
 * per sink except wait / release / pushed / abort against push. */
typedef struct adf_frames* adf_frames_t;
int32_t adf_frames_create(int32_t device, int64_t frame_floats, int32_t slots, adf_frames_t* out);
int32_t adf_frames_destroy(adf_frames_t f);
int32_t adf_frames_push(adf_frames_t f, const float* src, void* stream);
int32_t adf_frames_wait(adf_frames_t f, int64_t index, int32_t timeout_ms, const float** host_ptr);
int32_t adf_frames_release(adf_frames_t f, int64_t index);
int64_t adf_frames_pushed(adf_frames_t f);
int32_t adf_frames_abort(adf_frames_t f);
int32_t adf_sample_traj(adf_painn_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                        const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr_all, const float* z_rot_all,
                        int32_t early_stop_count, int32_t poll_every, int32_t* state, const int32_t* out_idx,
                        int32_t n_out, float* f1, float* f2, adf_frames_t sink, int32_t frame_every, void* stream);

/* Hand-off to the relaxation stage: the lift rule of scripts/create_lmdbs/pred_traj_to_lmdb.py:81-90 applied to the
 * sampled final frames on the device (in place).  lifted: optional [B] output = shift applied per system. */
int32_t adf_lift_adsorbates(float* pos, const int32_t* tags, const int32_t* atom_offset, int32_t B, float min_gap,
                            float* lifted, void* stream);

/* Multi-GPU exchange of the sharded sampler: systems are independent, every rank samples its shard
 * with no data-path collective, and ONE all-gather of the sampled adsorbate sites ends a pass.  Replaces the reference's
 * per-rank .npz + barrier + rank-0 merge (trainers/sde_denoising_trainer.py:862-909).  RCCL is loaded lazily (dlopen).
 *   adf_comm_unique_id   rank 0 draws a 128-byte id (ncclGetUniqueId) and hands it to the other ranks out of band
 *                        (adsorbdiff_amd/sampler.py broadcasts it through torch.distributed);
 *   adf_comm_create      collective over all ranks: one communicator per rank on the current device;
 *   adf_allgather_sites  out[r*bytes_per_rank ..] = rank r's `local`; device pointers, enqueued on `stream`;
 *                        every rank passes the same (padded) bytes_per_rank. */
#define ADF_COMM_ID_BYTES 128
typedef struct adf_comm* adf_comm_t;
int32_t adf_comm_unique_id(uint8_t* out128);
int32_t adf_comm_create(const uint8_t* id128, int32_t rank, int32_t world, adf_comm_t* out);
int32_t adf_comm_destroy(adf_comm_t comm);
int32_t adf_allgather_sites(adf_comm_t comm, const void* local, int64_t bytes_per_rank, void* out, void* stream);

/* ---- Training step (score matching; SURVEY.md 8f-1, BASELINE config 5).  Device ops that adsorbdiff_amd/train_step.py
 * strings together into forward-with-saved-activations, loss and backward of the PaiNN denoiser; they replace
 * torch.autograd through models/painn/painn_denoising.py + DenoisingTrainer._compute_loss
 * (trainers/sde_denoising_trainer.py:675-728) and torch.optim.AdamW + clip_grad_norm_ + the EMA update
 * (trainers/base_trainer.py:787-820).  Exact f32.  All pointers are device pointers; ld* are row strides in floats;
 * "acc" flags select accumulate-into instead of overwrite.  The graph ops use the handle's current graph. */
int32_t adf_op_linear_fwd(const float* A, int32_t lda, const float* W, const float* bias, float* C, int32_t ldc, int64_t M,
                          int32_t N, int32_t K, void* stream);
int64_t adf_op_linear_bwd_scratch(int64_t M, int32_t N, int32_t K); /* floats of scratch adf_op_linear_bwd needs */
int32_t adf_op_linear_bwd(const float* A, int32_t lda, const float* W, const float* dC, int32_t ldc, float* dA, int32_t ldda,
                          int32_t acc_dA, float* dW, float* db, int32_t acc_dW, int64_t M, int32_t N, int32_t K,
                          float* scratch, void* stream);
int32_t adf_op_ssilu_fwd(const float* h, float* y, int64_t n, void* stream);
int32_t adf_op_ssilu_bwd(const float* h, const float* dy, float* dh, int64_t n, void* stream);
int32_t adf_op_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* stats, int32_t N, int32_t H,
                             void* stream);
/* dx is accumulated into; dw, db [H] are written; scratch: 512 * 2 * H floats. */
int32_t adf_op_layernorm_bwd(const float* x, const float* w, const float* stats, const float* dy, float* dx, float* dw,
                             float* db, int32_t N, int32_t H, float* scratch, void* stream);
int32_t adf_op_embed_fwd(adf_painn_t h, const int32_t* Z, int32_t N, float* x, void* stream);
int32_t adf_op_embed_bwd(const float* dx, const int32_t* Z, float* demb, int32_t N, int32_t H, void* stream);
int32_t adf_op_rbf(adf_painn_t h, float* rbf, void* stream);
int32_t adf_op_message_fwd(adf_painn_t h, const float* xh, const float* vec, const float* rbfh, const float* x, float* x1,
                           float* vec1, int32_t vec_is_zero, void* stream);
/* The same forward through the sampler's fused message kernel (no [E,3H] operand: the radial-basis projection runs inside the
 * kernel on the matrix cores, painn_denoising.py:530-567); the layer's rbf_proj images are rebuilt from the bound weight
 * tensors first (the optimizer updates them in place).  Outputs agree with adf_op_message_fwd to ~1e-6 relative. */
int32_t adf_op_message_fwd_fused(adf_painn_t h, int32_t layer, const float* xh, const float* vec, const float* x, float* x1,
                                 float* vec1, int32_t vec_is_zero, void* stream);
int32_t adf_op_message_bwd(adf_painn_t h, const float* xh, const float* vec, const float* rbfh, const float* gx1,
                           const float* gv1, float* dxh, float* drbfh, float* dvec, float* dx, int32_t vec_is_zero,
                           void* stream);
/* The same backward with rbfh REGENERATED inside the kernel on the matrix cores (message_bwd.hip; autograd through
 * painn_denoising.py:530-567): no [E,3H] operand is kept from the forward or recomputed by a dense product.  drbfh is written
 * ([num_edges + 1, 3H]: one spare row that the kernel's padded edge rows write) with its 3H columns in the kernel's own
 * order; adf_op_message_bwd_perm fills perm[c'] = the column of rbf_proj's output
 * that kernel-order column c' holds (host array of 3H entries), so that the weight-gradient product of rbf_proj can run on it
 * and its rows be permuted back.  Needs the f16x3 arithmetic and equally spaced Gaussian centres
 * (adf_op_message_bwd_fused_supported returns 1) and the layer's rbf_proj images of this step (adf_op_message_fwd_fused
 * builds them).  Gradients agree with adf_op_message_bwd to ~1e-6 relative. */
int32_t adf_op_message_bwd_fused_supported(adf_painn_t h);
int32_t adf_op_message_bwd_perm(adf_painn_t h, int32_t* perm_host, int32_t n);
int32_t adf_op_message_bwd_fused(adf_painn_t h, int32_t layer, const float* xh, const float* vec, const float* gx1,
                                 const float* gv1, float* dxh, float* drbfh_lane_order, int64_t num_edges, float* dvec,
                                 float* dx, int32_t vec_is_zero, float* dbias_rows, void* stream);
/* The default since round 5: d(rbfh) never reaches memory.  adf_op_message_bwd_fused is called with drbfh_lane_order = NULL and
 * dbias_rows = [N, 3H] (it then writes, per atom, the column sums of the atom's d(rbfh) rows - the bias gradient of rbf_proj is
 * their sum over the atoms - and leaves the packed gradient records of the layer in the handle); adf_op_rbf_wgrad_fused then
 * ACCUMULATES dW [3H, R] of the layer's rbf_proj (reference row order), forming d(rbfh) again from those records, xh / vec
 * of the owning atom and the edge geometry while it stages the product (three-term bf16 split, six products, as
 * adf_op_linear_bwd's weight-gradient kernel).  edge_owner [num_edges] = the atom whose CSR segment holds an edge row
 * (adf_op_edge_owner, once per graph); rbf_image: adf_op_rbf_image; scratch: adf_op_rbf_wgrad_fused_scratch(h) floats.
 * Replaces autograd's dW = d(rbfh)^T edge_rbf for models/painn/painn_denoising.py:530-567 (PaiNNMessage.rbf_proj). */
int32_t adf_op_edge_owner(adf_painn_t h, int32_t* edge_owner, int64_t num_edges, void* stream);
/* the radial basis [num_edges, num_rbf] (adf_op_rbf) as the three bf16 terms of the product in the kernel's own layout, once per
 * step (it does not depend on the layer); `image`: adf_op_rbf_image_bytes(num_edges) bytes */
int64_t adf_op_rbf_image_bytes(int64_t num_edges);
int32_t adf_op_rbf_image(adf_painn_t h, const float* rbf, int64_t num_edges, void* image, void* stream);
int64_t adf_op_rbf_wgrad_fused_scratch(adf_painn_t h);
int32_t adf_op_rbf_wgrad_fused(adf_painn_t h, const float* xh, const float* vec, const void* rbf_image, const int32_t* edge_owner,
                               int64_t num_edges, float* dW, float* scratch, int32_t vec_is_zero, void* stream);
int32_t adf_op_vdot_fwd(const float* vv, float* dot, float* nrm, int32_t ldn, int64_t N, int32_t C, float eps, void* stream);
int32_t adf_op_vdot_bwd(const float* vv, const float* nrm, int32_t ldn, const float* ddot, const float* dnrm, int32_t lddn,
                        const float* dv1, float* dvv, int64_t N, int32_t C, void* stream);
int32_t adf_op_update_out_fwd(const float* x1, const float* vec1, const float* a, const float* dot, const float* vv, float s,
                              float* x2, float* vec2, int64_t N, int32_t H, void* stream);
int32_t adf_op_update_out_bwd(const float* a, const float* dot, const float* vv, float s, const float* dx2,
                              const float* dvec2, float* da, float* ddot, float* dv1, float* dx1, float* dvec1, int64_t N,
                              int32_t H, void* stream);
int32_t adf_op_vnorm_fwd(const float* t1, float* nrm, int32_t ldn, int64_t N, int32_t C, void* stream);
int32_t adf_op_vnorm_bwd(const float* t1, const float* nrm, int32_t ldn, const float* dnrm, int32_t lddn, float* dt1,
                         int64_t N, int32_t C, void* stream);
int32_t adf_op_gate_fwd(const float* o, const float* t2, float* xs, int32_t ldx, float* vout, int64_t N, int32_t C,
                        void* stream);
int32_t adf_op_gate_bwd(const float* o, const float* t2, const float* dxs, int32_t lddx, const float* dvout, float* d_o,
                        float* dt2, int64_t N, int32_t C, void* stream);
int32_t adf_op_copy_rows(const float* src, int32_t lds_, float* dst, int32_t ldd, int64_t M, int32_t C, int32_t accumulate,
                         void* stream);
int32_t adf_op_score_loss(const float* f1, const float* f2, const int32_t* tags, const int32_t* atom_offset,
                          const float* tr_sigma, const float* rot_sigma, const float* tr_score, const float* rot_score,
                          const float* rot_norm, float* loss, float* df1, float* df2, int32_t B, float* scratch, void* stream);
int32_t adf_op_sqnorm_accumulate(const float* g, int64_t n, float* out, void* stream);
int32_t adf_op_adamw_step(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* sqnorm,
                          float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                          float ema_decay, void* stream);

/* ---- EquiformerV2 denoiser (BASELINE config 4).  Replaces
 * EquiformerV2S_OC20_DenoisingPos.forward(data) (models/equiformer_v2/equiformer_v2_denoising.py:185-318) for the
 * configuration the repository ships (configs/denoising/eqv2_so3.yml): one resolution, layer_norm_sh, SiLU attention
 * with re-normalised alpha, separable S2 activation, grid MLP feed-forward, per-block atom edge embeddings, Gaussian
 * distance expansion with 600 functions (equiformer_v2_oc20.py:251-264), FOR_denoising = two force blocks. */
typedef struct adf_eqv2* adf_eqv2_t;
typedef struct {
    int32_t lmax, mmax;              /* lmax_list[0] (1..6), mmax_list[0] (<= lmax)                        */
    int32_t num_layers;
    int32_t sphere_channels;         /* C                                                               */
    int32_t attn_hidden_channels;
    int32_t num_heads, attn_alpha_channels, attn_value_channels;
    int32_t ffn_hidden_channels;
    int32_t grid_resolution;         /* res: the S2 grid has res x res points                           */
    int32_t edge_channels;
    int32_t num_distance_basis;      /* 600 in the reference, whatever the constructor argument says    */
    int32_t max_num_elements;
    int32_t max_neighbors;           /* K of the strict top-K cap                                       */
    float max_radius;                /* cutoff, Angstrom                                                */
    float avg_degree;                /* rescale of the edge-degree embedding (_AVG_DEGREE)              */
} adf_eqv2_hparams;

int32_t adf_eqv2_create(const adf_eqv2_hparams* hp, adf_eqv2_t* out);
int32_t adf_eqv2_destroy(adf_eqv2_t h);

/* Constant tables the reference takes from e3nn / Jd.pt (so3.py:509-531,566-613; wigner.py:8), HOST float32 arrays
 * computed by adsorbdiff_amd/so3_math.py:  jd = J_l row-major, l = 0..lmax concatenated;  to_red / from_red
 * [res*res, S_r] with the |m| <= mmax coefficients in m-major order;  to_full / from_full [res*res, (lmax+1)^2]. */
int32_t adf_eqv2_set_constants(adf_eqv2_t h, const float* jd, const float* to_red, const float* from_red,
                               const float* to_full, const float* from_full);

/* Bind the weights: caller-owned DEVICE float32 tensors in torch layout, reference state_dict names:
 *   0 atom_radii [101]  1 sphere_embedding.weight  2,3 edge_degree_embedding.{source,target}_embedding.weight
 *   4..13 edge_degree_embedding.rad_func  (RAD = net.0.weight, net.0.bias, net.1.weight, net.1.bias, net.3.weight,
 *         net.3.bias, net.4.weight, net.4.bias, net.6.weight, net.6.bias)
 *   then per block i:  norm_1 (NORM = affine_weight, norm_l0.weight, norm_l0.bias), ga (ATTN), norm_2 (NORM),
 *         ffn (so3_linear_1.weight, .bias, scalar_mlp.0.weight, .bias, grid_mlp.0.weight, grid_mlp.2.weight,
 *         grid_mlp.4.weight, so3_linear_2.weight, .bias)
 *   then norm (NORM), force_block (ATTN), force_block2 (ATTN)
 *   ATTN = alpha_dot, source_embedding.weight, target_embedding.weight, so2_conv_1.fc_m0.weight, .bias,
 *          so2_conv_1.so2_m_conv.{0..mmax-1}.fc.weight, so2_conv_1.rad_func (RAD), alpha_norm.weight, .bias,
 *          so2_conv_2.fc_m0.weight, .bias, so2_conv_2.so2_m_conv.{0..mmax-1}.fc.weight, proj.weight, proj.bias */
int32_t adf_eqv2_set_weights(adf_eqv2_t h, int32_t n_weights, const void* const* weights, void* stream);

/* 0 = f16x3 split products on the f16 matrix cores where the shapes allow (default), 1 = exact f32 everywhere. */
int32_t adf_eqv2_set_arithmetic(adf_eqv2_t h, int32_t exact_f32);

/* Use this edge list (source, target, vector target -> source image; target non-decreasing; DEVICE arrays, copied)
 * instead of building one, until adf_eqv2_set_edges(h, 0, ...) — parity tests against reference runs whose pick among
 * exactly tied K-th neighbours is implementation-defined. */
int32_t adf_eqv2_set_edges(adf_eqv2_t h, int64_t num_edges, const int32_t* src, const int32_t* dst, const float* vec,
                           int32_t max_in_degree, void* stream);
/* Static-atom cache of the neighbour search, as adf_graph_set_moving. */
int32_t adf_eqv2_set_moving(adf_eqv2_t h, const int32_t* moving, const int32_t* mov_idx, const int32_t* mov_off);

/* The forward: graph (models/base.py:33-123, NOT symmetrised), edge frames + Wigner rows, embeddings, num_layers
 * transformer blocks, final norm, two force blocks.  f1, f2: [N,3] = the l = 1 coefficients (m = -1, 0, 1) of the two
 * force blocks (equiformer_v2_denoising.py:307-318).  x_blocks (optional, may be NULL): [num_layers + 1][N][S][C]
 * node embeddings after the edge-degree embedding and after every block (parity tests). */
int32_t adf_eqv2_forward(adf_eqv2_t h, const adf_batch* b, float* f1, float* f2, float* x_blocks, void* stream);
/* The sampler's forward (cf. adf_painn_forward_subset): only the force blocks read the last node embedding, and the
 * update reads the adsorbate rows of f1 / f2 alone (denoising_torch.py:263-268, 460-467) - the force blocks run for the
 * listed target atoms only (n_out ascending indices, device int32), on a compacted copy of their incoming edges.  Rows
 * out_idx[*] of f1 / f2 are bit-identical to adf_eqv2_forward's; the other rows are NOT written. */
int32_t adf_eqv2_forward_subset(adf_eqv2_t h, const adf_batch* b, const int32_t* out_idx, int32_t n_out, float* f1,
                                float* f2, void* stream);
int32_t adf_eqv2_check_flags(adf_eqv2_t h, void* stream);

/* Reverse-diffusion stepper on an EquiformerV2 handle: same contracts as adf_sde_init_placement,
 * adf_sde_step_scheduled and adf_sample above. */
/* Incremental blocks (off until switched on; the host mirror switches it on for a sampling run, denoising_pos_params
 * ["incremental_layers"], default True).  While a static-atom promise is in force (adf_eqv2_set_moving: same batch, only
 * flagged atoms move) the handle keeps the embedding and every transformer block's output; a forward compares every
 * target's incoming edge list (sources and vectors, bit for bit) with the previous forward's and block i recomputes only
 * the targets within i + 1 hops of a changed one - on a compacted copy of their edges, as adf_eqv2_forward_subset does for
 * the force blocks.  Every output is bit-identical to a full forward; the reference recomputes every row at every step
 * (denoising_torch.py:498 -> equiformer_v2_denoising.py:232-318).  Costs (num_layers + 2) x S x C x 4 bytes per atom
 * (250 KB at config 4) and one 4 x num_layers-byte read-back (a stream synchronisation) per forward, which sizes the
 * launches.  Calling this (with either value) drops the kept state and zeroes the inc_* counters. */
int32_t adf_eqv2_set_incremental(adf_eqv2_t h, int32_t on);

int32_t adf_eqv2_init_placement(adf_eqv2_t h, const adf_batch* b, float* pos, const int32_t* tags, const float* noise,
                                void* stream);
int32_t adf_eqv2_sde_step(adf_eqv2_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                          const float* f1, const float* f2, const adf_step_coef* coef, const adf_step_coef* coefs_dev,
                          int32_t num_steps, const float* z_tr, const float* z_rot, int32_t early_stop_count,
                          int32_t* state, float* dcom, float* drot, void* stream);
int32_t adf_eqv2_sample(adf_eqv2_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                        const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr_all, const float* z_rot_all,
                        int32_t early_stop_count, int32_t poll_every, int32_t* state, const int32_t* out_idx, int32_t n_out,
                        float* f1, float* f2, void* stream);
int32_t adf_eqv2_sample_traj(adf_eqv2_t h, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                             const adf_step_coef* coefs_dev, int32_t num_steps, const float* z_tr_all, const float* z_rot_all,
                             int32_t early_stop_count, int32_t poll_every, int32_t* state, const int32_t* out_idx, int32_t n_out,
                             float* f1, float* f2, adf_frames_t sink, int32_t frame_every, void* stream);

/* Stand-alone torch.nn.functional.linear (+ optional SiLU, act = 2) through this path's dense-product kernels (unit tests
 * and micro-benchmarks of so2_ops.py:12-79,158-238 / so3.py:694-745 shapes).  A [M,K], W [N,K], bias [N] or NULL, C [M,N],
 * device pointers.  mode 0: exact f32; 1: f16x3 split with per-row power-of-two lifts (fp32 rows split in the kernel;
 * K % 32 == 0, N % 4 == 0); 2: the same product on rows pre-split into fp16 hi / lo images; 3: as 2 with the weights
 * streamed from their MFMA-fragment image (the sampler's first SO(2) convolution; N % 32 == 0, N >= 128, K % 64 == 0;
 * bit-identical to 2).  repeat > 1 re-runs the product kernel alone.  Synchronises. */
int32_t adf_eqv2_linear_forward(const float* A, const float* W, const float* bias, float* C, int64_t M, int32_t N, int32_t K,
                                int32_t act, int32_t mode, int32_t repeat, void* stream);

/* Work of the last forward and HIP-event time per kernel group (bench.py roofline).  Categories: 0 graph + Wigner,
 * 1 radial MLPs, 2 rotate in / out, 3 SO(2) convolution products, 4 S2 activation, 5 attention weights, 6 node-side
 * norms / SO(3) linears, 7 feed-forward grid MLP, 8 stepper. */
#define ADF_EQV2_PROF_NCAT 9
typedef struct {
    int64_t num_edges, num_atoms;
    int64_t dense_flops;       /* 2 x multiply-adds of every dense product of the REFERENCE's forward (f32-equivalent work) */
    int64_t conv_flops;        /* 2 x multiply-adds the SO(2)-convolution kernels of the last forward executed (force blocks:
                                * l = 1 columns only; subset forward / incremental blocks: the listed targets' edges only,
                                * estimated) */
    int64_t inc_rows, inc_rows_full;  /* incremental blocks, totals since adf_eqv2_set_incremental: block rows recomputed /
                                       * block rows full forwards would have computed (blocks x atoms) */
    int64_t forwards_total, conv_flops_total;  /* forwards since adf_eqv2_profile_enable(1) and conv_flops summed over them
                                                * (on the last forward's edge count) */
} adf_eqv2_counters;
int32_t adf_eqv2_get_counters(adf_eqv2_t h, adf_eqv2_counters* out, void* stream);
int32_t adf_eqv2_profile_enable(adf_eqv2_t h, int32_t on);
int32_t adf_eqv2_profile_read(adf_eqv2_t h, float* ms, int64_t* count, void* stream);

const char* adf_last_error(void);
const char* adf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ADSORBDIFF_HIP_H */
