/*
 * esr_hip.h -- C ABI of libesr_hip.so, the MI355X (gfx950) embedding-training hot path.
 *
 * The reference (BBischof/ESRecsys) has no plugin / operator / FFI boundary: its
 * hot path is a handful of Python functions executed by XLA through JAX/Flax/Optax
 * (SURVEY.md section 8b).  This header is therefore the boundary a maintainer
 * would bind from the reference's Python (ctypes -- see INTEGRATION.md); each
 * entry point cites the reference lines whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - returns ESR_OK (0) or a negative ESR_E* code; never throws.  esr_last_error()
 *     returns a thread-local message for the last failing call on this thread.
 *   - every pointer is DEVICE memory owned by the caller unless marked "host".
 *     The library never allocates, frees or synchronises; work is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *   - scratch memory is passed explicitly: `workspace` of at least the size the
 *     matching esr_*_workspace_bytes() query returns (16-byte aligned).
 *   - tables are row-major [V, D]; ids are int32 in [0, V).  Rows are processed
 *     in 16-byte chunks when D * sizeof(elt) is a multiple of 16, else scalar.
 *   - re-entrant across streams/threads; no global mutable state.
 */
#ifndef ESR_HIP_H_
#define ESR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESR_OK 0
#define ESR_EINVAL (-1)     /* bad argument (null pointer, negative size, unsupported D/k/...) */
#define ESR_ELAUNCH (-2)    /* HIP launch / runtime error (message has hipGetErrorString) */
#define ESR_EWORKSPACE (-3) /* workspace too small or misaligned */
#define ESR_ENODEVICE (-4)  /* no gfx950 device visible */

#define ESR_F32 0
#define ESR_BF16 1

#define ESR_GLOVE_REFERENCE 0 /* (B,B)-broadcast loss of wikipedia/train_cooccurence.py:83 */
#define ESR_GLOVE_DIAGONAL 1  /* textbook per-pair GloVe loss (build-defined option) */
/* OR-ed into esr_glove_fwd_bwd's `mode` / esr_triplet_fwd_bwd's `with_reg`: the gradient of the occurrence that read
 * table row r is written at output row r instead of at its occurrence index.  For "tables" that are a private copy
 * of the looked-up rows with one row per occurrence (the row-sharded step: rows in exchange order, ids = the inverse
 * routing permutation) this emits the gradients directly in exchange order; the three triplet gradient pointers may
 * then alias one buffer. */
#define ESR_GRADS_AT_IDS 0x100

typedef void* esr_stream_t;

const char* esr_last_error(void);
int esr_version(void);
/* host out-params; arch_len bytes at arch receive e.g. "gfx950". */
int esr_device_info(int* cu_count, int* wave_size, size_t* hbm_bytes, char* arch, int arch_len);

/* ---- measurement: per-kernel launch durations ("GPU kernel times ... from rocprofv3"; bench.py's
 * `roofline.achieved` wants the dominant kernel's duration measured live with HIP events on the stream the kernel is
 * launched on -- the kernels are launched inside the library, so the events are recorded here).  The reference has no
 * counterpart (it never measures: wikipedia/train_cooccurence.py:185-186 logs the loss only).
 * esr_kernel_timing(1): from now on the instrumented launch sites record a HIP event pair around their launch (the
 * in-batch head's kernels, the one-pass GloVe / triplet update kernels, the retrieval GEMM and selects); (0): stop.
 * Either call drops unread records.  Off by default: one load per launch site.
 * esr_kernel_timing_read: the ONE entry point that synchronises (it waits for the recorded events): writes
 * "name\tcalls\ttotal_ms\tmin_ms\tmax_ms\n" per kernel name to the HOST buffer `buf` (NUL-terminated, truncated to cap)
 * and clears the records; returns the bytes the whole text needs. */
int esr_kernel_timing(int enable);
long esr_kernel_timing_read(char* buf, size_t cap);
/* rocprofv3 markers ("tracing / profiling": absent in the reference -- only a commented-out
 * jax_debug_nans toggle, spotify/train_spotify.py:162-163): esr_trace_markers(1), or ESR_ROCTX=1 in the environment when the
 * library is loaded, makes every instrumented launch site and every step entry point a roctx range (push / pop on the
 * calling thread), so `rocprofv3 --marker-trace --kernel-trace` shows which phase of which step a kernel belongs to;
 * (0): off, the default (one load per site).  ESR_ENODEVICE when no roctx library can be loaded. */
int esr_trace_markers(int enable);

/* ---- G2 / S1: embedding-row gather ------------------------------------------------------
 * nn.Embed lookup == jnp.take(table, ids, axis=0): wikipedia/models.py:31-34 (and the id towers
 * that replace pinterest/models.py:64-70).  out[i, :] = table[ids[i], :], bit-exact. */
int esr_gather_rows(const void* table, int dtype, int64_t V, int D, const int32_t* ids, int64_t n,
                    void* out, esr_stream_t stream);

/* Debug screen for DEVICE-resident ids (the hot path trusts them: a host-side check would force a sync per step).
 * report is device int64 [2], initialised by the caller to {0, INT64_MAX}: report[0] += number of ids outside
 * [0, V), report[1] = min(position of such an id).  jnp.take clamps out-of-range ids silently
 * (wikipedia/models.py:31-34 [upstream jax]); here they would be wild reads / read-modify-writes, so the Python
 * layer runs this screen on every device id tensor when ESR_CHECK_IDS=1 and raises IndexError. */
int esr_check_ids(const int32_t* ids, int64_t n, int64_t V, int64_t* report, esr_stream_t stream);

/* ---- G2: Glove.__call__ pieces -- wikipedia/models.py:30-37 ------------------------------
 * dot[j] = E[t1[j]] . E[t2[j]],  s[i] = Bias[t1[i]] + Bias[t2[i]];  the (B,B) output is
 * dot[None,:] + s[:,None].  inputs is int32 [2, B] row-major (cooccurrence_matrix.py:103-104). */
int esr_glove_forward(const float* emb, const float* bias, int64_t V, int D, const int32_t* inputs,
                      int64_t B, float* dot, float* s, esr_stream_t stream);

/* ---- G3: apply_model / glove_loss value_and_grad -- wikipedia/train_cooccurence.py:76-89 ---
 * Fused gather + dot + weighted log10 loss + per-occurrence row gradients.
 *   loss[0]                       scalar loss
 *   grad_rows [2B, D]             occurrence j   : dL/ddot_j * E[t2[j]]   (row id t1[j])
 *                                 occurrence B+j : dL/ddot_j * E[t1[j]]   (row id t2[j])
 *   grad_bias [2B]                occurrence i and B+i: dL/ds_i
 * The occurrence ids are simply inputs[0..2B) read as a flat array. */
size_t esr_glove_workspace_bytes(int64_t B);
int esr_glove_fwd_bwd(const float* emb, const float* bias, int64_t V, int D, const int32_t* inputs,
                      const float* target, int64_t B, int mode, float* loss, float* grad_rows,
                      float* grad_bias, void* workspace, size_t workspace_bytes, esr_stream_t stream);

/* ---- G3 + G4 in one pass: loss, gradients and the sparse Adagrad update of BOTH tables ------------------------
 * (wikipedia/train_cooccurence.py:71-101 with the build's row-sparse Adagrad in place of dense optax.adam.)
 * No gradient row is ever written to memory: the update kernel walks the sorted occurrences of every distinct row,
 * re-reads each occurrence's partner row, forms gdot and accumulates gdot * partner on chip, then does the row's one
 * read-modify-write.  Reading partner rows while other rows are being rewritten is made safe by DOUBLE-BUFFERING
 * the embedding table: `emb` and `emb_shadow` are two [V, D] buffers and emb_loc [V] holds one STAMPED byte per row:
 * bit 0 = the buffer with the row's current value, bits 1..7 = the stamp of the step that last moved the row (0 = long
 * ago).  The step reads every row where it lived when the step began -- a byte that already carries THIS step's stamp
 * says "moved during this step, the old value is in the other buffer" -- writes each updated row into the other
 * buffer and stamps its byte.  `stamp`: 1 .. 127, a value no byte of the table carries from an EARLIER step: count the
 * steps on a table 1, 2, ... 127 and call esr_rows_restamp (one pass over V bytes, clears the stamps) before starting
 * over at 1.  Callers that need a plain [V, D] table run esr_rows_consolidate (copies the rows that live in `shadow`
 * into `primary`, zeroes every byte).  A fresh state is emb_loc = 0.
 * Same sort, same cut points and the same association of every row sum as esr_glove_fwd_bwd +
 * esr_sparse_adagrad_scatter: tables and accumulators agree with that path to an f32 rounding (bias sums are carried
 * in fp64 here; the bias statistics sum s, sum s^2 of K_A in exact integer arithmetic).
 * loss [1]; bias / bias_accum [V] are updated in place (every bias read of a step precedes its bias writes).
 * presorted_ids / presorted_perm (both or neither): the output of esr_segment_sort_ids on inputs[0 .. 2B) -- the sort
 * depends on the ids only, so a training loop runs it ahead; NULL = sort here.  plan (optional, needs the presorted
 * arrays): esr_glove_plan's record of this batch, made ahead as well -- it holds what the step needs from ids and
 * counts alone and the zeroed accumulators of ONE step (a plan feeds exactly one esr_glove_train_step); NULL = made
 * here.  long_runs: 0 = the caller knows from esr_glove_plan's hint that no run of equal ids outgrows a 32-position
 * chunk, and the long-run launch is skipped; anything else = launched (it returns at once when nothing was parked).
 * Launches per step with a plan and uniform ids: update, finalize.  blocks_per_cu > 0 caps the update kernel's
 * residency (experiment knob); 0 = fill the chip.  start_flag (optional, device uint32): the update kernel's first
 * workgroup stores start_value there as it starts -- a second stream gated on the word (esr_stream_gate: the id sort
 * of a coming batch) is released while that kernel runs, i.e. arrives after it has taken its wave slots, and the main
 * queue carries no event marker for it.
 * dtype (round 6): the embedding rows' type -- ESR_F32, or ESR_BF16 (BASELINE config 4's dtype: both buffers bf16 [V, D],
 * fp32 accumulator, fp32 bias tables): rows are widened on the load (exact), stepped in f32 and rounded to nearest even on
 * their one store; ~7 200 instead of 10 292 bytes per pair at D = 256. */
size_t esr_glove_step_workspace_bytes(int64_t B, int D);
int esr_glove_train_step(void* emb, void* emb_shadow, uint8_t* emb_loc, float* emb_accum, float* bias,
                         float* bias_accum, int64_t V, int dtype, int D, const int32_t* inputs, const float* target, int64_t B,
                         int mode, float lr, float eps, uint32_t stamp, const int32_t* presorted_ids,
                         const int32_t* presorted_perm, void* plan, int long_runs, int blocks_per_cu,
                         uint32_t* start_flag, uint32_t start_value, float* loss, void* workspace,
                         size_t workspace_bytes, esr_stream_t stream);
/* Holds `stream` (one sleeping wave) until the device word *flag has reached `value` -- sequence numbers, compared
 * wrap-safe as (int32)(*flag - value) >= 0 -- or timeout_us have passed (then the stream goes on: the gate orders work
 * for speed and for producers that are certain to run; it never hangs a queue).  The cheap cross-stream edge of the
 * training loops: no event marker on the producing stream, whose kernel announces itself with one store
 * (esr_glove_train_step's start_flag).  Build-defined: the reference's loops are single-stream JAX dispatch
 * (wikipedia/train_cooccurence.py:103-112). */
int esr_stream_gate(const uint32_t* flag, uint32_t value, uint32_t timeout_us, esr_stream_t stream);
/* The steps of nbatch <= 8 planned batches issued by ONE call (a training loop's per-step host work is then one
 * foreign call per group: at the reference's batch of 2048 pairs a step is ~20 us of kernels, less than a ctypes call
 * with 25 arguments plus the Python around it).  inputs / targets / sorted_ids / perm / plans as esr_glove_plan took and
 * made them; stamps first_stamp, first_stamp + 1, ... (all <= 127); long_runs: host int32 [nbatch] (0 / 1 / -1 as above)
 * or NULL = unknown; losses [nbatch]. */
int esr_glove_train_steps(void* emb, void* emb_shadow, uint8_t* emb_loc, float* emb_accum, float* bias,
                          float* bias_accum, int64_t V, int dtype, int D, int nbatch, const int32_t* const* inputs,
                          const float* const* targets, int64_t B, int mode, float lr, float eps, uint32_t first_stamp,
                          const int32_t* sorted_ids, const int32_t* perm, void* plans, const int32_t* long_runs,
                          float* losses, void* workspace, size_t workspace_bytes, esr_stream_t stream);
/* Plans of nbatch <= 8 coming batches of B pairs in one launch: inputs[b] = int32 [2, B], targets[b] = f32 [B],
 * sorted_ids / perm = [nbatch, 2B] (esr_segment_sort_ids_batched layout, or one esr_segment_sort_ids result with
 * nbatch = 1), plans = nbatch x esr_glove_plan_bytes(B) bytes, 256-byte aligned.  hints (optional) [nbatch]:
 * hints[b] = gen when a run of equal ids in list b is longer than 32 positions -- compare with `gen` (any value the
 * caller has not passed for this array before) after the stream has passed the call; words need no clearing. */
size_t esr_glove_plan_bytes(int64_t B);
int esr_glove_plan(const int32_t* const* inputs, const float* const* targets, int nbatch, int64_t B,
                   const int32_t* sorted_ids, const int32_t* perm, void* plans, int32_t* hints, int32_t gen,
                   esr_stream_t stream);
/* The hint alone, for lists whose steps resolve their records themselves (more than 32 768 ids: esr_glove_train_step
 * then ignores `plan`): hint[0] = gen when sorted_ids [n] has a run of equal ids longer than `chunk` positions (32 for
 * the GloVe step, 8 for the triplet step).  `hint` may be device memory or pinned host memory. */
int esr_long_run_hint(const int32_t* sorted_ids, int64_t n, int chunk, int32_t* hint, int32_t gen, esr_stream_t stream);
int esr_rows_consolidate(void* primary, const void* shadow, uint8_t* loc, int64_t V, int dtype, int D,
                         esr_stream_t stream);
/* loc[r] &= 1 for every row: forget the stamps, keep the locations (see esr_glove_train_step's `stamp`). */
int esr_rows_restamp(uint8_t* loc, int64_t V, esr_stream_t stream);

/* ---- S1-S3: STL score head + triplet loss -- pinterest/models.py:67-72,
 * pinterest/train_shop_the_look.py:93-122 --------------------------------------------------
 * Towers are id-embedding tables; pos and neg rows may come from different tables (pass the product
 * table twice for the id towers).  ids == NULL means "row b of the table", so the three (B,D)
 * embedding matrices of the reference head can be passed directly as scene/pos/neg tables.
 *   loss = (sum_b relu(1 + neg_b - pos_b) + regularization * sum_b reg) / batch_size   (train)
 *   eval_step (:118) = the same call with with_reg = 0, batch_size = 1 and g_* = NULL.
 * pos_score / neg_score [B] may be NULL.  g_scene/g_pos/g_neg [B, D] per-occurrence gradients. */
size_t esr_triplet_workspace_bytes(int64_t B);
int esr_triplet_fwd_bwd(const float* scene_table, int64_t Vs, const float* pos_table, int64_t Vp,
                        const float* neg_table, int64_t Vn, int D, const int32_t* scene_ids, const int32_t* pos_ids,
                        const int32_t* neg_ids, int64_t B, float regularization, float batch_size,
                        int with_reg, float* loss, float* pos_score, float* neg_score,
                        float* g_scene, float* g_pos, float* g_neg, void* workspace,
                        size_t workspace_bytes, esr_stream_t stream);

/* ---- S2 in one pass: train_step (pinterest/train_shop_the_look.py:93-109) = loss + gradients + sparse Adagrad ----
 * on both id towers, no [3B, D] gradient in memory: the update kernel walks the sorted occurrences of every distinct row
 * and forms each occurrence's gradient row on chip from the two OTHER rows of its triplet (both partner rows give the
 * pos / neg scores and the hinge mask; the own row gives the regulariser term).  Both towers are DOUBLE-BUFFERED with
 * stamped location bytes as in esr_glove_train_step (`scene` / `scene_shadow` + scene_loc [Vs], `product` /
 * `product_shadow` + product_loc [Vp]; one `stamp` per step for both; esr_rows_restamp / esr_rows_consolidate per
 * table).  Occurrence ids are the virtual rows [scene_ids ; Vs + pos_ids ; Vs + neg_ids]; presorted_ids /
 * presorted_perm (both or neither) = their esr_segment_sort_ids_multi output computed ahead, NULL = sort here; plan /
 * long_runs as for esr_glove_train_step (esr_triplet_plan; chunks of 8 positions here).  With a plan and uniform ids a
 * step is ONE launch: the loss leaves the update kernel through an exact integer reduction (a mean loss beyond 2048
 * comes out +inf).  Same element arithmetic (trip_grad, adagrad_elem), same sort and the same association of every row
 * sum as esr_triplet_fwd_bwd + esr_sparse_adagrad_scatter_multi.
 * loss [1] = (sum_b relu(1 + neg_b - pos_b) + regularization * reg) / batch_size.
 *
 * DIRECT mode (round 4, the default; ESR_TRIPLET_STEP=stamped keeps the walk described above): one row group per TRIPLET
 * reads its three rows once and steps every row that occurs once in the batch IN PLACE (row + accumulator read and
 * written once: the fused minimum of 4 row transfers per row, against 6 per occurrence for the stamped walk); a row with
 * 2 .. 8 occurrences is stepped by whichever of its triplets finishes last (gradient rows parked at their sorted
 * positions, one atomic arrival per occurrence); longer runs by a second launch, made only when the plan's hint says
 * one exists.  Nothing is double-buffered: the *_shadow / *_loc arguments and `stamp` are ignored and may be NULL / 0,
 * rows never leave `scene` / `product`.  Same sums, same association as the stamped walk and the six-launch path.
 * dtype (round 6): ESR_F32, or ESR_BF16 -- bf16 table rows [V, D] with fp32 accumulators (BASELINE config 4's dtype;
 * direct mode only): a row is widened on the load (exact), stepped in f32 and rounded to nearest even on its one store,
 * as esr_sparse_adagrad_scatter does; 5 376 instead of 7 680 bytes per triplet at D = 128. */
size_t esr_triplet_step_workspace_bytes(int64_t B, int D);
int esr_triplet_train_step(void* scene, void* scene_shadow, uint8_t* scene_loc, float* scene_accum, int64_t Vs,
                           void* product, void* product_shadow, uint8_t* product_loc, float* product_accum,
                           int64_t Vp, int dtype, int D, const int32_t* scene_ids, const int32_t* pos_ids,
                           const int32_t* neg_ids, int64_t B, float regularization, float batch_size, float lr,
                           float eps, uint32_t stamp, const int32_t* presorted_ids, const int32_t* presorted_perm,
                           void* plan, int long_runs, float* loss, void* workspace, size_t workspace_bytes,
                           esr_stream_t stream);
/* The steps of nbatch <= 8 planned batches by one call (see esr_glove_train_steps): ids[3 b + {0, 1, 2}] as for
 * esr_triplet_plan, losses [nbatch]. */
int esr_triplet_train_steps(void* scene, void* scene_shadow, uint8_t* scene_loc, float* scene_accum, int64_t Vs,
                            void* product, void* product_shadow, uint8_t* product_loc, float* product_accum,
                            int64_t Vp, int dtype, int D, int nbatch, const int32_t* const* ids, int64_t B, float regularization,
                            float batch_size, float lr, float eps, uint32_t first_stamp, const int32_t* sorted_ids,
                            const int32_t* perm, void* plans, const int32_t* long_runs, float* losses, void* workspace,
                            size_t workspace_bytes, esr_stream_t stream);
/* Plans of nbatch <= 8 coming batches of B triplets in one launch: ids[3 b + {0, 1, 2}] = scene / pos / neg id lists
 * of batch b, sorted_ids / perm = [nbatch, 3B] (esr_segment_sort_ids_batched layout), plans = nbatch x
 * esr_triplet_plan_bytes(B) bytes, 256-byte aligned; hints / gen as for esr_glove_plan (runs longer than 8 positions).
 * Direct mode: gen != 0 also tags the plan's long-run counter with the full 32-bit value (a count left in the buffer by
 * an earlier plan call is replaced, no fill launch): pass a generation that differs from the one the buffer was last
 * planned with (a counter that grows by one per call does), and ZERO a plan buffer before its first use.  gen == 0: the
 * counter is cleared by a fill in front of the launch.  A direct-mode plan is left as planned by the step it feeds
 * (the arrival that completes a run clears the run's counter), so it may feed another step of the SAME batch; a
 * stamped-mode plan (ESR_TRIPLET_STEP=stamped) feeds exactly one step. */
size_t esr_triplet_plan_bytes(int64_t B);
int esr_triplet_plan(const int32_t* const* ids, int nbatch, int64_t B, int64_t Vs, const int32_t* sorted_ids,
                     const int32_t* perm, void* plans, int32_t* hints, int32_t gen, esr_stream_t stream);

/* ---- north_star: in-batch-negative sampled softmax on the dense B x B score matrix --------
 * (build-defined; the closest reference precedent is spotify/models.py:74-87).
 *   S = scale * Q C^T (FP32 MFMA), ce_i = logsumexp_j S_ij - S_ii,
 *   loss = (sum_i ce_i + regularization * sum_i [reg(q_i) + reg(c_i)]) / batch_size
 *   gQ = scale * (softmax(S) - I) C / batch_size + dreg ; gC likewise with S^T.
 * Q, C, gQ, gC are [B, D] row-major, any B >= 1 (tiles of 32 rows; a ragged last tile is masked out of every
 * softmax); D is 32, 64 or 128. */
size_t esr_inbatch_workspace_bytes(int64_t B, int D);
int esr_inbatch_softmax_fwd_bwd(const float* Q, const float* C, int64_t B, int D, float scale,
                                float regularization, float batch_size, float* loss, float* lse,
                                float* gQ, float* gC, void* workspace, size_t workspace_bytes,
                                esr_stream_t stream);

/* Same contract with FP32-EQUIVALENT products on the bf16 matrix cores: every operand is split exactly
 * into three bf16 planes and a product is the six leading cross terms (dropped terms <= 2^-23 |a||b|),
 * 2.67x fewer matrix-pipe cycles than v_mfma_f32_32x32x2_f32.  D <= 128 and a multiple of 4 (narrower rows are
 * zero-padded into the 128-column tiles), B a multiple of 128. */
size_t esr_inbatch3_workspace_bytes(int64_t B, int D);
int esr_inbatch_softmax_fwd_bwd_bf16x3(const float* Q, const float* C, int64_t B, int D, float scale,
                                       float regularization, float batch_size, float* loss, float* lse,
                                       float* gQ, float* gC, void* workspace, size_t workspace_bytes,
                                       esr_stream_t stream);
/* The whole in-batch step head without materialised Q / C: row i of Q is query_table[query_ids[i]], row i of C is
 * cand_table[cand_ids[i]] (tables f32 or bf16, dtype = ESR_F32 / ESR_BF16; the id-embedding towers that replace
 * pinterest/models.py:64-70).  The gather is folded into the bf16-plane split and into the merge kernels; gQ / gC
 * are the per-occurrence gradient rows [B, 128].  gq_rows / gc_rows (optional, may be NULL): gradient row i is
 * written at gQ[gq_rows[i]] / gC[gc_rows[i]] instead of row i -- the row-sharded step passes the bucket positions
 * of the occurrences, with gQ == gC == the buffer that goes straight into the gradient all-to-all.
 * Same workspace as above. */
int esr_inbatch_towers_fwd_bwd_bf16x3(const void* query_table, int64_t Vq, const void* cand_table, int64_t Vc,
                                      int dtype, int D, const int32_t* query_ids, const int32_t* cand_ids,
                                      const int32_t* gq_rows, const int32_t* gc_rows, int64_t B, float scale,
                                      float regularization, float batch_size, float* loss, float* lse,
                                      float* gQ, float* gC, void* workspace, size_t workspace_bytes,
                                      esr_stream_t stream);

/* Same two contracts with the products formed from TWO fp16 planes per operand (x * 2^e = x1 + x2 to 2^-24 relative with
 * round-to-nearest planes; a.b ~= a2 b1 + a1 b2 + a1 b1, dropped term <= 2^-24 |a||b|): three v_mfma_f32_32x32x16_f16 per
 * product instead of six bf16 ones, i.e. half the matrix-core work of the bf16x3 entry points at the same f32-grade
 * error (the kernels run at the chip's power limit, so fewer MFMAs is what shortens the step).  fp16's narrow range is
 * handled inside: a per-matrix power-of-two scale from the batch's largest |element|, and an exponent reference tied
 * to the true row maximum.  Pass C reads the B x B probabilities pass Q stored (as the two fp16 planes pass Q forms for
 * its own product; their per-row factors ride on per-split scaled copies of Q -- esr_inbatch2h_pass_c_forms): D <= 128 and a multiple of 4 (narrower
 * rows are zero-padded into the 128-column tiles; gQ / gC are [B, D]), B a multiple of 128 and <= 16384 (workspace ~ 4 B^2 bytes); esr_inbatch2h_workspace_bytes returns 256 for a B it cannot serve.
 * GUARANTEED ERROR (tests/test_gpu_kernels.py): an operand element x of a matrix whose largest |element| is M enters the
 * products as x (1 + d) + a with |d| <= 2^-24 and |a| <= 2^-26 M 2^-16 -- relative for elements within 2^-16 of M,
 * absolute below.  Consequences that are asserted against the fp64 oracle: loss, lse and norm-wise gradients within 1e-5
 * (every shape / range test); every gradient entry within 1e-4 of max(|entry|, 1e-3 of the largest entry) at C2 size;
 * with row norms spread over FOUR decades inside each matrix (a few hot rows 100 x the median) every 128-row block of
 * gQ / gC within 2e-5 of the block's own largest entry and every row within 1e-4 of its own (measured 1.6e-6 / 1.7e-6,
 * the exact-f32 MFMA path measures 1.5e-6 / 3.5e-6); a 20-step training trajectory at C2 size within 1e-5 on every loss
 * and norm-wise on tables and accumulators (measured 1e-6).  A matrix whose rows span MORE than ~2^16 in magnitude should
 * take the bf16x3 entry points, whose three planes are an exact split of every element. */
size_t esr_inbatch2h_workspace_bytes(int64_t B, int D);
int esr_inbatch_softmax_fwd_bwd_f16x2(const float* Q, const float* C, int64_t B, int D, float scale,
                                      float regularization, float batch_size, float* loss, float* lse,
                                      float* gQ, float* gC, void* workspace, size_t workspace_bytes,
                                      esr_stream_t stream);
int esr_inbatch_towers_fwd_bwd_f16x2(const void* query_table, int64_t Vq, const void* cand_table, int64_t Vc,
                                     int dtype, int D, const int32_t* query_ids, const int32_t* cand_ids,
                                     const int32_t* gq_rows, const int32_t* gc_rows, int64_t B, float scale,
                                     float regularization, float batch_size, float* loss, float* lse,
                                     float* gQ, float* gC, void* workspace, size_t workspace_bytes,
                                     esr_stream_t stream);
/* Which form pass C of the LAST f16x2 call on `workspace` took, per pass-Q split (forms[8], device memory, copied on
 * `stream`): 0 = the scaled copy of Q for that split fed the matrix cores as stored (round 6: dC_j = sum_i P'_ij (f_is q_i),
 * the stored fp16 planes of P' are the MFMA operand, no VALU work on probabilities), non-zero = some row of the copy lay
 * more than ~17 binades under the copy's largest element and the workgroups of that split applied the factors to the
 * probabilities in registers instead (the general form; also taken for every split when B / 32 / splits is not a
 * multiple of 8, where the words stay 0).  Diagnostics and tests; no reference counterpart. */
int esr_inbatch2h_pass_c_forms(const void* workspace, size_t workspace_bytes, int64_t B, int32_t* forms,
                               esr_stream_t stream);

/* The whole in-batch training step of the two towers as ONE call (fp16 x 2 score path + the build's row-sparse Adagrad;
 * what train_step(state, scene, pos, None, ...) of esrecsys_amd/pinterest/train_shop_the_look.py issues per batch):
 * gather + split -> pass Q -> [merge<Q> -> Adagrad on the query tower] beside [factors -> pass C -> merge<C> -> Adagrad on
 * the candidate tower].  side_stream (optional): a second stream of the same device; the first bracket then runs on it
 * while the second runs on `stream` (pass C needs only the factors of merge<Q>: 14 + 6 us of the step leave its critical
 * path at B = 8192), joined before the call returns control of `stream`; NULL: everything in order on `stream`.  Both
 * forms run the same kernels on the same values -- tables, accumulators, loss and lse are bit-identical to
 * esr_inbatch_towers_fwd_bwd_f16x2 + esr_sparse_adagrad_scatter_multi on the occurrence list [query ids ; Vq + cand ids].
 * presorted_vids / presorted_perm [2B] (both or neither): that list sorted by esr_segment_sort_ids_multi /
 * _batched ahead of the call (offsets {0, Vq}); NULL: sorted here.  long_runs: 0 = the caller knows (esr_long_run_hint) that
 * no id occurs more than 32 times, else -1.  lse [B] optional.  Tables f32 or bf16 [V, D] with fp32 accumulators, D <= 128
 * (a multiple of 4), B a multiple of 128, <= 16384.  Workspace: esr_inbatch_train_step_workspace_bytes, 256-byte aligned. */
size_t esr_inbatch_train_step_workspace_bytes(int64_t B, int D);
int esr_inbatch_train_step_f16x2(void* query_table, float* query_accum, int64_t Vq, void* cand_table, float* cand_accum,
                                 int64_t Vc, int dtype, int D, const int32_t* query_ids, const int32_t* cand_ids,
                                 int64_t B, float scale, float regularization, float batch_size, float lr, float eps,
                                 const int32_t* presorted_vids, const int32_t* presorted_perm, int long_runs, float* loss,
                                 float* lse, void* workspace, size_t workspace_bytes, esr_stream_t stream,
                                 esr_stream_t side_stream);

/* ---- G4 (build's production optimizer): sort + segment-reduce + sparse Adagrad -----------
 * Replaces the dense V x D gradient + dense optimizer sweep of
 * wikipedia/train_cooccurence.py:86-101 with a row-sparse update.
 * esr_segment_sort_ids: stable sort of occurrence ids; perm[k] = original occurrence index.  Every size runs this
 * library's own kernels (one-workgroup bitonic sort, tile sort + rank merge, 11-bit LSD radix passes; no device-library
 * sort): n up to 2^30 ids.
 * The scatter entry points below may OVERWRITE grad_rows: runs of equal ids that cross a 32-position boundary are
 * summed chunk-wise, the partial sums parked in the gradient rows themselves, and combined in a fixed order. */
size_t esr_segment_sort_workspace_bytes(int64_t n);
int esr_segment_sort_ids(const int32_t* ids, int64_t n, int64_t V, int32_t* sorted_ids,
                         int32_t* perm, void* workspace, size_t workspace_bytes,
                         esr_stream_t stream);
/* The same sort over a list given as up to four segments [ids_k + offsets[k]] (the towers of one step as virtual rows
 * offsets[k] + id of their concatenation): read in place, no concatenated copy.  Same workspace query with
 * n = the sum of the counts. */
int esr_segment_sort_ids_multi(const int32_t* const* ids, const int64_t* counts, const int64_t* offsets, int nseg,
                               int64_t V, int32_t* sorted_ids, int32_t* perm, void* workspace,
                               size_t workspace_bytes, esr_stream_t stream);
/* The lists of `nbatch` (<= 8) coming batches sorted in ONE launch sequence: a training loop knows the ids of the next
 * batches before it needs them (the reference's loop draws them from its data iterator,
 * pinterest/train_shop_the_look.py:195-204), and at the reference's batch sizes the two-launch sort is latency, not
 * work -- eight lists cost what one does.  ids[b * nseg + k] = segment k of list b; counts / offsets are the same for
 * every list (n = their sum); sorted_ids / perm are [nbatch][n], list b exactly what esr_segment_sort_ids_multi would
 * give for it (perm indexes list b's own occurrences).  Up to 2 048 ids: one launch for all lists; up to 32 768: two;
 * up to 2 097 152 (and V < 2^33: at most three 11-bit digits): three per radix pass; beyond that the lists are sorted
 * one after the other. */
size_t esr_segment_sort_batched_workspace_bytes(int64_t n, int nbatch);
int esr_segment_sort_ids_batched(const int32_t* const* ids, const int64_t* counts, const int64_t* offsets, int nseg,
                                 int nbatch, int64_t V, int32_t* sorted_ids, int32_t* perm, void* workspace,
                                 size_t workspace_bytes, esr_stream_t stream);
/* For each distinct id: G = sum of its grad rows (occurrence order); acc += G*G;
 * p -= lr * G * rsqrt(acc + eps).  table dtype f32 or bf16 (fp32 accumulator either way). */
int esr_sparse_adagrad_scatter(void* table, int dtype, float* accum, int64_t V, int D,
                               const int32_t* sorted_ids, const int32_t* perm, int64_t n,
                               float* grad_rows, float lr, float eps, esr_stream_t stream);
/* Several tables (same D, same dtype) updated from ONE sorted occurrence list: ids are virtual rows
 * vid = row_offsets[t] + id of the concatenation of <= 4 tables, so a two-tower step needs one sort chain
 * and one update launch.  tables / accums / row_offsets (ntables + 1 entries) are HOST arrays of device
 * pointers / int64.  esr_concat_offset_ids builds the virtual ids: out = [ids[0] + offsets[0] ; ids[1] + ...]
 * (ids / counts / offsets are host arrays; the id buffers are device memory).
 * esr_sparse_adagrad_scatter_multi's long_runs: 0 = the caller knows (esr_long_run_hint with chunk = 32 on the sorted
 * list) that no id fills a whole 32-position block and the next position, and the launch that combines the chunk
 * partials of such runs is skipped; anything else = launched (on a list without them it only screens the boundaries). */
int esr_concat_offset_ids(const int32_t* const* ids, const int64_t* counts, const int64_t* offsets, int nseg,
                          int32_t* out, esr_stream_t stream);
/* out[i, :] = tables[t][vids[i] - row_offsets[t], :] (row bytes must be a multiple of 16). */
int esr_gather_rows_multi(const void* const* tables, const int64_t* row_offsets, int ntables, int dtype, int D,
                          const int32_t* vids, int64_t n, void* out, esr_stream_t stream);
int esr_sparse_adagrad_scatter_multi(void* const* tables, float* const* accums, const int64_t* row_offsets,
                                     int ntables, int dtype, int D, const int32_t* sorted_vids,
                                     const int32_t* perm, int64_t n, float* grad_rows, float lr, float eps,
                                     int long_runs, esr_stream_t stream);
/* Row-sparse SGD (p -= lr * G), same segment reduction. */
int esr_sparse_sgd_scatter(void* table, int dtype, int64_t V, int D, const int32_t* sorted_ids,
                           const int32_t* perm, int64_t n, float* grad_rows, float lr,
                           esr_stream_t stream);
/* Reference-faithful dense gradient: dense[V, D] = 0 then dense[id] = segment sum
 * (the scatter-add JAX's autodiff performs for nn.Embed, train_cooccurence.py:86-87). */
int esr_rows_to_dense(float* dense, int64_t V, int D, const int32_t* sorted_ids,
                      const int32_t* perm, int64_t n, float* grad_rows, esr_stream_t stream);
/* out[id, :] = the left-to-right sum of the gradient rows of the occurrences with sorted id `id`, for every id that
 * occurs -- esr_rows_to_dense without the zero fill, for callers whose ids cover [0, rows_out) (the distinct-row index
 * of esr_unique_by_owner: the per-occurrence gradient rows of a batch become ONE row per distinct row before they cross
 * the exchange).  May overwrite grad_rows (chunk partials of long runs are parked there). */
int esr_segment_sum_rows(float* out, int64_t rows_out, int D, const int32_t* sorted_ids, const int32_t* perm, int64_t n,
                         float* grad_rows, esr_stream_t stream);
/* optax.adam over every element -- wikipedia/train_cooccurence.py:99-101,171.
 * step = the 1-based count AFTER this update. */
int esr_dense_adam(float* param, float* mu, float* nu, const float* grad, int64_t numel, float lr,
                   float b1, float b2, float eps, int64_t step, esr_stream_t stream);

/* ---- G6: Glove.score_all + find_knn -- wikipedia/models.py:50-55, train_cooccurence.py:91-97
 * scores[v, t] = E[v] . E[token[t]]  ([V, T] row-major, no bias);
 * indices = stable ascending argsort of every column ([V, T] int32). */
int esr_score_all(const float* emb, int64_t V, int D, const int32_t* token, int T, float* scores,
                  esr_stream_t stream);
/* The k last rows of that argsort without sorting the columns (what dump_knn reads: train_cooccurence.py:114-126 takes
 * indices[-10:]): a radix select per column, k <= 1024.  out_indices / out_scores [T, k], best first:
 * out_indices[t][j] = argsort(scores[:, t])[V - 1 - j], ties exactly as the stable ascending argsort leaves them. */
int esr_topk_columns(const float* scores, int64_t V, int T, int k, float* out_scores, int32_t* out_indices,
                     esr_stream_t stream);
size_t esr_argsort_columns_workspace_bytes(int64_t V, int T);
int esr_argsort_columns(const float* scores, int64_t V, int T, int32_t* indices, void* workspace,
                        size_t workspace_bytes, esr_stream_t stream);

/* ---- find_top_k -- pinterest/make_recommendations.py:49-65 --------------------------------
 * scores[q, n] = queries[q] . candidates[n]; top-k per query, descending, ties -> lower index. */
size_t esr_score_topk_workspace_bytes(int64_t nq, int64_t N, int k);
int esr_score_topk(const float* queries, const float* candidates, int64_t nq, int64_t N, int D,
                   int k, float* out_scores, int32_t* out_indices, void* workspace,
                   size_t workspace_bytes, esr_stream_t stream);

/* ---- N3 / config 5: batched brute-force retrieval ------------------------------------------
 * find_top_k (pinterest/make_recommendations.py:49-65) for a BATCH of queries -- the scenes loop of
 * :123-132 in one call -- and the eval of spotify/train_spotify.py:120 (top_k(500) over every track):
 *   scores[q, n] = queries[q] . candidates[n];  per query the k best, descending, ties -> lower index.
 * queries f32 [nq, D], candidates f32 [N, D], k <= min(N, 1024).  The reported index of local candidate
 * n is index_base + n * index_step (row-sharded candidates: base = rank, step = world).
 * mode ESR_RETRIEVE_EXACT: products from three exact bf16 planes per operand (six MFMA cross terms,
 *   f32 accumulate) -- f32-equivalent scores, the brute-force answer.
 * mode ESR_RETRIEVE_BF16: one bf16 plane per operand -- the approximate candidate stage; follow with
 *   esr_rescore_candidates + esr_topk_merge for an exact re-rank of k' > k candidates.
 * mode ESR_RETRIEVE_F16X2: products from two fp16 planes of x * 2^e per operand (e per matrix, from its largest
 *   |element|; three MFMA cross terms, f32 accumulate) -- f32-grade scores (<= ~3 * 2^-24 |a||b| per elementary product)
 *   at half the matrix-core work of ESR_RETRIEVE_EXACT; one extra read of both matrices for the scales.
 * mode ESR_RETRIEVE_F16R (round 6): the EXACT top-k of the f32 scores from a ONE-term filter.  Every candidate is scored
 *   with the hi fp16 plane alone (one MFMA term instead of three); that score is off by at most b = 2^-10 (1 + 2 %) |q|
 *   max |c| (two roundings to 11 bits, Cauchy-Schwarz; row norms from the scaling pass), so the true top-k lies among
 *   the candidates whose one-term score reaches (k-th best one-term score) - 2 b: k of them have true scores >= that
 *   k-th best - b, and a member of the true top-k cannot score lower than those in truth.  The filter keeps exactly that
 *   band per query; the survivors (~1.2 k at N = 1 M, D = 512, k = 500) are re-scored as f32 dot products and selected.
 *   A query whose band outgrows its list (near-equal scores for thousands of candidates) is switched to exact scores on
 *   the spot and filtered against (exact k-th best) - b from then on: the answer never depends on the band being small.
 *   Scores are the f32 dot products (fmaf over 16 lanes + tree), ties -> lower index, like ESR_RETRIEVE_EXACT. */
#define ESR_RETRIEVE_EXACT 0
#define ESR_RETRIEVE_BF16 1
#define ESR_RETRIEVE_F16X2 2
#define ESR_RETRIEVE_F16R 3
size_t esr_retrieve_workspace_bytes(int64_t nq, int64_t N, int D, int k, int mode);
int esr_retrieve_topk(const float* queries, const float* candidates, int64_t nq, int64_t N, int D, int k,
                      int mode, int32_t index_base, int32_t index_step, float* out_scores,
                      int32_t* out_indices, void* workspace, size_t workspace_bytes, esr_stream_t stream);
/* A PREPARED corpus (round 6; every mode): everything a call does with the candidates before it looks at a query -- the
 * statistics pass (exponent, largest row norm: the scaled fp16 modes) and the split into the mode's planes, up to
 * 2 x 4 N D bytes of reads and 2 P N D of writes -- done ONCE for a corpus that serves many calls (the reference scores every batch of
 * scenes against the same product embeddings: pinterest/make_recommendations.py:123-132).  esr_retrieve_prepare fills
 * `prepared` (esr_retrieve_prepared_bytes(N, D, mode) bytes, 256-byte aligned, the caller's; valid while the candidate
 * matrix is unchanged; made for ONE mode); esr_retrieve_topk_prepared is esr_retrieve_topk of that mode with those passes
 * left out -- the same products, filter, band, re-score (from the f32 `candidates`, still an argument) and answer, bit for
 * bit. */
size_t esr_retrieve_prepared_bytes(int64_t N, int D, int mode);
int esr_retrieve_prepare(const float* candidates, int64_t N, int D, int mode, void* prepared, size_t prepared_bytes,
                         esr_stream_t stream);
int esr_retrieve_topk_prepared(const float* queries, const float* candidates, const void* prepared, int64_t nq, int64_t N,
                               int D, int k, int mode, int32_t index_base, int32_t index_step, float* out_scores,
                               int32_t* out_indices, void* workspace, size_t workspace_bytes, esr_stream_t stream);
/* scores[q, j] = queries[q] . candidates[(indices[q, j] - index_base) / index_step] in f32 (fmaf chain per
 * lane, wave reduction); indices < 0 give -inf.  indices int32 [nq, kc]. */
int esr_rescore_candidates(const float* queries, const float* candidates, int64_t nq, int64_t N, int D,
                           const int32_t* indices, int kc, int32_t index_base, int32_t index_step,
                           float* scores, esr_stream_t stream);
/* top-k of per-query (score, index) lists [nq, n] -> [nq, k], descending, ties -> lower index: the merge
 * of the shards' answers (after an all-gather) and of re-scored candidate lists. */
int esr_topk_merge(const float* scores, const int32_t* indices, int64_t nq, int n, int k,
                   float* out_scores, int32_t* out_indices, esr_stream_t stream);
/* hits[0] = sum over queries of |{j : exact[q, j] occurs in approx[q, :]}| (the word is zeroed by the call): recall@k of an
 * approximate answer against the brute-force one is hits / (nq * ke).  approx int32 [nq, ka], exact int32 [nq, ke], ka <=
 * 8192; O(ka + ke) per query (a hash set in LDS).  Build-defined: the reference has no ANN path to measure. */
int esr_recall_at_k(const int32_t* approx, int64_t nq, int ka, const int32_t* exact, int ke, unsigned long long* hits,
                    esr_stream_t stream);

/* ---- config 5's ANN leg: IVF search (build-defined; the reference has no ANN index -- esr_retrieve_topk stays the exact
 * answer and the yardstick for recall).  The index (esrecsys_amd/ivf.py builds it with the kernels above): candidates
 * grouped by coarse centroid -- cands_sorted [N, D] list after list, list_off int32 [nlist + 1], orig int32 [N] = the
 * row each one has in the caller's matrix, max_list = the longest list.  probe_lists int32 [nq, nprobe]: the lists each
 * query looks into, best first (the nprobe best centroids: esr_retrieve_topk of the queries against the centroids).
 * Scores inside the probed lists are exact f32 (grouped GEMM on the FP32 matrix cores, one 64 x 64 tile per workgroup)
 * and the answer is the exact top-k of the probed lists: the first probe slots are scored densely and give every query
 * a threshold (its k-th best so far), the others pass through a filtered epilogue in rounds of eight slots with a
 * compacting select in between -- the scheme of esr_retrieve_topk.  out [nq, k], best first, entries a query's lists
 * could not fill: score -inf, index -1.  k <= 1024. */
size_t esr_ivf_search_workspace_bytes(int64_t nq, int nlist, int max_list, int nprobe, int k);
int esr_ivf_search(const float* queries, int64_t nq, int D, const float* cands_sorted, const int32_t* list_off,
                   const int32_t* orig, int nlist, int max_list, const int32_t* probe_lists, int nprobe, int k,
                   float* out_scores, int32_t* out_indices, void* workspace, size_t workspace_bytes, esr_stream_t stream);
/* Index build helpers (esrecsys_amd/ivf.py).  esr_run_offsets: list_off[v] = first position of the ASCENDING list
 * sorted [n] whose value is >= v, for v = 0 .. nvalues (int32 [nvalues + 1]: the inverted lists' boundaries from the sorted
 * assignments); max_len (optional, int32 [1]) = the longest run.  esr_ivf_centroids: centroids[v] = the unit vector of
 * sums[v] [nlist, D] where list v has members (list_off[v + 1] > list_off[v]; list_off NULL: every list), else of training
 * row fallback_rows[v] -- the centroid step of spherical k-means. */
int esr_run_offsets(const int32_t* sorted, int64_t n, int nvalues, int32_t* list_off, int32_t* max_len, esr_stream_t stream);
int esr_ivf_centroids(const float* sums, const int32_t* list_off, const float* train, const int32_t* fallback_rows,
                      int nlist, int D, float* centroids, esr_stream_t stream);

/* ---- N1: Spotify id-embedding two-tower -- spotify/models.py:27-90, spotify/train_spotify.py:77-131 ----
 * A track embeds as concat(album_table[album mod n_album_rows], artist_table[artist]) ([.., 2F]).  One call is one
 * playlist: album_ids / artist_ids are int32 [n + m + o] = context, next, neg occurrences in that order (raw ids:
 * the +0.1 isin boosts of models.py:76-81 compare un-hashed ids).  n <= 32, 2F <= 256. */
size_t esr_spotify_workspace_bytes(int n, int m, int o, int F);
/* SpotifyModel.get_embeddings (models.py:37-51): out [count, 2F] = concat(album row (hashed), artist row); l2 [count]. */
int esr_spotify_get_embeddings(const float* album_table, int64_t n_album_rows, const float* artist_table,
                               int64_t n_artists, int F, const int32_t* album_ids, const int32_t* artist_ids,
                               int64_t count, float* out, float* l2, esr_stream_t stream);
/* SpotifyModel.__call__: pos [m], neg [o], the three flipped self-affinity matrices [n,n] [m,m] [o,o], l2 [n+m+o]. */
int esr_spotify_forward(const float* album_table, int64_t n_album_rows, const float* artist_table,
                        int64_t n_artists, int F, const int32_t* album_ids, const int32_t* artist_ids, int n,
                        int m, int o, float* pos, float* neg, float* ctx_self, float* next_self,
                        float* neg_self, float* l2, void* workspace, size_t workspace_bytes,
                        esr_stream_t stream);
/* value_and_grad of train_step's loss_fn (train_spotify.py:78-109): loss [1]; the gradient as per-occurrence rows
 * g_album_rows / g_artist_rows [n+m+o, F] (occurrence r -> album row album_rows[r] = hashed id, artist row
 * artist_ids[r]).  max / min split their cotangent evenly over ties; relu'(0) = 0 [upstream jax]. */
int esr_spotify_fwd_bwd(const float* album_table, int64_t n_album_rows, const float* artist_table,
                        int64_t n_artists, int F, const int32_t* album_ids, const int32_t* artist_ids, int n,
                        int m, int o, float regularization, float* loss, int32_t* album_rows,
                        float* g_album_rows, float* g_artist_rows, void* workspace, size_t workspace_bytes,
                        esr_stream_t stream);
/* eval_step's result[1] (train_spotify.py:113-119): affinity [T] of every track of the corpus to the n context
 * tracks (row max over the context + boosts). */
int esr_spotify_affinity_all(const float* album_table, int64_t n_album_rows, const float* artist_table,
                             int64_t n_artists, int F, const int32_t* ctx_album, const int32_t* ctx_artist,
                             int n, const int32_t* all_albums, const int32_t* all_artists, int64_t T,
                             float* affinity, esr_stream_t stream);
/* optax.sgd(lr, momentum) [upstream] in two halves: the decay over the whole table (trace *= momentum;
 * p -= lr * trace) and the row-sparse gradient (trace[row] += g; p[row] -= lr * g, duplicates summed in
 * occurrence order first).  Together: trace' = g + momentum * trace, p' = p - lr * trace'. */
int esr_dense_momentum_decay(float* param, float* trace, int64_t count, float lr, float momentum,
                             esr_stream_t stream);
int esr_sparse_momentum_scatter(float* table, float* trace, int64_t V, int D, const int32_t* sorted_ids,
                                const int32_t* perm, int64_t n, float* grad_rows, float lr,
                                esr_stream_t stream);
/* Lazy form of the same optimizer (row-sparse steps): no dense decay pass.  last [V] (int32, zero for a fresh state) is
 * the step each row is up to date with.  A step t >= 1 is: esr_momentum_catchup_rows on the ids the step will read
 * (ids[i] % modulus when modulus > 0: the album hash of spotify/models.py:37-41; duplicates welcome) -- the rows are
 * brought up to step t - 1 (n missed steps of trace *= momentum ; p -= lr * trace: one by one up to 2048, the closed form
 * beyond) and marked t -- then the forward / backward on current rows, then esr_sparse_momentum_step: trace = g +
 * momentum * trace ; p -= lr * trace on the touched rows (optax's order).  esr_momentum_flush brings EVERY row up to
 * `step` (before an eval, a checkpoint, any read of the whole table). */
int esr_momentum_catchup_rows(float* table, float* trace, int32_t* last, int64_t V, int D, const int32_t* ids, int64_t n,
                              int modulus, int step, float lr, float momentum, esr_stream_t stream);
int esr_sparse_momentum_step(float* table, float* trace, int64_t V, int D, const int32_t* sorted_ids,
                             const int32_t* perm, int64_t n, float* grad_rows, float lr, float momentum,
                             esr_stream_t stream);
int esr_momentum_flush(float* table, float* trace, int32_t* last, int64_t V, int D, int step, float lr, float momentum,
                       esr_stream_t stream);
int esr_momentum_catchup_rows2(float* table0, float* trace0, int32_t* last0, const int32_t* ids0, int modulus0,
                               float* table1, float* trace1, int32_t* last1, const int32_t* ids1, int modulus1, int D,
                               int64_t n, int step, float lr, float momentum, esr_stream_t stream);
int esr_sparse_momentum_step_multi(float* const* tables, float* const* traces, const int64_t* row_offsets, int ntables,
                                   int D, const int32_t* sorted_vids, const int32_t* perm, int64_t n, float* grad_rows,
                                   float lr, float momentum, esr_stream_t stream);
/* spotify/train_spotify.py:77-111 + 238-241 as ONE call (N1): catch-up of the playlist's rows (both tables, one launch),
 * esr_spotify_fwd_bwd, one sort of the virtual rows [album mod rows ; n_album_rows + artist], the whole momentum step on
 * the touched rows of both tables.  album_last / artist_last, step, lr, momentum as for esr_momentum_catchup_rows;
 * loss [1]. */
size_t esr_spotify_train_step_workspace_bytes(int n, int m, int o, int F);
int esr_spotify_train_step(float* album_table, float* album_trace, int32_t* album_last, int64_t n_album_rows,
                           float* artist_table, float* artist_trace, int32_t* artist_last, int64_t n_artists, int F,
                           const int32_t* album_ids, const int32_t* artist_ids, int n, int m, int o, float regularization,
                           int step, float lr, float momentum, float* loss, void* workspace, size_t workspace_bytes,
                           esr_stream_t stream);

/* ---- 8e: row-shard routing (owner = id mod world, local row = id div world) ---------------
 * Stable bucket of ids by owner: local_rows[k] = ids[perm[k]] / world, counts[g] = #ids owned by g
 * (int64, device); inverse (optional, may be NULL): inverse[perm[k]] = k, the bucket position of occurrence i.
 * Build-defined; the reference is single-device. */
size_t esr_bucket_workspace_bytes(int64_t n);
int esr_bucket_ids_by_owner(const int32_t* ids, int64_t n, int world, int32_t* local_rows,
                            int32_t* perm, int32_t* inverse, int64_t* counts, void* workspace,
                            size_t workspace_bytes, esr_stream_t stream);
/* The same bucket over a list given as up to four segments [ids_k + offsets[k]] (the lookups of one step as virtual
 * rows of the concatenated tables), read in place; perm / inverse index the concatenation.  ids / seg_counts /
 * offsets are host arrays as in esr_concat_offset_ids; workspace query with n = the sum of the counts. */
int esr_bucket_ids_by_owner_multi(const int32_t* const* ids, const int64_t* seg_counts, const int64_t* offsets,
                                  int nseg, int world, int32_t* local_rows, int32_t* perm, int32_t* inverse,
                                  int64_t* counts, void* workspace, size_t workspace_bytes, esr_stream_t stream);
/* The lists of `nbatch` (<= 8) coming batches bucketed in ONE launch pair (row-sharded loops make the routing plans of
 * several batches at once: esrecsys_amd/sharded.py begin_plans).  ids[b * nseg + k] = segment k of list b; seg_counts /
 * offsets are common to the lists (n = their sum); local_rows / perm / inverse are [nbatch][n], counts [nbatch][world]:
 * list b exactly what esr_bucket_ids_by_owner_multi gives for it. */
size_t esr_bucket_batched_workspace_bytes(int64_t n, int nbatch);
int esr_bucket_ids_by_owner_batched(const int32_t* const* ids, const int64_t* seg_counts, const int64_t* offsets,
                                    int nseg, int nbatch, int world, int32_t* local_rows, int32_t* perm,
                                    int32_t* inverse, int64_t* counts, void* workspace, size_t workspace_bytes,
                                    esr_stream_t stream);
/* The same routing with every DISTINCT row asked for once (esr_shard.hip): the occurrence list given as segments (as
 * esr_bucket_ids_by_owner_multi) -> ulocal [<= n]: the distinct local rows, owner-major and ascending inside an owner
 * (slice o, ucounts[o] entries, is what this rank asks of owner o and the order the rows come back in); ucounts [world]
 * (int64, device); uidx [n]: occurrence i reads row uidx[i] of the rows that came back; sorted_uidx / perm [n]: the
 * occurrences grouped by distinct row (sorted_uidx ascending; perm[p] = the occurrence) for esr_segment_sum_rows.
 * local_rows = the owner-local row space (max over owners of virtual rows div world, e.g. ceil(total / world)).
 * Owners need nothing new: they serve the list they are sent and segment-reduce what comes back by row. */
size_t esr_unique_by_owner_workspace_bytes(int64_t n);
int esr_unique_by_owner(const int32_t* const* ids, const int64_t* seg_counts, const int64_t* offsets, int nseg, int world,
                        int64_t local_rows, int32_t* ulocal, int32_t* uidx, int32_t* sorted_uidx, int32_t* perm,
                        int64_t* ucounts, void* workspace, size_t workspace_bytes, esr_stream_t stream);
/* The plan phase of the overlapped row-sharded loop (esrecsys_amd/sharded.py begin_stale_sets), batched over nlists lists:
 * esr_sorted_membership: flags[l][j] = 1 iff cur[l][j] != sentinel and it occurs in the ascending list seq[l][0 .. m)
 *   (cur int32 [nlists][n], seq int32 [nlists][m], flags uint8 [nlists][n]).
 * esr_flagged_first: out[l][0 .. c_l) = the flagged entries of values[l] (values NULL: their positions) in order -- a
 *   stable partition, only the flagged prefix is written -- and counts[l * counts_stride_list + g * counts_stride_slice]
 *   = flagged entries inside the g-th of G consecutive slices of lengths slice_len[l][g] (int64 [nlists][G], device; G
 *   <= 64); the count words are zeroed by the call.  Workspace: esr_flagged_first_workspace_bytes. */
int esr_sorted_membership(const int32_t* cur, int64_t n, const int32_t* seq, int64_t m, int nlists, int32_t sentinel,
                          uint8_t* flags, esr_stream_t stream);
size_t esr_flagged_first_workspace_bytes(int64_t n, int nlists);
int esr_flagged_first(const uint8_t* flags, const int32_t* values, int64_t n, int nlists, const int64_t* slice_len, int G,
                      int32_t* out, int64_t* counts, int64_t counts_stride_list, int64_t counts_stride_slice,
                      void* workspace, size_t workspace_bytes, esr_stream_t stream);
/* out[perm[k], :] = rows[k, :]  (undo the bucket order for rows that came back). */
int esr_unpermute_rows(const void* rows, int dtype, int D, const int32_t* perm, int64_t n, void* out,
                       esr_stream_t stream);

/* out[perm[k], :] = (float) rows_bf16[k, :] (perm == NULL: identity).  Requester side of a sharded lookup on a
 * bf16 table (BASELINE config 4): rows cross xGMI as bf16, the loss kernels consume f32.  D % 4 == 0. */
int esr_unpermute_rows_bf16_to_f32(const void* rows_bf16, int D, const int32_t* perm, int64_t n, float* out,
                                   esr_stream_t stream);

/* ---- 8e: the exchange itself -- all-to-all(v) of ids / rows / gradient rows over RCCL (xGMI) -------------------
 * Build-defined (the reference is single-device).  RCCL is bound at run time with dlopen: the library
 * has no link-time dependency on it.  An exchange is ncclGroupStart ; per peer ncclSend + ncclRecv ; ncclGroupEnd
 * enqueued on `stream`, i.e. in stream order with the kernels around it (no internal stream, no hand-over events);
 * on the xGMI full mesh every peer slice rides its own direct link.
 *   esr_comm_load       host: bind librccl (path NULL/"" = the librccl.so.1 already loaded in the process, else the
 *                       loader path).  Optional: the other entry points call it with NULL on first use.
 *   esr_comm_unique_id  host: 128 bytes of ncclUniqueId (rank 0 calls it and hands the bytes to every rank).
 *   esr_comm_init       COLLECTIVE over the `world` ranks, on the calling thread's current HIP device.
 *   esr_comm_count      host out-params: what RCCL itself reports for this communicator.
 *   esr_comm_async_error  ESR_OK or ESR_ELAUNCH when RCCL has recorded an asynchronous failure.
 *   esr_comm_abort / esr_comm_destroy  free the communicator (abort also terminates enqueued operations).
 * send_counts / recv_counts are HOST int64 [world] arrays: slice p of the send buffer (send_counts[p] ids / rows)
 * goes to peer p, slice p of the receive buffer comes from peer p.  Arguments are validated before the group opens
 * and the group is always closed, so a failing call never leaves peers inside an open group. */
typedef void* esr_comm_t;
int esr_comm_load(const char* librccl_path);
int esr_comm_unique_id(void* uid128);
int esr_comm_init(const void* uid128, int world, int rank, esr_comm_t* comm);
int esr_comm_count(esr_comm_t comm, int* world, int* rank);
int esr_comm_async_error(esr_comm_t comm);
int esr_comm_abort(esr_comm_t comm);
int esr_comm_destroy(esr_comm_t comm);
int esr_alltoall_bytes(esr_comm_t comm, const void* send, const int64_t* send_bytes, void* recv,
                       const int64_t* recv_bytes, esr_stream_t stream);
/* n_ops all-to-all(v)s as ONE RCCL group: operation o sends slice p of send[o] (send_bytes[o * world + p] bytes, host
 * arrays) to peer p and receives slice p of recv[o] from it.  The ids exchanges of a group of routing plans. */
int esr_alltoall_bytes_multi(esr_comm_t comm, int n_ops, const void* const* send, const int64_t* send_bytes,
                             void* const* recv, const int64_t* recv_bytes, esr_stream_t stream);
/* All-gather of equal blocks: block r of `recv` (world x bytes) = rank r's `send` (bytes).  The replicated-table mode
 * gathers every rank's ids and gradient rows with it; the sharded retrieval its queries. */
int esr_allgather_bytes(esr_comm_t comm, const void* send, int64_t bytes, void* recv, esr_stream_t stream);
/* int32 virtual local rows -> their owners (step 2 of SURVEY 8e). */
int esr_alltoall_ids(esr_comm_t comm, const int32_t* send_ids, const int64_t* send_counts, int32_t* recv_ids,
                     const int64_t* recv_counts, esr_stream_t stream);
/* looked-up rows [n, D] in the table dtype, owners -> requesters (step 3). */
int esr_alltoall_rows(esr_comm_t comm, const void* send_rows, int dtype, int D, const int64_t* send_counts,
                      void* recv_rows, const int64_t* recv_counts, esr_stream_t stream);
/* fp32 gradient rows [n, D], requesters -> owners (step 5). */
int esr_alltoall_grads(esr_comm_t comm, const float* send_grads, int D, const int64_t* send_counts,
                       float* recv_grads, const int64_t* recv_counts, esr_stream_t stream);

/* ---- a row-sharded step's exchange halves as ONE call each (esr_shard_step.hip; the loss kernel sits between them) ----
 * The steps being sharded: pinterest/train_shop_the_look.py:93-109, wikipedia/train_cooccurence.py:71-101 (the reference
 * is single-device and has no counterpart).  Count arrays are host int64 [world], peer-major, as in esr_alltoall_*:
 *   asked_counts[p]  rows peer p asks of THIS rank (their virtual local rows: asked_rows, concatenated peer by peer)
 *   ask_counts[p]    rows this rank asks of peer p (the routing plan's send counts)
 * esr_sharded_lookup: gather asked_rows from this rank's shards (esr_gather_rows_multi) into `served`
 *   [sum asked_counts, D], exchange (esr_alltoall_rows) into `back` [sum ask_counts, D], both in the tables' dtype.
 * esr_sharded_update: unique plans (sorted_uidx / occ_perm of esr_unique_by_owner non-NULL): ONE summed row per distinct
 *   row (esr_segment_sum_rows into `summed` [sum ask_counts, D]; grad_rows [n_occ, D] may be overwritten), else grad_rows
 *   are already one row per exchanged row; rows to their owners (esr_alltoall_grads into recv_grads [sum asked_counts,
 *   D]); the owner's fused segment-reduce + Adagrad (esr_sparse_adagrad_scatter_multi) with the sort of asked_rows the
 *   plan phase made (owner_sorted / owner_perm).  grad_dtype ESR_BF16: the rows cross the exchange as bf16 (rounded to
 *   nearest-even after the per-row sum into send_bf16 [sum ask_counts, D], received into recv_raw, widened into
 *   recv_grads; element error <= 2^-9 relative; D % 8 == 0) -- SURVEY 8d's config-4 budget of bf16-sized gradients.
 * world == 1 (comm may be NULL): nothing is exchanged or copied -- the gather writes into `back`, the update reads the
 * asker's rows in place; served / recv_grads / send_bf16 / recv_raw are not touched. */
/* rows [n, D] f32 -> bf16, round to nearest even (NaN stays NaN); D % 8 == 0.  The narrowing half of the bf16 gradient
 * exchange (esr_unpermute_rows_bf16_to_f32 with perm = NULL is the widening half). */
int esr_rows_f32_to_bf16(const float* rows, int64_t n, int D, void* rows_bf16, esr_stream_t stream);
int esr_sharded_lookup(esr_comm_t comm, int world, const void* const* tables, const int64_t* row_offsets, int ntables,
                       int dtype, int D, const int32_t* asked_rows, const int64_t* asked_counts,
                       const int64_t* ask_counts, void* served, void* back, esr_stream_t stream);
int esr_sharded_update(esr_comm_t comm, int world, void* const* tables, float* const* accums,
                       const int64_t* row_offsets, int ntables, int dtype, int D, float* grad_rows, int64_t n_occ,
                       const int32_t* sorted_uidx, const int32_t* occ_perm, float* summed, const int64_t* ask_counts,
                       const int64_t* asked_counts, int grad_dtype, void* send_bf16, void* recv_raw, float* recv_grads,
                       const int32_t* owner_sorted, const int32_t* owner_perm, float lr, float eps, int long_runs,
                       esr_stream_t stream);

/* ---- a whole row-sharded step -- lookup, loss kernel, update -- as ONE call (esr_shard_step.hip) -----------------------
 * A group = the same-width tables a step looks up and updates together, as this rank holds them (host struct of host
 * arrays of device pointers); a routing plan = what esrecsys_amd/sharded.py's plan phase made for one batch: */
typedef struct {
  esr_comm_t comm;              /* esr_comm_init's communicator; may be NULL at world 1 */
  int world;
  void* const* tables;          /* [ntables] this rank's shards */
  float* const* accums;         /* [ntables] their Adagrad accumulators */
  const int64_t* row_offsets;   /* [ntables + 1] virtual LOCAL row boundaries of the concatenated shards */
  int ntables;
  int dtype;                    /* ESR_F32 / ESR_BF16 (whole steps: ESR_F32) */
  int D;
  int grad_dtype;               /* ESR_F32, or ESR_BF16: gradient rows cross the exchange as bf16 (esr_sharded_update) */
} esr_shard_group_t;
typedef struct {
  const int32_t* asked_rows;    /* device [sum asked_counts]: virtual local rows the peers ask of this rank */
  const int64_t* asked_counts;  /* host [world] */
  const int64_t* ask_counts;    /* host [world]: rows this rank asks of each peer */
  const int32_t* index;         /* device [occurrences]: occurrence i reads row index[i] of the rows that come back */
  const int32_t* sorted_uidx;   /* device [occurrences] -- esr_unique_by_owner's grouping of the occurrences by distinct */
  const int32_t* occ_perm;      /*   row; both NULL for a per-occurrence plan (index = the inverse bucket permutation) */
  const int32_t* owner_sorted;  /* device [sum asked_counts]: esr_segment_sort_ids of asked_rows ... */
  const int32_t* owner_perm;    /*   ... and its permutation, for the owner-side update */
  int long_runs;                /* esr_sparse_adagrad_scatter_multi's hint (-1: unknown) */
} esr_routing_plan_t;
/* esr_sharded_triplet_step: the reference triplet loss (pinterest/train_shop_the_look.py:93-109) on row-sharded towers
 * (group = [scene table, product table]; occurrences = [scene ; pos ; neg] ids, 3 B of them): esr_sharded_lookup ->
 * esr_triplet_fwd_bwd on the rows where they landed -> esr_sharded_update.  loss [1] = this rank's share (a sum over
 * triplets / batch_size: all-reduce for the global value).
 * esr_sharded_glove_step: wikipedia/train_cooccurence.py:71-101 on a row-sharded embedding table and its [V, 1] bias table
 * (two single-table groups, ONE plan: same ids, same sharding; occurrences = inputs [2, B] flattened): both lookups ->
 * esr_glove_fwd_bwd -> both updates.  The loss is over the local batch.
 * Workspace: caller-owned, 256-byte aligned, at least the *_workspace_bytes of the SAME group(s), plan and B. */
size_t esr_sharded_triplet_step_workspace_bytes(const esr_shard_group_t* towers, const esr_routing_plan_t* plan,
                                                int64_t B);
int esr_sharded_triplet_step(const esr_shard_group_t* towers, const esr_routing_plan_t* plan, int64_t B,
                             float regularization, float batch_size, float lr, float eps, float* loss, void* workspace,
                             size_t workspace_bytes, esr_stream_t stream);
size_t esr_sharded_glove_step_workspace_bytes(const esr_shard_group_t* emb, const esr_shard_group_t* bias,
                                              const esr_routing_plan_t* plan, int64_t B);
int esr_sharded_glove_step(const esr_shard_group_t* emb, const esr_shard_group_t* bias, const esr_routing_plan_t* plan,
                           const float* target, int64_t B, int mode, float lr, float eps, float* loss, void* workspace,
                           size_t workspace_bytes, esr_stream_t stream);

/* ---- the same steps with the NEXT batch's lookup overlapped (build-defined) ---------------------------------
 * A training loop knows its coming batches (the plans are made a group ahead), so batch k + 1's gather + rows exchange
 * need not wait for batch k: call k issues them on `side` with the second communicator `comm2` right after batch k's own
 * rows are in place -- they run under batch k's loss kernel, gradient exchange and update -- and call k + 1 receives the
 * buffers as `back`.  Rows that batch k's update writes were fetched too early; which ones is a function of the ids alone
 * (owner side: the rows of batch k + 1's asked list that are in batch k's; esrecsys_amd/sharded.py begin_stale_sets),
 * and call k + 1 serves exactly those again on the main stream before its loss kernel: gather stale_rows -> rows
 * exchange (first communicator; counts stale_asked / stale_ask) -> scatter to rows stale_pos of `back`.  The results
 * equal the plain steps' bit for bit (tests/test_gpu_sharded.py, tests/test_sharded_gloo.py).
 * One struct per call; `ready` / `next_ready` hand the side stream's completion event from call to call (the consumer
 * releases it; esr_sharded_overlap_release for one that is never consumed).  Buffers in next_back / next_served belong to
 * the caller and must stay untouched until the call that takes them as `back` has been issued. */
typedef struct {
  void* back[2];                /* in: THIS batch's rows per group (GloVe: embedding, bias) = the previous call's next_back;
                                   NULL: look them up in line (first step of a loop) */
  void* ready;                  /* in: the previous call's next_ready (waited for on `stream`, then released), or NULL */
  const int32_t* stale_rows;    /* device [sum stale_asked]: virtual local rows this rank serves again, asker by asker */
  const int64_t* stale_asked;   /* host [world] */
  const int32_t* stale_pos;     /* device [sum stale_ask]: the rows of `back` that come again, owner by owner */
  const int64_t* stale_ask;     /* host [world] */
  const esr_routing_plan_t* next_plan; /* the NEXT batch's plan, or NULL (last step) */
  void* next_back[2];           /* per group: [sum next ask_counts, D], filled on `side` */
  void* next_served[2];         /* per group: scratch [sum next asked_counts, D] (world > 1) */
  esr_comm_t comm2;             /* a second esr_comm_init communicator over the same ranks (may be NULL at world 1) */
  esr_stream_t side;            /* the stream the next lookup runs on */
  void* next_ready;             /* out: recorded on `side` behind the next lookup (NULL when next_plan is NULL) */
} esr_step_overlap_t;
/* extra workspace of ONE group for its stale rows (add it to the step's *_workspace_bytes for every group of the step) */
size_t esr_sharded_step_overlap_workspace_bytes(const esr_shard_group_t* group, const esr_step_overlap_t* overlap);
void esr_sharded_overlap_release(void* ready);
/* overlap == NULL: exactly esr_sharded_triplet_step / esr_sharded_glove_step */
int esr_sharded_triplet_step_overlapped(const esr_shard_group_t* towers, const esr_routing_plan_t* plan,
                                        esr_step_overlap_t* overlap, int64_t B, float regularization, float batch_size,
                                        float lr, float eps, float* loss, void* workspace, size_t workspace_bytes,
                                        esr_stream_t stream);
int esr_sharded_glove_step_overlapped(const esr_shard_group_t* emb, const esr_shard_group_t* bias,
                                      const esr_routing_plan_t* plan, esr_step_overlap_t* overlap, const float* target,
                                      int64_t B, int mode, float lr, float eps, float* loss, void* workspace,
                                      size_t workspace_bytes, esr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ESR_HIP_H_ */
