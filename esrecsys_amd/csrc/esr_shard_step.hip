// The two halves of a row-sharded step that are not the loss kernel, each as ONE library call (build-defined
// -- the reference is single-device: pinterest/train_shop_the_look.py:93-109 and wikipedia/train_cooccurence.py:71-101 are
// the steps being sharded).  A step on row-sharded tables is
//
//   lookup   owner: gather the rows it was asked for  ->  rows back to the askers            esr_sharded_lookup
//   loss     the asker's loss kernel on the rows where they landed (esr_triplet_fwd_bwd / esr_glove_fwd_bwd /
//            esr_inbatch_towers_fwd_bwd_*, indexing them through the routing plan)
//   update   asker: ONE summed gradient row per distinct row (unique plans)  ->  gradient rows to the owners  ->
//            owner: fused segment-reduce + Adagrad on its shards                             esr_sharded_update
//
// Round 3 issued these as six library calls from Python with an allocation in front of each: at world 1 half of a
// triplet step's 96 us was an idle queue.  Here the sequence is enqueued by one call into caller-owned scratch; the
// exchange is the direct RCCL one of esr_comm.hip on the same stream.  world == 1: nothing is exchanged -- the gather
// writes straight into `back` and the owner-side update reads the asker's gradient rows in place (no self copy).
//
// Gradient rows may cross the exchange as bf16 (grad_dtype = ESR_BF16; BASELINE config 4 budgets bf16-sized gradients:
// SURVEY 8d, ~910 B/pair over xGMI): rounded to nearest-even after the per-distinct-row sum, widened again on the owner;
// relative error of a gradient element <= 2^-9, the accumulator and the parameter update stay f32.
#include "esr_common.h"
#include "../../include/esr_hip.h"

namespace esr {

// rows of f32 -> bf16 (round to nearest even; NaN stays NaN), 8 elements per thread and trip
__global__ __launch_bounds__(kBlock) void rows_f32_to_bf16_kernel(const float4* __restrict__ src, uint4* __restrict__ dst,
                                                                 int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (int64_t)gridDim.x * kBlock) {
    const float4 a = src[2 * i], b = src[2 * i + 1];
    auto rn = [](float x) -> uint32_t {
      const uint32_t u = __float_as_uint(x);
      if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;  // NaN: keep it one
      return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    };
    uint4 o;
    o.x = rn(a.x) | (rn(a.y) << 16);
    o.y = rn(a.z) | (rn(a.w) << 16);
    o.z = rn(b.x) | (rn(b.y) << 16);
    o.w = rn(b.z) | (rn(b.w) << 16);
    dst[i] = o;
  }
}

static int64_t sum_counts(const int64_t* c, int world) {
  int64_t s = 0;
  for (int p = 0; p < world; ++p) s += c[p];
  return s;
}

// one table: the single-table calls (any row width -- the GloVe bias column is one float); several: the fused ones
static int gather_any(const void* const* tables, const int64_t* row_offsets, int ntables, int dtype, int D,
                      const int32_t* rows, int64_t n, void* out, esr_stream_t stream) {
  if (ntables == 1 && tables && row_offsets)
    return esr_gather_rows(tables[0], dtype, row_offsets[1] - row_offsets[0], D, rows, n, out, stream);
  return esr_gather_rows_multi(tables, row_offsets, ntables, dtype, D, rows, n, out, stream);
}

}  // namespace esr

using namespace esr;

extern "C" {

int esr_rows_f32_to_bf16(const float* rows, int64_t n, int D, void* rows_bf16, esr_stream_t stream) {
  ESR_REQUIRE(n >= 0 && D > 0 && D % 8 == 0, "esr_rows_f32_to_bf16: bad sizes n=%lld D=%d (D %% 8 == 0)", (long long)n, D);
  if (n == 0) return ESR_OK;
  ESR_REQUIRE(rows && rows_bf16 && !(((uintptr_t)rows | (uintptr_t)rows_bf16) & 15), "esr_rows_f32_to_bf16: null or misaligned pointer");
  const int64_t n8 = n * D / 8;
  hipLaunchKernelGGL(rows_f32_to_bf16_kernel, dim3((unsigned)std::min<int64_t>(kMaxGrid, cdiv(n8, kBlock))), dim3(kBlock), 0,
                     as_stream(stream), (const float4*)rows, (uint4*)rows_bf16, n8);
  return check_launch("esr_rows_f32_to_bf16");
}

int esr_sharded_lookup(esr_comm_t comm, int world, const void* const* tables, const int64_t* row_offsets, int ntables,
                       int dtype, int D, const int32_t* asked_rows, const int64_t* asked_counts,
                       const int64_t* ask_counts, void* served, void* back, esr_stream_t stream) {
  TraceScope trace_scope_("esr_sharded_lookup");
  ESR_REQUIRE(world >= 1 && asked_counts && ask_counts, "esr_sharded_lookup: world=%d, or null count arrays", world);
  for (int p = 0; p < world; ++p)
    ESR_REQUIRE(asked_counts[p] >= 0 && ask_counts[p] >= 0, "esr_sharded_lookup: negative count for peer %d", p);
  const int64_t n_served = sum_counts(asked_counts, world), n_back = sum_counts(ask_counts, world);
  ESR_REQUIRE(n_back == 0 || back, "esr_sharded_lookup: null `back`");
  if (world == 1) {  // the only peer is this rank: its rows are gathered where the loss kernel reads them
    ESR_REQUIRE(n_served == n_back, "esr_sharded_lookup: a world of one rank asks itself %lld rows and serves %lld",
                (long long)n_back, (long long)n_served);
    return gather_any(tables, row_offsets, ntables, dtype, D, asked_rows, n_served, back, stream);
  }
  ESR_REQUIRE(comm, "esr_sharded_lookup: null communicator at world %d", world);
  ESR_REQUIRE(n_served == 0 || served, "esr_sharded_lookup: null `served` scratch");
  if (int rc = gather_any(tables, row_offsets, ntables, dtype, D, asked_rows, n_served, served, stream)) return rc;
  return esr_alltoall_rows(comm, served, dtype, D, asked_counts, back, ask_counts, stream);
}

int esr_sharded_update(esr_comm_t comm, int world, void* const* tables, float* const* accums,
                       const int64_t* row_offsets, int ntables, int dtype, int D, float* grad_rows, int64_t n_occ,
                       const int32_t* sorted_uidx, const int32_t* occ_perm, float* summed, const int64_t* ask_counts,
                       const int64_t* asked_counts, int grad_dtype, void* send_bf16, void* recv_raw, float* recv_grads,
                       const int32_t* owner_sorted, const int32_t* owner_perm, float lr, float eps, int long_runs,
                       esr_stream_t stream) {
  TraceScope trace_scope_("esr_sharded_update");
  ESR_REQUIRE(world >= 1 && asked_counts && ask_counts, "esr_sharded_update: world=%d, or null count arrays", world);
  ESR_REQUIRE(grad_dtype == ESR_F32 || grad_dtype == ESR_BF16, "esr_sharded_update: grad_dtype must be ESR_F32 or ESR_BF16");
  ESR_REQUIRE(D > 0 && n_occ >= 0, "esr_sharded_update: bad sizes D=%d n=%lld", D, (long long)n_occ);
  for (int p = 0; p < world; ++p)
    ESR_REQUIRE(asked_counts[p] >= 0 && ask_counts[p] >= 0, "esr_sharded_update: negative count for peer %d", p);
  const int64_t n_rows = sum_counts(ask_counts, world);     // rows that leave this rank: n_occ, or the distinct rows
  const int64_t n_recv = sum_counts(asked_counts, world);   // rows this rank owns and receives gradients for
  const bool unique = sorted_uidx != nullptr;
  ESR_REQUIRE(unique || n_rows == n_occ, "esr_sharded_update: %lld rows leave but %lld gradient rows were given "
              "(per-occurrence plans send one row per occurrence)", (long long)n_rows, (long long)n_occ);
  ESR_REQUIRE(n_occ == 0 || grad_rows, "esr_sharded_update: null gradient rows");
  float* out_rows = grad_rows;
  if (unique && n_occ > 0) {  // one summed row per distinct row, in the order the rows were asked for
    ESR_REQUIRE(occ_perm && summed && n_rows > 0, "esr_sharded_update: unique plan without occ_perm / `summed` scratch");
    if (int rc = esr_segment_sum_rows(summed, n_rows, D, sorted_uidx, occ_perm, n_occ, grad_rows, stream)) return rc;
    out_rows = summed;
  }
  float* owner_rows = out_rows;  // world 1: the owner IS the asker -- its update reads the rows in place
  if (world > 1) {
    ESR_REQUIRE(comm, "esr_sharded_update: null communicator at world %d", world);
    ESR_REQUIRE(n_recv == 0 || recv_grads, "esr_sharded_update: null `recv_grads` scratch");
    if (grad_dtype == ESR_BF16) {
      ESR_REQUIRE(D % 8 == 0, "esr_sharded_update: bf16 gradient rows need D %% 8 == 0 (D=%d)", D);
      ESR_REQUIRE((n_rows == 0 || send_bf16) && (n_recv == 0 || recv_raw),
                  "esr_sharded_update: bf16 gradient exchange without its scratch buffers");
      if (int rc = esr_rows_f32_to_bf16(out_rows, n_rows, D, send_bf16, stream)) return rc;
      if (int rc = esr_alltoall_rows(comm, send_bf16, ESR_BF16, D, ask_counts, recv_raw, asked_counts, stream)) return rc;
      if (n_recv > 0)
        if (int rc = esr_unpermute_rows_bf16_to_f32(recv_raw, D, nullptr, n_recv, recv_grads, stream)) return rc;
    } else {
      if (int rc = esr_alltoall_grads(comm, out_rows, D, ask_counts, recv_grads, asked_counts, stream)) return rc;
    }
    owner_rows = recv_grads;
  } else {
    ESR_REQUIRE(n_recv == n_rows, "esr_sharded_update: a world of one rank sends itself %lld rows and expects %lld",
                (long long)n_rows, (long long)n_recv);
  }
  if (n_recv == 0) return ESR_OK;
  ESR_REQUIRE(owner_sorted && owner_perm, "esr_sharded_update: null owner-side sort");
  if (ntables == 1 && tables && accums && row_offsets)
    return esr_sparse_adagrad_scatter(tables[0], dtype, accums[0], row_offsets[1] - row_offsets[0], D, owner_sorted,
                                      owner_perm, n_recv, owner_rows, lr, eps, stream);
  return esr_sparse_adagrad_scatter_multi(tables, accums, row_offsets, ntables, dtype, D, owner_sorted, owner_perm,
                                          n_recv, owner_rows, lr, eps, long_runs, stream);
}


// ---- whole steps: lookup -> loss kernel -> update as ONE call -----------------------------------------------------------
namespace {
struct StepScratch {
  char* p;
  size_t left;
  void* take(size_t bytes) {
    bytes = align_up(bytes, 256);
    if (bytes > left) return nullptr;
    void* r = p;
    p += bytes;
    left -= bytes;
    return r;
  }
};
// scratch of one group's exchange halves (rows of D elements): served, back, summed, recv_grads, bf16 send / recv
size_t group_scratch_bytes(const esr_shard_group_t* g, int64_t n_occ, int64_t n_rows, int64_t n_recv, bool unique) {
  const size_t es = g->dtype == ESR_BF16 ? 2 : 4, D = (size_t)g->D;
  size_t b = align_up((size_t)n_rows * D * es, 256);                          // back
  if (g->world > 1) b += align_up((size_t)n_recv * D * es, 256);              // served
  if (unique) b += align_up((size_t)n_rows * D * 4, 256);                     // summed
  if (g->world > 1) b += align_up((size_t)n_recv * D * 4, 256);               // recv_grads
  if (g->world > 1 && g->grad_dtype == ESR_BF16 && g->D % 8 == 0)
    b += align_up((size_t)n_rows * D * 2, 256) + align_up((size_t)n_recv * D * 2, 256);
  (void)n_occ;
  return b;
}
int plan_counts(const char* who, const esr_shard_group_t* g, const esr_routing_plan_t* plan, int64_t* n_rows,
                int64_t* n_recv) {
  ESR_REQUIRE(g && plan && g->world >= 1 && plan->ask_counts && plan->asked_counts, "%s: null group / plan", who);
  *n_rows = sum_counts(plan->ask_counts, g->world);
  *n_recv = sum_counts(plan->asked_counts, g->world);
  return ESR_OK;
}
int group_lookup(const esr_shard_group_t* g, const esr_routing_plan_t* plan, int64_t n_rows, int64_t n_recv,
                 StepScratch& sc, void** back, esr_stream_t stream) {
  const size_t es = g->dtype == ESR_BF16 ? 2 : 4;
  *back = sc.take((size_t)n_rows * g->D * es);
  void* served = g->world > 1 ? sc.take((size_t)n_recv * g->D * es) : nullptr;
  if (!*back || (g->world > 1 && !served)) {
    set_error("sharded step: workspace too small");
    return ESR_EWORKSPACE;
  }
  return esr_sharded_lookup(g->comm, g->world, g->tables, g->row_offsets, g->ntables, g->dtype, g->D, plan->asked_rows,
                            plan->asked_counts, plan->ask_counts, served, *back, stream);
}
int group_update(const esr_shard_group_t* g, const esr_routing_plan_t* plan, float* grad_rows, int64_t n_occ,
                 int64_t n_rows, int64_t n_recv, bool unique, StepScratch& sc, float lr, float eps,
                 esr_stream_t stream) {
  const bool bf16 = g->world > 1 && g->grad_dtype == ESR_BF16 && g->D % 8 == 0;
  float* summed = unique ? (float*)sc.take((size_t)n_rows * g->D * 4) : nullptr;
  float* recv = g->world > 1 ? (float*)sc.take((size_t)n_recv * g->D * 4) : nullptr;
  void* send_h = bf16 ? sc.take((size_t)n_rows * g->D * 2) : nullptr;
  void* recv_h = bf16 ? sc.take((size_t)n_recv * g->D * 2) : nullptr;
  if ((unique && !summed) || (g->world > 1 && !recv) || (bf16 && (!send_h || !recv_h))) {
    set_error("sharded step: workspace too small");
    return ESR_EWORKSPACE;
  }
  return esr_sharded_update(g->comm, g->world, g->tables, g->accums, g->row_offsets, g->ntables, g->dtype, g->D,
                            grad_rows, n_occ, unique ? plan->sorted_uidx : nullptr, unique ? plan->occ_perm : nullptr,
                            summed, plan->ask_counts, plan->asked_counts, bf16 ? ESR_BF16 : ESR_F32, send_h, recv_h, recv,
                            plan->owner_sorted, plan->owner_perm, lr, eps, plan->long_runs, stream);
}
#define ESR_HIP(call)                                                                  \
  do {                                                                                 \
    const hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess) {                                                            \
      set_error("sharded step (overlapped): %s: %s", #call, hipGetErrorString(e_));   \
      return ESR_ELAUNCH;                                                              \
    }                                                                                  \
  } while (0)

// ---- the overlap of SURVEY 8e: this batch's rows were looked up by the PREVIOUS call, under its kernels ---------------
int64_t stale_out(const esr_shard_group_t* g, const esr_step_overlap_t* ov) { return sum_counts(ov->stale_asked, g->world); }
int64_t stale_in(const esr_shard_group_t* g, const esr_step_overlap_t* ov) { return sum_counts(ov->stale_ask, g->world); }
size_t patch_scratch_bytes(const esr_shard_group_t* g, const esr_step_overlap_t* ov) {
  if (!ov || !ov->stale_asked || !ov->stale_ask) return 0;
  const size_t es = g->dtype == ESR_BF16 ? 2 : 4;
  return align_up((size_t)stale_out(g, ov) * g->D * es, 256) + (g->world > 1 ? align_up((size_t)stale_in(g, ov) * g->D * es, 256) : 0);
}
// The rows of `back` that the previous step's update wrote after they were fetched: served again (gather -> rows exchange
// on the FIRST communicator and the main stream -> scatter to their places).  Every rank calls the exchange, also with
// nothing to send.
int group_patch(const esr_shard_group_t* g, const esr_step_overlap_t* ov, void* back, StepScratch& sc, esr_stream_t stream) {
  ESR_REQUIRE(ov->stale_asked && ov->stale_ask, "sharded step (overlapped): rows looked up ahead need the stale-row lists");
  const int64_t n_out = stale_out(g, ov), n_in = stale_in(g, ov);
  const size_t es = g->dtype == ESR_BF16 ? 2 : 4;
  ESR_REQUIRE((n_out == 0 || ov->stale_rows) && (n_in == 0 || ov->stale_pos), "sharded step (overlapped): null stale-row list");
  void* served = sc.take((size_t)n_out * g->D * es);
  void* got = g->world > 1 ? sc.take((size_t)n_in * g->D * es) : served;
  if (!served || !got) {
    set_error("sharded step (overlapped): workspace too small for the stale rows");
    return ESR_EWORKSPACE;
  }
  if (n_out > 0)
    if (int rc = gather_any(g->tables, g->row_offsets, g->ntables, g->dtype, g->D, ov->stale_rows, n_out, served, stream)) return rc;
  if (g->world > 1) {
    ESR_REQUIRE(g->comm, "sharded step (overlapped): null communicator at world %d", g->world);
    if (int rc = esr_alltoall_rows(g->comm, served, g->dtype, g->D, ov->stale_asked, got, ov->stale_ask, stream)) return rc;
  } else {
    ESR_REQUIRE(n_out == n_in, "sharded step (overlapped): a world of one rank re-serves %lld rows and expects %lld",
                (long long)n_out, (long long)n_in);
  }
  if (n_in == 0) return ESR_OK;
  return esr_unpermute_rows(got, g->dtype, g->D, ov->stale_pos, n_in, back, stream);
}
// this batch's rows of group `gi`: patched if they came from the previous call, else looked up in line
int group_rows(const esr_shard_group_t* g, const esr_routing_plan_t* plan, const esr_step_overlap_t* ov, int gi,
               int64_t n_rows, int64_t n_recv, StepScratch& sc, void** back, esr_stream_t stream) {
  if (ov && ov->back[gi]) {
    *back = ov->back[gi];
    return group_patch(g, ov, *back, sc, stream);
  }
  return group_lookup(g, plan, n_rows, n_recv, sc, back, stream);
}
// main stream: wait for the lookup the previous call left on the side stream
int overlap_enter(esr_step_overlap_t* ov, esr_stream_t stream) {
  if (!ov) return ESR_OK;
  ov->next_ready = nullptr;
  if (ov->ready) {
    hipEvent_t ev = (hipEvent_t)ov->ready;
    ESR_HIP(hipStreamWaitEvent(as_stream(stream), ev, 0));
    ESR_HIP(hipEventDestroy(ev));  // (released when the wait has been consumed)
    ov->ready = nullptr;
  }
  return ESR_OK;
}
// The NEXT batch's lookups: on the side stream, behind everything the main stream holds now (the previous update and this
// batch's patch), on the second communicator -- they run under this batch's loss kernel, gradient exchange and update.
int overlap_prefetch(const esr_shard_group_t* const* groups, int ngroups, esr_step_overlap_t* ov, esr_stream_t stream) {
  if (!ov || !ov->next_plan) return ESR_OK;
  const esr_routing_plan_t* np = ov->next_plan;
  ESR_REQUIRE(ov->side && np->asked_counts && np->ask_counts, "sharded step (overlapped): next plan without a side stream / counts");
  hipEvent_t here = nullptr, done = nullptr;
  ESR_HIP(hipEventCreateWithFlags(&here, hipEventDisableTiming));
  ESR_HIP(hipEventRecord(here, as_stream(stream)));
  ESR_HIP(hipStreamWaitEvent(as_stream(ov->side), here, 0));
  ESR_HIP(hipEventDestroy(here));
  for (int gi = 0; gi < ngroups; ++gi) {
    const esr_shard_group_t* g = groups[gi];
    ESR_REQUIRE(g->world == 1 || ov->comm2, "sharded step (overlapped): null second communicator at world %d", g->world);
    if (int rc = esr_sharded_lookup(ov->comm2, g->world, g->tables, g->row_offsets, g->ntables, g->dtype, g->D,
                                    np->asked_rows, np->asked_counts, np->ask_counts, ov->next_served[gi], ov->next_back[gi],
                                    ov->side))
      return rc;
  }
  ESR_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  ESR_HIP(hipEventRecord(done, as_stream(ov->side)));
  ov->next_ready = done;
  return ESR_OK;
}
}  // namespace

size_t esr_sharded_triplet_step_workspace_bytes(const esr_shard_group_t* towers, const esr_routing_plan_t* plan,
                                                int64_t B) {
  int64_t n_rows = 0, n_recv = 0;
  if (!towers || !plan || B <= 0 || plan_counts("esr_sharded_triplet_step_workspace_bytes", towers, plan, &n_rows, &n_recv))
    return 0;
  return group_scratch_bytes(towers, 3 * B, n_rows, n_recv, plan->sorted_uidx != nullptr) +
         align_up((size_t)3 * B * towers->D * 4, 256) + align_up(esr_triplet_workspace_bytes(B), 256) + 1024;
}

size_t esr_sharded_step_overlap_workspace_bytes(const esr_shard_group_t* group, const esr_step_overlap_t* overlap) {
  return group && group->world >= 1 ? patch_scratch_bytes(group, overlap) : 0;
}

void esr_sharded_overlap_release(void* ready) {
  if (ready) (void)hipEventDestroy((hipEvent_t)ready);
}

int esr_sharded_triplet_step(const esr_shard_group_t* towers, const esr_routing_plan_t* plan, int64_t B,
                             float regularization, float batch_size, float lr, float eps, float* loss, void* workspace,
                             size_t workspace_bytes, esr_stream_t stream) {
  return esr_sharded_triplet_step_overlapped(towers, plan, nullptr, B, regularization, batch_size, lr, eps, loss, workspace,
                                             workspace_bytes, stream);
}

int esr_sharded_triplet_step_overlapped(const esr_shard_group_t* towers, const esr_routing_plan_t* plan,
                                        esr_step_overlap_t* ov, int64_t B, float regularization, float batch_size, float lr,
                                        float eps, float* loss, void* workspace, size_t workspace_bytes, esr_stream_t stream) {
  TraceScope trace_scope_("esr_sharded_triplet_step_overlapped");
  int64_t n_rows = 0, n_recv = 0;
  if (int rc = plan_counts("esr_sharded_triplet_step", towers, plan, &n_rows, &n_recv)) return rc;
  ESR_REQUIRE(B > 0 && loss && plan->index, "esr_sharded_triplet_step: B=%lld, or null loss / plan index", (long long)B);
  ESR_REQUIRE(towers->dtype == ESR_F32, "esr_sharded_triplet_step: f32 tables only (bf16 towers: lookup + the f32 head)");
  const size_t need = esr_sharded_triplet_step_workspace_bytes(towers, plan, B) + patch_scratch_bytes(towers, ov);
  ESR_REQUIRE(workspace && !((uintptr_t)workspace & 255) && workspace_bytes >= need,
              "esr_sharded_triplet_step: workspace %zu bytes < %zu required (or not 256-byte aligned)", workspace_bytes, need);
  const bool unique = plan->sorted_uidx != nullptr;
  ESR_REQUIRE(unique || n_rows == 3 * B, "esr_sharded_triplet_step: a per-occurrence plan of %lld rows for B=%lld",
              (long long)n_rows, (long long)B);
  StepScratch sc{(char*)workspace, workspace_bytes};
  void* back = nullptr;
  if (int rc = overlap_enter(ov, stream)) return rc;
  if (int rc = group_rows(towers, plan, ov, 0, n_rows, n_recv, sc, &back, stream)) return rc;
  if (int rc = overlap_prefetch(&towers, 1, ov, stream)) return rc;
  float* grads = (float*)sc.take((size_t)3 * B * towers->D * 4);
  const size_t tws_bytes = esr_triplet_workspace_bytes(B);
  void* tws = sc.take(tws_bytes);
  if (!grads || !tws) {
    set_error("esr_sharded_triplet_step: workspace too small");
    return ESR_EWORKSPACE;
  }
  // the loss kernel reads the rows where they landed through the plan's index.  Per-occurrence plans: every gradient row
  // is written at the row it was read from (exchange order); unique plans: per-occurrence rows, summed per distinct row
  const int32_t* ix = plan->index;
  const float* rows = (const float*)back;
  const int flags = 1 | (unique ? 0 : ESR_GRADS_AT_IDS);
  if (int rc = esr_triplet_fwd_bwd(rows, n_rows, rows, n_rows, rows, n_rows, towers->D, ix, ix + B, ix + 2 * B, B,
                                   regularization, batch_size, flags, loss, nullptr, nullptr, grads,
                                   unique ? grads + (size_t)B * towers->D : grads,
                                   unique ? grads + (size_t)2 * B * towers->D : grads, tws, tws_bytes, stream))
    return rc;
  return group_update(towers, plan, grads, 3 * B, n_rows, n_recv, unique, sc, lr, eps, stream);
}

size_t esr_sharded_glove_step_workspace_bytes(const esr_shard_group_t* emb, const esr_shard_group_t* bias,
                                              const esr_routing_plan_t* plan, int64_t B) {
  int64_t n_rows = 0, n_recv = 0;
  if (!emb || !bias || !plan || B <= 0 || plan_counts("esr_sharded_glove_step_workspace_bytes", emb, plan, &n_rows, &n_recv))
    return 0;
  const bool unique = plan->sorted_uidx != nullptr;
  return group_scratch_bytes(emb, 2 * B, n_rows, n_recv, unique) + group_scratch_bytes(bias, 2 * B, n_rows, n_recv, unique) +
         align_up((size_t)2 * B * emb->D * 4, 256) + align_up((size_t)2 * B * 4, 256) +
         align_up(esr_glove_workspace_bytes(B), 256) + 1024;
}

int esr_sharded_glove_step(const esr_shard_group_t* emb, const esr_shard_group_t* bias, const esr_routing_plan_t* plan,
                           const float* target, int64_t B, int mode, float lr, float eps, float* loss, void* workspace,
                           size_t workspace_bytes, esr_stream_t stream) {
  return esr_sharded_glove_step_overlapped(emb, bias, plan, nullptr, target, B, mode, lr, eps, loss, workspace,
                                           workspace_bytes, stream);
}

int esr_sharded_glove_step_overlapped(const esr_shard_group_t* emb, const esr_shard_group_t* bias,
                                      const esr_routing_plan_t* plan, esr_step_overlap_t* ov, const float* target, int64_t B,
                                      int mode, float lr, float eps, float* loss, void* workspace, size_t workspace_bytes,
                                      esr_stream_t stream) {
  TraceScope trace_scope_("esr_sharded_glove_step_overlapped");
  int64_t n_rows = 0, n_recv = 0;
  if (int rc = plan_counts("esr_sharded_glove_step", emb, plan, &n_rows, &n_recv)) return rc;
  ESR_REQUIRE(bias && bias->world == emb->world && bias->D == 1 && bias->ntables == 1 && emb->ntables == 1,
              "esr_sharded_glove_step: one embedding table and one [V, 1] bias table, sharded alike");
  ESR_REQUIRE(B > 0 && loss && target && plan->index, "esr_sharded_glove_step: B=%lld, or a null pointer", (long long)B);
  ESR_REQUIRE(emb->dtype == ESR_F32 && bias->dtype == ESR_F32, "esr_sharded_glove_step: f32 tables only");
  ESR_REQUIRE(mode == ESR_GLOVE_REFERENCE || mode == ESR_GLOVE_DIAGONAL, "esr_sharded_glove_step: bad mode %d", mode);
  const size_t need = esr_sharded_glove_step_workspace_bytes(emb, bias, plan, B) + patch_scratch_bytes(emb, ov) +
                      patch_scratch_bytes(bias, ov);
  ESR_REQUIRE(workspace && !((uintptr_t)workspace & 255) && workspace_bytes >= need,
              "esr_sharded_glove_step: workspace %zu bytes < %zu required (or not 256-byte aligned)", workspace_bytes, need);
  const bool unique = plan->sorted_uidx != nullptr;
  ESR_REQUIRE(unique || n_rows == 2 * B, "esr_sharded_glove_step: a per-occurrence plan of %lld rows for B=%lld",
              (long long)n_rows, (long long)B);
  StepScratch sc{(char*)workspace, workspace_bytes};
  void *rows = nullptr, *brow = nullptr;
  if (int rc = overlap_enter(ov, stream)) return rc;
  if (int rc = group_rows(emb, plan, ov, 0, n_rows, n_recv, sc, &rows, stream)) return rc;
  if (int rc = group_rows(bias, plan, ov, 1, n_rows, n_recv, sc, &brow, stream)) return rc;
  const esr_shard_group_t* both[2] = {emb, bias};
  if (int rc = overlap_prefetch(both, 2, ov, stream)) return rc;
  float* grad_rows = (float*)sc.take((size_t)2 * B * emb->D * 4);
  float* grad_bias = (float*)sc.take((size_t)2 * B * 4);
  const size_t gws_bytes = esr_glove_workspace_bytes(B);
  void* gws = sc.take(gws_bytes);
  if (!grad_rows || !grad_bias || !gws) {
    set_error("esr_sharded_glove_step: workspace too small");
    return ESR_EWORKSPACE;
  }
  if (int rc = esr_glove_fwd_bwd((const float*)rows, (const float*)brow, n_rows, emb->D, plan->index, target, B,
                                 mode | (unique ? 0 : ESR_GRADS_AT_IDS), loss, grad_rows, grad_bias, gws, gws_bytes, stream))
    return rc;
  if (int rc = group_update(emb, plan, grad_rows, 2 * B, n_rows, n_recv, unique, sc, lr, eps, stream)) return rc;
  return group_update(bias, plan, grad_bias, 2 * B, n_rows, n_recv, unique, sc, lr, eps, stream);
}

}  // extern "C"
