/*
 * marius_hip.h — C-ABI of libmarius_hip.so: the MI355X (gfx950) kernels behind Marius's link-prediction
 * training hot path.  Plain pointers and sizes only; no torch types.  All device pointers are HBM addresses
 * on the current HIP device; `stream` is a hipStream_t (NULL = default stream).  Every entry point returns
 * MARIUS_OK (0) or an error code; marius_hip_last_error() gives the message.  Nothing here allocates or frees
 * device memory and nothing synchronises the device: calls only enqueue work on `stream`.
 *
 * The reference (marius-team/marius) has NO FFI boundary for this path: its operator API is C++ classes over
 * torch::Tensor.  Each entry point below names the reference function(s) it replaces (paths relative to
 * /root/reference/src/cpp); the C++ host classes in marius_amd/csrc/host keep the reference's class/method
 * names and call these.  INTEGRATION.md shows the binding a reference maintainer would add.
 */
#ifndef MARIUS_HIP_H
#define MARIUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* marius_stream_t; /* hipStream_t */

enum { MARIUS_OK = 0, MARIUS_ERR_INVALID = 1, MARIUS_ERR_HIP = 2, MARIUS_ERR_UNSUPPORTED = 3 };

/* RelationOperator subclasses: include/nn/decoders/edge/relation_operators.h:11-43 */
enum { MARIUS_OP_HADAMARD = 0, MARIUS_OP_COMPLEX_HADAMARD = 1, MARIUS_OP_TRANSLATION = 2, MARIUS_OP_NOOP = 3 };
/* Comparator subclasses: include/nn/decoders/edge/comparators.h:13-35 */
enum { MARIUS_CMP_DOT = 0, MARIUS_CMP_L2 = 1, MARIUS_CMP_COSINE = 2 };
/* LossReduction: include/configuration/options.h */
enum { MARIUS_REDUCE_SUM = 0, MARIUS_REDUCE_MEAN = 1 };
/* LossFunction subclasses of src/nn/loss.cpp:50-187 on (pos [B'], neg [B', N]) scores.  CROSS_ENTROPY over [pos, neg...] with label 0
 * (loss.cpp:89-103) is the same function as SOFTMAX_CE (log-sum-exp over the same 1 + N scores) and shares its kernels. */
enum {
    MARIUS_LOSS_SOFTMAX_CE = 0,
    MARIUS_LOSS_RANKING = 1,           /* margin_ranking_loss(neg, pos[:,None], -1, margin): max(0, neg - pos + margin), mean over B' N */
    MARIUS_LOSS_CROSS_ENTROPY = 2,
    MARIUS_LOSS_BCE_AFTER_SIGMOID = 3, /* binary_cross_entropy(sigmoid([pos, neg.flatten]), [1.., 0..]), mean over B' (1 + N)          */
    MARIUS_LOSS_BCE_WITH_LOGITS = 4,
    MARIUS_LOSS_MSE = 5,
    MARIUS_LOSS_SOFTPLUS = 6           /* softplus(-(2 y - 1) x)                                                                       */
};

/* Bumped whenever a struct of this header changes size or meaning, or an entry point changes its signature (3: marius_lp_desc.flags /
 * reserved_, marius_lp_layout.adjrec / negrec / fpart / flash, planned segment update, zero-initialised sort workspace; 4: MARIUS_LP_KEEP_DADJ,
 * layout.dadj doubled on the flash path; 5: marius_lp_desc.absmax, marius_table_absmax, the *_tracked update entry points; 6:
 * marius_segment_update, marius_segment_adagrad_scatter_group, marius_hip_struct_bytes(2); 7: marius_lp_desc.absmax_rel,
 * marius_table_absmax_counted; marius_lp_layout lost the operand planes of the removed bf16x6 kernels and the stream-K partials; 8:
 * marius_lp_desc.upd_*, marius_segment_update.fused_below, marius_lp_fuses_endpoint_update, marius_segment_plan_occ_single; 9: the
 * fixed-capacity exchange entry points marius_a2a_capacity / marius_a2a_rows_post / marius_a2a_rows_wait, negative ids = padding slots in
 * marius_merge_unique_runs / marius_segment_plan; 10: marius_a2a_publish / marius_a2a_record_* / marius_owner_offsets_counts, marius_layer_post_hook*, marius_prepare_maps; 11: marius_lp_layout.flash_cfg, marius_prepare_maps_preferred, the neighbour-sampling / GraphSage entry points marius_nbr_* and marius_segment_gather_sum).  Every binder
 * compares the value it was built against with what the loaded library returns and refuses to run on a mismatch: marius_amd/hip.py lib(),
 * the host module's init (bindings.cpp), and the plug-in recipe of INTEGRATION.md. */
#define MARIUS_HIP_ABI_VERSION 11
int marius_hip_abi_version(void);
/* The MARIUS_* environment switches of the kernel library (test / A-B selectors; a production run sets none) are read ONCE, when the
 * library is loaded.  A process that changes one afterwards (the parity tests do, to reach a non-default kernel) calls this to re-read them. */
int marius_config_reload(void);
/* sizeof(marius_lp_desc) / sizeof(marius_lp_layout) as the library was compiled: a second line of defence for ctypes mirrors */
int marius_hip_struct_bytes(int which /* 0: marius_lp_desc, 1: marius_lp_layout, 2: marius_segment_update */);
const char* marius_hip_last_error(void);

/* Optional HIP-event profiler (bench.py's roofline): when enabled, the library records hipEvents on the launch stream
 * around its main kernels; marius_profile_read waits for them and returns the accumulated kernel time.
 * marius_profile_enable(on): 0 = off, 1 = every instrumented kernel, 2 + id = only kernel `id` (two events per launch of that kernel
 * and nothing else: each event pair costs a few microseconds of stream time, which matters inside a timed region). */
int marius_profile_enable(int on);
int marius_profile_reset(void);
int marius_profile_kernel_count(void);
const char* marius_profile_kernel_name(int id);
int marius_profile_read(int id, double* total_ms, int64_t* launches);
/* debug only: device buffer of >= 256*2*64 uint64 that receives per-phase cycle stamps of the score kernel (NULL = off) */
int marius_debug_set_timeline(unsigned long long* buf);

/* ------------------------------------------------------------------------------------------------ storage */

/* out[i, 0:d] = table[ids[i], 0:d]            replaces InMemory::indexRead  src/storage/storage.cpp:606-649
 *                                             (PartitionBuffer::indexRead    src/storage/buffer.cpp:441-455)   */
int marius_gather_rows(const float* table, int64_t table_ld, const int64_t* ids, int64_t n, int32_t d,
                       float* out, int64_t out_ld, marius_stream_t stream);

/* The same gather for a capacity-sized id list whose valid length lives on the device (the unique count of map_tensors, never read
 * back by the fused training step): rows [*num_rows_dev, capacity) of `out` are left untouched. */
int marius_gather_rows_counted(const float* table, int64_t table_ld, const int64_t* ids, int64_t capacity, const int64_t* num_rows_dev, int32_t d,
                               float* out, int64_t out_ld, marius_stream_t stream);

/* both node tables with one id list (embeddings + Adagrad state): DataLoader::loadGPUParameters
 * src/data/dataloader.cpp:529-548 */
int marius_gather_rows2(const float* table_a, const float* table_b, int64_t table_ld, const int64_t* ids, int64_t n,
                        int32_t d, float* out_a, float* out_b, int64_t out_ld, marius_stream_t stream);

/* table[ids[i], 0:d] += delta[i, 0:d]; ids UNIQUE (reference contract)   replaces InMemory::indexAdd
 * src/storage/storage.cpp:651-673 (PartitionBuffer::indexAdd src/storage/buffer.cpp:460-480). No atomics. */
int marius_scatter_add_rows(float* table, int64_t table_ld, const int64_t* ids, int64_t n, int32_t d,
                            const float* delta, int64_t delta_ld, marius_stream_t stream);

/* ds = g*g; state += ds; dw = -lr * g / (sqrt(state) + eps)     replaces Batch::accumulateGradients
 * src/data/batch.cpp:62-79 (eps = 1e-10 there). n = number of floats. */
int marius_adagrad_rule(const float* grad, float* state, float* dw, float* ds, int64_t n, float lr, float eps,
                        marius_stream_t stream);

/* dense AdagradOptimizer::step  src/nn/optim.cpp:114-145:  sum += g*g; w -= lr * g / (sqrt(sum) + eps) */
int marius_dense_adagrad_step(float* param, float* state_sum, const float* grad, int64_t n, float lr, float eps,
                              float weight_decay, marius_stream_t stream);

/* dense AdamOptimizer::step  src/nn/optim.cpp:186-232 (the reference's fb15k_237 example trains the relation tables with it):
 * exp_avg = exp_avg*b1 + g*(1-b1); exp_avg_sq = exp_avg_sq*b2 + (1-b2)*g*g; denom = sqrt(exp_avg_sq or its running max)/sqrt(1-b2^t) + eps;
 * w -= lr/(1-b1^t) * exp_avg/denom, t = num_steps + 1.  max_exp_avg_sq = NULL unless amsgrad. */
int marius_dense_adam_step(float* param, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, const float* grad, int64_t n, float lr,
                           float beta1, float beta2, float eps, float weight_decay, int64_t num_steps, marius_stream_t stream);

/* ------------------------------------------------------------------------------------------------ encoder (embedding-only)
 * Layer::post_hook  src/nn/layers/layer.cpp:9-16, applied by GeneralEncoder::forward (src/nn/encoders/encoder.cpp:195-257) to the [n, d] rows the
 * embedding layer hands on (EmbeddingLayer::forward is a column view, embedding.cpp:17):  out = act(x + bias), bias [d] or NULL (config bias:
 * false), act = apply_activation (src/nn/activation.cpp:7-21).  out may alias x. */
enum { MARIUS_ACT_NONE = 0, MARIUS_ACT_RELU = 1, MARIUS_ACT_SIGMOID = 2 };
int marius_layer_post_hook(const float* x, int64_t x_ld, const float* bias, int32_t activation, int64_t n, int32_t d, float* out, int64_t out_ld,
                           marius_stream_t stream);
/* its backward (what autograd derives for the reference): gx = gy * act'(.) from the forward OUTPUT y (relu: [y > 0]; sigmoid: y (1 - y); y may
 * be NULL for NONE), gx may alias gy; bias_grad [d] (NULL: no bias) = column sums of gx, computed in two deterministic stages through
 * `workspace` (marius_layer_post_hook_workspace_bytes(n, d) bytes): no float atomics, the same bits on every run. */
size_t marius_layer_post_hook_workspace_bytes(int64_t n, int32_t d);
int marius_layer_post_hook_backward(const float* gy, int64_t gy_ld, const float* y, int64_t y_ld, int32_t activation, int64_t n, int32_t d, float* gx,
                                    int64_t gx_ld, float* bias_grad, void* workspace, size_t workspace_bytes, marius_stream_t stream);

/* ------------------------------------------------------------------------------------------------ sampling */

/* ATen CPU generator stream (at::mt19937 as used by torch::randint / torch::randperm on CPU tensors).
 * state = 624 words + [624] = index of next word (624 => twist first).  Host helpers are plain C. */
#define MARIUS_MT_STATE_WORDS 625
void marius_mt19937_seed_host(uint32_t* state_host, uint64_t seed);            /* torch::manual_seed, src/marius.cpp:47 */
void marius_mt19937_fill_host(uint32_t* state_host, uint32_t* out, int64_t n); /* n raw 32-bit draws */
/* torch::randperm(n) on the CPU generator: DataLoader::setActiveEdges src/data/dataloader.cpp:176-182 */
int marius_mt19937_randperm_host(uint32_t* state_host, int64_t* out, int64_t n);

/* n raw tempered 32-bit draws from a DEVICE-resident state (advances it).  One workgroup; stream-ordered. */
int marius_mt19937_fill(uint32_t* state_dev, uint32_t* out_dev, int64_t n, marius_stream_t stream);

/* number of raw 32-bit words one getNegatives() call consumes */
int64_t marius_negatives_raw_words(int64_t num_nodes, int64_t B, int32_t num_chunks, int32_t num_negatives,
                                   int32_t num_deg);

/* raw words -> negative ids.   replaces CorruptNodeNegativeSampler::getNegatives src/data/samplers/negative.cpp:328-366
 * and batch_sample :7-19.  Per chunk: (N - num_deg) uniform ids `draw % num_nodes` (drawn first), then num_deg edge
 * positions `draw % B`; row layout cat({deg, uniform}).  range >= 2^28 consumes two words per draw (ATen).
 * edges: [B, edge_cols] int64; inverse != 0 -> column 0 (src) else last column (dst).
 * out_ids [C, N] int64; deg_pos [C, num_deg] int64 (sampled edge positions; may be NULL when num_deg == 0). */
int marius_sample_negatives(const uint32_t* raw, const int64_t* edges, int64_t B, int32_t edge_cols, int32_t inverse,
                            int64_t num_nodes, int32_t num_chunks, int32_t num_negatives, int32_t num_deg,
                            int64_t* out_ids, int64_t* deg_pos, marius_stream_t stream);

/* deg_negative_local_filter  src/data/samplers/negative.cpp:21-39 (LocalFilterMode::DEG, training):
 * out [C * num_deg, 2] int64, row (c * num_deg + k) = (e, k) if e = deg_pos[c][k] lies in chunk c (e / ceil(B / C) == c), else (-1, -1).
 * Rows with -1 are ignored by marius_lp_forward's filter; dropping them yields the reference's tensor in its order. */
int marius_deg_filter(const int64_t* deg_pos, int32_t num_chunks, int32_t num_deg, int64_t B, int64_t* out, marius_stream_t stream);

/* Filtered evaluation, global filter: compute_filter_corruption  src/data/samplers/negative.cpp:50-205 (CPU) / :212-293 (GPU, libtorch ops).
 * sorted_edges [n_sorted, cols] int64 = every known edge sorted by source (inverse = 0) or by destination (inverse = 1)
 * (MariusGraph::sortAllEdges, src/data/graph.cpp:233-236); edges [B, cols] the batch (global ids).
 * _offsets: counts[B] scratch, offsets[B + 1] = exclusive prefix of the number of true edges that share each batch edge's uncorrupted
 * endpoint and relation; the caller reads offsets[B] (= F) to size the output.  _emit: filter [F, 2] = (edge id, corrupted node),
 * ordered by edge id then by position in sorted_edges (the reference's order); apply_score_filter (negative.cpp:306-311) is
 * marius_lp_forward's dst_filter / src_filter. */
int marius_true_edge_filter_offsets(const int64_t* sorted_edges, int64_t n_sorted, int32_t cols, int32_t inverse, const int64_t* edges, int64_t B,
                                    int64_t* counts, int64_t* offsets, marius_stream_t stream);
int marius_true_edge_filter_emit(const int64_t* sorted_edges, int64_t n_sorted, int32_t cols, int32_t inverse, const int64_t* edges, int64_t B,
                                 const int64_t* offsets, int64_t* filter, marius_stream_t stream);

/* out[i, :] = edges[perm[start + i], :] cast to int64      replaces active_edges_ index_select + RandomEdgeSampler::getEdges
 * src/data/dataloader.cpp:180-182, src/data/samplers/edge.cpp:12-14.  edges_in is int32 or int64 ([E, cols]). */
int marius_select_edges(const void* edges_in, int32_t in_is_int64, int32_t cols, const int64_t* perm, int64_t start,
                        int64_t B, int64_t* out, marius_stream_t stream);

/* all_ids = cat({src, dst, src_neg.flatten(), dst_neg.flatten()})   DataLoader::edgeSample src/data/dataloader.cpp:400-409
 * (src_neg / dst_neg may be NULL).  out has 2B + (#neg tensors) * CN entries. */
int marius_assemble_ids(const int64_t* edges, int64_t B, int32_t edge_cols, const int64_t* src_neg, const int64_t* dst_neg,
                        int64_t CN, int64_t* out, marius_stream_t stream);

/* batch->edges_ = stack({src_mapping, rel, dst_mapping}) with the inverse of marius_sort_unique over all_ids
 * src/data/dataloader.cpp:460-466.  (The negative mappings are the views inverse[2B ..] — no kernel needed.) */
int marius_remap_edges(const int64_t* edges, const int64_t* inverse, int64_t B, int32_t edge_cols, int64_t* out,
                       marius_stream_t stream);

/* ------------------------------------------------------------------------------------------------ unique map */

/* Workspace contract: the first 256 bytes are the sort's control block (a hand-shake word and the tile counters its launches draw their
 * tile ids from).  They must be ZERO the first time a workspace is used — allocate it zero-filled — and every call leaves them zero again, so
 * a workspace can be reused for any n up to the one it was sized for, and a captured call can be replayed.  One workspace serves one
 * stream at a time. */
size_t marius_sort_unique_workspace_bytes(int64_t n);
/* replaces map_tensors src/common/util.cpp:180-205 (_unique2 sorted + inverse).
 * ids[n] int64 (>= 0) -> uniq[<=n] ascending, inverse[n] (index into uniq per input position),
 * perm[n] (input position of the k-th smallest id; stable), seg_offsets[<=n+1] (run starts in sorted order),
 * *num_unique_dev (device int64).  uniq[U..n) is zero-filled.  key_bits = number of significant id bits (<= 63). */
int marius_sort_unique(const int64_t* ids, int64_t n, int32_t key_bits, int64_t* uniq, int64_t* inverse, int32_t* perm,
                       int32_t* seg_offsets, int64_t* num_unique_dev, void* workspace, size_t workspace_bytes,
                       marius_stream_t stream);

/* Same outputs as marius_sort_unique for an input that is the concatenation of num_runs (<= 64) strictly ascending runs
 * (run q = ids[run_offsets_host[q] .. run_offsets_host[q+1]); the owner side of the sharded exchange receives one such run per sender):
 * the sorted position of every element comes from num_runs - 1 binary searches instead of radix passes; ties keep input order (stable).
 * run_offsets_host is a HOST array of num_runs + 1 offsets.  Workspace as for marius_sort_unique.
 * A run may open with any number of -1 entries (the unused slots of a fixed-capacity exchange block, marius_a2a_rows_post): they merge
 * into one leading segment with unique id -1, which marius_segment_plan marks dead. */
int marius_merge_unique_runs(const int64_t* ids, int64_t n, const int64_t* run_offsets_host, int32_t num_runs, int64_t* uniq, int64_t* inverse,
                             int32_t* perm, int32_t* seg_offsets, int64_t* num_unique_dev, void* workspace, size_t workspace_bytes,
                             marius_stream_t stream);

/* The whole map chain of one batch in ONE persistent launch (round 6): for up to two id lists at once — the batch's node ids and its relation
 * ids — assemble the ids (DataLoader::edgeSample src/data/dataloader.cpp:400-409: cat(src, dst, src_neg, dst_neg); or one column of the edges),
 * map_tensors (src/common/util.cpp:180-205: marius_sort_unique's outputs, bit for bit), the batch's edges in batch-local ids
 * (marius_remap_edges, dataloader.cpp:460-466) and the index plan of the segmented update (marius_segment_plan).  Replaces
 * marius_assemble_ids + marius_sort_unique (1 + passes + 1 launches) + marius_remap_edges + marius_segment_plan per list: 13 dependent launches
 * on the preparation stream become one.  Work items are drawn from a queue in dependency order and phases are separated by completion counters
 * with agent-scope release / acquire, so no co-residency of workgroups is assumed (a single workgroup would complete the launch).
 * A job's ids are either given (ids_in [n]) or assembled into ids_out [n] by the launch: col < 0 -> cat(edges[:, 0], edges[:, last], src_neg,
 * dst_neg) (either negative list may be NULL; n = 2 B + CN per list given), col >= 0 -> edges[:, col] (n = B).  edges_out (optional,
 * col < 0): [B, edge_cols] batch-local edges.  plan (optional): marius_segment_plan_bytes(n) bytes.  workspace: as marius_sort_unique
 * (zero-initialised once; one per job; left reusable by either form).  marius_prepare_maps_supported: 1 when every job fits the fused launch
 * (0 < n <= 2 M ids, key_bits <= 36) — otherwise call the separate entry points. */
typedef struct marius_map_job {
    const int64_t* ids_in;
    int64_t* ids_out;
    const int64_t* edges;
    const int64_t* src_neg;
    const int64_t* dst_neg;
    int64_t B, CN, n;
    int32_t edge_cols, col, key_bits, reserved_;
    int64_t* uniq;
    int64_t* inverse;
    int32_t* perm;
    int32_t* seg_offsets;
    int64_t* num_unique_dev;
    void* plan;
    int64_t* edges_out;
    void* workspace;
    size_t workspace_bytes;
} marius_map_job;
int marius_prepare_maps_supported(const marius_map_job* jobs, int32_t num_jobs);
int marius_prepare_maps_preferred(void); /* policy, not capability: 1 when MARIUS_MAPS=fused asks the DataLoader for the one-launch form (default: the separate launches) */
int marius_prepare_maps(const marius_map_job* jobs, int32_t num_jobs, marius_stream_t stream);

/* Sharded node table (partition axis of src/storage/storage.cpp:75 / buffer.cpp:340-356: shard q owns ids
 * [q * shard_rows, (q+1) * shard_rows)):  out[q] = first position in the ascending list uniq[0..*num_unique_dev) with
 * id >= q * shard_rows, q = 0..num_shards (out[num_shards] = U).  These are the all-to-all split points. */
int marius_owner_offsets(const int64_t* uniq, const int64_t* num_unique_dev, int64_t shard_rows, int32_t num_shards, int64_t* out,
                         marius_stream_t stream);

/* ---- fixed-capacity row exchange of the sharded node table (SURVEY.md 8(b) a2a_rows_{post,wait}; replaces the host-memory embeddings +
 * per-device replicas of src/cpp/src/pipeline/pipeline_gpu.cpp:23-80 and the rows that Batch::to / embeddingsToHost move over PCIe,
 * src/cpp/src/data/batch.cpp:21-60,81-103).  The transport between the two halves is ONE equal-split all-to-all per payload, issued by the
 * host on its communicator (ncclAllToAll(send, recv, cap * width, type, comm, stream) / c10d alltoall_base with empty split vectors): every
 * (requester, owner) pair owns `cap` slots, so no split size — and no device -> host read-back — is needed.  See INTEGRATION.md.
 * Neither half takes a communicator (SURVEY.md 8(b) sketched `a2a_rows_{post,wait}(comm, ...)`): by design this library issues no
 * collective and does not link RCCL — the communicator belongs to whoever hosts the training loop (c10d's ProcessGroupNCCL here, an
 * ncclComm_t per device in a binding of the reference), and the same two halves then serve any transport.
 *
 * marius_a2a_capacity: the planned maximum of rows one requester asks of one owner — max_rows (the batch's id capacity 2 B + 2 C N) for
 * world 1, else ceil(slack * max_rows / world) rounded up to 256 (slack >= 1; ids are spread evenly over the owners when node ids are
 * shuffled at preprocessing, the reference's default). */
int64_t marius_a2a_capacity(int64_t max_rows, int32_t world, double slack);
/* Requester, before the id all-to-all.  uniq: the batch's ascending unique global ids (map_tensors), owner_offsets[world + 1]: their split
 * points by owner (marius_owner_offsets).  Writes req_send[world * cap]: block q = (cap - cnt_q) entries of -1 followed by the cnt_q LOCAL
 * row ids (global - q * shard_rows) asked of owner q, ascending; place[u] = slot of unique index u in that layout (the row of u in the
 * row payload the owners send back, and the row its gradient is written to on the way out: marius_segment_sum_rows_planned's out_rows);
 * *overflow_flag |= 1 if some cnt_q > cap (the caller must not use the batch: raise the slack; the first cap rows of such an owner are
 * served, the others get place 0 — a defined slot, so that nothing indexes out of the payload before the flag is read).  Optional (n_occ > 0): slot_of_occ[i] =
 * place[inverse[i]] for the n_occ occurrences of map_tensors (inverse of marius_sort_unique) — the batch's local indices in slot terms
 * (marius_remap_edges with it gives Batch::edges_, its tail the negatives' mappings), so that the decoder reads the received row payload
 * IN PLACE instead of a compacted copy. */
int marius_a2a_rows_post(const int64_t* uniq, const int64_t* owner_offsets, int64_t shard_rows, int32_t world, int64_t cap, int64_t* req_send,
                         int64_t* place, int32_t* overflow_flag, const int64_t* inverse, int64_t n_occ, int64_t* slot_of_occ, marius_stream_t stream);
/* Requester, after the row all-to-all: *absmax = max(*absmax, max |x| over the `rows` = world * cap rows of the payload) when absmax != NULL
 * (marius_lp_desc.absmax of the batch: the rows came from every rank's shard, the requester bounds what it received; the buffer must have
 * been zero-initialised once: unused slots keep rows of earlier batches).  With emb != NULL it also leaves a copy in batch order —
 * emb[u, 0:d] = rows_recv[place[u], 0:d] for u < *num_unique_dev (what Batch::node_embeddings_ holds in the reference; capacity = rows of
 * emb) — for callers that did not rewrite their indices; the bound then covers exactly the copied rows. */
int marius_a2a_rows_wait(const float* rows_recv, int64_t recv_ld, int64_t rows, int32_t d, float* absmax, const int64_t* place,
                         const int64_t* num_unique_dev, int64_t capacity, float* emb, int64_t emb_ld, marius_stream_t stream);
/* No `comm` parameter on the pair above, on purpose (SURVEY.md 8(b) sketches a2a_rows_{post,wait}(comm, ...)): this library never calls RCCL.  The
 * collective between the two halves is issued by the HOST on whatever communicator it owns (c10d::ProcessGroupNCCL in sharded_trainer.cpp,
 * ncclAllToAll in a plug-in: INTEGRATION.md) on the same stream, so libmarius_hip.so links against nothing but the HIP runtime and works under
 * any transport the host has (the world-2 / world-8 tests run the same kernels under gloo).
 *
 * The host side of an all-to-all(v) needs a batch's split sizes.  marius_owner_offsets_counts = marius_owner_offsets that also writes
 * counts[q] = out[q + 1] - out[q] (the send counts; counts may be NULL).  marius_a2a_publish hands the batch's whole exchange header to the host
 * in ONE ordered record: record_mapped is device-mapped pinned host memory (hipHostMalloc, fine-grained: a torch pinned tensor) of
 * marius_a2a_record_words(world) = 2 world + 4 int64 words — [0] stamp, [1 .. world + 1] owner_offsets, [world + 2 .. 2 world + 1] recv_counts
 * (NULL: the send counts owner_offsets[q + 1] - owner_offsets[q], i.e. world 1 / no count exchange), [2 world + 2] *overflow_flag (NULL: 0),
 * [2 world + 3] marius_a2a_record_checksum of the words before it.  One single-wave kernel stores the payload, fences at system scope and
 * then stores the stamp (!= 0) with release semantics; the host acquires the stamp, copies the record, and accepts it only if the checksum
 * matches (a torn or stale read is re-polled, never acted on).  Replaces the three unordered D2H copies + polled stamp of ABI 9's host. */
int marius_owner_offsets_counts(const int64_t* uniq, const int64_t* num_unique_dev, int64_t shard_rows, int32_t num_shards, int64_t* out,
                                int64_t* counts, marius_stream_t stream);
int32_t marius_a2a_record_words(int32_t world);
uint64_t marius_a2a_record_checksum(const int64_t* record_host, int32_t world); /* host function: plain C over a host copy of the record */
int marius_a2a_publish(const int64_t* owner_offsets, const int64_t* recv_counts, const int32_t* overflow_flag, int32_t world, int64_t stamp,
                       int64_t* record_mapped, marius_stream_t stream);

/* ------------------------------------------------------------------------------------------------ decoder */

/* Descriptor of one CORRUPT_NODE link-prediction batch in batch-local ids (after map_tensors):
 * everything Model::forward_lp / train_batch (src/nn/model.cpp:252-333) reads. */
typedef struct marius_lp_desc {
    int32_t relop;        /* MARIUS_OP_*  */
    int32_t cmp;          /* MARIUS_CMP_* */
    int32_t d;            /* embedding dim */
    int32_t edge_cols;    /* 3 = (src, rel, dst), 2 = (src, dst) */
    int64_t B;            /* positives */
    int32_t C;            /* num_chunks */
    int32_t N;            /* negatives per chunk */
    int32_t use_inverse;  /* decoder->use_inverse_relations_ (needs edge_cols == 3) */
    int32_t reduction;    /* MARIUS_REDUCE_* */
    const float* emb;     /* [U, emb_ld] batch node embeddings (Batch::node_embeddings_) */
    int64_t emb_ld;
    int64_t U;            /* rows in emb (upper bound is fine; only used for validation) */
    const int64_t* edges; /* [B, edge_cols] batch-local (Batch::edges_) */
    const int64_t* dst_neg; /* [C, N] batch-local (Batch::dst_neg_indices_mapping_) */
    const int64_t* src_neg; /* [C, N] or NULL (Batch::src_neg_indices_mapping_) */
    const float* rel;     /* [R, rel_ld] relation_embeddings (EdgeDecoder::relations_) or NULL */
    const float* inv_rel; /* [R, rel_ld] inverse_relation_embeddings or NULL */
    int64_t rel_ld;
    int64_t R;
    const int64_t* dst_filter; /* [n_dst_filter, 2] (edge, negative column) -> score -1e9, or NULL */
    int64_t n_dst_filter;
    const int64_t* src_filter;
    int64_t n_src_filter;
    int32_t loss;         /* MARIUS_LOSS_* (0 = SoftmaxCrossEntropy: a zero-initialised descriptor keeps its old meaning) */
    float margin;         /* RankingLoss margin (loss.h:43-55) */
    int32_t flags;        /* MARIUS_LP_* (0 = the API contract: adj / pos / neg are all materialised) */
    int32_t free_cus;     /* flash path: compute units the persistent matrix launches of this batch leave WITHOUT a workgroup, for kernels the caller
                           * runs beside them on other streams (the sharded trainer's row exchange and batch preparation); 0 = only what the
                           * workgroup-count rule leaves anyway.  Results do not depend on it. */
    /* Optional, flash path only: DEVICE float[2] = { bound on |entries of the node table the batch rows come from|, bound on |entries of the
     * relation tables| } (any upper bound; marius_table_absmax computes one, the *_tracked Adagrad entry points keep it current).  When
     * given, the operand records hold fp16 halves of power-of-two scaled rows (22 significand bits per operand) instead of bf16 halves
     * (16 bits): same speed, split error 3 2^-24 instead of 3 2^-18 of sum|a_k b_k|.  NULL: bf16 records.  The kernels read the bounds on
     * the device when they run: no host read-back, and the values must not change between marius_lp_forward and marius_lp_backward. */
    const float* absmax;
    /* Optional: DEVICE float[1], the relation-table bound when it does not live behind the node bound (then absmax is float[1] too).  A
     * caller whose batch rows are a gathered copy (sharded table, partition buffer, Model::train_batch) bounds the rows it gathered —
     * marius_table_absmax over the copy — while the relation bound belongs to the model.  NULL: absmax[1]. */
    const float* absmax_rel;
    /* Optional, training only (MARIUS_LP_TRAIN_ONLY) and only when `emb` IS the node table (edges / negatives hold table rows): the edge backward
     * applies the sparse Adagrad step (batch.cpp:67-69 op order, the arithmetic of marius_segment_adagrad_scatter bit for bit) of every endpoint
     * occurrence whose node occurs ONCE in the batch straight to its table row — the row, its gradient and nothing else of that node are in the
     * half-wave's registers at that point — instead of writing the gradient to `gocc` for the segment update to read back: 1,200 bytes of
     * traffic less per such occurrence.  upd_occ_single: DEVICE uint8 per occurrence (marius_segment_plan_occ_single), upd_state: the table's
     * Adagrad state (row pitch emb_ld), upd_absmax: optional magnitude tracking as in the *_tracked entry points.  The caller's segment update
     * must then leave those rows alone: marius_segment_update.fused_below = 2 B.  Honoured only where marius_lp_fuses_endpoint_update() says so
     * (the 16-byte-row kernels: d % 4 == 0, d <= 128, packed rows); ask before relying on it.  NULL upd_occ_single / upd_state: off.
     * CALLER'S CONTRACT (the library cannot check it): `emb` is the MUTABLE node table itself, U its row count, `edges` hold table row ids, and
     * upd_state is that table's optimizer state with the SAME row pitch emb_ld — the backward writes w and s through these pointers.  Binding a
     * gathered [U, d] copy, or a state of another pitch, corrupts memory.  After marius_lp_backward the `gocc` rows of the occurrences
     * flagged in upd_occ_single below 2 B are UNDEFINED (never written): only a segment update carrying fused_below = 2 B may consume gocc. */
    const uint8_t* upd_occ_single;
    float* upd_state;
    float* upd_absmax;
    float upd_lr, upd_eps;
} marius_lp_desc;
/* 1: marius_lp_backward on this descriptor will apply the endpoint singletons' update itself (see upd_occ_single), 0: it will not */
int marius_lp_fuses_endpoint_update(const marius_lp_desc* desc);

/* marius_lp_desc.flags */
enum {
    /* The caller trains and never reads `neg` (Model::train_batch, model.cpp:290-333, as opposed to forward_lp's return value):
     * the library may run the flash-style path (lp_flash.hip) that keeps only the SoftmaxCE row statistics and recomputes score
     * tiles in the backward; layout.neg is then not allocated.  Honoured for SoftmaxCE + DotCompare without score filters and
     * d in (16, 128]; every other case silently takes the materialised-score kernels. */
    MARIUS_LP_TRAIN_ONLY = 1,
    /* with MARIUS_LP_TRAIN_ONLY: additionally store the recomputed scores into layout.neg (parity tests of the split arithmetic) */
    MARIUS_LP_STORE_SCORES = 2,
    /* flash path: layout.dadj must hold dL/dadj after marius_lp_backward (parity tests read it).  Without the flag the two blocks behind
     * layout.dadj[0] hold the unnormalised partials of the forward sweep and the edge backward combines them on the fly. */
    MARIUS_LP_KEEP_DADJ = 4
};

/* Workspace layout (all offsets in BYTES from the workspace base; dir 0 = (src,rel)->dst "rhs", dir 1 = inverse "lhs").
 * Bp = C * ceil(B / C) (pad_and_reshape, comparators.cpp:7-20); n_ld = N rounded up to 4. */
typedef struct marius_lp_layout {
    int64_t Bp, n_ld, d_ld;
    size_t total_bytes;
    size_t adj[2];    /* [Bp, d_ld]  op(src, rel) rows (zero rows for i >= B); NOT written on the flash path (flash != 0 and
                       *              no MARIUS_LP_STORE_SCORES): adj then exists as operand records only (adjrec)            */
    size_t pos[2];    /* [Bp]        positive scores (zero padded)  -> forward_lp pos / inv_pos                  */
    size_t neg[2];    /* [Bp, n_ld]  negative scores                -> forward_lp neg / inv_neg                  */
    size_t lse[2];    /* [Bp]        per-row scalar of the loss: SoftmaxCE log(e^pos + sum_j e^neg); Ranking pos - margin        */
    size_t rowloss[2];/* [Bp]        per-row loss                                                                */
    size_t loss;      /* [4] floats: total, dir0, dir1, unused                                                   */
    size_t dadj[2];   /* [Bp, d_ld]  dL/d adj (negative part); flash path: see MARIUS_LP_KEEP_DADJ                       */
    size_t gocc;      /* [2B + 2CN, d] occurrence gradients in map_tensors order (src, dst, src_neg, dst_neg)    */
    size_t grel[2];   /* [B, d]      per-edge relation gradients (dir 0 -> relations_, dir 1 -> inverse)         */
    size_t aux;       /* scratch (row norms etc.)                                                                */
    size_t lsepart;   /* [groups][ndir][Bp][2] partial (max, sum exp) of the score epilogue (fused SoftmaxCE)                */
    size_t dpos[2];   /* [Bp]        dL/d pos, written by marius_lp_loss, read by the edge backward                               */
    size_t vlog;      /* [ndir][Bp, n_ld] log(dL/dneg / scale) for the non-negative-gradient losses (0 = not allocated)            */
    size_t adjrec;    /* flash path: adj operand records      [ndir C][Bc rounded to 32][4 kp + 16 B] (hi | lo | lsec)             */
    size_t negrec;    /* flash path: negative operand records [ndir C][N  rounded to 32][4 kp + 16 B]                              */
    size_t fpart;     /* flash path: SoftmaxCE partials [2][ndir Bp] (max, sum exp)                                                  */
    int32_t flash;    /* 1 when marius_lp_plan selected the flash path for this descriptor (then neg[] = 0 unless STORE_SCORES)      */
    int32_t flash_cfg;/* flash path: the record layout the plan sized its buffers for (column chunks, folded tail), checked by every launch that uses
                       * the layout: MARIUS_ERR_INVALID instead of records written past their buffers when MARIUS_FLASH_* changed in between
                       * (marius_config_reload)                                                                                      */
} marius_lp_layout;

int marius_lp_plan(const marius_lp_desc* desc, marius_lp_layout* layout);

/* forward_lp: encoder pass-through + node_corrupt_forward (src/nn/decoders/edge/decoder_methods.cpp:57-114:
 * select_relations, apply_relation, compute_scores for both directions, pos padding) + apply_score_filter
 * (src/data/samplers/negative.cpp:306-311).  Fills adj/pos/neg. */
int marius_lp_forward(const marius_lp_desc* desc, const marius_lp_layout* layout, void* workspace, marius_stream_t stream);

/* desc->loss (SoftmaxCrossEntropy src/nn/loss.cpp:50-67 by default; the others :69-187) of both directions, summed (model.cpp:309-312).
 * Fills lse/rowloss/dpos/loss. */
int marius_lp_loss(const marius_lp_desc* desc, const marius_lp_layout* layout, void* workspace, marius_stream_t stream);

/* loss.backward() (model.cpp:324) restricted to this graph: fills gocc (per-occurrence node gradients) and grel. */
int marius_lp_backward(const marius_lp_desc* desc, const marius_lp_layout* layout, void* workspace, marius_stream_t stream);

/* SoftmaxCrossEntropy::operator() on materialised scores  src/nn/loss.cpp:50-67 (API-level; the training path uses marius_lp_loss).
 * lse[rows], rowloss[rows], loss[4] are outputs (loss[0] = reduced loss).  neg_ld must be a multiple of 4. */
int marius_softmax_ce(const float* pos, const float* neg, int64_t rows, int32_t N, int64_t neg_ld, int32_t reduction, float* lse,
                      float* rowloss, float* loss, marius_stream_t stream);

/* Any LossFunction::operator()(pos, neg, scores = true) on materialised scores (loss.cpp:50-187): scratch[2 * rows] floats, loss[4]
 * output (loss[0] = reduced loss).  neg_ld must be a multiple of 4 (the columns N..neg_ld of a row are read but ignored). */
int marius_loss_scores(int32_t loss_type, float margin, const float* pos, const float* neg, int64_t rows, int32_t N, int64_t neg_ld,
                       int32_t reduction, float* scratch, float* loss, marius_stream_t stream);

/* ranks = (neg >= pos[:,None]).sum(1) + 1   replaces LinkPredictionReporter::computeRanks src/reporting/reporting.cpp:55-57 */
int marius_compute_ranks(const float* pos, const float* neg, int64_t rows, int32_t N, int64_t neg_ld, int64_t* ranks,
                         marius_stream_t stream);

/* ------------------------------------------------------------------------------------------------ gradient reduce */

/* Segmented row sum in sorted-id order (autograd's index_select backward = index_add_, src/nn/model.cpp:324), no atomics:
 *   out[r(u), 0:d] = sum_{k in [seg_offsets[u], seg_offsets[u+1])} rows[perm[k], 0:d],  r(u) = out_rows ? out_rows[u] : u
 * perm / inverse / seg_offsets come from marius_sort_unique over the n occurrence ids (n = rows in `rows`).
 * With out_rows = the unique relation ids this builds the dense relation gradient (rows not hit stay untouched).
 * carry: scratch of marius_segment_carry_bytes(n, d) bytes. */
size_t marius_segment_carry_bytes(int64_t n, int32_t d);
int marius_segment_sum_rows(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                            const int32_t* seg_offsets, int64_t n, int32_t d, const int64_t* out_rows, float* out,
                            int64_t out_ld, void* carry, marius_stream_t stream);

/* Fused tail of the step: segmented sum + Batch::accumulateGradients (src/data/batch.cpp:62-79) + both indexAdd calls of
 * DataLoader::updateEmbeddings (src/data/dataloader.cpp:550-564):
 *   g = sum rows; ds = g*g; s = state[id] + ds; table[id] += -lr*(g/(sqrt(s)+eps)); state[id] = s   (uniq_ids ascending, unique) */
int marius_segment_adagrad_scatter(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                                   const int32_t* seg_offsets, int64_t n, int32_t d, const int64_t* uniq_ids, float* table,
                                   float* state, int64_t table_ld, float lr, float eps, void* carry, marius_stream_t stream);

/* The index work of the two calls above depends on perm / inverse / seg_offsets / uniq_ids only — known when the batch is prepared, a step
 * before its gradients exist.  marius_segment_plan precomputes it (per sorted position: occurrence row, unique index, inside-one-chunk and
 * singleton flags; per chunk: the boundary-crossing segment it owns; per unique row: table row id and the occurrence row of a singleton) on
 * whatever stream prepares batches; marius_segment_adagrad_scatter_planned then runs the same three kernels with one coalesced load where
 * the unplanned form walks a chain of dependent index loads.  Same results bit for bit.  plan: marius_segment_plan_bytes(n) bytes.
 * A unique id < 0 is padding, not a row (marius_a2a_rows_post): its positions are planned dead — the planned kernels neither load their
 * gradient rows nor write anything for them. */
size_t marius_segment_plan_bytes(int64_t n);
/* the per-occurrence singleton flags inside a plan (uint8[n]: 1 = the occurrence's id occurs once among the n): marius_lp_desc.upd_occ_single */
const uint8_t* marius_segment_plan_occ_single(const void* plan, int64_t n);
int marius_segment_plan(const int32_t* perm, const int64_t* inverse, const int32_t* seg_offsets, const int64_t* uniq_ids, int64_t n, void* plan,
                        marius_stream_t stream);
int marius_segment_sum_rows_planned(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse, const int32_t* seg_offsets,
                                    int64_t n, int32_t d, const int64_t* out_rows, float* out, int64_t out_ld, void* carry, const void* plan,
                                    marius_stream_t stream);
int marius_segment_adagrad_scatter_planned(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                                           const int32_t* seg_offsets, int64_t n, int32_t d, const int64_t* uniq_ids, float* table, float* state,
                                           int64_t table_ld, float lr, float eps, void* carry, const void* plan, marius_stream_t stream);
/* The same update (plan may be NULL: the unplanned form) that also keeps *absmax (device float) >= every |w| it writes: marius_lp_desc.absmax
 * stays a valid bound without a pass over the table.  marius_table_absmax max'es max |x| of a whole table into *absmax (zero it first). */
int marius_segment_adagrad_scatter_tracked(const float* rows, int64_t rows_ld, const int32_t* perm, const int64_t* inverse,
                                           const int32_t* seg_offsets, int64_t n, int32_t d, const int64_t* uniq_ids, float* table, float* state,
                                           int64_t table_ld, float lr, float eps, void* carry, const void* plan, float* absmax, marius_stream_t stream);

/* The update of several independent tables — one training step's node table and relation tables (dataloader.cpp:550-564 for the former,
 * the optimizer step over model parameters, model.cpp:328-331, restricted to touched rows for the latter) — as ONE pair of launches instead of
 * one pair per table.  Every job carries the arguments of marius_segment_adagrad_scatter_tracked (plan and absmax may be NULL); results per
 * table are those of the separate calls, bit for bit.  Up to four jobs, all planned with 16-byte-aligned rows of the same width class and with
 * distinct tables, states and carry buffers, run grouped (side by side); anything else falls back to the separate calls, one after the other, on
 * the same stream. */
typedef struct marius_segment_update {
    const float* rows;
    int64_t rows_ld;
    const int32_t* perm;
    const int64_t* inverse;
    const int32_t* seg_offsets;
    int64_t n;
    int32_t d;
    const int64_t* uniq_ids;
    float* table;
    float* state;
    int64_t table_ld;
    float lr, eps;
    void* carry;
    const void* plan;
    float* absmax;
    int64_t fused_below; /* > 0 (needs a plan): unique rows whose single occurrence is an occurrence row < fused_below were already updated by the
                          * producer of `rows` (marius_lp_desc.upd_occ_single: the edge backward, occurrences [0, 2 B)) and are skipped */
    /* A job with sum_out != NULL only REDUCES (marius_segment_sum_rows_planned as a job of the group; table / state / uniq_ids / lr / eps / absmax
     * are ignored, a plan is required): sum_out[r(u), 0:d] = sum of the rows of unique index u, r(u) = sum_out_rows ? sum_out_rows[u] : u.  The sharded
     * trainer's step has two relation-table updates and this reduction of the node gradients at its tail: one launch pair instead of three. */
    float* sum_out;
    int64_t sum_out_ld;
    const int64_t* sum_out_rows;
} marius_segment_update;
int marius_segment_adagrad_scatter_group(const marius_segment_update* jobs, int32_t njobs, marius_stream_t stream);
int marius_table_absmax(const float* table, int64_t rows, int64_t ld, int32_t d, float* absmax, marius_stream_t stream);
/* The same over the first *num_rows_dev rows (a device count: the unique count of map_tensors) of a capacity-sized row buffer. */
int marius_table_absmax_counted(const float* table, int64_t capacity, const int64_t* num_rows_dev, int64_t ld, int32_t d, float* absmax,
                                marius_stream_t stream);

/* ---- DENSE neighbour sampling and GraphSage aggregation (cfg4; first slice, round 6): neighbor.hip ------------------------------------------
 * The graph is the reference's MariusGraph (src/cpp/src/data/graph.cpp:16-44): an edge list sorted by destination (incoming neighbours) or by
 * source (outgoing), [E, cols] int64, with per-node first-edge offsets and degrees [num_nodes].
 *
 * marius_nbr_degrees: MariusGraph::getNeighborsForNodeIds up to the sampler (graph.cpp:128-189) + the capping of sample_uniform_gpu
 *   (src/cpp/src/data/samplers/neighbor.cpp:84-88): num[i] = degree of node_ids[i], global_offsets[i] = its first edge, capped[i] = min(num[i],
 *   max_neighbors) (max_neighbors < 0: NeighborSamplingLayer::ALL, capped = num), local_offsets = exclusive scan of capped, *total_dev = its sum.
 * marius_nbr_gather: sample_all_gpu (neighbor.cpp:9-17) / sample_uniform_gpu (neighbor.cpp:81-105): out_edges[p] = sorted_edges[global_offsets[i] + j], i the
 *   owner of output p, j = p - local_offsets[i] where the node keeps all its neighbours, rand_samples[p] % num[i] where it has more than
 *   max_neighbors (rand_samples [total]: the reference's torch::randint(max_degree, [total]) — drawn by the caller, as the MT19937 words of the
 *   negative sampler are; NULL for ALL).  total = the host's copy of *total_dev (the reference reads it with .item() at the same point).
 * marius_nbr_delta_ids: the ids the next hop expands (neighbor.cpp:515-529, device branch: bitmap + nonzero()): ascending unique ids among
 *   column 0 of in_edges and the last column of out_edges (either list may be empty) that are not in node_ids.  marks: [num_nodes] bytes, all zero
 *   on entry and on return (allocated once per graph); keys: [n_in + n_out] scratch; uniq .. workspace: as marius_sort_unique over n_in + n_out ids
 *   (uniq[0 .. *num_unique_dev) is the result).  O(batch), where the reference zero-fills and scans num_nodes entries per hop.
 * marius_nbr_positions: DENSEGraph::performMap (graph.cpp:361-398): out[t] = position in node_ids of edges[t, col]; table: [num_nodes] int64
 *   scratch (no initialisation needed: every id looked up is in node_ids).
 * marius_segment_gather_sum: GraphSageLayer::forward up to the matmuls (src/cpp/src/nn/layers/gnn/graph_sage_layer.cpp:37-90, layer_helpers.cpp:11-30):
 *   out[s] = sum over list a's segment s of rows[index_a[t]] (+ the same over list b: outgoing + incoming), rows added IN INDEX ORDER (the CPU
 *   index_add_ order: bit-identical), then mode 1 (MEAN): / where(deg != 0, deg, 1), mode 2 (GCN): (sum + self_rows[s]) / (deg + 1), deg = deg_a
 *   (+ deg_b).  offsets_*: [n] segment starts.  pre_div (optional, [rows]): every gathered row is divided by (float)pre_div[its row index] first —
 *   with index = the segment id of every occurrence of an input row (sorted by input row: marius_sort_unique's perm) this is the backward of
 *   the gather + mean.  d <= 512. */
size_t marius_nbr_workspace_bytes(int64_t n);
int marius_nbr_degrees(const int64_t* node_ids, int64_t n, const int64_t* num_neighbors_tbl, const int64_t* offsets_tbl, int64_t max_neighbors, int64_t* num,
                       int64_t* global_offsets, int64_t* capped, int64_t* local_offsets, int64_t* total_dev, void* workspace, size_t workspace_bytes,
                       marius_stream_t stream);
int marius_nbr_gather(const int64_t* sorted_edges, int32_t cols, const int64_t* num, const int64_t* global_offsets, const int64_t* local_offsets,
                      const int64_t* capped, int64_t n, const int64_t* rand_samples, int64_t total, int64_t* out_edges, marius_stream_t stream);
/* NeighborSamplingLayer::DROPOUT: sample_dropout_gpu (src/cpp/src/data/samplers/neighbor.cpp:236-253).  After marius_nbr_degrees with max_neighbors < 0 (num,
 * global_offsets, local_offsets, total): keep_rand [total] = the reference's torch::rand draw, neighbour p survives iff keep_rand[p] >= rate.
 * _offsets: keep [total] / scan [total] int64 scratch (flags and their exclusive scan), new_local_offsets [n], *total_kept_dev; workspace:
 * marius_nbr_workspace_bytes(total).  _emit: out_edges [total_kept, cols] in the reference's (masked_select) order. */
int marius_nbr_dropout_offsets(const int64_t* local_offsets, int64_t n, int64_t total, const float* keep_rand, float rate, int64_t* keep, int64_t* scan,
                               int64_t* new_local_offsets, int64_t* total_kept_dev, void* workspace, size_t workspace_bytes, marius_stream_t stream);
int marius_nbr_dropout_emit(const int64_t* sorted_edges, int32_t cols, const int64_t* global_offsets, const int64_t* local_offsets, int64_t n, const int64_t* keep,
                            const int64_t* scan, int64_t total, int64_t* out_edges, marius_stream_t stream);
int marius_nbr_delta_ids(const int64_t* in_edges, int64_t n_in, const int64_t* out_edges, int64_t n_out, int32_t cols, const int64_t* node_ids,
                         int64_t n_node_ids, int64_t num_nodes, uint8_t* marks, int64_t* keys, int64_t* uniq, int64_t* inverse, int32_t* perm,
                         int32_t* seg_offsets, int64_t* num_unique_dev, void* sort_workspace, size_t sort_workspace_bytes, marius_stream_t stream);
int marius_nbr_positions(const int64_t* node_ids, int64_t n, const int64_t* edges, int32_t cols, int32_t col, int64_t T, int64_t* table, int64_t* out,
                         marius_stream_t stream);
int marius_segment_gather_sum(const float* rows, int64_t rows_ld, int32_t d, const int64_t* index_a, const int64_t* offsets_a, int64_t T_a,
                              const int64_t* index_b, const int64_t* offsets_b, int64_t T_b, int64_t n, const int64_t* pre_div, const int64_t* deg_a,
                              const int64_t* deg_b, int32_t mode, const float* self_rows, int64_t self_ld, float* out, int64_t out_ld, marius_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MARIUS_HIP_H */
