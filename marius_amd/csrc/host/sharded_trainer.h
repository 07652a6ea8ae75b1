// Multi-GPU training step, one process per GPU: node table sharded by contiguous id range, rows fetched / gradients returned by
// RCCL all-to-all(v), relation tables replicated (the schedule, its staleness semantics and the CPU gloo tests of the
// control flow live in marius_amd/sharded.py: PipelineSchedule — this is the same schedule issued from C++ so that the host is not the
// bottleneck: the Python driver needs ~1.2 ms per step to issue ~60 launches, more than the GPU needs to execute them).
//
// Collectives go through torch.distributed's C++ layer (c10d::ProcessGroup), looked up by name (c10d::resolve_process_group), so nothing but
// a string crosses the Python boundary.
//
// Round 5: two forms of the exchange, same schedule.
//   exact (default)  all-to-all(v): the split sizes of a batch reach the host as ONE ordered record (marius_a2a_publish: payload, system fence,
//                    stamp; checksummed), written by the batch's preparation AHEAD steps before the exchange needs it (Slot::rec_host).
//   fixed            (MARIUS_EXCHANGE=fixed) FIXED-CAPACITY payloads: every (requester, owner) pair owns `cap` slots of each payload —
//                    marius_a2a_capacity: the batch's id capacity at world 1, slack x capacity / world otherwise — the three payloads (ids, rows,
//                    gradients) travel by equal-split all-to-alls, the counts ride in the id payload as -1 padding (exchange.hip,
//                    marius_a2a_rows_{post,wait}), the decoder scores the row payload where it landed (the batch's local indices are rewritten to
//                    payload slots), and the owner plans its segmented update when the IDS arrive, a scoring pass before the gradients.  The host
//                    reads nothing from the device.
// Measured at world 1 (profiles/r5_sharded_exchange_forms.txt): what round 4 read as "the loop blocks 0.35-0.45 ms per step for split points" is
// BACK-PRESSURE — the host runs ahead of a device that is the bottleneck, and blocks at whatever its first dependence on the device is; with the
// fixed form that dependence is gone and the loop blocks the same 0.25-0.4 ms at the slot-reuse check instead, while every exchange stage moves
// 25 % more rows (capacity 200,000 against 159,000 unique ids per batch): 0.757-0.79 ms per step against 0.696-0.73 exact.  On xGMI the padding
// would be wire bytes (slack 1.5: +50 %).  So the exact form stays the default and the fixed form is kept, tested at world 1 / 2 / 8, for
// transports or hosts where a per-batch read-back is not to be had.
#pragma once
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include "marius_host.h"

namespace marius_amd {

class ShardedTrainer {
   public:
    static constexpr int RING = 8;   // slots in flight
    static constexpr int AHEAD = 4;  // batches prepared ahead of the one being scored: their split points must be on the host when fetch needs them

    // loader: prepares this rank's batches (its node storage only supplies num_nodes = dim0_size_); shard_table / shard_state: rows
    // [rank * S, min((rank + 1) * S, num_nodes)) of the node table and its Adagrad state, S = ceil(num_nodes / world) (storage.cpp:75).
    // staleness 0: synchronous (every row a batch reads carries all earlier updates); s >= 1: the fetch of batch t + s and the gradient
    // return of batch t run on an exchange stream underneath the scoring (rows at most s updates stale, s <= AHEAD; the reference's own
    // multi-GPU trainer is the asynchronous pipeline with staleness_bound 16).  sync_interval: pipeline.gpu_sync_interval (relation tables and
    // their Adagrad state averaged every K steps; 1 = all-reduce the relation gradients every step, model.cpp:136-159).
    ShardedTrainer(shared_ptr<DataLoader> loader, shared_ptr<Model> model, Tensor shard_table, Tensor shard_state, int rank, int world, int64_t num_nodes,
                   const std::string& group_name, const std::string& side_group_name, int staleness, int sync_interval);
    ~ShardedTrainer();
    void step();
    void train_steps(int64_t n);
    void finish();  // drain the device (prefetched batches are dropped: preparation and fetch have no side effects)
    double host_seconds_ = 0;  // time spent issuing steps (diagnostic: the host must stay ahead of the GPU)
    // bytes this rank put on the wire (to other ranks) in ids / rows served / gradients returned since the counters were last cleared
    int64_t exchange_bytes_[3] = {0, 0, 0};
    int ranks() const { return pg_ ? pg_->getSize() : 1; }
    std::string backend() const { return pg_ ? pg_->getBackendName() : std::string("none"); }
    int64_t steps_ = 0;
    double phase_seconds_[6] = {0, 0, 0, 0, 0, 0};  // host time in: prepare, wait for split points, fetch, compute, update, dense
    bool fixed_capacity() const { return fixed_; }
    // records whose first read failed the checksum and were polled again (wait_prepared): 0 unless the platform reorders the kernel's stores
    int64_t torn_reads_ = 0;
    std::string describe_state() const;  // every slot's stamp and events, every stream's status: what a deadline failure reports
    int64_t pair_capacity() const { return cap_; }  // rows per (requester, owner) pair and payload (fixed-capacity exchange); 0 before the first batch
    // rows this rank actually asked of [0] / was asked for by [1] OTHER ranks (from the published header): the useful part of exchange_bytes_
    // ([1] needs the receive counts: exact form only; the fixed-capacity form exchanges no counts and leaves it 0)
    int64_t useful_rows_[2] = {0, 0};
    // device time from the first to the last operation of a stage on its stream (HIP events; collected when a slot is reused, RING steps
    // later): prepare (prep stream), fetch and update (exchange stream), compute (main stream).  A stage's span includes the time its
    // kernels wait for CUs held by the other streams: span >> the stage's own work means that stream is starved.
    double span_ms_[7] = {0, 0, 0, 0, 0, 0, 0};  // 4..6: fetch split into ids all-to-all / owner gather / rows all-to-all
    int64_t span_n_[7] = {0, 0, 0, 0, 0, 0, 0};
    bool spans_ = false;
    void enable_spans(bool on) { spans_ = on; }
    void reset_counters() {  // after a warm-up: its first steps carry one-time costs (allocations, code-object loads, collective setup)
        host_seconds_ = 0;
        for (auto& p : phase_seconds_) p = 0;
        steps_ = 0;
        exchange_bytes_[0] = exchange_bytes_[1] = exchange_bytes_[2] = 0;
        useful_rows_[0] = useful_rows_[1] = 0;
        fine_seconds_[0] = fine_seconds_[1] = 0;
    }

   private:
    struct Slot {
        shared_ptr<Batch> batch;
        Tensor offs_dev;                    // [world + 1] split points of the batch's ascending unique ids by owner
        Tensor cnt_send_dev, cnt_recv_dev;  // [world] rows this rank asks of every owner / is asked for by every requester
        void* ready = nullptr;       // prep stream: batch prepared, header published
        void* fetched = nullptr;     // rows of this batch have arrived
        void* computed = nullptr;    // per-row gradients complete
        void* free_ = nullptr;       // owners applied the gradients: every buffer of the slot is reusable
        void* span_b[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // timing events (stage begin / end), see span_ms_
        void* span_e[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        bool span_live[7] = {false, false, false, false, false, false, false};
        bool used = false;
        std::vector<int64_t> send_counts, recv_counts;
        int64_t U = 0, nrecv = 0;
        Tensor emb, grad, local_ids;
        Tensor row_bound;  // device float[1] >= every |x| of the rows this slot's batches gathered (Batch::row_bound_; max'ed in on the exchange stream)
        // The batch's exchange header on the host: pinned int64 [2 world + 4] = stamp | split points | receive counts | overflow | checksum, written by
        // the LAST kernel of the preparation (marius_a2a_publish: payload stores, system fence, then the stamp = batch index + 1 with release
        // semantics).  The host polls the stamp (acquire) instead of hipEventSynchronize(ready) — on this runtime an event wait on a stream with
        // younger work queued behind the event returned only when that younger work had finished (profiles/r5_sharded_timeline_before.txt) —
        // copies the record to `hdr` and accepts it only if the checksum matches.
        Tensor rec_host;
        std::vector<int64_t> hdr;   // the accepted copy
        int64_t hdr_stamp = 0;      // stamp of the record in hdr (== stamp_value once wait_prepared returned)
        int64_t stamp_value = 0;
        int64_t batch_index = -1;
        // fixed-capacity exchange
        Tensor req_send, place;             // [world * cap] ids asked of every owner (-1 padded), [L] slot of unique index u
        Tensor slot_of_occ, edges_slot;     // [L] payload slot of every occurrence, [B, cols] Batch::edges_ in slot terms: the decoder reads the row payload in place
        Tensor overflow_dev;  // int32: some owner was asked for more than cap rows — all-reduced (MAX) over the ranks on the preparation stream and
                              // published in the header, so EVERY rank refuses the batch before anything of it is exchanged (fetch_fixed)
        Tensor grad_send;                   // [world * cap, d] per-row gradients in the owners' slot order
        Tensor r_uniq, r_inverse, r_perm, r_seg, r_count, r_plan;  // owner side: merged runs of the received ids + their segment plan
    };
    shared_ptr<DataLoader> loader_;
    shared_ptr<Model> model_;
    Tensor table_, state_;
    int rank_, world_, staleness_, sync_interval_;
    int64_t num_nodes_, S_, lo_;
    int d_;
    c10::intrusive_ptr<c10d::ProcessGroup> pg_, side_pg_;
    void* main_stream_ = nullptr;  // forward / loss / backward: a high-priority stream, so that its workgroups win CU slots over the side streams
    void* prep_stream_ = nullptr;
    void* xchg_stream_ = nullptr;  // == main stream when staleness == 0
    Slot slots_[RING];
    int64_t next_prepared_ = 0, next_fetched_ = 0, step_index_ = 0;
    // grow-only buffers (per-step sizes vary with the number of unique ids)
    Tensor buf_req_, buf_rows_, buf_recv_grad_, emb_[RING], grad_[RING], local_[RING];
    // owner-side dedupe of the received ids
    Tensor r_ws_, r_carry_;
    int64_t r_cap_ = 0;
    // fixed-capacity exchange: shared staging (each is produced and consumed inside one stage on the exchange stream)
    bool fixed_ = false;
    double slack_ = 1.5;
    int64_t L_ = 0, cap_ = 0, ncap_ = 0;  // id capacity of a batch, slots per pair, slots per payload (world * cap)
    Tensor req_recv_, rows_send_, grad_recv_;
    std::vector<int64_t> run_offsets_;  // q * cap: the `world` runs of a received id payload
    void setup_fixed(Slot& s, int64_t L);
    void fetch_fixed(int64_t t);
    void update_fixed(int64_t t);
    void retire(Slot& s);  // a slot is about to be reused: its preparation finished long ago — useful-row accounting
    void wait_prepared(Slot& s);  // the slot's header is on the host, checksum-verified, in s.hdr
    void check_splits(const Slot& s, const char* what, const Tensor& in, const std::vector<int64_t>& in_split, const Tensor& out, const std::vector<int64_t>& out_split) const;
    void refuse_overflow(const Slot& s) const;
    double deadline_s_ = 60.0;          // MARIUS_SHARDED_DEADLINE_S
    bool failed_ = false;               // a step threw: the pipeline state is half-advanced, every later call refuses
    double fine_seconds_[2] = {0, 0};   // MARIUS_SHARDED_FINE=1: host seconds inside prepareBatch / the rest of prepare()
    // what the constructor changed on objects it does not own, put back by the destructor
    int saved_free_cus_ = 0;
    bool saved_prefetch_ = true, saved_full_batches_ = false, saved_plan_ahead_ = false, saved_run_ahead_ = true;
    int saved_pool_requests_ = 0;

    Slot& slot(int64_t t) { return slots_[t % RING]; }
    Tensor view(Tensor& buf, int64_t n, std::vector<int64_t> tail, torch::ScalarType dtype);
    void prepare(int64_t t);
    void prepare_through(int64_t t);
    void fetch(int64_t t);
    void fetch_through(int64_t t);
    void compute(int64_t t);
    void update(int64_t t);
    void dense(int64_t t);
    void plan_local(Slot& s);
    void apply_local(Slot& s, const Tensor& grads);
    Tensor a2a(const Slot& s, const char* what, const Tensor& in, const std::vector<int64_t>& send_counts, const std::vector<int64_t>& recv_counts, Tensor out);
    void prime();
    void span_begin(Slot& s, int stage, void* stream);
    void span_end(Slot& s, int stage, void* stream);
    void span_collect(Slot& s);
};

// The collective calls of ShardedTrainer in isolation, on tensors of any device the group supports (the CPU gloo tests run this with
// world size 2): count exchange (equal split), all-to-all(v) of `send` rows with those counts, all-reduce.  Returns {recv_counts, recv, sum}.
std::vector<Tensor> c10d_exchange_selftest(const std::string& group_name, Tensor send, std::vector<int64_t> send_counts, Tensor to_reduce);

}  // namespace marius_amd
