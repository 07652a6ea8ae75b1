// ShardedTrainer: see sharded_trainer.h.  Stream order on the exchange stream:  fetch(t+1) | grads(t), update(t) | fetch(t+2) | ...
#include "sharded_trainer.h"

#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/csrc/distributed/c10d/GroupRegistry.hpp>

#include <chrono>
#include <cmath>
#include <sstream>
#include <thread>

namespace marius_amd {

#define ST_HIPCHECK(x)                                                                                             \
    do {                                                                                                            \
        hipError_t e_ = (x);                                                                                        \
        if (e_ != hipSuccess) throw MariusRuntimeException(std::string("ShardedTrainer HIP: ") + hipGetErrorString(e_)); \
    } while (0)

namespace {
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}
struct Phase {  // adds the enclosed host time to one slot of ShardedTrainer::phase_seconds_
    double& acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit Phase(double& a) : acc(a) {}
    ~Phase() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
struct Scope {  // `s` is the current torch stream of this thread while the object lives
    c10::hip::HIPStream prev;
    explicit Scope(c10::hip::HIPStream s) : prev(c10::hip::getCurrentHIPStream(s.device_index())) { c10::hip::setCurrentHIPStream(s); }
    ~Scope() { c10::hip::setCurrentHIPStream(prev); }
};
c10::hip::HIPStream& strm(void* p) { return *(c10::hip::HIPStream*)p; }
hipEvent_t new_event() {
    hipEvent_t e;
    ST_HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return e;
}
int key_bits(int64_t n) {
    int b = 1;
    while ((1ll << b) <= n && b < 63) ++b;
    return b;
}
}  // namespace

ShardedTrainer::ShardedTrainer(shared_ptr<DataLoader> loader, shared_ptr<Model> model, Tensor shard_table, Tensor shard_state, int rank, int world,
                               int64_t num_nodes, const std::string& group_name, const std::string& side_group_name, int staleness, int sync_interval)
    : loader_(loader), model_(model), table_(shard_table), state_(shard_state), rank_(rank), world_(world), staleness_(staleness),
      sync_interval_(sync_interval), num_nodes_(num_nodes) {
    if (staleness_ < 0 || staleness_ > AHEAD) throw MariusRuntimeException("ShardedTrainer: staleness must be in [0, " + std::to_string(AHEAD) + "]");
    require_device(table_, "ShardedTrainer");
    require_device(state_, "ShardedTrainer");
    S_ = (num_nodes_ + world_ - 1) / world_;  // storage.cpp:75
    lo_ = std::min<int64_t>((int64_t)rank_ * S_, num_nodes_);
    if (table_.size(0) != std::min<int64_t>(lo_ + S_, num_nodes_) - lo_) throw MariusRuntimeException("ShardedTrainer: shard has the wrong number of rows");
    d_ = (int)table_.size(1);
    saved_full_batches_ = loader_->full_batches_only_;
    saved_plan_ahead_ = loader_->plan_ahead_;
    loader_->full_batches_only_ = true;  // prepare() wraps to the next epoch when fewer than a full batch remains
    loader_->plan_ahead_ = true;         // the reductions' index work rides with the preparation (marius_segment_plan)
    pg_ = c10d::resolve_process_group(group_name);
    (void)side_group_name;  // kept in the signature: earlier builds exchanged the receive counts over a CPU (gloo) group
    if (pg_->getSize() != world_ || pg_->getRank() != rank_) throw MariusRuntimeException("ShardedTrainer: process group does not match rank / world");
    const auto dev = table_.device();
    // The compute stream is the caller's current stream (a high-priority compute stream was measured: 1.87 vs 1.40 ms per step — the side
    // streams starve and the compute stream then waits for them).  The side streams' kernels are short and sit on the critical path of the
    // NEXT step (rows must have arrived before it can be scored), while the compute stream's persistent kernels hold most of the chip: high
    // priority lets the side streams' workgroups take the first slots that free up.
    main_stream_ = new c10::hip::HIPStream(c10::hip::getCurrentHIPStream(dev.index()));
    prep_stream_ = new c10::hip::HIPStream(c10::hip::getStreamFromPool(/*isHighPriority=*/true, dev.index()));
    xchg_stream_ = new c10::hip::HIPStream(staleness_ ? c10::hip::getStreamFromPool(/*isHighPriority=*/true, dev.index()) : strm(main_stream_));
    {
        // exact (default): all-to-all(v), split sizes read from pinned memory a few steps after the device wrote them; fixed: fixed-capacity
        // equal-split payloads, nothing read on the host (sharded_trainer.h: measured, 25 % more rows through every exchange stage at world 1)
        const char* e = getenv("MARIUS_EXCHANGE");
        fixed_ = e && e[0] == 'f';
        const char* sl = getenv("MARIUS_EXCHANGE_SLACK");  // planned maximum per (requester, owner) pair = slack * capacity / world (world > 1)
        if (sl && atof(sl) >= 1.0) slack_ = atof(sl);
    }
    {
        const char* e = getenv("MARIUS_SHARDED_DEADLINE_S");
        if (e && atof(e) > 0) deadline_s_ = atof(e);
    }
    const int64_t rec_words = marius_a2a_record_words(world_);
    for (auto& s : slots_) {
        s.rec_host = torch::zeros({rec_words}, torch::TensorOptions().dtype(torch::kInt64).pinned_memory(true));  // stamp 0 = nothing published
        s.hdr.assign(rec_words, 0);
        s.overflow_dev = torch::zeros({1}, torch::TensorOptions().dtype(torch::kInt32).device(dev));
        s.offs_dev = torch::empty({world_ + 1}, torch::TensorOptions().dtype(torch::kInt64).device(dev));
        s.cnt_send_dev = torch::zeros({world_}, torch::TensorOptions().dtype(torch::kInt64).device(dev));
        s.cnt_recv_dev = torch::zeros({world_}, torch::TensorOptions().dtype(torch::kInt64).device(dev));
        s.row_bound = torch::zeros({1}, torch::TensorOptions().dtype(torch::kFloat32).device(dev));
        s.ready = new_event();
        s.fetched = new_event();
        s.computed = new_event();
        s.free_ = new_event();
    }
    saved_run_ahead_ = loader_->run_ahead_;
    loader_->run_ahead_ = false;
    loader_->num_relations_ = model_->decoder_->num_relations_;
    // The exchange and preparation streams carry ~0.9 GB of row traffic per step in kernels that can only run where the persistent matrix
    // launches leave a CU free: 32 CUs without a flash workgroup (400 instead of 480 workgroups at the bench shape: whole tiles, no split tile)
    // cost the matrix launches 4 % and take the step from 0.726 to 0.696 ms at world 1 (sweep 0 / 16 / 32 / 48 / 64 / 96: profiles/r5_sharded_free_cus.txt)
    {
        const char* e = getenv("MARIUS_SHARDED_FREE_CUS");
        saved_free_cus_ = model_->ctx_.free_cus;  // put back by the destructor: a Model later driven by SynchronousTrainer fills the chip again
        model_->ctx_.free_cus = e ? atoi(e) : 32;
    }
    // Where the MT19937 words come from.  The run-ahead pool's own stream would be a FIFTH stream beside compute / preparation / exchange / RCCL, and
    // with the runtime's four hardware queues it lands on the compute stream's queue — its 0.9 ms single-workgroup fill then stands in front of the
    // matrix launches every eighth batch (0.762-0.845 ms per step at world 1; a fifth hardware queue, GPU_MAX_HW_QUEUES=5, is worse still: 1.20 ms).
    // Default: generated on the preparation stream itself, request by request (0.661 ms on the box of the A/B, profiles/r5_sharded_streams.txt).
    // MARIUS_MT_FILL=xchg keeps the pool, one batch per pool, filled on the EXCHANGE stream (tried because the preparation stream's ~25 dependent
    // launches are the step's critical path and the two fills are 0.15 ms of it: 0.749 ms — the fills delay the exchange more than they spared the
    // preparation); =own: the pool on a stream of its own, round 4's form.
    if (loader_->generator_) {
        saved_prefetch_ = loader_->generator_->prefetch_;
        saved_pool_requests_ = loader_->generator_->pool_requests_;
        static const char mode = [] { const char* e = getenv("MARIUS_MT_FILL"); return e ? e[0] : 'p'; }();
        if (mode == 'x' && staleness_ > 0) {
            loader_->generator_->prefetch_ = true;
            loader_->generator_->pool_requests_ = 2;
            loader_->generator_->use_fill_stream((void*)strm(xchg_stream_).stream());
        } else if (mode != 'o') {
            loader_->generator_->prefetch_ = false;
        }
    }
    {
        Scope scope(strm(main_stream_));  // the permutation upload is ordered before the first preparation (which waits for this stream)
        loader_->initializeBatches(true);
    }
    prime();
}

ShardedTrainer::~ShardedTrainer() {
    (void)hipDeviceSynchronize();
    // what the constructor changed on the model / loader / generator it was handed
    model_->ctx_.free_cus = saved_free_cus_;
    loader_->full_batches_only_ = saved_full_batches_;
    loader_->plan_ahead_ = saved_plan_ahead_;
    loader_->run_ahead_ = saved_run_ahead_;
    if (loader_->generator_) {
        loader_->generator_->release_fill_stream();  // (MARIUS_MT_FILL=xchg lent it the exchange stream, which dies with this object)
        loader_->generator_->prefetch_ = saved_prefetch_;
        loader_->generator_->pool_requests_ = saved_pool_requests_;
    }
    for (auto& s : slots_)
        for (void* e : {s.ready, s.fetched, s.computed, s.free_})
            if (e) (void)hipEventDestroy((hipEvent_t)e);
    delete (c10::hip::HIPStream*)main_stream_;
    delete (c10::hip::HIPStream*)prep_stream_;
    delete (c10::hip::HIPStream*)xchg_stream_;
}

// RCCL sets a collective up lazily on its first use per communicator (tens of milliseconds for the first all-reduce); the relation-table
// averaging only happens every sync_interval steps, so without this its one-time cost would land in the middle of a run.
void ShardedTrainer::prime() {
    torch::NoGradGuard ng;  // the relation tables are leaves that require grad: averaging them in place is not an autograd operation
    const auto dev = table_.device();
    Scope scope(strm(main_stream_));
    for (auto& t : model_->dense_state()) {  // every replica starts from the same tables and zero sums: sum / world leaves them unchanged
        std::vector<Tensor> v{t};
        pg_->allreduce(v)->wait();
        t.div_((double)world_);
    }
    model_->touch_relations();
    std::vector<int64_t> ones(world_, 1);
    Tensor rows = torch::zeros({world_, d_}, torch::TensorOptions().dtype(torch::kFloat32).device(dev)), rows_out = torch::empty_like(rows);
    Tensor ids = torch::zeros({world_}, torch::TensorOptions().dtype(torch::kInt64).device(dev)), ids_out = torch::empty_like(ids);
    pg_->alltoall_base(rows_out, rows, ones, ones)->wait();
    pg_->alltoall_base(ids_out, ids, ones, ones)->wait();
    if (world_ > 1 || fixed_) {
        Tensor a = torch::zeros({world_}, torch::TensorOptions().dtype(torch::kInt64).device(dev)), b = torch::zeros_like(a);
        std::vector<int64_t> none;
        pg_->alltoall_base(b, a, none, none)->wait();  // the equal-split form (count exchange; every payload of the fixed-capacity exchange)
        pg_->alltoall_base(rows_out, rows, none, none)->wait();
    }
    strm(main_stream_).synchronize();
}

// [n, *tail] view of a grow-only buffer: a fresh allocation per step keeps the caching allocator splitting / merging blocks
Tensor ShardedTrainer::view(Tensor& buf, int64_t n, std::vector<int64_t> tail, torch::ScalarType dtype) {
    if (!buf.defined() || buf.size(0) < n) {
        if (buf.defined()) {  // the old block may still be read on either stream
            c10::hip::HIPCachingAllocator::recordStream(buf.storage().data_ptr(), strm(main_stream_));
            c10::hip::HIPCachingAllocator::recordStream(buf.storage().data_ptr(), strm(xchg_stream_));
        }
        std::vector<int64_t> shape{std::max<int64_t>(n + n / 4, 1)};
        shape.insert(shape.end(), tail.begin(), tail.end());
        buf = torch::empty(shape, torch::TensorOptions().dtype(dtype).device(table_.device()));
    }
    return buf.narrow(0, 0, n);
}

// An all-to-all(v) whose two sides disagree never completes (the c10d watchdog ends the process ten minutes later): every split vector is
// checked against the tensors it describes before the collective is issued, and a mismatch names the batch.
void ShardedTrainer::check_splits(const Slot& s, const char* what, const Tensor& in, const std::vector<int64_t>& in_split, const Tensor& out,
                                  const std::vector<int64_t>& out_split) const {
    int64_t ns = 0, nr = 0;
    bool neg = false;
    for (auto c : in_split) { ns += c; neg = neg || c < 0; }
    for (auto c : out_split) { nr += c; neg = neg || c < 0; }
    const bool self_ok = world_ > 1 || in_split[0] == out_split[0];  // world 1: what this rank sends itself is what it receives
    if (neg || (int)in_split.size() != world_ || (int)out_split.size() != world_ || in.size(0) != ns || out.size(0) != nr || !self_ok) {
        std::ostringstream m;
        m << "ShardedTrainer: " << what << " all-to-all(v) of batch " << s.batch_index << " (slot " << (s.batch_index % RING) << ", rank " << rank_ << "): send rows " << in.size(0)
          << " vs split sum " << ns << ", receive rows " << out.size(0) << " vs split sum " << nr << "; send splits [";
        for (auto c : in_split) m << c << " ";
        m << "] receive splits [";
        for (auto c : out_split) m << c << " ";
        m << "]";
        throw MariusRuntimeException(m.str());
    }
}

Tensor ShardedTrainer::a2a(const Slot& s, const char* what, const Tensor& in, const std::vector<int64_t>& send_counts, const std::vector<int64_t>& recv_counts, Tensor out) {
    Tensor src = in.contiguous();
    std::vector<int64_t> out_split = recv_counts, in_split = send_counts;
    check_splits(s, what, src, in_split, out, out_split);
    pg_->alltoall_base(out, src, out_split, in_split)->wait();  // NCCL work: orders the current stream behind the collective, the host does not block
    return out;
}

// stage 1 (prep stream): everything that does not read the table — edge slice, negatives, sort / unique, owner split points
void ShardedTrainer::span_begin(Slot& s, int stage, void* stream) {
    if (!spans_) return;
    if (!s.span_b[stage]) {
        hipEvent_t b, e;
        ST_HIPCHECK(hipEventCreate(&b));
        ST_HIPCHECK(hipEventCreate(&e));
        s.span_b[stage] = b;
        s.span_e[stage] = e;
    }
    ST_HIPCHECK(hipEventRecord((hipEvent_t)s.span_b[stage], strm(stream).stream()));
}
void ShardedTrainer::span_end(Slot& s, int stage, void* stream) {
    if (!spans_ || !s.span_b[stage]) return;
    ST_HIPCHECK(hipEventRecord((hipEvent_t)s.span_e[stage], strm(stream).stream()));
    s.span_live[stage] = true;
}
void ShardedTrainer::span_collect(Slot& s) {
    for (int k = 0; k < 7; ++k) {
        if (!s.span_live[k]) continue;
        s.span_live[k] = false;
        float ms = 0.f;
        if (hipEventQuery((hipEvent_t)s.span_e[k]) == hipSuccess && hipEventElapsedTime(&ms, (hipEvent_t)s.span_b[k], (hipEvent_t)s.span_e[k]) == hipSuccess) {
            span_ms_[k] += ms;
            ++span_n_[k];
        }
    }
}

void ShardedTrainer::prepare(int64_t t) {
    Slot& s = slot(t);
    if (s.used) {
        Phase pw(phase_seconds_[1]);  // (fixed-capacity form: the only place the loop can wait — for a preparation issued RING steps ago)
        retire(s);
    }
    Phase ph(phase_seconds_[0]);
    span_collect(s);
    auto& prep = strm(prep_stream_);
    if (s.used) {
        ST_HIPCHECK(hipStreamWaitEvent(prep.stream(), (hipEvent_t)s.free_, 0));  // the batch that used this slot RING steps ago is fully retired
    } else {
        hipEvent_t e = new_event();
        ST_HIPCHECK(hipEventRecord(e, strm(main_stream_).stream()));
        ST_HIPCHECK(hipStreamWaitEvent(prep.stream(), e, 0));
        ST_HIPCHECK(hipEventDestroy(e));
    }
    span_begin(s, 0, prep_stream_);
    {
        Scope scope(prep);
        const int64_t B = loader_->batch_size_;
        if ((loader_->batch_id_ + 1) * B > loader_->num_edges_) loader_->initializeBatches(true);  // next epoch: a new permutation (full batches only)
        {
            Phase pf(fine_seconds_[0]);
            s.batch = loader_->prepareBatch(/*exact_unique=*/false);
        }
        Phase pf2(fine_seconds_[1]);
        s.batch_index = t;
        auto st = (marius_stream_t)prep.stream();
        // split points by owner and the all-to-all(v) send counts, one launch
        mcheck(marius_owner_offsets_counts(s.batch->unique_node_indices_.data_ptr<int64_t>(), s.batch->num_unique_dev_.data_ptr<int64_t>(), S_, world_,
                                           s.offs_dev.data_ptr<int64_t>(), s.cnt_send_dev.data_ptr<int64_t>(), st));
        const int64_t* recv_counts_dev = nullptr;   // NULL: the header carries the send counts (world 1: what this rank asks itself for)
        const int32_t* overflow_dev = nullptr;
        if (fixed_) {
            // fixed-capacity exchange: the id payload (-1 padded blocks of cap slots per owner) and the slot of every unique row; the exchange
            // itself needs nothing of this batch on the host (the header is read for the overflow flag and the byte accounting only)
            setup_fixed(s, s.batch->occ_perm_.size(0));
            s.overflow_dev.zero_();
            mcheck(marius_a2a_rows_post(s.batch->unique_node_indices_.data_ptr<int64_t>(), s.offs_dev.data_ptr<int64_t>(), S_, world_, cap_,
                                        s.req_send.data_ptr<int64_t>(), s.place.data_ptr<int64_t>(), s.overflow_dev.data_ptr<int32_t>(),
                                        s.batch->occ_inverse_.data_ptr<int64_t>(), L_, s.slot_of_occ.data_ptr<int64_t>(), st));
            if (world_ > 1) {
                // every rank must refuse an overflowing batch TOGETHER (a rank that throws alone leaves its peers blocked in the next all-to-all
                // until the c10d watchdog fires): the flag is max-reduced over the ranks here, four batches before anybody exchanges the batch
                std::vector<Tensor> v{s.overflow_dev};
                c10d::AllreduceOptions opts;
                opts.reduceOp = c10d::ReduceOp::MAX;
                pg_->allreduce(v, opts)->wait();
            }
            overflow_dev = s.overflow_dev.data_ptr<int32_t>();
            // the batch's local indices in slot terms (dataloader.cpp:460-466 with place o inverse instead of inverse): the decoder then reads
            // the row payload where the all-to-all left it — no compacted [U, d] copy
            const int64_t Bb = s.batch->global_edges_.size(0), cols = s.batch->global_edges_.size(1), CN = s.batch->src_neg_indices_.numel();
            if (!s.edges_slot.defined() || s.edges_slot.size(0) != Bb) s.edges_slot = torch::empty({Bb, cols}, s.batch->global_edges_.options());
            mcheck(marius_remap_edges(s.batch->global_edges_.data_ptr<int64_t>(), s.slot_of_occ.data_ptr<int64_t>(), Bb, (int32_t)cols, s.edges_slot.data_ptr<int64_t>(), st));
            s.batch->edges_ = s.edges_slot;
            s.batch->src_neg_indices_mapping_ = s.slot_of_occ.narrow(0, 2 * Bb, CN).view(s.batch->src_neg_indices_.sizes());
            s.batch->dst_neg_indices_mapping_ = s.slot_of_occ.narrow(0, 2 * Bb + CN, CN).view(s.batch->dst_neg_indices_.sizes());
        } else if (world_ > 1) {
            // The receive counts of the all-to-all(v) travel on the device as well: a `world`-integer all-to-all of the send counts on this
            // (preparation) stream.  Every rank issues its collectives in the same program order, which is all a communicator requires.
            std::vector<int64_t> none;
            pg_->alltoall_base(s.cnt_recv_dev, s.cnt_send_dev, none, none)->wait();
            recv_counts_dev = s.cnt_recv_dev.data_ptr<int64_t>();
        }
        // the LAST operation of the preparation: the whole header in one ordered record (exchange.hip: payload, system fence, stamp)
        s.stamp_value = t + 1;
        mcheck(marius_a2a_publish(s.offs_dev.data_ptr<int64_t>(), recv_counts_dev, overflow_dev, world_, s.stamp_value, s.rec_host.data_ptr<int64_t>(), st));
    }
    span_end(s, 0, prep_stream_);
    ST_HIPCHECK(hipEventRecord((hipEvent_t)s.ready, prep.stream()));
    s.used = true;
}

void ShardedTrainer::prepare_through(int64_t t) {
    while (next_prepared_ <= t) prepare(next_prepared_++);
}

// what a stuck pipeline looks like from the host: which headers arrived, which events fired, which streams still hold work
std::string ShardedTrainer::describe_state() const {
    std::ostringstream m;
    auto q = [](void* e) { return !e ? "-" : hipEventQuery((hipEvent_t)e) == hipSuccess ? "done" : "pending"; };
    m << "rank " << rank_ << "/" << world_ << " step " << step_index_ << " prepared<" << next_prepared_ << " fetched<" << next_fetched_ << " exchange " << (fixed_ ? "fixed" : "exact")
      << " torn_reads " << torn_reads_ << "; streams:";
    const char* names[3] = {"main", "prep", "xchg"};
    void* streams[3] = {main_stream_, prep_stream_, xchg_stream_};
    for (int k = 0; k < 3; ++k)
        m << " " << names[k] << "=" << (streams[k] && hipStreamQuery(((c10::hip::HIPStream*)streams[k])->stream()) == hipSuccess ? "idle" : "busy");
    for (int k = 0; k < RING; ++k) {
        const Slot& s = slots_[k];
        if (!s.used) continue;
        const int64_t seen = __atomic_load_n(s.rec_host.data_ptr<int64_t>(), __ATOMIC_ACQUIRE);
        m << "; slot " << k << " batch " << s.batch_index << " stamp " << seen << "/" << s.stamp_value << " ready=" << q(s.ready) << " fetched=" << q(s.fetched)
          << " computed=" << q(s.computed) << " free=" << q(s.free_);
    }
    return m.str();
}

// stage 2 (host + exchange stream): split sizes, then ids -> owners, rows -> requesters.
// The header of a batch prepared at least one step ago: poll the stamp (acquire), copy the record, verify its checksum; a record that fails
// the checksum is polled again (and counted), one that never arrives fails the step with the pipeline's state in the message.
void ShardedTrainer::wait_prepared(Slot& s) {
    if (s.hdr_stamp == s.stamp_value) return;
    const int64_t* rec = s.rec_host.data_ptr<int64_t>();
    const int n = (int)s.hdr.size();
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t spins = 0;; ++spins) {
        if (__atomic_load_n(rec, __ATOMIC_ACQUIRE) == s.stamp_value) {
            s.hdr[0] = s.stamp_value;
            for (int w = 1; w < n; ++w) s.hdr[w] = __atomic_load_n(rec + w, __ATOMIC_RELAXED);
            if ((uint64_t)s.hdr[n - 1] == marius_a2a_record_checksum(s.hdr.data(), world_)) {
                s.hdr_stamp = s.stamp_value;
                return;
            }
            ++torn_reads_;
        }
        if ((spins & 0xfff) == 0xfff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadline_s_) {
            failed_ = true;
            throw MariusRuntimeException("ShardedTrainer: the header of batch " + std::to_string(s.batch_index) + " did not reach the host within " +
                                         std::to_string(deadline_s_) + " s (MARIUS_SHARDED_DEADLINE_S) — " + describe_state());
        }
        cpu_relax();
    }
}

void ShardedTrainer::refuse_overflow(const Slot& s) const {
    if (s.hdr[2 * world_ + 2] != 0)
        throw MariusRuntimeException("ShardedTrainer: batch " + std::to_string(s.batch_index) + " asked one owner for more than the planned maximum of " + std::to_string(cap_) +
                                     " rows on some rank (fixed-capacity exchange, slack " + std::to_string(slack_) + "): raise MARIUS_EXCHANGE_SLACK, or MARIUS_EXCHANGE=exact." +
                                     "  Nothing of the batch was exchanged or applied; every rank raises at this batch.");
}

// a slot is reused RING steps after its batch was prepared: the header is long there, so this reads pinned memory without waiting
void ShardedTrainer::retire(Slot& s) {
    wait_prepared(s);
    if (fixed_) {
        for (int q = 0; q < world_; ++q)
            if (q != rank_) useful_rows_[0] += s.hdr[2 + q] - s.hdr[1 + q];
    }
}

void ShardedTrainer::setup_fixed(Slot& s, int64_t L) {
    const auto dev = table_.device();
    auto i64o = torch::TensorOptions().dtype(torch::kInt64).device(dev), i32o = torch::TensorOptions().dtype(torch::kInt32).device(dev);
    auto f32o = torch::TensorOptions().dtype(torch::kFloat32).device(dev), u8o = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    if (L_ == 0) {
        L_ = L;
        cap_ = marius_a2a_capacity(L, world_, slack_);
        ncap_ = cap_ * world_;
        run_offsets_.resize(world_ + 1);
        for (int q = 0; q <= world_; ++q) run_offsets_[q] = (int64_t)q * cap_;
        req_recv_ = torch::empty({ncap_}, i64o);
        rows_send_ = torch::empty({ncap_, d_}, f32o);
        grad_recv_ = torch::empty({ncap_, d_}, f32o);
        r_ws_ = torch::zeros({(int64_t)marius_sort_unique_workspace_bytes(ncap_)}, u8o);
        r_carry_ = torch::empty({(int64_t)marius_segment_carry_bytes(ncap_, d_)}, u8o);
    }
    if (L != L_) throw MariusRuntimeException("ShardedTrainer: the fixed-capacity exchange needs batches of one size (full batches only)");
    if (s.req_send.defined()) return;
    s.req_send = torch::empty({ncap_}, i64o);
    s.place = torch::empty({L_}, i64o);
    s.slot_of_occ = torch::empty({L_}, i64o);
    s.grad_send = torch::empty({ncap_, d_}, f32o);
    s.emb = torch::zeros({ncap_, d_}, f32o);  // the row payload as received (zeroed ONCE: unused slots then hold finite values for the bound scan)
    s.r_uniq = torch::empty({ncap_}, i64o);
    s.r_inverse = torch::empty({ncap_}, i64o);
    s.r_perm = torch::empty({ncap_}, i32o);
    s.r_seg = torch::empty({ncap_ + 1}, i32o);
    s.r_count = torch::zeros({1}, i64o);
    s.r_plan = torch::empty({(int64_t)marius_segment_plan_bytes(ncap_)}, u8o);
}

// stage 2, fixed-capacity form (exchange stream; the host reads nothing): ids -> owners, owners gather + plan their update, rows -> requesters
void ShardedTrainer::fetch_fixed(int64_t t) {
    Slot& s = slot(t);
    {
        // the header was published AHEAD batches ago: no wait in steady state.  The overflow flag in it is the MAX over all ranks, so every rank
        // refuses the batch here, BEFORE any of its payloads is exchanged, scored or applied (no corrupted update ever reaches a shard)
        Phase pw(phase_seconds_[1]);
        wait_prepared(s);
        if (s.hdr[2 * world_ + 2] != 0) failed_ = true;
        refuse_overflow(s);
    }
    Phase ph(phase_seconds_[2]);
    auto& xchg = strm(xchg_stream_);
    ST_HIPCHECK(hipStreamWaitEvent(xchg.stream(), (hipEvent_t)s.ready, 0));
    span_begin(s, 1, xchg_stream_);
    {
        Scope scope(xchg);
        auto st = (marius_stream_t)xchg.stream();
        std::vector<int64_t> none;
        span_begin(s, 4, xchg_stream_);
        pg_->alltoall_base(req_recv_, s.req_send, none, none)->wait();  // orders this stream behind the collective; the host does not block
        span_end(s, 4, xchg_stream_);
        span_begin(s, 5, xchg_stream_);
        // owner: rows of the ids asked for (-1 = unused slot: skipped), in the requesters' slot order
        mcheck(marius_gather_rows(table_.data_ptr<float>(), table_.stride(0), req_recv_.data_ptr<int64_t>(), ncap_, d_, rows_send_.data_ptr<float>(),
                                  rows_send_.stride(0), st));
        span_end(s, 5, xchg_stream_);
        span_begin(s, 6, xchg_stream_);
        pg_->alltoall_base(s.emb, rows_send_, none, none)->wait();
        span_end(s, 6, xchg_stream_);
        // requester: magnitude bound of the received payload (marius_lp_desc.absmax: fp16 operand halves); the rows are scored where they landed
        if (Model::flash_f16_enabled()) {
            s.row_bound.zero_();
            mcheck(marius_a2a_rows_wait(s.emb.data_ptr<float>(), s.emb.stride(0), ncap_, d_, s.row_bound.data_ptr<float>(), nullptr, nullptr, 0, nullptr, 0, st));
        }
        // owner: the received id payload is `world` non-decreasing runs of cap slots (-1 padding first): merge them and plan the segmented
        // update NOW — the ids are here a whole scoring pass before the gradients
        mcheck(marius_merge_unique_runs(req_recv_.data_ptr<int64_t>(), ncap_, run_offsets_.data(), world_, s.r_uniq.data_ptr<int64_t>(), s.r_inverse.data_ptr<int64_t>(),
                                        s.r_perm.data_ptr<int32_t>(), s.r_seg.data_ptr<int32_t>(), s.r_count.data_ptr<int64_t>(), r_ws_.data_ptr(), (size_t)r_ws_.numel(), st));
        mcheck(marius_segment_plan(s.r_perm.data_ptr<int32_t>(), s.r_inverse.data_ptr<int64_t>(), s.r_seg.data_ptr<int32_t>(), s.r_uniq.data_ptr<int64_t>(), ncap_,
                                   s.r_plan.data_ptr(), st));
    }
    span_end(s, 1, xchg_stream_);
    ST_HIPCHECK(hipEventRecord((hipEvent_t)s.fetched, xchg.stream()));
    // what crosses the wire is the padded payload: (world - 1) blocks of cap slots each way
    exchange_bytes_[0] += (int64_t)(world_ - 1) * cap_ * 8;
    exchange_bytes_[1] += (int64_t)(world_ - 1) * cap_ * d_ * 4;
    exchange_bytes_[2] += (int64_t)(world_ - 1) * cap_ * d_ * 4;
}

// stage 4, fixed-capacity form: gradients -> owners, one grouped launch pair applies them (plan from fetch_fixed)
void ShardedTrainer::update_fixed(int64_t t) {
    Phase ph(phase_seconds_[4]);
    Slot& s = slot(t);
    auto& xchg = strm(xchg_stream_);
    ST_HIPCHECK(hipStreamWaitEvent(xchg.stream(), (hipEvent_t)s.computed, 0));
    span_begin(s, 3, xchg_stream_);
    {
        Scope scope(xchg);
        std::vector<int64_t> none;
        pg_->alltoall_base(grad_recv_, s.grad_send, none, none)->wait();
        marius_segment_update u = {};
        u.rows = grad_recv_.data_ptr<float>();
        u.rows_ld = grad_recv_.stride(0);
        u.perm = s.r_perm.data_ptr<int32_t>();
        u.inverse = s.r_inverse.data_ptr<int64_t>();
        u.seg_offsets = s.r_seg.data_ptr<int32_t>();
        u.n = ncap_;
        u.d = d_;
        u.uniq_ids = s.r_uniq.data_ptr<int64_t>();  // LOCAL row ids (what the requesters sent); the -1 run is planned dead
        u.table = table_.data_ptr<float>();
        u.state = state_.data_ptr<float>();
        u.table_ld = table_.stride(0);
        u.lr = model_->sparse_lr_;
        u.eps = 1e-10f;
        u.carry = r_carry_.data_ptr();
        u.plan = s.r_plan.data_ptr();
        mcheck(marius_segment_adagrad_scatter_group(&u, 1, (marius_stream_t)xchg.stream()));
    }
    span_end(s, 3, xchg_stream_);
    ST_HIPCHECK(hipEventRecord((hipEvent_t)s.free_, xchg.stream()));
}

void ShardedTrainer::fetch(int64_t t) {
    if (fixed_) return fetch_fixed(t);
    Slot& s = slot(t);
    {
        Phase ph(phase_seconds_[1]);
        wait_prepared(s);  // a batch prepared at least one step ago: no stream drains for this
    }
    Phase ph(phase_seconds_[2]);
    const int64_t* offs = s.hdr.data() + 1;
    const int64_t* rc = s.hdr.data() + world_ + 2;
    s.send_counts.assign(world_, 0);
    for (int i = 0; i < world_; ++i) s.send_counts[i] = offs[i + 1] - offs[i];
    s.recv_counts.assign(rc, rc + world_);
    s.U = offs[world_];
    s.nrecv = 0;
    for (auto c : s.recv_counts) s.nrecv += c;
    {   // a header that passed its checksum is what the device wrote; this guards the device side (a wrong unique count, a shard-size mismatch)
        bool ok = offs[0] == 0 && s.U >= 0 && s.U <= s.batch->unique_node_indices_.size(0);
        for (int i = 0; i < world_; ++i) ok = ok && s.send_counts[i] >= 0 && s.recv_counts[i] >= 0 && s.recv_counts[i] <= table_.size(0);
        if (!ok) {
            failed_ = true;
            throw MariusRuntimeException("ShardedTrainer: implausible exchange header for batch " + std::to_string(s.batch_index) + " — " + describe_state());
        }
    }
    const int k = (int)(t % RING);
    auto& xchg = strm(xchg_stream_);
    ST_HIPCHECK(hipStreamWaitEvent(xchg.stream(), (hipEvent_t)s.ready, 0));
    span_begin(s, 1, xchg_stream_);
    {
        Scope scope(xchg);
        span_begin(s, 4, xchg_stream_);
        Tensor req = a2a(s, "id", s.batch->unique_node_indices_.narrow(0, 0, s.U), s.send_counts, s.recv_counts, view(buf_req_, s.nrecv, {}, torch::kInt64));
        span_end(s, 4, xchg_stream_);
        span_begin(s, 5, xchg_stream_);
        s.local_ids = view(local_[k], s.nrecv, {}, torch::kInt64);
        torch::sub_out(s.local_ids, req, lo_);
        Tensor rows = view(buf_rows_, s.nrecv, {d_}, torch::kFloat32);
        if (s.nrecv > 0)
            mcheck(marius_gather_rows(table_.data_ptr<float>(), table_.stride(0), s.local_ids.data_ptr<int64_t>(), s.nrecv, d_, rows.data_ptr<float>(),
                                      rows.stride(0), (marius_stream_t)xchg.stream()));
        span_end(s, 5, xchg_stream_);
        span_begin(s, 6, xchg_stream_);
        s.emb = a2a(s, "row", rows, s.recv_counts, s.send_counts, view(emb_[k], s.U, {d_}, torch::kFloat32));
        span_end(s, 6, xchg_stream_);
        // Magnitude bound of the rows this batch will read (marius_lp_desc.absmax: fp16 operand halves on the flash path).  The rows come from
        // every rank's shard, so no rank's own table bound covers them — but the requester holds all of them right here: one pass over the
        // gathered copy (80 MB at the bench shape, on this stream, underneath the scoring of the previous batch) is exact and needs no
        // collective.  One float per slot: the scoring of batch t reads its slot's bound while batch t + 1's is being written.
        if (s.U > 0 && Model::flash_f16_enabled())
            mcheck(marius_table_absmax(s.emb.data_ptr<float>(), s.U, s.emb.stride(0), d_, s.row_bound.data_ptr<float>(), (marius_stream_t)xchg.stream()));
        plan_local(s);  // the owner's merge + segment plan of this batch's update: the ids are here a scoring pass before the gradients
    }
    span_end(s, 1, xchg_stream_);
    ST_HIPCHECK(hipEventRecord((hipEvent_t)s.fetched, xchg.stream()));
    for (int q = 0; q < world_; ++q) {
        if (q == rank_) continue;
        exchange_bytes_[0] += s.send_counts[q] * 8;             // ids to the owners
        exchange_bytes_[1] += s.recv_counts[q] * (int64_t)d_ * 4;  // rows served to the requesters
        exchange_bytes_[2] += s.send_counts[q] * (int64_t)d_ * 4;  // gradients returned to the owners (update())
        useful_rows_[0] += s.send_counts[q];
        useful_rows_[1] += s.recv_counts[q];
    }
}

void ShardedTrainer::fetch_through(int64_t t) {
    while (next_fetched_ <= t) {
        prepare_through(next_fetched_);
        fetch(next_fetched_++);
    }
}

// stage 3 (main stream): the same forward / loss / backward kernels as the single-GPU step
void ShardedTrainer::compute(int64_t t) {
    Phase ph(phase_seconds_[3]);
    Slot& s = slot(t);
    auto& main = strm(main_stream_);
    Scope scope(main);
    ST_HIPCHECK(hipStreamWaitEvent(main.stream(), (hipEvent_t)s.fetched, 0));
    span_begin(s, 2, main_stream_);
    s.batch->node_embeddings_ = s.emb;
    if (Model::flash_f16_enabled()) s.batch->row_bound_ = s.row_bound;
    // replicas step on their own relation gradients between averaging points (sync_interval > 1); with sync_interval 1 the dense
    // gradients are all-reduced first
    if (fixed_) {  // per-row gradient sums straight into the owners' slot order (the gradient payload)
        model_->backward_to_unique_grads(s.batch, s.grad_send, sync_interval_ > 1, s.place);
    } else {
        s.grad = view(grad_[t % RING], s.U, {d_}, torch::kFloat32);
        model_->backward_to_unique_grads(s.batch, s.grad, sync_interval_ > 1);
    }
    span_end(s, 2, main_stream_);
    ST_HIPCHECK(hipEventRecord((hipEvent_t)s.computed, main.stream()));
}

// owner side: a row may have been requested by several ranks -> merge the senders' ascending id runs, sum per row, one Adagrad step per row.
// plan_local runs when the IDS arrive (fetch: a scoring pass before the gradients), apply_local when the gradients do: one grouped launch pair.
void ShardedTrainer::plan_local(Slot& s) {
    const int64_t n = s.local_ids.size(0);
    if (n == 0) return;
    const auto dev = table_.device();
    auto i64o = torch::TensorOptions().dtype(torch::kInt64).device(dev), i32o = torch::TensorOptions().dtype(torch::kInt32).device(dev);
    auto u8o = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    if (!s.r_uniq.defined() || s.r_uniq.size(0) < n) {  // per slot: the plan of batch t + 1 is written while the update of batch t still reads its own
        const int64_t cap = std::max<int64_t>(n + n / 2, 1024);
        s.r_uniq = torch::empty({cap}, i64o);
        s.r_inverse = torch::empty({cap}, i64o);
        s.r_perm = torch::empty({cap}, i32o);
        s.r_seg = torch::empty({cap + 1}, i32o);
        s.r_count = torch::zeros({1}, i64o);
        s.r_plan = torch::empty({(int64_t)marius_segment_plan_bytes(cap)}, u8o);
    }
    if (r_cap_ < n) {  // shared scratch: the merge workspace (used inside fetch only) and the update's carries (inside update only)
        r_cap_ = std::max<int64_t>(n + n / 2, 1024);
        r_ws_ = torch::zeros({(int64_t)marius_sort_unique_workspace_bytes(r_cap_)}, u8o);
        r_carry_ = torch::empty({(int64_t)marius_segment_carry_bytes(r_cap_, d_)}, u8o);
    }
    auto st = cur_stream();
    // every sender's list is ascending and duplicate-free: merge the `world` runs (binary searches) instead of radix-sorting them
    std::vector<int64_t> runs(s.recv_counts.size() + 1, 0);
    for (size_t q = 0; q < s.recv_counts.size(); ++q) runs[q + 1] = runs[q] + s.recv_counts[q];
    if ((int)s.recv_counts.size() <= 64)
        mcheck(marius_merge_unique_runs(s.local_ids.data_ptr<int64_t>(), n, runs.data(), (int32_t)s.recv_counts.size(), s.r_uniq.data_ptr<int64_t>(),
                                        s.r_inverse.data_ptr<int64_t>(), s.r_perm.data_ptr<int32_t>(), s.r_seg.data_ptr<int32_t>(), s.r_count.data_ptr<int64_t>(),
                                        r_ws_.data_ptr(), (size_t)r_ws_.numel(), st));
    else
        mcheck(marius_sort_unique(s.local_ids.data_ptr<int64_t>(), n, key_bits(table_.size(0)), s.r_uniq.data_ptr<int64_t>(), s.r_inverse.data_ptr<int64_t>(),
                                  s.r_perm.data_ptr<int32_t>(), s.r_seg.data_ptr<int32_t>(), s.r_count.data_ptr<int64_t>(), r_ws_.data_ptr(), (size_t)r_ws_.numel(), st));
    // (the plan's size is part of its layout: it is written and read with the same n)
    mcheck(marius_segment_plan(s.r_perm.data_ptr<int32_t>(), s.r_inverse.data_ptr<int64_t>(), s.r_seg.data_ptr<int32_t>(), s.r_uniq.data_ptr<int64_t>(), n,
                               s.r_plan.data_ptr(), st));
}

void ShardedTrainer::apply_local(Slot& s, const Tensor& grads) {
    const int64_t n = s.local_ids.size(0);
    if (n == 0) return;
    marius_segment_update u = {};
    u.rows = grads.data_ptr<float>();
    u.rows_ld = grads.stride(0);
    u.perm = s.r_perm.data_ptr<int32_t>();
    u.inverse = s.r_inverse.data_ptr<int64_t>();
    u.seg_offsets = s.r_seg.data_ptr<int32_t>();
    u.n = n;
    u.d = d_;
    u.uniq_ids = s.r_uniq.data_ptr<int64_t>();
    u.table = table_.data_ptr<float>();
    u.state = state_.data_ptr<float>();
    u.table_ld = table_.stride(0);
    u.lr = model_->sparse_lr_;
    u.eps = 1e-10f;
    u.carry = r_carry_.data_ptr();
    u.plan = s.r_plan.data_ptr();
    mcheck(marius_segment_adagrad_scatter_group(&u, 1, cur_stream()));  // (falls back to the planned single-table launches where the grouped form does not apply)
}

// stage 4 (exchange stream): gradients -> owners, owners update their rows
void ShardedTrainer::update(int64_t t) {
    if (fixed_) return update_fixed(t);
    Phase ph(phase_seconds_[4]);
    Slot& s = slot(t);
    auto& xchg = strm(xchg_stream_);
    ST_HIPCHECK(hipStreamWaitEvent(xchg.stream(), (hipEvent_t)s.computed, 0));
    span_begin(s, 3, xchg_stream_);
    {
        Scope scope(xchg);
        Tensor recv_grad = a2a(s, "gradient", s.grad, s.send_counts, s.recv_counts, view(buf_recv_grad_, s.nrecv, {d_}, torch::kFloat32));
        apply_local(s, recv_grad);
    }
    span_end(s, 3, xchg_stream_);
    ST_HIPCHECK(hipEventRecord((hipEvent_t)s.free_, xchg.stream()));
}

void ShardedTrainer::dense(int64_t t) {
    Phase ph(phase_seconds_[5]);
    torch::NoGradGuard ng;
    Scope scope(strm(main_stream_));
    if (sync_interval_ <= 1) {  // model.cpp:136-159: all-reduce the relation gradients, every replica takes the same dense step
        for (Tensor* g : {&model_->relations_grad_, &model_->inverse_relations_grad_}) {
            if (!g->defined()) continue;
            std::vector<Tensor> v{*g};
            pg_->allreduce(v)->wait();
        }
        model_->step();
    } else if ((t + 1) % sync_interval_ == 0) {  // pipeline_gpu.cpp:52-80 (gpu_model_average): tables and optimizer state averaged every K steps
        for (auto& tt : model_->dense_state()) {
            std::vector<Tensor> v{tt};
            pg_->allreduce(v)->wait();
            tt.div_((double)world_);
        }
        model_->touch_relations();  // the averaged tables are not the ones whose magnitude the local steps tracked
    }
}

void ShardedTrainer::step() {
    if (failed_) throw MariusRuntimeException("ShardedTrainer: an earlier step failed; the pipeline is half-advanced and cannot continue");
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t t = step_index_;
    fetch_through(t);  // no-op except on the first step
    // The scoring of batch t goes to the device FIRST: fetching the next batch may block the host (it needs the split points of a batch
    // whose preparation runs in the gaps the big kernels leave), and the compute stream must not run dry meanwhile.
    compute(t);
    prepare_through(t + AHEAD);  // preparation never reads the table: how far ahead it runs changes nothing but latency hiding
    // rows of batch t + s move while batch t is scored; on the exchange stream this precedes update(t).  s = 1: the fetch of batch t + 1 still
    // queues behind update(t - 1), i.e. behind the scoring of t - 1, and has only the rest of scoring t to finish in: the critical cycle is
    // score(t) -> update(t) -> fetch(t + 2) -> score(t + 2), period >= (score + update + fetch) / 2.  s = 2 takes the fetch off that
    // cycle (period >= (score + update + fetch) / 3); rows are then up to two updates stale.  At world 1 (no wire time) the choice does not
    // matter — 1.02 / 1.00 / 1.02 ms per step for s = 1 / 2 / 3: the device is simply full (device_span_ms) — so the default stays 1.
    if (staleness_) fetch_through(t + staleness_);
    update(t);
    dense(t);
    ++step_index_;
    ++steps_;
    host_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void ShardedTrainer::train_steps(int64_t n) {
    for (int64_t i = 0; i < n; ++i) step();
}

void ShardedTrainer::finish() {
    ST_HIPCHECK(hipDeviceSynchronize());
    if (getenv("MARIUS_SHARDED_FINE"))
        fprintf(stderr, "[sharded fine] per step ms: prepareBatch %.4f, rest of prepare %.4f (steps %ld)\n", fine_seconds_[0] / std::max<int64_t>(steps_, 1) * 1e3,
                fine_seconds_[1] / std::max<int64_t>(steps_, 1) * 1e3, (long)steps_);
    // batches prepared ahead but never fetched are dropped; an overflow among them was never acted on and is nobody's error
}

std::vector<Tensor> c10d_exchange_selftest(const std::string& group_name, Tensor send, std::vector<int64_t> send_counts, Tensor to_reduce) {
    auto pg = c10d::resolve_process_group(group_name);
    const int world = pg->getSize();
    if ((int)send_counts.size() != world) throw MariusRuntimeException("c10d_exchange_selftest: one count per rank");
    Tensor sc = torch::from_blob(send_counts.data(), {world}, torch::kInt64).clone(), rc = torch::empty({world}, torch::kInt64);
    std::vector<int64_t> none;
    pg->alltoall_base(rc, sc, none, none)->wait();  // as ShardedTrainer::fetch does over the side group
    std::vector<int64_t> recv_counts(rc.data_ptr<int64_t>(), rc.data_ptr<int64_t>() + world);
    int64_t nrecv = 0;
    for (auto c : recv_counts) nrecv += c;
    std::vector<int64_t> shape = send.sizes().vec();
    shape[0] = nrecv;
    Tensor out = torch::empty(shape, send.options()), src = send.contiguous();
    pg->alltoall_base(out, src, recv_counts, send_counts)->wait();  // ShardedTrainer::a2a
    std::vector<Tensor> v{to_reduce};
    pg->allreduce(v)->wait();
    return {rc, out, to_reduce};
}

}  // namespace marius_amd
