// Implementation of the host layer (see marius_host.h).  Every device operation is a call into libmarius_hip.so.
#include "marius_host.h"

#include <condition_variable>
#include <array>
#include <deque>
#include <map>
#include <exception>
#include <mutex>
#include <thread>
#include "partition_buffer.h"

#include <c10/hip/HIPStream.h>
#include <torch/csrc/distributed/c10d/GroupRegistry.hpp>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <c10/hip/HIPFunctions.h>
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <functional>
#include <sstream>

namespace marius_amd {

// ------------------------------------------------------------------------------------------------ helpers
std::string TensorSizeMismatchException::describe(const Tensor& t, const std::string& msg) {
    std::stringstream ss;
    ss << "Tensor size mismatch. Size: " << (t.defined() ? t.sizes() : at::IntArrayRef{}) << " " << msg;
    return ss.str();
}

void mcheck(int rc) {
    if (rc != MARIUS_OK) throw MariusRuntimeException(std::string("libmarius_hip: ") + marius_hip_last_error());
}

marius_stream_t cur_stream() { return (marius_stream_t)c10::hip::getCurrentHIPStream().stream(); }

void require_device(const Tensor& t, const char* what) {
    if (!t.defined()) throw UndefinedTensorException();
    if (!t.is_cuda()) throw MariusRuntimeException(std::string(what) + ": tensor must live on the MI355X (no CPU fallback)");
}

static inline float* fp(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
static inline int64_t* ip(const Tensor& t) { return t.defined() ? t.data_ptr<int64_t>() : nullptr; }
static inline int key_bits_for(int64_t n) {
    int b = 1;
    while ((1ll << b) <= n && b < 63) ++b;
    return b;
}
static torch::TensorOptions i64(torch::Device d) { return torch::TensorOptions().dtype(torch::kInt64).device(d); }
static torch::TensorOptions i32(torch::Device d) { return torch::TensorOptions().dtype(torch::kInt32).device(d); }
static torch::TensorOptions f32(torch::Device d) { return torch::TensorOptions().dtype(torch::kFloat32).device(d); }

// ------------------------------------------------------------------------------------------------ generator
#define HIPCHECK(x)                                                                                         \
    do {                                                                                                    \
        hipError_t e_ = (x);                                                                                \
        if (e_ != hipSuccess) throw MariusRuntimeException(std::string("HIP: ") + hipGetErrorString(e_));   \
    } while (0)

namespace {
struct StreamScope {  // make `s` the current torch stream of this thread for the lifetime of the object
    c10::hip::HIPStream prev;
    explicit StreamScope(c10::hip::HIPStream s) : prev(c10::hip::getCurrentHIPStream(s.device_index())) { c10::hip::setCurrentHIPStream(s); }
    ~StreamScope() { c10::hip::setCurrentHIPStream(prev); }
};
}  // namespace

// Flags of every event this file uses to order one stream of the device after another (loader -> main, main -> loader gate, pool fills, the
// relation side stream): no timing and NO system-scope fence.  A default event makes its record a system-scope release — an L2 write-back the
// following kernel waits for (measured: the 7 us hole between edge_bwd and seg_reduce on the main queue, where the loader's gate is recorded) —
// which only a HOST reader of device-written host memory needs, and none of these events has one (the host reads results through
// torch's own synchronising copies).  MARIUS_EVENT_FENCE=system restores the default (A/B runs).
unsigned order_event_flags() {
    static const unsigned f = [] {
        const char* e = getenv("MARIUS_EVENT_FENCE");
        return (e && e[0] == 's') ? (unsigned)hipEventDisableTiming : (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence);
    }();
    return f;
}

void* aux_stream(int device_index, int which) {
    static std::mutex mu;
    static std::map<int, std::array<hipStream_t, AUX_COUNT>> streams;
    std::lock_guard<std::mutex> lk(mu);
    auto it = streams.find(device_index);
    if (it == streams.end()) {
        int prev = 0;
        HIPCHECK(hipGetDevice(&prev));
        HIPCHECK(hipSetDevice(device_index));
        std::array<hipStream_t, AUX_COUNT> a{};
        for (auto& s : a) HIPCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));  // all three at once, in a fixed order: the same queues in every process
        HIPCHECK(hipSetDevice(prev));
        it = streams.emplace(device_index, a).first;
    }
    return it->second[which];
}

MariusGenerator::MariusGenerator(uint64_t seed) {
    state_host_ = torch::zeros({MARIUS_MT_STATE_WORDS}, torch::kInt32);
    marius_mt19937_seed_host((uint32_t*)state_host_.data_ptr<int32_t>(), seed);
}
MariusGenerator::~MariusGenerator() {
    for (auto& p : pools_) {
        if (p.ready) (void)hipEventDestroy((hipEvent_t)p.ready);
        if (p.done) (void)hipEventDestroy((hipEvent_t)p.done);
    }
}
void MariusGenerator::use_fill_stream(void* hip_stream) {
    if (side_stream_ && side_stream_owned_) (void)hipStreamSynchronize((hipStream_t)side_stream_);  // the process-wide fill stream: finish what it holds for us
    side_stream_ = hip_stream;
    side_stream_owned_ = false;
    side_ordered_ = false;
}
void MariusGenerator::release_fill_stream() {
    if (side_stream_ && !side_stream_owned_) {
        to_host();  // waits for the borrowed stream, rewinds the state to exactly the consumed position, drops the pools filled on that stream
        side_stream_ = nullptr;
        side_stream_owned_ = true;
        side_ordered_ = false;
    }
}
void MariusGenerator::drop_pools() {
    for (auto& p : pools_) {
        p.filled = p.waited = p.has_done = false;
        p.used = 0;
    }
    cur_ = 0;
}
void MariusGenerator::to_host() {
    if (!state_dev_.defined()) return;
    if (side_stream_) HIPCHECK(hipStreamSynchronize((hipStream_t)side_stream_));
    Pool& p = pools_[cur_];
    if (prefetch_ && p.filled) {
        // the device state has run ahead: rewind to the start of the pool being consumed and replay what was consumed
        state_host_ = p.state_before.cpu();
        if (p.used > 0) {
            Tensor scratch = torch::empty({p.used}, torch::kInt32);
            marius_mt19937_fill_host((uint32_t*)state_host_.data_ptr<int32_t>(), (uint32_t*)scratch.data_ptr<int32_t>(), p.used);
        }
    } else {
        state_host_ = state_dev_.cpu();
    }
    state_dev_ = Tensor();
    drop_pools();
}
void MariusGenerator::to_device(torch::Device dev) {
    if (!state_dev_.defined()) state_dev_ = state_host_.to(dev);
}
Tensor MariusGenerator::randperm(int64_t n) {
    to_host();
    Tensor out = torch::empty({n}, torch::kInt64);
    mcheck(marius_mt19937_randperm_host((uint32_t*)state_host_.data_ptr<int32_t>(), out.data_ptr<int64_t>(), n));
    return out;
}
void MariusGenerator::fill_pool(int i, torch::Device dev) {
    Pool& p = pools_[i];
    hipStream_t side = (hipStream_t)side_stream_;
    if (p.has_done) HIPCHECK(hipStreamWaitEvent(side, (hipEvent_t)p.done, 0));  // previous contents fully consumed
    HIPCHECK(hipMemcpyAsync(p.state_before.data_ptr(), state_dev_.data_ptr(), MARIUS_MT_STATE_WORDS * 4, hipMemcpyDeviceToDevice, side));
    mcheck(marius_mt19937_fill((uint32_t*)state_dev_.data_ptr<int32_t>(), (uint32_t*)p.buf.data_ptr<int32_t>(), p.size, (marius_stream_t)side));
    HIPCHECK(hipEventRecord((hipEvent_t)p.ready, side));
    p.filled = true;
    p.waited = false;
    p.used = 0;
    p.has_done = false;
    (void)dev;
}
Tensor MariusGenerator::raw_words(int64_t n, torch::Device dev) {
    to_device(dev);
    hipStream_t main = (hipStream_t)cur_stream();
    if (!prefetch_) {
        Tensor out = torch::empty({n}, i32(dev));
        mcheck(marius_mt19937_fill((uint32_t*)state_dev_.data_ptr<int32_t>(), (uint32_t*)out.data_ptr<int32_t>(), n, (marius_stream_t)main));
        return out;
    }
    if (!side_stream_) side_stream_ = aux_stream(dev.index(), AUX_FILL);  // (shared by every generator of the process; never destroyed)
    if (!side_ordered_) {
        // the state upload / earlier direct fills were enqueued on the main stream: order the fill stream (own or the caller's) after them
        hipEvent_t e;
        HIPCHECK(hipEventCreateWithFlags(&e, order_event_flags()));
        HIPCHECK(hipEventRecord(e, main));
        HIPCHECK(hipStreamWaitEvent((hipStream_t)side_stream_, e, 0));
        HIPCHECK(hipEventDestroy(e));
        side_ordered_ = true;
    }
    const int64_t want = std::max<int64_t>(n * pool_requests_, 1 << 16);
    for (int i = 0; i < 2; ++i) {
        Pool& p = pools_[i];
        if (!p.buf.defined() || p.size < n || p.buf.device() != dev) {
            if (p.filled) throw MariusRuntimeException("MariusGenerator: request larger than the run-ahead pool");
            p.buf = torch::empty({want}, i32(dev));
            p.state_before = torch::empty({MARIUS_MT_STATE_WORDS}, i32(dev));
            p.size = want;
            if (!p.ready) {
                hipEvent_t a, b;
                HIPCHECK(hipEventCreateWithFlags(&a, order_event_flags()));
                HIPCHECK(hipEventCreateWithFlags(&b, order_event_flags()));
                p.ready = a;
                p.done = b;
            }
        }
    }
    if (!pools_[cur_].filled) {  // cold start: fill the current pool and the next one
        fill_pool(cur_, dev);
        fill_pool(cur_ ^ 1, dev);
    }
    auto take = [&](int64_t cnt) {
        Pool& p = pools_[cur_];
        if (!p.waited) {
            HIPCHECK(hipStreamWaitEvent(main, (hipEvent_t)p.ready, 0));
            p.waited = true;
        }
        Tensor v = p.buf.narrow(0, p.used, cnt);
        p.used += cnt;
        return v;
    };
    auto advance = [&]() {  // current pool exhausted: hand it back to the producer and move on
        Pool& p = pools_[cur_];
        HIPCHECK(hipEventRecord((hipEvent_t)p.done, main));
        p.has_done = true;
        p.filled = false;
        const int old = cur_;
        cur_ ^= 1;
        if (!pools_[cur_].filled) fill_pool(cur_, dev);
        fill_pool(old, dev);  // becomes the pool after the one now current
    };
    // A pool that the previous request exhausted exactly is handed back only NOW: its `done` event must follow the kernel that reads
    // the view returned last time in stream order, and that kernel is enqueued by the caller after raw_words returns (recording the
    // event right away let the producer overwrite the tail of the buffer under the consumer).
    if (pools_[cur_].filled && pools_[cur_].used == pools_[cur_].size) advance();
    Pool& p = pools_[cur_];
    if (p.size - p.used >= n) return take(n);
    // request straddles two pools
    const int64_t first = p.size - p.used;
    Tensor a = first > 0 ? take(first).clone() : Tensor();
    advance();
    Tensor b = take(n - first);
    return first > 0 ? torch::cat({a, b}) : b;
}

// ------------------------------------------------------------------------------------------------ storage
InMemory::InMemory(std::string filename, int64_t dim0_size, int64_t dim1_size, torch::Dtype dtype, torch::Device device) {
    filename_ = std::move(filename);
    dim0_size_ = dim0_size;
    dim1_size_ = dim1_size;
    dtype_ = dtype;
    device_ = device;
}
InMemory::InMemory(Tensor data) {
    require_device(data, "InMemory");
    data_ = data;
    dim0_size_ = data.size(0);
    dim1_size_ = data.dim() > 1 ? data.size(1) : 1;
    dtype_ = data.scalar_type();
    device_ = data.device();
    loaded_ = true;
}
void InMemory::load() {  // storage.cpp:547-573: pread the raw row-major file, then .to(device)
    if (loaded_) return;
    Tensor host = torch::empty({dim0_size_, dim1_size_}, torch::TensorOptions().dtype(dtype_));
    std::ifstream f(filename_, std::ios::binary);
    if (!f) throw std::runtime_error("");  // storage.cpp:179-201 behaviour: bare runtime_error after logging
    f.read(reinterpret_cast<char*>(host.data_ptr()), (std::streamsize)host.nbytes());
    if (!f) throw std::runtime_error("");
    data_ = host.to(device_);
    loaded_ = true;
}
void InMemory::write() {
    if (!loaded_ || filename_.empty()) return;
    Tensor host = data_.cpu().contiguous();
    std::ofstream f(filename_, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("");
    f.write(reinterpret_cast<const char*>(host.data_ptr()), (std::streamsize)host.nbytes());
}
void InMemory::unload(bool perform_write) {
    if (perform_write) write();
    data_ = Tensor();
    loaded_ = false;
}
Tensor InMemory::indexRead(Tensor indices) {  // storage.cpp:606-649
    if (indices.sizes().size() != 1) throw std::runtime_error("");
    require_device(data_, "indexRead");
    require_device(indices, "indexRead");
    if (dtype_ != torch::kFloat32) return data_.index_select(0, indices);  // edge lists: not on the float path
    Tensor out = torch::empty({indices.size(0), dim1_size_}, data_.options());
    mcheck(marius_gather_rows(fp(data_), data_.stride(0), ip(indices), indices.size(0), (int32_t)dim1_size_, fp(out), out.stride(0), cur_stream()));
    return out;
}
Tensor InMemory::indexReadCounted(Tensor indices, Tensor count_dev) {
    if (indices.sizes().size() != 1) throw std::runtime_error("");
    require_device(data_, "indexRead");
    require_device(indices, "indexRead");
    Tensor out = torch::empty({indices.size(0), dim1_size_}, data_.options());
    mcheck(marius_gather_rows_counted(fp(data_), data_.stride(0), ip(indices), indices.size(0), ip(count_dev), (int32_t)dim1_size_, fp(out), out.stride(0),
                                      cur_stream()));
    return out;
}
void InMemory::indexAdd(Tensor indices, Tensor values) {  // storage.cpp:651-673 (ids unique)
    if (!values.defined() || indices.sizes().size() != 1 || values.size(0) != indices.size(0) || data_.dim() != values.dim())
        throw std::runtime_error("");
    require_device(data_, "indexAdd");
    mcheck(marius_scatter_add_rows(fp(data_), data_.stride(0), ip(indices), indices.size(0), (int32_t)dim1_size_, fp(values), values.stride(0),
                                   cur_stream()));
    // written through a raw pointer: tell ATen's version counter, which is what a model that tracks this table's magnitude bound looks at
    // (Model::Scanned; every other write path of this class is an ATen op and bumps it by itself)
    data_.unsafeGetTensorImpl()->bump_version();
}
void Storage::readPartitionSizes(const std::string& filename) {
    std::ifstream f(filename);
    if (!f) throw std::runtime_error("");
    edge_bucket_sizes_.clear();
    int64_t v;
    while (f >> v) edge_bucket_sizes_.push_back(v);
}
static void permute_rows(Tensor& data, const std::vector<int64_t>& buckets, const std::function<Tensor(const Tensor&)>& order) {
    if (buckets.empty()) {
        data.copy_(data.index_select(0, order(data)));
        return;
    }
    int64_t start = 0;
    for (int64_t n : buckets) {
        if (n > 1) {
            Tensor b = data.narrow(0, start, n);
            b.copy_(b.index_select(0, order(b)));
        }
        start += n;
    }
}
void InMemory::shuffle() {  // storage.cpp:709-737: randperm per edge bucket
    if (!loaded_) load();
    // the reference draws from the one global generator its whole run uses; here that stream is the MariusGenerator handed to
    // setGenerator (host randperm, same MT19937 sequence as torch::randperm on the CPU); without one, torch's device generator
    auto gen = generator_;
    permute_rows(data_, edge_bucket_sizes_, [gen](const Tensor& b) {
        if (gen) return gen->randperm(b.size(0)).to(b.device());
        return torch::randperm(b.size(0), torch::TensorOptions().dtype(torch::kInt64).device(b.device()));
    });
}
void InMemory::sort(bool src) {  // storage.cpp:739-766: stable, so equal keys keep their file order (reproducible bucket contents)
    if (!loaded_) load();
    permute_rows(data_, edge_bucket_sizes_, [src](const Tensor& b) { return torch::argsort(b.select(1, src ? 0 : -1), /*stable=*/true, /*dim=*/0, /*descending=*/false); });
}
Tensor InMemory::range(int64_t offset, int64_t n) {
    if (!data_.defined()) throw std::runtime_error("");
    return data_.narrow(0, offset, n);
}
void InMemory::indexPut(Tensor indices, Tensor values) { data_.index_copy_(0, indices, values); }
void InMemory::rangePut(int64_t offset, Tensor values) { data_.narrow(0, offset, values.size(0)).copy_(values); }

// ------------------------------------------------------------------------------------------------ negative sampler
CorruptNodeNegativeSampler::CorruptNodeNegativeSampler(int num_chunks, int num_negatives, float degree_fraction, bool filtered,
                                                       LocalFilterMode local_filter_mode, shared_ptr<MariusGenerator> generator)
    : num_chunks_(num_chunks), num_negatives_(num_negatives), degree_fraction_(degree_fraction), filtered_(filtered),
      local_filter_mode_(local_filter_mode), generator_(generator) {
    if (filtered_) {  // negative.cpp:321-325
        num_chunks_ = 1;
        num_negatives_ = -1;
        degree_fraction_ = 0.0;
    }
    if (!generator_) generator_ = std::make_shared<MariusGenerator>(0);
}

std::tuple<Tensor, Tensor> CorruptNodeNegativeSampler::getNegatives(shared_ptr<MariusGraph> graph, Tensor edges, bool inverse) {
    require_device(edges, "getNegatives");
    auto dev = edges.device();
    const int64_t num_nodes = graph->num_nodes_in_memory_;
    if (num_negatives_ == -1) {  // filtered evaluation: every node is a negative (negative.cpp:354-356), true edges are masked
        Tensor ids = torch::arange(num_nodes, i64(dev)).unsqueeze(0);
        return std::forward_as_tuple(ids, compute_filter_corruption_global(graph, edges, inverse));
    }
    // negative.cpp:198-200, 295-301: on device edges only the DEG local filter exists; every other local mode ends in this exception
    if (local_filter_mode_ != LocalFilterMode::DEG)
        throw MariusRuntimeException("Local filtering against all edges in the batch not yet supported on GPU.");
    const int n_deg = (int)(num_negatives_ * degree_fraction_);
    const int64_t B = edges.size(0);
    const int64_t words = marius_negatives_raw_words(num_nodes, B, num_chunks_, num_negatives_, n_deg);
    Tensor raw = generator_->raw_words(words, dev);
    Tensor ids = torch::empty({num_chunks_, num_negatives_}, i64(dev));
    Tensor deg = n_deg > 0 ? torch::empty({num_chunks_, n_deg}, i64(dev)) : Tensor();
    mcheck(marius_sample_negatives((const uint32_t*)raw.data_ptr<int32_t>(), ip(edges), B, (int32_t)edges.size(1), inverse ? 1 : 0, num_nodes,
                                   num_chunks_, num_negatives_, n_deg, ip(ids), ip(deg), cur_stream()));
    Tensor filter = torch::empty({0, 2}, i64(dev));
    if (n_deg > 0 && local_filter_mode_ == LocalFilterMode::DEG) {
        filter = torch::empty({(int64_t)num_chunks_ * n_deg, 2}, i64(dev));
        mcheck(marius_deg_filter(ip(deg), num_chunks_, n_deg, B, ip(filter), cur_stream()));
        if (compact_filter_) filter = filter.index({filter.select(1, 0) >= 0});  // the reference's tensor, in its nonzero() order
    } else if (n_deg > 0) {
        throw MariusRuntimeException("Local filtering against all edges in the batch not yet supported on GPU.");  // negative.cpp:301
    }
    return std::forward_as_tuple(ids, filter);
}

void MariusGraph::sortAllEdges(Tensor all_edges) {  // graph.cpp:233-236
    all_src_sorted_edges_ = all_edges.index_select(0, all_edges.select(1, 0).argsort(0, false)).to(torch::kInt64);
    all_dst_sorted_edges_ = all_edges.index_select(0, all_edges.select(1, -1).argsort(0, false)).to(torch::kInt64);
}

// negative.cpp:50-293, global branch.  The reference builds the pairs with a chain of libtorch ops on the GPU; here two small kernels
// (count, emit) walk the sorted edge list per batch edge (eval_filter.hip); one 8-byte read-back sizes the output tensor.
Tensor compute_filter_corruption_global(shared_ptr<MariusGraph> graph, Tensor edges, bool inverse) {
    if (edges.dim() == 3) edges = edges.flatten(0, 1);
    else if (edges.dim() != 2) throw TensorSizeMismatchException(edges, "Edge list must have three (if chunked) or two dimensions");
    if (edges.size(-1) != 3 && edges.size(-1) != 2) throw TensorSizeMismatchException(edges, "Edge list tensor must have 3 or 2 columns.");
    require_device(edges, "compute_filter_corruption");
    Tensor all_sorted_edges = inverse ? graph->all_dst_sorted_edges_ : graph->all_src_sorted_edges_;
    if (!all_sorted_edges.defined()) throw MariusRuntimeException("filtered evaluation needs MariusGraph::sortAllEdges (all known edges) first");
    if (all_sorted_edges.size(1) != edges.size(1)) throw TensorSizeMismatchException(all_sorted_edges, "sorted edge list and batch edges differ in column count");
    edges = edges.contiguous();
    all_sorted_edges = all_sorted_edges.contiguous();
    const int64_t B = edges.size(0);
    auto dev = edges.device();
    Tensor counts = torch::empty({std::max<int64_t>(B, 1)}, i64(dev)), offsets = torch::empty({B + 1}, i64(dev));
    mcheck(marius_true_edge_filter_offsets(ip(all_sorted_edges), all_sorted_edges.size(0), (int32_t)edges.size(1), inverse ? 1 : 0, ip(edges), B, ip(counts),
                                           ip(offsets), cur_stream()));
    const int64_t F = offsets[B].item<int64_t>();
    Tensor filter = torch::empty({F, 2}, i64(dev));
    if (F > 0)
        mcheck(marius_true_edge_filter_emit(ip(all_sorted_edges), all_sorted_edges.size(0), (int32_t)edges.size(1), inverse ? 1 : 0, ip(edges), B, ip(offsets),
                                            ip(filter), cur_stream()));
    return filter;
}

// ------------------------------------------------------------------------------------------------ batch
void Batch::accumulateGradients(float learning_rate) {
    if (node_embeddings_.defined()) {
        if (!node_embeddings_grad_.defined()) throw UndefinedTensorException();
        node_gradients_ = torch::empty_like(node_embeddings_grad_);
        node_state_update_ = torch::empty_like(node_embeddings_grad_);
        mcheck(marius_adagrad_rule(fp(node_embeddings_grad_), fp(node_embeddings_state_), fp(node_gradients_), fp(node_state_update_),
                                   node_embeddings_grad_.numel(), learning_rate, 1e-10f, cur_stream()));
    }
    node_embeddings_state_ = Tensor();
}
void Batch::clear() {
    unique_node_indices_ = node_embeddings_ = node_embeddings_grad_ = node_gradients_ = node_state_update_ = node_embeddings_state_ = Tensor();
    edges_ = global_edges_ = table_ = src_neg_indices_ = dst_neg_indices_ = src_neg_indices_mapping_ = dst_neg_indices_mapping_ = Tensor();
    src_neg_filter_ = dst_neg_filter_ = occ_perm_ = occ_inverse_ = occ_seg_offsets_ = num_unique_dev_ = Tensor();
    rel_uniq_ = rel_inverse_ = rel_perm_ = rel_seg_ = rel_count_ = occ_plan_ = rel_plan_ = Tensor();
    row_bound_ = table_to_update_ = Tensor();
}

// ------------------------------------------------------------------------------------------------ LP context / fused decoder calls
Tensor LpContext::view(size_t off, std::vector<int64_t> shape, std::vector<int64_t> strides) const {
    int64_t n = 1;
    if (strides.empty()) {
        for (auto s : shape) n *= s;
        return workspace.narrow(0, (int64_t)off, n * 4).view(torch::kFloat32).view(shape);
    }
    int64_t span = 1;
    for (size_t i = 0; i < shape.size(); ++i) span += (shape[i] - 1) * strides[i];
    return workspace.narrow(0, (int64_t)off, span * 4).view(torch::kFloat32).as_strided(shape, strides);
}

static void lp_setup(LpContext& ctx, shared_ptr<EdgeDecoder> dec, const Tensor& edges, const Tensor& emb, const Tensor& dst_negs, const Tensor& src_negs,
                     const Tensor& dst_filter, const Tensor& src_filter, LossReduction reduction, int loss_kind = MARIUS_LOSS_SOFTMAX_CE, float margin = 0.f,
                     int lp_flags = 0) {
    require_device(emb, "forward_lp");
    require_device(edges, "forward_lp");
    if (edges.dim() != 2 || (edges.size(1) != 3 && edges.size(1) != 2))
        throw TensorSizeMismatchException(edges, "Edge list must be a 3 or 2 column tensor");
    marius_lp_desc& d = ctx.desc;
    d = marius_lp_desc{};
    const bool has_rel = edges.size(1) == 3;
    d.relop = has_rel ? dec->relation_operator_->kind() : MARIUS_OP_NOOP;
    d.cmp = dec->comparator_->kind();
    d.d = (int32_t)emb.size(1);
    d.edge_cols = (int32_t)edges.size(1);
    d.B = edges.size(0);
    d.C = (int32_t)dst_negs.size(0);
    d.N = (int32_t)dst_negs.size(1);
    d.use_inverse = (has_rel && dec->use_inverse_relations_ && dec->inverse_relations_.defined() && src_negs.defined()) ? 1 : 0;
    d.reduction = reduction == LossReduction::MEAN ? MARIUS_REDUCE_MEAN : MARIUS_REDUCE_SUM;
    d.loss = loss_kind;
    d.margin = margin;
    d.flags = lp_flags;
    d.free_cus = (lp_flags & MARIUS_LP_TRAIN_ONLY) ? ctx.free_cus : 0;
    d.absmax = (ctx.absmax.defined() && (lp_flags & MARIUS_LP_TRAIN_ONLY)) ? ctx.absmax.data_ptr<float>() : nullptr;
    d.absmax_rel = (d.absmax && ctx.absmax_rel.defined()) ? ctx.absmax_rel.data_ptr<float>() : nullptr;
    Tensor e = edges.contiguous(), dn = dst_negs.contiguous(), sn = src_negs.defined() ? src_negs.contiguous() : Tensor();
    d.emb = fp(emb);
    d.emb_ld = emb.stride(0);
    d.U = emb.size(0);
    d.edges = ip(e);
    d.dst_neg = ip(dn);
    d.src_neg = ip(sn);
    d.rel = has_rel ? fp(dec->relations_) : nullptr;
    d.inv_rel = d.use_inverse ? fp(dec->inverse_relations_) : nullptr;
    d.rel_ld = has_rel ? dec->relations_.stride(0) : 0;
    d.R = has_rel ? dec->relations_.size(0) : 0;
    Tensor df = (dst_filter.defined() && dst_filter.numel() > 0) ? dst_filter.contiguous() : Tensor();
    Tensor sf = (src_filter.defined() && src_filter.numel() > 0) ? src_filter.contiguous() : Tensor();
    d.dst_filter = ip(df);
    d.n_dst_filter = df.defined() ? df.size(0) : 0;
    d.src_filter = ip(sf);
    d.n_src_filter = sf.defined() ? sf.size(0) : 0;
    ctx.keep = {emb, e, dn, sn, df, sf};
    mcheck(marius_lp_plan(&d, &ctx.layout));
    if (!ctx.workspace.defined() || (size_t)ctx.workspace.numel() < ctx.layout.total_bytes || ctx.workspace.device() != emb.device())
        ctx.workspace = torch::empty({(int64_t)ctx.layout.total_bytes}, torch::TensorOptions().dtype(torch::kUInt8).device(emb.device()));
    ctx.has_loss = false;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> node_corrupt_forward(shared_ptr<EdgeDecoder> decoder, Tensor positive_edges, Tensor node_embeddings,
                                                                Tensor dst_negs, Tensor src_negs, LpContext* ctx, Tensor dst_filter, Tensor src_filter,
                                                                LossReduction reduction, int loss_kind, float margin, int lp_flags) {
    LpContext local;
    LpContext& c = ctx ? *ctx : local;
    lp_setup(c, decoder, positive_edges, node_embeddings, dst_negs, src_negs, dst_filter, src_filter, reduction, loss_kind, margin, lp_flags);
    mcheck(marius_lp_forward(&c.desc, &c.layout, c.workspace.data_ptr(), cur_stream()));
    const int64_t Bp = c.layout.Bp, N = c.desc.N, nld = c.layout.n_ld;
    const bool have_neg = !(c.layout.flash && c.layout.neg[0] == 0);  // the flash path keeps row statistics only
    Tensor pos = c.view(c.layout.pos[0], {Bp});
    Tensor neg = have_neg ? c.view(c.layout.neg[0], {Bp, N}, {nld, 1}) : Tensor();
    Tensor inv_pos, inv_neg;
    if (c.desc.use_inverse) {
        inv_pos = c.view(c.layout.pos[1], {Bp});
        if (have_neg) inv_neg = c.view(c.layout.neg[1], {Bp, N}, {nld, 1});
    }
    if (!ctx) {  // the workspace dies with `local`: hand out owning copies
        pos = pos.clone();
        if (neg.defined()) neg = neg.clone();
        if (inv_pos.defined()) {
            inv_pos = inv_pos.clone();
            if (inv_neg.defined()) inv_neg = inv_neg.clone();
        }
    }
    return std::forward_as_tuple(pos, neg, inv_pos, inv_neg);
}

std::tuple<Tensor, Tensor> only_pos_forward(shared_ptr<EdgeDecoder> decoder, Tensor edges, Tensor node_embeddings) {
    // decoder_methods.cpp:7-42: same positive-score path; run the fused forward with one dummy negative per direction
    auto dev = node_embeddings.device();
    Tensor dummy = torch::zeros({1, 1}, i64(dev));
    auto t = node_corrupt_forward(decoder, edges, node_embeddings, dummy, dummy, nullptr);
    const int64_t B = edges.size(0);
    Tensor pos = std::get<0>(t).narrow(0, 0, B);
    Tensor inv = std::get<2>(t);
    if (inv.defined()) inv = inv.narrow(0, 0, B);
    return std::forward_as_tuple(pos, inv);
}

// API-level operator calls (not on the fused training path).  Plain tensors: expressed through the same fused forward so that they,
// too, run on the HIP kernels only.  Tensors that require grad (the generic path taken when a user plug-in is part of the model): the
// reference's own libtorch expressions, so that autograd sees them.
static Tensor relop_device(int kind, const Tensor& embs, const Tensor& rels) {
    if (!rels.defined()) return embs;
    require_device(embs, "RelationOperator");
    const int64_t B = embs.size(0);
    auto dev = embs.device();
    struct D : EdgeDecoder {};
    auto dec = std::make_shared<D>();
    dec->comparator_ = std::make_shared<DotCompare>();
    dec->relations_ = rels.contiguous();
    dec->use_inverse_relations_ = false;
    struct Op : RelationOperator {
        int k;
        int kind() const override { return k; }
        Tensor operator()(const Tensor& e, const Tensor&) override { return e; }
    };
    auto op = std::make_shared<Op>();
    op->k = kind;
    dec->relation_operator_ = op;
    Tensor idx = torch::arange(B, i64(dev));
    Tensor edges = torch::stack({idx, idx, idx}, 1);
    Tensor dummy = torch::zeros({1, 1}, i64(dev));
    LpContext ctx;
    node_corrupt_forward(dec, edges, embs.contiguous(), dummy, Tensor(), &ctx);
    return ctx.view(ctx.layout.adj[0], {B, embs.size(1)}, {ctx.layout.d_ld, 1}).clone();
}
static bool wants_grad(const Tensor& a, const Tensor& b) {
    return torch::GradMode::is_enabled() && ((a.defined() && a.requires_grad()) || (b.defined() && b.requires_grad()));
}
Tensor HadamardOperator::operator()(const Tensor& embs, const Tensor& rels) {  // relation_operators.cpp:7-12
    if (!rels.defined()) return embs;
    return wants_grad(embs, rels) ? embs * rels : relop_device(kind(), embs, rels);
}
Tensor ComplexHadamardOperator::operator()(const Tensor& embs, const Tensor& rels) {  // relation_operators.cpp:14-34
    if (!rels.defined()) return embs;
    if (!wants_grad(embs, rels)) return relop_device(kind(), embs, rels);
    const int64_t h = embs.size(1) / 2;
    Tensor re = embs.narrow(1, 0, h), im = embs.narrow(1, h, h), rre = rels.narrow(1, 0, h), rim = rels.narrow(1, h, h);
    return torch::cat({re * rre - im * rim, re * rim + im * rre}, 1);
}
Tensor TranslationOperator::operator()(const Tensor& embs, const Tensor& rels) {  // relation_operators.cpp:36-41
    if (!rels.defined()) return embs;
    return wants_grad(embs, rels) ? embs + rels : relop_device(kind(), embs, rels);
}
Tensor NoOp::operator()(const Tensor& embs, const Tensor&) { return embs; }

Tensor pad_and_reshape(Tensor input, int num_chunks) {  // comparators.cpp:7-20
    const int64_t num_pos = input.size(0);
    const int64_t per = (int64_t)std::ceil((float)num_pos / num_chunks);
    if (per * num_chunks != num_pos) {
        const int64_t new_size = per * num_chunks;
        Tensor pad = torch::zeros({new_size - num_pos, input.size(1)}, input.options());
        input = torch::cat({input, pad}, 0);
    }
    return input.view({num_chunks, per, input.size(1)});
}

static Tensor compare_device(int kind, Tensor src, Tensor dst) {
    if (!src.defined() || !dst.defined()) throw UndefinedTensorException();  // comparators.cpp:23-25
    require_device(src, "Comparator");
    auto dev = src.device();
    struct D : EdgeDecoder {};
    auto dec = std::make_shared<D>();
    dec->relation_operator_ = std::make_shared<NoOp>();
    struct Cmp : Comparator {
        int k;
        int kind() const override { return k; }
        Tensor operator()(Tensor, Tensor) override { return Tensor(); }
    };
    auto cmp = std::make_shared<Cmp>();
    cmp->k = kind;
    dec->comparator_ = cmp;
    dec->use_inverse_relations_ = false;
    const int64_t B = src.size(0);
    Tensor idx = torch::arange(B, i64(dev));
    if (src.sizes() == dst.sizes()) {
        Tensor emb = torch::cat({src, dst}, 0).contiguous();
        Tensor edges = torch::stack({idx, idx + B}, 1);
        Tensor dummy = torch::zeros({1, 1}, i64(dev));
        auto t = node_corrupt_forward(dec, edges, emb, dummy, Tensor(), nullptr);
        return std::get<0>(t).narrow(0, 0, B);
    }
    const int64_t C = dst.size(0), N = dst.size(1);
    Tensor emb = torch::cat({src, dst.reshape({C * N, dst.size(2)})}, 0).contiguous();
    Tensor edges = torch::stack({idx, idx}, 1);
    Tensor negs = (torch::arange(C * N, i64(dev)) + B).reshape({C, N});
    auto t = node_corrupt_forward(dec, edges, emb, negs, Tensor(), nullptr);
    return std::get<1>(t);
}
Tensor DotCompare::operator()(Tensor src, Tensor dst) {  // comparators.cpp:22-28
    if (!src.defined() || !dst.defined()) throw UndefinedTensorException();
    if (!wants_grad(src, dst)) return compare_device(kind(), src, dst);
    if (src.sizes() == dst.sizes()) return (src * dst).sum(-1);
    Tensor s = pad_and_reshape(src, (int)dst.size(0));
    return s.bmm(dst.transpose(-1, -2)).flatten(0, 1);
}
Tensor L2Compare::operator()(Tensor src, Tensor dst) {  // comparators.cpp:30-41
    if (!src.defined() || !dst.defined()) throw UndefinedTensorException();
    if (!wants_grad(src, dst)) return compare_device(kind(), src, dst);
    if (src.sizes() == dst.sizes()) return torch::pairwise_distance(src, dst);
    Tensor s = pad_and_reshape(src, (int)dst.size(0));
    Tensor x2 = s.pow(2).sum(2).unsqueeze(2), y2 = dst.pow(2).sum(2).unsqueeze(1);
    Tensor xy = s.bmm(dst.transpose(1, 2));
    return torch::clamp_min(x2 + y2 - 2 * xy, 1e-8).sqrt().flatten(0, 1);
}
Tensor CosineCompare::operator()(Tensor src, Tensor dst) {  // comparators.cpp:43-60 (scores the tensors as given)
    if (!src.defined() || !dst.defined()) throw UndefinedTensorException();
    if (!wants_grad(src, dst)) return compare_device(kind(), src, dst);
    if (src.sizes() == dst.sizes()) return (src * dst).sum(-1);
    Tensor s = pad_and_reshape(src, (int)dst.size(0));
    return s.bmm(dst.transpose(-1, -2)).flatten(0, 1);
}

// ------------------------------------------------------------------------------------------------ edge decoders
Tensor EdgeDecoder::apply_relation(Tensor nodes, Tensor relations) { return (*relation_operator_)(nodes, relations); }
Tensor EdgeDecoder::compute_scores(Tensor src, Tensor dst) { return (*comparator_)(src, dst); }
Tensor EdgeDecoder::select_relations(Tensor indices, bool inverse) {  // edge_decoder.cpp:11-20
    Tensor& table = inverse ? inverse_relations_ : relations_;
    if (!table.defined()) throw UndefinedTensorException();
    Tensor out = torch::empty({indices.size(0), table.size(1)}, table.options());
    mcheck(marius_gather_rows(fp(table), table.stride(0), ip(indices), indices.size(0), (int32_t)table.size(1), fp(out), out.stride(0), cur_stream()));
    return out;
}
static void init_decoder(EdgeDecoder* d, int num_relations, int dim, torch::TensorOptions opts, bool inv, EdgeDecoderMethod m) {
    d->num_relations_ = num_relations;
    d->embedding_size_ = dim;
    d->tensor_options_ = opts;
    d->use_inverse_relations_ = inv;
    d->decoder_method_ = m;
    d->learning_task_ = LearningTask::LINK_PREDICTION;
}
DistMult::DistMult(int num_relations, int embedding_dim, torch::TensorOptions o, bool inv, EdgeDecoderMethod m) {
    comparator_ = std::make_shared<DotCompare>();
    relation_operator_ = std::make_shared<HadamardOperator>();
    init_decoder(this, num_relations, embedding_dim, o, inv, m);
    reset();
}
// register_parameter the first time (the constructor; Cloneable::clone() clears the dictionaries first), replace the tensor on a repeated
// reset() — the reference's reset() throws "Parameter already defined" there; re-initialising is the more useful superset
#define MARIUS_SET_PARAM(FIELD, NAME, VALUE)                                                     \
    do {                                                                                          \
        Tensor v_ = (VALUE);                                                                      \
        v_.set_requires_grad(true);                                                               \
        if (parameters_.contains(NAME)) FIELD = parameters_[NAME] = v_;                           \
        else FIELD = register_parameter(NAME, v_, /*requires_grad=*/true);                        \
    } while (0)
void DistMult::reset() {  // distmult.cpp:21-27
    MARIUS_SET_PARAM(relations_, "relation_embeddings", torch::ones({num_relations_, embedding_size_}, tensor_options_));
    if (use_inverse_relations_) MARIUS_SET_PARAM(inverse_relations_, "inverse_relation_embeddings", torch::ones({num_relations_, embedding_size_}, tensor_options_));
}
ComplEx::ComplEx(int num_relations, int embedding_dim, torch::TensorOptions o, bool inv, EdgeDecoderMethod m) {
    comparator_ = std::make_shared<DotCompare>();
    relation_operator_ = std::make_shared<ComplexHadamardOperator>();
    init_decoder(this, num_relations, embedding_dim, o, inv, m);
    reset();
}
void ComplEx::reset() {  // complex.cpp:21-29
    Tensor r = torch::zeros({num_relations_, embedding_size_}, tensor_options_);
    r.narrow(1, 0, embedding_size_ / 2).fill_(1);
    MARIUS_SET_PARAM(relations_, "relation_embeddings", r);
    if (use_inverse_relations_) {
        Tensor ir = torch::zeros({num_relations_, embedding_size_}, tensor_options_);
        ir.narrow(1, 0, embedding_size_ / 2).fill_(1);
        MARIUS_SET_PARAM(inverse_relations_, "inverse_relation_embeddings", ir);
    }
}
TransE::TransE(int num_relations, int embedding_dim, torch::TensorOptions o, bool inv, EdgeDecoderMethod m) {
    comparator_ = std::make_shared<L2Compare>();
    relation_operator_ = std::make_shared<TranslationOperator>();
    init_decoder(this, num_relations, embedding_dim, o, inv, m);
    reset();
}
void TransE::reset() {  // transe.cpp:21-28
    MARIUS_SET_PARAM(relations_, "relation_embeddings", torch::zeros({num_relations_, embedding_size_}, tensor_options_));
    if (use_inverse_relations_) MARIUS_SET_PARAM(inverse_relations_, "inverse_relation_embeddings", torch::zeros({num_relations_, embedding_size_}, tensor_options_));
}
#undef MARIUS_SET_PARAM
shared_ptr<EdgeDecoder> get_edge_decoder(DecoderType type, EdgeDecoderMethod method, int num_relations, int dim, torch::TensorOptions opts, bool inv) {
    switch (type) {  // model_helpers.h:23-38
        case DecoderType::DISTMULT: return std::make_shared<DistMult>(num_relations, dim, opts, inv, method);
        case DecoderType::TRANSE: return std::make_shared<TransE>(num_relations, dim, opts, inv, method);
        case DecoderType::COMPLEX: return std::make_shared<ComplEx>(num_relations, dim, opts, inv, method);
    }
    throw MariusRuntimeException("Decoder currently not supported.");
}

// ------------------------------------------------------------------------------------------------ loss / optimizers / reporter
// loss.cpp:37-187 as differentiable libtorch expressions (generic path only: a user plug-in elsewhere in the model)
static Tensor loss_autograd(int kind, float margin, LossReduction red, Tensor pos, Tensor neg) {
    namespace F = torch::nn::functional;
    const bool mean = red == LossReduction::MEAN;
    auto reduce = [&](Tensor t) { return mean ? t.mean() : t.sum(); };
    auto scores_labels = [&]() {  // scores_to_labels, loss.cpp:37-48
        Tensor y = torch::cat({pos, neg.flatten(0, 1)});
        Tensor l = torch::cat({torch::ones_like(pos), torch::zeros_like(neg.flatten(0, 1))});
        return std::make_pair(y, l);
    };
    switch (kind) {
        case MARIUS_LOSS_SOFTMAX_CE:
        case MARIUS_LOSS_CROSS_ENTROPY: {
            Tensor scores = torch::cat({pos.unsqueeze(1), neg.logsumexp(1, true)}, 1);
            Tensor labels = torch::zeros({pos.size(0)}, torch::TensorOptions().dtype(torch::kInt64).device(pos.device()));
            if (mean) return F::cross_entropy(scores, labels, F::CrossEntropyFuncOptions().reduction(torch::kMean));
            return F::cross_entropy(scores, labels, F::CrossEntropyFuncOptions().reduction(torch::kSum));
        }
        case MARIUS_LOSS_RANKING:
            return reduce(torch::relu(neg - pos.unsqueeze(1) + margin));
        case MARIUS_LOSS_BCE_AFTER_SIGMOID: {
            auto yl = scores_labels();
            return reduce(F::binary_cross_entropy(yl.first.sigmoid(), yl.second, F::BinaryCrossEntropyFuncOptions().reduction(torch::kNone)));
        }
        case MARIUS_LOSS_BCE_WITH_LOGITS: {
            auto yl = scores_labels();
            return reduce(F::binary_cross_entropy_with_logits(yl.first, yl.second, F::BinaryCrossEntropyWithLogitsFuncOptions().reduction(torch::kNone)));
        }
        case MARIUS_LOSS_MSE: {
            auto yl = scores_labels();
            return reduce((yl.first - yl.second).pow(2));
        }
        case MARIUS_LOSS_SOFTPLUS: {
            auto yl = scores_labels();
            return reduce(F::softplus(-(2 * yl.second - 1) * yl.first));
        }
    }
    throw MariusRuntimeException("LossFunction: a user-defined loss must override operator()");
}

Tensor LossFunction::operator()(Tensor pos, Tensor neg, bool scores) {
    if (!scores) {
        if (kind() == MARIUS_LOSS_SOFTMAX_CE)
            throw MariusRuntimeException(
                "Input to SoftmaxCrossEntropy loss function must be scores. SoftmaxCrossEntropy is currently unsupported for classification.");
        if (kind() == MARIUS_LOSS_RANKING)
            throw MariusRuntimeException("Input to ranking loss function must be scores. This loss function is unsupported for classification.");
        throw MariusRuntimeException(std::string(name()) + ": classification input (scores = false) is outside the link-prediction path of this build");
    }
    if (!pos.defined() || !neg.defined()) throw UndefinedTensorException();                    // check_score_shapes, loss.cpp:7-29
    if (pos.dim() != 1) throw TensorSizeMismatchException(pos, "Positive scores should be 1-dimensional");
    if (neg.dim() != 2) throw TensorSizeMismatchException(neg, "Negative scores should be 2-dimensional");
    if (pos.size(0) != neg.size(0)) throw TensorSizeMismatchException(pos, "First dimension of pos_scores and neg_scores should match.");
    if (wants_grad(pos, neg)) return loss_autograd(kind(), margin(), reduction_type_, pos, neg);
    require_device(pos, name());
    require_device(neg, name());
    Tensor n = neg;
    if (n.stride(1) != 1 || n.stride(0) % 4 != 0) {  // give the kernel a 16-B aligned row pitch
        const int64_t nld = (neg.size(1) + 3) / 4 * 4;
        Tensor buf = torch::empty({neg.size(0), nld}, neg.options());
        buf.narrow(1, 0, neg.size(1)).copy_(neg);
        n = buf.narrow(1, 0, neg.size(1));
    }
    Tensor p = pos.contiguous();
    Tensor scratch = torch::empty({2 * std::max<int64_t>(p.size(0), 1)}, p.options()), loss = torch::empty({4}, p.options());
    mcheck(marius_loss_scores(kind(), margin(), fp(p), fp(n), p.size(0), (int32_t)n.size(1), n.stride(0),
                              reduction_type_ == LossReduction::MEAN ? MARIUS_REDUCE_MEAN : MARIUS_REDUCE_SUM, fp(scratch), fp(loss), cur_stream()));
    return loss[0];
}

shared_ptr<LossFunction> getLossFunction(const std::string& type, LossReduction reduction, float margin) {
    if (type == "SOFTMAX_CE") return std::make_shared<SoftmaxCrossEntropy>(reduction);
    if (type == "RANKING") return std::make_shared<RankingLoss>(reduction, margin);
    if (type == "CROSS_ENTROPY") return std::make_shared<CrossEntropyLoss>(reduction);
    if (type == "BCE_AFTER_SIGMOID") return std::make_shared<BCEAfterSigmoidLoss>(reduction);
    if (type == "BCE_WITH_LOGITS") return std::make_shared<BCEWithLogitsLoss>(reduction);
    if (type == "MSE") return std::make_shared<MSELoss>(reduction);
    if (type == "SOFTPLUS") return std::make_shared<SoftPlusLoss>(reduction);
    throw std::runtime_error("Unsupported loss function type");  // loss.cpp:207
}

void Optimizer::save(torch::serialize::OutputArchive& archive, const std::vector<std::string>& keys) {  // optim.cpp:25-40
    archive.write("num_steps", torch::IValue(num_steps()));
    auto slots = state_slots();
    for (size_t i = 0; i < keys.size() && i < params_.size(); ++i) {
        torch::serialize::OutputArchive tmp;
        for (auto& sl : slots) tmp.write(sl.first, (*sl.second)[i]);
        archive.write(keys[i], tmp);
    }
}
void Optimizer::load(torch::serialize::InputArchive& archive, const std::vector<std::string>& keys) {  // optim.cpp:7-23
    torch::IValue tmp;
    archive.read("num_steps", tmp);
    set_num_steps(tmp.toInt());
    auto slots = state_slots();
    for (size_t i = 0; i < keys.size() && i < params_.size(); ++i) {
        torch::serialize::InputArchive sub;
        archive.read(keys[i], sub);
        for (auto& sl : slots) {
            Tensor t;
            sub.read(sl.first, t);
            (*sl.second)[i].copy_(t);
        }
    }
}
void Optimizer::clear_grad() {
    for (auto& pg : params_) pg.second.zero_();
}
AdagradOptimizer::AdagradOptimizer(std::vector<std::pair<Tensor, Tensor>> params, float lr, float eps, float init_value) {
    params_ = std::move(params);
    learning_rate_ = lr;
    eps_ = eps;
    for (auto& pg : params_) state_.push_back(torch::full_like(pg.first, init_value));
}
void AdagradOptimizer::step() {  // optim.cpp:114-145
    for (size_t i = 0; i < params_.size(); ++i)
        mcheck(marius_dense_adagrad_step(fp(params_[i].first), fp(state_[i]), fp(params_[i].second), params_[i].first.numel(), learning_rate_, eps_,
                                         weight_decay_, cur_stream()));
    num_steps_++;
}
AdamOptimizer::AdamOptimizer(std::vector<std::pair<Tensor, Tensor>> params, float lr, float eps, float beta_1, float beta_2, float weight_decay, bool amsgrad) {
    params_ = std::move(params);
    learning_rate_ = lr;
    eps_ = eps;
    beta_1_ = beta_1;
    beta_2_ = beta_2;
    weight_decay_ = weight_decay;
    amsgrad_ = amsgrad;
    for (auto& pg : params_) {  // reset_state (optim.cpp:160-184)
        state_.push_back(torch::zeros_like(pg.first));
        exp_avg_sq_.push_back(torch::zeros_like(pg.first));
        if (amsgrad_) max_exp_avg_sq_.push_back(torch::zeros_like(pg.first));
    }
}
void AdamOptimizer::step() {  // optim.cpp:186-232
    for (size_t i = 0; i < params_.size(); ++i)
        mcheck(marius_dense_adam_step(fp(params_[i].first), fp(state_[i]), fp(exp_avg_sq_[i]), amsgrad_ ? fp(max_exp_avg_sq_[i]) : nullptr,
                                      fp(params_[i].second), params_[i].first.numel(), learning_rate_, beta_1_, beta_2_, eps_, weight_decay_, num_steps_,
                                      cur_stream()));
    num_steps_++;
}
SGDOptimizer::SGDOptimizer(std::vector<std::pair<Tensor, Tensor>> params, float lr) {
    params_ = std::move(params);
    learning_rate_ = lr;
}
void SGDOptimizer::step() {  // optim.cpp:59-79: param -= lr * grad (plain libtorch elementwise op, as in the reference)
    torch::NoGradGuard ng;  // the parameters are leaves that require grad (distmult.cpp:21-27): in-place steps belong outside autograd
    for (auto& pg : params_) pg.first.add_(pg.second, -learning_rate_);
}

Tensor LinkPredictionReporter::computeRanks(Tensor pos, Tensor neg) {
    require_device(pos, "computeRanks");
    Tensor p = pos.contiguous();
    Tensor n = neg.stride(1) == 1 ? neg : neg.contiguous();
    Tensor ranks = torch::empty({p.size(0)}, i64(p.device()));
    mcheck(marius_compute_ranks(fp(p), fp(n), p.size(0), (int32_t)n.size(1), n.stride(0), ip(ranks), cur_stream()));
    return ranks;
}
void LinkPredictionReporter::addResult(Tensor pos, Tensor neg) { ranks_.push_back(computeRanks(pos, neg)); }
std::vector<double> LinkPredictionReporter::report() {  // reporting.cpp:11-31, model.cpp:29-38
    if (ranks_.empty()) return {};
    Tensor r = torch::cat(ranks_).to(torch::kFloat64).cpu();
    std::vector<double> out;
    out.push_back(r.reciprocal().mean().item<double>());
    out.push_back(r.mean().item<double>());
    for (int k : {1, 3, 5, 10, 50, 100}) out.push_back(r.le(k).to(torch::kFloat64).mean().item<double>());
    return out;
}

// ------------------------------------------------------------------------------------------------ model
Model::Model(shared_ptr<EdgeDecoder> decoder, shared_ptr<LossFunction> loss, shared_ptr<LinkPredictionReporter> reporter, torch::Device device)
    : decoder_(decoder), loss_function_(loss), reporter_(reporter), device_(device) {
    if (decoder_->relations_.defined()) relations_grad_ = torch::zeros_like(decoder_->relations_);
    if (decoder_->inverse_relations_.defined()) inverse_relations_grad_ = torch::zeros_like(decoder_->inverse_relations_);
    // model.cpp:52-57: the decoder is the submodule "decoder" (its reset() registered relation_embeddings / inverse_relation_embeddings).
    // A user decoder that is not a torch module keeps the round-2 form: its tables as parameters of the model itself.
    if (auto mod = std::dynamic_pointer_cast<torch::nn::Module>(decoder_)) {
        register_module("decoder", mod);
    } else {
        if (decoder_->relations_.defined()) register_parameter("relation_embeddings", decoder_->relations_, /*requires_grad=*/true);
        if (decoder_->inverse_relations_.defined()) register_parameter("inverse_relation_embeddings", decoder_->inverse_relations_, /*requires_grad=*/true);
    }
    learning_task_ = decoder_->learning_task_;
    devices_ = {device};
}
// ------------------------------------------------------------------------------------------------ encoder (one embedding layer + post-hook)
GeneralEncoder::GeneralEncoder(int output_dim, bool bias, ActivationFunction activation, torch::Device device, Tensor bias_init)
    : activation_(activation), output_dim_(output_dim) {
    if (output_dim < 1) throw MariusRuntimeException("GeneralEncoder: output_dim must be positive");
    if (bias) {  // Layer::init_bias (layer.cpp:18-22): registered as "bias", requires grad
        Tensor b = bias_init.defined() ? bias_init.to(device, torch::kFloat32).reshape({output_dim}).contiguous().clone()
                                       : torch::zeros({output_dim}, torch::TensorOptions().dtype(torch::kFloat32).device(device));
        bias_ = register_parameter("bias", b.set_requires_grad(true));
        bias_grad_ = torch::zeros_like(b);
    }
}

Tensor GeneralEncoder::forward(Tensor embeddings) {
    if (!embeddings.defined()) throw MariusRuntimeException("Encoder requires embeddings and/or features as input");  // encoder.cpp:210
    if (!has_post_hook()) return embeddings;
    if (output_dim_ && embeddings.size(1) != output_dim_) throw TensorSizeMismatchException(embeddings, "embedding layer output_dim and the node rows differ in width");
    if (torch::GradMode::is_enabled() && (embeddings.requires_grad() || (bias_.defined() && bias_.grad_fn()))) {
        Tensor x = bias_.defined() ? embeddings + bias_ : embeddings;  // layer.cpp:10-12
        if (activation_ == ActivationFunction::RELU) return torch::relu(x);
        if (activation_ == ActivationFunction::SIGMOID) return torch::sigmoid(x);
        return x;
    }
    require_device(embeddings, "GeneralEncoder::forward");
    Tensor x = embeddings.stride(1) == 1 ? embeddings : embeddings.contiguous();
    Tensor out = torch::empty({x.size(0), x.size(1)}, x.options());
    torch::NoGradGuard ng;
    mcheck(marius_layer_post_hook(fp(x), x.stride(0), bias_.defined() ? fp(bias_) : nullptr, (int32_t)activation_, x.size(0), (int32_t)x.size(1), fp(out), out.stride(0),
                                  cur_stream()));
    return out;
}

Tensor GeneralEncoder::backward(Tensor grad_encoded, Tensor encoded) {
    if (!has_post_hook()) return grad_encoded;
    require_device(grad_encoded, "GeneralEncoder::backward");
    torch::NoGradGuard ng;
    Tensor g = grad_encoded.stride(1) == 1 ? grad_encoded : grad_encoded.contiguous();
    const int64_t n = g.size(0);
    const int32_t d = (int32_t)g.size(1);
    const size_t wsb = bias_.defined() ? marius_layer_post_hook_workspace_bytes(n, d) : 0;
    if (wsb && (!ws_.defined() || (size_t)ws_.numel() < wsb || ws_.device() != g.device()))
        ws_ = torch::empty({(int64_t)wsb}, torch::TensorOptions().dtype(torch::kUInt8).device(g.device()));
    Tensor gx = torch::empty({n, (int64_t)d}, g.options());
    const bool need_y = activation_ != ActivationFunction::NONE;
    mcheck(marius_layer_post_hook_backward(fp(g), g.stride(0), need_y ? fp(encoded) : nullptr, need_y ? encoded.stride(0) : 0, (int32_t)activation_, n, d, fp(gx), gx.stride(0),
                                           bias_.defined() ? fp(bias_grad_) : nullptr, wsb ? ws_.data_ptr() : nullptr, wsb, cur_stream()));
    return gx;
}

void Model::set_encoder(shared_ptr<GeneralEncoder> encoder) {
    encoder_ = encoder;
    if (!encoder_) return;
    if (named_children().contains("encoder")) replace_module("encoder", encoder_);
    else register_module("encoder", encoder_);
}

static shared_ptr<EdgeDecoder> as_edge_decoder(shared_ptr<Decoder> d) {
    auto e = std::dynamic_pointer_cast<EdgeDecoder>(d);
    if (!e) throw MariusRuntimeException("Decoder currently not supported.");  // this build: edge decoders (link prediction) only
    return e;
}
Model::Model(shared_ptr<GeneralEncoder> encoder, shared_ptr<Decoder> decoder, shared_ptr<LossFunction> loss, shared_ptr<Reporter> reporter,
             std::vector<shared_ptr<Optimizer>> optimizers)
    : Model(as_edge_decoder(decoder), loss,
            reporter ? std::dynamic_pointer_cast<LinkPredictionReporter>(reporter) : std::make_shared<LinkPredictionReporter>(),
            as_edge_decoder(decoder)->tensor_options_.device()) {
    if (!reporter_) throw MariusRuntimeException("Reporter must be specified for this learning task.");  // model.cpp:44: not a link-prediction reporter
    encoder_ = encoder;  // a pass-through, or one embedding layer with a post-hook (GeneralEncoder): its bias is a dense parameter
    if (encoder_) register_module("encoder", encoder_);
    optimizers_ = optimizers;
}
void Model::setup_optimizers(shared_ptr<ModelConfig> c) {  // model.cpp:161-250 for the decoder's parameters
    if (!c) throw MariusRuntimeException("Model::setup_optimizers: null model configuration");  // UnexpectedNullPtrException in the reference
    setup_optimizer(c->dense_optimizer, c->dense_lr, c->eps, c->beta_1, c->beta_2, c->weight_decay, c->amsgrad);
}
void Model::setup_optimizers(float dense_lr) {
    std::vector<std::pair<Tensor, Tensor>> params;
    if (encoder_ && encoder_->bias_.defined()) params.emplace_back(encoder_->bias_, encoder_->bias_grad_);  // "embedding:0_0_bias": encoder parameters first (model.cpp:175-183)
    if (decoder_->relations_.defined()) params.emplace_back(decoder_->relations_, relations_grad_);
    if (decoder_->inverse_relations_.defined()) params.emplace_back(decoder_->inverse_relations_, inverse_relations_grad_);
    optimizers_ = {std::make_shared<AdagradOptimizer>(params, dense_lr)};
}
void Model::setup_optimizer(const std::string& type, float lr, float eps, float beta_1, float beta_2, float weight_decay, bool amsgrad) {
    std::vector<std::pair<Tensor, Tensor>> params;
    if (encoder_ && encoder_->bias_.defined()) params.emplace_back(encoder_->bias_, encoder_->bias_grad_);  // "embedding:0_0_bias": encoder parameters first (model.cpp:175-183)
    if (decoder_->relations_.defined()) params.emplace_back(decoder_->relations_, relations_grad_);
    if (decoder_->inverse_relations_.defined()) params.emplace_back(decoder_->inverse_relations_, inverse_relations_grad_);
    if (type == "ADAGRAD") {
        auto o = std::make_shared<AdagradOptimizer>(params, lr, eps);
        o->weight_decay_ = weight_decay;
        optimizers_ = {o};
    } else if (type == "ADAM") {
        optimizers_ = {std::make_shared<AdamOptimizer>(params, lr, eps, beta_1, beta_2, weight_decay, amsgrad)};
    } else if (type == "SGD") {
        optimizers_ = {std::make_shared<SGDOptimizer>(params, lr)};
    } else {
        throw MariusRuntimeException("Unrecognized optimizer type: " + type);
    }
}
static std::vector<std::string> decoder_param_keys(const Model& m) {  // the dense optimizer's parameter keys: encoder layers first (model.cpp:175-183), then
    std::vector<std::string> k;                                      // named_parameters() order of the decoder (distmult.cpp:21-27)
    if (m.encoder_ && m.encoder_->bias_.defined()) k.push_back("embedding:0_0_bias");
    if (m.decoder_->relations_.defined()) k.push_back("relation_embeddings");
    if (m.decoder_->inverse_relations_.defined()) k.push_back("inverse_relation_embeddings");
    return k;
}
void Model::save(const std::string& directory) {  // model.cpp:82-106
    torch::serialize::OutputArchive model_archive, state_archive;
    // encoder_->save: GeneralEncoder registers its embedding layer as the (parameter-less) submodule "embedding:0_0" (encoder.cpp:43-45)
    torch::serialize::OutputArchive embedding_layer;
    if (encoder_ && encoder_->bias_.defined()) embedding_layer.write("bias", encoder_->bias_);  // EmbeddingLayer's registered parameter (layer.cpp:18-22)
    model_archive.write("embedding:0_0", embedding_layer);
    // decoder_->save: its parameters by registered name
    if (decoder_->relations_.defined()) model_archive.write("relation_embeddings", decoder_->relations_);
    if (decoder_->inverse_relations_.defined()) model_archive.write("inverse_relation_embeddings", decoder_->inverse_relations_);
    const auto keys = decoder_param_keys(*this);
    for (size_t i = 0; i < optimizers_.size(); ++i) {
        torch::serialize::OutputArchive optim_archive;
        optimizers_[i]->save(optim_archive, keys);
        state_archive.write(std::to_string(i), optim_archive);
    }
    model_archive.save_to(directory + "model.pt");
    state_archive.save_to(directory + "model_state.pt");
}
void Model::load(const std::string& directory, bool train) {  // model.cpp:108-134
    touch_relations();  // the relation tables are about to be overwritten: their bound is rescanned before the next training forward
    torch::serialize::InputArchive model_archive, state_archive;
    model_archive.load_from(directory + "model.pt");
    const auto keys = decoder_param_keys(*this);
    if (train) {
        state_archive.load_from(directory + "model_state.pt");
        for (size_t i = 0; i < optimizers_.size(); ++i) {
            torch::serialize::InputArchive tmp;
            state_archive.read(std::to_string(i), tmp);
            optimizers_[i]->load(tmp, keys);
        }
    }
    torch::NoGradGuard ng;
    if (encoder_ && encoder_->bias_.defined()) {
        torch::serialize::InputArchive embedding_layer;
        model_archive.read("embedding:0_0", embedding_layer);
        Tensor t;
        embedding_layer.read("bias", t);
        encoder_->bias_.copy_(t);
    }
    if (decoder_->relations_.defined()) {
        Tensor t;
        model_archive.read("relation_embeddings", t);
        decoder_->relations_.copy_(t);
    }
    if (decoder_->inverse_relations_.defined()) {
        Tensor t;
        model_archive.read("inverse_relation_embeddings", t);
        decoder_->inverse_relations_.copy_(t);
    }
}
void Model::clear_grad() {
    for (auto& o : optimizers_) o->clear_grad();
}
void Model::step() {
    torch::NoGradGuard ng;
    for (auto& o : optimizers_) o->step();
    touch_relations();  // a dense step moved the relation tables without tracking their magnitude
}

// decoder_methods.cpp:57-114 through the decoder's virtual operators, as differentiable libtorch ops: the path a user-defined comparator
// or relation operator takes (and what autograd differentiates when any plug-in is part of the model)
static std::tuple<Tensor, Tensor, Tensor, Tensor> node_corrupt_forward_generic(shared_ptr<EdgeDecoder> dec, Tensor edges, Tensor emb, Tensor dst_negs,
                                                                               Tensor src_negs, Tensor dst_filter, Tensor src_filter) {
    if (edges.dim() != 2 || (edges.size(1) != 3 && edges.size(1) != 2)) throw TensorSizeMismatchException(edges, "Edge list must be a 3 or 2 column tensor");
    const bool has_rel = edges.size(1) == 3;
    Tensor src = emb.index_select(0, edges.select(1, 0)), dst = emb.index_select(0, edges.select(1, -1));
    Tensor rels, inv_rels;
    if (has_rel) {
        Tensor rel_ids = edges.select(1, 1);
        rels = dec->relations_.index_select(0, rel_ids);
        if (dec->use_inverse_relations_ && dec->inverse_relations_.defined()) inv_rels = dec->inverse_relations_.index_select(0, rel_ids);
    }
    auto neg_rows = [&](const Tensor& ids) { return emb.index_select(0, ids.flatten()).reshape({ids.size(0), ids.size(1), emb.size(1)}); };
    auto filt = [](Tensor scores, const Tensor& f) {  // apply_score_filter, negative.cpp:306-311
        if (f.defined() && f.numel() > 0) scores = scores.index_put({f.select(1, 0), f.select(1, 1)}, torch::full({}, -1e9, scores.options()));
        return scores;
    };
    Tensor adj_src = dec->apply_relation(src, rels);
    Tensor pos = dec->compute_scores(adj_src, dst);
    Tensor neg = filt(dec->compute_scores(adj_src, neg_rows(dst_negs)), dst_filter);
    Tensor inv_pos, inv_neg;
    if (inv_rels.defined() && src_negs.defined()) {
        Tensor adj_dst = dec->apply_relation(dst, inv_rels);
        inv_pos = dec->compute_scores(adj_dst, src);
        inv_neg = filt(dec->compute_scores(adj_dst, neg_rows(src_negs)), src_filter);
    }
    if (pos.size(0) != neg.size(0)) {  // decoder_methods.cpp:103-111: pos padded to the chunked row count
        namespace F = torch::nn::functional;
        const int64_t extra = neg.size(0) - pos.size(0);
        pos = F::pad(pos, F::PadFuncOptions({0, extra}));
        if (inv_pos.defined()) inv_pos = F::pad(inv_pos, F::PadFuncOptions({0, extra}));
    }
    return std::forward_as_tuple(pos, neg, inv_pos, inv_neg);
}

bool Model::fused_ok() const {
    return !custom_forward() && decoder_ && decoder_->comparator_ && decoder_->relation_operator_ && decoder_->comparator_->kind() >= 0 &&
           decoder_->relation_operator_->kind() >= 0 && (!loss_function_ || loss_function_->kind() >= 0);
}

void Model::broadcast(std::vector<torch::Device> devices) {  // model.cpp:136-147
    if (devices.size() > 1)
        throw MariusRuntimeException(
            "Model::broadcast: this build runs one process per GPU (torch.distributed over RCCL); replicas live in the other ranks — "
            "register the process group with set_process_group() and use all_reduce()");
    devices_ = devices;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> Model::forward_lp(shared_ptr<Batch> batch, bool train) {
    (void)train;  // evaluation also calls forward_lp(batch, true) in the reference (model.cpp:337)
    const bool builtin = decoder_->comparator_->kind() >= 0 && decoder_->relation_operator_->kind() >= 0;
    // model.cpp:253: encoded_nodes = encoder_->forward(batch->node_embeddings_, ...) — a view unless the embedding layer has a post-hook
    Tensor nodes = has_post_hook() ? encoder_->forward(batch->node_embeddings_) : batch->node_embeddings_;
    if (decoder_->decoder_method_ == EdgeDecoderMethod::CORRUPT_NODE &&
        (!builtin || (torch::GradMode::is_enabled() && nodes.requires_grad())))
        return node_corrupt_forward_generic(decoder_, batch->edges_, nodes, batch->dst_neg_indices_mapping_, batch->src_neg_indices_mapping_,
                                            batch->dst_neg_filter_, batch->src_neg_filter_);
    if (decoder_->decoder_method_ == EdgeDecoderMethod::ONLY_POS) {
        auto t = only_pos_forward(decoder_, batch->edges_, nodes);
        return std::forward_as_tuple(std::get<0>(t), Tensor(), std::get<1>(t), Tensor());
    }
    if (decoder_->decoder_method_ != EdgeDecoderMethod::CORRUPT_NODE) throw MariusRuntimeException("Decoder method currently unsupported.");
    return node_corrupt_forward(decoder_, batch->edges_, nodes, batch->dst_neg_indices_mapping_, batch->src_neg_indices_mapping_, &ctx_,
                                batch->dst_neg_filter_, batch->src_neg_filter_, loss_function_ ? loss_function_->reduction_type_ : LossReduction::SUM,
                                loss_function_ ? loss_function_->kind() : MARIUS_LOSS_SOFTMAX_CE, loss_function_ ? loss_function_->margin() : 0.f);
}

std::tuple<Tensor, Tensor, Tensor, Tensor> Model::forward_lp_train(shared_ptr<Batch> batch) {
    if (decoder_->decoder_method_ != EdgeDecoderMethod::CORRUPT_NODE) return forward_lp(batch, true);
    const bool direct = batch->table_.defined();
    if (direct && has_post_hook())
        throw MariusRuntimeException("Model: an embedding layer with a bias or an activation trains through Model::train_batch (gathered rows); the table-direct fused "
                                     "step reads node rows in place and has no post-hook");
    bind_ranges(batch, direct);
    return node_corrupt_forward(decoder_, direct ? batch->global_edges_ : batch->edges_, direct ? batch->table_ : batch->node_embeddings_,
                                direct ? batch->dst_neg_indices_ : batch->dst_neg_indices_mapping_,
                                direct ? batch->src_neg_indices_ : batch->src_neg_indices_mapping_, &ctx_,
                                batch->dst_neg_filter_, batch->src_neg_filter_, loss_function_ ? loss_function_->reduction_type_ : LossReduction::SUM,
                                loss_function_ ? loss_function_->kind() : MARIUS_LOSS_SOFTMAX_CE, loss_function_ ? loss_function_->margin() : 0.f,
                                MARIUS_LP_TRAIN_ONLY);
}

static void ensure(Tensor& t, int64_t bytes, torch::Device dev) {
    if (!t.defined() || t.numel() < bytes || t.device() != dev) t = torch::empty({bytes}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
}

// forward + loss + hand-derived backward; leaves per-occurrence node gradients in ctx_ (gocc) and dense relation gradients in *_grad_
static void model_backward(Model& m, shared_ptr<Batch> batch) {
    auto st = cur_stream();
    LpContext& c = m.ctx_;
    mcheck(marius_lp_loss(&c.desc, &c.layout, c.workspace.data_ptr(), st));
    mcheck(marius_lp_backward(&c.desc, &c.layout, c.workspace.data_ptr(), st));
    m.loss_ = c.view(c.layout.loss, {4});
    (void)st;
}

// relation ids of the batch grouped by value: taken from the loader when it prepared them ahead, sorted here otherwise
struct RelMap {
    const int64_t* uniq;
    const int64_t* inverse;
    const int32_t* perm;
    const int32_t* seg;
    const void* plan = nullptr;  // marius_segment_plan of this map, when the loader prepared one
};
static RelMap relation_map(Model& m, shared_ptr<Batch> batch) {
    LpContext& c = m.ctx_;
    const int64_t B = c.desc.B;
    if (batch->rel_perm_.defined() && batch->rel_perm_.size(0) == B)
        return {ip(batch->rel_uniq_), ip(batch->rel_inverse_), batch->rel_perm_.data_ptr<int32_t>(), batch->rel_seg_.data_ptr<int32_t>(),
                batch->rel_plan_.defined() ? batch->rel_plan_.data_ptr() : nullptr};
    auto dev = c.workspace.device();
    if (!m.rel_ids_.defined() || m.rel_ids_.size(0) != B) {
        m.rel_ids_ = torch::empty({B}, i64(dev));
        m.rel_uniq_ = torch::empty({B}, i64(dev));
        m.rel_inverse_ = torch::empty({B}, i64(dev));
        m.rel_perm_ = torch::empty({B}, i32(dev));
        m.rel_seg_ = torch::empty({B + 1}, i32(dev));
        m.rel_count_ = torch::zeros({1}, i64(dev));
        {   // zero-initialised: the sort's control block (include/marius_hip.h)
            const int64_t wsb = (int64_t)marius_sort_unique_workspace_bytes(B);
            if (!m.rel_ws_.defined() || m.rel_ws_.numel() < wsb || m.rel_ws_.device() != dev) m.rel_ws_ = torch::zeros({wsb}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
        }
    }
    m.rel_ids_.copy_(batch->edges_.select(1, 1));
    mcheck(marius_sort_unique(ip(m.rel_ids_), B, key_bits_for(c.desc.R), ip(m.rel_uniq_), ip(m.rel_inverse_), m.rel_perm_.data_ptr<int32_t>(),
                              m.rel_seg_.data_ptr<int32_t>(), ip(m.rel_count_), m.rel_ws_.data_ptr(), (size_t)m.rel_ws_.numel(), cur_stream()));
    return {ip(m.rel_uniq_), ip(m.rel_inverse_), m.rel_perm_.data_ptr<int32_t>(), m.rel_seg_.data_ptr<int32_t>()};
}

// relations_.grad / inverse_relations_.grad [R, d] (dense, as autograd leaves them): index_select backward without atomics
static void relation_grads_dense(Model& m, shared_ptr<Batch> batch) {
    LpContext& c = m.ctx_;
    if (c.desc.edge_cols != 3) return;
    const int64_t B = c.desc.B;
    RelMap rm = relation_map(m, batch);
    ensure(m.rel_carry_, (int64_t)marius_segment_carry_bytes(B, c.desc.d), c.workspace.device());
    Tensor* grads[2] = {&m.relations_grad_, &m.inverse_relations_grad_};
    for (int dir = 0; dir < (c.desc.use_inverse ? 2 : 1); ++dir) {
        grads[dir]->zero_();
        const float* rows = (const float*)((const char*)c.workspace.data_ptr() + c.layout.grel[dir]);
        mcheck(marius_segment_sum_rows(rows, c.layout.d_ld, rm.perm, rm.inverse, rm.seg, B, c.desc.d, rm.uniq, fp(*grads[dir]), grads[dir]->stride(0),
                                       m.rel_carry_.data_ptr(), cur_stream()));
    }
}

// the sparse relation step below as jobs of marius_segment_adagrad_scatter_group (one per direction); false: not expressible (see there)
static bool relation_step_jobs(Model& m, shared_ptr<Batch> batch, marius_segment_update* jobs, int& njobs) {
    LpContext& c = m.ctx_;
    if (c.desc.edge_cols != 3) return true;
    if (m.optimizers_.size() != 1) return false;
    auto* opt = dynamic_cast<AdagradOptimizer*>(m.optimizers_[0].get());
    if (!opt || opt->weight_decay_ != 0.f) return false;
    const int ndir = c.desc.use_inverse ? 2 : 1;
    if ((int)opt->params_.size() != ndir) return false;
    if (!(batch->rel_perm_.defined() && batch->rel_perm_.size(0) == c.desc.B && batch->rel_plan_.defined())) return false;  // the loader's map + plan only
    const int64_t B = c.desc.B;
    RelMap rm = relation_map(m, batch);
    const int64_t one = (int64_t)marius_segment_carry_bytes(B, c.desc.d);
    ensure(m.rel_carry_, ndir * one, c.workspace.device());  // the directions run side by side: a carry each
    for (int dir = 0; dir < ndir; ++dir) {
        Tensor& w = opt->params_[dir].first;
        marius_segment_update& u = jobs[njobs++];
        u.rows = (const float*)((const char*)c.workspace.data_ptr() + c.layout.grel[dir]);
        u.rows_ld = c.layout.d_ld;
        u.perm = rm.perm;
        u.inverse = rm.inverse;
        u.seg_offsets = rm.seg;
        u.n = B;
        u.d = c.desc.d;
        u.uniq_ids = rm.uniq;
        u.table = fp(w);
        u.state = fp(opt->state_[dir]);
        u.table_ld = w.stride(0);
        u.lr = opt->learning_rate_;
        u.eps = opt->eps_;
        u.carry = (char*)m.rel_carry_.data_ptr() + dir * one;
        u.plan = rm.plan;
        u.absmax = m.relation_bound();
    }
    return true;
}

// Dense Adagrad step on the relation tables restricted to the rows the batch touched.  A row with zero gradient is a fixed point of
// the dense rule (sum += 0; w -= lr*0/(sqrt(sum)+eps)), so this equals AdagradOptimizer::step() on the dense gradient bit for bit
// while skipping the [R, d] zero-fill, the dense scatter target and the full-table pass.  Returns false if the optimizer is not
// plain Adagrad (then the caller takes the dense route).
static bool relation_step_sparse(Model& m, shared_ptr<Batch> batch) {
    LpContext& c = m.ctx_;
    if (c.desc.edge_cols != 3) return true;
    if (m.optimizers_.size() != 1) return false;
    auto* opt = dynamic_cast<AdagradOptimizer*>(m.optimizers_[0].get());
    if (!opt || opt->weight_decay_ != 0.f) return false;
    const int ndir = c.desc.use_inverse ? 2 : 1;
    if ((int)opt->params_.size() != ndir) return false;
    const int64_t B = c.desc.B;
    RelMap rm = relation_map(m, batch);
    ensure(m.rel_carry_, (int64_t)marius_segment_carry_bytes(B, c.desc.d), c.workspace.device());
    for (int dir = 0; dir < ndir; ++dir) {
        Tensor& w = opt->params_[dir].first;
        const float* rows = (const float*)((const char*)c.workspace.data_ptr() + c.layout.grel[dir]);
        if (m.rel_ranges_valid_)
            mcheck(marius_segment_adagrad_scatter_tracked(rows, c.layout.d_ld, rm.perm, rm.inverse, rm.seg, B, c.desc.d, rm.uniq, fp(w), fp(opt->state_[dir]),
                                                          w.stride(0), opt->learning_rate_, opt->eps_, m.rel_carry_.data_ptr(), rm.plan, m.relation_bound(),
                                                          cur_stream()));
        else if (rm.plan)
            mcheck(marius_segment_adagrad_scatter_planned(rows, c.layout.d_ld, rm.perm, rm.inverse, rm.seg, B, c.desc.d, rm.uniq, fp(w), fp(opt->state_[dir]),
                                                          w.stride(0), opt->learning_rate_, opt->eps_, m.rel_carry_.data_ptr(), rm.plan, cur_stream()));
        else
            mcheck(marius_segment_adagrad_scatter(rows, c.layout.d_ld, rm.perm, rm.inverse, rm.seg, B, c.desc.d, rm.uniq, fp(w), fp(opt->state_[dir]), w.stride(0),
                                                  opt->learning_rate_, opt->eps_, m.rel_carry_.data_ptr(), cur_stream()));
    }
    return true;
}

// model.cpp:290-333 as written there: forward through the (possibly overridden) forward_lp and loss, loss.backward(), dense step,
// sparse Adagrad rule.  Taken when a user plug-in is part of the model; the built-in configuration never comes here.
void Model::train_batch_generic(shared_ptr<Batch> batch, bool call_step) {
    if (call_step) clear_grad();
    Tensor emb_plain = batch->node_embeddings_, rel_plain = decoder_->relations_, inv_plain = decoder_->inverse_relations_;
    Tensor emb = emb_plain.detach().requires_grad_(true);
    Tensor rel = rel_plain.defined() ? rel_plain.detach().requires_grad_(true) : Tensor();
    Tensor inv = inv_plain.defined() ? inv_plain.detach().requires_grad_(true) : Tensor();
    Tensor bias_plain = encoder_ ? encoder_->bias_ : Tensor(), bias = bias_plain.defined() ? bias_plain.detach().requires_grad_(true) : Tensor();
    batch->node_embeddings_ = emb;
    decoder_->relations_ = rel;
    decoder_->inverse_relations_ = inv;
    if (bias.defined()) encoder_->bias_ = bias;
    Tensor loss;
    try {
        auto t = forward_lp(batch, true);
        if (!loss_function_) throw MariusRuntimeException("Model::train_batch: no loss function");
        loss = (*loss_function_)(std::get<0>(t), std::get<1>(t), true);
        Tensor rhs = loss, lhs;
        if (std::get<3>(t).defined()) {
            lhs = (*loss_function_)(std::get<2>(t), std::get<3>(t), true);
            loss = lhs + rhs;  // model.cpp:309-312
        }
        if (!loss.requires_grad()) throw MariusRuntimeException("Model::train_batch: the loss does not depend on the parameters (forward_lp must return differentiable scores)");
        loss.backward();
        loss_ = torch::stack({loss.detach(), rhs.detach(), lhs.defined() ? lhs.detach() : torch::zeros_like(rhs.detach()), torch::zeros_like(rhs.detach())});
    } catch (...) {
        batch->node_embeddings_ = emb_plain;
        decoder_->relations_ = rel_plain;
        decoder_->inverse_relations_ = inv_plain;
        if (bias.defined()) encoder_->bias_ = bias_plain;
        throw;
    }
    batch->node_embeddings_ = emb_plain;
    decoder_->relations_ = rel_plain;
    decoder_->inverse_relations_ = inv_plain;
    if (bias.defined()) {
        encoder_->bias_ = bias_plain;
        encoder_->bias_grad_.copy_(bias.grad().defined() ? bias.grad() : torch::zeros_like(bias_plain));
    }
    if (rel.defined()) relations_grad_.copy_(rel.grad().defined() ? rel.grad() : torch::zeros_like(rel_plain));
    if (inv.defined() && inverse_relations_grad_.defined()) inverse_relations_grad_.copy_(inv.grad().defined() ? inv.grad() : torch::zeros_like(inv_plain));
    batch->node_embeddings_grad_ = emb.grad().defined() ? emb.grad() : torch::zeros_like(emb_plain);
    publish_grads();
    if (call_step) step();
    if (batch->node_embeddings_.defined()) batch->accumulateGradients(sparse_lr_);
}

// model.cpp:324 leaves relations_.grad() / inverse_relations_.grad() for whoever steps the parameters (the model's optimizers, or a user's
// optimizer over named_parameters() after train_batch(batch, false)): the hand-derived backward writes relations_grad_ /
// inverse_relations_grad_, which are made the parameters' .grad() here (aliases: clear_grad() zeroes both views of the same memory)
void Model::publish_grads() {
    if (encoder_ && encoder_->bias_.defined()) encoder_->bias_.mutable_grad() = encoder_->bias_grad_;
    if (decoder_->relations_.defined() && relations_grad_.defined()) decoder_->relations_.mutable_grad() = relations_grad_;
    if (decoder_->inverse_relations_.defined() && inverse_relations_grad_.defined()) decoder_->inverse_relations_.mutable_grad() = inverse_relations_grad_;
}

void Model::all_reduce() {  // model.cpp:149-159: sum of the dense gradients over the replicas
    if (process_group_.empty()) return;  // a single process holds the only replica
    auto pg = c10d::resolve_process_group(process_group_);
    std::vector<Tensor> grads;
    if (relations_grad_.defined()) grads.push_back(relations_grad_);
    if (inverse_relations_grad_.defined()) grads.push_back(inverse_relations_grad_);
    for (auto& g : grads) {
        std::vector<Tensor> v{g};
        pg->allreduce(v)->wait();
    }
}

shared_ptr<Model> initModelFromConfig(shared_ptr<ModelConfig> c, std::vector<torch::Device> devices, int num_relations, bool train) {
    if (!c) throw MariusRuntimeException("initModelFromConfig: null model configuration");
    return initModelFromConfig(*c, std::move(devices), num_relations, train);
}
shared_ptr<Model> initModelFromConfig(const ModelConfig& c, std::vector<torch::Device> devices, int num_relations, bool train) {  // model.cpp:361-440
    if (devices.empty()) throw MariusRuntimeException("initModelFromConfig: no device");
    auto dev = devices[0];
    auto opts = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
    DecoderType dt;
    if (c.decoder == "DISTMULT") dt = DecoderType::DISTMULT;
    else if (c.decoder == "COMPLEX") dt = DecoderType::COMPLEX;
    else if (c.decoder == "TRANSE") dt = DecoderType::TRANSE;
    else throw MariusRuntimeException("Decoder currently not supported.");
    EdgeDecoderMethod m = c.decoder_method == "ONLY_POS" ? EdgeDecoderMethod::ONLY_POS : EdgeDecoderMethod::CORRUPT_NODE;
    auto decoder = get_edge_decoder(dt, m, num_relations, c.embedding_dim, opts, c.inverse_edges);
    auto loss = getLossFunction(c.loss, c.loss_reduction == "MEAN" ? LossReduction::MEAN : LossReduction::SUM, c.margin);
    auto model = std::make_shared<Model>(decoder, loss, std::make_shared<LinkPredictionReporter>(), dev);
    model->sparse_lr_ = c.sparse_lr;
    {   // the embedding layer's post-hook (LayerConfig bias / bias_init / activation; default: none — a pass-through encoder)
        ActivationFunction act = ActivationFunction::NONE;
        if (c.encoder_activation == "RELU") act = ActivationFunction::RELU;
        else if (c.encoder_activation == "SIGMOID") act = ActivationFunction::SIGMOID;
        else if (c.encoder_activation != "NONE") throw MariusRuntimeException("Unsupported activation function");  // activation.cpp:19
        if (c.encoder_bias || act != ActivationFunction::NONE)
            model->set_encoder(std::make_shared<GeneralEncoder>(c.embedding_dim, c.encoder_bias, act, dev, c.encoder_bias_init));
    }
    if (train) model->setup_optimizer(c.dense_optimizer, c.dense_lr, c.eps, c.beta_1, c.beta_2, c.weight_decay, c.amsgrad);
    model->broadcast({dev});
    return model;
}

void Model::train_batch(shared_ptr<Batch> batch, bool call_step) {
    if (!fused_ok()) return train_batch_generic(batch, call_step);
    if (call_step) clear_grad();
    // Layer::post_hook of the embedding layer (layer.cpp:9-16): the decoder scores act(rows + bias); the rows themselves stay in the batch for
    // the sparse update (Batch::accumulateGradients works on the raw embeddings, batch.cpp:62-79)
    Tensor raw_rows;
    if (has_post_hook()) {
        raw_rows = batch->node_embeddings_;
        batch->node_embeddings_ = encoder_->forward(raw_rows);
    }
    struct Restore {  // exception-safe: the batch gets its raw rows back whatever happens below
        shared_ptr<Batch>& b;
        Tensor& raw;
        ~Restore() { if (raw.defined()) b->node_embeddings_ = raw; }
    } restore{batch, raw_rows};
    forward_lp_train(batch);
    model_backward(*this, batch);
    relation_grads_dense(*this, batch);
    // node_embeddings_.grad [U, d]: sum of the occurrence gradients per unique row (autograd's index_add), atomic-free
    const int64_t L = batch->occ_perm_.size(0);
    const int64_t U = batch->node_embeddings_.size(0);
    ensure(carry_, (int64_t)marius_segment_carry_bytes(L, ctx_.desc.d), device_);
    batch->node_embeddings_grad_ = torch::zeros({U, (int64_t)ctx_.desc.d}, batch->node_embeddings_.options());
    const float* gocc = (const float*)((const char*)ctx_.workspace.data_ptr() + ctx_.layout.gocc);
    mcheck(marius_segment_sum_rows(gocc, ctx_.layout.d_ld, batch->occ_perm_.data_ptr<int32_t>(), ip(batch->occ_inverse_),
                                   batch->occ_seg_offsets_.data_ptr<int32_t>(), L, ctx_.desc.d, nullptr, fp(batch->node_embeddings_grad_),
                                   batch->node_embeddings_grad_.stride(0), carry_.data_ptr(), cur_stream()));
    if (raw_rows.defined()) {  // through the post-hook: d/d(rows) = d/d(encoded) * act'(.), bias.grad = its column sums
        batch->node_embeddings_grad_ = encoder_->backward(batch->node_embeddings_grad_, batch->node_embeddings_);
        batch->node_embeddings_ = raw_rows;
    }
    publish_grads();
    if (call_step) step();
    if (batch->node_embeddings_.defined()) batch->accumulateGradients(sparse_lr_);
}

void Model::backward_into_tables(shared_ptr<Batch> batch, Tensor table, Tensor state, bool table_direct) {
    // one row pitch for both (the update kernels take a single ld): contiguous tensors, or the two halves of an interleaved [row | state] buffer
    if (table.stride(0) != state.stride(0) || table.stride(1) != 1 || state.stride(1) != 1)
        throw MariusRuntimeException("backward_into_tables: the table and its optimizer state must share one row pitch and have unit column stride");
    if (table_direct) {
        if (!batch->global_edges_.defined()) throw MariusRuntimeException("backward_into_tables: table-direct step without the batch's global-id edges");
        batch->table_ = table;
    }
    forward_lp_train(batch);
    batch->table_ = Tensor();
    const int64_t L = batch->occ_perm_.size(0);
    // Endpoint occurrences whose node occurs once in the batch (about half of them at the bench shape) take their Adagrad step inside the edge
    // backward, which holds the row and its whole gradient in registers: no gocc store, no re-read next to the same table row in the update below
    // (marius_lp_desc.upd_*; needs the table itself behind `emb` and the loader's plan for the per-occurrence flags).
    int64_t fused_below = 0;
    static const bool fuse_env = [] { const char* e = getenv("MARIUS_FUSE_ENDPOINT_UPDATE"); return !(e && e[0] == '0'); }();
    if (table_direct && fuse_env && fuse_endpoint_update_ && batch->occ_plan_.defined()) {
        marius_lp_desc& d = ctx_.desc;  // (rebuilt from zero by every forward: nothing to undo)
        d.upd_occ_single = marius_segment_plan_occ_single(batch->occ_plan_.data_ptr(), L);
        d.upd_state = fp(state);
        d.upd_absmax = node_track_;
        d.upd_lr = sparse_lr_;
        d.upd_eps = 1e-10f;
        if (marius_lp_fuses_endpoint_update(&d)) fused_below = 2 * d.B;
        else d.upd_occ_single = nullptr, d.upd_state = nullptr, d.upd_absmax = nullptr;
    }
    last_fused_below_ = fused_below;
    model_backward(*this, batch);
    if (ev_grads_) HIPCHECK(hipEventRecord((hipEvent_t)ev_grads_, c10::hip::getCurrentHIPStream(device_.index()).stream()));
    const auto dev_index = device_.index();
    ensure(carry_, (int64_t)marius_segment_carry_bytes(L, ctx_.desc.d), device_);
    const float* gocc = (const float*)((const char*)ctx_.workspace.data_ptr() + ctx_.layout.gocc);
    // The node-table update and the relation-table updates are independent.  With the loader's plans at hand all of them are ONE pair of
    // launches (marius_segment_adagrad_scatter_group); the side stream the relation step used to run on, its fork / join events and four
    // launches are gone from the step's tail.
    marius_segment_update node_job = {};
    if (batch->occ_plan_.defined()) {
        // relation jobs first: a hub relation's segment spans hundreds of chunks and is finished by ONE wave, so those workgroups should start
        // at the head of the launch, with the node table's many short ones filling in behind them
        marius_segment_update jobs[3] = {};
        int njobs = 0;
        const bool rel_ok = relation_step_jobs(*this, batch, jobs, njobs);
        marius_segment_update& u = jobs[njobs++];
        u.rows = gocc;
        u.rows_ld = ctx_.layout.d_ld;
        u.perm = batch->occ_perm_.data_ptr<int32_t>();
        u.inverse = ip(batch->occ_inverse_);
        u.seg_offsets = batch->occ_seg_offsets_.data_ptr<int32_t>();
        u.n = L;
        u.d = ctx_.desc.d;
        u.uniq_ids = ip(batch->unique_node_indices_);
        u.table = fp(table);
        u.state = fp(state);
        u.table_ld = table.stride(0);
        u.lr = sparse_lr_;
        u.eps = 1e-10f;
        u.carry = carry_.data_ptr();
        u.plan = batch->occ_plan_.data_ptr();
        u.absmax = node_track_;
        u.fused_below = fused_below;
        if (rel_ok) {
            mcheck(marius_segment_adagrad_scatter_group(jobs, njobs, cur_stream()));
            return;
        }
        node_job = u;
    }
    // otherwise: the relation step (6 small, latency-bound launches) on a side stream underneath the node-table update, joined before
    // returning (the next forward reads the relation tables)
    c10::hip::HIPStream main = c10::hip::getCurrentHIPStream(dev_index);
    if (!side_stream_) {
        side_stream_ = new c10::hip::HIPStream(c10::hip::getStreamFromExternal((hipStream_t)aux_stream(dev_index, AUX_RELATIONS), dev_index));
        hipEvent_t e0, e1;
        HIPCHECK(hipEventCreateWithFlags(&e0, order_event_flags()));
        HIPCHECK(hipEventCreateWithFlags(&e1, order_event_flags()));
        ev_fork_ = e0;
        ev_join_ = e1;
    }
    c10::hip::HIPStream side = *(c10::hip::HIPStream*)side_stream_;
    HIPCHECK(hipEventRecord((hipEvent_t)ev_fork_, main.stream()));
    HIPCHECK(hipStreamWaitEvent(side.stream(), (hipEvent_t)ev_fork_, 0));
    bool sparse_ok;
    {
        StreamScope scope(side);
        sparse_ok = relation_step_sparse(*this, batch);
    }
    HIPCHECK(hipEventRecord((hipEvent_t)ev_join_, side.stream()));
    // whatever happens below, the caller's stream must not run ahead of the relation step on the side stream (a throw from the node
    // update used to leave ev_join_ unwaited, with the tables half-stepped AND the next forward racing the side stream)
    struct Join {
        hipStream_t main;
        hipEvent_t ev;
        ~Join() { (void)hipStreamWaitEvent(main, ev, 0); }
    } join{main.stream(), (hipEvent_t)ev_join_};
    if (!sparse_ok) {
        clear_grad();
        relation_grads_dense(*this, batch);
        step();  // (Adam, SGD, weight decay: the relation bound is rescanned before the next forward — touch_relations())
    }
    if (fused_below > 0)  // (the job form carries fused_below; one job)
        mcheck(marius_segment_adagrad_scatter_group(&node_job, 1, cur_stream()));
    else if (node_track_)
        mcheck(marius_segment_adagrad_scatter_tracked(gocc, ctx_.layout.d_ld, batch->occ_perm_.data_ptr<int32_t>(), ip(batch->occ_inverse_),
                                                      batch->occ_seg_offsets_.data_ptr<int32_t>(), L, ctx_.desc.d, ip(batch->unique_node_indices_), fp(table),
                                                      fp(state), table.stride(0), sparse_lr_, 1e-10f, carry_.data_ptr(),
                                                      batch->occ_plan_.defined() ? batch->occ_plan_.data_ptr() : nullptr, node_track_, cur_stream()));
    else if (batch->occ_plan_.defined())
        mcheck(marius_segment_adagrad_scatter_planned(gocc, ctx_.layout.d_ld, batch->occ_perm_.data_ptr<int32_t>(), ip(batch->occ_inverse_),
                                                      batch->occ_seg_offsets_.data_ptr<int32_t>(), L, ctx_.desc.d, ip(batch->unique_node_indices_), fp(table),
                                                      fp(state), table.stride(0), sparse_lr_, 1e-10f, carry_.data_ptr(), batch->occ_plan_.data_ptr(), cur_stream()));
    else
        mcheck(marius_segment_adagrad_scatter(gocc, ctx_.layout.d_ld, batch->occ_perm_.data_ptr<int32_t>(), ip(batch->occ_inverse_),
                                              batch->occ_seg_offsets_.data_ptr<int32_t>(), L, ctx_.desc.d, ip(batch->unique_node_indices_), fp(table),
                                              fp(state), table.stride(0), sparse_lr_, 1e-10f, carry_.data_ptr(), cur_stream()));
}

void Model::backward_to_unique_grads(shared_ptr<Batch> batch, Tensor grad_out, bool local_relation_step, Tensor out_rows) {
    if (has_post_hook()) throw MariusRuntimeException("Model: the sharded trainer scores the exchanged rows as they are; an embedding layer with a bias or an activation is not supported there");
    forward_lp_train(batch);  // (sharded table: the rows came from other ranks' shards; the bound is the one of the gathered copy itself, Batch::row_bound_)
    model_backward(*this, batch);
    const int64_t L = batch->occ_perm_.size(0);
    ensure(carry_, (int64_t)marius_segment_carry_bytes(L, ctx_.desc.d), device_);
    const float* gocc = (const float*)((const char*)ctx_.workspace.data_ptr() + ctx_.layout.gocc);
    const int64_t* orows = out_rows.defined() ? ip(out_rows) : nullptr;  // where unique row u's gradient goes (fixed-capacity exchange: marius_a2a_rows_post's place)
    bool done = false;
    if (local_relation_step) {
        // The step's tail as ONE launch pair when the loader planned the maps: both relation tables' touched-rows Adagrad steps and the reduction of the
        // node gradients are jobs of marius_segment_adagrad_scatter_group (the last one reduce-only: marius_segment_update.sum_out) — five launches
        // otherwise (two pairs for the relation tables of round 4's four, reduce + fix-up for the nodes).  Same results bit for bit.
        marius_segment_update jobs[3] = {};
        int njobs = 0;
        static const bool group_env = [] { const char* e = getenv("MARIUS_REL_GROUP"); return !(e && e[0] == '0'); }();
        if (group_env && relation_step_jobs(*this, batch, jobs, njobs)) {
            bool nodes_in = false;
            if (batch->occ_plan_.defined()) {
                marius_segment_update& u = jobs[njobs++];
                u.rows = gocc;
                u.rows_ld = ctx_.layout.d_ld;
                u.perm = batch->occ_perm_.data_ptr<int32_t>();
                u.inverse = ip(batch->occ_inverse_);
                u.seg_offsets = batch->occ_seg_offsets_.data_ptr<int32_t>();
                u.n = L;
                u.d = ctx_.desc.d;
                u.carry = carry_.data_ptr();
                u.plan = batch->occ_plan_.data_ptr();
                u.sum_out = fp(grad_out);
                u.sum_out_ld = grad_out.stride(0);
                u.sum_out_rows = orows;
                nodes_in = true;
            }
            if (njobs > 0) mcheck(marius_segment_adagrad_scatter_group(jobs, njobs, cur_stream()));
            if (nodes_in) return;
            done = true;
        } else {
            done = relation_step_sparse(*this, batch);
        }
    }
    if (!done) {
        relation_grads_dense(*this, batch);
        if (local_relation_step) step();  // optimizer without a touched-rows form: dense step on this replica
    }
    if (batch->occ_plan_.defined())
        mcheck(marius_segment_sum_rows_planned(gocc, ctx_.layout.d_ld, batch->occ_perm_.data_ptr<int32_t>(), ip(batch->occ_inverse_),
                                               batch->occ_seg_offsets_.data_ptr<int32_t>(), L, ctx_.desc.d, orows, fp(grad_out), grad_out.stride(0),
                                               carry_.data_ptr(), batch->occ_plan_.data_ptr(), cur_stream()));
    else
        mcheck(marius_segment_sum_rows(gocc, ctx_.layout.d_ld, batch->occ_perm_.data_ptr<int32_t>(), ip(batch->occ_inverse_),
                                       batch->occ_seg_offsets_.data_ptr<int32_t>(), L, ctx_.desc.d, orows, fp(grad_out), grad_out.stride(0),
                                       carry_.data_ptr(), cur_stream()));
}

static bool flash_f16_env() {
    static const bool on = [] { const char* e = getenv("MARIUS_FLASH_F16"); return !(e && e[0] == '0'); }();
    return on;
}
bool Model::flash_f16_enabled() { return flash_f16_env(); }

void Model::ensure_relation_ranges() {
    Tensor* rels[2] = {&decoder_->relations_, &decoder_->inverse_relations_};
    bool same = rel_ranges_valid_;
    for (int i = 0; i < 2 && same; ++i) same = rels[i]->defined() ? tracked_rel_[i].is(*rels[i]) : tracked_rel_[i].ptr == nullptr;
    if (same) return;
    if (!range_state_.defined()) range_state_ = torch::zeros({2}, f32(device_));
    else range_state_.narrow(0, 1, 1).zero_();
    for (int i = 0; i < 2; ++i) {
        tracked_rel_[i] = Scanned{};
        if (!rels[i]->defined()) continue;
        require_device(*rels[i], "Model::ensure_relation_ranges");
        mcheck(marius_table_absmax(fp(*rels[i]), rels[i]->size(0), rels[i]->stride(0), (int32_t)rels[i]->size(1), range_state_.data_ptr<float>() + 1, cur_stream()));
        tracked_rel_[i].set(*rels[i]);
    }
    rel_ranges_valid_ = true;
}

void Model::track_ranges(Tensor table) {
    if (!range_state_.defined()) range_state_ = torch::zeros({2}, f32(device_));
    else range_state_.narrow(0, 0, 1).zero_();
    mcheck(marius_table_absmax(fp(table), table.size(0), table.stride(0), (int32_t)table.size(1), range_state_.data_ptr<float>(), cur_stream()));
    tracked_table_.set(table);
    ranges_valid_ = true;
    ensure_relation_ranges();
}
void Model::drop_ranges() {
    ranges_valid_ = false;
    tracked_table_ = Scanned{};
    ctx_.absmax = ctx_.absmax_rel = Tensor();
    node_track_ = nullptr;
}

void Model::bind_ranges(shared_ptr<Batch> batch, bool direct) {
    ctx_.absmax = ctx_.absmax_rel = Tensor();
    node_track_ = nullptr;
    if (!flash_f16_env()) return;
    if (direct) {
        if (external_node_bound_.defined()) {  // partition-buffer slab: the buffer's running bound
            ensure_relation_ranges();
            ctx_.absmax = external_node_bound_;
            ctx_.absmax_rel = range_state_.narrow(0, 1, 1);
            node_track_ = external_node_bound_.data_ptr<float>();
        } else if (tracks(batch->table_)) {
            ensure_relation_ranges();
            ctx_.absmax = range_state_;
            node_track_ = range_state_.data_ptr<float>();
        }  // else nobody scanned this table: bf16 records
        return;
    }
    // a gathered copy: bound the rows the decoder is about to read — by the gatherer's scan, or here
    const Tensor& emb = batch->node_embeddings_;
    if (!emb.defined() || !emb.is_cuda() || emb.dim() != 2 || emb.scalar_type() != torch::kFloat32 || emb.requires_grad()) return;
    Tensor nb = batch->row_bound_;
    if (!nb.defined()) {
        if (!row_bound_.defined()) row_bound_ = torch::zeros({1}, f32(device_));
        if (batch->num_unique_dev_.defined() && batch->num_unique_dev_.is_cuda())  // capacity-sized copies carry their row count on the device
            mcheck(marius_table_absmax_counted(fp(emb), emb.size(0), ip(batch->num_unique_dev_), emb.stride(0), (int32_t)emb.size(1), row_bound_.data_ptr<float>(), cur_stream()));
        else
            mcheck(marius_table_absmax(fp(emb), emb.size(0), emb.stride(0), (int32_t)emb.size(1), row_bound_.data_ptr<float>(), cur_stream()));
        nb = row_bound_;
    }
    ensure_relation_ranges();
    ctx_.absmax = nb;
    ctx_.absmax_rel = range_state_.narrow(0, 1, 1);
    if (tracks(batch->table_to_update_)) node_track_ = range_state_.data_ptr<float>();  // gathered A/B form of the fused step on a tracked table
}

std::vector<Tensor> Model::dense_state() {
    std::vector<Tensor> out;
    for (auto& o : optimizers_) {
        for (auto& p : o->params_) out.push_back(p.first);
        for (auto& sl : o->state_slots())
            for (auto& t : *sl.second) out.push_back(t);
    }
    return out;
}

Model::~Model() {
    if (ev_fork_) (void)hipEventDestroy((hipEvent_t)ev_fork_);
    if (ev_join_) (void)hipEventDestroy((hipEvent_t)ev_join_);
    delete (c10::hip::HIPStream*)side_stream_;
}

void Model::evaluate_batch(shared_ptr<Batch> batch) {
    auto t = forward_lp(batch, true);
    if (std::get<1>(t).defined()) reporter_->addResult(std::get<0>(t), std::get<1>(t));
    if (std::get<3>(t).defined()) reporter_->addResult(std::get<2>(t), std::get<3>(t));
}

// ------------------------------------------------------------------------------------------------ dataloader
DataLoader::DataLoader(shared_ptr<InMemory> edges, shared_ptr<Storage> node_embeddings, shared_ptr<Storage> node_embeddings_state,
                       shared_ptr<CorruptNodeNegativeSampler> negative_sampler, shared_ptr<MariusGenerator> generator, int64_t batch_size, bool train)
    : edges_(edges), node_embeddings_(node_embeddings), node_embeddings_state_(node_embeddings_state), negative_sampler_(negative_sampler),
      generator_(generator), batch_size_(batch_size), train_(train) {
    graph_ = std::make_shared<MariusGraph>();
    pb_embeddings_ = std::dynamic_pointer_cast<PartitionBufferStorage>(node_embeddings_);
    pb_state_ = std::dynamic_pointer_cast<PartitionBufferStorage>(node_embeddings_state_);
    if (pb_embeddings_ && node_embeddings_state_ && !pb_state_)
        throw MariusRuntimeException("DataLoader: embeddings and optimizer state must use the same storage backend");
    // graph_storage.h:255-268: the id range negatives are drawn from is what is in memory
    graph_->num_nodes_in_memory_ = pb_embeddings_ ? pb_embeddings_->getNumInMemory() : node_embeddings_->dim0_size_;
    num_edges_ = edges_->dim0_size_;
    key_bits_ = key_bits_for(graph_->num_nodes_in_memory_);
    if (negative_sampler_) negative_sampler_->generator_ = generator_;
}

void DataLoader::setEdgeBucketSizes(std::vector<int64_t> sizes) {
    buckets_validated_ = false;
    edge_bucket_starts_.assign(sizes.size() + 1, 0);
    for (size_t i = 0; i < sizes.size(); ++i) edge_bucket_starts_[i + 1] = edge_bucket_starts_[i] + sizes[i];
    if (edge_bucket_starts_.back() != edges_->dim0_size_) throw MariusRuntimeException("edge bucket sizes do not add up to the number of edges");
}

void DataLoader::loadStorage() {
    if (!partitioned()) return;
    auto o = pb_embeddings_->options_;
    if (edge_bucket_starts_.empty() && !edges_->edge_bucket_sizes_.empty()) setEdgeBucketSizes(edges_->edge_bucket_sizes_);
    if ((int64_t)edge_bucket_starts_.size() != (int64_t)o->num_partitions * o->num_partitions + 1)
        throw MariusRuntimeException("DataLoader: partitioned training needs the edge bucket sizes (num_partitions^2 entries)");
    std::tie(buffer_states_, edge_buckets_per_buffer_) = getEdgeBucketOrdering(o->edge_bucket_ordering, o->num_partitions, o->buffer_capacity,
                                                                                o->fine_to_coarse_ratio, o->num_cache_partitions,
                                                                                o->randomly_assign_edge_buckets, generator_);
    pb_embeddings_->setBufferOrdering(buffer_states_);
    pb_embeddings_->load();
    if (pb_state_) {
        pb_state_->setBufferOrdering(buffer_states_);
        pb_state_->load();
    }
    buffer_cursor_ = 0;
}

struct DataLoader::LoaderWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    struct Req {
        void* wait_ev;  // main-stream progress the loader stream must see first
        bool exact;
    };
    std::deque<Req> req;
    std::deque<shared_ptr<Batch>> done;
    std::exception_ptr err;
    bool stop = false;
};

// Opt-in (MARIUS_LOADER_THREAD=1): measured neutral at the current kernel times (0.760 vs 0.767 ms per step) — the step is bound by the
// training stream's own kernels, not by launch issue; kept for when it is not.
static bool loader_thread_enabled() {
    const char* e = getenv("MARIUS_LOADER_THREAD");
    return e && e[0] == '1';
}

void DataLoader::nextEpoch(bool write) {
    if (!partitioned()) return;
    drain_worker();
    // the two tables' write-backs are independent files and independent staging buffers: side by side (each is bound by its own
    // device-to-host copy + page-cache writes; cfg5 scale: 4.2 s one after the other)
    std::string err;
    std::thread st;
    if (pb_state_ && write) {
        const auto dev_index = pb_state_->device_.index();
        c10::hip::getCurrentHIPStream(dev_index).synchronize();  // (the helper thread's "current stream" is not this thread's: drain here)
        st = std::thread([this, write, &err, dev_index] {
            try {
                if (hipSetDevice(dev_index) != hipSuccess) throw MariusRuntimeException("nextEpoch: hipSetDevice failed in the write-back thread");
                pb_state_->unload(write);
            } catch (const std::exception& e) {
                err = e.what();
            }
        });
    } else if (pb_state_) {
        pb_state_->unload(write);
    }
    try {
        pb_embeddings_->unload(write);
    } catch (...) {
        if (st.joinable()) st.join();
        throw;
    }
    if (st.joinable()) st.join();
    if (!err.empty()) throw MariusRuntimeException(err);
}

// Every edge of an active bucket must have both endpoints in the buffer, or a batch is handed rows of partitions that are on disk (the
// reference's index_select throws on the -1 such a node maps to).  That follows from two facts, checked where they are cheap (a
// reduction + .item() per buffer state made the host wait for the whole preceding state at every swap): (1) the edge list really is sorted by
// edge bucket and the bucket sizes describe it — ONE pass over the edges on the device, the first time a buffer state is laid out; (2) every
// bucket assigned to a buffer state has both its partitions in that state — a host loop over a few dozen pairs per state.
void DataLoader::validate_edge_buckets() {
    // the verdict is kept for exactly the edge list it was reached on: storage pointer, row count and ATen version (replacing or
    // reordering edges_->data_ after setEdgeBucketSizes used to go unnoticed)
    const void* key_ptr = edges_->data_.defined() ? edges_->data_.data_ptr() : nullptr;
    const int64_t key_rows = edges_->dim0_size_;
    const uint32_t key_ver = edges_->data_.defined() ? (uint32_t)edges_->data_._version() : 0u;
    if (buckets_validated_ && validated_ptr_ == key_ptr && validated_rows_ == key_rows && validated_version_ == key_ver) return;
    auto dev = edges_->device_;
    const int64_t P = pb_embeddings_->options_->num_partitions;
    const int64_t ps = (pb_embeddings_->dim0_size_ + P - 1) / P;
    Tensor starts = torch::from_blob(edge_bucket_starts_.data(), {(int64_t)edge_bucket_starts_.size()}, torch::kInt64).clone().to(dev);
    const int64_t E = edges_->dim0_size_, piece = 1ll << 24;  // (a few int64 temporaries of `piece` elements live at once: 128 MB each)
    Tensor bad = torch::zeros({}, i64(dev));
    for (int64_t lo = 0; lo < E; lo += piece) {
        const int64_t n = std::min(piece, E - lo);
        Tensor e = edges_->data_.narrow(0, lo, n);
        Tensor bucket = e.select(1, 0).to(torch::kInt64).floor_divide(ps) * P + e.select(1, -1).to(torch::kInt64).floor_divide(ps);
        Tensor expect = torch::bucketize(torch::arange(lo, lo + n, i64(dev)), starts, /*out_int32=*/false, /*right=*/true) - 1;
        bad += bucket.ne(expect).sum();
    }
    const int64_t nbad = bad.item<int64_t>();
    if (nbad != 0)
        throw MariusRuntimeException("DataLoader: " + std::to_string(nbad) +
                                     " edges have an endpoint outside the partitions in memory whenever their bucket is active (edge list not sorted by edge "
                                     "bucket, or wrong edge_bucket_sizes)");
    buckets_validated_ = true;
    validated_ptr_ = key_ptr;
    validated_rows_ = key_rows;
    validated_version_ = key_ver;
}

void DataLoader::setActiveEdges() {
    auto dev = edges_->device_;
    const int64_t P = pb_embeddings_->options_->num_partitions;
    validate_edge_buckets();
    Tensor buckets = edge_buckets_per_buffer_.at(buffer_cursor_);
    auto b = buckets.accessor<int64_t, 2>();
    {
        Tensor st = buffer_states_.at(buffer_cursor_).to(torch::kCPU, torch::kInt64).contiguous();
        const int64_t* sp = st.data_ptr<int64_t>();
        auto resident = [&](int64_t part) { return std::find(sp, sp + st.numel(), part) != sp + st.numel(); };
        for (int64_t i = 0; i < buckets.size(0); ++i)
            if (!resident(b[i][0]) || !resident(b[i][1]))
                throw MariusRuntimeException("DataLoader: edge bucket (" + std::to_string(b[i][0]) + ", " + std::to_string(b[i][1]) + ") is assigned to buffer state " +
                                             std::to_string(buffer_cursor_) + ", which does not hold both partitions (outside the partitions in memory)");
    }
    std::vector<Tensor> parts;
    for (int64_t i = 0; i < buckets.size(0); ++i) {
        const int64_t id = b[i][0] * P + b[i][1];
        const int64_t n = edge_bucket_starts_[id + 1] - edge_bucket_starts_[id];
        if (n > 0) parts.push_back(edges_->data_.narrow(0, edge_bucket_starts_[id], n));
    }
    const int64_t cols = edges_->dim1_size_;
    if (parts.empty()) {
        active_edges_ = torch::empty({0, cols}, i64(dev));
        return;
    }
    Tensor act = torch::cat(parts).to(torch::kInt64);
    Tensor g2l = pb_embeddings_->getGlobalToLocalMapDevice();  // graph_storage.cpp:395-420: node columns -> buffer rows
    (void)dev;
    std::vector<Tensor> columns{g2l.index_select(0, act.select(1, 0))};
    if (cols == 3) columns.push_back(act.select(1, 1));
    columns.push_back(g2l.index_select(0, act.select(1, -1)));
    active_edges_ = torch::stack(columns, 1).contiguous();
}

void* DataLoader::gate_event() {
    if (!gate_event_) {
        hipEvent_t e;
        HIPCHECK(hipEventCreateWithFlags(&e, order_event_flags()));
        gate_event_ = e;
    }
    return gate_event_;
}

bool DataLoader::hasNextBatch() {
    if (batches_left_ > 0 || !partitioned()) return batches_left_ > 0;
    // dataloader.cpp:308-340: all batches of this buffer state are done -> swap and lay out the next state's batches
    while (batches_left_ == 0 && pb_embeddings_->hasSwap()) {
        pb_embeddings_->performNextSwap();
        if (pb_state_) pb_state_->performNextSwap();
        ++buffer_cursor_;
        initializeBatches(true);
    }
    return batches_left_ > 0;
}

struct DataLoader::ShuffleAhead {
    std::thread th;
    std::vector<uint32_t> start, after;  // generator state the permutation was drawn from / left behind
    Tensor perm;                         // pinned host int64 [n]
    int64_t n = 0;
    double seconds = 0;
};

int64_t DataLoader::wordsPerEpoch(bool with_permutation) {
    int64_t words = 0;
    if (with_permutation && num_edges_ > 0)  // ATen randperm_cpu: n - 1 32-bit draws below 2^32 / 20 elements, 2 n above (rng.hip)
        words += ((uint64_t)num_edges_ >= (0xffffffffull / 20ull)) ? 2 * num_edges_ : num_edges_ - 1;
    if (negative_sampler_ && negative_sampler_->num_negatives_ >= 0) {  // two getNegatives per batch (dataloader.cpp:498-503), a fixed count per call
        const int n_deg = (int)(negative_sampler_->num_negatives_ * negative_sampler_->degree_fraction_);
        const int64_t batches = full_batches_only_ ? num_edges_ / batch_size_ : (num_edges_ + batch_size_ - 1) / batch_size_;
        words += 2 * batches * marius_negatives_raw_words(graph_->num_nodes_in_memory_, batch_size_, negative_sampler_->num_chunks_, negative_sampler_->num_negatives_, n_deg);
    }
    return words;
}

void DataLoader::start_shuffle_ahead() {
    const char* e0 = getenv("MARIUS_SHUFFLE_AHEAD");
    const char* e1 = getenv("MARIUS_SHUFFLE_AHEAD_MIN");  // below this many edges the serial draw is cheaper than a thread (tests set 0)
    const bool enabled = !(e0 && e0[0] == '0');
    const int64_t min_edges = e1 ? (int64_t)atoll(e1) : (int64_t)200000;
    if (!enabled || !train_ || num_edges_ < min_edges || !negative_sampler_ || negative_sampler_->num_negatives_ < 0) return;
    if (negative_sampler_->local_filter_mode_ != LocalFilterMode::DEG) return;
    // how many edges the NEXT permutation covers: all of them, or (out-of-core) the edge buckets of the next buffer state — the layouts of
    // all states are fixed when the epoch's ordering is drawn, so the count is known now (the state after the last one belongs to the next
    // epoch, whose ordering has not been drawn yet: no prediction there)
    int64_t next_n = num_edges_;
    if (partitioned()) {
        if (buffer_cursor_ + 1 >= (int64_t)edge_buckets_per_buffer_.size()) return;
        const int64_t P = pb_embeddings_->options_->num_partitions;
        Tensor buckets = edge_buckets_per_buffer_.at(buffer_cursor_ + 1);
        auto b = buckets.accessor<int64_t, 2>();
        next_n = 0;
        for (int64_t i = 0; i < buckets.size(0); ++i) {
            const int64_t id = b[i][0] * P + b[i][1];
            next_n += edge_bucket_starts_[id + 1] - edge_bucket_starts_[id];
        }
        if (next_n <= 0) return;
    }
    // words this epoch's sampling (and whoever else draws before the next epoch) will consume
    const int64_t words = wordsPerEpoch(false) + words_between_epochs_;
    generator_->to_host();  // right after this epoch's randperm: the state lives on the host
    auto* a = new ShuffleAhead();
    a->n = next_n;
    a->start.assign((const uint32_t*)generator_->state_host_.data_ptr<int32_t>(), (const uint32_t*)generator_->state_host_.data_ptr<int32_t>() + MARIUS_MT_STATE_WORDS);
    try {
        a->perm = torch::empty({next_n}, torch::TensorOptions().dtype(torch::kInt64).pinned_memory(true));
    } catch (const std::exception&) {  // no room for a second (pinned) permutation: the serial draw at the boundary stays
        delete a;
        return;
    }
    int64_t* out = a->perm.data_ptr<int64_t>();
    a->th = std::thread([a, words, out] {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint32_t> scratch(1 << 20);
        for (int64_t done = 0; done < words;) {
            const int64_t m = std::min<int64_t>(words - done, (int64_t)scratch.size());
            marius_mt19937_fill_host(a->start.data(), scratch.data(), m);
            done += m;
        }
        a->after = a->start;  // `start` now is the state at the epoch's end: where the next permutation begins
        marius_mt19937_randperm_host(a->after.data(), out, a->n);
        a->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    });
    ahead_ = a;
}

bool DataLoader::take_shuffle_ahead(Tensor& perm) {
    if (!ahead_) return false;
    ShuffleAhead* a = ahead_;
    ahead_ = nullptr;
    a->th.join();
    bool hit = false;
    if (a->n == num_edges_) {
        generator_->to_host();
        if (std::memcmp(generator_->state_host_.data_ptr<int32_t>(), a->start.data(), MARIUS_MT_STATE_WORDS * 4) == 0) {
            std::memcpy(generator_->state_host_.data_ptr<int32_t>(), a->after.data(), MARIUS_MT_STATE_WORDS * 4);
            perm = a->perm;
            hit = true;
        }
    }
    (hit ? shuffle_ahead_hits_ : shuffle_ahead_misses_)++;
    delete a;
    return hit;
}

void DataLoader::initializeBatches(bool shuffle) {
    drain_worker();  // nothing may be preparing while the generator draws the epoch permutation on this thread
    if (partitioned()) {
        setActiveEdges();
        num_edges_ = active_edges_.size(0);
    }
    // setActiveEdges (dataloader.cpp:176-182): randperm over all edges on the generator stream, consumed even for evaluation
    Tensor perm;
    if (!(shuffle && take_shuffle_ahead(perm))) {
        if (ahead_) {  // a permutation drawn ahead that this call cannot use (shuffle off): drop it
            Tensor unused;
            take_shuffle_ahead(unused);
        }
        perm = num_edges_ > 0 ? generator_->randperm(num_edges_) : torch::empty({0}, torch::kInt64);
    }
    active_perm_ = shuffle ? perm.to(edges_->device_) : torch::arange(num_edges_, i64(edges_->device_));
    total_batches_ = (num_edges_ + batch_size_ - 1) / batch_size_;
    batches_left_ = total_batches_;
    prepared_left_ = total_batches_;
    batch_id_ = 0;
    if (shuffle) start_shuffle_ahead();
}

DataLoader::~DataLoader() {
    if (ahead_) {
        ahead_->th.join();
        delete ahead_;
        ahead_ = nullptr;
    }
    if (worker_) {
        {
            std::lock_guard<std::mutex> lk(worker_->m);
            worker_->stop = true;
        }
        worker_->cv.notify_all();
        if (worker_->th.joinable()) worker_->th.join();
        delete worker_;
        worker_ = nullptr;
    }
    next_.reset();
    for (auto& e : ev_pool_)
        if (e) (void)hipEventDestroy((hipEvent_t)e);
    for (auto& e : ev_main_)
        if (e) (void)hipEventDestroy((hipEvent_t)e);
    if (gate_event_) (void)hipEventDestroy((hipEvent_t)gate_event_);
    delete (c10::hip::HIPStream*)loader_stream_;
}

void DataLoader::post_prepare(bool exact_unique) {
    const auto dev_index = edges_->device_.index();
    c10::hip::HIPStream main = c10::hip::getCurrentHIPStream(dev_index);
    // Everything the compute stream has been given so far (the previous step; at an epoch start also the permutation upload) must
    // be complete before the loader runs: it orders the reads of active_perm_ and makes the caching allocator's per-stream reuse
    // safe (blocks freed by the previous batch return to the loader stream's pool while that step may still be executing).
    hipEvent_t em = (hipEvent_t)ev_main_[ev_main_next_];
    ev_main_next_ = (ev_main_next_ + 1) & 3;
    static const bool gate_env = [] { const char* e = getenv("MARIUS_LOADER_GATE"); return !(e && e[0] == '0'); }();
    if (gate_env && gate_valid_ && gate_event_ && !worker_) em = (hipEvent_t)gate_event_;  // the previous step's gradients exist: early enough (see header)
    else HIPCHECK(hipEventRecord(em, main.stream()));
    prepared_left_--;
    pending_ = true;
    next_exact_ = exact_unique;
    if (!worker_) {  // inline: prepare on the loader stream from this thread
        c10::hip::HIPStream loader = *(c10::hip::HIPStream*)loader_stream_;
        HIPCHECK(hipStreamWaitEvent(loader.stream(), em, 0));
        shared_ptr<Batch> b;
        {
            StreamScope scope(loader);
            b = prepareBatch(exact_unique);
        }
        b->ready_ = ev_pool_[ev_next_];
        ev_next_ = (ev_next_ + 1) & 3;
        HIPCHECK(hipEventRecord((hipEvent_t)b->ready_, loader.stream()));
        next_ = b;
        return;
    }
    {
        std::lock_guard<std::mutex> lk(worker_->m);
        worker_->req.push_back({(void*)em, exact_unique});
    }
    worker_->cv.notify_all();
}

shared_ptr<Batch> DataLoader::take_prepared() {
    pending_ = false;
    if (!worker_) {
        shared_ptr<Batch> b = next_;
        next_.reset();
        return b;
    }
    std::unique_lock<std::mutex> lk(worker_->m);
    worker_->cv.wait(lk, [&] { return !worker_->done.empty() || worker_->err; });
    if (worker_->err) {
        std::exception_ptr e = worker_->err;
        worker_->err = nullptr;
        std::rethrow_exception(e);
    }
    shared_ptr<Batch> b = worker_->done.front();
    worker_->done.pop_front();
    return b;
}

void DataLoader::drain_worker() {
    if (pending_) {
        try {
            (void)take_prepared();
        } catch (...) {
        }
    }
    held_.reset();
    held_prev_.reset();
    next_.reset();
    gate_valid_ = false;  // an epoch boundary uploads a new permutation on the training stream: the next preparation waits for all of it
}

shared_ptr<Batch> DataLoader::getBatch(bool exact_unique) {
    if (!run_ahead_ && !pending_) {
        batches_left_--;
        prepared_left_--;
        auto b = prepareBatch(exact_unique);
        last_num_unique_ = b->num_unique_dev_;
        return b;
    }
    const auto dev_index = edges_->device_.index();
    c10::hip::HIPStream main = c10::hip::getCurrentHIPStream(dev_index);
    if (!loader_stream_) {
        loader_stream_ = new c10::hip::HIPStream(c10::hip::getStreamFromExternal((hipStream_t)aux_stream(dev_index, AUX_LOADER), dev_index));
        for (auto& e : ev_pool_) {
            hipEvent_t ev;
            HIPCHECK(hipEventCreateWithFlags(&ev, order_event_flags()));
            e = ev;
        }
        for (auto& e : ev_main_) {
            hipEvent_t ev;
            HIPCHECK(hipEventCreateWithFlags(&ev, order_event_flags()));
            e = ev;
        }
        if (loader_thread_enabled()) {
            worker_ = new LoaderWorker();
            worker_->th = std::thread([this, dev_index] {
                (void)hipSetDevice(dev_index);
                c10::hip::HIPStream loader = *(c10::hip::HIPStream*)loader_stream_;
                LoaderWorker& w = *worker_;
                for (;;) {
                    LoaderWorker::Req r;
                    {
                        std::unique_lock<std::mutex> lk(w.m);
                        w.cv.wait(lk, [&] { return w.stop || !w.req.empty(); });
                        if (w.stop) return;
                        r = w.req.front();
                        w.req.pop_front();
                    }
                    try {
                        if (hipStreamWaitEvent(loader.stream(), (hipEvent_t)r.wait_ev, 0) != hipSuccess) throw MariusRuntimeException("loader thread: hipStreamWaitEvent failed");
                        shared_ptr<Batch> b;
                        {
                            StreamScope scope(loader);
                            b = prepareBatch(r.exact);
                        }
                        b->ready_ = ev_pool_[ev_next_];
                        ev_next_ = (ev_next_ + 1) & 3;
                        if (hipEventRecord((hipEvent_t)b->ready_, loader.stream()) != hipSuccess) throw MariusRuntimeException("loader thread: hipEventRecord failed");
                        std::lock_guard<std::mutex> lk(w.m);
                        w.done.push_back(b);
                    } catch (...) {
                        std::lock_guard<std::mutex> lk(w.m);
                        w.err = std::current_exception();
                    }
                    w.cv.notify_all();
                }
            });
        }
    }
    if (pending_ && next_exact_ != exact_unique) throw MariusRuntimeException("DataLoader: exact_unique changed while a batch was prepared ahead");
    if (!pending_) post_prepare(exact_unique);
    shared_ptr<Batch> batch = take_prepared();
    batches_left_--;
    last_num_unique_ = batch->num_unique_dev_;
    held_prev_ = held_;  // the batch before the previous one is released here: the updates of the previous step may still be reading its maps
    held_ = batch;
    if (run_ahead_ && prepared_left_ > 0) post_prepare(exact_unique);
    HIPCHECK(hipStreamWaitEvent(main.stream(), (hipEvent_t)batch->ready_, 0));
    return batch;
}

shared_ptr<Batch> DataLoader::prepareBatch(bool exact_unique) {
    auto st = cur_stream();
    auto dev = edges_->device_;
    auto batch = std::make_shared<Batch>(train_);
    batch->batch_id_ = (int)batch_id_;
    batch->start_idx_ = batch_id_ * batch_size_;
    batch->batch_size_ = std::min(batch_size_, num_edges_ - batch->start_idx_);
    batch_id_++;
    const int64_t B = batch->batch_size_;
    const int cols = (int)edges_->dim1_size_;
    // edge_sampler_->getEdges (edge.cpp:12-14): slice of the shuffled edges, cast to int64
    Tensor edges = torch::empty({B, cols}, i64(dev));
    if (partitioned())
        mcheck(marius_select_edges(active_edges_.data_ptr(), 1, cols, ip(active_perm_), batch->start_idx_, B, ip(edges), st));
    else
        mcheck(marius_select_edges(edges_->data_.data_ptr(), edges_->dtype_ == torch::kInt64 ? 1 : 0, cols, ip(active_perm_), batch->start_idx_, B, ip(edges), st));
    // negativeSample (dataloader.cpp:498-503): inverse (src corruption) first, then dst.  The reference's filter tensor (rows in nonzero() order)
    // needs its size on the host — a stream drain; the fused step keeps the uncompacted [C * n_deg, 2] form instead, whose (-1, -1) rows every
    // consumer behind the C-ABI ignores (measured at degree_fraction 0.5: 0.82 -> ms per step with the two drains per batch gone: see DESIGN)
    negative_sampler_->compact_filter_ = exact_unique;
    std::tie(batch->src_neg_indices_, batch->src_neg_filter_) = negative_sampler_->getNegatives(graph_, edges, true);
    std::tie(batch->dst_neg_indices_, batch->dst_neg_filter_) = negative_sampler_->getNegatives(graph_, edges, false);
    const int64_t CN = batch->dst_neg_indices_.numel();
    const int64_t L = 2 * B + 2 * CN;
    // map_tensors (util.cpp:180-205)
    all_ids_ = torch::empty({L}, i64(dev));
    uniq_ = torch::empty({L}, i64(dev));
    inverse_ = torch::empty({L}, i64(dev));
    perm_ = torch::empty({L}, i32(dev));
    seg_ = torch::empty({L + 1}, i32(dev));
    count_ = torch::empty({1}, i64(dev));  // (written by the sort's emit launch — or its empty-input launch — before anything reads it: no zero fill)
    const size_t wsb = marius_sort_unique_workspace_bytes(L);
    if (!sort_ws_.defined() || (size_t)sort_ws_.numel() < wsb) sort_ws_ = torch::zeros({(int64_t)wsb}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));  // zeroed: the sort's control block
    batch->global_edges_ = edges;
    batch->edges_ = torch::empty({B, cols}, i64(dev));
    const bool plan = train_ && (run_ahead_ || plan_ahead_);  // the fused update's index work, done here (this stream runs a step ahead of the gradients)
    const bool rels = train_ && cols == 3 && num_relations_ > 0;  // index_select backward into [R, d] wants the relation ids grouped: sort them here
    if (plan) batch->occ_plan_ = torch::empty({(int64_t)marius_segment_plan_bytes(L)}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
    Tensor rel_ids;
    if (rels) {
        batch->rel_uniq_ = torch::empty({B}, i64(dev));
        batch->rel_inverse_ = torch::empty({B}, i64(dev));
        batch->rel_perm_ = torch::empty({B}, i32(dev));
        batch->rel_seg_ = torch::empty({B + 1}, i32(dev));
        batch->rel_count_ = torch::empty({1}, i64(dev));
        if (plan) batch->rel_plan_ = torch::empty({(int64_t)marius_segment_plan_bytes(B)}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
        const size_t wsr = marius_sort_unique_workspace_bytes(B);  // its own workspace: the fused launch works on both lists at once
        if (!sort_ws_rel_.defined() || (size_t)sort_ws_rel_.numel() < wsr) sort_ws_rel_ = torch::zeros({(int64_t)wsr}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
    }
    // The separate launches (default), or — MARIUS_MAPS=fused — ONE persistent launch for the whole chain of both id lists (marius_prepare_maps: ids
    // assembled, unique maps, batch-local edges, segment plans).  Same bits either way; the one-launch form measured 6 % slower per step inside the
    // pipeline (its ~90 resident workgroups hold CU slots the main stream's persistent kernels want: DESIGN 4.2), so it is opt-in
    marius_map_job jobs[2] = {};
    jobs[0].ids_out = ip(all_ids_);
    jobs[0].edges = ip(edges);
    jobs[0].src_neg = ip(batch->src_neg_indices_);
    jobs[0].dst_neg = ip(batch->dst_neg_indices_);
    jobs[0].B = B;
    jobs[0].CN = CN;
    jobs[0].n = L;
    jobs[0].edge_cols = cols;
    jobs[0].col = -1;
    jobs[0].key_bits = key_bits_;
    jobs[0].uniq = ip(uniq_);
    jobs[0].inverse = ip(inverse_);
    jobs[0].perm = perm_.data_ptr<int32_t>();
    jobs[0].seg_offsets = seg_.data_ptr<int32_t>();
    jobs[0].num_unique_dev = ip(count_);
    jobs[0].plan = plan ? batch->occ_plan_.data_ptr() : nullptr;
    jobs[0].edges_out = ip(batch->edges_);
    jobs[0].workspace = sort_ws_.data_ptr();
    jobs[0].workspace_bytes = (size_t)sort_ws_.numel();
    if (rels) {
        rel_ids = torch::empty({B}, i64(dev));
        jobs[1].ids_out = ip(rel_ids);
        jobs[1].edges = ip(edges);
        jobs[1].B = B;
        jobs[1].n = B;
        jobs[1].edge_cols = cols;
        jobs[1].col = 1;
        jobs[1].key_bits = key_bits_for(num_relations_);
        jobs[1].uniq = ip(batch->rel_uniq_);
        jobs[1].inverse = ip(batch->rel_inverse_);
        jobs[1].perm = batch->rel_perm_.data_ptr<int32_t>();
        jobs[1].seg_offsets = batch->rel_seg_.data_ptr<int32_t>();
        jobs[1].num_unique_dev = ip(batch->rel_count_);
        jobs[1].plan = plan ? batch->rel_plan_.data_ptr() : nullptr;
        jobs[1].workspace = sort_ws_rel_.data_ptr();
        jobs[1].workspace_bytes = (size_t)sort_ws_rel_.numel();
    }
    const int njobs = rels ? 2 : 1;
    if (marius_prepare_maps_preferred() && marius_prepare_maps_supported(jobs, njobs)) {
        mcheck(marius_prepare_maps(jobs, njobs, st));
    } else {
        mcheck(marius_assemble_ids(ip(edges), B, cols, ip(batch->src_neg_indices_), ip(batch->dst_neg_indices_), CN, ip(all_ids_), st));
        mcheck(marius_sort_unique(ip(all_ids_), L, key_bits_, ip(uniq_), ip(inverse_), perm_.data_ptr<int32_t>(), seg_.data_ptr<int32_t>(), ip(count_),
                                  sort_ws_.data_ptr(), (size_t)sort_ws_.numel(), st));
        mcheck(marius_remap_edges(ip(edges), ip(inverse_), B, cols, ip(batch->edges_), st));
        if (plan) mcheck(marius_segment_plan(perm_.data_ptr<int32_t>(), ip(inverse_), seg_.data_ptr<int32_t>(), ip(uniq_), L, batch->occ_plan_.data_ptr(), st));
        if (rels) {
            rel_ids = edges.select(1, 1).contiguous();
            mcheck(marius_sort_unique(ip(rel_ids), B, key_bits_for(num_relations_), ip(batch->rel_uniq_), ip(batch->rel_inverse_),
                                      batch->rel_perm_.data_ptr<int32_t>(), batch->rel_seg_.data_ptr<int32_t>(), ip(batch->rel_count_), sort_ws_rel_.data_ptr(),
                                      (size_t)sort_ws_rel_.numel(), st));
            if (plan)
                mcheck(marius_segment_plan(batch->rel_perm_.data_ptr<int32_t>(), ip(batch->rel_inverse_), batch->rel_seg_.data_ptr<int32_t>(), ip(batch->rel_uniq_), B,
                                           batch->rel_plan_.data_ptr(), st));
        }
    }
    batch->src_neg_indices_mapping_ = inverse_.narrow(0, 2 * B, CN).view(batch->src_neg_indices_.sizes());
    batch->dst_neg_indices_mapping_ = inverse_.narrow(0, 2 * B + CN, CN).view(batch->dst_neg_indices_.sizes());
    // exact_unique: the reference's tensor [U] (one 8-byte D2H copy); otherwise capacity-sized with a zero tail (no host sync)
    const int64_t U = exact_unique ? count_.item<int64_t>() : L;
    batch->unique_node_indices_ = uniq_.narrow(0, 0, U);
    batch->occ_perm_ = perm_;
    batch->occ_inverse_ = inverse_;
    batch->occ_seg_offsets_ = seg_;
    batch->num_unique_dev_ = count_;
    return batch;
}

void DataLoader::loadGPUParameters(shared_ptr<Batch> batch) {
    if (train_ && node_embeddings_state_) {
        const int64_t U = batch->unique_node_indices_.size(0), d = node_embeddings_->dim1_size_;
        batch->node_embeddings_ = torch::empty({U, d}, node_embeddings_->data_.options());
        batch->node_embeddings_state_ = torch::empty({U, d}, node_embeddings_->data_.options());
        mcheck(marius_gather_rows2(fp(node_embeddings_->data_), fp(node_embeddings_state_->data_), node_embeddings_->data_.stride(0),
                                   ip(batch->unique_node_indices_), U, (int32_t)d, fp(batch->node_embeddings_), fp(batch->node_embeddings_state_), d,
                                   cur_stream()));
    } else {
        batch->node_embeddings_ = node_embeddings_->indexRead(batch->unique_node_indices_);
    }
}

void DataLoader::updateEmbeddings(shared_ptr<Batch> batch, bool gpu) {
    (void)gpu;
    node_embeddings_->indexAdd(batch->unique_node_indices_, batch->node_gradients_);          // graph_storage.cpp:289
    node_embeddings_state_->indexAdd(batch->unique_node_indices_, batch->node_state_update_);  // graph_storage.cpp:319-323
}

// ------------------------------------------------------------------------------------------------ trainer / evaluator
void SynchronousTrainer::train_one(bool fused) {
    fused = fused && model_->fused_ok() && !model_->has_post_hook();  // user plug-ins (virtual calls + autograd) and an embedding layer with a post-hook
                                                                       // (gathered rows through marius_layer_post_hook) train through the API-granular path
    dataloader_->run_ahead_ = fused;
    dataloader_->num_relations_ = fused ? model_->decoder_->num_relations_ : 0;
    if (fused) {
        auto batch = dataloader_->getBatch(/*exact_unique=*/false);  // no host sync anywhere in the step
        // capacity-sized id list (no host sync on the unique count): gather the U real rows only
        auto mem = std::dynamic_pointer_cast<InMemory>(dataloader_->node_embeddings_);
        // A device-resident table is read in place by global id: no gathered [U, d] copy is written and re-read (MARIUS_TABLE_DIRECT=0: the
        // gathered form, for A/B runs).
        static const bool direct_env = [] { const char* e = getenv("MARIUS_TABLE_DIRECT"); return !(e && e[0] == '0'); }();
        // The partition buffer's slab is such a table too: batches of a buffer state carry slab row ids (edges remapped by setActiveEdges,
        // negatives drawn from the rows in memory), and the slab stays where it is across swaps.
        auto pbs = std::dynamic_pointer_cast<PartitionBufferStorage>(dataloader_->node_embeddings_);
        const bool direct = direct_env && ((mem && mem->data_.is_cuda()) || (pbs && pbs->data_.defined() && pbs->data_.is_cuda()));
        // fp16 operand records need magnitude bounds (Model::bind_ranges): a device-resident table is scanned the first time the trainer meets it
        // and kept current by the fused update; the partition buffer keeps its own running bound of the slab (MARIUS_FLASH_F16=0: bf16 records)
        if (flash_f16_env()) {
            if (mem && mem->data_.is_cuda() && !model_->tracks(mem->data_)) model_->track_ranges(mem->data_);
            if (pbs && direct) {
                pbs->buffer_->enable_absmax();
                model_->external_node_bound_ = pbs->buffer_->absmax_;
            } else {
                model_->external_node_bound_ = Tensor();
            }
        }
        batch->table_to_update_ = dataloader_->node_embeddings_->data_;
        if (!direct)
            batch->node_embeddings_ = (mem && batch->num_unique_dev_.defined()) ? mem->indexReadCounted(batch->unique_node_indices_, batch->num_unique_dev_)
                                                                                  : dataloader_->node_embeddings_->indexRead(batch->unique_node_indices_);
        model_->ev_grads_ = dataloader_->gate_event();
        try {
            model_->backward_into_tables(batch, dataloader_->node_embeddings_->data_, dataloader_->node_embeddings_state_->data_, direct);
        } catch (...) {
            model_->ev_grads_ = nullptr;
            throw;
        }
        model_->ev_grads_ = nullptr;
        dataloader_->gate_valid_ = true;
    } else {  // API-granular path, call for call the reference's loop (trainer.cpp:106-138)
        if (model_->ranges_valid_) model_->drop_ranges();  // updateEmbeddings writes the table without tracking its magnitude: the table-wide bound is void
        auto batch = dataloader_->getBatch(true);               // (the step itself bounds the rows it gathers: Model::bind_ranges)
        dataloader_->loadGPUParameters(batch);
        model_->train_batch(batch);
        dataloader_->updateEmbeddings(batch, true);
        batch->clear();
    }
}

void SynchronousTrainer::train(int num_epochs) {
    for (int epoch = 0; epoch < num_epochs; ++epoch) {
        dataloader_->loadStorage();  // out-of-core tables: a fresh ordering per epoch (trainer.cpp:96 -> setTrainSet -> loadStorage); else a no-op
        dataloader_->initializeBatches(true);
        c10::hip::getCurrentHIPStream().synchronize();
        auto t0 = std::chrono::steady_clock::now();
        while (dataloader_->hasNextBatch()) train_one(fused_update_);
        c10::hip::getCurrentHIPStream().synchronize();
        dataloader_->nextEpoch();  // out-of-core: write the buffer back (inside the timed region, as in the reference)
        last_epoch_seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        last_edges_per_second_ = (double)dataloader_->edges_->dim0_size_ / last_epoch_seconds_;  // trainer.cpp:156-159
    }
}

void SynchronousTrainer::train_steps(int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        if (!dataloader_->hasNextBatch()) {
            dataloader_->nextEpoch();
            dataloader_->loadStorage();
            dataloader_->initializeBatches(true);
        }
        train_one(fused_update_);
    }
}

void PipelineTrainer::train(int num_epochs) {
    if (!stale_parameters_) {  // device-resident parameters: gathered by the compute stage, i.e. the synchronous result; the loader runs ahead
        SynchronousTrainer t(dataloader_, model_);
        t.train(num_epochs);
        last_epoch_seconds_ = t.last_epoch_seconds_;
        last_edges_per_second_ = t.last_edges_per_second_;
        return;
    }
    for (int epoch = 0; epoch < num_epochs; ++epoch) {
        dataloader_->loadStorage();
        dataloader_->initializeBatches(true);
        c10::hip::getCurrentHIPStream().synchronize();
        auto t0 = std::chrono::steady_clock::now();
        dataloader_->run_ahead_ = false;
        dataloader_->num_relations_ = 0;
        std::deque<shared_ptr<Batch>> in_flight;
        while (dataloader_->hasNextBatch() || !in_flight.empty()) {
            while ((int)in_flight.size() < staleness_bound_ && dataloader_->hasNextBatch()) {  // LoadBatchWorker: admit, prepare, read parameters
                auto b = dataloader_->getBatch(true);
                dataloader_->loadGPUParameters(b);
                in_flight.push_back(b);
            }
            auto b = in_flight.front();  // ComputeWorker + UpdateBatchWorker
            in_flight.pop_front();
            model_->train_batch(b);
            dataloader_->updateEmbeddings(b, true);
            b->clear();
        }
        c10::hip::getCurrentHIPStream().synchronize();
        dataloader_->nextEpoch();
        last_epoch_seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        last_edges_per_second_ = (double)dataloader_->edges_->dim0_size_ / last_epoch_seconds_;
    }
}

std::vector<double> SynchronousEvaluator::evaluate() {
    model_->reporter_->clear();
    // partitioned evaluation (storage.full_graph_evaluation: false): the evaluation edges are walked buffer state by buffer state like the
    // training edges, negatives come from the nodes in memory (dataloader.cpp:296-345 applies to both modes)
    dataloader_->loadStorage();
    dataloader_->initializeBatches(false);
    while (dataloader_->hasNextBatch()) {
        auto batch = dataloader_->getBatch(true);
        dataloader_->loadGPUParameters(batch);
        model_->evaluate_batch(batch);
    }
    dataloader_->nextEpoch(/*write=*/false);  // nothing was modified
    return model_->reporter_->report();
}

}  // namespace marius_amd
