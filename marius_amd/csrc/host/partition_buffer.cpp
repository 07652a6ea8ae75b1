// Partition buffer in HBM + edge-bucket orderings (see partition_buffer.h).
#include "partition_buffer.h"

#include <c10/hip/HIPStream.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

namespace marius_amd {

#define PB_HIPCHECK(x)                                                                                               \
    do {                                                                                                              \
        hipError_t e_ = (x);                                                                                          \
        if (e_ != hipSuccess) throw MariusRuntimeException(std::string("PartitionBuffer HIP: ") + hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------ file
PartitionedFile::PartitionedFile(const std::string& filename) : filename_(filename) {
    fd_ = open(filename_.c_str(), O_RDWR);
    if (fd_ == -1) throw std::runtime_error("");  // buffer.cpp:78-81: logs, then a bare runtime_error
}
PartitionedFile::~PartitionedFile() {
    if (fd_ != -1) close(fd_);
}
// One pread / pwrite stream moves ~5 GB/s out of the page cache (or a RAM-backed file): a 2 GB partition then takes longer than a buffer
// state of cfg5's shape trains (0.7 s).  The transfer is cut into 64 MB slices handled by up to 8 threads (positional IO: no shared file
// offset), which is what lets the prefetch of the next swap finish under the compute of the current state.
static void full_io(int fd, char* buf, int64_t n, int64_t off, bool write_) {
    const int64_t slice = 64ll << 20;
    const int64_t nslices = (n + slice - 1) / slice;
    int failed = 0;
    int err_no = 0;
    static const int max_threads = [] { const char* e = getenv("MARIUS_PB_IO_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();
    const int nthreads = (int)std::min<int64_t>(nslices, max_threads);
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1) if (nslices > 1)
    for (int64_t sidx = 0; sidx < nslices; ++sidx) {
        int64_t o = sidx * slice, left = std::min<int64_t>(slice, n - o);
        while (left > 0) {  // pread / pwrite may move less than asked
            const ssize_t r = write_ ? pwrite(fd, buf + o, (size_t)left, off + o) : pread(fd, buf + o, (size_t)left, off + o);
            if (r <= 0) {
#pragma omp critical
                {
                    failed = 1;
                    err_no = errno;
                }
                break;
            }
            o += r;
            left -= r;
        }
    }
    if (failed) throw MariusRuntimeException(std::string("PartitionedFile: ") + (write_ ? "pwrite" : "pread") + " failed: " + std::strerror(err_no));
}
void PartitionedFile::readPartition(void* host_addr, const Partition& p) {
    if (host_addr == nullptr) throw std::runtime_error("");
    full_io(fd_, (char*)host_addr, p.total_size_, p.file_offset_, false);
}
void PartitionedFile::writePartition(const void* host_addr, const Partition& p) {
    if (host_addr == nullptr) throw std::runtime_error("");
    full_io(fd_, (char*)host_addr, p.total_size_, p.file_offset_, true);
}

// ------------------------------------------------------------------------------------------------ buffer
PartitionBuffer::PartitionBuffer(int capacity, int num_partitions, int fine_to_coarse_ratio, int64_t partition_size, int embedding_size,
                                 int64_t total_embeddings, torch::Dtype dtype, std::string filename, bool prefetching, torch::Device device)
    : capacity_(capacity), num_partitions_(num_partitions), fine_to_coarse_ratio_(fine_to_coarse_ratio), embedding_size_(embedding_size),
      partition_size_(partition_size), total_embeddings_(total_embeddings), dtype_(dtype), filename_(std::move(filename)), prefetching_(prefetching),
      device_(device) {
    if (dtype_ != torch::kFloat32) throw MariusRuntimeException("PartitionBuffer: float32 rows only (the gather / scatter kernels are fp32)");
    if (!device_.is_cuda()) throw MariusRuntimeException("PartitionBuffer: the buffer lives in HBM (no CPU fallback)");
    dtype_size_ = 4;
    int64_t idx = 0, off = 0;
    for (int i = 0; i < num_partitions_; ++i) {  // buffer.cpp:345-362
        Partition p;
        p.partition_id_ = i;
        p.partition_size_ = (i == num_partitions_ - 1) ? total_embeddings_ - idx : partition_size_;
        p.idx_offset_ = idx;
        p.file_offset_ = off;
        p.total_size_ = p.partition_size_ * embedding_size_ * dtype_size_;
        partition_table_.push_back(p);
        idx += p.partition_size_;
        off += p.total_size_;
    }
    file_ = std::make_unique<PartitionedFile>(filename_);
}

PartitionBuffer::~PartitionBuffer() {
    try {
        unload(true);  // buffer.cpp:366
    } catch (...) {
    }
}

char* PartitionBuffer::slot_ptr(int64_t slot) const { return (char*)buffer_tensor_view_.data_ptr() + slot * slot_bytes(); }

void PartitionBuffer::alloc_staging() {
    lanes_ = std::max(1, fine_to_coarse_ratio_);
    for (size_t i = 0; i + 1 < buffer_states_.size(); ++i) {  // widest exchange of the ordering
        int diff = 0;
        for (auto p : buffer_states_[i + 1])
            if (std::find(buffer_states_[i].begin(), buffer_states_[i].end(), p) == buffer_states_[i].end()) ++diff;
        lanes_ = std::max(lanes_, diff);
    }
    admit_mem_.assign(lanes_, nullptr);
    evict_mem_.assign(lanes_, nullptr);
    for (int i = 0; i < lanes_; ++i) {
        PB_HIPCHECK(hipHostMalloc(&admit_mem_[i], (size_t)slot_bytes(), hipHostMallocDefault));
        PB_HIPCHECK(hipHostMalloc(&evict_mem_[i], (size_t)slot_bytes(), hipHostMallocDefault));
    }
    hipStream_t s;
    PB_HIPCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    swap_stream_ = s;
    PB_HIPCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    swap_stream2_ = s;
    static const bool dev_env = [] { const char* e = getenv("MARIUS_PB_DEVICE_STAGING"); return !(e && e[0] == '0'); }();
    if (prefetching_ && dev_env && buffer_states_.size() > 1) {
        dev_admit_.assign(lanes_, nullptr);
        dev_evict_.assign(lanes_, nullptr);
        bool ok = true;
        for (int i = 0; i < lanes_ && ok; ++i)
            ok = hipMalloc(&dev_admit_[i], (size_t)slot_bytes()) == hipSuccess && hipMalloc(&dev_evict_[i], (size_t)slot_bytes()) == hipSuccess;
        if (!ok) {  // HBM is full of slab: keep the exchange at the swap point (the path below)
            (void)hipGetLastError();
            for (auto m : dev_admit_)
                if (m) (void)hipFree(m);
            for (auto m : dev_evict_)
                if (m) (void)hipFree(m);
            dev_admit_.clear();
            dev_evict_.clear();
            return;
        }
        hipEvent_t e;
        PB_HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_compute_ = e;
        PB_HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_swapped_ = e;
        PB_HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_evict_host_ = e;
        PB_HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_admit_ready_ = e;
    }
}

void PartitionBuffer::free_staging() {
    for (auto m : admit_mem_)
        if (m) (void)hipHostFree(m);
    for (auto m : evict_mem_)
        if (m) (void)hipHostFree(m);
    admit_mem_.clear();
    evict_mem_.clear();
    for (auto m : dev_admit_)
        if (m) (void)hipFree(m);
    for (auto m : dev_evict_)
        if (m) (void)hipFree(m);
    dev_admit_.clear();
    dev_evict_.clear();
    for (void** e : {&ev_compute_, &ev_swapped_, &ev_evict_host_, &ev_admit_ready_}) {
        if (*e) (void)hipEventDestroy((hipEvent_t)*e);
        *e = nullptr;
    }
    if (swap_stream_) (void)hipStreamDestroy((hipStream_t)swap_stream_);
    if (swap_stream2_) (void)hipStreamDestroy((hipStream_t)swap_stream2_);
    swap_stream_ = swap_stream2_ = nullptr;
}

void PartitionBuffer::stage_in(const Partition& p, int64_t slot, void* staging) {
    hipStream_t s = (hipStream_t)swap_stream_;
    PB_HIPCHECK(hipMemcpyAsync(slot_ptr(slot), staging, (size_t)p.total_size_, hipMemcpyHostToDevice, s));
    if (p.total_size_ < slot_bytes())  // short last partition: rows past its end read as zeros (readPartition memsets, buffer.cpp:92)
        PB_HIPCHECK(hipMemsetAsync(slot_ptr(slot) + p.total_size_, 0, (size_t)(slot_bytes() - p.total_size_), s));
}

void PartitionBuffer::scan_slot(int64_t slot, void* stream) {
    if (!absmax_.defined() || dtype_ != torch::kFloat32) return;
    mcheck(marius_table_absmax((const float*)slot_ptr(slot), partition_size_, embedding_size_, embedding_size_, absmax_.data_ptr<float>(), (marius_stream_t)stream));
}

void PartitionBuffer::enable_absmax() {
    if (absmax_.defined() || dtype_ != torch::kFloat32) return;
    absmax_ = torch::zeros({1}, torch::TensorOptions().dtype(torch::kFloat32).device(device_));
    if (!loaded_) return;
    // the slab as it is now, on the caller's stream (updates and swaps so far are ordered before it; what follows keeps the bound current)
    for (int64_t slot = 0; slot < capacity_; ++slot) scan_slot(slot, c10::hip::getCurrentHIPStream(device_.index()).stream());
}

void PartitionBuffer::load() {  // buffer.cpp:372-418
    if (loaded_) return;
    if (buffer_state_.empty()) throw MariusRuntimeException("PartitionBuffer::load: setBufferOrdering first");
    buffer_tensor_view_ = torch::zeros({(int64_t)capacity_ * partition_size_, (int64_t)embedding_size_}, torch::TensorOptions().dtype(dtype_).device(device_));
    c10::hip::getCurrentHIPStream(device_.index()).synchronize();  // the zero fill runs on torch's stream, the copies below on ours
    alloc_staging();
    for (size_t i = 0; i < buffer_state_.size(); ++i) {
        Partition& p = partition_table_[buffer_state_[i]];
        void* st = admit_mem_[i % lanes_];
        if (i >= (size_t)lanes_) PB_HIPCHECK(hipStreamSynchronize((hipStream_t)swap_stream_));  // staging lane still in flight
        file_->readPartition(st, p);
        stage_in(p, (int64_t)i, st);
        scan_slot((int64_t)i, swap_stream_);
        p.present_ = true;
        p.buffer_idx_ = (int)i;
    }
    PB_HIPCHECK(hipStreamSynchronize((hipStream_t)swap_stream_));
    loaded_ = true;
    staged_admits_.clear();
    if (prefetching_) {
        io_stop_ = false;
        io_thread_ = std::thread(&PartitionBuffer::io_loop, this);
        std::vector<int> next = getNextAdmit();
        if (!next.empty()) {
            staged_admits_ = next;
            io_submit([this, next] {
                for (size_t i = 0; i < next.size(); ++i) file_->readPartition(admit_mem_[i], partition_table_[next[i]]);
                if (dev_staging()) {
                    hipStream_t s2 = (hipStream_t)swap_stream2_;
                    for (size_t i = 0; i < next.size(); ++i)
                        PB_HIPCHECK(hipMemcpyAsync(dev_admit_[i], admit_mem_[i], (size_t)partition_table_[next[i]].total_size_, hipMemcpyHostToDevice, s2));
                    PB_HIPCHECK(hipEventRecord((hipEvent_t)ev_admit_ready_, s2));
                }
            });
        }
    }
}

void PartitionBuffer::sync() {
    if (!loaded_) return;
    if (prefetching_) io_wait();
    hipStream_t s = (hipStream_t)swap_stream_;
    c10::hip::getCurrentHIPStream(device_.index()).synchronize();  // every update enqueued so far has landed in the slab
    PB_HIPCHECK(hipStreamSynchronize(s));
    if (swap_stream2_) PB_HIPCHECK(hipStreamSynchronize((hipStream_t)swap_stream2_));
    // two staging buffers (nothing is being admitted any more: the admit lane is free): partition k + 1 comes off the device while partition k
    // is written to its file
    void* stage[2] = {evict_mem_[0], admit_mem_[0]};
    std::thread writer;
    std::string werr;
    int k = 0;
    try {
        for (auto& p : partition_table_) {
            if (!p.present_) continue;
            void* buf = stage[k & 1];
            PB_HIPCHECK(hipMemcpyAsync(buf, slot_ptr(p.buffer_idx_), (size_t)p.total_size_, hipMemcpyDeviceToHost, s));
            PB_HIPCHECK(hipStreamSynchronize(s));
            if (writer.joinable()) writer.join();  // the other buffer's write; this one's buffer was released by the join before it
            if (!werr.empty()) throw MariusRuntimeException(werr);
            Partition* pp = &p;
            writer = std::thread([this, buf, pp, &werr] {
                try {
                    file_->writePartition(buf, *pp);
                } catch (const std::exception& e) {
                    werr = e.what();
                }
            });
            p.present_ = false;
            p.buffer_idx_ = -1;
            ++k;
        }
    } catch (...) {  // a HIP error while a write-back is in flight: unwinding past a joinable std::thread would terminate the process
        if (writer.joinable()) writer.join();
        throw;
    }
    if (writer.joinable()) writer.join();
    if (!werr.empty()) throw MariusRuntimeException(werr);
}

void PartitionBuffer::unload(bool write) {  // buffer.cpp:420-439
    if (!loaded_) return;
    if (write) sync();
    if (prefetching_) {
        {
            std::unique_lock<std::mutex> lk(io_mu_);
            io_stop_ = true;
        }
        io_cv_.notify_all();
        if (io_thread_.joinable()) io_thread_.join();
    }
    if (!write) {
        c10::hip::getCurrentHIPStream(device_.index()).synchronize();
        for (auto& p : partition_table_) {
            p.present_ = false;
            p.buffer_idx_ = -1;
        }
    }
    free_staging();
    buffer_tensor_view_ = Tensor();
    staged_admits_.clear();
    loaded_ = false;
}

void PartitionBuffer::setBufferOrdering(std::vector<Tensor> buffer_states) {  // buffer.cpp:488-497
    buffer_states_.clear();
    for (auto& t : buffer_states) {
        Tensor h = t.to(torch::kCPU, torch::kInt64).contiguous();
        buffer_states_.emplace_back(h.data_ptr<int64_t>(), h.data_ptr<int64_t>() + h.numel());
    }
    if (buffer_states_.empty()) throw MariusRuntimeException("setBufferOrdering: empty ordering");
    for (auto& st : buffer_states_) {
        if ((int)st.size() > capacity_) throw MariusRuntimeException("setBufferOrdering: a buffer state exceeds the capacity");
        for (auto p : st)
            if (p < 0 || p >= num_partitions_) throw MariusRuntimeException("setBufferOrdering: partition id out of range");
    }
    buffer_state_ = buffer_states_[0];
    next_state_ = 1;
    if (loaded_) {
        unload(true);
        load();
    }
}

bool PartitionBuffer::hasSwap() { return next_state_ < buffer_states_.size(); }

std::vector<int> PartitionBuffer::getNextAdmit() {  // buffer.cpp:549-567: in the order of the next state
    std::vector<int> out;
    if (!hasSwap()) return out;
    for (auto p : buffer_states_[next_state_])
        if (std::find(buffer_state_.begin(), buffer_state_.end(), p) == buffer_state_.end()) out.push_back((int)p);
    return out;
}

std::vector<int> PartitionBuffer::getNextEvict() {  // buffer.cpp:569-585: in the order of the current state
    std::vector<int> out;
    if (!hasSwap()) return out;
    const auto& nxt = buffer_states_[next_state_];
    for (auto p : buffer_state_)
        if (std::find(nxt.begin(), nxt.end(), p) == nxt.end()) out.push_back((int)p);
    return out;
}

void PartitionBuffer::performNextSwap() {  // buffer.cpp:501-547 (+ evict :637-652, admit :654-686)
    if (buffer_state_.empty() || !hasSwap()) return;
    if (!loaded_) throw MariusRuntimeException("performNextSwap: buffer not loaded");
    if (dev_staging()) {
        perform_next_swap_staged();
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int> evict = getNextEvict(), admit = getNextAdmit();
    if (admit.size() > evict.size() || (int)evict.size() > lanes_) throw std::runtime_error("");
    std::vector<int64_t> slots;
    for (int e : evict) slots.push_back(partition_table_[e].buffer_idx_);
    hipStream_t s = (hipStream_t)swap_stream_;
    // the batches of the current state are enqueued on the caller's stream: the swap starts after them
    hipEvent_t ev;
    PB_HIPCHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    PB_HIPCHECK(hipEventRecord(ev, c10::hip::getCurrentHIPStream(device_.index()).stream()));
    PB_HIPCHECK(hipStreamWaitEvent(s, ev, 0));
    // The host runs ahead of the device (the fused step never synchronises): waiting here for the state's last batch is compute time,
    // not swap time.  Accounted separately so that swap_seconds_ is what the exchange itself costs.
    PB_HIPCHECK(hipEventSynchronize(ev));
    drain_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const auto t1 = std::chrono::steady_clock::now();
    PB_HIPCHECK(hipEventDestroy(ev));
    bool staged = false;
    if (prefetching_) {
        io_wait();  // previous write-back done (evict staging free) and the look-ahead reads complete
        staged = staged_admits_ == admit;
        if (staged) ++prefetch_hits_;
    }
    // PCIe is full duplex: the slot is drained (D2H, swap stream) and refilled (H2D, second stream) in chunks, chunk k of the
    // admission following chunk k of the eviction by an event, so both directions are busy for most of the swap
    hipStream_t s2 = (hipStream_t)swap_stream2_;
    const int64_t CH = 32ll << 20;
    std::vector<hipEvent_t> events;
    for (size_t i = 0; i < evict.size(); ++i) {
        const Partition& pe = partition_table_[evict[i]];
        Partition* pa = i < admit.size() ? &partition_table_[admit[i]] : nullptr;
        if (pa && !staged) file_->readPartition(admit_mem_[i], *pa);  // a partition evicted earlier was written before this point (FIFO / synchronous)
        const int64_t span = std::max<int64_t>(pe.total_size_, pa ? pa->total_size_ : 0);
        for (int64_t off = 0; off < span; off += CH) {
            const int64_t ne = std::min(CH, pe.total_size_ - off), na = pa ? std::min(CH, pa->total_size_ - off) : 0;
            if (ne > 0) PB_HIPCHECK(hipMemcpyAsync((char*)evict_mem_[i] + off, slot_ptr(slots[i]) + off, (size_t)ne, hipMemcpyDeviceToHost, s));
            if (na > 0) {
                hipEvent_t e;
                PB_HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                PB_HIPCHECK(hipEventRecord(e, s));
                PB_HIPCHECK(hipStreamWaitEvent(s2, e, 0));
                events.push_back(e);
                PB_HIPCHECK(hipMemcpyAsync(slot_ptr(slots[i]) + off, (char*)admit_mem_[i] + off, (size_t)na, hipMemcpyHostToDevice, s2));
            }
        }
        if (pa && pa->total_size_ < slot_bytes()) {  // short last partition: zero tail, after the eviction has drained the slot
            hipEvent_t e;
            PB_HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            PB_HIPCHECK(hipEventRecord(e, s));
            PB_HIPCHECK(hipStreamWaitEvent(s2, e, 0));
            events.push_back(e);
            PB_HIPCHECK(hipMemsetAsync(slot_ptr(slots[i]) + pa->total_size_, 0, (size_t)(slot_bytes() - pa->total_size_), s2));
        }
        if (pa) scan_slot(slots[i], s2);
    }
    PB_HIPCHECK(hipStreamSynchronize(s));
    PB_HIPCHECK(hipStreamSynchronize(s2));
    for (auto e : events) PB_HIPCHECK(hipEventDestroy(e));
    for (int e : evict) partition_table_[e].present_ = false;  // buffer_idx_ stays, as in the reference
    for (size_t i = 0; i < admit.size(); ++i) {
        partition_table_[admit[i]].present_ = true;
        partition_table_[admit[i]].buffer_idx_ = (int)slots[i];
    }
    buffer_state_ = buffer_states_[next_state_++];
    if (prefetching_) {
        std::vector<int> next = getNextAdmit();
        staged_admits_ = next;
        io_submit([this, evict, next] {
            for (size_t i = 0; i < evict.size(); ++i) file_->writePartition(evict_mem_[i], partition_table_[evict[i]]);
            for (size_t i = 0; i < next.size(); ++i) file_->readPartition(admit_mem_[i], partition_table_[next[i]]);
        });
    } else {
        for (size_t i = 0; i < evict.size(); ++i) file_->writePartition(evict_mem_[i], partition_table_[evict[i]]);
    }
    ++swaps_;
    swap_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
}

// The same exchange with the PCIe transfers moved off the swap point (see the header): at the swap the device copies the evicted partitions into
// dev_evict_ and the staged admissions out of dev_admit_ (two HBM-to-HBM copies, ~1 ms per GB), the next state's batches wait for exactly that, and
// the evicted data goes to the host, into its file, and the FOLLOWING admission from its file onto the device while those batches run.  The host
// never waits for the device here; it is throttled by io_wait() to at most one buffer state ahead of the IO thread.
void PartitionBuffer::perform_next_swap_staged() {
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<int> evict = getNextEvict(), admit = getNextAdmit();
    if (admit.size() > evict.size() || (int)evict.size() > lanes_) throw std::runtime_error("");
    std::vector<int64_t> slots;
    for (int e : evict) slots.push_back(partition_table_[e].buffer_idx_);
    hipStream_t s = (hipStream_t)swap_stream_, s2 = (hipStream_t)swap_stream2_;
    hipStream_t compute = c10::hip::getCurrentHIPStream(device_.index()).stream();
    io_wait();  // the previous swap's write-back and this swap's look-ahead (file -> pinned -> dev_admit_, enqueued on s2) are done / enqueued
    const bool staged = staged_admits_ == admit;
    if (staged) ++prefetch_hits_;
    if (!staged) {  // an ordering the look-ahead did not predict (setBufferOrdering in between): read and stage now
        PB_HIPCHECK(hipStreamSynchronize(s2));
        for (size_t i = 0; i < admit.size(); ++i) {
            file_->readPartition(admit_mem_[i], partition_table_[admit[i]]);
            PB_HIPCHECK(hipMemcpyAsync(dev_admit_[i], admit_mem_[i], (size_t)partition_table_[admit[i]].total_size_, hipMemcpyHostToDevice, s2));
        }
        PB_HIPCHECK(hipEventRecord((hipEvent_t)ev_admit_ready_, s2));
    }
    PB_HIPCHECK(hipEventRecord((hipEvent_t)ev_compute_, compute));
    PB_HIPCHECK(hipStreamWaitEvent(s, (hipEvent_t)ev_compute_, 0));        // the ending state's batches have updated the slab
    if (!admit.empty()) PB_HIPCHECK(hipStreamWaitEvent(s, (hipEvent_t)ev_admit_ready_, 0));
    for (size_t i = 0; i < evict.size(); ++i) {
        const Partition& pe = partition_table_[evict[i]];
        PB_HIPCHECK(hipMemcpyAsync(dev_evict_[i], slot_ptr(slots[i]), (size_t)pe.total_size_, hipMemcpyDeviceToDevice, s));
        if (i < admit.size()) {
            const Partition& pa = partition_table_[admit[i]];
            PB_HIPCHECK(hipMemcpyAsync(slot_ptr(slots[i]), dev_admit_[i], (size_t)pa.total_size_, hipMemcpyDeviceToDevice, s));
            if (pa.total_size_ < slot_bytes()) PB_HIPCHECK(hipMemsetAsync(slot_ptr(slots[i]) + pa.total_size_, 0, (size_t)(slot_bytes() - pa.total_size_), s));
            scan_slot(slots[i], s);  // ahead of ev_swapped_: the next state's first batch reads a bound that covers the admitted rows
        }
    }
    PB_HIPCHECK(hipEventRecord((hipEvent_t)ev_swapped_, s));
    PB_HIPCHECK(hipStreamWaitEvent(compute, (hipEvent_t)ev_swapped_, 0));  // the next state's first batch
    for (size_t i = 0; i < evict.size(); ++i)
        PB_HIPCHECK(hipMemcpyAsync(evict_mem_[i], dev_evict_[i], (size_t)partition_table_[evict[i]].total_size_, hipMemcpyDeviceToHost, s));
    PB_HIPCHECK(hipEventRecord((hipEvent_t)ev_evict_host_, s));
    for (int e : evict) partition_table_[e].present_ = false;  // buffer_idx_ stays, as in the reference
    for (size_t i = 0; i < admit.size(); ++i) {
        partition_table_[admit[i]].present_ = true;
        partition_table_[admit[i]].buffer_idx_ = (int)slots[i];
    }
    buffer_state_ = buffer_states_[next_state_++];
    std::vector<int> next = getNextAdmit();
    staged_admits_ = next;
    io_submit([this, evict, next] {
        hipStream_t s2 = (hipStream_t)swap_stream2_;
        // the look-ahead read does not have to queue behind the write-back (different partitions, different pinned buffers): side by side,
        // unless a partition leaving now is the one coming back next (then the file must have it first)
        bool overlap = !next.empty();
        for (int n : next)
            if (std::find(evict.begin(), evict.end(), n) != evict.end()) overlap = false;
        std::string rerr;
        std::thread reader;
        auto read_next = [this, &next, s2] {
            PB_HIPCHECK(hipStreamSynchronize(s2));  // the previous look-ahead's copies have left admit_mem_ (dev_admit_ was consumed before ev_swapped_)
            for (size_t i = 0; i < next.size(); ++i) file_->readPartition(admit_mem_[i], partition_table_[next[i]]);
        };
        if (overlap)
            reader = std::thread([this, &rerr, &read_next] {
                try {
                    (void)hipSetDevice(device_.index());
                    read_next();
                } catch (const std::exception& e) {
                    rerr = e.what();
                }
            });
        try {
            PB_HIPCHECK(hipEventSynchronize((hipEvent_t)ev_evict_host_));  // (also: ev_swapped_ has passed, i.e. dev_admit_ was consumed)
            for (size_t i = 0; i < evict.size(); ++i) file_->writePartition(evict_mem_[i], partition_table_[evict[i]]);
        } catch (...) {
            if (reader.joinable()) reader.join();
            throw;
        }
        if (reader.joinable()) reader.join();
        if (!rerr.empty()) throw MariusRuntimeException(rerr);
        if (next.empty()) return;
        if (!overlap) read_next();
        PB_HIPCHECK(hipEventSynchronize((hipEvent_t)ev_evict_host_));
        for (size_t i = 0; i < next.size(); ++i)
            PB_HIPCHECK(hipMemcpyAsync(dev_admit_[i], admit_mem_[i], (size_t)partition_table_[next[i]].total_size_, hipMemcpyHostToDevice, s2));
        PB_HIPCHECK(hipEventRecord((hipEvent_t)ev_admit_ready_, s2));
    });
    ++swaps_;
    swap_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
}

Tensor PartitionBuffer::getGlobalToLocalMap(bool get_current) {  // buffer.cpp:587-635
    Tensor map = torch::full({total_embeddings_}, -1, torch::kInt64);
    int64_t* m = map.data_ptr<int64_t>();
    auto fill = [&](const Partition& p, int64_t slot) {
        for (int64_t r = 0; r < p.partition_size_; ++r) m[p.idx_offset_ + r] = slot * partition_size_ + r;
    };
    if (get_current) {
        for (auto id : buffer_state_) fill(partition_table_[id], partition_table_[id].buffer_idx_);
        return map;
    }
    if (!hasSwap()) throw MariusRuntimeException("getGlobalToLocalMap(false): no further buffer state");
    std::vector<int> evict = getNextEvict(), admit = getNextAdmit();
    for (auto id : buffer_states_[next_state_])
        if (partition_table_[id].buffer_idx_ != -1) fill(partition_table_[id], partition_table_[id].buffer_idx_);
    for (size_t i = 0; i < evict.size() && i < admit.size(); ++i) fill(partition_table_[admit[i]], partition_table_[evict[i]].buffer_idx_);
    return map;
}

Tensor PartitionBuffer::getGlobalToLocalMapDevice() {
    auto opt = torch::TensorOptions().dtype(torch::kInt64).device(device_);
    Tensor map = torch::full({total_embeddings_}, -1, opt);
    for (auto id : buffer_state_) {
        const Partition& p = partition_table_[id];
        const int64_t b = (int64_t)p.buffer_idx_ * partition_size_;
        if (p.partition_size_ > 0) map.narrow(0, p.idx_offset_, p.partition_size_).copy_(torch::arange(b, b + p.partition_size_, opt));
    }
    return map;
}

Tensor PartitionBuffer::indexRead(Tensor indices) {  // buffer.cpp:434-448
    if (indices.sizes().size() != 1) throw std::runtime_error("");
    require_device(buffer_tensor_view_, "PartitionBuffer::indexRead");
    require_device(indices, "PartitionBuffer::indexRead");
    Tensor out = torch::empty({indices.size(0), (int64_t)embedding_size_}, buffer_tensor_view_.options());
    mcheck(marius_gather_rows(buffer_tensor_view_.data_ptr<float>(), buffer_tensor_view_.stride(0), indices.data_ptr<int64_t>(), indices.size(0),
                              (int32_t)embedding_size_, out.data_ptr<float>(), out.stride(0), cur_stream()));
    return out;
}

void PartitionBuffer::indexAdd(Tensor indices, Tensor values) {  // buffer.cpp:453-475 (ids unique)
    if (!values.defined() || indices.sizes().size() != 1 || indices.size(0) != values.size(0) || values.dim() != 2 || values.size(1) != embedding_size_)
        throw std::runtime_error("");
    require_device(buffer_tensor_view_, "PartitionBuffer::indexAdd");
    require_device(values, "PartitionBuffer::indexAdd");
    mcheck(marius_scatter_add_rows(buffer_tensor_view_.data_ptr<float>(), buffer_tensor_view_.stride(0), indices.data_ptr<int64_t>(), indices.size(0),
                                   (int32_t)embedding_size_, values.data_ptr<float>(), values.stride(0), cur_stream()));
}

Tensor PartitionBuffer::getRandomIds(int64_t size) {  // buffer.cpp:450: over the rows the current state holds
    int64_t n = 0;
    for (auto id : buffer_state_) n += partition_table_[id].partition_size_;
    return torch::randint(n, {size}, torch::kInt64).to(device_);
}

void PartitionBuffer::io_loop() {
    (void)hipSetDevice(device_.index());  // the look-ahead enqueues copies from this thread
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(io_mu_);
            io_cv_.wait(lk, [this] { return io_stop_ || !io_jobs_.empty(); });
            if (io_jobs_.empty()) return;  // stop requested and nothing queued
            job = std::move(io_jobs_.front());
            io_jobs_.pop_front();
            io_busy_ = true;
        }
        try {
            job();
        } catch (const std::exception& e) {
            std::unique_lock<std::mutex> lk(io_mu_);
            io_error_ = e.what();
        }
        {
            std::unique_lock<std::mutex> lk(io_mu_);
            io_busy_ = false;
        }
        io_cv_.notify_all();
    }
}

void PartitionBuffer::io_submit(std::function<void()> job) {
    {
        std::unique_lock<std::mutex> lk(io_mu_);
        io_jobs_.push_back(std::move(job));
    }
    io_cv_.notify_all();
}

void PartitionBuffer::io_wait() {
    std::unique_lock<std::mutex> lk(io_mu_);
    io_cv_.wait(lk, [this] { return io_jobs_.empty() && !io_busy_; });
    if (!io_error_.empty()) {
        std::string e = io_error_;
        io_error_.clear();
        throw MariusRuntimeException("PartitionBuffer IO thread: " + e);
    }
}

// ------------------------------------------------------------------------------------------------ storage wrapper
PartitionBufferStorage::PartitionBufferStorage(std::string filename, int64_t dim0_size, int64_t dim1_size, shared_ptr<PartitionBufferOptions> options,
                                               torch::Device device) {
    filename_ = std::move(filename);
    dim0_size_ = dim0_size;
    dim1_size_ = dim1_size;
    options_ = options;
    dtype_ = torch::kFloat32;
    device_ = device;
    const int64_t partition_size = (int64_t)std::ceil((double)dim0_size_ / options_->num_partitions);  // storage.cpp:75
    buffer_ = std::make_unique<PartitionBuffer>(options_->buffer_capacity, options_->num_partitions, options_->fine_to_coarse_ratio, partition_size,
                                                (int)dim1_size_, dim0_size_, dtype_, filename_, options_->prefetching, device_);
}
void PartitionBufferStorage::load() {
    if (loaded_) return;
    buffer_->load();
    data_ = buffer_->buffer_tensor_view_;
    loaded_ = true;
}
void PartitionBufferStorage::write() {
    if (loaded_) buffer_->sync();
}
void PartitionBufferStorage::unload(bool perform_write) {
    if (!loaded_) return;
    buffer_->unload(perform_write);
    data_ = Tensor();
    loaded_ = false;
}
void PartitionBufferStorage::performNextSwap() {
    buffer_->performNextSwap();
    data_ = buffer_->buffer_tensor_view_;
}
void PartitionBufferStorage::setBufferOrdering(std::vector<Tensor> buffer_states) {
    buffer_->setBufferOrdering(std::move(buffer_states));
    if (loaded_) data_ = buffer_->buffer_tensor_view_;
}
Tensor PartitionBufferStorage::range(int64_t, int64_t) { throw std::runtime_error(""); }     // storage.cpp:178-181: unsupported
void PartitionBufferStorage::indexPut(Tensor, Tensor) { throw std::runtime_error(""); }        // storage.cpp:183-186
void PartitionBufferStorage::rangePut(int64_t offset, Tensor values) {                         // storage.cpp:112-128: straight to the file
    if (loaded_) throw MariusRuntimeException("PartitionBufferStorage::rangePut: only before load()");
    Tensor host = values.to(torch::kCPU, dtype_).contiguous();
    int fd = open(filename_.c_str(), O_RDWR);
    if (fd == -1) throw std::runtime_error("");
    try {
        full_io(fd, (char*)host.data_ptr(), (int64_t)host.nbytes(), offset * dim1_size_ * 4, true);
    } catch (...) {
        close(fd);
        throw;
    }
    close(fd);
}

// ------------------------------------------------------------------------------------------------ orderings
namespace {
using States = std::vector<std::vector<int>>;

std::vector<int64_t> draw_perm(const shared_ptr<MariusGenerator>& g, int64_t n) {
    Tensor t = g->randperm(n);
    return std::vector<int64_t>(t.data_ptr<int64_t>(), t.data_ptr<int64_t>() + n);
}
template <class T>
std::vector<T> permuted(const std::vector<T>& v, const std::vector<int64_t>& perm) {
    std::vector<T> out(v.size());
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[perm[i]];
    return out;
}

// ordering.cpp:78-129: keep `capacity` partitions resident; cycle every on-disk partition through the last slot (each meets all
// residents), then retire up to capacity-1 residents in favour of on-disk partitions, which thereby are finished; repeat.
States beta_states(int num_partitions, int capacity, const shared_ptr<MariusGenerator>& g) {
    States states;
    std::vector<int64_t> all = draw_perm(g, num_partitions);
    std::vector<int> in_buffer(all.begin(), all.begin() + capacity);
    std::vector<int> on_disk;
    for (int p = 0; p < num_partitions; ++p)
        if (std::find(in_buffer.begin(), in_buffer.end(), p) == in_buffer.end()) on_disk.push_back(p);  // ascending, like the sorted unique
    states.push_back(in_buffer);
    while (!on_disk.empty()) {
        in_buffer = permuted(in_buffer, draw_perm(g, (int64_t)in_buffer.size()));
        on_disk = permuted(on_disk, draw_perm(g, (int64_t)on_disk.size()));
        for (size_t i = 0; i < on_disk.size(); ++i) {
            std::swap(on_disk[i], in_buffer.back());
            states.push_back(in_buffer);
        }
        on_disk = permuted(on_disk, draw_perm(g, (int64_t)on_disk.size()));
        size_t replaced = 0;
        for (int i = 0; i < capacity - 1 && (size_t)i < on_disk.size(); ++i) {
            in_buffer[i] = on_disk[i];
            states.push_back(in_buffer);
            ++replaced;
        }
        on_disk.erase(on_disk.begin(), on_disk.begin() + replaced);
    }
    return states;
}
}  // namespace

std::tuple<std::vector<Tensor>, std::vector<Tensor>> getEdgeBucketOrdering(EdgeBucketOrdering ordering, int num_partitions, int buffer_capacity,
                                                                           int fine_to_coarse_ratio, int num_cache_partitions,
                                                                           bool randomly_assign_edge_buckets, shared_ptr<MariusGenerator> generator) {
    // ordering.cpp:12-34: OLD_BETA / NEW_BETA are the two-level scheme with ratio 1, no cache, greedy / random assignment
    int ratio = fine_to_coarse_ratio, cache = num_cache_partitions;
    bool random_assign = randomly_assign_edge_buckets;
    switch (ordering) {
        case EdgeBucketOrdering::OLD_BETA: ratio = 1; cache = 0; random_assign = false; break;
        case EdgeBucketOrdering::NEW_BETA: ratio = 1; cache = 0; random_assign = true; break;
        case EdgeBucketOrdering::COMET: break;
        default: throw MariusRuntimeException("edge bucket ordering not implemented");  // ALL_BETA / CUSTOM: "Not implemented" upstream too
    }
    if (ratio < 1 || num_partitions % ratio || buffer_capacity % ratio) throw MariusRuntimeException("fine_to_coarse_ratio must divide partitions and capacity");
    const int cp = num_partitions / ratio - cache, cc = buffer_capacity / ratio - cache;
    if (cc < 2 && cp > cc) throw MariusRuntimeException("buffer capacity too small: two (coarse) partitions must fit");
    // ordering.cpp:245-297
    States coarse = beta_states(cp, cc, generator);
    const int cached_fine = cache * ratio;
    std::vector<int> fine_map(num_partitions);
    for (int i = 0; i < cached_fine; ++i) fine_map[i] = i;
    std::vector<int64_t> rest = draw_perm(generator, num_partitions - cached_fine);
    for (int i = 0; i < num_partitions - cached_fine; ++i) fine_map[cached_fine + i] = (int)rest[i] + cached_fine;
    States states;
    for (auto& cs : coarse) {
        std::vector<int> st = cs;
        for (auto& x : st) x += cache;
        for (int j = 0; j < cache; ++j) st.push_back(j);
        std::vector<int> fine;
        for (int c : st)
            for (int k = 0; k < ratio; ++k) fine.push_back(fine_map[c * ratio + k]);
        states.push_back(fine);
    }
    // edge buckets
    const int P = num_partitions;
    std::vector<std::vector<std::pair<int, int>>> buckets(states.size());
    if (!random_assign) {  // ordering.cpp:131-152: first state that holds both partitions
        std::vector<char> seen((size_t)P * P, 0);
        for (size_t i = 0; i < states.size(); ++i)
            for (int s : states[i])
                for (int t : states[i])
                    if (!seen[(size_t)s * P + t]) {
                        seen[(size_t)s * P + t] = 1;
                        buckets[i].emplace_back(s, t);
                    }
    } else {  // ordering.cpp:154-243: uniform choice among the states that hold both; the reference draws with libc rand_r per OpenMP
              // thread (not reproducible), here: first element of a generator randperm over the candidates
        std::vector<std::vector<char>> holds(states.size(), std::vector<char>(P, 0));
        for (size_t i = 0; i < states.size(); ++i)
            for (int s : states[i]) holds[i][s] = 1;
        for (int s = 0; s < P; ++s)
            for (int t = 0; t < P; ++t) {
                std::vector<int> options;
                for (size_t i = 0; i < states.size(); ++i)
                    if (holds[i][s] && holds[i][t]) options.push_back((int)i);
                if (options.empty()) throw MariusRuntimeException("ordering: a partition pair is never co-resident");
                const int pick = options[(size_t)draw_perm(generator, (int64_t)options.size())[0]];
                buckets[pick].emplace_back(s, t);
            }
    }
    std::vector<Tensor> ts, tb;
    for (auto& st : states) ts.push_back(torch::tensor(std::vector<int64_t>(st.begin(), st.end()), torch::kInt64));
    for (auto& bs : buckets) {
        Tensor t = torch::zeros({(int64_t)bs.size(), 2}, torch::kInt64);
        auto a = t.accessor<int64_t, 2>();
        for (size_t i = 0; i < bs.size(); ++i) {
            a[i][0] = bs[i].first;
            a[i][1] = bs[i].second;
        }
        tb.push_back(t);
    }
    return std::make_tuple(ts, tb);
}

}  // namespace marius_amd
