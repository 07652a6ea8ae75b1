// Out-of-core node storage for the link-prediction path: the reference's PartitionBuffer API
// (include/storage/buffer.h:13-190, src/storage/buffer.cpp:324-713; PartitionBufferStorage include/storage/storage.h:89-146) re-designed
// for the MI355X:
//   * the buffer slab [capacity * partition_size, d] lives in HBM (the reference keeps it in host DRAM and ships every batch over
//     PCIe); a batch reads / updates it with the same gather / scatter kernels as a DEVICE_MEMORY table, ids being buffer-local rows;
//   * a swap moves whole partitions: evicted slots -> pinned host staging (hipMemcpyAsync D2H on a dedicated swap stream) -> pwrite,
//     admitted partitions pread -> pinned staging -> H2D into the freed slots; with `prefetching` one background IO thread writes the
//     evicted partitions and then reads the NEXT swap's admissions while the current buffer state trains (the reference's
//     LookaheadBlock + AsyncWriteBlock pair, buffer.cpp:122-322; a single FIFO thread gives the write-before-re-read order their
//     `evicting_` flag exists for), so a swap costs two PCIe copies, not file IO;
//   * 288 GB of HBM per GPU: `buffer_capacity` can usually be the whole table (cfg5, Twitter d=400: 2 x 66.6 GB) and then no swap
//     ever happens; capacity < num_partitions is for tables beyond that.
// Orderings (include/data/ordering.h, src/data/ordering.cpp:78-297): BETA / COMET buffer-state sequences + edge-bucket assignment.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#include "marius_host.h"

namespace marius_amd {

struct Partition {  // buffer.h:13-43 (the fields the device design keeps)
    int partition_id_ = 0;
    int64_t partition_size_ = 0;  // rows (the last partition may be short)
    int64_t idx_offset_ = 0;      // first global row id
    int64_t file_offset_ = 0;     // bytes
    int64_t total_size_ = 0;      // bytes
    int buffer_idx_ = -1;         // slot in the slab
    bool present_ = false;
};

class PartitionedFile {  // buffer.cpp:64-120
   public:
    std::string filename_;
    int fd_ = -1;
    explicit PartitionedFile(const std::string& filename);
    ~PartitionedFile();
    void readPartition(void* host_addr, const Partition& p);         // pread the whole partition
    void writePartition(const void* host_addr, const Partition& p);  // pwrite the whole partition
};

class PartitionBuffer {
   public:
    PartitionBuffer(int capacity, int num_partitions, int fine_to_coarse_ratio, int64_t partition_size, int embedding_size, int64_t total_embeddings,
                    torch::Dtype dtype, std::string filename, bool prefetching, torch::Device device);
    ~PartitionBuffer();

    void load();
    void unload(bool write);
    void sync();  // write every present partition back and mark it absent (buffer.cpp:688-699)
    std::vector<int> getNextAdmit();
    std::vector<int> getNextEvict();
    Tensor getRandomIds(int64_t size);
    Tensor indexRead(Tensor indices);               // buffer-local ids, device
    void indexAdd(Tensor indices, Tensor values);   // ids unique
    Tensor getGlobalToLocalMap(bool get_current);   // host int64 [total_embeddings], -1 = not in the buffer
    Tensor getGlobalToLocalMapDevice();             // the current map built on the device (a few slice fills instead of a host loop + upload)
    void setBufferOrdering(std::vector<Tensor> buffer_states);
    bool hasSwap();
    void performNextSwap();
    int64_t getNumInMemory() const { return (int64_t)capacity_ * partition_size_; }
    std::vector<int64_t> getBufferState() const { return buffer_state_; }

    Tensor buffer_tensor_view_;  // [capacity * partition_size, d] on the device
    // Running bound on |entries of the slab| (device float[1]; marius_lp_desc.absmax of the table-direct training step): the whole slab once
    // when enabled / loaded, every admitted partition right behind the copy that fills its slot (same stream, ahead of the event the next buffer
    // state's first batch waits for), and the fused update's own tracking.  Never lowered: evictions only make it looser.
    Tensor absmax_;
    void enable_absmax();
    std::vector<Partition> partition_table_;
    // counters for tests / reporting
    int64_t swaps_ = 0, prefetch_hits_ = 0;
    double swap_seconds_ = 0;
    double drain_seconds_ = 0;  // host time spent at swap points waiting for the device to finish the batches of the ending buffer state

   private:
    int capacity_, num_partitions_, fine_to_coarse_ratio_, embedding_size_;
    int64_t partition_size_, total_embeddings_;
    torch::Dtype dtype_;
    int dtype_size_;
    std::string filename_;
    bool prefetching_;
    torch::Device device_;
    bool loaded_ = false;
    std::unique_ptr<PartitionedFile> file_;
    std::vector<std::vector<int64_t>> buffer_states_;
    size_t next_state_ = 0;
    std::vector<int64_t> buffer_state_;

    // pinned staging: `lanes_` partitions each way
    int lanes_ = 1;
    std::vector<void*> admit_mem_, evict_mem_;
    void* swap_stream_ = nullptr;   // evictions (D2H), initial load
    void* swap_stream2_ = nullptr;  // admissions (H2D) during a swap
    std::vector<int> staged_admits_;  // partition ids currently (being) read into admit_mem_ by the IO thread
    // Device-side staging (prefetching mode; 288 GB of HBM make two more slot-sized buffers per lane a rounding error): the next admission is
    // already ON the device when its swap comes, and the eviction leaves its slot by a device-to-device copy — the PCIe transfers of a swap run
    // under the compute of the neighbouring buffer states instead of between them.
    std::vector<void*> dev_admit_, dev_evict_;
    void* ev_compute_ = nullptr;      // the ending state's last batch (compute stream)
    void* ev_swapped_ = nullptr;      // slots exchanged (swap stream): the next state's first batch waits for it
    void* ev_evict_host_ = nullptr;   // evicted partitions have reached the pinned buffers
    void* ev_admit_ready_ = nullptr;  // the staged admissions have reached dev_admit_
    bool dev_staging() const { return prefetching_ && !dev_admit_.empty(); }
    void perform_next_swap_staged();

    // one FIFO IO thread
    std::thread io_thread_;
    std::mutex io_mu_;
    std::condition_variable io_cv_;
    std::deque<std::function<void()>> io_jobs_;
    bool io_busy_ = false, io_stop_ = false;
    std::string io_error_;
    void io_loop();
    void io_submit(std::function<void()> job);
    void io_wait();

    int64_t slot_bytes() const { return partition_size_ * embedding_size_ * dtype_size_; }
    char* slot_ptr(int64_t slot) const;
    void stage_in(const Partition& p, int64_t slot, void* staging);  // staging (already filled) -> slot, zero tail
    void scan_slot(int64_t slot, void* stream);  // max |x| of one slot into absmax_ (no-op unless enabled)
    void alloc_staging();
    void free_staging();
};

enum class EdgeBucketOrdering { OLD_BETA, NEW_BETA, ALL_BETA, COMET, CUSTOM };

struct PartitionBufferOptions {  // configuration/options.h (PartitionBufferOptions) / marius_config.py
    int num_partitions = 16;
    int buffer_capacity = 8;
    bool prefetching = true;
    int fine_to_coarse_ratio = 1;
    int num_cache_partitions = 0;
    EdgeBucketOrdering edge_bucket_ordering = EdgeBucketOrdering::COMET;  // datatypes.py:161-169 defaults
    bool randomly_assign_edge_buckets = true;
};

class PartitionBufferStorage : public Storage {  // storage.h:89-146
   public:
    shared_ptr<PartitionBufferOptions> options_;
    std::unique_ptr<PartitionBuffer> buffer_;
    PartitionBufferStorage(std::string filename, int64_t dim0_size, int64_t dim1_size, shared_ptr<PartitionBufferOptions> options, torch::Device device);
    Tensor indexRead(Tensor indices) override { return buffer_->indexRead(indices); }
    void indexAdd(Tensor indices, Tensor values) override { buffer_->indexAdd(indices, values); }
    Tensor range(int64_t offset, int64_t n) override;
    void indexPut(Tensor indices, Tensor values) override;
    void rangePut(int64_t offset, Tensor values) override;  // straight to the file (storage.cpp:112-128): initialisation before load()
    void load() override;
    void write() override;
    void unload(bool perform_write) override;
    Tensor getRandomIds(int64_t size) { return buffer_->getRandomIds(size); }
    bool hasSwap() { return buffer_->hasSwap(); }
    void performNextSwap();
    Tensor getGlobalToLocalMap(bool get_current) { return buffer_->getGlobalToLocalMap(get_current); }
    Tensor getGlobalToLocalMapDevice() { return buffer_->getGlobalToLocalMapDevice(); }
    void sync() { buffer_->sync(); }
    void setBufferOrdering(std::vector<Tensor> buffer_states);
    std::vector<int> getNextAdmit() { return buffer_->getNextAdmit(); }
    std::vector<int> getNextEvict() { return buffer_->getNextEvict(); }
    int64_t getNumInMemory() { return buffer_->getNumInMemory(); }
};

// (buffer_states, edge_buckets_per_buffer): buffer_states[i] int64 [capacity]; edge_buckets_per_buffer[i] int64 [k_i, 2] (src, dst partition).
// Random draws come from `generator` (MT19937, same call sequence as the reference's torch::randperm calls on the default generator).
std::tuple<std::vector<Tensor>, std::vector<Tensor>> getEdgeBucketOrdering(EdgeBucketOrdering ordering, int num_partitions, int buffer_capacity,
                                                                           int fine_to_coarse_ratio, int num_cache_partitions,
                                                                           bool randomly_assign_edge_buckets, shared_ptr<MariusGenerator> generator);

}  // namespace marius_amd
