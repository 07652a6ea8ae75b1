// C++ host layer on libtorch (PyTorch-ROCm): the reference's operator API for the link-prediction hot path, with every
// device operation routed through the C-ABI of libmarius_hip.so (include/marius_hip.h).  Class / method / field names follow
// the reference (paths relative to /root/reference/src/cpp) so that code written against Marius reads the same here.
//
//   include/common/exception.h            -> MariusRuntimeException, TensorSizeMismatchException, UndefinedTensorException
//   include/storage/storage.h:35-86       -> Storage, InMemory (device-resident node table / edge list)
//   include/data/samplers/negative.h      -> NegativeSampler, CorruptNodeNegativeSampler
//   include/data/batch.h                  -> Batch
//   include/nn/decoders/edge/*.h          -> RelationOperator, Comparator, EdgeDecoder, DistMult, ComplEx, TransE, decoder methods
//   include/nn/loss.h                     -> LossFunction, SoftmaxCrossEntropy
//   include/nn/optim.h                    -> Optimizer, AdagradOptimizer, SGDOptimizer
//   include/nn/model.h                    -> Model (forward_lp, train_batch, evaluate_batch, step)
//   include/data/dataloader.h             -> DataLoader (initializeBatches, getBatch, loadGPUParameters, updateEmbeddings)
//   include/pipeline/trainer.h            -> SynchronousTrainer ; include/pipeline/evaluator.h -> SynchronousEvaluator
//   include/reporting/reporting.h         -> LinkPredictionReporter (ranks, MRR, Hits@k)
// There is no CPU fallback: tensors handed to these classes must live on the MI355X.
#pragma once
#include <torch/torch.h>
#include <typeinfo>

#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "marius_hip.h"

namespace marius_amd {

using std::shared_ptr;
using torch::Tensor;

// ------------------------------------------------------------------------------------------------ exceptions (exception.h:12-42)
struct MariusRuntimeException : public std::runtime_error {
    explicit MariusRuntimeException(const std::string& msg) : std::runtime_error(msg) {}
};
struct UndefinedTensorException : public MariusRuntimeException {
    UndefinedTensorException() : MariusRuntimeException("Tensor undefined") {}
};
struct TensorSizeMismatchException : public MariusRuntimeException {
    TensorSizeMismatchException(const Tensor& t, const std::string& msg) : MariusRuntimeException(describe(t, msg)) {}
    static std::string describe(const Tensor& t, const std::string& msg);
};

void mcheck(int rc);                 // non-zero C-ABI status -> MariusRuntimeException(marius_hip_last_error())
marius_stream_t cur_stream();        // current HIP stream of the current device
void require_device(const Tensor& t, const char* what);

// ------------------------------------------------------------------------------------------------ options (configuration/options.h)
enum class LossReduction { MEAN, SUM };
enum class EdgeDecoderMethod { ONLY_POS, POS_AND_NEG, CORRUPT_NODE, CORRUPT_REL };
enum class LocalFilterMode { ALL, DEG };
enum class DecoderType { DISTMULT, TRANSE, COMPLEX };

// ------------------------------------------------------------------------------------------------ generator (ATen CPU MT19937 stream)
// torch::manual_seed(seed) + the global CPU generator of the reference (marius.cpp:47), as an explicit object whose state can
// live on the host (randperm) or on the device (negative sampling).
// Process-wide auxiliary streams, one set per device, created together on first use and never destroyed: AUX_LOADER (the DataLoader's batch
// preparation), AUX_FILL (the generator's run-ahead MT19937 pools), AUX_RELATIONS (the relation-table step underneath the node-table update).
// Every DataLoader / MariusGenerator / Model of the process shares them.  Round 6 finding: with a stream per OBJECT (torch's pool, or
// hipStreamCreate in the generator) the second trainer of a process — bench.py builds one for the arithmetic check's pretraining before the timed
// one — got streams number 4 and 5; the runtime maps streams onto 4 hardware queues, so the timed trainer's loader stream shared a queue with the
// main stream, its kernels no longer ran underneath the matrix launches, and every step was 60-70 us slower (0.64 vs 0.57 ms) for the rest of
// the process.
enum AuxStream { AUX_LOADER = 0, AUX_FILL = 1, AUX_RELATIONS = 2, AUX_COUNT = 3 };
void* aux_stream(int device_index, int which);  // hipStream_t
unsigned order_event_flags();  // hipEventCreateWithFlags flags of stream-ordering events (no timing, no system-scope fence)

class MariusGenerator {
   public:
    explicit MariusGenerator(uint64_t seed);
    ~MariusGenerator();
    Tensor randperm(int64_t n);                       // host int64, consumes the stream like torch::randperm on CPU
    Tensor raw_words(int64_t n, torch::Device dev);   // n raw 32-bit draws on `dev` (int32 tensor), advances the stream
    void to_device(torch::Device dev);
    void to_host();
    Tensor state_host_;  // [625] int32
    Tensor state_dev_;   // defined while the state lives on the device
    // Run-ahead pools: the single-workgroup MT19937 kernel produces the next `pool_requests_` requests' worth of raw words on a
    // side stream while the main stream trains (same stream of numbers, produced early; two pools alternate).  to_host() rewinds
    // to exactly the consumed position, so host draws (randperm at the epoch boundary) stay in sequence.
    bool prefetch_ = true;
    int pool_requests_ = 16;
    // the pool fills run on a stream of the caller's instead of one of their own (not owned; set before the first draw).  The sharded trainer hands
    // over its exchange stream: a stream of the generator's own would be the process's fifth (sharded_trainer.cpp)
    void use_fill_stream(void* hip_stream);
    void release_fill_stream();  // forget a borrowed fill stream (its owner is going away): the next pool fill creates the generator's own again

   private:
    struct Pool {
        Tensor buf, state_before;
        int64_t size = 0, used = 0;
        bool filled = false, waited = false;
        void* ready = nullptr;  // hipEvent_t: fill finished (side stream)
        void* done = nullptr;   // hipEvent_t: last consumer enqueued (main stream)
        bool has_done = false;
    };
    Pool pools_[2];
    int cur_ = 0;
    void* side_stream_ = nullptr;  // hipStream_t: the process-wide fill stream (aux_stream) unless a caller lends its own (use_fill_stream)
    bool side_stream_owned_ = true;  // false: borrowed through use_fill_stream
    bool side_ordered_ = false;  // the fill stream has been ordered behind the state upload
    void fill_pool(int i, torch::Device dev);
    void drop_pools();
};

// ------------------------------------------------------------------------------------------------ storage (storage.h:35-86)
class Storage {
   public:
    virtual ~Storage() = default;
    int64_t dim0_size_ = 0;
    int64_t dim1_size_ = 0;
    torch::Dtype dtype_ = torch::kFloat32;
    Tensor data_;
    torch::Device device_ = torch::kCPU;
    std::string filename_;
    bool loaded_ = false;
    std::vector<int64_t> edge_bucket_sizes_;  // storage.h:50: sizes of the (src partition, dst partition) buckets of a bucket-sorted edge list

    void readPartitionSizes(const std::string& filename);  // storage.cpp:203-214: one size per line (edges/train_partition_offsets.txt)
    virtual Tensor indexRead(Tensor indices) = 0;
    virtual void indexAdd(Tensor indices, Tensor values) = 0;
    virtual Tensor range(int64_t offset, int64_t n) = 0;
    virtual void indexPut(Tensor indices, Tensor values) = 0;
    virtual void rangePut(int64_t offset, Tensor values) = 0;
    virtual void load() = 0;
    virtual void write() = 0;
    virtual void unload(bool perform_write) = 0;
};

// DEVICE_MEMORY backend (storage.cpp:515-760) on the MI355X: raw row-major binary file <-> HBM tensor.
class InMemory : public Storage {
   public:
    InMemory(std::string filename, int64_t dim0_size, int64_t dim1_size, torch::Dtype dtype, torch::Device device);
    InMemory(Tensor data);  // tensor constructor (storage.cpp:538-545)
    Tensor indexRead(Tensor indices) override;
    shared_ptr<MariusGenerator> generator_;  // stream shuffle() draws from (the run's global generator); null: torch's device generator
    void setGenerator(shared_ptr<MariusGenerator> g) { generator_ = g; }
    Tensor indexReadCounted(Tensor indices, Tensor count_dev);  // capacity-sized ids, valid length on the device (fused step: no host sync)
    void indexAdd(Tensor indices, Tensor values) override;
    Tensor range(int64_t offset, int64_t n) override;
    void indexPut(Tensor indices, Tensor values) override;
    void rangePut(int64_t offset, Tensor values) override;
    void load() override;
    void write() override;
    void unload(bool perform_write) override;
    // storage.cpp:709-790: permute / sort the rows (within each edge bucket when edge_bucket_sizes_ is set).  torch device ops: this is
    // data preparation, not the training path; the reference draws the permutation from the device generator, so no bit parity is defined
    void shuffle();
    void sort(bool src);
};

// ------------------------------------------------------------------------------------------------ graph stub + samplers
struct MariusGraph {  // the fields the LP sampler reads (graph.h)
    int64_t num_nodes_in_memory_ = 0;
    // every known edge (train + validation + test [+ filter edges]) sorted by source / by destination, int64: the "true edge" lookup
    // of filtered evaluation (graph.cpp:233-236; assembled by GraphModelStorage::sortAllEdges, graph_storage.cpp:745-777)
    Tensor all_src_sorted_edges_, all_dst_sorted_edges_;
    void sortAllEdges(Tensor all_edges);
};
// compute_filter_corruption, global (filtered-evaluation) branch, negative.cpp:212-293: (edge_id, node) pairs of every true edge that
// shares the uncorrupted endpoint (and the relation) with batch edge edge_id; apply_score_filter sets those scores to -1e9
Tensor compute_filter_corruption_global(shared_ptr<MariusGraph> graph, Tensor edges, bool inverse);

class NegativeSampler {
   public:
    virtual ~NegativeSampler() = default;
    virtual std::tuple<Tensor, Tensor> getNegatives(shared_ptr<MariusGraph> graph, Tensor edges = Tensor(), bool inverse = false) = 0;
};

class CorruptNodeNegativeSampler : public NegativeSampler {
   public:
    int num_chunks_;
    int num_negatives_;
    float degree_fraction_;
    bool filtered_;
    LocalFilterMode local_filter_mode_;
    shared_ptr<MariusGenerator> generator_;

    CorruptNodeNegativeSampler(int num_chunks, int num_negatives, float degree_fraction, bool filtered = false,
                               LocalFilterMode local_filter_mode = LocalFilterMode::DEG, shared_ptr<MariusGenerator> generator = nullptr);
    // (ids [C,N] int64, filter [F,2] int64) — negative.cpp:328-366.  compact_filter=false keeps the uncompacted filter (rows of -1)
    std::tuple<Tensor, Tensor> getNegatives(shared_ptr<MariusGraph> graph, Tensor edges = Tensor(), bool inverse = false) override;
    bool compact_filter_ = true;
};

// ------------------------------------------------------------------------------------------------ batch (batch.h:32-89)
class Batch {
   public:
    explicit Batch(bool train) : train_(train) {}
    bool train_;
    int batch_id_ = -1;
    int64_t start_idx_ = 0;
    int64_t batch_size_ = 0;
    Tensor edges_;                      // [B, 3|2] int64 (global ids before map, batch-local after)
    Tensor global_edges_;               // [B, 3|2] int64 global ids, kept beside the batch-local copy (table-direct fused step)
    Tensor table_;                      // defined: the fused step reads node rows straight from this table by global id (no gathered copy)
    Tensor unique_node_indices_;        // [U] ascending
    Tensor node_embeddings_;            // [U, d]
    Tensor node_embeddings_state_;      // [U, d]
    Tensor node_embeddings_grad_;       // [U, d]  (node_embeddings_.grad() in the reference)
    Tensor node_gradients_;             // dw
    Tensor node_state_update_;          // ds
    Tensor src_neg_indices_, dst_neg_indices_;                   // [C, N] global ids
    Tensor src_neg_indices_mapping_, dst_neg_indices_mapping_;   // [C, N] batch-local
    Tensor src_neg_filter_, dst_neg_filter_;                     // [F, 2]
    // device-side products of the unique map that the fused update consumes
    Tensor occ_perm_, occ_inverse_, occ_seg_offsets_, num_unique_dev_;
    Tensor occ_plan_, rel_plan_;  // marius_segment_plan of the node / relation unique maps (prepared with the batch, off the critical path)
    // sorted-unique map of the batch's relation ids (column 1), prepared by the loader so the relation-gradient reduction needs no
    // sort on the compute stream: uniq [B] (zero tail), inverse [B], perm [B] int32, seg [B+1] int32
    Tensor rel_uniq_, rel_inverse_, rel_perm_, rel_seg_, rel_count_;
    Tensor table_to_update_; // the table the fused update of this batch writes (set by the trainer; lets the model recognise a table it tracks)
    Tensor row_bound_;       // optional device float[1] >= every |x| of node_embeddings_ (whoever gathered the rows scanned them: ShardedTrainer)
    void* ready_ = nullptr;  // hipEvent_t recorded on the loader stream when the batch was prepared ahead (DataLoader owns it)

    void accumulateGradients(float learning_rate);  // batch.cpp:62-79
    void clear();                                   // batch.cpp:105-...
};

// ------------------------------------------------------------------------------------------------ decoder pieces
// The reference's plug-in points (relation_operators.h:11-15, comparators.h:13-17): a virtual call operator on tensors.  The built-in
// subclasses additionally report a kind(): when every component of a model reports one, training runs the fused HIP path keyed on it;
// a user subclass (kind() == -1, the default) makes the model take the generic path through these virtual calls and libtorch
// autograd, exactly as the reference does (nn/model.cpp:290-333).  The built-ins themselves run on the device kernels for plain
// tensors and as differentiable libtorch ops when an argument requires grad (so they compose with user code on the generic path).
class RelationOperator {
   public:
    virtual ~RelationOperator() = default;
    virtual int kind() const { return -1; }  // MARIUS_OP_* for the built-ins
    virtual Tensor operator()(const Tensor& embs, const Tensor& rels) = 0;  // undefined rels => identity (relation_operators.cpp)
};
#define MARIUS_RELOP(NAME, KIND)                                                \
    struct NAME : RelationOperator {                                            \
        int kind() const override { return KIND; }                              \
        Tensor operator()(const Tensor& embs, const Tensor& rels) override;     \
    }
MARIUS_RELOP(HadamardOperator, MARIUS_OP_HADAMARD);
MARIUS_RELOP(ComplexHadamardOperator, MARIUS_OP_COMPLEX_HADAMARD);
MARIUS_RELOP(TranslationOperator, MARIUS_OP_TRANSLATION);
MARIUS_RELOP(NoOp, MARIUS_OP_NOOP);
#undef MARIUS_RELOP

class Comparator {
   public:
    virtual ~Comparator() = default;
    virtual int kind() const { return -1; }  // MARIUS_CMP_* for the built-ins
    virtual Tensor operator()(Tensor src, Tensor dst) = 0;  // dst [B,d] -> [B]; dst [C,N,d] -> [C*ceil(B/C), N]  (comparators.cpp)
};
#define MARIUS_CMP(NAME, KIND)                                   \
    struct NAME : Comparator {                                   \
        int kind() const override { return KIND; }               \
        Tensor operator()(Tensor src, Tensor dst) override;      \
    }
MARIUS_CMP(DotCompare, MARIUS_CMP_DOT);
MARIUS_CMP(L2Compare, MARIUS_CMP_L2);
MARIUS_CMP(CosineCompare, MARIUS_CMP_COSINE);
#undef MARIUS_CMP
Tensor pad_and_reshape(Tensor input, int num_chunks);  // comparators.cpp:7-20

// options.h:12, nn/decoders/decoder.h:12-17
enum class LearningTask { NODE_CLASSIFICATION, LINK_PREDICTION, ENCODE };
class Decoder {
   public:
    LearningTask learning_task_ = LearningTask::LINK_PREDICTION;
    virtual ~Decoder() {}
};

class EdgeDecoder : public Decoder {  // edge_decoder.h:13-31
   public:
    shared_ptr<Comparator> comparator_;
    shared_ptr<RelationOperator> relation_operator_;
    Tensor relations_;
    Tensor inverse_relations_;
    int num_relations_ = 0;
    int embedding_size_ = 0;
    torch::TensorOptions tensor_options_;
    EdgeDecoderMethod decoder_method_ = EdgeDecoderMethod::CORRUPT_NODE;
    bool use_inverse_relations_ = true;

    Tensor apply_relation(Tensor nodes, Tensor relations);
    Tensor compute_scores(Tensor src, Tensor dst);
    Tensor select_relations(Tensor indices, bool inverse = false);
};
// distmult.h:10-16, complex.h, transe.h: torch::nn::Cloneable modules whose reset() (re)creates the relation tables and registers them as
// `relation_embeddings` / `inverse_relation_embeddings` with requires_grad(true) (distmult.cpp:21-27).  The fused path updates them in place
// through the C-ABI (raw pointers: outside autograd); Model::train_batch leaves the hand-derived gradients in their .grad(), so an optimizer a
// user builds over named_parameters() steps them as in the reference; the generic (plug-in) path differentiates detached aliases.
class DistMult : public EdgeDecoder, public torch::nn::Cloneable<DistMult> {
   public:
    DistMult(int num_relations, int embedding_dim, torch::TensorOptions tensor_options = torch::TensorOptions(), bool use_inverse_relations = true,
             EdgeDecoderMethod decoder_method = EdgeDecoderMethod::CORRUPT_NODE);
    void reset() override;
};
class ComplEx : public EdgeDecoder, public torch::nn::Cloneable<ComplEx> {
   public:
    ComplEx(int num_relations, int embedding_dim, torch::TensorOptions tensor_options = torch::TensorOptions(), bool use_inverse_relations = true,
            EdgeDecoderMethod decoder_method = EdgeDecoderMethod::CORRUPT_NODE);
    void reset() override;
};
class TransE : public EdgeDecoder, public torch::nn::Cloneable<TransE> {
   public:
    TransE(int num_relations, int embedding_dim, torch::TensorOptions tensor_options = torch::TensorOptions(), bool use_inverse_relations = true,
           EdgeDecoderMethod decoder_method = EdgeDecoderMethod::CORRUPT_NODE);
    void reset() override;
};
shared_ptr<EdgeDecoder> get_edge_decoder(DecoderType type, EdgeDecoderMethod method, int num_relations, int dim, torch::TensorOptions opts,
                                         bool use_inverse_relations);

// One workspace per batch shape holding everything the fused forward / loss / backward produce (marius_lp_layout).
struct LpContext {
    marius_lp_desc desc{};
    marius_lp_layout layout{};
    Tensor workspace;
    std::vector<Tensor> keep;  // tensors whose pointers sit in desc
    bool has_loss = false;
    Tensor absmax;             // optional device float (marius_lp_desc.absmax): bound on the node rows the step reads; set per step by Model::bind_ranges, used only with MARIUS_LP_TRAIN_ONLY
    Tensor absmax_rel;         // optional device float[1] (marius_lp_desc.absmax_rel); undefined: absmax is float[2] and holds both
    int free_cus = 0;          // marius_lp_desc.free_cus of the training forward (set by a trainer that runs other streams beside the matrix launches)
    Tensor view(size_t off, std::vector<int64_t> shape, std::vector<int64_t> strides = {}) const;
};

// decoder_methods.h:11-21
std::tuple<Tensor, Tensor> only_pos_forward(shared_ptr<EdgeDecoder> decoder, Tensor edges, Tensor node_embeddings);
std::tuple<Tensor, Tensor, Tensor, Tensor> node_corrupt_forward(shared_ptr<EdgeDecoder> decoder, Tensor positive_edges, Tensor node_embeddings,
                                                                Tensor dst_negs, Tensor src_negs, LpContext* ctx = nullptr,
                                                                Tensor dst_filter = Tensor(), Tensor src_filter = Tensor(),
                                                                LossReduction reduction = LossReduction::SUM, int loss_kind = MARIUS_LOSS_SOFTMAX_CE,
                                                                float margin = 0.f, int lp_flags = 0);

// ------------------------------------------------------------------------------------------------ loss / optimizers / reporter
class LossFunction {  // loss.h:21-31; every subclass of loss.h:33-107 evaluates on the device (marius_loss_scores / marius_lp_loss)
   public:
    virtual ~LossFunction() = default;
    LossReduction reduction_type_ = LossReduction::SUM;
    virtual int kind() const { return -1; }  // MARIUS_LOSS_* for the built-ins; -1: user-defined (generic autograd path)
    virtual float margin() const { return 0.f; }
    virtual bool scores_only() const { return false; }  // SoftmaxCE / Ranking throw for classification input (loss.cpp:51-55, 72-74)
    virtual const char* name() const { return "LossFunction"; }
    // (pos [B'], neg [B', N], scores = true) -> scalar loss.  scores = false (classification logits + labels) is outside the link-prediction path.
    virtual Tensor operator()(Tensor y_pred, Tensor targets, bool scores);
};
#define MARIUS_LOSS_CLASS(NAME, KIND, SCORES_ONLY)                                                        \
    class NAME : public LossFunction {                                                                    \
       public:                                                                                            \
        explicit NAME(LossReduction reduction = LossReduction::SUM) { reduction_type_ = reduction; }      \
        int kind() const override { return KIND; }                                                        \
        bool scores_only() const override { return SCORES_ONLY; }                                         \
        const char* name() const override { return #NAME; }                                               \
    }
MARIUS_LOSS_CLASS(SoftmaxCrossEntropy, MARIUS_LOSS_SOFTMAX_CE, true);        // loss.cpp:50-67
MARIUS_LOSS_CLASS(CrossEntropyLoss, MARIUS_LOSS_CROSS_ENTROPY, false);      // loss.cpp:89-103
MARIUS_LOSS_CLASS(BCEAfterSigmoidLoss, MARIUS_LOSS_BCE_AFTER_SIGMOID, false);  // loss.cpp:105-123
MARIUS_LOSS_CLASS(BCEWithLogitsLoss, MARIUS_LOSS_BCE_WITH_LOGITS, false);   // loss.cpp:125-143
MARIUS_LOSS_CLASS(MSELoss, MARIUS_LOSS_MSE, false);                         // loss.cpp:145-163
MARIUS_LOSS_CLASS(SoftPlusLoss, MARIUS_LOSS_SOFTPLUS, false);               // loss.cpp:165-187
#undef MARIUS_LOSS_CLASS
class RankingLoss : public LossFunction {  // loss.cpp:69-87, loss.h:43-55
   public:
    float margin_;
    explicit RankingLoss(LossReduction reduction = LossReduction::SUM, float margin = 0.1f) : margin_(margin) { reduction_type_ = reduction; }
    int kind() const override { return MARIUS_LOSS_RANKING; }
    float margin() const override { return margin_; }
    bool scores_only() const override { return true; }
    const char* name() const override { return "RankingLoss"; }
};
shared_ptr<LossFunction> getLossFunction(const std::string& type, LossReduction reduction, float margin = 0.1f);  // loss.cpp:189-209

class Optimizer {
   public:
    virtual ~Optimizer() = default;
    // optim.cpp:7-40: "num_steps" + one nested archive per parameter key holding its state tensors
    virtual void save(torch::serialize::OutputArchive& archive, const std::vector<std::string>& keys);
    virtual void load(torch::serialize::InputArchive& archive, const std::vector<std::string>& keys);
    virtual int64_t num_steps() const { return 0; }
    virtual void set_num_steps(int64_t) {}
    // (name, tensors) of the per-parameter state in the reference's naming: Adagrad {"sum"}, Adam {"exp_avg", "exp_avg_sq"[, "max_exp_avg_sq"]}
    virtual std::vector<std::pair<std::string, std::vector<Tensor>*>> state_slots() { return {}; }
    float learning_rate_ = 0.1f;
    std::vector<std::pair<Tensor, Tensor>> params_;  // (param, grad)
    std::vector<Tensor> state_;
    virtual void step() = 0;
    void clear_grad();
};
class AdagradOptimizer : public Optimizer {  // optim.cpp:82-145
   public:
    float eps_ = 1e-10f, weight_decay_ = 0.f;
    AdagradOptimizer(std::vector<std::pair<Tensor, Tensor>> params, float lr, float eps = 1e-10f, float init_value = 0.f);
    void step() override;
    int64_t num_steps_ = 0;
    int64_t num_steps() const override { return num_steps_; }
    void set_num_steps(int64_t n) override { num_steps_ = n; }
    std::vector<std::pair<std::string, std::vector<Tensor>*>> state_slots() override { return {{"sum", &state_}}; }
};
class AdamOptimizer : public Optimizer {  // optim.cpp:147-232
   public:
    float eps_ = 1e-8f, beta_1_ = 0.9f, beta_2_ = 0.999f, weight_decay_ = 0.f;
    bool amsgrad_ = false;
    int64_t num_steps_ = 0;
    std::vector<Tensor> exp_avg_sq_, max_exp_avg_sq_;  // state_ holds exp_avg
    AdamOptimizer(std::vector<std::pair<Tensor, Tensor>> params, float lr, float eps = 1e-8f, float beta_1 = 0.9f, float beta_2 = 0.999f,
                  float weight_decay = 0.f, bool amsgrad = false);
    void step() override;
    int64_t num_steps() const override { return num_steps_; }
    void set_num_steps(int64_t n) override { num_steps_ = n; }
    std::vector<std::pair<std::string, std::vector<Tensor>*>> state_slots() override {
        std::vector<std::pair<std::string, std::vector<Tensor>*>> v = {{"exp_avg", &state_}, {"exp_avg_sq", &exp_avg_sq_}};
        if (amsgrad_) v.push_back({"max_exp_avg_sq", &max_exp_avg_sq_});
        return v;
    }
};
class SGDOptimizer : public Optimizer {  // optim.cpp:59-79
   public:
    SGDOptimizer(std::vector<std::pair<Tensor, Tensor>> params, float lr);
    void step() override;
};

class Reporter {  // reporting.h:58-76 (metrics are fixed to the link-prediction set of model.cpp:28-38 in this build)
   public:
    virtual ~Reporter() = default;
};
class LinkPredictionReporter : public Reporter {  // reporting.cpp:11-57
   public:
    std::vector<Tensor> ranks_;
    Tensor computeRanks(Tensor pos_scores, Tensor neg_scores);
    void addResult(Tensor pos_scores, Tensor neg_scores);
    void clear() { ranks_.clear(); }
    // (MRR, MeanRank, Hits@1, @3, @5, @10, @50, @100)
    std::vector<double> report();
};

// ------------------------------------------------------------------------------------------------ model (model.h:16-65)
// What initModelFromConfig needs of the reference's ModelConfig (configuration/config.h) for an embedding-only link-prediction model.
struct ModelConfig {
    std::string decoder = "DISTMULT";            // model.decoder.type
    int embedding_dim = 50;                      // model.encoder ... output_dim of the embedding layer
    bool inverse_edges = true;                   // model.decoder.options.inverse_edges
    std::string decoder_method = "CORRUPT_NODE"; // model.decoder.options.edge_decoder_method
    std::string loss = "SOFTMAX_CE";             // model.loss.type
    std::string loss_reduction = "SUM";          // model.loss.options.reduction
    float margin = 0.1f;                         // model.loss.options.margin
    std::string dense_optimizer = "ADAGRAD";     // model.dense_optimizer.type
    float dense_lr = 0.1f, eps = 1e-10f, beta_1 = 0.9f, beta_2 = 0.999f, weight_decay = 0.f;
    bool amsgrad = false;
    float sparse_lr = 0.1f;                      // model.sparse_optimizer.options.learning_rate
    bool encoder_bias = false;                   // model.encoder.layers[0][0].bias (LayerConfig, marius_config.py:190-199)
    std::string encoder_activation = "NONE";     // model.encoder.layers[0][0].activation
    Tensor encoder_bias_init;                    // [embedding_dim] from bias_init (undefined: ZEROS)
};

// encoder.h / layer.h: the general encoder of an embedding-only model = ONE EmbeddingLayer: a column view of the batch's rows
// (EmbeddingLayer::forward, embedding.cpp:17) followed by Layer::post_hook (layer.cpp:9-16): `+ bias` when LayerConfig::bias, then the
// activation (activation.cpp:7-21: NONE / RELU / SIGMOID).  Default-constructed: a pass-through (no parameters), which is what every
// BASELINE configuration uses and the only form the table-direct fused step accepts (its kernels read node rows in place); with a bias or an
// activation the model trains through the API-granular step (gathered [U, d] rows -> marius_layer_post_hook -> decoder -> its backward ->
// sparse Adagrad on the raw rows), the bias as a dense parameter "embedding:0_0_bias" of the dense optimizer (model.cpp:175-183).
enum class ActivationFunction { NONE = 0, RELU = 1, SIGMOID = 2 };
class GeneralEncoder : public torch::nn::Module {
   public:
    GeneralEncoder() = default;
    // bias_init: the initial bias [output_dim] (initialize_tensor(config_->bias_init, ...), layer.cpp:18-22; undefined = ZEROS, the default)
    GeneralEncoder(int output_dim, bool bias, ActivationFunction activation, torch::Device device, Tensor bias_init = Tensor());
    bool has_post_hook() const { return bias_.defined() || activation_ != ActivationFunction::NONE; }
    // encoder.cpp:195-257 for the single embedding layer.  Differentiable libtorch ops when autograd is recording through `embeddings` (user
    // plug-ins: Model::train_batch_generic), the HIP kernel otherwise.
    Tensor forward(Tensor embeddings);
    // gradient w.r.t. the embeddings from the gradient w.r.t. forward's output (and that output); fills bias_grad_ (deterministic column sums)
    Tensor backward(Tensor grad_encoded, Tensor encoded);
    Tensor bias_, bias_grad_;
    ActivationFunction activation_ = ActivationFunction::NONE;
    int output_dim_ = 0;

   private:
    Tensor ws_;
};

class Model : public torch::nn::Module {
   public:
    shared_ptr<GeneralEncoder> encoder_;  // nullptr / pass-through, or one embedding layer with a post-hook (bias, activation)
    bool has_post_hook() const { return encoder_ && encoder_->has_post_hook(); }
    void set_encoder(shared_ptr<GeneralEncoder> encoder);  // (re)binds the encoder; call setup_optimizer* afterwards so that its bias is stepped
    LearningTask learning_task_ = LearningTask::LINK_PREDICTION;
    // Multi-GPU (model.h:33): the reference keeps one replica per device inside one process.  This build runs one process per GPU
    // (torch.distributed over RCCL), so a process only ever holds its own replica; broadcast() records the device list and all_reduce()
    // sums the dense (relation) gradients over the ranks of the registered process group.
    std::vector<torch::Device> devices_;
    std::string process_group_;  // c10d group name (torch.distributed: group.group_name); empty = single process
    void broadcast(std::vector<torch::Device> devices);  // model.cpp:136-147
    void all_reduce();                                    // model.cpp:149-159
    void set_process_group(const std::string& name) { process_group_ = name; }
    // true when every component is a built-in (kind() >= 0) and forward_lp is not overridden: the fused HIP training path applies
    bool fused_ok() const;
    virtual bool custom_forward() const { return typeid(*this) != typeid(Model); }
    shared_ptr<EdgeDecoder> decoder_;
    shared_ptr<LossFunction> loss_function_;
    shared_ptr<LinkPredictionReporter> reporter_;
    std::vector<shared_ptr<Optimizer>> optimizers_;
    float sparse_lr_ = 0.1f;
    // table-direct fused step: endpoint occurrences whose node occurs once take their Adagrad step inside the edge backward (marius_lp_desc.upd_*);
    // false (or MARIUS_FUSE_ENDPOINT_UPDATE=0 in the environment): every row goes through the segment update — same bits either way
    bool fuse_endpoint_update_ = true;
    int64_t last_fused_below_ = 0;  // diagnostic: occurrences [0, this) of the last table-direct step were eligible for the in-backward update (0: not fused)
    torch::Device device_ = torch::kCPU;
    Tensor relations_grad_, inverse_relations_grad_;
    LpContext ctx_;
    Tensor loss_;  // [4] device floats: total, rhs, lhs, -
    // scratch for the gradient reductions
    Tensor carry_, rel_carry_, rel_ws_, rel_uniq_, rel_inverse_, rel_perm_, rel_seg_, rel_count_, rel_ids_;
    void* side_stream_ = nullptr;  // relation-table update runs here, underneath the node-table update (backward_into_tables)
    void* ev_fork_ = nullptr;
    void* ev_join_ = nullptr;
    void* ev_grads_ = nullptr;  // caller-owned hipEvent_t (the data loader's gate): recorded on the training stream when a step's gradients exist, i.e.
                                // after the edge backward and before the updates; set around one backward_into_tables call, never owned here
    virtual ~Model();

    Model(shared_ptr<EdgeDecoder> decoder, shared_ptr<LossFunction> loss, shared_ptr<LinkPredictionReporter> reporter, torch::Device device);
    // the reference's constructor (model.h:33, model.cpp:18-58): the device is the decoder's, a null reporter becomes the link-prediction one
    Model(shared_ptr<GeneralEncoder> encoder, shared_ptr<Decoder> decoder, shared_ptr<LossFunction> loss, shared_ptr<Reporter> reporter = nullptr,
          std::vector<shared_ptr<Optimizer>> optimizers = {});
    virtual std::tuple<Tensor, Tensor, Tensor, Tensor> forward_lp(shared_ptr<Batch> batch, bool train);  // model.cpp:252-288
    // the same forward for callers that only train (nobody reads the negative scores): lets the library take the flash-style path
    // (MARIUS_LP_TRAIN_ONLY, include/marius_hip.h); neg / inv_neg of the returned tuple are then undefined
    std::tuple<Tensor, Tensor, Tensor, Tensor> forward_lp_train(shared_ptr<Batch> batch);
    void train_batch(shared_ptr<Batch> batch, bool call_step = true);                            // model.cpp:290-333
    void evaluate_batch(shared_ptr<Batch> batch);                                                // model.cpp:335-359
    void clear_grad();
    void step();
    void setup_optimizers(float dense_lr);
    void setup_optimizers(shared_ptr<ModelConfig> model_config);  // model.h:58, model.cpp:161-250: the dense optimizer of the configuration
    // model.cpp:82-134: model.pt (encoder + decoder parameters) and model_state.pt (optimizer state) as torch::serialize archives with
    // the reference's key structure, so a model directory written here loads in the reference and vice versa
    void save(const std::string& directory);
    void load(const std::string& directory, bool train);
    // dense optimizer by name (ModelConfig::dense_optimizer, model.cpp:381-440): "ADAGRAD", "ADAM" or "SGD"
    void setup_optimizer(const std::string& type, float lr, float eps, float beta_1, float beta_2, float weight_decay, bool amsgrad);
    // fused tail used by the trainer for DEVICE_MEMORY tables: backward products -> table/state update in one call
    // table_direct: the decoder kernels index `table` by the batch's global ids (global_edges_, *_neg_indices_) instead of a gathered
    // [U, d] copy by batch-local ids: the occurrence order, and with it every gradient and update, is unchanged
    void backward_into_tables(shared_ptr<Batch> batch, Tensor table, Tensor state, bool table_direct = false);
    // sharded node table (sharded_trainer.h): forward + loss + backward, then the per-unique-row gradient sums into grad_out [>= U, d] for
    // the owners of the rows.  local_relation_step: touched-rows Adagrad step on this replica's relation tables (replicas are averaged every
    // gpu_sync_interval steps); otherwise the dense gradients are left in relations_grad_ / inverse_relations_grad_ for an all-reduce + step()
    void backward_to_unique_grads(shared_ptr<Batch> batch, Tensor grad_out, bool local_relation_step, Tensor out_rows = Tensor());
    std::vector<Tensor> dense_state();  // relation tables + their optimizer state (what gpu_model_average averages, pipeline_gpu.cpp:52-80)
    // Magnitude bounds on the device (marius_lp_desc.absmax / absmax_rel): with them the flash path packs fp16 operand halves (22 significand
    // bits per operand) instead of bf16 ones (16).  Three sources for the bound on the node rows a step reads, one per way the rows reach it:
    //   * a device-resident table read in place: track_ranges(table) scans it once (one pass, no temporary) and the fused update keeps the bound
    //     current (marius_segment_adagrad_scatter_tracked); the scan remembers WHICH table it saw (pointer, rows, ATen version) and is redone
    //     when the trainer meets another one or an ATen op wrote it in between;
    //   * a partition-buffer slab read in place: the buffer's own running bound (PartitionBuffer::absmax — initial fill, every admitted
    //     partition at its swap, every tracked update);
    //   * a gathered copy [U, d] (sharded table, Model::train_batch, gathered fused step): the copy itself is scanned — exact, local, no
    //     collective — into Batch::row_bound_ (the sharded trainer does it on its exchange stream) or the model's own running row_bound_.
    // The relation bound (range_state_[1]) belongs to the model: the touched-rows update tracks it, everything else that writes the relation
    // tables (dense optimizer steps, load, replica averaging) calls touch_relations() and the bound is rescanned (two small tables) before
    // the next training forward.
    Tensor range_state_;  // device float[2]: [0] bound on the tracked table, [1] bound on the relation tables
    Tensor row_bound_;    // device float[1]: running bound on the rows of gathered batches (monotone: every batch is max'ed in)
    bool ranges_valid_ = false, rel_ranges_valid_ = false;
    struct Scanned {
        const void* ptr = nullptr;
        int64_t rows = 0;
        uint32_t version = 0;
        bool is(const Tensor& t) const { return t.defined() && t.data_ptr() == ptr && t.size(0) == rows && t._version() == version; }
        void set(const Tensor& t) { ptr = t.data_ptr(); rows = t.size(0); version = t._version(); }
    };
    Scanned tracked_table_, tracked_rel_[2];
    Tensor external_node_bound_;  // set by the trainer for a partition-buffer slab (the buffer keeps it current)
    float* node_track_ = nullptr; // the bound the node-table update of this step must keep current (set by bind_ranges; nullptr: none)
    static bool flash_f16_enabled();  // MARIUS_FLASH_F16=0 (read once): bf16 operand halves everywhere, no bounds kept
    void track_ranges(Tensor table);
    bool tracks(const Tensor& table) const { return ranges_valid_ && tracked_table_.is(table); }
    void drop_ranges();
    void touch_relations() { rel_ranges_valid_ = false; }
    // The node-table counterpart: anything that writes a TRACKED table through a raw pointer — a ctypes / C-ABI scatter, a user
    // kernel — without going through InMemory::indexAdd (which bumps the ATen version the scan remembers) calls this; the table is rescanned
    // before the next training forward instead of being packed against a stale bound (fp16 halves would saturate silently).
    void touch_table() { drop_ranges(); }
    void ensure_relation_ranges();
    float* relation_bound() { return rel_ranges_valid_ ? range_state_.data_ptr<float>() + 1 : nullptr; }
    // chooses the bounds of one training forward (ctx_.absmax / absmax_rel); direct: the batch reads batch->table_ in place
    void bind_ranges(shared_ptr<Batch> batch, bool direct);

    void publish_grads();  // relations_.grad() / inverse_relations_.grad() = the gradients of the last backward (model.cpp:324)

   private:
    void train_batch_generic(shared_ptr<Batch> batch, bool call_step);  // the reference's autograd formulation, for user plug-ins
};
// model.cpp:361-440 for the embedding-only link-prediction models of this build
shared_ptr<Model> initModelFromConfig(const ModelConfig& config, std::vector<torch::Device> devices, int num_relations, bool train);
shared_ptr<Model> initModelFromConfig(shared_ptr<ModelConfig> model_config, std::vector<torch::Device> devices, int num_relations, bool train);  // model.h:65

// ------------------------------------------------------------------------------------------------ dataloader (dataloader.h)
class PartitionBufferStorage;  // partition_buffer.h
class DataLoader {
   public:
    shared_ptr<InMemory> edges_;             // [E, 3|2] int32/int64 on device
    shared_ptr<Storage> node_embeddings_;    // [num_nodes, d]: InMemory (DEVICE_MEMORY) or PartitionBufferStorage (out-of-core)
    shared_ptr<Storage> node_embeddings_state_;
    shared_ptr<CorruptNodeNegativeSampler> negative_sampler_;
    shared_ptr<MariusGraph> graph_;
    shared_ptr<MariusGenerator> generator_;
    int64_t batch_size_;
    bool train_;
    int64_t num_edges_ = 0;
    int64_t batches_left_ = 0, batch_id_ = 0, total_batches_ = 0;
    Tensor active_perm_;  // device int64 permutation of the epoch
    // unique-map scratch (capacity-sized)
    Tensor all_ids_, uniq_, inverse_, perm_, seg_, count_, sort_ws_, sort_ws_rel_;  // (two sort workspaces: the fused map launch works on the node ids and the relation ids at once)
    int key_bits_ = 63;
    // run-ahead: getBatch() hands out a batch prepared on the loader stream while the previous step was computing, then starts the
    // next one.  Preparation (edge slice, negatives, map_tensors, relation-id sort) never reads the tables and consumes the generator
    // in the same order, so results are identical to the serial loop; only small launch-bound kernels leave the compute stream.
    bool run_ahead_ = false;
    int num_relations_ = 0;   // > 0: also prepare the relation-id map (Batch::rel_*)
    void* loader_stream_ = nullptr;
    void* ev_pool_[4] = {nullptr, nullptr, nullptr, nullptr};
    void* ev_main_[4] = {nullptr, nullptr, nullptr, nullptr};
    int ev_next_ = 0, ev_main_next_ = 0;
    shared_ptr<Batch> next_;
    bool next_exact_ = false;
    int64_t prepared_left_ = 0;
    // Batch preparation ahead of the step runs on its own host thread (the reference's batch_loader_threads, pipeline.cpp:19-49):
    // ~35 of a step's ~60 launches are the sampler / unique-map kernels of the NEXT batch, and issuing them from the training
    // thread made the step launch-bound once the contraction kernels left the FP32 matrix pipe.
    struct LoaderWorker;
    LoaderWorker* worker_ = nullptr;
    bool pending_ = false;  // a request has been posted and not yet taken
    // Memory of a batch is allocated on the loader stream and read by the training stream.  The preparing thread may only reuse it for a
    // batch whose loader-stream work waits for that training step, so a batch is released HERE, between taking the next prepared batch
    // and posting the following request — never while a preparation is in flight on the other thread.
    shared_ptr<Batch> held_;
    // Early gate (MARIUS_LOADER_GATE=0 disables it): the preparation of batch t + 1 may start as soon as the GRADIENTS of step t - 1 exist
    // (gate_event_, recorded by the model) instead of when that step has finished: its two hundred microseconds of small kernels then run
    // underneath the HBM-bound tail of step t - 1 (segmented sums, Adagrad) and the row packing of step t, and are mostly gone when the
    // persistent matrix launches of step t need every CU.  Preparation never reads the tables; what it must not do is reuse memory the
    // running updates still read, so a batch is kept one step longer (held_prev_).
    void* gate_event_ = nullptr;  // owned here (created on first use, destroyed with the loader)
    bool gate_valid_ = false;
    void* gate_event();
    shared_ptr<Batch> held_prev_;
    // The NEXT epoch's permutation, drawn by a host thread while this epoch trains (dataloader.cpp:176-182 puts a serial randperm of all
    // edges between two epochs: 0.12 s at 10 M edges, seconds at Freebase86m's 338 M — as long as the epoch itself on this device).  The words
    // an epoch's sampling draws from the generator are known in advance (two requests per batch, a fixed count each), so the thread advances a
    // COPY of the state by that many words and draws the permutation from there.  At the epoch boundary the copy's starting point is compared
    // with the generator's actual state: equal -> the permutation and the state after it are adopted (bit-identical to drawing it now);
    // different (anything else consumed the generator) -> the permutation is drawn now as before.  MARIUS_SHUFFLE_AHEAD=0 disables it.
    struct ShuffleAhead;
    ShuffleAhead* ahead_ = nullptr;
    bool full_batches_only_ = false;  // the caller wraps to the next epoch when fewer than batch_size_ edges remain (ShardedTrainer)
    bool plan_ahead_ = false;         // prepare marius_segment_plan of the node / relation maps with every batch even without run_ahead_ (ShardedTrainer)
    int64_t shuffle_ahead_hits_ = 0, shuffle_ahead_misses_ = 0;
    // words other users of the same generator draw between two of this loader's epochs (e.g. the per-epoch evaluation passes: the sum of their
    // wordsPerEpoch(true)); the permutation drawn ahead starts that much further down the stream
    int64_t words_between_epochs_ = 0;
    int64_t wordsPerEpoch(bool with_permutation);  // generator words one pass over this loader consumes (non-partitioned)
    void start_shuffle_ahead();
    bool take_shuffle_ahead(Tensor& perm);
    Tensor last_num_unique_;  // device count of the batch getBatch returned last (count_ itself belongs to the preparing thread)
    void post_prepare(bool exact_unique);
    shared_ptr<Batch> take_prepared();
    void drain_worker();
    ~DataLoader();
    shared_ptr<Batch> prepareBatch(bool exact_unique);  // the body of getBatch, on the current stream

    DataLoader(shared_ptr<InMemory> edges, shared_ptr<Storage> node_embeddings, shared_ptr<Storage> node_embeddings_state,
               shared_ptr<CorruptNodeNegativeSampler> negative_sampler, shared_ptr<MariusGenerator> generator, int64_t batch_size, bool train);
    void initializeBatches(bool shuffle = true);    // dataloader.cpp:202-248 (+ setActiveEdges :120-183)
    bool hasNextBatch();                            // out-of-core: swaps to the next buffer state when the current one is exhausted
    // ---- out-of-core mode (both tables are PartitionBufferStorage; edges_ sorted by edge bucket, torch_partitioner.py:12-46)
    shared_ptr<PartitionBufferStorage> pb_embeddings_, pb_state_;
    std::vector<int64_t> edge_bucket_starts_;       // [p*p + 1] prefix sums of the bucket sizes
    std::vector<Tensor> buffer_states_, edge_buckets_per_buffer_;
    size_t buffer_cursor_ = 0;
    Tensor active_edges_;                           // [n, cols] int64, buffer-local node ids: the edges assigned to the current buffer state
    bool partitioned() const { return pb_embeddings_ != nullptr; }
    void setEdgeBucketSizes(std::vector<int64_t> sizes);  // Storage::edge_bucket_sizes_ (storage.h:50) of the train edges
    void loadStorage();                             // dataloader.cpp:566-600: new ordering (consumes the generator), load the first buffer state
    void nextEpoch(bool write = true);              // dataloader.cpp:108-118: write the buffer back (training), unload
    void setActiveEdges();                          // dataloader.cpp:120-175 for the current buffer state
    bool buckets_validated_ = false;
    const void* validated_ptr_ = nullptr;  // the edge list validate_edge_buckets() last passed: data pointer, rows, ATen version
    int64_t validated_rows_ = 0;
    uint32_t validated_version_ = 0;
    void validate_edge_buckets();                   // once per edge list: it is sorted by edge bucket and the bucket sizes describe it
    shared_ptr<Batch> getBatch(bool exact_unique = true);  // dataloader.cpp:360-471
    void loadGPUParameters(shared_ptr<Batch> batch);       // dataloader.cpp:529-548
    void updateEmbeddings(shared_ptr<Batch> batch, bool gpu = true);  // dataloader.cpp:550-564
    int64_t getNumEdges() const { return num_edges_; }
};

// ------------------------------------------------------------------------------------------------ trainer / evaluator
class SynchronousTrainer {  // trainer.cpp:94-161
   public:
    shared_ptr<DataLoader> dataloader_;
    shared_ptr<Model> model_;
    bool fused_update_ = true;  // DEVICE_MEMORY fast path: segmented sum + Adagrad + scatter in one C-ABI call
    double last_epoch_seconds_ = 0, last_edges_per_second_ = 0;
    SynchronousTrainer(shared_ptr<DataLoader> dataloader, shared_ptr<Model> model) : dataloader_(dataloader), model_(model) {}
    void train(int num_epochs = 1);
    // run `n` batches of the current epoch (starting a new epoch when the batches run out); no host synchronisation inside
    void train_steps(int64_t n);
    void train_one(bool fused);
};
// PipelineTrainer::train (trainer.cpp:35-74) on one device: the reference's five worker stages (pipeline.cpp, pipeline_gpu.cpp) collapse
// to "batches are admitted and prepared ahead of the training step".  Kept semantics:
//   * admission control (pipeline.cpp:24-45): at most `staleness_bound` batches are in flight (admitted, not yet applied);
//   * where the parameters of a batch are read: with device-resident storage the compute stage gathers them right before the step
//     (pipeline_gpu.cpp:52 loadGPUParameters) — no staleness, only the sampler runs ahead, results equal the synchronous trainer; with
//     host storage the LOADER stage gathers rows and optimizer state at admission (getBatch -> loadCPUParameters, dataloader.cpp:505-527)
//     and the update lands later: rows are up to staleness_bound - 1 updates old.  `stale_parameters` selects the second behaviour
//     (storage.embeddings.type: HOST_MEMORY in the YAML); this build admits deterministically, so the staleness is exactly
//     min(staleness_bound, batches left) - 1 and runs are reproducible (the reference's thread interleaving is not).
class PipelineTrainer {
   public:
    shared_ptr<DataLoader> dataloader_;
    shared_ptr<Model> model_;
    int staleness_bound_;
    bool stale_parameters_;
    double last_epoch_seconds_ = 0, last_edges_per_second_ = 0;
    PipelineTrainer(shared_ptr<DataLoader> dataloader, shared_ptr<Model> model, int staleness_bound = 16, bool stale_parameters = false)
        : dataloader_(dataloader), model_(model), staleness_bound_(staleness_bound < 1 ? 1 : staleness_bound), stale_parameters_(stale_parameters) {}
    void train(int num_epochs = 1);
};
class SynchronousEvaluator {  // evaluator.cpp:58-97
   public:
    shared_ptr<DataLoader> dataloader_;
    shared_ptr<Model> model_;
    SynchronousEvaluator(shared_ptr<DataLoader> dataloader, shared_ptr<Model> model) : dataloader_(dataloader), model_(model) {}
    std::vector<double> evaluate();
};

}  // namespace marius_amd
