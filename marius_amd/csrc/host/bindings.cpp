// pybind11 module `_marius_host`: the C++ host classes (marius_host.h) for Python drivers / tests.
// Mirrors the parts of the reference's python_bindings (src/cpp/python_bindings) that sit on the link-prediction path.
#include <torch/extension.h>

#include "marius_host.h"
#include "partition_buffer.h"
#include "sharded_trainer.h"

namespace py = pybind11;
using namespace marius_amd;

PYBIND11_MODULE(_marius_host, m) {
    m.doc() = "MI355X-native host layer of the Marius link-prediction hot path (C++ on libtorch, kernels via libmarius_hip.so)";
    py::register_exception<MariusRuntimeException>(m, "MariusRuntimeException");
    // the structs of include/marius_hip.h are passed by pointer: a library built from another revision of the header would read past them
    if (marius_hip_abi_version() != MARIUS_HIP_ABI_VERSION || marius_hip_struct_bytes(0) != (int)sizeof(marius_lp_desc) ||
        marius_hip_struct_bytes(1) != (int)sizeof(marius_lp_layout) || marius_hip_struct_bytes(2) != (int)sizeof(marius_segment_update))
        throw std::runtime_error("_marius_host was built against C-ABI version " + std::to_string(MARIUS_HIP_ABI_VERSION) + " of libmarius_hip.so, the loaded library reports " +
                                 std::to_string(marius_hip_abi_version()) + ": rebuild both with `python -m marius_amd.build --host`");

    py::enum_<LossReduction>(m, "LossReduction").value("MEAN", LossReduction::MEAN).value("SUM", LossReduction::SUM);
    py::enum_<EdgeDecoderMethod>(m, "EdgeDecoderMethod")
        .value("ONLY_POS", EdgeDecoderMethod::ONLY_POS)
        .value("POS_AND_NEG", EdgeDecoderMethod::POS_AND_NEG)
        .value("CORRUPT_NODE", EdgeDecoderMethod::CORRUPT_NODE)
        .value("CORRUPT_REL", EdgeDecoderMethod::CORRUPT_REL);
    py::enum_<LocalFilterMode>(m, "LocalFilterMode").value("ALL", LocalFilterMode::ALL).value("DEG", LocalFilterMode::DEG);
    py::enum_<DecoderType>(m, "DecoderType").value("DISTMULT", DecoderType::DISTMULT).value("TRANSE", DecoderType::TRANSE).value("COMPLEX", DecoderType::COMPLEX);

    py::class_<MariusGenerator, std::shared_ptr<MariusGenerator>>(m, "MariusGenerator")
        .def(py::init<uint64_t>(), py::arg("seed"))
        .def("randperm", &MariusGenerator::randperm)
        .def("raw_words", [](MariusGenerator& g, int64_t n, torch::Device d) { return g.raw_words(n, d); })
        .def("to_host", &MariusGenerator::to_host)
        .def_readwrite("prefetch", &MariusGenerator::prefetch_)
        .def_readwrite("pool_requests", &MariusGenerator::pool_requests_)
        .def_readwrite("state_host", &MariusGenerator::state_host_);

    py::class_<Storage, std::shared_ptr<Storage>>(m, "Storage")
        .def_readwrite("data", &Storage::data_)
        .def_readwrite("dim0_size", &Storage::dim0_size_)
        .def_readwrite("dim1_size", &Storage::dim1_size_)
        .def_readwrite("filename", &Storage::filename_)
        .def_readwrite("edge_bucket_sizes", &Storage::edge_bucket_sizes_)
        .def("readPartitionSizes", &Storage::readPartitionSizes);
    py::class_<InMemory, Storage, std::shared_ptr<InMemory>>(m, "InMemory")
        .def(py::init<torch::Tensor>(), py::arg("data"))
        .def(py::init([](std::string filename, int64_t dim0, int64_t dim1, py::object dtype, torch::Device device) {
                 return std::make_shared<InMemory>(filename, dim0, dim1, torch::python::detail::py_object_to_dtype(dtype), device);
             }),
             py::arg("filename"), py::arg("dim0_size"), py::arg("dim1_size"), py::arg("dtype"), py::arg("device"))
        .def("indexRead", &InMemory::indexRead)
        .def("indexAdd", &InMemory::indexAdd)
        .def("range", &InMemory::range)
        .def("indexPut", &InMemory::indexPut)
        .def("rangePut", &InMemory::rangePut)
        .def("load", &InMemory::load)
        .def("write", &InMemory::write)
        .def("unload", &InMemory::unload, py::arg("perform_write") = false)
        .def("shuffle", &InMemory::shuffle)
        .def("setGenerator", &InMemory::setGenerator, py::arg("generator"))
        .def("sort", &InMemory::sort, py::arg("src"));

    py::enum_<EdgeBucketOrdering>(m, "EdgeBucketOrdering")
        .value("OLD_BETA", EdgeBucketOrdering::OLD_BETA)
        .value("NEW_BETA", EdgeBucketOrdering::NEW_BETA)
        .value("ALL_BETA", EdgeBucketOrdering::ALL_BETA)
        .value("COMET", EdgeBucketOrdering::COMET)
        .value("CUSTOM", EdgeBucketOrdering::CUSTOM);
    py::class_<PartitionBufferOptions, std::shared_ptr<PartitionBufferOptions>>(m, "PartitionBufferOptions")
        .def(py::init<>())
        .def_readwrite("num_partitions", &PartitionBufferOptions::num_partitions)
        .def_readwrite("buffer_capacity", &PartitionBufferOptions::buffer_capacity)
        .def_readwrite("prefetching", &PartitionBufferOptions::prefetching)
        .def_readwrite("fine_to_coarse_ratio", &PartitionBufferOptions::fine_to_coarse_ratio)
        .def_readwrite("num_cache_partitions", &PartitionBufferOptions::num_cache_partitions)
        .def_readwrite("edge_bucket_ordering", &PartitionBufferOptions::edge_bucket_ordering)
        .def_readwrite("randomly_assign_edge_buckets", &PartitionBufferOptions::randomly_assign_edge_buckets);
    py::class_<PartitionBuffer, std::shared_ptr<PartitionBuffer>>(m, "PartitionBuffer")  // buffer.h:128-190, float32 rows, slab on `device`
        .def(py::init([](int capacity, int num_partitions, int fine_to_coarse_ratio, int64_t partition_size, int embedding_size, int64_t total_embeddings,
                         std::string filename, bool prefetching, torch::Device device) {
                 return std::make_shared<PartitionBuffer>(capacity, num_partitions, fine_to_coarse_ratio, partition_size, embedding_size, total_embeddings,
                                                          torch::kFloat32, filename, prefetching, device);
             }),
             py::arg("capacity"), py::arg("num_partitions"), py::arg("fine_to_coarse_ratio"), py::arg("partition_size"), py::arg("embedding_size"),
             py::arg("total_embeddings"), py::arg("filename"), py::arg("prefetching"), py::arg("device"))
        .def("load", &PartitionBuffer::load, py::call_guard<py::gil_scoped_release>())
        .def("unload", &PartitionBuffer::unload, py::arg("write"), py::call_guard<py::gil_scoped_release>())
        .def("sync", &PartitionBuffer::sync, py::call_guard<py::gil_scoped_release>())
        .def("getNextAdmit", &PartitionBuffer::getNextAdmit)
        .def("getNextEvict", &PartitionBuffer::getNextEvict)
        .def("getRandomIds", &PartitionBuffer::getRandomIds)
        .def("indexRead", &PartitionBuffer::indexRead)
        .def("indexAdd", &PartitionBuffer::indexAdd)
        .def("getGlobalToLocalMap", &PartitionBuffer::getGlobalToLocalMap, py::arg("get_current") = true)
        .def("setBufferOrdering", &PartitionBuffer::setBufferOrdering, py::arg("buffer_states"))
        .def("hasSwap", &PartitionBuffer::hasSwap)
        .def("performNextSwap", &PartitionBuffer::performNextSwap, py::call_guard<py::gil_scoped_release>())
        .def("getNumInMemory", &PartitionBuffer::getNumInMemory)
        .def_readonly("swaps", &PartitionBuffer::swaps_)
        .def_readonly("prefetch_hits", &PartitionBuffer::prefetch_hits_)
        .def("enable_absmax", &PartitionBuffer::enable_absmax)
        .def_readonly("absmax", &PartitionBuffer::absmax_)  // running bound on |slab entries| (device float[1]); undefined until enabled
        .def_readonly("buffer_tensor_view", &PartitionBuffer::buffer_tensor_view_);
    py::class_<PartitionBufferStorage, Storage, std::shared_ptr<PartitionBufferStorage>>(m, "PartitionBufferStorage")
        .def(py::init<std::string, int64_t, int64_t, std::shared_ptr<PartitionBufferOptions>, torch::Device>(), py::arg("filename"), py::arg("dim0_size"),
             py::arg("dim1_size"), py::arg("options"), py::arg("device"))
        .def("indexRead", &PartitionBufferStorage::indexRead)
        .def("indexAdd", &PartitionBufferStorage::indexAdd)
        .def("range", &PartitionBufferStorage::range)
        .def("indexPut", &PartitionBufferStorage::indexPut)
        .def("rangePut", &PartitionBufferStorage::rangePut)
        .def("load", &PartitionBufferStorage::load, py::call_guard<py::gil_scoped_release>())
        .def("write", &PartitionBufferStorage::write, py::call_guard<py::gil_scoped_release>())
        .def("unload", &PartitionBufferStorage::unload, py::arg("perform_write") = false, py::call_guard<py::gil_scoped_release>())
        .def("getRandomIds", &PartitionBufferStorage::getRandomIds)
        .def("hasSwap", &PartitionBufferStorage::hasSwap)
        .def("performNextSwap", &PartitionBufferStorage::performNextSwap, py::call_guard<py::gil_scoped_release>())
        .def("getGlobalToLocalMap", &PartitionBufferStorage::getGlobalToLocalMap, py::arg("get_current") = true)
        .def("sync", &PartitionBufferStorage::sync, py::call_guard<py::gil_scoped_release>())
        .def("setBufferOrdering", &PartitionBufferStorage::setBufferOrdering, py::arg("buffer_states"))
        .def("getNextAdmit", &PartitionBufferStorage::getNextAdmit)
        .def("getNextEvict", &PartitionBufferStorage::getNextEvict)
        .def("getNumInMemory", &PartitionBufferStorage::getNumInMemory)
        .def_property_readonly("swaps", [](PartitionBufferStorage& s) { return s.buffer_->swaps_; })
        .def_property_readonly("prefetch_hits", [](PartitionBufferStorage& s) { return s.buffer_->prefetch_hits_; })
        .def_property_readonly("swap_seconds", [](PartitionBufferStorage& s) { return s.buffer_->swap_seconds_; })
        .def_property_readonly("drain_seconds", [](PartitionBufferStorage& s) { return s.buffer_->drain_seconds_; })
        .def_property_readonly("absmax", [](PartitionBufferStorage& s) { return s.buffer_->absmax_; })
        .def_readonly("options", &PartitionBufferStorage::options_);
    m.def("getEdgeBucketOrdering", &getEdgeBucketOrdering, py::arg("edge_bucket_ordering"), py::arg("num_partitions"), py::arg("buffer_capacity"),
          py::arg("fine_to_coarse_ratio"), py::arg("num_cache_partitions"), py::arg("randomly_assign_edge_buckets"), py::arg("generator"));

    py::class_<MariusGraph, std::shared_ptr<MariusGraph>>(m, "MariusGraph")
        .def(py::init([](int64_t n) {
                 auto g = std::make_shared<MariusGraph>();
                 g->num_nodes_in_memory_ = n;
                 return g;
             }),
             py::arg("num_nodes_in_memory"))
        .def_readwrite("num_nodes_in_memory", &MariusGraph::num_nodes_in_memory_)
        .def_readwrite("all_src_sorted_edges", &MariusGraph::all_src_sorted_edges_)
        .def_readwrite("all_dst_sorted_edges", &MariusGraph::all_dst_sorted_edges_)
        .def("sortAllEdges", &MariusGraph::sortAllEdges, py::arg("all_edges"));
    m.def("compute_filter_corruption_global", &compute_filter_corruption_global, py::arg("graph"), py::arg("edges"), py::arg("inverse"));

    py::class_<CorruptNodeNegativeSampler, std::shared_ptr<CorruptNodeNegativeSampler>>(m, "CorruptNodeNegativeSampler")
        .def(py::init<int, int, float, bool, LocalFilterMode, std::shared_ptr<MariusGenerator>>(), py::arg("num_chunks") = 1, py::arg("num_negatives") = 500,
             py::arg("degree_fraction") = 0.0f, py::arg("filtered") = false, py::arg("local_filter_mode") = LocalFilterMode::DEG,
             py::arg("generator") = nullptr)
        .def("getNegatives", &CorruptNodeNegativeSampler::getNegatives, py::arg("graph"), py::arg("edges"), py::arg("inverse") = false)
        .def_readwrite("num_chunks", &CorruptNodeNegativeSampler::num_chunks_)
        .def_readwrite("num_negatives", &CorruptNodeNegativeSampler::num_negatives_)
        .def_readwrite("degree_fraction", &CorruptNodeNegativeSampler::degree_fraction_)
        .def_readwrite("generator", &CorruptNodeNegativeSampler::generator_);

    py::class_<Batch, std::shared_ptr<Batch>>(m, "Batch")
        .def(py::init<bool>(), py::arg("train"))
        .def_readwrite("train", &Batch::train_)
        .def_readwrite("batch_id", &Batch::batch_id_)
        .def_readwrite("edges", &Batch::edges_)
        .def_readwrite("unique_node_indices", &Batch::unique_node_indices_)
        .def_readwrite("node_embeddings", &Batch::node_embeddings_)
        .def_readwrite("node_embeddings_state", &Batch::node_embeddings_state_)
        .def_readwrite("node_embeddings_grad", &Batch::node_embeddings_grad_)
        .def_readwrite("node_gradients", &Batch::node_gradients_)
        .def_readwrite("node_state_update", &Batch::node_state_update_)
        .def_readwrite("src_neg_indices", &Batch::src_neg_indices_)
        .def_readwrite("dst_neg_indices", &Batch::dst_neg_indices_)
        .def_readwrite("src_neg_indices_mapping", &Batch::src_neg_indices_mapping_)
        .def_readwrite("dst_neg_indices_mapping", &Batch::dst_neg_indices_mapping_)
        .def_readwrite("src_neg_filter", &Batch::src_neg_filter_)
        .def_readwrite("dst_neg_filter", &Batch::dst_neg_filter_)
        .def_readwrite("occ_perm", &Batch::occ_perm_)
        .def_readwrite("occ_inverse", &Batch::occ_inverse_)
        .def_readwrite("occ_seg_offsets", &Batch::occ_seg_offsets_)
        .def("accumulateGradients", &Batch::accumulateGradients, py::arg("learning_rate"))
        .def("clear", &Batch::clear);

    // Python subclasses plug in exactly like C++ ones: override __call__ (test_nn.py-style user code); kind() stays -1, so a model that
    // holds one trains through the generic autograd path
    struct PyRelationOperator : RelationOperator {
        using RelationOperator::RelationOperator;
        torch::Tensor operator()(const torch::Tensor& e, const torch::Tensor& r) override {
            PYBIND11_OVERRIDE_PURE_NAME(torch::Tensor, RelationOperator, "__call__", operator(), e, r);
        }
    };
    struct PyComparator : Comparator {
        using Comparator::Comparator;
        torch::Tensor operator()(torch::Tensor s, torch::Tensor d) override { PYBIND11_OVERRIDE_PURE_NAME(torch::Tensor, Comparator, "__call__", operator(), s, d); }
    };
    py::class_<RelationOperator, PyRelationOperator, std::shared_ptr<RelationOperator>>(m, "RelationOperator")
        .def(py::init<>())
        .def("__call__", [](RelationOperator& op, torch::Tensor e, torch::Tensor r) { return op(e, r); });
    py::class_<HadamardOperator, RelationOperator, std::shared_ptr<HadamardOperator>>(m, "HadamardOperator").def(py::init<>());
    py::class_<ComplexHadamardOperator, RelationOperator, std::shared_ptr<ComplexHadamardOperator>>(m, "ComplexHadamardOperator").def(py::init<>());
    py::class_<TranslationOperator, RelationOperator, std::shared_ptr<TranslationOperator>>(m, "TranslationOperator").def(py::init<>());
    py::class_<Comparator, PyComparator, std::shared_ptr<Comparator>>(m, "Comparator")
        .def(py::init<>())
        .def("__call__", [](Comparator& c, torch::Tensor s, torch::Tensor d) { return c(s, d); });
    m.def("pad_and_reshape", &pad_and_reshape, py::arg("input"), py::arg("num_chunks"));
    py::class_<DotCompare, Comparator, std::shared_ptr<DotCompare>>(m, "DotCompare").def(py::init<>());
    py::class_<L2Compare, Comparator, std::shared_ptr<L2Compare>>(m, "L2Compare").def(py::init<>());
    py::class_<CosineCompare, Comparator, std::shared_ptr<CosineCompare>>(m, "CosineCompare").def(py::init<>());

    py::class_<EdgeDecoder, std::shared_ptr<EdgeDecoder>>(m, "EdgeDecoder")
        .def_readwrite("relations", &EdgeDecoder::relations_)
        .def_readwrite("inverse_relations", &EdgeDecoder::inverse_relations_)
        .def_readwrite("comparator", &EdgeDecoder::comparator_)
        .def_readwrite("relation_operator", &EdgeDecoder::relation_operator_)
        .def_readwrite("use_inverse_relations", &EdgeDecoder::use_inverse_relations_)
        .def_readwrite("decoder_method", &EdgeDecoder::decoder_method_)
        .def_readonly("num_relations", &EdgeDecoder::num_relations_)
        .def_readonly("embedding_size", &EdgeDecoder::embedding_size_)
        .def("apply_relation", &EdgeDecoder::apply_relation)
        .def("compute_scores", &EdgeDecoder::compute_scores)
        .def("select_relations", &EdgeDecoder::select_relations, py::arg("indices"), py::arg("inverse") = false);
    auto dec_init = [](auto tag) {
        using T = decltype(tag);
        return py::init([](int num_relations, int embedding_dim, torch::Device device, bool use_inverse_relations, EdgeDecoderMethod method) {
            return std::make_shared<T>(num_relations, embedding_dim, torch::TensorOptions().dtype(torch::kFloat32).device(device), use_inverse_relations, method);
        });
    };
    (void)dec_init;
#define MARIUS_DECODER(NAME)                                                                                                                       \
    py::class_<NAME, EdgeDecoder, std::shared_ptr<NAME>>(m, #NAME)                                                                                 \
        .def(py::init([](int num_relations, int embedding_dim, torch::Device device, bool use_inverse_relations, EdgeDecoderMethod method) {       \
                 return std::make_shared<NAME>(num_relations, embedding_dim, torch::TensorOptions().dtype(torch::kFloat32).device(device),        \
                                               use_inverse_relations, method);                                                                     \
             }),                                                                                                                                   \
             py::arg("num_relations"), py::arg("embedding_dim"), py::arg("device"), py::arg("use_inverse_relations") = true,                       \
             py::arg("decoder_method") = EdgeDecoderMethod::CORRUPT_NODE)                                                                          \
        .def("reset", [](NAME& self) { self.reset(); })                                                                                           \
        .def("named_parameters", [](NAME& self) {                                                                                                  \
            py::dict d;                                                                                                                            \
            for (auto& kv : self.named_parameters()) d[py::str(kv.key())] = kv.value();                                                            \
            return d;                                                                                                                              \
        })                                                                                                                                         \
        .def("clone", [](NAME& self) { return std::dynamic_pointer_cast<NAME>(self.clone()); })
    MARIUS_DECODER(DistMult);
    MARIUS_DECODER(ComplEx);
    MARIUS_DECODER(TransE);
#undef MARIUS_DECODER

    m.def("only_pos_forward", &only_pos_forward, py::arg("decoder"), py::arg("edges"), py::arg("node_embeddings"));
    m.def("node_corrupt_forward",
          [](std::shared_ptr<EdgeDecoder> dec, torch::Tensor edges, torch::Tensor emb, torch::Tensor dst_negs, std::optional<torch::Tensor> src_negs) {
              return node_corrupt_forward(dec, edges, emb, dst_negs, src_negs.has_value() ? *src_negs : torch::Tensor(), nullptr);
          },
          py::arg("decoder"), py::arg("positive_edges"), py::arg("node_embeddings"), py::arg("dst_negs"), py::arg("src_negs") = py::none());

    py::class_<LossFunction, std::shared_ptr<LossFunction>>(m, "LossFunction")
        .def("__call__", [](LossFunction& l, torch::Tensor a, torch::Tensor b, bool scores) { return l(a, b, scores); }, py::arg("y_pred"),
             py::arg("targets"), py::arg("scores") = true);
    auto red = [](const std::string& r) { return (r == "mean" || r == "MEAN") ? LossReduction::MEAN : LossReduction::SUM; };
#define MARIUS_LOSS_BIND(NAME)                                                                                                        \
    py::class_<NAME, LossFunction, std::shared_ptr<NAME>>(m, #NAME)                                                                   \
        .def(py::init([red](std::string reduction) { return std::make_shared<NAME>(red(reduction)); }), py::arg("reduction") = "sum")
    MARIUS_LOSS_BIND(SoftmaxCrossEntropy);
    MARIUS_LOSS_BIND(CrossEntropyLoss);
    MARIUS_LOSS_BIND(BCEAfterSigmoidLoss);
    MARIUS_LOSS_BIND(BCEWithLogitsLoss);
    MARIUS_LOSS_BIND(MSELoss);
    MARIUS_LOSS_BIND(SoftPlusLoss);
#undef MARIUS_LOSS_BIND
    py::class_<RankingLoss, LossFunction, std::shared_ptr<RankingLoss>>(m, "RankingLoss")
        .def(py::init([red](std::string reduction, float margin) { return std::make_shared<RankingLoss>(red(reduction), margin); }),
             py::arg("reduction") = "sum", py::arg("margin") = 0.1f)
        .def_readwrite("margin", &RankingLoss::margin_);
    m.def("getLossFunction", [red](std::string type, std::string reduction, float margin) { return getLossFunction(type, red(reduction), margin); },
          py::arg("type"), py::arg("reduction") = "sum", py::arg("margin") = 0.1f);

    py::class_<LinkPredictionReporter, std::shared_ptr<LinkPredictionReporter>>(m, "LinkPredictionReporter")
        .def(py::init<>())
        .def("computeRanks", &LinkPredictionReporter::computeRanks)
        .def("addResult", &LinkPredictionReporter::addResult)
        .def("clear", &LinkPredictionReporter::clear)
        .def("report", &LinkPredictionReporter::report);

    // model_wrap.cpp:10-13 + a real trampoline: a Python subclass that overrides forward_lp is called by train_batch / evaluate_batch
    struct PyModel : Model {
        using Model::Model;
        std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> forward_lp(std::shared_ptr<Batch> batch, bool train) override {
            using Ret = std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>;
            PYBIND11_OVERRIDE(Ret, Model, forward_lp, batch, train);
        }
        bool custom_forward() const override {
            py::gil_scoped_acquire gil;
            return static_cast<bool>(py::get_override(static_cast<const Model*>(this), "forward_lp"));
        }
    };
    py::class_<ModelConfig>(m, "ModelConfig")
        .def(py::init<>())
        .def_readwrite("decoder", &ModelConfig::decoder)
        .def_readwrite("embedding_dim", &ModelConfig::embedding_dim)
        .def_readwrite("inverse_edges", &ModelConfig::inverse_edges)
        .def_readwrite("decoder_method", &ModelConfig::decoder_method)
        .def_readwrite("loss", &ModelConfig::loss)
        .def_readwrite("loss_reduction", &ModelConfig::loss_reduction)
        .def_readwrite("margin", &ModelConfig::margin)
        .def_readwrite("dense_optimizer", &ModelConfig::dense_optimizer)
        .def_readwrite("dense_lr", &ModelConfig::dense_lr)
        .def_readwrite("eps", &ModelConfig::eps)
        .def_readwrite("beta_1", &ModelConfig::beta_1)
        .def_readwrite("beta_2", &ModelConfig::beta_2)
        .def_readwrite("weight_decay", &ModelConfig::weight_decay)
        .def_readwrite("amsgrad", &ModelConfig::amsgrad)
        .def_readwrite("sparse_lr", &ModelConfig::sparse_lr)
        .def_readwrite("encoder_bias", &ModelConfig::encoder_bias)
        .def_readwrite("encoder_activation", &ModelConfig::encoder_activation)
        .def_readwrite("encoder_bias_init", &ModelConfig::encoder_bias_init);
    py::enum_<ActivationFunction>(m, "ActivationFunction").value("NONE", ActivationFunction::NONE).value("RELU", ActivationFunction::RELU).value("SIGMOID", ActivationFunction::SIGMOID);
    py::class_<GeneralEncoder, std::shared_ptr<GeneralEncoder>>(m, "GeneralEncoder")
        .def(py::init<>())
        .def(py::init([](int output_dim, bool bias, ActivationFunction act, torch::Device dev) { return std::make_shared<GeneralEncoder>(output_dim, bias, act, dev); }),
             py::arg("output_dim"), py::arg("bias"), py::arg("activation"), py::arg("device"))
        .def(py::init<int, bool, ActivationFunction, torch::Device, torch::Tensor>(), py::arg("output_dim"), py::arg("bias"), py::arg("activation"), py::arg("device"),
             py::arg("bias_init"))
        .def("forward", &GeneralEncoder::forward, py::arg("embeddings"))
        .def("backward", &GeneralEncoder::backward, py::arg("grad_encoded"), py::arg("encoded"))
        .def("has_post_hook", &GeneralEncoder::has_post_hook)
        .def_readonly("bias", &GeneralEncoder::bias_)
        .def_readonly("bias_grad", &GeneralEncoder::bias_grad_)
        .def_readonly("activation", &GeneralEncoder::activation_);
    m.def("initModelFromConfig", static_cast<std::shared_ptr<Model> (*)(const ModelConfig&, std::vector<torch::Device>, int, bool)>(&initModelFromConfig),
          py::arg("model_config"), py::arg("devices"), py::arg("num_relations"), py::arg("train"));
    py::class_<Model, PyModel, std::shared_ptr<Model>>(m, "Model", py::dynamic_attr())
        .def(py::init<std::shared_ptr<EdgeDecoder>, std::shared_ptr<LossFunction>, std::shared_ptr<LinkPredictionReporter>, torch::Device>(),
             py::arg("decoder"), py::arg("loss"), py::arg("reporter"), py::arg("device"))
        .def("forward_lp", [](Model& self, std::shared_ptr<Batch> b, bool train) { return self.Model::forward_lp(b, train); }, py::arg("batch"),
             py::arg("train") = true)
        .def("fused_ok", &Model::fused_ok)
        .def("has_post_hook", &Model::has_post_hook)
        .def("set_encoder", &Model::set_encoder, py::arg("encoder"))
        .def_readonly("encoder", &Model::encoder_)
        .def("broadcast", &Model::broadcast, py::arg("devices"))
        .def("all_reduce", &Model::all_reduce)
        .def("set_process_group", &Model::set_process_group, py::arg("group_name"))
        .def("named_parameters", [](Model& self) {
            py::dict d;
            for (auto& kv : self.named_parameters()) d[py::str(kv.key())] = kv.value();
            return d;
        })
        .def("train_batch", &Model::train_batch, py::arg("batch"), py::arg("call_step") = true)
        .def("evaluate_batch", &Model::evaluate_batch)
        .def("save", &Model::save, py::arg("directory"))
        .def("load", &Model::load, py::arg("directory"), py::arg("train") = true)
        .def("setup_optimizers", static_cast<void (Model::*)(float)>(&Model::setup_optimizers), py::arg("dense_lr"))
        .def("setup_optimizers", [](Model& self, const ModelConfig& c) { self.setup_optimizers(std::make_shared<ModelConfig>(c)); }, py::arg("model_config"))
        .def("setup_optimizer", &Model::setup_optimizer, py::arg("type"), py::arg("lr"), py::arg("eps") = 1e-10f, py::arg("beta_1") = 0.9f,
             py::arg("beta_2") = 0.999f, py::arg("weight_decay") = 0.f, py::arg("amsgrad") = false)
        .def("step", &Model::step)
        .def("clear_grad", &Model::clear_grad)
        .def_readwrite("sparse_lr", &Model::sparse_lr_)
        .def_readwrite("fuse_endpoint_update", &Model::fuse_endpoint_update_)
        .def_readonly("last_step_fused_below", &Model::last_fused_below_)
        .def_readwrite("decoder", &Model::decoder_)
        .def_readwrite("reporter", &Model::reporter_)
        .def_readonly("loss", &Model::loss_)
        .def("dense_state", &Model::dense_state)  // [parameters..., optimizer state tensors...] of the dense (relation-table) optimizers, aliased
        .def_readonly("ranges_valid", &Model::ranges_valid_)  // the flash path packs fp16 operand halves (table magnitude bounds are tracked)
        .def_readonly("range_state", &Model::range_state_)
        .def("track_ranges", &Model::track_ranges, py::arg("table"))
        .def("drop_ranges", &Model::drop_ranges)
        .def_readonly("rel_ranges_valid", &Model::rel_ranges_valid_)
        .def_readonly("row_bound", &Model::row_bound_)  // running bound of the rows of gathered batches (Model::bind_ranges)
        .def("tracks", &Model::tracks, py::arg("table"))
        .def("touch_relations", &Model::touch_relations)  // call after writing the relation tables through a raw pointer (ATen writes are seen by their version)
        .def("touch_table", &Model::touch_table)          // the same for the tracked node table (raw-pointer writers: ctypes scatter calls, user kernels)
        // what the last training forward packed its operand records with: "fp16" (22 significand bits per operand) or "bf16" (16)
        .def_property_readonly("last_step_records", [](Model& self) { return std::string(self.ctx_.layout.flash ? (self.ctx_.desc.absmax ? "fp16" : "bf16") : "none"); })
        .def_property_readonly("last_step_flash", [](Model& self) { return self.ctx_.layout.flash != 0; })  // did the last fused step take the flash decoder path
        .def_readonly("relations_grad", &Model::relations_grad_)
        .def_readonly("inverse_relations_grad", &Model::inverse_relations_grad_);

    py::class_<DataLoader, std::shared_ptr<DataLoader>>(m, "DataLoader")
        .def(py::init<std::shared_ptr<InMemory>, std::shared_ptr<Storage>, std::shared_ptr<Storage>, std::shared_ptr<CorruptNodeNegativeSampler>,
                      std::shared_ptr<MariusGenerator>, int64_t, bool>(),
             py::arg("edges"), py::arg("node_embeddings"), py::arg("node_embeddings_state"), py::arg("negative_sampler"), py::arg("generator"),
             py::arg("batch_size"), py::arg("train") = true)
        .def("initializeBatches", &DataLoader::initializeBatches, py::arg("shuffle") = true)
        .def_readonly("shuffle_ahead_hits", &DataLoader::shuffle_ahead_hits_)
        .def_readonly("shuffle_ahead_misses", &DataLoader::shuffle_ahead_misses_)
        .def_readwrite("full_batches_only", &DataLoader::full_batches_only_)
        .def_readwrite("generator", &DataLoader::generator_)
        .def_readwrite("words_between_epochs", &DataLoader::words_between_epochs_)
        .def("wordsPerEpoch", &DataLoader::wordsPerEpoch, py::arg("with_permutation"))
        .def("hasNextBatch", &DataLoader::hasNextBatch)
        .def("getBatch", &DataLoader::getBatch, py::arg("exact_unique") = true)
        .def("loadGPUParameters", &DataLoader::loadGPUParameters)
        .def("updateEmbeddings", &DataLoader::updateEmbeddings, py::arg("batch"), py::arg("gpu") = true)
        .def("getNumEdges", &DataLoader::getNumEdges)
        .def("setEdgeBucketSizes", &DataLoader::setEdgeBucketSizes, py::arg("sizes"))
        .def("loadStorage", &DataLoader::loadStorage)
        .def("nextEpoch", &DataLoader::nextEpoch, py::arg("write") = true)
        .def_readonly("buffer_states", &DataLoader::buffer_states_)
        .def_readonly("edge_buckets_per_buffer", &DataLoader::edge_buckets_per_buffer_)
        .def_readonly("active_edges", &DataLoader::active_edges_)
        .def_readonly("graph", &DataLoader::graph_)
        .def_readonly("active_perm", &DataLoader::active_perm_)
        .def_readonly("num_unique", &DataLoader::last_num_unique_);

    py::class_<SynchronousTrainer, std::shared_ptr<SynchronousTrainer>>(m, "SynchronousTrainer")
        .def(py::init<std::shared_ptr<DataLoader>, std::shared_ptr<Model>>())
        .def("train", &SynchronousTrainer::train, py::arg("num_epochs") = 1, py::call_guard<py::gil_scoped_release>())
        .def("train_steps", &SynchronousTrainer::train_steps, py::arg("n"), py::call_guard<py::gil_scoped_release>())
        .def("train_one", &SynchronousTrainer::train_one, py::arg("fused") = true, py::call_guard<py::gil_scoped_release>())
        .def_readwrite("fused_update", &SynchronousTrainer::fused_update_)
        .def_readonly("last_epoch_seconds", &SynchronousTrainer::last_epoch_seconds_)
        .def_readonly("last_edges_per_second", &SynchronousTrainer::last_edges_per_second_);
    py::class_<PipelineTrainer, std::shared_ptr<PipelineTrainer>>(m, "PipelineTrainer")
        .def(py::init<std::shared_ptr<DataLoader>, std::shared_ptr<Model>, int, bool>(), py::arg("dataloader"), py::arg("model"),
             py::arg("staleness_bound") = 16, py::arg("stale_parameters") = false)
        .def("train", &PipelineTrainer::train, py::arg("num_epochs") = 1, py::call_guard<py::gil_scoped_release>())
        .def_readonly("staleness_bound", &PipelineTrainer::staleness_bound_)
        .def_readonly("stale_parameters", &PipelineTrainer::stale_parameters_)
        .def_readonly("last_epoch_seconds", &PipelineTrainer::last_epoch_seconds_)
        .def_readonly("last_edges_per_second", &PipelineTrainer::last_edges_per_second_);
    py::class_<ShardedTrainer, std::shared_ptr<ShardedTrainer>>(m, "ShardedTrainer")
        .def(py::init<std::shared_ptr<DataLoader>, std::shared_ptr<Model>, torch::Tensor, torch::Tensor, int, int, int64_t, std::string, std::string, int, int>(),
             py::arg("dataloader"), py::arg("model"), py::arg("shard_table"), py::arg("shard_state"), py::arg("rank"), py::arg("world"), py::arg("num_nodes"),
             py::arg("group_name"), py::arg("side_group_name"), py::arg("staleness") = 1, py::arg("sync_interval") = 16)
        .def("step", &ShardedTrainer::step, py::call_guard<py::gil_scoped_release>())
        .def("train_steps", &ShardedTrainer::train_steps, py::arg("n"), py::call_guard<py::gil_scoped_release>())
        .def("finish", &ShardedTrainer::finish, py::call_guard<py::gil_scoped_release>())
        .def_readwrite("host_seconds", &ShardedTrainer::host_seconds_)
        .def("ranks", &ShardedTrainer::ranks)
        .def("backend", &ShardedTrainer::backend)
        .def_property_readonly("exchange_bytes", [](ShardedTrainer& t) { return std::vector<int64_t>(t.exchange_bytes_, t.exchange_bytes_ + 3); })
        .def("reset_exchange_bytes", [](ShardedTrainer& t) { t.exchange_bytes_[0] = t.exchange_bytes_[1] = t.exchange_bytes_[2] = 0; })
        .def_readonly("steps", &ShardedTrainer::steps_)
        .def_property_readonly("fixed_capacity", &ShardedTrainer::fixed_capacity)
        .def_property_readonly("pair_capacity", &ShardedTrainer::pair_capacity)
        .def_property_readonly("useful_rows", [](ShardedTrainer& t) { return std::vector<int64_t>(t.useful_rows_, t.useful_rows_ + 2); })
        .def_property_readonly("phase_seconds", [](ShardedTrainer& t) { return std::vector<double>(t.phase_seconds_, t.phase_seconds_ + 6); })
        .def_readonly("torn_reads", &ShardedTrainer::torn_reads_)
        .def("describe_state", &ShardedTrainer::describe_state)
        .def("enable_spans", &ShardedTrainer::enable_spans)
        .def("reset_counters", &ShardedTrainer::reset_counters)
        .def_property_readonly("span_ms", [](ShardedTrainer& t) {
            std::vector<double> v(7, 0.0);
            for (int k = 0; k < 7; ++k) v[k] = t.span_n_[k] ? t.span_ms_[k] / (double)t.span_n_[k] : 0.0;
            return v;
        });
    m.def("c10d_exchange_selftest", &c10d_exchange_selftest, py::arg("group_name"), py::arg("send"), py::arg("send_counts"), py::arg("to_reduce"),
          py::call_guard<py::gil_scoped_release>());
    py::class_<SynchronousEvaluator, std::shared_ptr<SynchronousEvaluator>>(m, "SynchronousEvaluator")
        .def(py::init<std::shared_ptr<DataLoader>, std::shared_ptr<Model>>())
        .def_readonly("dataloader", &SynchronousEvaluator::dataloader_)
        .def("evaluate", &SynchronousEvaluator::evaluate);
}
