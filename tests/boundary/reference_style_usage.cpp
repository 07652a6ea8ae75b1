// Compile-only check (tests/test_boundary_cpu.py): user code written against the signatures of the REFERENCE's headers
//   nn/decoders/edge/comparators.h:13-17, relation_operators.h:11-15, edge_decoder.h:13-31, distmult.h:10-16 (complex.h, transe.h),
//   nn/decoders/decoder.h:12-17, nn/decoders/edge/decoder_methods.h:11-21, nn/loss.h:21-31, nn/model.h:16-65,
//   data/samplers/negative.h:31-57, data/batch.h:32-89, storage/storage.h:35-86
// must compile unchanged against marius_amd/csrc/host/marius_host.h (only the namespace differs).  Nothing here is executed.
#include "marius_host.h"
#include "partition_buffer.h"

using namespace marius_amd;
using torch::Tensor;

struct MyComparator : public Comparator {  // comparators.h:16: virtual torch::Tensor operator()(torch::Tensor src, torch::Tensor dst)
    Tensor operator()(Tensor src, Tensor dst) override { return (src * dst).sum(-1); }
};
struct MyOperator : public RelationOperator {  // relation_operators.h:14: virtual torch::Tensor operator()(const torch::Tensor &embs, const torch::Tensor &rels)
    Tensor operator()(const Tensor& embs, const Tensor& rels) override { return rels.defined() ? embs + rels : embs; }
};
struct MyLoss : public LossFunction {  // loss.h:30: virtual torch::Tensor operator()(torch::Tensor y_pred, torch::Tensor targets, bool scores)
    Tensor operator()(Tensor y_pred, Tensor targets, bool scores) override { return scores ? y_pred.sum() - targets.sum() : y_pred.sum(); }
};
struct MyModel : public Model {  // model.h:38 overridden from user code (test_nn.py:113-127 does it from Python)
    using Model::Model;
    std::tuple<Tensor, Tensor, Tensor, Tensor> forward_lp(shared_ptr<Batch> batch, bool train) override { return Model::forward_lp(batch, train); }
};

void reference_style_usage() {
    torch::TensorOptions opts;
    // distmult.h:12-13 — every argument after embedding_dim defaulted
    shared_ptr<DistMult> dm = std::make_shared<DistMult>(10, 8, opts, true, EdgeDecoderMethod::CORRUPT_NODE);
    shared_ptr<ComplEx> cx = std::make_shared<ComplEx>(10, 8);
    shared_ptr<TransE> te = std::make_shared<TransE>(10, 8, opts, false, EdgeDecoderMethod::ONLY_POS);
    std::shared_ptr<torch::nn::Module> copy = dm->clone();  // torch::nn::Cloneable<DistMult>
    dm->reset();
    auto named = dm->named_parameters();                     // relation_embeddings / inverse_relation_embeddings (distmult.cpp:21-27)
    (void)named["relation_embeddings"];
    shared_ptr<EdgeDecoder> ed = dm;                         // edge_decoder.h:13: class EdgeDecoder : public Decoder
    shared_ptr<Decoder> dec = ed;
    LearningTask task = dec->learning_task_;
    (void)task;
    ed->comparator_ = std::make_shared<MyComparator>();
    ed->relation_operator_ = std::make_shared<MyOperator>();
    int nr = ed->num_relations_ + ed->embedding_size_;
    bool inv = ed->use_inverse_relations_;
    (void)nr; (void)inv; (void)ed->tensor_options_; (void)ed->decoder_method_;
    Tensor rels = ed->select_relations(torch::zeros({2}, torch::kInt64));  // inverse = false by default
    Tensor adj = ed->apply_relation(rels, ed->relations_);
    Tensor sc = ed->compute_scores(adj, rels);
    // decoder_methods.h:11-21
    std::tuple<Tensor, Tensor> pos_only = only_pos_forward(ed, Tensor(), Tensor());
    std::tuple<Tensor, Tensor, Tensor, Tensor> all4 = node_corrupt_forward(ed, Tensor(), Tensor(), Tensor(), Tensor());
    // model.h:33 — (encoder, decoder, loss, reporter = nullptr, optimizers = {})
    shared_ptr<LossFunction> loss = std::make_shared<SoftmaxCrossEntropy>(LossReduction::SUM);
    shared_ptr<GeneralEncoder> enc = nullptr;
    shared_ptr<Model> model = std::make_shared<Model>(enc, dec, loss);
    shared_ptr<Reporter> rep = std::make_shared<LinkPredictionReporter>();
    shared_ptr<Model> model2 = std::make_shared<MyModel>(enc, dec, std::make_shared<MyLoss>(), rep, std::vector<shared_ptr<Optimizer>>{});
    shared_ptr<ModelConfig> cfg = std::make_shared<ModelConfig>();
    model->setup_optimizers(cfg);                            // model.h:58
    shared_ptr<Model> model3 = initModelFromConfig(cfg, {torch::Device(torch::kCPU)}, 5, true);  // model.h:65
    model->broadcast({torch::Device(torch::kCPU)});
    model->all_reduce();
    float lr = model->sparse_lr_;
    (void)lr; (void)model->device_; (void)model->learning_task_; (void)model->encoder_; (void)model->decoder_; (void)model->loss_function_;
    (void)model->reporter_; (void)model->optimizers_;
    shared_ptr<Batch> batch = std::make_shared<Batch>(true);
    std::tuple<Tensor, Tensor, Tensor, Tensor> fw = model->forward_lp(batch, true);
    model->train_batch(batch);        // call_step = true
    model->train_batch(batch, false);
    model->evaluate_batch(batch);
    model->clear_grad();
    model->step();
    // a user optimiser over the model's parameters (the reference registers them with requires_grad(true), distmult.cpp:21-27):
    // train_batch(batch, false) leaves relations_.grad(), the optimiser steps them
    torch::optim::SGD user_opt(model->parameters(), torch::optim::SGDOptions(0.1));
    model->train_batch(batch, false);
    for (auto& kv : model->named_parameters()) (void)kv.value().grad();
    user_opt.step();
    user_opt.zero_grad();
    model->save("dir/");
    model->load("dir/", true);
    // negative.h:45-57
    shared_ptr<NegativeSampler> ns = std::make_shared<CorruptNodeNegativeSampler>(10, 500, 0.5f, false, LocalFilterMode::DEG);
    std::tuple<Tensor, Tensor> negs = ns->getNegatives(std::make_shared<MariusGraph>());  // edges = {}, inverse = false
    // batch.h: public tensor fields + the two calls the trainer makes
    (void)batch->edges_; (void)batch->unique_node_indices_; (void)batch->node_embeddings_; (void)batch->node_embeddings_state_;
    (void)batch->node_gradients_; (void)batch->node_state_update_; (void)batch->src_neg_indices_mapping_; (void)batch->dst_neg_indices_mapping_;
    (void)batch->src_neg_filter_; (void)batch->dst_neg_filter_;
    batch->accumulateGradients(0.1f);
    batch->clear();
    // storage.h:35-86
    shared_ptr<Storage> st = std::make_shared<InMemory>("embeddings.bin", 100, 8, torch::kFloat32, torch::Device(torch::kCPU));
    Tensor rows = st->indexRead(torch::zeros({2}, torch::kInt64));
    st->indexAdd(torch::zeros({2}, torch::kInt64), rows);
    st->load(); st->write(); st->unload(true);
    (void)st->dim0_size_; (void)st->dim1_size_; (void)st->dtype_; (void)st->data_; (void)st->device_; (void)st->filename_; (void)st->edge_bucket_sizes_;
    (void)pos_only; (void)all4; (void)fw; (void)negs; (void)sc; (void)copy; (void)cx; (void)te; (void)model2; (void)model3;
}
