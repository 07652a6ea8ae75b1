// ORACLE (test infrastructure, NOT product code).
//
// Thin C-ABI driver around the REFERENCE's own score arithmetic, compiled from the sources where they lie:
//   /root/reference/src/cpp/src/nn/decoders/edge/comparators.cpp          (DotCompare / L2Compare / CosineCompare, pad_and_reshape)
//   /root/reference/src/cpp/src/nn/decoders/edge/relation_operators.cpp   (Hadamard / ComplexHadamard / Translation / NoOp)
// These two translation units need only the reference headers + libtorch (common/datatypes.h -> torch/torch.h).
// Every other file of the path (decoder_methods.cpp, negative.cpp, batch.cpp, loss.cpp, util.cpp, storage.cpp)
// pulls in reporting/logger.h -> spdlog, an un-vendored submodule (third_party/spdlog is empty): unbuildable here
// without writing stand-ins, so they are NOT built; oracle/lp_oracle.py restates them instead.
//
// Built by oracle/Makefile into oracle/_ref/libmarius_ref.so (git-ignored; travels to the GPU box).
// Used by tests to validate oracle/lp_oracle.py and, optionally, as the "reference" CPU leg.
#include <torch/torch.h>

#include "nn/decoders/edge/comparators.h"
#include "nn/decoders/edge/relation_operators.h"

namespace {
torch::Tensor wrap(float* p, std::vector<int64_t> sizes) { return torch::from_blob(p, sizes, torch::kFloat32); }

std::shared_ptr<RelationOperator> make_op(int op) {
    switch (op) {
        case 0: return std::make_shared<HadamardOperator>();
        case 1: return std::make_shared<ComplexHadamardOperator>();
        case 2: return std::make_shared<TranslationOperator>();
        default: return std::make_shared<NoOp>();
    }
}
std::shared_ptr<Comparator> make_cmp(int cmp) {
    switch (cmp) {
        case 0: return std::make_shared<DotCompare>();
        case 1: return std::make_shared<L2Compare>();
        default: return std::make_shared<CosineCompare>();
    }
}
}  // namespace

extern "C" {

// out[B,d] = op(embs[B,d], rels[B,d])
int ref_relation_op(int op, float* embs, float* rels, int64_t B, int64_t d, float* out) {
    try {
        auto r = (*make_op(op))(wrap(embs, {B, d}), wrap(rels, {B, d}));
        wrap(out, {B, d}).copy_(r);
        return 0;
    } catch (...) { return 1; }
}

// out[B] = cmp(src[B,d], dst[B,d])
int ref_compare_same(int cmp, float* src, float* dst, int64_t B, int64_t d, float* out) {
    try {
        auto r = (*make_cmp(cmp))(wrap(src, {B, d}), wrap(dst, {B, d}));
        wrap(out, {B}).copy_(r);
        return 0;
    } catch (...) { return 1; }
}

// out[C*ceil(B/C), N] = cmp(src[B,d], negs[C,N,d])
int ref_compare_neg(int cmp, float* src, float* negs, int64_t B, int64_t C, int64_t N, int64_t d, float* out) {
    try {
        auto r = (*make_cmp(cmp))(wrap(src, {B, d}), wrap(negs, {C, N, d}));
        wrap(out, {r.size(0), r.size(1)}).copy_(r);
        return 0;
    } catch (...) { return 1; }
}

// One direction of node_corrupt scoring + SoftmaxCE(SUM) through the reference's operators, with autograd:
//   adj = op(src, rel); pos = cmp(adj, dst); neg = cmp(adj, negs); loss = sum_i -pos_i + log(e^pos_i + sum_j e^neg_ij)
// Writes pos[B], neg[B',N], loss[1], and gradients w.r.t. src, rel, dst, negs. (B % C == 0 required here.)
int ref_score_fwd_bwd(int op, int cmp, float* src, float* rel, float* dst, float* negs, int64_t B, int64_t C, int64_t N,
                      int64_t d, float* pos_out, float* neg_out, float* loss_out, float* g_src, float* g_rel, float* g_dst,
                      float* g_negs) {
    try {
        auto s = wrap(src, {B, d}).clone().requires_grad_(true);
        auto r = wrap(rel, {B, d}).clone().requires_grad_(true);
        auto t = wrap(dst, {B, d}).clone().requires_grad_(true);
        auto n = wrap(negs, {C, N, d}).clone().requires_grad_(true);
        auto relop = make_op(op);
        auto comp = make_cmp(cmp);
        auto adj = (*relop)(s, r);
        auto pos = (*comp)(adj, t);
        auto neg = (*comp)(adj, n);
        auto y = torch::cat({pos.unsqueeze(1), neg.logsumexp(1, true)}, -1);
        auto labels = torch::zeros({pos.size(0)}, torch::kInt64);
        auto loss = torch::nn::functional::cross_entropy(y, labels, torch::nn::functional::CrossEntropyFuncOptions().reduction(torch::kSum));
        loss.backward();
        wrap(pos_out, {B}).copy_(pos.detach());
        wrap(neg_out, {neg.size(0), N}).copy_(neg.detach());
        loss_out[0] = loss.item<float>();
        wrap(g_src, {B, d}).copy_(s.grad());
        wrap(g_rel, {B, d}).copy_(r.grad());
        wrap(g_dst, {B, d}).copy_(t.grad());
        wrap(g_negs, {C, N, d}).copy_(n.grad());
        return 0;
    } catch (...) { return 1; }
}

int ref_set_num_threads(int n) {
    at::set_num_threads(n);
    return at::get_num_threads();
}
}
