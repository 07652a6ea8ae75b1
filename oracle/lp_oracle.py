"""ORACLE — test infrastructure, NOT product code.

CPU restatement (libtorch CPU ops, op-for-op in the reference's order) of Marius's link-prediction
training hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Every function cites the reference file:line it follows (paths relative to /root/reference/src/cpp).
The tensor arithmetic of the reference is libtorch (ATen) — a third-party dependency not vendored under
/root/reference (setup.cfg:39 `torch>=1.7.1`; pinned here to the container's torch 2.10.0).  The restatement
therefore calls the same ATen CPU ops the reference calls, in the same order, so on the same inputs it is
the reference's CPU result.  Pinning: tests/test_oracle_*.py check it against
  * the reference's own known answers (test/python/bindings/integration/test_nn.py:148-160,
    test_data.py:34-47, test_nn.py:197-207) committed as tests/golden/ref_known_answers.json,
  * oracle/_ref (the reference's comparators.cpp + relation_operators.cpp compiled where they lie),
  * the RNG stream of torch.randint/randperm (see oracle/mt19937_aten.c).
"""
import math
from typing import List, Optional, Tuple

import torch


# ----------------------------------------------------------------------------- a5: map_tensors
def map_tensors(unmapped: List[torch.Tensor]) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """common/util.cpp:180-205 — cat -> _unique2(sorted=True, return_inverse=True) -> narrow per input."""
    for t in unmapped:
        if t.dim() > 1:
            raise RuntimeError("Input tensors must be 1D")
    all_ids = torch.cat(unmapped)
    uniq, inverse = torch.unique(all_ids, sorted=True, return_inverse=True)
    out, off = [], 0
    for t in unmapped:
        out.append(inverse.narrow(0, off, t.size(0)))
        off += t.size(0)
    return uniq, out


# ----------------------------------------------------------------------------- a9: relation operators
def hadamard(embs, rels):
    """relation_operators.cpp:7-12"""
    return embs if rels is None else embs * rels


def complex_hadamard(embs, rels):
    """relation_operators.cpp:14-35 — halves split at d/2: [re | im]."""
    if rels is None:
        return embs
    dim = embs.size(1)
    real_len = dim // 2
    imag_len = dim - dim // 2
    re_e, im_e = embs.narrow(1, 0, real_len), embs.narrow(1, real_len, imag_len)
    re_r, im_r = rels.narrow(1, 0, real_len), rels.narrow(1, real_len, imag_len)
    out = torch.zeros_like(embs)
    out = torch.cat([(re_e * re_r) - (im_e * im_r), (re_e * im_r) + (im_e * re_r)], dim=1)
    return out


def translation(embs, rels):
    """relation_operators.cpp:37-42"""
    return embs if rels is None else embs + rels


REL_OPS = {"hadamard": hadamard, "complex_hadamard": complex_hadamard, "translation": translation,
           "noop": lambda e, r: e}


# ----------------------------------------------------------------------------- a10: comparators
def pad_and_reshape(x: torch.Tensor, num_chunks: int) -> torch.Tensor:
    """comparators.cpp:7-20 — note ceil((float)num_pos / num_chunks) in float32."""
    num_pos = x.size(0)
    per_chunk = int(math.ceil(float(torch.tensor(num_pos, dtype=torch.float32) / num_chunks)))
    if per_chunk != num_pos // num_chunks:
        new_size = per_chunk * num_chunks
        x = torch.nn.functional.pad(x, (0, 0, 0, new_size - num_pos))
    return x.view(num_chunks, per_chunk, x.size(1))


def dot_compare(src, dst):
    """comparators.cpp:62-73"""
    if src.shape == dst.shape:
        return (src * dst).sum(-1)
    src = pad_and_reshape(src, dst.size(0))
    return src.bmm(dst.transpose(-1, -2)).flatten(0, 1)


def l2_compare(src, dst):
    """comparators.cpp:22-41 — returns a POSITIVE distance (reference quirk, kept)."""
    if src.shape == dst.shape:
        return torch.pairwise_distance(src, dst)
    src = pad_and_reshape(src, dst.size(0))
    x2 = src.pow(2).sum(2).unsqueeze(2)
    y2 = dst.pow(2).sum(2).unsqueeze(1)
    xy = torch.matmul(src, dst.transpose(1, 2))
    return torch.sqrt(torch.clamp_min(x2 + y2 - 2 * xy, 1e-8)).flatten(0, 1).clone()


def cosine_compare(src, dst):
    """comparators.cpp:43-60 — normalised tensors are computed then NOT used (reference quirk, kept)."""
    if src.shape == dst.shape:
        return (src * dst).sum(-1)
    src = pad_and_reshape(src, dst.size(0))
    return src.bmm(dst.transpose(-1, -2)).flatten(0, 1)


COMPARATORS = {"dot": dot_compare, "l2": l2_compare, "cosine": cosine_compare}

# decoder name -> (relation operator, comparator, relation init)   distmult.cpp:7-27, complex.cpp:7-29, transe.cpp:7-28
DECODERS = {"DISTMULT": ("hadamard", "dot"), "COMPLEX": ("complex_hadamard", "dot"), "TRANSE": ("translation", "l2")}


def init_relations(decoder: str, num_relations: int, d: int):
    """distmult.cpp:21-27 (ones), complex.cpp:21-29 (first d/2 columns 1, rest 0), transe.cpp:21-28 (zeros)."""
    if decoder == "DISTMULT":
        return torch.ones(num_relations, d)
    if decoder == "COMPLEX":
        r = torch.zeros(num_relations, d)
        r[:, : d // 2] = 1
        return r
    if decoder == "TRANSE":
        return torch.zeros(num_relations, d)
    raise ValueError(decoder)


# ----------------------------------------------------------------------------- a11: decoder methods
def only_pos_forward(decoder, edges, node_embeddings, relations=None, inverse_relations=None):
    """decoder_methods.cpp:7-42"""
    op, cmp = REL_OPS[DECODERS[decoder][0]], COMPARATORS[DECODERS[decoder][1]]
    if edges.size(1) not in (2, 3):
        raise RuntimeError("Edge list must be a 3 or 2 column tensor")
    src = node_embeddings.index_select(0, edges[:, 0])
    dst = node_embeddings.index_select(0, edges[:, -1])
    inv_pos = None
    if edges.size(1) == 3:
        rel_ids = edges[:, 1]
        pos = cmp(op(src, relations.index_select(0, rel_ids)), dst)
        if inverse_relations is not None:
            inv_pos = cmp(op(dst, inverse_relations.index_select(0, rel_ids)), src)
    else:
        pos = cmp(src, dst)
    return pos, inv_pos


def node_corrupt_forward(decoder, edges, node_embeddings, dst_negs, src_negs, relations=None, inverse_relations=None):
    """decoder_methods.cpp:57-114 — returns (pos, neg, inv_pos, inv_neg); pos padded to B' when B % C != 0."""
    op, cmp = REL_OPS[DECODERS[decoder][0]], COMPARATORS[DECODERS[decoder][1]]
    if edges.size(1) not in (2, 3):
        raise RuntimeError("Edge list must be a 3 or 2 column tensor")
    src = node_embeddings.index_select(0, edges[:, 0])
    dst = node_embeddings.index_select(0, edges[:, -1])
    dst_neg_embs = node_embeddings.index_select(0, dst_negs.flatten(0, 1)).reshape(dst_negs.size(0), dst_negs.size(1), -1)
    inv_pos = inv_neg = None
    if edges.size(1) == 3:
        rel_ids = edges[:, 1]
        adj_src = op(src, relations.index_select(0, rel_ids))
        pos = cmp(adj_src, dst)
        neg = cmp(adj_src, dst_neg_embs)
        if inverse_relations is not None:
            adj_dst = op(dst, inverse_relations.index_select(0, rel_ids))
            src_neg_embs = node_embeddings.index_select(0, src_negs.flatten(0, 1)).reshape(src_negs.size(0), src_negs.size(1), -1)
            inv_pos = cmp(adj_dst, src)
            inv_neg = cmp(adj_dst, src_neg_embs)
    else:
        pos = cmp(src, dst)
        neg = cmp(src, dst_neg_embs)
    if pos.size(0) != neg.size(0):
        extra = neg.size(0) - pos.size(0)
        pos = torch.nn.functional.pad(pos, (0, extra))
        if inv_pos is not None:
            inv_pos = torch.nn.functional.pad(inv_pos, (0, extra))
    return pos, neg, inv_pos, inv_neg


# ----------------------------------------------------------------------------- a4: score filter
def deg_negative_local_filter(deg_sample_indices: Optional[torch.Tensor], edges: torch.Tensor) -> torch.Tensor:
    """negative.cpp:21-39"""
    if deg_sample_indices is None:
        return torch.empty(0, 2, dtype=torch.int64)
    num_chunks = deg_sample_indices.size(0)
    chunk_size = math.ceil(edges.size(0) / num_chunks)
    num_deg = deg_sample_indices.size(1)
    chunk_ids = deg_sample_indices.div(chunk_size, rounding_mode="trunc")
    inv_mask = chunk_ids - torch.arange(0, num_chunks).view(num_chunks, -1)
    mask = inv_mask == 0
    temp_idx = torch.nonzero(mask)
    id_offsets = deg_sample_indices.flatten(0, 1).index_select(0, temp_idx[:, 0] * num_deg + temp_idx[:, 1])
    return torch.stack([id_offsets, temp_idx[:, 1]]).transpose(0, 1)


def apply_score_filter(scores, filt):
    """negative.cpp:306-311 — in place index_put_(-1e9)."""
    if filt is not None and filt.numel() > 0:
        scores.index_put_((filt[:, 0], filt[:, 1]), torch.tensor(-1e9, dtype=scores.dtype))
    return scores


# ----------------------------------------------------------------------------- a13: loss
def softmax_cross_entropy(pos, neg, reduction="sum"):
    """loss.cpp:50-67 — CE([pos, logsumexp(neg,1)], label 0)."""
    y_pred = torch.cat([pos.unsqueeze(1), neg.logsumexp(1, True)], -1)
    labels = torch.zeros(pos.size(0), dtype=torch.int64)
    return torch.nn.functional.cross_entropy(y_pred, labels, reduction=reduction)


def _labels(pos, neg):
    """scores_to_labels(pos, neg.flatten(0, 1), one_hot=True), loss.cpp:37-48: 1-D [B' + B' N] predictions and 1 / 0 labels."""
    y = torch.cat([pos, neg.flatten(0, 1)], -1)
    return y, torch.cat([torch.ones_like(pos), torch.zeros_like(neg.flatten(0, 1))], -1)


def loss_function(kind, pos, neg, reduction="sum", margin=0.1):
    """Every LossFunction::operator()(pos, neg, scores=true) of loss.cpp:50-187, with the same torch.nn.functional calls."""
    F = torch.nn.functional
    kind = kind.upper()
    if kind == "SOFTMAX_CE":
        return softmax_cross_entropy(pos, neg, reduction)
    if kind == "RANKING":  # :69-87  margin_ranking_loss(neg, pos.unsqueeze(1), -1, margin)
        return F.margin_ranking_loss(neg, pos.unsqueeze(1), pos.new_full((1, 1), -1), margin=margin, reduction=reduction)
    if kind == "CROSS_ENTROPY":  # :89-103  CE over [pos, neg...] with label 0
        y = torch.cat([pos.unsqueeze(1), neg], -1)
        return F.cross_entropy(y, torch.zeros(pos.size(0), dtype=torch.int64), reduction=reduction)
    y, lab = _labels(pos, neg)
    if kind == "BCE_AFTER_SIGMOID":  # :105-123
        return F.binary_cross_entropy(y.sigmoid(), lab, reduction=reduction)
    if kind == "BCE_WITH_LOGITS":  # :125-143
        return F.binary_cross_entropy_with_logits(y, lab, reduction=reduction)
    if kind == "MSE":  # :145-163
        return F.mse_loss(y, lab, reduction=reduction)
    if kind == "SOFTPLUS":  # :165-187
        out = F.softplus(-1 * (2 * lab - 1) * y)
        return out.mean() if reduction == "mean" else out.sum()
    raise ValueError(kind)


# ----------------------------------------------------------------------------- a15: sparse Adagrad rule
def accumulate_gradients(grad, state, lr):
    """data/batch.cpp:62-79 — returns (node_gradients_ = dw, node_state_update_ = ds); mutates state like the reference."""
    ds = grad.pow(2)
    state.add_(ds)
    dw = -lr * (grad / (state.sqrt().add_(1e-10)))
    return dw, ds


# ----------------------------------------------------------------------------- a17: dense optimizers
def dense_adagrad_step(param, grad, state_sum, lr, eps=1e-10, weight_decay=0.0, lr_decay=0.0, step=1):
    """nn/optim.cpp:114-145 (AdagradOptimizer::step)."""
    clr = lr / (1 + (step - 1) * lr_decay)
    g = grad
    if weight_decay != 0:
        g = g + weight_decay * param
    state_sum.addcmul_(g, g, value=1.0)
    std = state_sum.sqrt().add_(eps)
    param.addcdiv_(g, std, value=-clr)


def dense_adam_step(param, grad, exp_avg, exp_avg_sq, lr, num_steps, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, max_exp_avg_sq=None):
    """nn/optim.cpp:186-232 (AdamOptimizer::step), same tensor ops in the same order; num_steps = steps taken before this one."""
    import math

    import numpy as np

    bc1 = float(np.float32(1) - np.float32(np.power(np.float32(beta1), np.float32(num_steps + 1))))
    bc2 = float(np.float32(1) - np.float32(np.power(np.float32(beta2), np.float32(num_steps + 1))))
    g = grad
    if weight_decay != 0:
        g = g.add(param, alpha=weight_decay)
    # beta_1_ / beta_2_ are C++ floats in the reference: `1 - beta_2_` is evaluated in float (0.999f -> 0.00099998713), not in double
    omb1 = float(np.float32(1) - np.float32(beta1))
    omb2 = float(np.float32(1) - np.float32(beta2))
    exp_avg.mul_(beta1).add_(g, alpha=omb1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=omb2)
    if max_exp_avg_sq is not None:
        torch.max(max_exp_avg_sq, exp_avg_sq, out=max_exp_avg_sq)
        denom = (max_exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    else:
        denom = (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-(lr / bc1))


# ----------------------------------------------------------------------------- a4 (evaluation): global score filter
def sort_all_edges(all_edges):
    """data/graph.cpp:233-236 (MariusGraph::sortAllEdges)"""
    all_edges = all_edges.to(torch.int64)
    return (all_edges.index_select(0, all_edges[:, 0].argsort()), all_edges.index_select(0, all_edges[:, -1].argsort()))


def compute_filter_corruption_global(all_src_sorted, all_dst_sorted, edges, inverse):
    """data/samplers/negative.cpp:50-205, global branch of compute_filter_corruption_cpu, loop for loop: for every batch edge, every known
    edge with the same uncorrupted endpoint (and relation) contributes (edge_id, corrupted endpoint).  Returns int64 [F, 2]."""
    has_rel = edges.size(1) == 3
    if inverse:
        tup_id, corrupt_id, srt = (2 if has_rel else 1), 0, all_dst_sorted
    else:
        tup_id, corrupt_id, srt = 0, (2 if has_rel else 1), all_src_sorted
    nodes = edges[:, tup_id].contiguous()
    sorted_nodes = srt[:, tup_id].contiguous()
    starts = torch.searchsorted(sorted_nodes, nodes)
    ends = torch.searchsorted(sorted_nodes, nodes + 1)
    out = []
    e_l, s_l = edges.tolist(), srt.tolist()
    for edge_id in range(edges.size(0)):
        for cur in range(int(starts[edge_id]), int(ends[edge_id])):
            if (not has_rel) or s_l[cur][1] == e_l[edge_id][1]:
                out.append((edge_id, s_l[cur][corrupt_id]))
    return torch.tensor(out, dtype=torch.int64).reshape(-1, 2)


def apply_score_filter(scores, flt):
    """data/samplers/negative.cpp:306-311"""
    if flt is not None and flt.numel() > 0:
        scores.index_put_((flt[:, 0], flt[:, 1]), torch.tensor(-1e9, dtype=scores.dtype))
    return scores


# ----------------------------------------------------------------------------- a18: ranks
def compute_ranks(pos, neg):
    """reporting/reporting.cpp:55-57"""
    return (neg >= pos.unsqueeze(1)).sum(1) + 1


# ----------------------------------------------------------------------------- a6 / a16: storage
def index_read(table, ids):
    """storage/storage.cpp:606-649"""
    return table.index_select(0, ids)


def index_add(table, ids, values):
    """storage/storage.cpp:651-673 (ids unique)"""
    table.index_add_(0, ids, values)


# ----------------------------------------------------------------------------- a12/a14/a17: one train step on a batch
def apply_activation(activation, x):
    """nn/activation.cpp:7-21"""
    if activation == "RELU":
        return torch.relu(x)
    if activation == "SIGMOID":
        return torch.sigmoid(x)
    if activation == "NONE":
        return x
    raise RuntimeError("Unsupported activation function")


def post_hook(x, bias=None, activation="NONE"):
    """Layer::post_hook nn/layers/layer.cpp:9-16: `input + bias_` when config_->bias, then apply_activation — what GeneralEncoder::forward
    (nn/encoders/encoder.cpp:221-224) applies to the rows EmbeddingLayer::forward (embedding.cpp:17, a column view) hands on"""
    if bias is not None:
        x = x + bias
    return apply_activation(activation, x)


def train_batch(decoder, node_embeddings, node_state, edges, dst_neg_map, src_neg_map, relations, inverse_relations,
                dst_filter=None, src_filter=None, reduction="sum", sparse_lr=0.1, loss="SOFTMAX_CE", margin=0.1, encoder_bias=None, encoder_activation="NONE"):
    """nn/model.cpp:290-333 (train_batch) + :252-288 (forward_lp) on batch-local tensors.  encoder_bias / encoder_activation: the embedding
    layer's post-hook (model.cpp:253 encoder_->forward; default: none).

    Returns dict with scores, loss, node grad [U,d], relation grads, dw, ds (and bias_grad when a bias is given).
    """
    emb = node_embeddings.clone().requires_grad_(True)
    rel = relations.clone().requires_grad_(True) if relations is not None else None
    inv = inverse_relations.clone().requires_grad_(True) if inverse_relations is not None else None
    bias = encoder_bias.clone().requires_grad_(True) if encoder_bias is not None else None
    encoded = post_hook(emb, bias, encoder_activation)
    pos, neg, inv_pos, inv_neg = node_corrupt_forward(decoder, edges, encoded, dst_neg_map, src_neg_map, rel, inv)
    neg = apply_score_filter(neg, dst_filter)
    if inv_neg is not None:
        inv_neg = apply_score_filter(inv_neg, src_filter)
        rhs = loss_function(loss, pos, neg, reduction, margin)
        lhs = loss_function(loss, inv_pos, inv_neg, reduction, margin)
        loss = lhs + rhs
    else:
        loss = loss_function(loss, pos, neg, reduction, margin)
    loss.backward()
    state = node_state.clone()
    dw, ds = accumulate_gradients(emb.grad, state, sparse_lr)
    return {
        "pos": pos.detach(), "neg": neg.detach(),
        "inv_pos": None if inv_pos is None else inv_pos.detach(),
        "inv_neg": None if inv_neg is None else inv_neg.detach(),
        "loss": loss.detach(), "node_grad": emb.grad, "rel_grad": None if rel is None else rel.grad,
        "inv_rel_grad": None if inv is None else inv.grad, "dw": dw, "ds": ds, "bias_grad": None if bias is None else bias.grad,
    }
