"""ORACLE (test infrastructure; only tests/ import this): DENSE multi-hop neighbour sampling and the GraphSage aggregation of the reference,
restated with the SAME ATen calls in the same order as the reference's device branch (cfg4: ogbn-papers100M GraphSage — SURVEY.md §8 (f.4) tail).

  MariusGraph             src/cpp/src/data/graph.cpp:16-44        CSR offsets / degrees of the src-sorted and dst-sorted edge lists
  neighbors_for_node_ids  src/cpp/src/data/graph.cpp:128-236      degrees + offsets of the requested nodes, then the one-hop sampler
  sample_all              src/cpp/src/data/samplers/neighbor.cpp:9-17
  sample_uniform          src/cpp/src/data/samplers/neighbor.cpp:81-105   (`rand_samples` = torch::randint(max_id, [total]) of the reference)
  layered_neighbors       src/cpp/src/data/samplers/neighbor.cpp:402-582  (device branch: bitmap `hash_map`, nonzero() -> ascending delta ids)
  perform_map             src/cpp/src/data/graph.cpp:361-398
  prepare_for_next_layer  src/cpp/src/data/graph.cpp:290-325
  graph_sage_forward      src/cpp/src/nn/layers/gnn/graph_sage_layer.cpp:37-96 with layer_helpers.cpp:11-30 (segmented sum = index_add_)

Pinning: the reference holds NO known-answer vectors for this path (its tests check shapes and `unique_node_indices == dense_graph.node_ids`,
test/python/bindings/integration/test_data.py:182-303), and its sources cannot be compiled here (spdlog).  What pins this file is that every
function is the reference's own ATen op sequence, call for call, plus the structural properties its tests and the algorithm imply
(tests/test_oracle_cpu.py: every sampled edge is an edge of the graph, ALL returns every neighbour, node_ids are unique and hop-ordered,
mappings point at the right ids).  The one substitution: the reference draws `rand_samples` from the DEVICE generator inside
sample_uniform_gpu; here (and in the HIP path) the caller passes the tensor, so both sides consume the same draw.  PARITY: pinned by
construction and properties, not by reference-held vectors."""
import torch


class MariusGraph:
    """graph.cpp:16-44 (the two-sorted-lists constructor) and :46-52 (from an edge list: argsort is NOT stable in the reference either —
    callers that compare neighbour ORDER must pass pre-sorted lists, as the tests here do)"""

    def __init__(self, src_sorted_edges, dst_sorted_edges, num_nodes_in_memory):
        self.num_nodes_in_memory = num_nodes_in_memory
        self.src_sorted_edges, self.dst_sorted_edges = src_sorted_edges, dst_sorted_edges
        contiguous_src = src_sorted_edges.select(1, 0).contiguous()
        contiguous_dst = dst_sorted_edges.select(1, -1).contiguous()
        arange = torch.arange(0, num_nodes_in_memory)
        self.out_offsets = torch.searchsorted(contiguous_src, arange)
        end = torch.tensor([contiguous_src.size(0)])
        self.out_num_neighbors = torch.cat([self.out_offsets, end]).narrow(0, 1, self.out_offsets.size(0)) - self.out_offsets
        self.in_offsets = torch.searchsorted(contiguous_dst, arange)
        end = torch.tensor([contiguous_dst.size(0)])
        self.in_num_neighbors = torch.cat([self.in_offsets, end]).narrow(0, 1, self.in_offsets.size(0)) - self.in_offsets
        self.max_out_num_neighbors = int(torch.max(self.out_num_neighbors))
        self.max_in_num_neighbors = int(torch.max(self.in_num_neighbors))

    @staticmethod
    def from_edges(edges, num_nodes):
        src_sorted = edges.index_select(0, edges.select(1, 0).argsort(stable=True))
        dst_sorted = edges.index_select(0, edges.select(1, -1).argsort(stable=True))
        return MariusGraph(src_sorted, dst_sorted, num_nodes)


def sample_all(edges, global_offsets, local_offsets, num_neighbors):
    """neighbor.cpp:9-17"""
    repeated_starts = global_offsets.repeat_interleave(num_neighbors)
    repeated_offsets = local_offsets.repeat_interleave(num_neighbors)
    arange = torch.arange(repeated_offsets.size(0))
    sorted_list_idx = repeated_starts + arange - repeated_offsets
    return edges.index_select(0, sorted_list_idx), local_offsets


def uniform_total(num_neighbors, max_neighbors):
    """how many samples sample_uniform draws (the size of its randint): sum of the capped degrees"""
    return int(num_neighbors.clamp(max=max_neighbors).sum())


def sample_uniform(edges, global_offsets, local_offsets, num_neighbors, max_neighbors, rand_samples):
    """neighbor.cpp:81-105; rand_samples: the reference's torch::randint(max_id, [total])"""
    mask = num_neighbors > max_neighbors
    capped = num_neighbors.masked_fill(mask, max_neighbors)
    local_offsets = capped.cumsum(0) - capped
    repeated_starts = global_offsets.repeat_interleave(capped)
    repeated_offsets = local_offsets.repeat_interleave(capped)
    arange = torch.arange(repeated_offsets.size(0))
    ranged = repeated_starts + arange - repeated_offsets
    repeated_num = num_neighbors.repeat_interleave(capped)
    rs = rand_samples.clone()
    rs.fmod_(repeated_num)
    sampled = repeated_starts + rs
    mask = mask.repeat_interleave(capped)
    idx = torch.where(mask, sampled, ranged)
    return edges.index_select(0, idx), local_offsets


def sample_dropout(edges, global_offsets, local_offsets, num_neighbors, rate, keep_rand):
    """neighbor.cpp:236-253; keep_rand: the reference's torch::rand(total)"""
    repeated_starts = global_offsets.repeat_interleave(num_neighbors)
    repeated_offsets = local_offsets.repeat_interleave(num_neighbors)
    arange = torch.arange(repeated_offsets.size(0))
    sorted_list_idx = repeated_starts + arange - repeated_offsets
    keep_mask = torch.ge(keep_rand, rate)
    sorted_list_idx = sorted_list_idx.masked_select(keep_mask)
    capped = segmented_sum_with_offsets(keep_mask.to(torch.int64), local_offsets)
    summed = capped.cumsum(0)
    local_offsets = summed - capped
    return edges.index_select(0, sorted_list_idx), local_offsets


def neighbors_for_node_ids(graph, node_ids, incoming, max_neighbors=-1, rand_samples=None, dropout_rate=None, keep_rand=None):
    """graph.cpp:128-236, device branch; max_neighbors < 0: NeighborSamplingLayer::ALL, else UNIFORM; dropout_rate: DROPOUT"""
    if incoming:
        num_neighbors = graph.in_num_neighbors.index_select(0, node_ids)
        global_offsets = graph.in_offsets.index_select(0, node_ids)
        edges = graph.dst_sorted_edges
    else:
        num_neighbors = graph.out_num_neighbors.index_select(0, node_ids)
        global_offsets = graph.out_offsets.index_select(0, node_ids)
        edges = graph.src_sorted_edges
    summed = num_neighbors.cumsum(0)
    local_offsets = summed - num_neighbors
    if dropout_rate is not None:
        return sample_dropout(edges, global_offsets, local_offsets, num_neighbors, dropout_rate, keep_rand)
    if max_neighbors < 0:
        return sample_all(edges, global_offsets, local_offsets, num_neighbors)
    return sample_uniform(edges, global_offsets, local_offsets, num_neighbors, max_neighbors, rand_samples)


class DENSEGraph:
    pass


def layered_neighbors(graph, node_ids, fanouts, use_incoming=True, use_outgoing=False, rand=None):
    """LayeredNeighborSampler::getNeighbors (neighbor.cpp:402-582), device branch.  fanouts: one entry per layer, -1 = ALL, k >= 0 = UNIFORM, ("dropout", rate).
    rand(layer, incoming: bool, total) -> int64 [total] in [0, max degree): the randint of sample_uniform_gpu."""
    hop_offsets = torch.zeros(1, dtype=torch.int64)
    delta_ids = node_ids
    incoming_offsets = outgoing_offsets = None
    incoming_vec, outgoing_vec = [], []
    hash_map = torch.zeros(graph.num_nodes_in_memory, dtype=torch.bool)
    for i, fan in enumerate(fanouts):
        d_in_edges = d_in_offs = d_out_edges = d_out_offs = None
        if delta_ids.size(0) > 0:
            for incoming in ((True,) if use_incoming else ()) + ((False,) if use_outgoing else ()):
                tbl = graph.in_num_neighbors if incoming else graph.out_num_neighbors
                if isinstance(fan, tuple):  # ("dropout", rate): NeighborSamplingLayer::DROPOUT; rand(...) -> the torch::rand draw, float32 [total degree]
                    kr = rand(i, incoming, int(tbl.index_select(0, delta_ids).sum()))
                    res = neighbors_for_node_ids(graph, delta_ids, incoming, -1, None, fan[1], kr)
                else:
                    rs = rand(i, incoming, uniform_total(tbl.index_select(0, delta_ids), fan)) if fan >= 0 else None
                    res = neighbors_for_node_ids(graph, delta_ids, incoming, fan, rs)
                if incoming:
                    d_in_edges, d_in_offs = res
                else:
                    d_out_edges, d_out_offs = res
        if incoming_offsets is not None:
            if d_in_offs is not None and d_in_offs.size(0) > 0:
                incoming_offsets = incoming_offsets + d_in_edges.size(0)
                incoming_offsets = torch.cat([d_in_offs, incoming_offsets], 0)
        else:
            incoming_offsets = d_in_offs
        if d_in_edges is not None and d_in_edges.size(0) > 0:
            incoming_vec.insert(0, d_in_edges)
        if outgoing_offsets is not None:
            if d_out_offs is not None and d_out_offs.size(0) > 0:
                outgoing_offsets = outgoing_offsets + d_out_edges.size(0)
                outgoing_offsets = torch.cat([d_out_offs, outgoing_offsets], 0)
        else:
            outgoing_offsets = d_out_offs
        if d_out_edges is not None and d_out_edges.size(0) > 0:
            outgoing_vec.insert(0, d_out_edges)
        # delta ids: the device branch (neighbor.cpp:515-529)
        if i > 0:
            hash_map = hash_map & False
        if d_in_edges is not None and d_in_edges.size(0) > 0:
            hash_map.index_fill_(0, d_in_edges.select(1, 0), True)
        if d_out_edges is not None and d_out_edges.size(0) > 0:
            hash_map.index_fill_(0, d_out_edges.select(1, -1), True)
        hash_map.index_fill_(0, node_ids, False)
        delta_ids = hash_map.nonzero().flatten(0, 1)
        hop_offsets = hop_offsets + delta_ids.size(0)
        hop_offsets = torch.cat([torch.zeros(1, dtype=torch.int64), hop_offsets])
        if delta_ids.size(0) > 0:
            node_ids = torch.cat([delta_ids, node_ids], 0)
    hop_offsets = torch.cat([hop_offsets, torch.tensor([node_ids.size(0)])])
    g = DENSEGraph()
    g.hop_offsets, g.node_ids = hop_offsets, node_ids
    g.in_offsets, g.in_neighbors_vec = incoming_offsets, incoming_vec
    g.out_offsets, g.out_neighbors_vec = outgoing_offsets, outgoing_vec
    g.num_nodes_in_memory = graph.num_nodes_in_memory
    g.in_neighbors_mapping = g.out_neighbors_mapping = None
    return g


def perform_map(g):
    """DENSEGraph::performMap (graph.cpp:361-398)"""
    m = torch.zeros(g.num_nodes_in_memory, dtype=torch.int64)
    m.index_copy_(0, g.node_ids, torch.arange(g.node_ids.size(0)))
    if len(g.out_neighbors_vec) > 0:
        g.src_sorted_edges = torch.cat(g.out_neighbors_vec, 0)
        g.out_neighbors_mapping = m.gather(0, g.src_sorted_edges.select(1, -1))
        tmp = torch.cat([g.out_offsets, torch.tensor([g.src_sorted_edges.size(0)])])
        g.out_num_neighbors = tmp.narrow(0, 1, g.out_offsets.size(0)) - tmp.narrow(0, 0, g.out_offsets.size(0))
    else:
        g.src_sorted_edges = torch.zeros(0, 2, dtype=torch.int64)
        g.out_num_neighbors = torch.zeros(g.node_ids.size(0), dtype=torch.int64)
    if len(g.in_neighbors_vec) > 0:
        g.dst_sorted_edges = torch.cat(g.in_neighbors_vec, 0)
        g.in_neighbors_mapping = m.gather(0, g.dst_sorted_edges.select(1, 0))
        tmp = torch.cat([g.in_offsets, torch.tensor([g.dst_sorted_edges.size(0)])])
        g.in_num_neighbors = tmp.narrow(0, 1, g.in_offsets.size(0)) - tmp.narrow(0, 0, g.in_offsets.size(0))
    else:
        g.dst_sorted_edges = torch.zeros(0, 2, dtype=torch.int64)
        g.in_num_neighbors = torch.zeros(g.node_ids.size(0), dtype=torch.int64)
    return g


def prepare_for_next_layer(g):
    """DENSEGraph::prepareForNextLayer (graph.cpp:290-325)"""
    num_nodes_to_remove = int(g.hop_offsets[1] - g.hop_offsets[0])
    num_finished_nodes = int(g.hop_offsets[2] - g.hop_offsets[1])
    if g.src_sorted_edges.size(0) > 0:
        if num_finished_nodes == g.out_offsets.size(0):
            return g
        fin = int(g.out_offsets[num_finished_nodes])
        g.src_sorted_edges = g.src_sorted_edges.narrow(0, fin, g.src_sorted_edges.size(0) - fin)
        g.out_neighbors_mapping = g.out_neighbors_mapping.narrow(0, fin, g.out_neighbors_mapping.size(0) - fin) - num_nodes_to_remove
        g.out_offsets = g.out_offsets.narrow(0, num_finished_nodes, g.out_offsets.size(0) - num_finished_nodes) - fin
    g.out_num_neighbors = g.out_num_neighbors.narrow(0, num_finished_nodes, g.out_num_neighbors.size(0) - num_finished_nodes)
    if g.dst_sorted_edges.size(0) > 0:
        if num_finished_nodes == g.in_offsets.size(0):
            return g
        fin = int(g.in_offsets[num_finished_nodes])
        g.dst_sorted_edges = g.dst_sorted_edges.narrow(0, fin, g.dst_sorted_edges.size(0) - fin)
        g.in_neighbors_mapping = g.in_neighbors_mapping.narrow(0, fin, g.in_neighbors_mapping.size(0) - fin) - num_nodes_to_remove
        g.in_offsets = g.in_offsets.narrow(0, num_finished_nodes, g.in_offsets.size(0) - num_finished_nodes) - fin
    g.in_num_neighbors = g.in_num_neighbors.narrow(0, num_finished_nodes, g.in_num_neighbors.size(0) - num_finished_nodes)
    g.node_ids = g.node_ids.narrow(0, num_nodes_to_remove, g.node_ids.size(0) - num_nodes_to_remove)
    g.hop_offsets = g.hop_offsets.narrow(0, 1, g.hop_offsets.size(0) - 1) - num_nodes_to_remove
    return g


def segment_ids_from_offsets(segment_offsets, input_size):
    """layer_helpers.cpp:11-17"""
    segment_ids = torch.zeros(input_size + 1, dtype=segment_offsets.dtype)
    segment_ids.index_add_(0, segment_offsets, torch.ones(segment_offsets.size(0), dtype=segment_offsets.dtype))
    segment_ids = segment_ids.cumsum(0) - 1
    return segment_ids.narrow(0, 0, segment_ids.size(0) - 1)


def segmented_sum_with_offsets(tensor, segment_offsets):
    """layer_helpers.cpp:19-30: zeros + index_add_ (CPU: rows are added in index order)"""
    segment_ids = segment_ids_from_offsets(segment_offsets, tensor.size(0))
    out = torch.zeros((segment_offsets.size(0),) + tuple(tensor.shape[1:]), dtype=tensor.dtype)
    out.index_add_(0, segment_ids, tensor)
    return out


def graph_sage_aggregate(inputs, g, aggregator="MEAN"):
    """the part of GraphSageLayer::forward (graph_sage_layer.cpp:37-96) in front of the matmuls: a_i (MEAN: the neighbours' mean; GCN: mean over
    the neighbours and the node itself) and the self rows"""
    total = a_i = None
    if g.out_neighbors_mapping is not None:
        total = g.out_num_neighbors
        a_i = segmented_sum_with_offsets(inputs.index_select(0, g.out_neighbors_mapping), g.out_offsets)
    if g.in_neighbors_mapping is not None:
        total = g.in_num_neighbors if total is None else total + g.in_num_neighbors
        s = segmented_sum_with_offsets(inputs.index_select(0, g.in_neighbors_mapping), g.in_offsets)
        a_i = s if a_i is None else a_i + s
    layer_offset = int(g.hop_offsets[1])
    self_embs = inputs.narrow(0, layer_offset, inputs.size(0) - layer_offset)
    if aggregator == "GCN":
        a_i = a_i + self_embs
        a_i = a_i / (total + 1).unsqueeze(-1)
    elif aggregator == "MEAN":
        if total is not None:
            denominator = torch.where(torch.not_equal(total, 0), total, 1).to(a_i.dtype).unsqueeze(-1)
            a_i = a_i / denominator
    else:
        raise RuntimeError("Unrecognized aggregator")
    return a_i, self_embs


def graph_sage_forward(inputs, g, w1, w2=None, bias=None, aggregator="MEAN"):
    """GraphSageLayer::forward (graph_sage_layer.cpp:37-96); bias / activation: Layer::post_hook (layer.cpp:9-16) is the caller's"""
    a_i, self_embs = graph_sage_aggregate(inputs, g, aggregator)
    if aggregator == "GCN":
        out = torch.matmul(w1, a_i.transpose(0, -1)).transpose(0, -1)
    elif a_i is not None:
        out = (torch.matmul(w1, self_embs.transpose(0, -1)) + torch.matmul(w2, a_i.transpose(0, -1))).transpose(0, -1)
    else:
        out = torch.matmul(w1, self_embs.transpose(0, -1)).transpose(0, -1)
    if bias is not None:
        out = out + bias
    return out


def post_hook(x, bias, activation):
    """Layer::post_hook (src/cpp/src/nn/layers/layer.cpp:9-16): x + bias, then apply_activation (src/cpp/src/nn/activation.cpp:7-21)"""
    if bias is not None:
        x = x + bias
    if activation == "RELU":
        x = torch.relu(x)
    elif activation == "SIGMOID":
        x = torch.sigmoid(x)
    return x


def encoder_forward(features, g, layers):
    """GeneralEncoder::forward (src/cpp/src/nn/encoders/encoder.cpp:195-257) for the cfg4 shape: a FEATURE stage followed by GraphSage stages —
    layers: [(w1, w2, bias, aggregator, activation)]; performMap first, prepareForNextLayer between GNN stages (not after the last)."""
    perform_map(g)
    out = features.narrow(0, 0, features.size(0))  # FeatureLayer::forward is the identity on the batch's feature rows
    for i, (w1, w2, bias, aggregator, activation) in enumerate(layers):
        out = graph_sage_forward(out, g, w1, w2, None, aggregator)
        out = post_hook(out, bias, activation)
        if i < len(layers) - 1:
            prepare_for_next_layer(g)
    return out


def node_classification_step(features, g, layers, labels, lr):
    """Model::train_batch, NODE_CLASSIFICATION branch (src/cpp/src/nn/model.cpp:317-328): y_pred = encoder output of the batch's target nodes,
    loss = cross_entropy(y_pred, labels) (CrossEntropyLoss, src/cpp/src/nn/loss.cpp:88-102, reduction sum), backward, dense Adagrad step of every
    layer parameter (optim.cpp:59-110: sum += g^2; w -= lr g / (sqrt(sum) + 1e-10)).  Returns (loss, y_pred); parameters are updated in place,
    their Adagrad sums live in `.adagrad_sum` attributes."""
    params = [p for layer in layers for p in layer[:3] if p is not None]
    for p in params:
        p.grad = None
    y = encoder_forward(features, g, layers)
    loss = torch.nn.functional.cross_entropy(y, labels, reduction="sum")
    loss.backward()
    with torch.no_grad():
        for p in params:
            if not hasattr(p, "adagrad_sum"):
                p.adagrad_sum = torch.zeros_like(p)
            p.adagrad_sum.addcmul_(p.grad, p.grad, value=1.0)
            p.addcdiv_(p.grad, p.adagrad_sum.sqrt().add_(1e-10), value=-lr)
    return loss.detach(), y.detach()
