"""cfg4, first slice (round 6): DENSE multi-hop neighbour sampling and the GraphSage layer on the HIP kernels of neighbor.hip, behind the
reference's names — MariusGraph (src/cpp/src/data/graph.cpp:16-44), LayeredNeighborSampler.getNeighbors (src/cpp/src/data/samplers/neighbor.cpp:402-582),
DENSEGraph.performMap / prepareForNextLayer (graph.cpp:290-398), GraphSageLayer.forward (src/cpp/src/nn/layers/gnn/graph_sage_layer.cpp:37-96).

Host side in Python for this slice (the C++ host module carries the link-prediction path; these classes are the next thing to move there):
what runs on the device per hop is marius_nbr_degrees -> marius_nbr_gather -> marius_nbr_delta_ids, then marius_nbr_positions once per batch and
marius_segment_gather_sum per layer; what stays ATen is the bookkeeping on [hops]-sized tensors, the randint draw (the reference's own call) and the
two matmuls of the layer (library GEMMs).  The two host reads per hop (total sampled edges, number of new ids) are the reference's own `.item()`s.
There is no CPU fallback: every method takes device tensors and goes through libmarius_hip.so."""
import torch

from . import hip as H


def _i64(dev):
    return dict(dtype=torch.int64, device=dev)


class MariusGraph:
    """graph.cpp:16-44: the src-sorted and dst-sorted edge lists, CSR offsets and degrees (ATen searchsorted, once per graph), plus the two persistent
    [num_nodes] scratch tables the O(batch) hop kernels work on (marks: zero between calls; positions: never needs clearing)"""

    def __init__(self, src_sorted_edges, dst_sorted_edges, num_nodes_in_memory):
        dev = src_sorted_edges.device
        H._dev(src_sorted_edges)
        self.num_nodes_in_memory_ = num_nodes_in_memory
        self.src_sorted_edges_, self.dst_sorted_edges_ = src_sorted_edges.contiguous(), dst_sorted_edges.contiguous()
        csrc = self.src_sorted_edges_.select(1, 0).contiguous()
        cdst = self.dst_sorted_edges_.select(1, -1).contiguous()
        ar = torch.arange(0, num_nodes_in_memory, device=dev)
        self.out_offsets_ = torch.searchsorted(csrc, ar)
        self.out_num_neighbors_ = torch.cat([self.out_offsets_, torch.tensor([csrc.size(0)], device=dev)]).narrow(0, 1, num_nodes_in_memory) - self.out_offsets_
        self.in_offsets_ = torch.searchsorted(cdst, ar)
        self.in_num_neighbors_ = torch.cat([self.in_offsets_, torch.tensor([cdst.size(0)], device=dev)]).narrow(0, 1, num_nodes_in_memory) - self.in_offsets_
        self.max_out_num_neighbors_ = int(self.out_num_neighbors_.max()) if num_nodes_in_memory else 0
        self.max_in_num_neighbors_ = int(self.in_num_neighbors_.max()) if num_nodes_in_memory else 0
        self.marks_ = torch.zeros(num_nodes_in_memory, dtype=torch.uint8, device=dev)
        self.positions_ = torch.empty(num_nodes_in_memory, **_i64(dev))

    def getNeighborsForNodeIds(self, node_ids, incoming, max_neighbors=-1, rand=None, dropout_rate=None, keep_rand=None):
        """graph.cpp:128-236 (ALL: max_neighbors < 0; UNIFORM otherwise).  rand(total) -> int64 [total] in [0, max degree): the draw of
        sample_uniform_gpu (neighbor.cpp:91); default: torch.randint on the device generator, as the reference.  Returns (edges, local_offsets)."""
        dev = node_ids.device
        n = node_ids.numel()
        tbl_num, tbl_off, edges, max_id = ((self.in_num_neighbors_, self.in_offsets_, self.dst_sorted_edges_, self.max_in_num_neighbors_) if incoming else
                                           (self.out_num_neighbors_, self.out_offsets_, self.src_sorted_edges_, self.max_out_num_neighbors_))
        num, goff, capped, loff = (torch.empty(n, **_i64(dev)) for _ in range(4))
        total_dev = torch.empty(1, **_i64(dev))
        wsb = H.lib().marius_nbr_workspace_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        H.check(H.lib().marius_nbr_degrees(H.ptr(node_ids), n, H.ptr(tbl_num), H.ptr(tbl_off), max_neighbors, H.ptr(num), H.ptr(goff), H.ptr(capped), H.ptr(loff),
                                           H.ptr(total_dev), H.ptr(ws), wsb, H.stream_ptr()), "nbr_degrees")
        total = int(total_dev.item())  # (the reference: summed_num_neighbors[-1].item())
        if dropout_rate is not None:  # NeighborSamplingLayer::DROPOUT (neighbor.cpp:236-253); keep_rand(total) -> float32 [total]: the reference's torch::rand
            assert max_neighbors < 0
            kr = keep_rand(total) if keep_rand is not None else torch.rand(total, device=dev)
            keep, scan, new_loff = torch.empty(max(total, 1), **_i64(dev)), torch.empty(max(total, 1), **_i64(dev)), torch.empty(n, **_i64(dev))
            kept_dev = torch.empty(1, **_i64(dev))
            wsb2 = H.lib().marius_nbr_workspace_bytes(total)
            ws2 = torch.empty(wsb2, dtype=torch.uint8, device=dev)
            H.check(H.lib().marius_nbr_dropout_offsets(H.ptr(loff), n, total, H.ptr(kr), float(dropout_rate), H.ptr(keep), H.ptr(scan), H.ptr(new_loff), H.ptr(kept_dev), H.ptr(ws2),
                                                       wsb2, H.stream_ptr()), "nbr_dropout_offsets")
            kept = int(kept_dev.item())
            out = torch.empty((kept, edges.size(1)), **_i64(dev))
            if kept > 0:
                H.check(H.lib().marius_nbr_dropout_emit(H.ptr(edges), edges.size(1), H.ptr(goff), H.ptr(loff), n, H.ptr(keep), H.ptr(scan), total, H.ptr(out), H.stream_ptr()),
                        "nbr_dropout_emit")
            return out, new_loff
        rs = None
        if max_neighbors >= 0 and total > 0:
            rs = rand(total) if rand is not None else torch.randint(max(max_id, 1), (total,), **_i64(dev))
        out = torch.empty((total, edges.size(1)), **_i64(dev))
        H.check(H.lib().marius_nbr_gather(H.ptr(edges), edges.size(1), H.ptr(num), H.ptr(goff), H.ptr(loff), H.ptr(capped), n, H.ptr(rs), total, H.ptr(out),
                                          H.stream_ptr()), "nbr_gather")
        return out, loff


class DENSEGraph:
    """graph.h DENSEGraph: hop_offsets_, node_ids_, in/out offsets, the per-hop edge lists and (after performMap) the batch-local mappings"""

    def __init__(self, hop_offsets, node_ids, in_offsets, in_neighbors_vec, out_offsets, out_neighbors_vec, graph):
        self.hop_offsets_, self.node_ids_ = hop_offsets, node_ids
        self.in_offsets_, self.in_neighbors_vec_ = in_offsets, in_neighbors_vec
        self.out_offsets_, self.out_neighbors_vec_ = out_offsets, out_neighbors_vec
        self.graph_ = graph
        self.in_neighbors_mapping_ = self.out_neighbors_mapping_ = None
        self.src_sorted_edges_ = self.dst_sorted_edges_ = None
        self.in_num_neighbors_ = self.out_num_neighbors_ = None

    def performMap(self):
        """graph.cpp:361-398"""
        dev = self.node_ids_.device
        n = self.node_ids_.numel()
        first = True

        def mapping(edges, col):
            nonlocal first
            out = torch.empty(edges.size(0), **_i64(dev))
            H.check(H.lib().marius_nbr_positions(H.ptr(self.node_ids_), n if first else 0, H.ptr(edges), edges.size(1), col, edges.size(0), H.ptr(self.graph_.positions_),
                                                 H.ptr(out), H.stream_ptr()), "nbr_positions")
            first = False
            return out

        if len(self.out_neighbors_vec_) > 0:
            self.src_sorted_edges_ = torch.cat(self.out_neighbors_vec_, 0)
            self.out_neighbors_mapping_ = mapping(self.src_sorted_edges_, self.src_sorted_edges_.size(1) - 1)
            self.out_neighbors_vec_ = []
            tmp = torch.cat([self.out_offsets_, torch.tensor([self.src_sorted_edges_.size(0)], device=dev)])
            self.out_num_neighbors_ = tmp.narrow(0, 1, self.out_offsets_.size(0)) - tmp.narrow(0, 0, self.out_offsets_.size(0))
        else:
            self.src_sorted_edges_ = torch.zeros((0, 2), **_i64(dev))
            self.out_num_neighbors_ = torch.zeros(n, **_i64(dev))
        if len(self.in_neighbors_vec_) > 0:
            self.dst_sorted_edges_ = torch.cat(self.in_neighbors_vec_, 0)
            self.in_neighbors_mapping_ = mapping(self.dst_sorted_edges_, 0)
            self.in_neighbors_vec_ = []
            tmp = torch.cat([self.in_offsets_, torch.tensor([self.dst_sorted_edges_.size(0)], device=dev)])
            self.in_num_neighbors_ = tmp.narrow(0, 1, self.in_offsets_.size(0)) - tmp.narrow(0, 0, self.in_offsets_.size(0))
        else:
            self.dst_sorted_edges_ = torch.zeros((0, 2), **_i64(dev))
            self.in_num_neighbors_ = torch.zeros(n, **_i64(dev))

    def getLayerOffset(self):
        return int(self.hop_offsets_[1])

    def prepareForNextLayer(self):
        """graph.cpp:290-325 (views only)"""
        rm = int(self.hop_offsets_[1] - self.hop_offsets_[0])
        fin_nodes = int(self.hop_offsets_[2] - self.hop_offsets_[1])
        if self.src_sorted_edges_.size(0) > 0:
            if fin_nodes == self.out_offsets_.size(0):
                return
            fin = int(self.out_offsets_[fin_nodes])
            self.src_sorted_edges_ = self.src_sorted_edges_.narrow(0, fin, self.src_sorted_edges_.size(0) - fin)
            self.out_neighbors_mapping_ = self.out_neighbors_mapping_.narrow(0, fin, self.out_neighbors_mapping_.size(0) - fin) - rm
            self.out_offsets_ = self.out_offsets_.narrow(0, fin_nodes, self.out_offsets_.size(0) - fin_nodes) - fin
        self.out_num_neighbors_ = self.out_num_neighbors_.narrow(0, fin_nodes, self.out_num_neighbors_.size(0) - fin_nodes)
        if self.dst_sorted_edges_.size(0) > 0:
            if fin_nodes == self.in_offsets_.size(0):
                return
            fin = int(self.in_offsets_[fin_nodes])
            self.dst_sorted_edges_ = self.dst_sorted_edges_.narrow(0, fin, self.dst_sorted_edges_.size(0) - fin)
            self.in_neighbors_mapping_ = self.in_neighbors_mapping_.narrow(0, fin, self.in_neighbors_mapping_.size(0) - fin) - rm
            self.in_offsets_ = self.in_offsets_.narrow(0, fin_nodes, self.in_offsets_.size(0) - fin_nodes) - fin
        self.in_num_neighbors_ = self.in_num_neighbors_.narrow(0, fin_nodes, self.in_num_neighbors_.size(0) - fin_nodes)
        self.node_ids_ = self.node_ids_.narrow(0, rm, self.node_ids_.size(0) - rm)
        self.hop_offsets_ = self.hop_offsets_.narrow(0, 1, self.hop_offsets_.size(0) - 1) - rm


class LayeredNeighborSampler:
    """neighbor.cpp:354-582.  num_neighbors: one entry per layer, -1 = all neighbours (NeighborSamplingLayer::ALL), k >= 0 = UNIFORM with max_neighbors k,
    ("dropout", rate) = DROPOUT"""

    def __init__(self, graph, num_neighbors, use_incoming_nbrs=True, use_outgoing_nbrs=False):
        self.graph_, self.num_neighbors_ = graph, list(num_neighbors)
        self.use_incoming_nbrs_, self.use_outgoing_nbrs_ = use_incoming_nbrs, use_outgoing_nbrs

    def getNeighbors(self, node_ids, rand=None):
        """rand(layer, incoming, total): the uniform sampler's draw (tests pass the oracle's); default: the device generator"""
        g = self.graph_
        dev = node_ids.device
        H._dev(node_ids)
        node_ids = node_ids.contiguous()
        hop_offsets = torch.zeros(1, **_i64(dev))
        delta_ids = node_ids
        in_offs = out_offs = None
        in_vec, out_vec = [], []
        for i, fan in enumerate(self.num_neighbors_):
            d_in = d_in_offs = d_out = d_out_offs = None
            if delta_ids.size(0) > 0:
                def hop(incoming):
                    draw = None if rand is None else (lambda t, i=i: rand(i, incoming, t))
                    if isinstance(fan, tuple):  # ("dropout", rate): NeighborSamplingLayer::DROPOUT
                        return g.getNeighborsForNodeIds(delta_ids, incoming, -1, None, fan[1], draw)
                    return g.getNeighborsForNodeIds(delta_ids, incoming, fan, draw)

                if self.use_incoming_nbrs_:
                    d_in, d_in_offs = hop(True)
                if self.use_outgoing_nbrs_:
                    d_out, d_out_offs = hop(False)
            if in_offs is not None:
                if d_in_offs is not None and d_in_offs.size(0) > 0:
                    in_offs = torch.cat([d_in_offs, in_offs + d_in.size(0)], 0)
            else:
                in_offs = d_in_offs
            if d_in is not None and d_in.size(0) > 0:
                in_vec.insert(0, d_in)
            if out_offs is not None:
                if d_out_offs is not None and d_out_offs.size(0) > 0:
                    out_offs = torch.cat([d_out_offs, out_offs + d_out.size(0)], 0)
            else:
                out_offs = d_out_offs
            if d_out is not None and d_out.size(0) > 0:
                out_vec.insert(0, d_out)
            # ---- the ids the next hop expands (neighbor.cpp:515-529): one O(batch) call instead of a num_nodes bitmap fill + nonzero
            n_in = 0 if d_in is None else d_in.size(0)
            n_out = 0 if d_out is None else d_out.size(0)
            n = n_in + n_out
            cols = (d_in if d_in is not None else d_out if d_out is not None else g.dst_sorted_edges_).size(1)
            um = H.UniqueMap(max(n, 1), dev)
            keys = torch.empty(max(n, 1), **_i64(dev))
            H.check(H.lib().marius_nbr_delta_ids(H.ptr(d_in) if n_in else None, n_in, H.ptr(d_out) if n_out else None, n_out, cols, H.ptr(node_ids), node_ids.numel(),
                                                 g.num_nodes_in_memory_, H.ptr(g.marks_), H.ptr(keys), H.ptr(um.uniq), H.ptr(um.inverse), H.ptr(um.perm), H.ptr(um.seg),
                                                 H.ptr(um.count), H.ptr(um.ws), um.ws_bytes, H.stream_ptr()), "nbr_delta_ids")
            U = int(um.count.item())  # (the reference: nonzero() synchronises here)
            delta_ids = um.uniq[:U].clone()
            hop_offsets = torch.cat([torch.zeros(1, **_i64(dev)), hop_offsets + U])
            if U > 0:
                node_ids = torch.cat([delta_ids, node_ids], 0)
        hop_offsets = torch.cat([hop_offsets, torch.tensor([node_ids.size(0)], device=dev)])
        return DENSEGraph(hop_offsets, node_ids, in_offs, in_vec, out_offs, out_vec, g)


class _Aggregate(torch.autograd.Function):
    """a_i of GraphSageLayer::forward: the segmented mean / GCN mean of gathered rows (marius_segment_gather_sum) and its backward (the same kernel
    over the occurrences of every input row, sorted by input row: marius_sort_unique's stable perm keeps the reference's index_add_ order)"""

    @staticmethod
    def forward(ctx, inputs, lists, n, mode, layer_offset):
        # lists: [(mapping, offsets, num_neighbors)] — outgoing then incoming, as the reference adds them
        dev = inputs.device
        d = inputs.size(1)
        out = torch.empty((n, d), dtype=torch.float32, device=dev)
        (ia, oa, da) = lists[0]
        (ib, ob, db) = lists[1] if len(lists) > 1 else (None, None, None)
        self_rows = inputs.narrow(0, layer_offset, n)
        H.check(H.lib().marius_segment_gather_sum(H.ptr(inputs), inputs.stride(0), d, H.ptr(ia), H.ptr(oa), ia.numel(), H.ptr(ib), H.ptr(ob), 0 if ib is None else ib.numel(), n,
                                                  None, H.ptr(da), H.ptr(db), mode, H.ptr(self_rows) if mode == 2 else None, inputs.stride(0), H.ptr(out), out.stride(0),
                                                  H.stream_ptr()), "segment_gather_sum")
        ctx.lists, ctx.n, ctx.mode, ctx.layer_offset, ctx.rows = lists, n, mode, layer_offset, inputs.size(0)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dev = grad_out.device
        n, d = ctx.n, grad_out.size(1)
        grad_out = grad_out.contiguous()
        total = ctx.lists[0][2] if len(ctx.lists) == 1 else ctx.lists[0][2] + ctx.lists[1][2]
        denom = (total + 1) if ctx.mode == 2 else torch.where(total != 0, total, torch.ones_like(total)) if ctx.mode == 1 else None
        grad_in = torch.zeros((ctx.rows, d), dtype=torch.float32, device=dev)
        for (idx, offs, _num) in ctx.lists:
            T = idx.numel()
            if T == 0:
                continue
            seg_id = torch.searchsorted(offs, torch.arange(T, device=dev), right=True) - 1  # owner of every gathered entry (ATen, [T])
            um = H.UniqueMap(T, dev).run(idx.contiguous(), max(1, (ctx.rows - 1).bit_length()))
            U = int(um.count.item())
            occ_seg = seg_id[um.perm[:T].long()].contiguous()  # segment of every occurrence, grouped by input row, occurrences in index order
            part = torch.empty((U, d), dtype=torch.float32, device=dev)
            starts = um.seg[:U].long().contiguous()
            H.check(H.lib().marius_segment_gather_sum(H.ptr(grad_out), grad_out.stride(0), d, H.ptr(occ_seg), H.ptr(starts), T, None, None, 0, U,
                                                      H.ptr(denom) if denom is not None else None, None, None, 0, None, 0, H.ptr(part), part.stride(0), H.stream_ptr()),
                    "segment_gather_sum(backward)")
            grad_in.index_add_(0, um.uniq[:U], part)  # unique rows: one add per row, no atomics race
        if ctx.mode == 2:  # GCN: the self rows are part of the mean
            grad_in.narrow(0, ctx.layer_offset, n).add_(grad_out / denom.unsqueeze(-1).to(grad_out.dtype))
        return grad_in, None, None, None, None


class _PostHook(torch.autograd.Function):
    """Layer::post_hook (layer.cpp:9-16) on the HIP kernels of encoder.hip: act(x + bias) and its backward (deterministic column sums for bias.grad)"""

    @staticmethod
    def forward(ctx, x, bias, activation):
        y = H.layer_post_hook(x.contiguous(), bias, activation)
        ctx.activation, ctx.with_bias = activation, bias is not None
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gx, bg = H.layer_post_hook_backward(gy.contiguous(), y, ctx.activation, with_bias=ctx.with_bias)
        return gx, bg, None


class GraphSageLayer(torch.nn.Module):
    """graph_sage_layer.cpp:10-96.  aggregator: "MEAN" (w1 self + w2 mean(neighbours)) or "GCN" (w1 mean(neighbours + self)); bias / activation:
    the layer's post-hook (layer.cpp:9-16; GeneralEncoder::forward applies it right after the layer, encoder.cpp:236-238)"""

    def __init__(self, input_dim, output_dim, aggregator="MEAN", bias=False, device="cuda:0", activation="NONE"):
        super().__init__()
        self.aggregator, self.activation = aggregator, activation
        glorot = lambda: torch.nn.init.xavier_uniform_(torch.empty(output_dim, input_dim, device=device))  # noqa: E731
        self.w1 = torch.nn.Parameter(glorot())
        self.w2 = torch.nn.Parameter(glorot()) if aggregator == "MEAN" else None
        self.bias = torch.nn.Parameter(torch.zeros(output_dim, device=device)) if bias else None

    def aggregate(self, inputs, dense_graph):
        lists = []
        if dense_graph.out_neighbors_mapping_ is not None:
            lists.append((dense_graph.out_neighbors_mapping_.contiguous(), dense_graph.out_offsets_.contiguous(), dense_graph.out_num_neighbors_.contiguous()))
        if dense_graph.in_neighbors_mapping_ is not None:
            lists.append((dense_graph.in_neighbors_mapping_.contiguous(), dense_graph.in_offsets_.contiguous(), dense_graph.in_num_neighbors_.contiguous()))
        layer_offset = dense_graph.getLayerOffset()
        n = inputs.size(0) - layer_offset
        if not lists:
            return None, inputs.narrow(0, layer_offset, n)
        a_i = _Aggregate.apply(inputs.contiguous(), lists, n, 2 if self.aggregator == "GCN" else 1, layer_offset)
        return a_i, inputs.narrow(0, layer_offset, n)

    def forward(self, inputs, dense_graph, train=True):
        a_i, self_embs = self.aggregate(inputs, dense_graph)
        if self.aggregator == "GCN":
            out = torch.matmul(self.w1, a_i.transpose(0, -1)).transpose(0, -1)
        elif a_i is not None:
            out = (torch.matmul(self.w1, self_embs.transpose(0, -1)) + torch.matmul(self.w2, a_i.transpose(0, -1))).transpose(0, -1)
        else:
            out = torch.matmul(self.w1, self_embs.transpose(0, -1)).transpose(0, -1)
        if self.bias is not None or self.activation != "NONE":
            out = _PostHook.apply(out, self.bias, self.activation)
        return out


class GraphSageEncoder(torch.nn.Module):
    """GeneralEncoder::forward (src/cpp/src/nn/encoders/encoder.cpp:195-257) for cfg4's shape: a FEATURE stage (the batch's feature rows as they are)
    followed by GraphSage stages; performMap first, prepareForNextLayer between GNN stages.  dims: [feature_dim, hidden ..., num_classes]."""

    def __init__(self, dims, aggregator="MEAN", activation="RELU", bias=True, device="cuda:0"):
        super().__init__()
        n = len(dims) - 1
        self.layers = torch.nn.ModuleList([GraphSageLayer(dims[i], dims[i + 1], aggregator, bias=bias, device=device, activation=activation if i < n - 1 else "NONE")
                                           for i in range(n)])

    def forward(self, features, dense_graph, train=True):
        dense_graph.performMap()
        out = features
        for i, layer in enumerate(self.layers):
            out = layer(out, dense_graph, train)
            if i < len(self.layers) - 1:
                dense_graph.prepareForNextLayer()
        return out


def node_classification_step(encoder, features, dense_graph, labels, lr, reduction="sum"):
    """Model::train_batch, NODE_CLASSIFICATION branch (src/cpp/src/nn/model.cpp:317-328): y_pred for the batch's target nodes, CrossEntropyLoss
    (src/cpp/src/nn/loss.cpp:88-102), backward, the dense Adagrad step of every layer parameter (marius_dense_adagrad_step: optim.cpp:114-145;
    the sums live in `.adagrad_sum` of the parameters).  Returns (loss, y_pred)."""
    for p in encoder.parameters():
        p.grad = None
    y = encoder(features, dense_graph, True)
    loss = torch.nn.functional.cross_entropy(y, labels.to(torch.int64), reduction=reduction)
    loss.backward()
    with torch.no_grad():
        for p in encoder.parameters():
            if p.grad is None:
                continue
            if not hasattr(p, "adagrad_sum"):
                p.adagrad_sum = torch.zeros_like(p)
            H.dense_adagrad_step(p.data, p.adagrad_sum, p.grad.contiguous(), lr)
    return loss.detach(), y.detach()
