"""(File name: sorts LAST among the GPU test files on purpose.  Everything in here was written after the pool closed GPU use for this repository and
has never run on hardware — DESIGN.md §10 — so that `pytest -x -m gpu` reaches it only after every test that HAS been verified on an MI355X.)

cfg4 slice (round 6): the HIP neighbour-sampling / GraphSage-aggregation path (neighbor.hip through marius_amd/gnn.py) against the oracle
(oracle/neighbor_oracle.py: the reference's own ATen op sequence — neighbor.cpp:9-105, 402-582; graph.cpp:16-44, 128-236, 290-398;
graph_sage_layer.cpp:37-96).  Integer outputs bit-exact; the aggregation's float sums bit-exact against the CPU op sequence (rows are added in
index order, as the CPU index_add_ does); the layer's output (two library GEMMs) and the backward within 1e-5 / 1e-6 relative."""
import math

import pytest
import torch

from test_gpu_flash import DEC as _FLASH_DEC, make_batch as _flash_make_batch

from oracle import neighbor_oracle as NO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    from marius_amd import gnn

    return gnn


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def make_graph(num_nodes, E, cols, seed, hubs=True):
    g = torch.Generator().manual_seed(seed)
    src, dst = torch.randint(num_nodes, (E,), generator=g), torch.randint(num_nodes, (E,), generator=g)
    if hubs and E > 10:
        dst[torch.rand(E, generator=g) < 0.2] = 3 % num_nodes
        src[torch.rand(E, generator=g) < 0.1] = 5 % num_nodes
    if num_nodes > 2:  # one node without any edge
        src[src == num_nodes - 1] = 0
        dst[dst == num_nodes - 1] = 1
    edges = torch.stack([src, torch.randint(7, (E,), generator=g), dst], 1) if cols == 3 else torch.stack([src, dst], 1)
    return NO.MariusGraph.from_edges(edges, num_nodes)


def to_device(G, og, dev):
    return G.MariusGraph(og.src_sorted_edges.to(dev), og.dst_sorted_edges.to(dev), og.num_nodes_in_memory)


@pytest.mark.parametrize("num_nodes,E,cols", [(200, 3000, 2), (200, 3000, 3), (5000, 200000, 2), (3, 4, 3), (70000, 70000, 2)])
@pytest.mark.parametrize("max_neighbors", [-1, 0, 1, 7, 100000])
def test_one_hop_sampler_bit_exact(G, dev, num_nodes, E, cols, max_neighbors):
    """MariusGraph::getNeighborsForNodeIds (graph.cpp:128-236) with sample_all_gpu / sample_uniform_gpu (neighbor.cpp:9-17, 81-105): both directions,
    repeated ids, the node without edges, the hub; 5000 requested nodes cross the scan's tile boundary."""
    og = make_graph(num_nodes, E, cols, seed=num_nodes + cols)
    dg = to_device(G, og, dev)
    g = torch.Generator().manual_seed(max_neighbors + 7)
    for n in (1, 6, 1025, 5000):
        ids = torch.randint(num_nodes, (n,), generator=g)
        ids[0] = 3 % num_nodes
        ids[-1] = num_nodes - 1
        for incoming in (True, False):
            tbl = og.in_num_neighbors if incoming else og.out_num_neighbors
            rs = None
            if max_neighbors >= 0:
                total = NO.uniform_total(tbl.index_select(0, ids), max_neighbors)
                rs = torch.randint(max(int(tbl.max()), 1), (total,), generator=g)
            want, want_offs = NO.neighbors_for_node_ids(og, ids, incoming, max_neighbors, rs)
            got, got_offs = dg.getNeighborsForNodeIds(ids.to(dev), incoming, max_neighbors, None if rs is None else (lambda t, rs=rs: rs.to(dev)))
            assert torch.equal(got.cpu(), want) and torch.equal(got_offs.cpu(), want_offs)


@pytest.mark.parametrize("cols", [2, 3])
@pytest.mark.parametrize("rate", [0.0, 0.3, 0.9, 1.5])
def test_one_hop_dropout_sampler_bit_exact(G, dev, cols, rate):
    """sample_dropout_gpu (neighbor.cpp:236-253): the kept neighbours in masked_select's order and the nodes' new offsets, from the same torch::rand draw"""
    og = make_graph(800, 12000, cols, seed=31 + cols)
    dg = to_device(G, og, dev)
    g = torch.Generator().manual_seed(int(rate * 10) + 3)
    for n in (1, 7, 1500):
        ids = torch.randint(800, (n,), generator=g)
        ids[0], ids[-1] = 3, 799
        for incoming in (True, False):
            tbl = og.in_num_neighbors if incoming else og.out_num_neighbors
            kr = torch.rand(int(tbl.index_select(0, ids).sum()), generator=g)
            want, want_offs = NO.neighbors_for_node_ids(og, ids, incoming, -1, None, rate, kr)
            got, got_offs = dg.getNeighborsForNodeIds(ids.to(dev), incoming, -1, None, rate, lambda t, kr=kr: kr.to(dev))
            assert torch.equal(got.cpu(), want) and torch.equal(got_offs.cpu(), want_offs)


def test_one_hop_sampler_empty_request(G, dev):
    og = make_graph(50, 300, 2, seed=1)
    dg = to_device(G, og, dev)
    got, offs = dg.getNeighborsForNodeIds(torch.zeros(0, dtype=torch.int64, device=dev), True)
    assert got.shape == (0, 2) and offs.numel() == 0


@pytest.mark.parametrize("cols", [2, 3])
@pytest.mark.parametrize("fanouts,inc,out", [([-1], True, False), ([10, 5], True, False), ([15, 10, 5], True, True), ([3, -1, 2], False, True), ([0, 4], True, True)])
def test_layered_sampler_dense_graph_bit_exact(G, dev, cols, fanouts, inc, out):
    """LayeredNeighborSampler::getNeighbors (neighbor.cpp:402-582, device branch) + DENSEGraph::performMap + prepareForNextLayer (graph.cpp:290-398):
    node ids, hop offsets, neighbour offsets, every hop's edges, mappings and degrees, then the views of every following layer."""
    og = make_graph(3000, 40000, cols, seed=11 + cols)
    dgraph = to_device(G, og, dev)
    seeds = torch.randperm(3000, generator=torch.Generator().manual_seed(5))[:64]
    seeds[0], seeds[1] = 3, 2999
    draws = {}

    def rand_cpu(i, incoming, t):
        draws[(i, incoming)] = torch.randint(1 << 40, (t,), generator=torch.Generator().manual_seed(100 + 2 * i + int(incoming)))
        return draws[(i, incoming)]

    want = NO.layered_neighbors(og, seeds, fanouts, inc, out, rand=rand_cpu)
    sampler = G.LayeredNeighborSampler(dgraph, fanouts, inc, out)
    got = sampler.getNeighbors(seeds.to(dev), rand=lambda i, incoming, t: draws[(i, incoming)].to(dev))
    assert torch.equal(got.node_ids_.cpu(), want.node_ids) and torch.equal(got.hop_offsets_.cpu(), want.hop_offsets)
    for a, b in ((got.in_offsets_, want.in_offsets), (got.out_offsets_, want.out_offsets)):
        assert (a is None) == (b is None) and (a is None or torch.equal(a.cpu(), b))
    for a, b in ((got.in_neighbors_vec_, want.in_neighbors_vec), (got.out_neighbors_vec_, want.out_neighbors_vec)):
        assert len(a) == len(b) and all(torch.equal(x.cpu(), y) for x, y in zip(a, b))
    assert int(dgraph.marks_.max()) == 0  # the mark array is zero again after every hop
    NO.perform_map(want)
    got.performMap()
    for layer in range(len(fanouts)):
        for name in ("in_neighbors_mapping", "out_neighbors_mapping", "in_num_neighbors", "out_num_neighbors", "src_sorted_edges", "dst_sorted_edges", "node_ids", "hop_offsets",
                     "in_offsets", "out_offsets"):
            a, b = getattr(got, name + "_"), getattr(want, name)
            assert (a is None) == (b is None), name
            if a is not None:
                assert torch.equal(a.cpu(), b), (layer, name)
        if layer + 1 < len(fanouts):
            NO.prepare_for_next_layer(want)
            got.prepareForNextLayer()


@pytest.mark.parametrize("d", [1, 7, 64, 100, 128, 256, 512])
@pytest.mark.parametrize("aggregator,inc,out", [("MEAN", True, False), ("MEAN", True, True), ("GCN", True, False), ("GCN", True, True), ("MEAN", False, True)])
def test_graph_sage_aggregation_bit_exact_and_layer_close(G, dev, d, aggregator, inc, out):
    """GraphSageLayer::forward (graph_sage_layer.cpp:37-96): a_i bit-identical to the CPU op sequence (index_select + zeros + index_add_ + divide),
    the layer's output within 1e-5 (two library GEMMs on either side), gradients of inputs / w1 / w2 within 1e-5 of the CPU autograd."""
    og = make_graph(2000, 30000, 2, seed=d)
    dgraph = to_device(G, og, dev)
    seeds = torch.randperm(2000, generator=torch.Generator().manual_seed(d))[:50]
    seeds[0], seeds[1] = 3, 1999
    fan = [8, 4]
    draws = {}

    def rand_cpu(i, incoming, t):
        draws[(i, incoming)] = torch.randint(1 << 40, (t,), generator=torch.Generator().manual_seed(7 * i + int(incoming)))
        return draws[(i, incoming)]

    want_g = NO.perform_map(NO.layered_neighbors(og, seeds, fan, inc, out, rand=rand_cpu))
    got_g = G.LayeredNeighborSampler(dgraph, fan, inc, out).getNeighbors(seeds.to(dev), rand=lambda i, incoming, t: draws[(i, incoming)].to(dev))
    got_g.performMap()
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(want_g.node_ids.numel(), d, generator=gen)
    out_dim = 24
    layer = G.GraphSageLayer(d, out_dim, aggregator, bias=True, device=dev)
    with torch.no_grad():
        layer.bias.copy_(torch.randn(out_dim, generator=gen))
    xd = x.to(dev).requires_grad_(True)
    a_i, self_rows = layer.aggregate(xd, got_g)
    want_a, want_self = NO.graph_sage_aggregate(x, want_g, aggregator)
    assert torch.equal(a_i.detach().cpu(), want_a) and torch.equal(self_rows.detach().cpu(), want_self)  # bit for bit
    y = layer(xd, got_g)
    xc = x.clone().requires_grad_(True)
    w1, w2, b = layer.w1.detach().cpu().requires_grad_(True), None if layer.w2 is None else layer.w2.detach().cpu().requires_grad_(True), layer.bias.detach().cpu()
    yc = NO.graph_sage_forward(xc, want_g, w1, w2, b, aggregator)
    scale = float(yc.abs().max())
    assert float((y.detach().cpu() - yc.detach()).abs().max()) <= 1e-5 * scale
    gy = torch.randn(yc.shape, generator=gen)
    y.backward(gy.to(dev))
    yc.backward(gy)
    for got, want, what in ((xd.grad, xc.grad, "inputs"), (layer.w1.grad, w1.grad, "w1")) + (((layer.w2.grad, w2.grad, "w2"),) if w2 is not None else ()):
        err = float((got.cpu() - want).abs().max())
        assert err <= 1e-5 * float(want.abs().max()), (what, err)


def test_papers100m_shaped_hop_properties(G, dev):
    """Size-independent properties at a scale the oracle cannot walk: 20 M nodes, 200 M edges (a fifth of ogbn-papers100M's node count; the full
    graph's two sorted edge lists alone are 52 GB), 1000 seed nodes, fan-outs 15-10-5 over incoming edges.  Every sampled edge is an edge of the node that
    owns its segment; capped nodes hold exactly their cap; the batch's ids are unique; every hop's new ids ascend; mappings point at the right ids."""
    num_nodes, E = 20_000_000, 200_000_000
    g = torch.Generator(device=dev).manual_seed(0)
    src = torch.randint(num_nodes, (E,), device=dev, generator=g)
    dst = (torch.rand(E, device=dev, generator=g).pow(3) * num_nodes).long().clamp_(max=num_nodes - 1)  # skewed in-degrees
    order = torch.argsort(dst, stable=True)
    dst_sorted = torch.stack([src[order], dst[order]], 1)
    del order
    order = torch.argsort(src, stable=True)
    src_sorted = torch.stack([src[order], dst[order]], 1)
    del order, src, dst
    graph = G.MariusGraph(src_sorted, dst_sorted, num_nodes)
    seeds = torch.randperm(num_nodes, device=dev, generator=g)[:1000]
    fan = [15, 10, 5]
    dg = G.LayeredNeighborSampler(graph, fan, True, False).getNeighbors(seeds)
    ids = dg.node_ids_
    assert ids.unique().numel() == ids.numel() and torch.equal(ids[-1000:], seeds)
    ho = dg.hop_offsets_.tolist()
    for a, b in zip(ho[:-2], ho[1:-1]):
        assert bool((ids[a:b][1:] > ids[a:b][:-1]).all())
    dg.performMap()
    assert torch.equal(ids[dg.in_neighbors_mapping_], dg.dst_sorted_edges_[:, 0])
    owners = ids[ho[1]:]
    T = dg.dst_sorted_edges_.size(0)
    seg = torch.searchsorted(dg.in_offsets_, torch.arange(T, device=dev), right=True) - 1
    assert torch.equal(owners[seg], dg.dst_sorted_edges_[:, 1])                       # every sampled edge ends at the node that owns its segment
    deg = graph.in_num_neighbors_[owners]
    # hop h (from the seeds outwards) was sampled with fan[h]; the owners of hop h sit in [ho[-2-h], ho[-1-h])
    caps = torch.empty_like(deg)
    for h, f in enumerate(fan):
        lo, hi = ho[len(ho) - 2 - h] - ho[1], ho[len(ho) - 1 - h] - ho[1]
        caps[lo:hi] = f
    assert torch.equal(dg.in_num_neighbors_, torch.minimum(deg, caps))
    # and it is an edge of the graph: (src, dst) occurs in dst's slice of the sorted list
    k = torch.randint(T, (2000,), device=dev, generator=g)
    e = dg.dst_sorted_edges_[k]
    start, num = graph.in_offsets_[e[:, 1]], graph.in_num_neighbors_[e[:, 1]]
    for s, c, row in zip(start.tolist()[:200], num.tolist()[:200], e.tolist()[:200]):
        assert row[0] in graph.dst_sorted_edges_[s:s + c, 0].tolist()
    # aggregation at this size: mean of ones is one (or zero for a node without neighbours), whatever the order
    x = torch.ones(ids.numel(), 128, device=dev)
    layer = G.GraphSageLayer(128, 16, "MEAN", device=dev)
    a_i, _ = layer.aggregate(x, dg)
    has = (dg.in_num_neighbors_ > 0).float().unsqueeze(-1)
    assert torch.equal(a_i, has.expand_as(a_i))


@pytest.mark.parametrize("aggregator,inc,out", [("MEAN", True, False), ("GCN", True, True)])
def test_three_layer_encoder_and_node_classification_steps(G, dev, aggregator, inc, out):
    """cfg4's shape end to end: DENSE sample -> three GraphSage stages (aggregation: neighbor.hip; bias + RELU: encoder.hip's post-hook kernels) ->
    cross-entropy over the target nodes -> backward -> marius_dense_adagrad_step, three consecutive steps, against the oracle's restatement of
    GeneralEncoder::forward (encoder.cpp:195-257) and Model::train_batch's NODE_CLASSIFICATION branch (model.cpp:317-328)."""
    dims, classes = [100, 64, 32, 7], 7
    og = make_graph(4000, 60000, 2, seed=13)
    dgraph = to_device(G, og, dev)
    gen = torch.Generator().manual_seed(3)
    feats = torch.randn(4000, dims[0], generator=gen)
    labels_all = torch.randint(classes, (4000,), generator=gen)
    enc = G.GraphSageEncoder(dims, aggregator, "RELU", bias=True, device=dev)
    ref_layers = []
    for layer in enc.layers:
        with torch.no_grad():
            layer.bias.copy_(0.1 * torch.randn(layer.bias.shape, generator=gen))
        ref_layers.append((layer.w1.detach().cpu().clone().requires_grad_(True), None if layer.w2 is None else layer.w2.detach().cpu().clone().requires_grad_(True),
                           layer.bias.detach().cpu().clone().requires_grad_(True), aggregator, layer.activation))
    fan = [8, 5, 3]
    for step in range(3):
        seeds = torch.randperm(4000, generator=gen)[:100]
        draws = {}

        def rand_cpu(i, incoming, t):
            draws[(i, incoming)] = torch.randint(1 << 40, (t,), generator=torch.Generator().manual_seed(50 * step + 2 * i + int(incoming)))
            return draws[(i, incoming)]

        want_g = NO.layered_neighbors(og, seeds, fan, inc, out, rand=rand_cpu)
        got_g = G.LayeredNeighborSampler(dgraph, fan, inc, out).getNeighbors(seeds.to(dev), rand=lambda i, incoming, t: draws[(i, incoming)].to(dev))
        assert torch.equal(got_g.node_ids_.cpu(), want_g.node_ids)
        x = feats[want_g.node_ids]
        labels = labels_all[seeds]
        want_loss, want_y = NO.node_classification_step(x, want_g, ref_layers, labels, 0.1)
        got_loss, got_y = G.node_classification_step(enc, x.to(dev), got_g, labels.to(dev), 0.1)
        scale = float(want_y.abs().max())
        assert float((got_y.cpu() - want_y).abs().max()) <= 1e-4 * scale and abs(float(got_loss) - float(want_loss)) <= 1e-4 * abs(float(want_loss))
        for layer, (w1, w2, b, _a, _act) in zip(enc.layers, ref_layers):
            # (Adagrad from a zero sum moves a weight by lr sign(g): compare where the reference's accumulated g^2 is not rounding noise — tests/tolerance.py)
            ok = w1.adagrad_sum > 1e-10
            assert float((layer.w1.detach().cpu() - w1.detach())[ok].abs().max()) <= 2e-3
            assert float((layer.bias.detach().cpu() - b.detach()).abs().max()) <= 2e-3


# ------------------------------------------------------------------------------------------------ also unverified on hardware: the layout's record-layout check
@pytest.fixture(scope="module")
def H():
    from marius_amd import hip

    hip.lib()
    return hip


def test_layout_refuses_a_record_layout_changed_after_the_plan(H, dev, monkeypatch):
    """the record pitch (folded column tail: d = 36 / 68 / 100) and the column chunking follow MARIUS_FLASH_* switches that
    marius_config_reload() can change between marius_lp_plan and a launch that reuses its layout — the pack kernels would then write 464-byte
    records into buffers planned for 432-byte ones.  The plan records what it sized for (marius_lp_layout.flash_cfg) and every launch checks it."""
    decoder, B, C, N, d = "COMPLEX", 256, 4, 64, 100
    emb, edges, dst_neg, src_neg, rel, inv = _flash_make_batch(decoder, B, C, N, d, 1000, 7, seed=3)
    relop, cmp = _FLASH_DEC[decoder]
    W = H.LpWorkspace(relop, cmp, d, B, C, N, True, H.REDUCE_SUM, 3, True, dev, flags=H.LP_TRAIN_ONLY)
    assert W.layout.flash == 1 and W.layout.flash_cfg != 0
    t = lambda x: x.to(dev)  # noqa: E731
    W.bind(t(emb), t(edges), t(dst_neg), t(src_neg), t(rel), t(inv))
    W.forward()
    torch.cuda.synchronize()
    monkeypatch.setenv("MARIUS_FLASH_TAIL4", "0")
    H.reload_env()
    try:
        with pytest.raises(H.MariusHipError, match="record layout changed"):
            W.forward()
    finally:
        monkeypatch.delenv("MARIUS_FLASH_TAIL4")
        H.reload_env()
    W.forward()  # the planned layout is valid again
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ also new: the true-edge filter through the C-ABI directly
@pytest.mark.parametrize("cols", [2, 3])
@pytest.mark.parametrize("num_nodes,E,B", [(50, 400, 64), (1000, 20000, 500), (7, 30, 9)])
def test_true_edge_filter_equals_the_reference_loop(H, dev, cols, num_nodes, E, B):
    """marius_true_edge_filter_offsets / _emit against compute_filter_corruption's global branch (negative.cpp:50-205, oracle/lp_oracle.py: loop for loop),
    both corruption sides, duplicate edges in the known list and in the batch, batch edges that are not known edges.  (Until round 6 these entry
    points were only exercised through the C++ host's filtered evaluation.)"""
    from oracle import lp_oracle as O

    g = torch.Generator().manual_seed(num_nodes + cols)
    known = torch.stack([torch.randint(num_nodes, (E,), generator=g)] + ([torch.randint(3, (E,), generator=g)] if cols == 3 else []) + [torch.randint(num_nodes, (E,), generator=g)], 1)
    src_sorted = known[known[:, 0].argsort(stable=True)]
    dst_sorted = known[known[:, -1].argsort(stable=True)]
    batch = torch.cat([known[torch.randint(E, (B - 3,), generator=g)], torch.stack([torch.randint(num_nodes, (3,), generator=g) for _ in range(cols)], 1)])
    for inverse in (False, True):
        want = O.compute_filter_corruption_global(src_sorted, dst_sorted, batch, inverse)
        got = H.true_edge_filter((dst_sorted if inverse else src_sorted).to(dev), batch.to(dev), inverse)
        assert torch.equal(got.cpu(), want)
