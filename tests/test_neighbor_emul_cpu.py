"""CPU build of the cfg4 kernel file: marius_amd/csrc/kernels/neighbor.hip compiled by g++ against tests/emul/common.h (HIP's execution model on
host threads: a std::thread per work-item, barriers, wave shuffles through a per-wave buffer) and run through the SAME C-ABI entry points against
the oracle (oracle/neighbor_oracle.py).  This is test infrastructure — nothing under marius_amd/ can reach it, and it proves the kernels' logic
(indexing, scans, ordering), not their performance or the hipcc build; the `-m gpu` tests (tests/test_gpu_zz_unverified_cfg4.py) are the parity tests proper.
It exists because the round's GPU access ended before the cfg4 slice could run on hardware."""
import ctypes as C
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))

from oracle import neighbor_oracle as NO  # noqa: E402


@pytest.fixture(scope="module")
def E():
    import build_emul
    from marius_amd import hip

    lib = C.CDLL(build_emul.build(sanitize=os.environ.get("MARIUS_EMUL_SANITIZE") == "1"))
    for name in ("marius_nbr_workspace_bytes", "marius_nbr_degrees", "marius_nbr_gather", "marius_nbr_delta_ids", "marius_nbr_positions", "marius_segment_gather_sum",
                 "marius_sort_unique_workspace_bytes", "marius_nbr_dropout_offsets", "marius_nbr_dropout_emit"):
        res, args = hip.SIGNATURES[name]  # the same signature table the HIP library is bound with
        getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
    return lib


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def graph(num_nodes, Ecount, cols, seed):
    g = torch.Generator().manual_seed(seed)
    src, dst = torch.randint(num_nodes, (Ecount,), generator=g), torch.randint(num_nodes, (Ecount,), generator=g)
    dst[torch.rand(Ecount, generator=g) < 0.2] = 3
    src[src == num_nodes - 1] = 0
    dst[dst == num_nodes - 1] = 1
    edges = torch.stack([src, torch.randint(7, (Ecount,), generator=g), dst], 1) if cols == 3 else torch.stack([src, dst], 1)
    return NO.MariusGraph.from_edges(edges, num_nodes)


def one_hop(E, og, ids, incoming, max_neighbors, rs):
    n = ids.numel()
    tbl_num, tbl_off, edges = (og.in_num_neighbors, og.in_offsets, og.dst_sorted_edges) if incoming else (og.out_num_neighbors, og.out_offsets, og.src_sorted_edges)
    num, goff, capped, loff = (torch.full((n,), -7, dtype=torch.int64) for _ in range(4))
    total = torch.full((1,), -1, dtype=torch.int64)
    wsb = E.marius_nbr_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8)
    assert E.marius_nbr_degrees(P(ids), n, P(tbl_num), P(tbl_off), max_neighbors, P(num), P(goff), P(capped), P(loff), P(total), P(ws), wsb, None) == 0
    T = int(total)
    out = torch.full((T, edges.size(1)), -1, dtype=torch.int64)
    assert E.marius_nbr_gather(P(edges.contiguous()), edges.size(1), P(num), P(goff), P(loff), P(capped), n, P(rs), T, P(out), None) == 0
    return out, loff


@pytest.mark.parametrize("cols", [2, 3])
@pytest.mark.parametrize("max_neighbors", [-1, 0, 3, 1000])
def test_emulated_one_hop_sampler_equals_the_oracle(E, cols, max_neighbors):
    og = graph(300, 2500, cols, seed=cols)
    g = torch.Generator().manual_seed(5 + max_neighbors)
    for n in ((6, 1025) if os.environ.get("MARIUS_EMUL_SANITIZE") == "1" else (1, 6, 1025, 5000)):  # 1025 / 5000: more than one scan tile
        ids = torch.randint(300, (n,), generator=g)
        ids[0], ids[-1] = 3, 299
        for incoming in (True, False):
            tbl = og.in_num_neighbors if incoming else og.out_num_neighbors
            if max_neighbors < 0 and int(tbl.index_select(0, ids).sum()) > 30000:
                continue  # (every sampled edge is a host thread here)
            rs = None
            if max_neighbors >= 0:
                rs = torch.randint(max(int(tbl.max()), 1), (NO.uniform_total(tbl.index_select(0, ids), max_neighbors),), generator=g)
            want, want_offs = NO.neighbors_for_node_ids(og, ids, incoming, max_neighbors, rs)
            got, got_offs = one_hop(E, og, ids, incoming, max_neighbors, rs)
            assert torch.equal(got, want) and torch.equal(got_offs, want_offs)


def test_emulated_delta_ids_and_positions_equal_the_oracle(E):
    from marius_amd import hip

    og = graph(500, 4000, 2, seed=9)
    seeds = torch.tensor([7, 3, 150, 499, 42])
    rg = torch.Generator().manual_seed(4)
    want = NO.layered_neighbors(og, seeds, [4, -1], True, True, rand=lambda i, inc, t: torch.randint(1 << 30, (t,), generator=torch.Generator().manual_seed(10 * i + int(inc))))
    # replay hop by hop through the emulated entry points
    marks = torch.zeros(500, dtype=torch.uint8)
    node_ids, delta = seeds.clone(), seeds.clone()
    for i, fan in enumerate([4, -1]):
        d_in = d_out = None
        if delta.numel():
            def rs(inc):
                tbl = og.in_num_neighbors if inc else og.out_num_neighbors
                return None if fan < 0 else torch.randint(1 << 30, (NO.uniform_total(tbl.index_select(0, delta), fan),), generator=torch.Generator().manual_seed(10 * i + int(inc)))
            d_in, _ = one_hop(E, og, delta, True, fan, rs(True))
            d_out, _ = one_hop(E, og, delta, False, fan, rs(False))
        n_in, n_out = (0 if d_in is None else d_in.size(0)), (0 if d_out is None else d_out.size(0))
        n = n_in + n_out
        keys, uniq, inverse = (torch.empty(max(n, 1), dtype=torch.int64) for _ in range(3))
        perm, seg = torch.empty(max(n, 1), dtype=torch.int32), torch.empty(max(n, 1) + 1, dtype=torch.int32)
        count = torch.zeros(1, dtype=torch.int64)
        wsb = E.marius_sort_unique_workspace_bytes(max(n, 1))
        ws = torch.zeros(wsb, dtype=torch.uint8)  # (zero-filled: the sort's control block — include/marius_hip.h)
        assert E.marius_nbr_delta_ids(P(d_in), n_in, P(d_out), n_out, 2, P(node_ids), node_ids.numel(), 500, P(marks), P(keys), P(uniq), P(inverse), P(perm), P(seg), P(count),
                                      P(ws), wsb, None) == 0
        assert int(marks.max()) == 0
        delta = uniq[:int(count)].clone()
        if delta.numel():
            node_ids = torch.cat([delta, node_ids])
    assert torch.equal(node_ids, want.node_ids)
    NO.perform_map(want)
    table = torch.full((500,), -99, dtype=torch.int64)
    for edges, col, ref in ((want.dst_sorted_edges, 0, want.in_neighbors_mapping), (want.src_sorted_edges, 1, want.out_neighbors_mapping)):
        out = torch.empty(edges.size(0), dtype=torch.int64)
        assert E.marius_nbr_positions(P(want.node_ids), want.node_ids.numel(), P(edges.contiguous()), 2, col, edges.size(0), P(table), P(out), None) == 0
        assert torch.equal(out, ref)
    assert hip.SIGNATURES["marius_nbr_positions"][1] is not None


@pytest.mark.parametrize("d", [1, 7, 64, 100, 130])
@pytest.mark.parametrize("aggregator,inc,out", [("MEAN", True, False), ("MEAN", True, True), ("GCN", True, True)])
def test_emulated_aggregation_is_bit_identical_to_the_cpu_op_sequence(E, d, aggregator, inc, out):
    og = graph(400, 3000, 2, seed=d)
    seeds = torch.tensor([3, 399, 17, 250])
    dg = NO.perform_map(NO.layered_neighbors(og, seeds, [6, 3], inc, out, rand=lambda i, incoming, t: torch.randint(1 << 30, (t,), generator=torch.Generator().manual_seed(i))))
    x = torch.randn(dg.node_ids.numel(), d, generator=torch.Generator().manual_seed(1))
    want, self_rows = NO.graph_sage_aggregate(x, dg, aggregator)
    lists = []
    if dg.out_neighbors_mapping is not None:
        lists.append((dg.out_neighbors_mapping.contiguous(), dg.out_offsets.contiguous(), dg.out_num_neighbors.contiguous()))
    if dg.in_neighbors_mapping is not None:
        lists.append((dg.in_neighbors_mapping.contiguous(), dg.in_offsets.contiguous(), dg.in_num_neighbors.contiguous()))
    (ia, oa, da) = lists[0]
    (ib, ob, db) = lists[1] if len(lists) > 1 else (None, None, None)
    n = want.size(0)
    got = torch.full((n, d), float("nan"))
    self_c = self_rows.contiguous()
    assert E.marius_segment_gather_sum(P(x), x.stride(0), d, P(ia), P(oa), ia.numel(), P(ib), P(ob), 0 if ib is None else ib.numel(), n, None, P(da), P(db),
                                       2 if aggregator == "GCN" else 1, P(self_c) if aggregator == "GCN" else None, self_c.stride(0), P(got), got.stride(0), None) == 0
    assert torch.equal(got, want)
    # the backward form: occurrences of every input row, grouped by row in index order, each gathered gradient row divided by its segment's denominator
    gy = torch.randn(n, d, generator=torch.Generator().manual_seed(2))
    total = da if db is None else da + db
    denom = (total + 1) if aggregator == "GCN" else torch.where(total != 0, total, torch.ones_like(total))
    xr = x.clone().requires_grad_(True)
    a, _ = NO.graph_sage_aggregate(xr, dg, aggregator)
    a.backward(gy)
    grad = torch.zeros_like(x)
    for (idx, offs, _num) in lists:
        T = idx.numel()
        seg_id = torch.searchsorted(offs, torch.arange(T), right=True) - 1
        order = torch.argsort(idx, stable=True)
        uniq, counts = torch.unique_consecutive(idx[order], return_counts=True)
        starts = (counts.cumsum(0) - counts).contiguous()
        occ_seg = seg_id[order].contiguous()
        part = torch.empty(uniq.numel(), d)
        assert E.marius_segment_gather_sum(P(gy), gy.stride(0), d, P(occ_seg), P(starts), T, None, None, 0, uniq.numel(), P(denom), None, None, 0, None, 0, P(part), part.stride(0),
                                           None) == 0
        grad.index_add_(0, uniq, part)
    if aggregator == "GCN":
        lo = int(dg.hop_offsets[1])
        grad[lo:] += gy / denom.unsqueeze(-1)
    assert torch.allclose(grad, xr.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("rate", [0.0, 0.3, 0.9, 1.5])
def test_emulated_dropout_sampler_equals_the_oracle(G_emul, rate):
    import test_gpu_zz_unverified_cfg4 as TZ

    for cols in (2, 3):
        TZ.test_one_hop_dropout_sampler_bit_exact(G_emul, torch.device("cpu"), cols, rate)


@pytest.fixture()
def G_emul(E, monkeypatch):
    """marius_amd/gnn.py — the host-side mirror of MariusGraph / LayeredNeighborSampler / DENSEGraph / GraphSageLayer — driven on CPU tensors with the
    ctypes layer pointed at the emulated library FOR THIS TEST ONLY (monkeypatch): the product module itself has no such switch and refuses host tensors."""
    from marius_amd import gnn, hip

    for name in ("marius_sort_unique", "marius_sort_unique_workspace_bytes"):
        res, args = hip.SIGNATURES[name]
        getattr(E, name).restype, getattr(E, name).argtypes = res, args
    monkeypatch.setattr(hip, "lib", lambda: E)
    monkeypatch.setattr(hip, "_dev", lambda t: t)
    monkeypatch.setattr(hip, "stream_ptr", lambda stream=None: None)
    # the post-hook and the dense optimizer live in other kernel files (encoder.hip, rows.hip: verified on hardware in their own GPU tests): torch
    # stand-ins with the same contracts for this CPU run
    act = {"NONE": lambda t: t, "RELU": torch.relu, "SIGMOID": torch.sigmoid}
    monkeypatch.setattr(hip, "layer_post_hook", lambda x, bias=None, activation="NONE", out=None: act[activation](x if bias is None else x + bias))

    def hook_bwd(gy, y, activation="NONE", with_bias=True):
        gx = gy if activation == "NONE" else gy * ((y > 0).float() if activation == "RELU" else y * (1 - y))
        return gx, (gx.sum(0) if with_bias else None)

    monkeypatch.setattr(hip, "layer_post_hook_backward", hook_bwd)

    def adagrad(param, state_sum, grad, lr, eps=1e-10, weight_decay=0.0):
        state_sum.addcmul_(grad, grad, value=1.0)
        param.addcdiv_(grad, state_sum.sqrt().add_(eps), value=-lr)

    monkeypatch.setattr(hip, "dense_adagrad_step", adagrad)
    return gnn


@pytest.mark.parametrize("fanouts,inc,out", [([-1], True, False), ([5, 3], True, True), ([4, 0, 2], True, False), ([2, -1], False, True), ([("dropout", 0.5), 3], True, True)])
def test_emulated_host_mirror_layered_sampler_equals_the_oracle(G_emul, fanouts, inc, out):
    og = graph(600, 5000, 3, seed=21)
    dgraph = G_emul.MariusGraph(og.src_sorted_edges, og.dst_sorted_edges, og.num_nodes_in_memory)
    seeds = torch.randperm(600, generator=torch.Generator().manual_seed(5))[:16]
    seeds[0], seeds[1] = 3, 599
    draws = {}

    def rand_cpu(i, incoming, t):
        gen = torch.Generator().manual_seed(100 + 2 * i + int(incoming))
        draws[(i, incoming)] = torch.rand(t, generator=gen) if isinstance(fanouts[i], tuple) else torch.randint(1 << 40, (t,), generator=gen)
        return draws[(i, incoming)]

    want = NO.layered_neighbors(og, seeds, fanouts, inc, out, rand=rand_cpu)
    got = G_emul.LayeredNeighborSampler(dgraph, fanouts, inc, out).getNeighbors(seeds, rand=lambda i, incoming, t: draws[(i, incoming)])
    assert torch.equal(got.node_ids_, want.node_ids) and torch.equal(got.hop_offsets_, want.hop_offsets)
    for a, b in ((got.in_offsets_, want.in_offsets), (got.out_offsets_, want.out_offsets)):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
    for a, b in ((got.in_neighbors_vec_, want.in_neighbors_vec), (got.out_neighbors_vec_, want.out_neighbors_vec)):
        assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))
    assert int(dgraph.marks_.max()) == 0
    NO.perform_map(want)
    got.performMap()
    for layer in range(len(fanouts)):
        for name in ("in_neighbors_mapping", "out_neighbors_mapping", "in_num_neighbors", "out_num_neighbors", "src_sorted_edges", "dst_sorted_edges", "node_ids", "hop_offsets",
                     "in_offsets", "out_offsets"):
            a, b = getattr(got, name + "_"), getattr(want, name)
            assert (a is None) == (b is None), name
            if a is not None:
                assert torch.equal(a, b), (layer, name)
        if layer + 1 < len(fanouts):
            NO.prepare_for_next_layer(want)
            got.prepareForNextLayer()


@pytest.mark.parametrize("aggregator,inc,out", [("MEAN", True, False), ("MEAN", True, True), ("GCN", True, True)])
def test_emulated_host_mirror_graph_sage_layer_forward_backward(G_emul, aggregator, inc, out):
    d, out_dim = 20, 6
    og = graph(400, 3000, 2, seed=3)
    dgraph = G_emul.MariusGraph(og.src_sorted_edges, og.dst_sorted_edges, og.num_nodes_in_memory)
    seeds = torch.tensor([3, 399, 17, 250, 8])
    draws = {}

    def rand_cpu(i, incoming, t):
        draws[(i, incoming)] = torch.randint(1 << 40, (t,), generator=torch.Generator().manual_seed(7 * i + int(incoming)))
        return draws[(i, incoming)]

    want_g = NO.perform_map(NO.layered_neighbors(og, seeds, [5, 2], inc, out, rand=rand_cpu))
    got_g = G_emul.LayeredNeighborSampler(dgraph, [5, 2], inc, out).getNeighbors(seeds, rand=lambda i, incoming, t: draws[(i, incoming)])
    got_g.performMap()
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(want_g.node_ids.numel(), d, generator=gen)
    layer = G_emul.GraphSageLayer(d, out_dim, aggregator, bias=True, device="cpu")
    xd = x.clone().requires_grad_(True)
    y = layer(xd, got_g)
    xc = x.clone().requires_grad_(True)
    w1, w2, b = layer.w1.detach().clone().requires_grad_(True), None if layer.w2 is None else layer.w2.detach().clone().requires_grad_(True), layer.bias.detach()
    yc = NO.graph_sage_forward(xc, want_g, w1, w2, b, aggregator)
    assert torch.allclose(y, yc, rtol=1e-6, atol=1e-6)
    gy = torch.randn(yc.shape, generator=gen)
    y.backward(gy)
    yc.backward(gy)
    assert torch.allclose(xd.grad, xc.grad, rtol=1e-5, atol=1e-6) and torch.allclose(layer.w1.grad, w1.grad, rtol=1e-5, atol=1e-6)
    if w2 is not None:
        assert torch.allclose(layer.w2.grad, w2.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("aggregator,inc,out", [("MEAN", True, False), ("GCN", True, True)])
def test_emulated_three_layer_encoder_and_node_classification_steps(G_emul, aggregator, inc, out):
    """cfg4's shape end to end on the emulated kernels: DENSE sample (15-10-5 style fan-outs) -> three GraphSage stages with bias + RELU hooks ->
    cross-entropy over the target nodes -> backward -> dense Adagrad, three consecutive steps, against the oracle's restatement of
    GeneralEncoder::forward (encoder.cpp:195-257) and Model::train_batch's NODE_CLASSIFICATION branch (model.cpp:317-328)."""
    dims, classes = [12, 16, 8, 5], 5
    og = graph(500, 6000, 2, seed=13)
    dgraph = G_emul.MariusGraph(og.src_sorted_edges, og.dst_sorted_edges, og.num_nodes_in_memory)
    gen = torch.Generator().manual_seed(3)
    feats = torch.randn(500, dims[0], generator=gen)
    labels_all = torch.randint(classes, (500,), generator=gen)
    enc = G_emul.GraphSageEncoder(dims, aggregator, "RELU", bias=True, device="cpu")
    ref_layers = []
    for layer in enc.layers:
        with torch.no_grad():
            layer.bias.copy_(0.1 * torch.randn(layer.bias.shape, generator=gen))
        ref_layers.append((layer.w1.detach().clone().requires_grad_(True), None if layer.w2 is None else layer.w2.detach().clone().requires_grad_(True),
                           layer.bias.detach().clone().requires_grad_(True), aggregator, layer.activation))
    fan = [4, 3, 2]
    for step in range(3):
        seeds = torch.randperm(500, generator=gen)[:12]
        draws = {}

        def rand_cpu(i, incoming, t):
            draws[(i, incoming)] = torch.randint(1 << 40, (t,), generator=torch.Generator().manual_seed(50 * step + 2 * i + int(incoming)))
            return draws[(i, incoming)]

        want_g = NO.layered_neighbors(og, seeds, fan, inc, out, rand=rand_cpu)
        got_g = G_emul.LayeredNeighborSampler(dgraph, fan, inc, out).getNeighbors(seeds, rand=lambda i, incoming, t: draws[(i, incoming)])
        assert torch.equal(got_g.node_ids_, want_g.node_ids)
        x = feats[want_g.node_ids]
        labels = labels_all[seeds]
        want_loss, want_y = NO.node_classification_step(x, want_g, ref_layers, labels, 0.1)
        got_loss, got_y = G_emul.node_classification_step(enc, x.clone(), got_g, labels, 0.1)
        assert got_y.shape == (12, classes) and torch.allclose(got_y, want_y, rtol=1e-5, atol=1e-6) and torch.allclose(got_loss, want_loss, rtol=1e-5)
        for layer, (w1, w2, b, _a, _act) in zip(enc.layers, ref_layers):
            assert torch.allclose(layer.w1, w1, rtol=1e-4, atol=1e-6) and torch.allclose(layer.bias, b, rtol=1e-4, atol=1e-6)
            if w2 is not None:
                assert torch.allclose(layer.w2, w2, rtol=1e-4, atol=1e-6)
