"""GPU parity tests: HIP path (through the C-ABI of libmarius_hip.so) vs the oracle on the same seeded inputs.

Bars: bit-exact for ids / indices / copies; floats within the three tiers of tests/tolerance.py at rtol 1e-4 (north_star: "within 1e-4
relative on float scores"): pure relative 1e-4 over entries >= 0.1 max, 3e-4 over entries >= 0.01 max, 3e-6 x max below.
"""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import lp_oracle as O
from oracle.mt_oracle import OracleGenerator
from tolerance import tiers

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def assert_close(got, want, what, rtol=RTOL, small_frac=None):
    """tests/tolerance.py:tiers — pure relative error <= rtol over entries >= 0.1 max |want|, <= 3 rtol over entries >= 0.01 max, absolute error
    <= 0.03 rtol x max below (round 6: the old form added atol = rtol x max to EVERY entry, which passed anything below 1 % of the maximum)"""
    tiers(got, want, what, rtol=rtol, small_frac=small_frac)


@pytest.fixture(scope="module")
def H():
    from marius_amd import hip

    hip.lib()
    return hip


# ------------------------------------------------------------------------------------------------ storage rows
@pytest.mark.parametrize("d", [2, 50, 100, 128, 400])
def test_gather_scatter_rows(H, dev, d):
    g = torch.Generator().manual_seed(d)
    num_nodes, n = 5000, 1777
    table = torch.randn(num_nodes, d, generator=g)
    state = torch.rand(num_nodes, d, generator=g)
    ids = torch.randperm(num_nodes, generator=g)[:n].sort().values
    t_d, s_d, ids_d = table.to(dev), state.to(dev), ids.to(dev)
    out = H.gather_rows(t_d, ids_d)
    assert torch.equal(out.cpu(), O.index_read(table, ids))  # bit exact copy
    a, b = H.gather_rows2(t_d, s_d, ids_d)
    assert torch.equal(a.cpu(), table[ids]) and torch.equal(b.cpu(), state[ids])
    delta = torch.randn(n, d, generator=g)
    H.scatter_add_rows(t_d, ids_d, delta.to(dev))
    ref = table.clone()
    O.index_add(ref, ids, delta)
    assert torch.equal(t_d.cpu(), ref)  # unique ids: one add per element, bit exact


def test_gather_empty_and_errors(H, dev):
    table = torch.randn(10, 8, device=dev)
    ids = torch.empty(0, dtype=torch.int64, device=dev)
    assert H.gather_rows(table, ids).shape == (0, 8)
    with pytest.raises(H.MariusHipError):
        H.check(H.lib().marius_gather_rows(H.ptr(table), 4, H.ptr(ids), 1, 8, H.ptr(table), 8, None), "bad ld")


def test_adagrad_rule_bit_exact(H, dev):
    g = torch.Generator().manual_seed(3)
    grad = torch.randn(1234, 100, generator=g)
    state = torch.rand(1234, 100, generator=g)
    st = state.clone()
    dw_ref, ds_ref = O.accumulate_gradients(grad, st, 0.1)
    st_d = state.to(dev)
    dw, ds = H.adagrad_rule(grad.to(dev), st_d, 0.1)
    assert torch.equal(ds.cpu(), ds_ref)
    assert torch.equal(st_d.cpu(), st)
    assert_close(dw, dw_ref, "dw", rtol=1e-6)


def test_dense_adagrad_step(H, dev):
    g = torch.Generator().manual_seed(4)
    w, gr, ssum = torch.randn(237, 100, generator=g), torch.randn(237, 100, generator=g), torch.rand(237, 100, generator=g)
    w_d, s_d = w.to(dev), ssum.to(dev)
    H.dense_adagrad_step(w_d, s_d, gr.to(dev), 0.1)
    O.dense_adagrad_step(w, gr, ssum, 0.1)
    assert_close(s_d, ssum, "sum", rtol=1e-6)
    assert_close(w_d, w, "w", rtol=1e-6)


# ------------------------------------------------------------------------------------------------ sampler
@pytest.mark.parametrize("n", [1, 623, 624, 625, 5000, 100000])
def test_mt19937_device_stream_bit_exact(H, dev, n):
    gen = H.Generator(42, dev)
    raw = gen.fill_device(n)
    want = OracleGenerator(42).raw(n)
    assert np.array_equal(raw.cpu().numpy().view(np.uint32), want)
    # stream continues correctly across calls and across host <-> device hand-over
    raw2 = gen.fill_device(700)
    og = OracleGenerator(42)
    og.raw(n)
    assert np.array_equal(raw2.cpu().numpy().view(np.uint32), og.raw(700))
    perm = gen.randperm_host(50)
    assert np.array_equal(perm.numpy(), og.randperm(50))


@pytest.mark.parametrize("B,C,N,f,num_nodes", [(6, 1, 5, 0.0, 6), (6, 3, 5, 0.5, 6), (1000, 10, 500, 0.0, 14541),
                                               (1000, 10, 500, 0.5, 14541), (5000, 50, 1000, 0.0, 86054151),
                                               (64, 2, 16, 0.25, 2 ** 28 + 5)])
def test_negative_sampler_bit_exact(H, dev, B, C, N, f, num_nodes):
    seed = 1234 + B
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(B)
    edges = torch.stack([torch.randint(num_nodes, (B,), generator=g), torch.randint(7, (B,), generator=g),
                         torch.randint(num_nodes, (B,), generator=g)], 1)
    og = OracleGenerator(seed)
    gen = H.Generator(seed, dev)
    edges_d = edges.to(dev)
    n_deg = int(N * f)
    for inverse in (True, False):  # DataLoader::negativeSample order: src negatives first (dataloader.cpp:498-503)
        want_ids, want_deg = og.get_negatives(edges.numpy(), num_nodes, C, N, f, inverse)
        words = H.negatives_raw_words(num_nodes, B, C, N, n_deg)
        raw = gen.fill_device(words)
        ids, deg = H.sample_negatives(raw, edges_d, num_nodes, C, N, f, inverse)
        assert np.array_equal(ids.cpu().numpy(), want_ids)
        if n_deg:
            assert np.array_equal(deg.cpu().numpy(), want_deg)
        # and the oracle itself is the torch stream (same generator calls as negative.cpp:340-357)
        if num_nodes < 2 ** 28 or True:
            chunks = []
            for _ in range(C):
                uni = torch.randint(num_nodes, (N - n_deg,))
                if f > 0:
                    pos = torch.randint(0, B, (n_deg,))
                    uni = torch.cat([edges[pos, 0 if inverse else 2], uni])
                chunks.append(uni)
            assert np.array_equal(torch.stack(chunks).numpy(), want_ids)


def test_select_edges(H, dev):
    g = torch.Generator().manual_seed(9)
    E = 5000
    edges32 = torch.randint(0, 1000, (E, 3), generator=g, dtype=torch.int32)
    gen = H.Generator(7, dev)
    perm = gen.randperm_host(E)
    torch.manual_seed(7)
    assert torch.equal(perm, torch.randperm(E))
    out = H.select_edges(edges32.to(dev), perm.to(dev), 1000, 777)
    assert torch.equal(out.cpu(), edges32[perm[1000:1777]].to(torch.int64))


# ------------------------------------------------------------------------------------------------ unique map
@pytest.mark.parametrize("n,hi", [(1, 5), (12, 6), (12000, 14541), (200000, 86054151), (200000, 300), (4095, 1 << 20), (4096, 1 << 20), (4097, 1 << 20),
                                  (16385, 1), (70001, (1 << 36) - 5), (70001, (1 << 40) + 3), (2200000, 86054151)])
def test_sort_unique_matches_map_tensors(H, dev, n, hi):
    g = torch.Generator().manual_seed(n + hi)
    parts = [torch.randint(hi, (n // 4 + 1,), generator=g) for _ in range(4)]
    uniq_ref, mapped_ref = O.map_tensors(parts)
    ids = torch.cat(parts).to(dev)
    um = H.UniqueMap(ids.numel(), dev).run(ids, key_bits=max(1, math.ceil(math.log2(hi + 1))))
    U = int(um.count.item())
    assert U == uniq_ref.numel()
    assert torch.equal(um.uniq[:U].cpu(), uniq_ref)
    assert torch.equal(um.inverse[: ids.numel()].cpu(), torch.cat(mapped_ref))
    seg = um.seg[: U + 1].cpu()
    perm = um.perm[: ids.numel()].cpu().long()
    assert seg[0] == 0 and seg[U] == ids.numel()
    sorted_ids = ids.cpu()[perm]
    assert torch.equal(sorted_ids, torch.sort(ids.cpu(), stable=True).values)
    assert torch.equal(perm, torch.sort(ids.cpu(), stable=True).indices)  # stable order


def test_sort_unique_reuses_its_workspace_across_calls_and_sizes(H, dev):
    """The hand-written sort keeps no state between calls: the same workspace sorts different inputs of different sizes back to back (the granule
    arrays are re-zeroed by the call's first kernel, the histogram ticket is per call), and a call on another stream's workspace in between
    does not disturb it."""
    g = torch.Generator().manual_seed(99)
    um, other = H.UniqueMap(150000, dev), H.UniqueMap(5000, dev)
    for n, hi in ((150000, 1 << 27), (4097, 300), (150000, 1000), (1, 2), (90000, 1 << 27), (150000, 1 << 27)):
        ids = torch.randint(hi, (n,), generator=g)
        other.run(torch.randint(77, (5000,), generator=g).to(dev), 7)
        um.run(ids.to(dev), 27)
        u, inv = torch.unique(ids, return_inverse=True)
        U = int(um.count.item())
        assert U == u.numel() and torch.equal(um.uniq[:U].cpu(), u) and torch.equal(um.inverse[:n].cpu(), inv)
        assert torch.equal(um.perm[:n].cpu().long(), torch.sort(ids, stable=True).indices)


def test_sort_unique_replays_from_a_captured_graph(H, dev):
    """A captured call carries its histogram ticket as a constant: the replay must still wait for the histogram to be zeroed (rs_emit_kernel puts
    the ready word back to 0), so every replay of the graph sorts the then-current content of the input buffer."""
    g = torch.Generator().manual_seed(5)
    n = 50000
    ids = torch.randint(1 << 27, (n,), generator=g).to(dev)
    um = H.UniqueMap(n, dev)
    um.run(ids, 27)  # warm
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            um.run(ids, 27)
    for trial in range(3):
        fresh = torch.randint(1 << 27, (n,), generator=g)
        ids.copy_(fresh.to(dev))
        graph.replay()
        torch.cuda.synchronize()
        u, inv = torch.unique(fresh, return_inverse=True)
        U = int(um.count.item())
        assert U == u.numel() and torch.equal(um.uniq[:U].cpu(), u) and torch.equal(um.inverse[:n].cpu(), inv)


def _plan_valid_bytes(H, plan, n, U):
    """the defined part of a marius_segment_plan buffer (padding between its four arrays is never written): pos_plan [n], chunk_plan, row_plan [n], occ_single [n]"""
    pos = (n * 16 + 255) // 256 * 256
    chunk = ((n + 31) // 32 * 16 + 255) // 256 * 256
    return torch.cat([plan[:n * 16], plan[pos:pos + (n + 31) // 32 * 16], plan[pos + chunk:pos + chunk + n * 16], plan[2 * pos + chunk:2 * pos + chunk + n]])


@pytest.mark.parametrize("B,C,N,num_nodes,R,cols,hubs", [(50000, 50, 1000, 86054151, 14824, 3, True),   # the bench batch: 49 + 13 tiles, 3 + 2 passes
                                                       (1000, 10, 500, 14541, 237, 3, False),          # FB15k-237 shape
                                                       (250, 5, 40, 4000, 11, 3, False), (7, 1, 3, 50, 2, 3, False), (1, 1, 1, 2, 1, 3, False),
                                                       (4096, 1, 4096, 1 << 20, 1, 2, False),           # 2-column edges: one job, n = 3 tiles exactly
                                                       (2049, 3, 683, 99999, 5, 3, True)])              # n = one past a tile boundary
def test_prepare_maps_one_launch_equals_the_separate_launches(H, dev, B, C, N, num_nodes, R, cols, hubs):
    """marius_prepare_maps (round 6): ids assembled, both unique maps, batch-local edges and both segment plans in ONE persistent launch (work items
    drawn from a queue, phases separated by completion counters) must leave bit for bit what marius_assemble_ids + marius_sort_unique +
    marius_remap_edges + marius_segment_plan leave, for the node ids and the relation ids of the same batch — and what map_tensors
    (util.cpp:180-205) gives on the CPU.  Repeated on the same workspaces, alternating with the separate form (both leave the control block zero)."""
    g = torch.Generator().manual_seed(B + N)
    CN = C * N
    nbits, rbits = max(1, math.ceil(math.log2(num_nodes))), max(1, math.ceil(math.log2(R + 1)))
    L = 2 * B + 2 * CN
    um, ur = H.UniqueMap(L, dev), H.UniqueMap(B, dev)        # fused
    vm, vr = H.UniqueMap(L, dev), H.UniqueMap(B, dev)        # separate launches
    for trial in range(3):
        src, dst = torch.randint(num_nodes, (B,), generator=g), torch.randint(num_nodes, (B,), generator=g)
        if hubs:  # a few nodes in a large share of the edges: segments spanning many tiles
            src[torch.rand(B, generator=g) < 0.3] = int(torch.randint(num_nodes, (1,), generator=g))
            dst[torch.rand(B, generator=g) < 0.1] = int(src[0])
        rel = torch.randint(R, (B,), generator=g)
        edges = (torch.stack([src, rel, dst], 1) if cols == 3 else torch.stack([src, dst], 1)).to(dev)
        sneg, dneg = torch.randint(num_nodes, (C, N), generator=g).to(dev), torch.randint(num_nodes, (C, N), generator=g).to(dev)
        # ---- separate launches
        ids = torch.empty(L, dtype=torch.int64, device=dev)
        H.check(H.lib().marius_assemble_ids(H.ptr(edges), B, cols, H.ptr(sneg), H.ptr(dneg), CN, H.ptr(ids), H.stream_ptr()), "assemble_ids")
        vm.run(ids, nbits)
        want_edges = torch.empty_like(edges)
        H.check(H.lib().marius_remap_edges(H.ptr(edges), H.ptr(vm.inverse), B, cols, H.ptr(want_edges), H.stream_ptr()), "remap_edges")
        want_plan = H.segment_plan(vm, L)
        if cols == 3:
            vr.run(edges[:, 1].contiguous(), rbits)
            want_rplan = H.segment_plan(vr, B)
        # ---- one launch
        ids_out, rel_out = torch.full((L,), -1, dtype=torch.int64, device=dev), torch.full((B,), -1, dtype=torch.int64, device=dev)
        got_edges = torch.full_like(edges, -1)
        plan = torch.empty(int(H.lib().marius_segment_plan_bytes(L)), dtype=torch.uint8, device=dev)
        rplan = torch.empty(int(H.lib().marius_segment_plan_bytes(B)), dtype=torch.uint8, device=dev)
        jobs = [H.map_job(um, nbits, edges=edges, src_neg=sneg, dst_neg=dneg, ids_out=ids_out, plan=plan, edges_out=got_edges)]
        if cols == 3:
            jobs.append(H.map_job(ur, rbits, edges=edges, col=1, ids_out=rel_out, plan=rplan))
        assert H.prepare_maps(jobs)
        torch.cuda.synchronize()
        assert torch.equal(ids_out, ids) and torch.equal(got_edges, want_edges)
        for a, b, n, gp, wp in ((um, vm, L, plan, want_plan),) + (((ur, vr, B, rplan, want_rplan),) if cols == 3 else ()):
            U = int(b.count.item())
            assert int(a.count.item()) == U
            assert torch.equal(a.uniq[:n], b.uniq[:n]) and torch.equal(a.inverse[:n], b.inverse[:n]) and torch.equal(a.perm[:n], b.perm[:n])   # (uniq: zero tail included)
            assert torch.equal(a.seg[:U + 1], b.seg[:U + 1])
            assert torch.equal(_plan_valid_bytes(H, gp, n, U), _plan_valid_bytes(H, wp, n, U))
        if cols == 3:
            assert torch.equal(rel_out, edges[:, 1])
        if trial == 0:  # and the CPU restatement of map_tensors
            e = edges.cpu()
            uniq_ref, mapped = O.map_tensors([e[:, 0], e[:, -1], sneg.cpu().flatten(), dneg.cpu().flatten()])
            U = int(um.count.item())
            assert torch.equal(um.uniq[:U].cpu(), uniq_ref) and torch.equal(um.inverse[:L].cpu(), torch.cat(mapped))
        # the separate form on the FUSED form's workspace, then the fused form again (next trial): either leaves it reusable
        um.run(ids, nbits)
        assert torch.equal(um.inverse[:L], vm.inverse[:L])


def test_prepare_maps_given_ids_many_replays_next_to_a_busy_stream(H, dev):
    """ids_in form (a list that already exists) and robustness of the queue: 300 back-to-back launches on a side stream while the main stream keeps
    every CU busy with large GEMMs — workgroups of the persistent launch then start late and out of step, which the item queue must not care about
    (nothing waits for a workgroup that has not started)."""
    g = torch.Generator().manual_seed(3)
    n = 200000
    ids = torch.randint(86054151, (n,), generator=g).to(dev)
    um = H.UniqueMap(n, dev)
    want = H.UniqueMap(n, dev).run(ids, 27)
    a = torch.randn(4096, 4096, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    for it in range(300):
        if it % 10 == 0:
            for _ in range(4):
                a = (a @ a) * 1e-3
        with torch.cuda.stream(side):
            um.uniq.fill_(-5)
            assert H.prepare_maps([H.map_job(um, 27, ids=ids)])
    torch.cuda.synchronize()
    assert torch.equal(um.uniq, want.uniq) and torch.equal(um.inverse, want.inverse) and torch.equal(um.perm, want.perm) and int(um.count.item()) == int(want.count.item())
    # outside the fused launch's range: refused by the predicate, not attempted
    assert not H.prepare_maps([H.map_job(H.UniqueMap(8, dev), 40, ids=torch.zeros(8, dtype=torch.int64, device=dev))])


def test_sort_unique_empty(H, dev):
    um = H.UniqueMap(8, dev)
    um.run(torch.empty(0, dtype=torch.int64, device=dev))
    assert int(um.count.item()) == 0


# ------------------------------------------------------------------------------------------------ decoder fwd / loss / bwd
DEC = {"DISTMULT": (0, 0), "COMPLEX": (1, 0), "TRANSE": (2, 1)}


def make_batch(decoder, B, C, N, d, U, R, seed, zipf=False):
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(U, d, generator=g) * 0.5
    state = torch.rand(U, d, generator=g)
    if zipf:  # heavy duplicates: hub nodes / hub relations
        src = (torch.rand(B, generator=g) ** 4 * U).long().clamp_(0, U - 1)
        rel = (torch.rand(B, generator=g) ** 4 * R).long().clamp_(0, R - 1)
    else:
        src = torch.randint(U, (B,), generator=g)
        rel = torch.randint(R, (B,), generator=g)
    dst = torch.randint(U, (B,), generator=g)
    edges = torch.stack([src, rel, dst], 1)
    dst_neg = torch.randint(U, (C, N), generator=g)
    src_neg = torch.randint(U, (C, N), generator=g)
    relations = O.init_relations(decoder, R, d) + 0.3 * torch.randn(R, d, generator=g)
    inv_relations = O.init_relations(decoder, R, d) + 0.3 * torch.randn(R, d, generator=g)
    return emb, state, edges, dst_neg, src_neg, relations, inv_relations


def run_hip_lp(H, dev, decoder, emb, edges, dst_neg, src_neg, rel, inv, use_inverse, reduction, dst_filter=None, src_filter=None):
    relop, cmp = DEC[decoder]
    B, (C, N), d = edges.size(0), dst_neg.shape, emb.size(1)
    W = H.LpWorkspace(relop, cmp, d, B, C, N, use_inverse, H.REDUCE_SUM if reduction == "sum" else H.REDUCE_MEAN, edges.size(1), True, dev)
    t = lambda x: None if x is None else x.to(dev)
    W.bind(t(emb), t(edges), t(dst_neg), t(src_neg), t(rel), t(inv) if use_inverse else None, t(dst_filter), t(src_filter))
    W.forward()
    W.loss()
    W.backward()
    torch.cuda.synchronize()
    return W


@pytest.mark.parametrize("decoder", ["DISTMULT", "COMPLEX", "TRANSE"])
@pytest.mark.parametrize("use_inverse", [True, False])
@pytest.mark.parametrize("B,C,N,d", [(6, 3, 5, 2), (100, 10, 50, 50), (1000, 10, 500, 100), (250, 7, 130, 100), (300, 4, 260, 200),
                                     (5, 4, 6, 8), (2, 4, 64, 100), (1, 3, 33, 100)])  # the last three: whole chunks of padding rows
@pytest.mark.parametrize("reduction", ["sum"])
def test_lp_forward_loss_backward(H, dev, decoder, use_inverse, B, C, N, d, reduction):
    U, R = max(40, B), 11
    emb, state, edges, dst_neg, src_neg, rel, inv = make_batch(decoder, B, C, N, d, U, R, seed=B + d, zipf=(B == 250))
    want = O.train_batch(decoder, emb, state, edges, dst_neg, src_neg, rel, inv if use_inverse else None, reduction=reduction)
    W = run_hip_lp(H, dev, decoder, emb, edges, dst_neg, src_neg, rel, inv, use_inverse, reduction)
    Bp = W.layout.Bp
    assert Bp == want["pos"].numel()
    assert_close(W.pos(0), want["pos"], "pos")
    assert_close(W.neg(0), want["neg"], "neg")
    if use_inverse:
        assert_close(W.pos(1), want["inv_pos"], "inv_pos")
        assert_close(W.neg(1), want["inv_neg"], "inv_neg")
    assert_close(W.loss_values()[0:1], want["loss"].reshape(1), "loss")
    # node gradient: segment-sum of the occurrence gradients in map order == autograd's index_add
    L = W.num_occ()
    occ_ids = torch.cat([edges[:, 0], edges[:, 2], src_neg.flatten(), dst_neg.flatten()])
    gocc = W.gocc()[:, :d].cpu()
    node_grad = torch.zeros(U, d, dtype=torch.float64).index_add_(0, occ_ids, gocc.double())
    assert_close(node_grad.float(), want["node_grad"], "node_grad")
    rel_grad = torch.zeros(R, d, dtype=torch.float64).index_add_(0, edges[:, 1], W.grel(0)[:, :d].cpu().double())
    assert_close(rel_grad.float(), want["rel_grad"], "rel_grad")
    if use_inverse:
        inv_grad = torch.zeros(R, d, dtype=torch.float64).index_add_(0, edges[:, 1], W.grel(1)[:, :d].cpu().double())
        assert_close(inv_grad.float(), want["inv_rel_grad"], "inv_rel_grad")


@pytest.mark.parametrize("decoder", ["DISTMULT", "COMPLEX", "TRANSE"])
@pytest.mark.parametrize("level", ["generic", "fast", "pp"])
def test_lp_lower_kernel_levels_also_match(H, dev, decoder, level, monkeypatch):
    """MARIUS_KERNELS forces the generic / fast contraction kernels at a shape the resident ones normally take."""
    monkeypatch.setenv("MARIUS_KERNELS", level)
    H.reload_env()
    B, C, N, d, U, R = 300, 4, 200, 100, 400, 7
    emb, state, edges, dst_neg, src_neg, rel, inv = make_batch(decoder, B, C, N, d, U, R, seed=31)
    want = O.train_batch(decoder, emb, state, edges, dst_neg, src_neg, rel, inv)
    W = run_hip_lp(H, dev, decoder, emb, edges, dst_neg, src_neg, rel, inv, True, "sum")
    assert_close(W.neg(0), want["neg"], "neg")
    assert_close(W.neg(1), want["inv_neg"], "inv_neg")
    occ_ids = torch.cat([edges[:, 0], edges[:, 2], src_neg.flatten(), dst_neg.flatten()])
    node_grad = torch.zeros(U, d, dtype=torch.float64).index_add_(0, occ_ids, W.gocc()[:, :d].cpu().double())
    assert_close(node_grad.float(), want["node_grad"], "node_grad")


@pytest.mark.parametrize("decoder", ["DISTMULT", "TRANSE"])
def test_lp_mean_reduction_and_filter(H, dev, decoder):
    B, C, N, d, U, R = 96, 4, 40, 20, 60, 5
    emb, state, edges, dst_neg, src_neg, rel, inv = make_batch(decoder, B, C, N, d, U, R, seed=77)
    g = torch.Generator().manual_seed(5)
    dst_filter = torch.stack([torch.randint(B, (30,), generator=g), torch.randint(N, (30,), generator=g)], 1)
    src_filter = torch.stack([torch.randint(B, (17,), generator=g), torch.randint(N, (17,), generator=g)], 1)
    want = O.train_batch(decoder, emb, state, edges, dst_neg, src_neg, rel, inv, dst_filter=dst_filter, src_filter=src_filter, reduction="mean")
    W = run_hip_lp(H, dev, decoder, emb, edges, dst_neg, src_neg, rel, inv, True, "mean", dst_filter, src_filter)
    assert_close(W.neg(0), want["neg"], "neg filtered")
    assert_close(W.neg(1), want["inv_neg"], "inv_neg filtered")
    assert_close(W.loss_values()[0:1], want["loss"].reshape(1), "loss")
    occ_ids = torch.cat([edges[:, 0], edges[:, 2], src_neg.flatten(), dst_neg.flatten()])
    node_grad = torch.zeros(U, d, dtype=torch.float64).index_add_(0, occ_ids, W.gocc()[:, :d].cpu().double())
    assert_close(node_grad.float(), want["node_grad"], "node_grad")


def test_lp_two_column_edges(H, dev):
    """num_relations == 1: edges [B,2], no relation operator, single direction (decoder_methods.cpp:98-101)."""
    B, C, N, d, U = 64, 4, 24, 16, 50
    g = torch.Generator().manual_seed(2)
    emb = torch.randn(U, d, generator=g)
    edges = torch.stack([torch.randint(U, (B,), generator=g), torch.randint(U, (B,), generator=g)], 1)
    dst_neg, src_neg = torch.randint(U, (C, N), generator=g), torch.randint(U, (C, N), generator=g)
    want = O.train_batch("DISTMULT", emb, torch.zeros(U, d), edges, dst_neg, src_neg, None, None)
    W = H.LpWorkspace(0, 0, d, B, C, N, False, H.REDUCE_SUM, 2, True, dev)
    W.bind(emb.to(dev), edges.to(dev), dst_neg.to(dev), src_neg.to(dev), None, None)
    W.forward(); W.loss(); W.backward()
    assert_close(W.pos(0), want["pos"], "pos")
    assert_close(W.neg(0), want["neg"], "neg")
    occ_ids = torch.cat([edges[:, 0], edges[:, 1], src_neg.flatten(), dst_neg.flatten()])
    node_grad = torch.zeros(U, d, dtype=torch.float64).index_add_(0, occ_ids, W.gocc()[:, :d].cpu().double())
    assert_close(node_grad.float(), want["node_grad"], "node_grad")


def test_lp_bad_edge_columns_raises(H, dev):
    with pytest.raises(H.MariusHipError, match="3 or 2 column"):
        H.LpWorkspace(0, 0, 8, 4, 1, 4, False, 0, 4, True, dev)


def test_compute_ranks(H, dev):
    g = torch.Generator().manual_seed(8)
    pos, neg = torch.randn(500, generator=g), torch.randn(500, 333, generator=g)
    neg[5, 7] = pos[5]  # tie counts as >=
    got = H.compute_ranks(pos.to(dev), neg.to(dev))
    assert torch.equal(got.cpu(), O.compute_ranks(pos, neg))


# ------------------------------------------------------------------------------------------------ gradient reduce + fused update
@pytest.mark.parametrize("n,U,d", [(1, 1, 4), (1000, 900, 100), (5000, 37, 50), (20000, 19000, 100), (4096, 3, 7), (3000, 2500, 400)])
def test_segment_sum_rows(H, dev, n, U, d):
    g = torch.Generator().manual_seed(n + U)
    ids = torch.randint(U, (n,), generator=g)
    rows = torch.randn(n, d, generator=g)
    um = H.UniqueMap(n, dev).run(ids.to(dev), key_bits=32)
    nu = int(um.count.item())
    out = torch.zeros(nu, d, device=dev)
    H.segment_sum_rows(rows.to(dev), um, n, d, out)
    uniq, inv = torch.unique(ids, return_inverse=True)
    want = torch.zeros(nu, d, dtype=torch.float64).index_add_(0, inv, rows.double())
    assert_close(out, want.float(), "segment sum", rtol=2e-6 * max(1, n // max(nu, 1)) ** 0.5 + 1e-6)
    # indexed variant (dense relation gradient)
    dense = torch.zeros(U, d, device=dev)
    H.segment_sum_rows(rows.to(dev), um, n, d, dense, out_rows=um.uniq)
    want_dense = torch.zeros(U, d, dtype=torch.float64).index_add_(0, ids, rows.double())
    assert_close(dense, want_dense.float(), "segment sum indexed", rtol=1e-5)
    # reproducible: same bits on a second run
    out2 = torch.zeros(nu, d, device=dev)
    H.segment_sum_rows(rows.to(dev), um, n, d, out2)
    assert torch.equal(out, out2)


def test_segment_adagrad_scatter_matches_reference_update(H, dev):
    g = torch.Generator().manual_seed(11)
    num_nodes, n, d = 3000, 8000, 100
    ids = (torch.rand(n, generator=g) ** 3 * num_nodes).long()
    rows = torch.randn(n, d, generator=g) * 0.1
    table, state = torch.randn(num_nodes, d, generator=g), torch.rand(num_nodes, d, generator=g)
    um = H.UniqueMap(n, dev).run(ids.to(dev), key_bits=16)
    t_d, s_d = table.to(dev), state.to(dev)
    H.segment_adagrad_scatter(rows.to(dev), um, n, d, t_d, s_d, lr=0.1)
    # reference sequence: grad = index_add; accumulateGradients on the gathered copies; indexAdd x2
    uniq, inv = torch.unique(ids, return_inverse=True)
    grad = torch.zeros(uniq.numel(), d).index_add_(0, inv, rows)
    st = O.index_read(state, uniq)
    dw, ds = O.accumulate_gradients(grad, st, 0.1)
    O.index_add(table, uniq, dw)
    O.index_add(state, uniq, ds)
    assert_close(t_d, table, "table after update", rtol=1e-5)
    assert_close(s_d, state, "state after update", rtol=1e-5)


@pytest.mark.parametrize("n,num_nodes,power,d", [(8000, 3000, 3, 100), (200000, 86054151, 1, 100), (50000, 40, 4, 64), (33, 5, 1, 20), (1, 9, 1, 8)])
def test_planned_segment_adagrad_scatter_is_bit_identical(H, dev, n, num_nodes, power, d):
    """marius_segment_plan precomputes the index chains of the fused update; the planned call must leave exactly the bits of the unplanned one:
    singletons only, hub rows spanning hundreds of chunks (40 ids over 50,000 occurrences), a single chunk, a single row."""
    g = torch.Generator().manual_seed(n + d)
    ids = (torch.rand(n, generator=g) ** power * num_nodes).long().clamp_(0, num_nodes - 1)
    rows = (torch.randn(n, d, generator=g) * 0.1).to(dev)
    rows_of_table = min(num_nodes, 200000)
    ids = ids % rows_of_table
    table, state = torch.randn(rows_of_table, d, generator=g), torch.rand(rows_of_table, d, generator=g)
    um = H.UniqueMap(n, dev).run(ids.to(dev), key_bits=28)
    ta, sa, tb, sb = table.to(dev), state.to(dev), table.to(dev), state.to(dev)
    H.segment_adagrad_scatter(rows, um, n, d, ta, sa, lr=0.1)
    plan = H.segment_plan(um, n)
    H.segment_adagrad_scatter(rows, um, n, d, tb, sb, lr=0.1, plan=plan)
    assert torch.equal(ta, tb) and torch.equal(sa, sb)
    assert not torch.equal(ta.cpu(), table)
    U = int(um.count.item())
    oa, ob = torch.zeros(n, d, device=dev), torch.zeros(n, d, device=dev)
    H.segment_sum_rows(rows, um, n, d, oa)
    H.segment_sum_rows(rows, um, n, d, ob, plan=plan)
    assert torch.equal(oa[:U], ob[:U])


@pytest.mark.parametrize("n,rows_of_table,d,planned", [(8000, 3000, 100, True), (50000, 40, 64, True), (33, 5, 20, False), (200000, 200000, 100, True)])
def test_tracked_update_keeps_the_magnitude_bound(H, dev, n, rows_of_table, d, planned):
    """marius_segment_adagrad_scatter_tracked: the same bits as the untracked update, and *absmax >= every |w| in the table afterwards, starting
    from marius_table_absmax — the bound marius_lp_desc.absmax needs for fp16 operand records, kept without a pass over the table."""
    g = torch.Generator().manual_seed(n + d)
    ids = (torch.rand(n, generator=g) ** 2 * rows_of_table).long().clamp_(0, rows_of_table - 1)
    rows = (torch.randn(n, d, generator=g) * 0.1).to(dev)
    table, state = torch.randn(rows_of_table, d, generator=g) * 1e-3, torch.zeros(rows_of_table, d)   # fresh rows: the first Adagrad step moves them by lr
    um = H.UniqueMap(n, dev).run(ids.to(dev), key_bits=28)
    plan = H.segment_plan(um, n) if planned else None
    ta, sa, tb, sb = table.to(dev), state.to(dev), table.to(dev), state.to(dev)
    bound = H.table_absmax(tb)
    assert float(bound) == float(table.abs().max())
    H.segment_adagrad_scatter(rows, um, n, d, ta, sa, lr=0.1, plan=plan)
    H.segment_adagrad_scatter(rows, um, n, d, tb, sb, lr=0.1, plan=plan, absmax=bound)
    assert torch.equal(ta, tb) and torch.equal(sa, sb)
    assert float(bound) == float(tb.abs().max()) > 0.05           # grew with the update (1e-3 -> ~lr), and is exact here: nothing shrank


@pytest.mark.parametrize("rows,d,ld", [(200000, 100, 100), (1000, 100, 100), (777, 50, 50), (513, 33, 33), (4096, 100, 112), (3, 7, 7), (40000, 400, 400)])
def test_table_absmax_flat_strided_and_counted(H, dev, rows, d, ld):
    """marius_table_absmax / _counted: the bound marius_lp_desc.absmax starts from — exact max |x| for contiguous rows (flat float4 stream), padded
    row pitch (the padding is NOT read: it holds larger values here), sizes that leave a ragged float4 tail, a running bound that is only ever
    raised, and a capacity-sized buffer of which only the first *count rows are valid (the rest holds NaN / huge garbage)."""
    g = torch.Generator().manual_seed(rows + d)
    buf = torch.full((rows, ld), 1e30)
    buf[:, :d] = torch.randn(rows, d, generator=g)
    r0, c0 = int(torch.randint(rows, (1,), generator=g)), int(torch.randint(d, (1,), generator=g))
    buf[r0, c0] = -7.5   # the maximum, negative, at an arbitrary place
    t = buf.to(dev)[:, :d] if ld != d else buf[:, :d].contiguous().to(dev)
    assert t.stride(0) == ld
    out = H.table_absmax(t)
    assert float(out) == 7.5
    H.table_absmax(t * 0.5, out=out)   # running bound: a smaller table does not lower it
    assert float(out) == 7.5
    H.table_absmax(t * 2.0, out=out)
    assert float(out) == 15.0
    # counted: rows [count, capacity) are garbage
    count = max(1, rows // 3)
    cap = t.clone() if ld == d else t
    garbage = cap.clone()
    garbage[count:] = 1e30
    garbage[count::2] = float("nan")
    if count > 1:
        garbage[count - 1, 0] = 3.25
    want = float(garbage[:count].abs().max())
    cnt = torch.tensor([count], dtype=torch.int64, device=dev)
    assert float(H.table_absmax(garbage, count=cnt)) == want
    assert float(H.table_absmax(garbage, count=torch.tensor([rows + 5], dtype=torch.int64, device=dev) * 0)) == 0.0   # zero valid rows


@pytest.mark.parametrize("planned", [True, False])
def test_grouped_update_of_three_tables_equals_the_separate_updates(H, dev, planned):
    """marius_segment_adagrad_scatter_group: a step's node table + both relation tables in one launch pair (the relation tables share one
    id map, as the two directions of a batch do) — every table and state bit-identical to its own marius_segment_adagrad_scatter_tracked call,
    magnitude bounds included; unplanned jobs take the documented fallback (separate launches) with the same result."""
    g = torch.Generator().manual_seed(7)
    d = 100
    shapes = [(200000, 150000), (50000, 3000)]   # (occurrences, table rows): the bench's node and relation shapes (a Zipf hub spans hundreds of chunks)
    maps, plans = [], []
    for n, nrows in shapes:
        ids = (torch.rand(n, generator=g) ** 3 * nrows).long().clamp_(0, nrows - 1)
        um = H.UniqueMap(n, dev).run(ids.to(dev), key_bits=28)
        maps.append(um)
        plans.append(H.segment_plan(um, n) if planned else None)
    jobs_src = [(0, 0.1), (1, 0.1), (1, 0.05)]
    want, got, jobs, bounds_w, bounds_g = [], [], [], [], []
    for which, lr in jobs_src:
        n, nrows = shapes[which]
        rows = (torch.randn(n, d, generator=g) * 0.1).to(dev)
        table, state = (torch.randn(nrows, d, generator=g) * 1e-3), torch.rand(nrows, d, generator=g) * 1e-4
        tw, sw, tg, sg = table.to(dev), state.to(dev), table.to(dev), state.to(dev)
        bw, bg = H.table_absmax(tw), H.table_absmax(tg)
        H.segment_adagrad_scatter(rows, maps[which], n, d, tw, sw, lr=lr, plan=plans[which], absmax=bw)
        want.append((tw, sw))
        got.append((tg, sg))
        bounds_w.append(bw)
        bounds_g.append(bg)
        jobs.append(dict(rows=rows, um=maps[which], n=n, d=d, table=tg, state=sg, lr=lr, plan=plans[which], absmax=bg))
    H.segment_adagrad_scatter_group(jobs)
    torch.cuda.synchronize()
    for (tw, sw), (tg, sg), bw, bg in zip(want, got, bounds_w, bounds_g):
        assert torch.equal(tw, tg) and torch.equal(sw, sg)
        assert float(bw) == float(bg) == float(tg.abs().max())


@pytest.mark.parametrize("with_rows", [False, True])
def test_group_with_a_reduce_only_job_equals_the_separate_calls(H, dev, with_rows):
    """marius_segment_update.sum_out: the sharded step's tail — both relation tables' Adagrad steps AND the reduction of the node gradients into the
    gradient payload — as one launch pair.  The reduce-only job must equal marius_segment_sum_rows_planned (bit for bit; the rows with a single
    occurrence are copied instead of summed from zero), with and without the output-row indirection, and the Adagrad jobs beside it their own calls."""
    g = torch.Generator().manual_seed(23)
    d, n, nrows, nrel, B = 100, 200000, 150000, 3000, 50000
    ids = (torch.rand(n, generator=g) ** 3 * nrows).long().clamp_(0, nrows - 1)
    um = H.UniqueMap(n, dev).run(ids.to(dev), key_bits=28)
    plan = H.segment_plan(um, n)
    U = int(um.count.item())
    rows = (torch.randn(n, d, generator=g) * 0.1).to(dev)
    out_rows = torch.randperm(n, generator=g).to(dev) if with_rows else None   # any injective map into an [n, d] payload
    want = torch.full((n, d), 7.0, device=dev)
    H.segment_sum_rows(rows, um, n, d, want, out_rows=out_rows, plan=plan)
    rel_ids = (torch.rand(B, generator=g) ** 3 * nrel).long().clamp_(0, nrel - 1)
    rm = H.UniqueMap(B, dev).run(rel_ids.to(dev), key_bits=14)
    rplan = H.segment_plan(rm, B)
    rrows = [(torch.randn(B, d, generator=g) * 0.1).to(dev) for _ in range(2)]
    tabs = [(torch.randn(nrel, d, generator=g) * 1e-3, torch.rand(nrel, d, generator=g) * 1e-4) for _ in range(2)]
    ref = []
    for (t, s_), r in zip(tabs, rrows):
        tw, sw = t.to(dev), s_.to(dev)
        H.segment_adagrad_scatter(r, rm, B, d, tw, sw, lr=0.1, plan=rplan)
        ref.append((tw, sw))
    got = torch.full((n, d), 7.0, device=dev)
    mine = [(t.to(dev), s_.to(dev)) for t, s_ in tabs]
    H.segment_adagrad_scatter_group([dict(rows=rrows[0], um=rm, n=B, d=d, table=mine[0][0], state=mine[0][1], lr=0.1, plan=rplan),
                                     dict(rows=rrows[1], um=rm, n=B, d=d, table=mine[1][0], state=mine[1][1], lr=0.1, plan=rplan),
                                     dict(rows=rows, um=um, n=n, d=d, plan=plan, sum_out=got, sum_out_rows=out_rows)])
    torch.cuda.synchronize()
    assert torch.equal(got, want)   # (untouched payload rows keep their 7.0 in both)
    touched = (got != 7.0).any(1).sum().item()
    assert touched == U
    for (tw, sw), (tg, sg) in zip(ref, mine):
        assert torch.equal(tw, tg) and torch.equal(sw, sg)


# ------------------------------------------------------------------------------------------------ whole steps vs the CPU path
@pytest.mark.parametrize("decoder,f", [("COMPLEX", 0.0), ("DISTMULT", 0.5), ("TRANSE", 0.0)])
def test_train_steps_match_cpu_reference_path(H, dev, decoder, f):
    """3 synchronous steps: sampled ids bit-exact (same generator stream), unique map bit-exact, loss/tables within tolerance."""
    from marius_amd.lp_step import DeviceLinkPredictionStep
    from oracle.cpu_step import CpuLinkPredictionStep

    num_nodes, R, d, B, C, N, E, seed = 5000, 13, 20, 200, 4, 50, 2000, 99
    g = torch.Generator().manual_seed(1)
    table = (torch.rand(num_nodes, d, generator=g) - 0.5) * 0.6
    state = torch.zeros(num_nodes, d)
    edges_all = torch.stack([torch.randint(num_nodes, (E,), generator=g), torch.randint(R, (E,), generator=g),
                             torch.randint(num_nodes, (E,), generator=g)], 1)
    cpu = CpuLinkPredictionStep(decoder, table.clone(), state.clone(), R, B, C, N, degree_fraction=f)
    t_d, s_d = table.to(dev), state.to(dev)
    hipstep = DeviceLinkPredictionStep(decoder, num_nodes, R, d, B, C, N, degree_fraction=f, seed=seed, device=dev, node_table=t_d, node_state=s_d)
    torch.manual_seed(seed)
    perm_ref = torch.randperm(E)                       # setActiveEdges draws first (dataloader.cpp:180)
    perm = hipstep.gen.randperm_host(E)
    assert torch.equal(perm, perm_ref)
    e32 = edges_all.to(torch.int32).to(dev)
    for s in range(3):
        batch = edges_all[perm_ref[s * B:(s + 1) * B]]
        want = cpu.step(batch)
        edges = H.select_edges(e32, perm.to(dev), s * B, B)
        assert torch.equal(edges.cpu(), batch)
        W = hipstep.step(edges)
        assert torch.equal(hipstep.last["src_neg"].cpu(), want["src_neg"])   # bit-exact sampled node indices
        assert torch.equal(hipstep.last["dst_neg"].cpu(), want["dst_neg"])
        U = int(hipstep.um.count.item())
        assert torch.equal(hipstep.um.uniq[:U].cpu(), want["uniq"])
        assert_close(W.pos(0), want["pos"], "pos step %d" % s)
        l2 = 1e-4 if decoder == "TRANSE" else None   # small distances: sqrt of a cancelling x^2 + y^2 - 2 x y (tests/tolerance.py:tiers)
        assert_close(W.neg(0), want["neg"], "neg step %d" % s, small_frac=l2)
        assert_close(W.neg(1), want["inv_neg"], "inv_neg step %d" % s, small_frac=l2)
        assert_close(W.loss_values()[0:1], want["loss"].reshape(1), "loss step %d" % s)
    assert_close(t_d, cpu.table, "node table after 3 steps", rtol=2e-4)
    assert_close(s_d, cpu.state, "adagrad state after 3 steps", rtol=2e-4)
    assert_close(hipstep.rel, cpu.rel, "relations after 3 steps", rtol=2e-4)
    assert_close(hipstep.inv_rel, cpu.inv_rel, "inverse relations after 3 steps", rtol=2e-4)


# ------------------------------------------------------------------------------------------------ sharded exchange (world 1 on the GPU)
def test_sharded_hip_backend_equals_single_gpu_step(H, dev):
    """The N>1 code path (owner split points, all-to-all(v) over RCCL, owner-side dedupe + Adagrad) run with world_size 1 must
    reproduce the fused single-GPU step."""
    import os

    import torch.distributed as dist

    from marius_amd.lp_step import DeviceLinkPredictionStep
    from marius_amd.sharded import HipBackend, sharded_step

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    import datetime

    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    try:
        _sharded_hip_backend_body(H, dev, dist)
    finally:  # a failed run must not leave its communicator to the rest of the process
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.destroy_process_group()


def _sharded_hip_backend_body(H, dev, dist):
    from marius_amd.lp_step import DeviceLinkPredictionStep
    from marius_amd.sharded import HipBackend, sharded_step

    num_nodes, R, d, B, C, N, E, seed = 3000, 9, 100, 200, 4, 60, 1200, 21
    g = torch.Generator().manual_seed(2)
    table = (torch.rand(num_nodes, d, generator=g) - 0.5) * 0.5
    edges_all = torch.stack([torch.randint(num_nodes, (E,), generator=g), torch.randint(R, (E,), generator=g),
                             torch.randint(num_nodes, (E,), generator=g)], 1).to(dev)
    ta, sa = table.to(dev), torch.zeros(num_nodes, d, device=dev)
    tb, sb = table.to(dev), torch.zeros(num_nodes, d, device=dev)
    fused = DeviceLinkPredictionStep("COMPLEX", num_nodes, R, d, B, C, N, seed=seed, device=dev, node_table=ta, node_state=sa)
    split = DeviceLinkPredictionStep("COMPLEX", num_nodes, R, d, B, C, N, seed=seed, device=dev)
    be = HipBackend(split, tb, sb)
    for s in range(3):
        batch = edges_all[s * B:(s + 1) * B].contiguous()
        W = fused.step(batch)
        loss = sharded_step(be, batch, 0, 1, num_nodes)
        assert_close(loss.reshape(1), W.loss_values()[0:1], "loss", rtol=1e-6)
    assert_close(tb, ta, "table", rtol=1e-6)
    assert_close(sb, sa, "state", rtol=1e-6)
    assert_close(split.rel, fused.rel, "rel", rtol=1e-6)
    # pipelined variant (preparation one step ahead on a side stream + second communicator): same numbers
    from marius_amd.sharded import PipelinedShardedTrainer

    tc, sc = table.to(dev), torch.zeros(num_nodes, d, device=dev)
    piped = DeviceLinkPredictionStep("COMPLEX", num_nodes, R, d, B, C, N, seed=seed, device=dev)
    side = dist.new_group(backend="gloo")
    tr = PipelinedShardedTrainer(piped, tc, sc, edges_all, None, 0, 1, num_nodes, sync_interval=1, side_group=side)
    for s in range(3):
        tr.step()
    torch.cuda.synchronize()
    assert_close(tc, ta, "pipelined table", rtol=1e-6)
    assert_close(sc, sa, "pipelined state", rtol=1e-6)
    assert_close(piped.rel, fused.rel, "pipelined rel", rtol=1e-6)
    # gpu_sync_interval > 1: replicas step locally on the relation rows their batch touched (no per-step all-reduce); at world 1 that
    # is the same trajectory as the dense step
    te, se = table.to(dev), torch.zeros(num_nodes, d, device=dev)
    loc = DeviceLinkPredictionStep("COMPLEX", num_nodes, R, d, B, C, N, seed=seed, device=dev)
    tr = PipelinedShardedTrainer(loc, te, se, edges_all, None, 0, 1, num_nodes, sync_interval=16, side_group=side)
    for s in range(3):
        tr.step()
    torch.cuda.synchronize()
    assert_close(te, ta, "sync_interval table", rtol=1e-6)
    assert_close(loc.rel, fused.rel, "sync_interval rel", rtol=1e-6)
    assert_close(loc.inv_rel, fused.inv_rel, "sync_interval inv_rel", rtol=1e-6)
    # overlapped exchange (staleness 1): rows of batch t are read after update t-2 and before update t-1 — replay the recorded
    # (ids, rows used, row gradients) of every step through the Adagrad rule and check exactly that.
    td, sd = table.to(dev), torch.zeros(num_nodes, d, device=dev)
    ovl = DeviceLinkPredictionStep("COMPLEX", num_nodes, R, d, B, C, N, seed=seed, device=dev)
    trace = []
    tr = PipelinedShardedTrainer(ovl, td, sd, edges_all, None, 0, 1, num_nodes, sync_interval=1, side_group=side, staleness=1, trace=trace)
    steps = 5
    for s in range(steps):
        tr.step()
    tr.finish()
    T, S_ = table.to(dev).clone(), torch.zeros(num_nodes, d, device=dev)
    snaps = [T.clone()]  # snaps[k] = table after updates 0..k-1
    for t in range(steps):
        u, g = trace[t]["uniq"], trace[t]["grad"]
        assert u.numel() == torch.unique(u).numel()
        S_[u] += g * g
        T[u] += -0.1 * g / (S_[u].sqrt() + 1e-10)
        snaps.append(T.clone())
    for t in range(steps):
        expect = snaps[max(t - 1, 0)][trace[t]["uniq"]]
        assert_close(trace[t]["emb"], expect, "rows of step %d carry updates < %d" % (t, max(t - 1, 0)), rtol=1e-6)
    assert not torch.equal(trace[2]["emb"], snaps[2][trace[2]["uniq"]])  # the test data does make consecutive batches share rows
    assert_close(td, T, "overlapped table", rtol=1e-5)
    assert_close(sd, S_, "overlapped state", rtol=1e-5)


# ------------------------------------------------------------------------------------------------ dense Adam (optim.cpp:186-232)
@pytest.mark.gpu
@pytest.mark.parametrize("amsgrad,wd", [(False, 0.0), (True, 0.0), (False, 0.01)])
def test_dense_adam_step_matches_reference_ops(H, dev, amsgrad, wd):
    g = torch.Generator().manual_seed(5)
    n = (37, 100)
    p_ref = torch.randn(n, generator=g)
    m_ref, v_ref = torch.zeros(n), torch.zeros(n)
    vm_ref = torch.zeros(n) if amsgrad else None
    p, m, v = p_ref.to(dev), m_ref.to(dev), v_ref.to(dev)
    vm = vm_ref.to(dev) if amsgrad else None
    for step in range(4):
        grad = torch.randn(n, generator=g) * (0.1 if step != 2 else 3.0)
        O.dense_adam_step(p_ref, grad, m_ref, v_ref, 0.1, step, weight_decay=wd, max_exp_avg_sq=vm_ref)
        H.dense_adam_step(p, m, v, grad.to(dev), 0.1, step, weight_decay=wd, max_exp_avg_sq=vm)
    torch.cuda.synchronize()
    assert_close(m, m_ref, "exp_avg", rtol=1e-6)
    assert_close(v, v_ref, "exp_avg_sq", rtol=1e-6)
    assert_close(p, p_ref, "param", rtol=1e-5)
    if amsgrad:
        assert_close(vm, vm_ref, "max_exp_avg_sq", rtol=1e-6)


# ------------------------------------------------------------------------------------------------ the other loss functions (loss.cpp:69-187)
LOSSES = ["RANKING", "CROSS_ENTROPY", "BCE_AFTER_SIGMOID", "BCE_WITH_LOGITS", "MSE", "SOFTPLUS"]


@pytest.mark.gpu
@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("decoder,use_inverse,B,C,N,d,reduction", [("DISTMULT", True, 100, 10, 50, 50, "sum"), ("COMPLEX", True, 250, 7, 130, 100, "mean"),
                                                                  ("TRANSE", True, 96, 4, 40, 20, "sum"), ("DISTMULT", False, 5, 4, 6, 8, "mean")])
@pytest.mark.parametrize("novlog", [False, True])
def test_lp_other_losses_forward_backward(H, dev, monkeypatch, novlog, loss, decoder, use_inverse, B, C, N, d, reduction):
    """loss value, node and relation gradients of every LossFunction subclass against the oracle (autograd through the restated forward
    with the same torch.nn.functional loss calls as loss.cpp), incl. B % C != 0 (padding rows enter these losses) and the L2 comparator."""
    if novlog:  # generic backward kernels with the loss-specific dL/dS instead of the log-gradient buffer + tuned kernels
        if decoder == "TRANSE":
            pytest.skip("the L2 comparator always takes the generic kernels")
        monkeypatch.setenv("MARIUS_NO_VLOG", "1")
        H.reload_env()
    U, R, margin = max(40, B), 11, 0.7
    emb, state, edges, dst_neg, src_neg, rel, inv = make_batch(decoder, B, C, N, d, U, R, seed=B + d + len(loss))
    want = O.train_batch(decoder, emb, state, edges, dst_neg, src_neg, rel, inv if use_inverse else None, reduction=reduction, loss=loss, margin=margin)
    relop, cmp = DEC[decoder]
    W = H.LpWorkspace(relop, cmp, d, B, C, N, use_inverse, H.REDUCE_SUM if reduction == "sum" else H.REDUCE_MEAN, 3, True, dev, loss=H.LOSS[loss], margin=margin)
    t = lambda x: x.to(dev)
    W.bind(t(emb), t(edges), t(dst_neg), t(src_neg), t(rel), t(inv) if use_inverse else None)
    W.forward()
    W.loss()
    W.backward()
    torch.cuda.synchronize()
    assert (W.layout.vlog != 0) == (not novlog and decoder != "TRANSE" and loss not in ("MSE", "CROSS_ENTROPY"))
    assert_close(W.neg(0), want["neg"], "neg")   # the scores themselves stay intact for the caller
    assert_close(W.loss_values()[0:1], want["loss"].reshape(1), "loss")
    occ_ids = torch.cat([edges[:, 0], edges[:, 2], src_neg.flatten(), dst_neg.flatten()])
    node_grad = torch.zeros(U, d, dtype=torch.float64).index_add_(0, occ_ids, W.gocc()[:, :d].cpu().double())
    assert_close(node_grad.float(), want["node_grad"], "node_grad")
    rel_grad = torch.zeros(R, d, dtype=torch.float64).index_add_(0, edges[:, 1], W.grel(0)[:, :d].cpu().double())
    assert_close(rel_grad.float(), want["rel_grad"], "rel_grad")
    if use_inverse:
        inv_grad = torch.zeros(R, d, dtype=torch.float64).index_add_(0, edges[:, 1], W.grel(1)[:, :d].cpu().double())
        assert_close(inv_grad.float(), want["inv_rel_grad"], "inv_rel_grad")


@pytest.mark.gpu
@pytest.mark.parametrize("runs", [[1], [5000], [0, 7, 0], [3000, 2500, 4000, 1], [25000] * 8, [1000, 0, 0, 3], [17] * 64])
def test_merge_unique_runs_equals_sort_unique(H, dev, runs):
    """Owner side of the sharded exchange: concatenated ascending duplicate-free runs (one per sender) merged by binary searches must give
    bit for bit what the stable radix sort gives (unique ids, inverse, stable permutation, segment offsets, count)."""
    import ctypes as C

    g = torch.Generator().manual_seed(len(runs) * 1000 + sum(runs))
    hi = max(10, sum(runs))  # dense id range -> many ids are requested by several senders
    parts = [torch.sort(torch.randperm(hi, generator=g)[:r])[0] for r in runs]
    ids = torch.cat(parts).to(dev)
    n = ids.numel()
    a, b = H.UniqueMap(max(n, 1), dev), H.UniqueMap(max(n, 1), dev)
    a.run(ids, 63)
    offs = [0]
    for r in runs:
        offs.append(offs[-1] + r)
    arr = (C.c_int64 * len(offs))(*offs)
    H.check(H.lib().marius_merge_unique_runs(H.ptr(ids), n, arr, len(runs), H.ptr(b.uniq), H.ptr(b.inverse), H.ptr(b.perm), H.ptr(b.seg), H.ptr(b.count),
                                             H.ptr(b.ws), b.ws.numel(), H.stream_ptr()), "merge_unique_runs")
    torch.cuda.synchronize()
    U = int(a.count.item())
    assert int(b.count.item()) == U == torch.unique(ids).numel()
    assert torch.equal(b.uniq[:n], a.uniq[:n]) and torch.equal(b.inverse[:n], a.inverse[:n])
    assert torch.equal(b.perm[:n], a.perm[:n]) and torch.equal(b.seg[:U + 1], a.seg[:U + 1])


# ------------------------------------------------------------------------------------------------ fixed-capacity exchange (exchange.hip)
@pytest.mark.gpu
@pytest.mark.parametrize("world,cap_slack", [(1, 1.0), (2, 1.5), (8, 1.5), (4, 1.0)])
def test_fixed_capacity_exchange_halves_against_numpy(H, dev, world, cap_slack):
    """marius_a2a_rows_post / _wait and the dead (-1) padding through merge -> plan -> grouped update, against a numpy restatement and against the
    same update on the compacted lists.  One process plays every rank: `world` requesters, each asking `world` owners (contiguous
    id ranges of ceil(num_nodes / world) rows, storage.cpp:75); the all-to-alls are index shuffles here."""
    from oracle import exchange_oracle as X

    g = torch.Generator().manual_seed(11 + world)
    num_nodes, d, L = 5000, 100, 1600
    S = (num_nodes + world - 1) // world
    cap = H.a2a_capacity(L, world, cap_slack)
    assert cap == X.capacity(L, world, cap_slack)
    table = torch.rand(num_nodes, d, generator=g) - 0.5
    reqs, places, ums, invs, slots = [], [], [], [], []
    for r in range(world):
        ids = torch.randint(num_nodes, (L,), generator=g)
        if r == 0:
            ids[: L // 3] = ids[0]  # a hub: a segment spanning many chunks on the requester side
        um = H.UniqueMap(L, dev).run(ids.to(dev), 63)
        U = int(um.count.item())
        offs = torch.empty(world + 1, dtype=torch.int64, device=dev)
        H.check(H.lib().marius_owner_offsets(H.ptr(um.uniq), H.ptr(um.count), S, world, H.ptr(offs), H.stream_ptr()), "owner_offsets")
        req, place, flag, slot = H.a2a_rows_post(um, offs, S, world, cap, inverse=um.inverse)
        torch.cuda.synchronize()
        uq, of = um.uniq[:U].cpu().numpy(), offs.cpu().numpy()
        assert np.array_equal(of, X.owner_offsets(uq, S, world))
        # the same split points with the send counts in one launch, and the header record the host side of the all-to-all(v) reads:
        # ONE kernel writes payload, fence, stamp into pinned memory (marius_a2a_publish) — bit-equal to the numpy record, checksum included
        offs2, cnt = H.owner_offsets_counts(um, S, world)
        rec = torch.zeros(2 * world + 4, dtype=torch.int64).pin_memory()
        recv_counts = torch.randint(0, 999, (world,), generator=g).to(dev) if r % 2 else None
        H.a2a_publish(offs2, rec, stamp=100 + r, recv_counts=recv_counts, overflow=flag)
        torch.cuda.synchronize()
        assert torch.equal(offs2, offs) and np.array_equal(cnt.cpu().numpy(), np.diff(of))
        assert np.array_equal(rec.numpy(), X.record(of, None if recv_counts is None else recv_counts.cpu().numpy(), int(flag.item()), 100 + r))
        assert H.a2a_record_ok(rec, world)
        want_req, want_place, over = X.post(uq, S, world, cap)   # oracle/exchange_oracle.py: the numpy restatement
        assert bool(flag.item()) == over
        assert np.array_equal(req.cpu().numpy(), want_req)
        assert np.array_equal(place[:U].cpu().numpy(), want_place)  # (rows beyond an owner's capacity: slot 0, a defined index inside the payload)
        if over:
            continue
        assert np.array_equal(slot.cpu().numpy(), want_place[um.inverse.cpu().numpy()])
        reqs.append(req)
        places.append(place)
        ums.append(um)
    if len(reqs) < world:
        return  # (capacity exceeded for some requester at slack 1.0: the flag was the point)
    # ---- rows: every owner gathers for every requester (negative ids skipped), the requester bounds / compacts what it received
    for r in range(world):
        recv = torch.zeros(world * cap, d, device=dev)
        for q in range(world):
            shard = table[q * S:min((q + 1) * S, num_nodes)].to(dev)
            H.gather_rows(shard, reqs[r][q * cap:(q + 1) * cap], out=recv[q * cap:(q + 1) * cap])
        U = int(ums[r].count.item())
        emb = torch.full((L, d), float("nan"), device=dev)
        bound = torch.zeros(1, device=dev)
        H.a2a_rows_wait(recv, absmax=bound, place=places[r], count_dev=ums[r].count, emb=emb)
        want = table[ums[r].uniq[:U].cpu()]
        assert torch.equal(emb[:U].cpu(), want) and bool(torch.isnan(emb[U:]).all())
        assert float(bound) == float(want.abs().max())
        bound2 = torch.zeros(1, device=dev)
        H.a2a_rows_wait(recv, absmax=bound2)
        assert float(bound2) == float(want.abs().max())  # (unused slots are zeros here)
        assert torch.equal(recv[places[r][:U]].cpu(), want)  # the payload read in place through the slot indices
    # ---- owner 0: merge the `world` received runs (with their -1 padding), plan, grouped update == the update over the compacted lists
    recv_ids = torch.cat([reqs[r][0:cap] for r in range(world)])  # block 0 of every requester
    n = world * cap
    grads = torch.rand(n, d, generator=g).to(dev) - 0.5
    grads[recv_ids < 0] = float("nan")  # a dead slot's gradient row must never be read
    arr = (ctypes.c_int64 * (world + 1))(*[q * cap for q in range(world + 1)])
    b = H.UniqueMap(n, dev)
    H.check(H.lib().marius_merge_unique_runs(H.ptr(recv_ids), n, arr, world, H.ptr(b.uniq), H.ptr(b.inverse), H.ptr(b.perm), H.ptr(b.seg), H.ptr(b.count),
                                             H.ptr(b.ws), b.ws.numel(), H.stream_ptr()), "merge_unique_runs")
    plan = H.segment_plan(b, n)
    S0 = min(S, num_nodes)
    t1, s1 = table[:S0].clone().to(dev), torch.full((S0, d), 0.01, device=dev)
    H.segment_adagrad_scatter_group([dict(rows=grads, um=b, n=n, d=d, table=t1, state=s1, lr=0.1, plan=plan)])
    keep = (recv_ids >= 0).nonzero().flatten()
    c = H.UniqueMap(n, dev).run(recv_ids[keep].contiguous(), 63)
    t2, s2 = table[:S0].clone().to(dev), torch.full((S0, d), 0.01, device=dev)
    if keep.numel():
        H.segment_adagrad_scatter(grads[keep].contiguous(), c, keep.numel(), d, t2, s2, 0.1)
    torch.cuda.synchronize()
    # same occurrence order inside every segment; the -1 run shifts the 32-position chunk boundaries, so a segment of 3+ occurrences may be
    # associated differently ((a + b) + c vs a + (b + c)): bit-equal up to two requesters per row, 1e-6 beyond
    assert not bool(torch.isnan(t1).any()) and not bool(torch.isnan(s1).any())
    assert torch.equal((s1 != 0.01).any(1), (s2 != 0.01).any(1)), "the set of updated rows differs"
    if world <= 2:
        assert torch.equal(t1, t2) and torch.equal(s1, s2)
    else:
        assert torch.allclose(t1, t2, rtol=1e-6, atol=1e-7) and torch.allclose(s1, s2, rtol=1e-6, atol=1e-8)
