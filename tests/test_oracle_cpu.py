"""CPU tests: pin the oracle against (1) the reference's own known answers, (2) golden vectors produced by the REFERENCE's
comparators/relation operators (oracle/_ref), (3) the torch RNG stream; plus host-side product logic that needs no GPU."""
import ctypes as C
import json
import math
import os
import re

import numpy as np
import pytest
import torch

from oracle import lp_oracle as O
from oracle.mt_oracle import OracleGenerator

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def known():
    with open(os.path.join(GOLD, "ref_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def refgold():
    return np.load(os.path.join(GOLD, "ref_scores_golden.npz"))


# ------------------------------------------------------------------------------------------------ reference known answers
def test_distmult_known_scores(known):
    k = known["distmult_forward"]
    emb, edges = torch.tensor(k["node_embeddings"]), torch.tensor(k["batch_edges"])
    rel = O.init_relations("DISTMULT", k["num_relations"], k["embedding_dim"])
    pos, inv = O.only_pos_forward("DISTMULT", edges, emb, rel, None)
    assert torch.equal(pos, torch.tensor(k["expected_scores"]))  # test_nn.py:158 uses torch.eq
    assert inv is None


def test_accumulate_gradients_known(known):
    k = known["accumulate_gradients"]
    grad, state = torch.tensor(k["grad"]), torch.tensor(k["state"])
    dw, ds = O.accumulate_gradients(grad, state, k["learning_rate"])
    assert torch.equal(ds, torch.tensor(k["expected_state_update"]))
    assert torch.equal(ds, grad.pow(2))
    assert torch.equal(dw, -1.0 * (grad / (ds.sqrt().add_(1e-10))))  # test_data.py:46-47


def test_train_batch_known_shapes(known):
    k = known["distmult_forward"]
    emb, edges = torch.tensor(k["node_embeddings"]), torch.tensor(k["batch_edges"])
    dst_neg = torch.tensor(known["train_batch_shapes"]["dst_neg_indices_mapping"])
    rel = O.init_relations("DISTMULT", 2, 2)
    out = O.train_batch("DISTMULT", emb, torch.zeros_like(emb), edges, dst_neg, None, rel, None)
    assert out["neg"].shape == (3, 2) and out["node_grad"].shape == emb.shape and out["inv_neg"] is None


# ------------------------------------------------------------------------------------------------ golden vectors from the reference code
OPS = {"hadamard": "hadamard", "complex_hadamard": "complex_hadamard", "translation": "translation"}


@pytest.mark.parametrize("name", sorted(OPS))
def test_relation_operators_match_reference(refgold, name):
    e, r = torch.from_numpy(refgold["op_in_e"]), torch.from_numpy(refgold["op_in_r"])
    assert torch.equal(O.REL_OPS[name](e, r), torch.from_numpy(refgold["op_" + name]))


@pytest.mark.parametrize("name", ["dot", "l2", "cosine"])
def test_comparators_match_reference(refgold, name):
    e, o = torch.from_numpy(refgold["op_in_e"]), torch.from_numpy(refgold["cmp_in_o"])
    assert torch.equal(O.COMPARATORS[name](e, o), torch.from_numpy(refgold["cmp_same_" + name]))
    for B in (12, 10):  # 10: B % C != 0 -> zero padded last chunk
        src, negs = torch.from_numpy(refgold["neg_in_src_%d" % B]), torch.from_numpy(refgold["neg_in_negs_%d" % B])
        got = O.COMPARATORS[name](src, negs)
        want = torch.from_numpy(refgold["cmp_neg_%s_%d" % (name, B)])
        assert got.shape == want.shape
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dec", ["DISTMULT", "COMPLEX", "TRANSE"])
def test_score_forward_backward_matches_reference(refgold, dec):
    g = lambda k: torch.from_numpy(refgold["fb_%s_%s" % (dec, k)])
    src, rel, dst, negs = [g(k).clone().requires_grad_(True) for k in ("src", "rel", "dst", "negs")]
    op, cmp = O.REL_OPS[O.DECODERS[dec][0]], O.COMPARATORS[O.DECODERS[dec][1]]
    adj = op(src, rel)
    pos, neg = cmp(adj, dst), cmp(adj, negs)
    loss = O.softmax_cross_entropy(pos, neg, "sum")
    loss.backward()
    tol = dict(rtol=1e-5, atol=1e-6)
    assert torch.allclose(pos, g("pos"), **tol) and torch.allclose(neg, g("neg"), **tol)
    assert torch.allclose(loss.reshape(1), g("loss"), **tol)
    for t, k in ((src, "g_src"), (rel, "g_rel"), (dst, "g_dst"), (negs, "g_negs")):
        assert torch.allclose(t.grad, g(k), **tol), k


def test_oracle_vs_live_reference_build():
    """When oracle/_ref/libmarius_ref.so is present (built from /root/reference), compare on fresh random inputs."""
    so = os.path.join(ROOT, "oracle", "_ref", "libmarius_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built")
    L = C.CDLL(so)
    fp = lambda a: a.ctypes.data_as(C.c_void_p)
    rs = np.random.RandomState(5)
    B, Cc, N, d = 21, 4, 9, 14
    src, negs = rs.randn(B, d).astype(np.float32), rs.randn(Cc, N, d).astype(np.float32)
    Bp = Cc * math.ceil(B / Cc)
    for cmp, name in [(0, "dot"), (1, "l2")]:
        res = np.zeros((Bp, N), np.float32)
        assert L.ref_compare_neg(cmp, fp(src), fp(negs), C.c_int64(B), C.c_int64(Cc), C.c_int64(N), C.c_int64(d), fp(res)) == 0
        got = O.COMPARATORS[name](torch.from_numpy(src), torch.from_numpy(negs))
        assert torch.allclose(got, torch.from_numpy(res), rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------------ decoder quirks the reference has
def test_pad_and_reshape_zero_rows_enter_loss():
    """comparators.cpp:7-20 + decoder_methods.cpp:103-111: B % C != 0 pads rows with zeros; they add log(1+N) each."""
    emb = torch.randn(30, 6)
    edges = torch.stack([torch.randint(30, (10,)), torch.zeros(10, dtype=torch.int64), torch.randint(30, (10,))], 1)
    dst_neg = torch.randint(30, (4, 5))
    rel = O.init_relations("DISTMULT", 1, 6)
    pos, neg, _, _ = O.node_corrupt_forward("DISTMULT", edges, emb, dst_neg, None, rel, None)
    assert pos.shape == (12,) and neg.shape == (12, 5)
    assert torch.equal(pos[10:], torch.zeros(2)) and torch.equal(neg[10:], torch.zeros(2, 5))
    full = O.softmax_cross_entropy(pos, neg)
    part = O.softmax_cross_entropy(pos[:10], neg[:10])
    assert torch.allclose(full - part, torch.tensor(2 * math.log(6.0)), atol=1e-5)


def test_transe_scores_are_positive_distances():
    emb = torch.randn(20, 8)
    edges = torch.stack([torch.randint(20, (6,)), torch.zeros(6, dtype=torch.int64), torch.randint(20, (6,))], 1)
    pos, neg, _, _ = O.node_corrupt_forward("TRANSE", edges, emb, torch.randint(20, (2, 4)), None, O.init_relations("TRANSE", 1, 8), None)
    assert (pos >= 0).all() and (neg > 0).all()


def test_edge_column_validation():
    with pytest.raises(RuntimeError, match="3 or 2 column"):
        O.only_pos_forward("DISTMULT", torch.zeros(3, 4, dtype=torch.int64), torch.randn(4, 2), None, None)


def test_deg_filter_entries_are_own_chunk():
    """test/cpp/unit/data/samplers/test_negative.cpp property: every DEG filter entry points at a positive of its own chunk."""
    B, Cc, n_deg = 12, 3, 5
    g = torch.Generator().manual_seed(0)
    deg = torch.randint(B, (Cc, n_deg), generator=g)
    edges = torch.zeros(B, 3, dtype=torch.int64)
    f = O.deg_negative_local_filter(deg, edges)
    chunk = math.ceil(B / Cc)
    for e, k in f.tolist():
        c = e // chunk
        assert deg[c, k] == e
    assert f.size(0) == int((deg // chunk == torch.arange(Cc).view(-1, 1)).sum())


# ------------------------------------------------------------------------------------------------ RNG stream
@pytest.fixture(scope="module")
def rnggold():
    with open(os.path.join(GOLD, "rng_golden.json")) as f:
        return json.load(f)


def test_c_oracle_rng_matches_golden_and_torch(rnggold):
    for case in rnggold["cases"]:
        if case["kind"] == "randint":
            got = OracleGenerator(case["seed"]).randint(case["high"], case["n"])
            assert got.tolist() == case["values"]
            torch.manual_seed(case["seed"])
            assert torch.randint(case["high"], (case["n"],)).tolist() == case["values"]  # this torch == golden torch
        elif case["kind"] == "randperm":
            og = OracleGenerator(case["seed"])
            assert og.randperm(case["n"]).tolist() == case["values"]
            assert og.randint(1000, 5).tolist() == case["next_randint_1000"]
        else:
            og = OracleGenerator(case["seed"])
            edges = np.array(case["edges"], dtype=np.int64)
            for call in case["calls"]:
                ids, deg = og.get_negatives(edges, case["num_nodes"], case["C"], case["N"], case["degree_fraction"], call["inverse"])
                assert ids.tolist() == call["ids"]
                if call["deg_pos"]:
                    assert deg.tolist() == call["deg_pos"]


# ------------------------------------------------------------------------------------------------ product: C-ABI library + host logic (no GPU)
def _declared_symbols():
    import re

    txt = open(os.path.join(ROOT, "include", "marius_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(marius_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from marius_amd import hip

    L = hip.lib()  # loads libmarius_hip.so and binds argtypes for every entry of SIGNATURES
    decl = _declared_symbols()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(L, name), "libmarius_hip.so does not export %s" % name
    assert sorted(hip.SIGNATURES) == decl, set(hip.SIGNATURES) ^ set(decl)
    # the version the header states, the library returns and the ctypes mirror expects are one number (a stale build is refused by hip.lib())
    hdr = open(os.path.join(ROOT, "include", "marius_hip.h")).read()
    assert L.marius_hip_abi_version() == hip.ABI_VERSION == int(re.search(r"#define MARIUS_HIP_ABI_VERSION (\d+)", hdr).group(1))
    assert L.marius_hip_struct_bytes(0) == C.sizeof(hip.LpDesc) and L.marius_hip_struct_bytes(1) == C.sizeof(hip.LpLayout)


def test_host_generator_matches_golden(rnggold):
    """Host half of the product sampler (seed / raw words / randperm are plain C, no device needed)."""
    from marius_amd import hip

    for case in rnggold["cases"]:
        if case["kind"] == "randperm":
            g = hip.Generator(case["seed"])
            assert g.randperm_host(case["n"]).tolist() == case["values"]
            raw = g.fill_host(5).numpy().view(np.uint32)
            assert (raw % 1000).tolist() == case["next_randint_1000"]
        elif case["kind"] == "randint" and case["high"] < 2 ** 28:
            raw = hip.Generator(case["seed"]).fill_host(case["n"]).numpy().view(np.uint32).astype(np.uint64)
            assert (raw % case["high"]).tolist() == case["values"]
    assert hip.negatives_raw_words(86054151, 50000, 50, 1000, 0) == 50000
    assert hip.negatives_raw_words(2 ** 28, 50000, 2, 10, 4) == 2 * (6 * 2 + 4)


def test_product_path_has_no_cpu_fallback():
    from marius_amd import hip

    with pytest.raises(hip.MariusHipError):
        hip.gather_rows(torch.zeros(4, 4), torch.zeros(2, dtype=torch.int64))


def test_lp_plan_validation_and_layout():
    from marius_amd import hip

    d = hip.LpDesc()
    d.relop, d.cmp, d.d, d.edge_cols, d.B, d.C, d.N, d.use_inverse, d.reduction = 1, 0, 100, 3, 1000, 10, 500, 1, 0
    d.src_neg, d.inv_rel = C.c_void_p(1), C.c_void_p(1)
    lay = hip.LpLayout()
    assert hip.lib().marius_lp_plan(C.byref(d), C.byref(lay)) == 0
    assert lay.Bp == 1000 and lay.n_ld == 500 and lay.d_ld == 100 and lay.total_bytes > 2 * 1000 * 500 * 4
    d.B = 1005  # ceil(1005/10) = 101 -> Bp = 1010 (pad_and_reshape)
    assert hip.lib().marius_lp_plan(C.byref(d), C.byref(lay)) == 0 and lay.Bp == 1010
    d.edge_cols = 4
    assert hip.lib().marius_lp_plan(C.byref(d), C.byref(lay)) == 1  # MARIUS_ERR_INVALID
    assert b"3 or 2 column" in hip.lib().marius_hip_last_error()
    d.edge_cols, d.d = 3, 7  # ComplEx needs even d
    assert hip.lib().marius_lp_plan(C.byref(d), C.byref(lay)) != 0


def test_global_score_filter_entries_are_exactly_the_true_edges():
    """Property the reference's own test states (test/cpp/unit/data/samplers/test_negative.cpp:108-166): every filter entry is a real
    edge sharing the uncorrupted endpoint and relation with its batch edge — and, for the global filter, none is missing."""
    from oracle import lp_oracle as O

    g = torch.Generator().manual_seed(3)
    num_nodes, R, E = 40, 3, 400
    all_edges = torch.stack([torch.randint(num_nodes, (E,), generator=g), torch.randint(R, (E,), generator=g), torch.randint(num_nodes, (E,), generator=g)], 1)
    src_sorted, dst_sorted = O.sort_all_edges(all_edges)
    batch = all_edges[torch.randperm(E, generator=g)[:25]]
    truth = set(map(tuple, all_edges.tolist()))
    for inverse in (False, True):
        flt = O.compute_filter_corruption_global(src_sorted, dst_sorted, batch, inverse)
        got = set()
        for eid, node in flt.tolist():
            s, r, d = batch[eid].tolist()
            assert ((node, r, d) if inverse else (s, r, node)) in truth
            got.add((eid, node))
        want = set()
        for eid, (s, r, d) in enumerate(batch.tolist()):
            for (s2, r2, d2) in truth:
                if r2 == r and ((d2 == d) if inverse else (s2 == s)):
                    want.add((eid, s2 if inverse else d2))
        assert got == want
        # the batch edge itself is always filtered (its own tail / head is a true edge)
        for eid, (s, r, d) in enumerate(batch.tolist()):
            assert (eid, s if inverse else d) in got


def test_oracle_adam_matches_torch_optim():
    """oracle.dense_adam_step (restatement of optim.cpp:186-232) against torch.optim.Adam on the same gradients (float-vs-double bias
    corrections differ in the last bits only)."""
    from oracle import lp_oracle as O

    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(13, 7, generator=g)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=0.1)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in range(5):
        grad = torch.randn(13, 7, generator=g)
        p_ref.grad = grad.clone()
        opt.step()
        O.dense_adam_step(p, grad, m, v, 0.1, step)
    assert torch.allclose(p, p_ref.detach(), rtol=2e-5, atol=1e-6)


# ---- the loss family (loss.cpp:50-187): the vectors and assertions of test/cpp/unit/nn/test_loss.cpp:8-16, 84, 119, 141-142, 180, 215-320
LOSS_POS4 = torch.tensor([.5, 2.5, 5.0, 7.5, 100.0, 250.0])
LOSS_NEG4 = torch.tensor([[.5, 10.0], [2.5, -1.0], [5.0, 1.0], [7.5, -5.0], [100.0, 20.0], [250.0, 10.0]])
LOSS_CASES = [(torch.tensor([500.0]), torch.tensor([[150.0, 100.0, 50.0, 25.0, 10.0]])), (torch.tensor([.1]), torch.tensor([[.001, -.001, -.005, -.1, -10.0]])),
              (torch.tensor([-500.0]), torch.tensor([[-150.0, -100.0, -50.0, -25.0, 10.0]])), (LOSS_POS4, LOSS_NEG4)]


@pytest.mark.parametrize("kind,terms", [("SOFTMAX_CE", 6), ("RANKING", 12), ("CROSS_ENTROPY", 6), ("BCE_AFTER_SIGMOID", 18), ("BCE_WITH_LOGITS", 18),
                                        ("MSE", 18), ("SOFTPLUS", 18)])
def test_loss_family_reductions_as_in_reference_tests(kind, terms):
    for pos, neg in LOSS_CASES:  # "ASSERT_NO_THROW" on every fixture, both reductions
        assert torch.isfinite(O.loss_function(kind, pos, neg, "mean", 0.0)) and torch.isfinite(O.loss_function(kind, pos, neg, "sum", 0.0))
    s, m = O.loss_function(kind, LOSS_POS4, LOSS_NEG4, "sum", 0.0), O.loss_function(kind, LOSS_POS4, LOSS_NEG4, "mean", 0.0)
    assert torch.equal(s / terms, m)  # sum / (number of loss terms) == mean, the denominators the reference asserts


def test_ranking_loss_grows_with_margin():  # test_loss.cpp:120-142
    l1, l2, l3 = (O.loss_function("RANKING", LOSS_POS4, LOSS_NEG4, "sum", mg) for mg in (-10.0, 5.0, 10.0))
    assert l1 < l2 < l3


def test_cross_entropy_equals_softmax_ce_on_scores():
    """Why the device path folds CROSS_ENTROPY into SOFTMAX_CE: log-sum-exp over [pos, neg...] either way."""
    g = torch.Generator().manual_seed(0)
    pos, neg = torch.randn(50, generator=g), torch.randn(50, 30, generator=g)
    a, b = O.loss_function("CROSS_ENTROPY", pos, neg, "sum"), O.loss_function("SOFTMAX_CE", pos, neg, "sum")
    assert abs(a.item() - b.item()) <= 1e-5 * abs(b.item())


def test_randperm_wide_path_matches_torch():
    """n >= 2^32 / 20: ATen's randperm_cpu switches to the inside-out shuffle with random64() draws (an epoch over > 214.7 M training edges,
    e.g. Twitter-2010 / Freebase86m at full size).  Oracle and library host function both reproduce torch.randperm bit for bit."""
    import ctypes as C

    from marius_amd import hip as H

    n = 0xFFFFFFFF // 20 + 3
    torch.manual_seed(77)
    ref = torch.randperm(n)
    og = OracleGenerator(77)
    assert np.array_equal(og.randperm(n), ref.numpy())
    L = H.lib()
    st = torch.zeros(625, dtype=torch.int32)
    L.marius_mt19937_seed_host(st.data_ptr(), 77)
    out = torch.empty(n, dtype=torch.int64)
    assert L.marius_mt19937_randperm_host(st.data_ptr(), out.data_ptr(), n) == 0
    assert torch.equal(out, ref)


def test_arith_check_is_neutral_on_the_reference_fp32_evaluation():
    """oracle/arith_check.error_pairs compares a device evaluation and the reference's fp32 evaluation with the float64 oracle.  Fed the fp32
    evaluation itself as the "device" result every ratio is exactly 1; fed the float64 result every device error is 0."""
    from oracle.arith_check import error_pairs, occurrence_oracle

    g = torch.Generator().manual_seed(0)
    B, Cn, N, d, U, R = 60, 3, 20, 16, 50, 5
    emb = torch.randn(U, d, generator=g) * 0.5
    edges = torch.stack([torch.randint(U, (B,), generator=g), torch.randint(R, (B,), generator=g), torch.randint(U, (B,), generator=g)], 1)
    dn, sn = torch.randint(U, (Cn, N), generator=g), torch.randint(U, (Cn, N), generator=g)
    rel = O.init_relations("COMPLEX", R, d) + 0.3 * torch.randn(R, d, generator=g)
    inv = O.init_relations("COMPLEX", R, d) + 0.3 * torch.randn(R, d, generator=g)
    for dtype, want_ratio in ((torch.float32, 1.0), (torch.float64, 0.0)):
        w, _ = occurrence_oracle("COMPLEX", emb, edges, dn, sn, rel, inv, dtype=dtype)
        lse = torch.logsumexp(torch.cat([w["pos"][:, None], w["neg"]], 1), 1)
        ilse = torch.logsumexp(torch.cat([w["inv_pos"][:, None], w["inv_neg"]], 1), 1)
        got = {"neg": w["neg"], "inv_neg": w["inv_neg"], "lse": lse, "inv_lse": ilse, "rowloss": lse - w["pos"], "inv_rowloss": ilse - w["inv_pos"],
               "loss": w["loss"], "gocc": w["node_grad"]}
        pairs = error_pairs("COMPLEX", emb, edges, dn, sn, rel, inv, got)
        for q in ("scores", "lse", "row_loss", "loss", "occ_grad"):
            assert pairs[q]["ratio_max"] == want_ratio and pairs[q]["ratio_rms"] == want_ratio, (q, pairs[q])
            assert 0 < pairs[q]["fp32_max"] < 1e-5


def test_only_the_checker_legs_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under marius_amd/ or include/ names it, and bench.py imports it only inside its two checker legs
    (cpu_baseline_leg: the timed CPU port; arith_check_leg: the float64 / float32 yardsticks) — never in the timed device region."""
    import re

    for base, _, files in os.walk(os.path.join(ROOT, "marius_amd")):
        if os.sep + "lib" in base:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(base, f)
                if f != "build.py":  # (building the checker is not using it)
                    assert "oracle/_build" not in text and "libmt_oracle" not in text and "oracle/_ref" not in text, os.path.join(base, f)
    src = open(os.path.join(ROOT, "bench.py")).read()
    legs = {}
    for m in re.finditer(r"^def (\w+)\(", src, re.M):
        legs[m.start()] = m.group(1)
    starts = sorted(legs)
    for m in re.finditer(r"^\s*(from|import)\s+oracle\b", src, re.M):
        owner = legs[max(s for s in starts if s <= m.start())]
        assert owner in ("cpu_baseline_leg", "arith_check_leg"), owner


# ------------------------------------------------------------------------------------------------ cfg4 slice: neighbour sampling / GraphSage oracle
def _nbr_graph(num_nodes, E, cols, seed, hubs=True):
    from oracle import neighbor_oracle as NO

    g = torch.Generator().manual_seed(seed)
    src, dst = torch.randint(num_nodes, (E,), generator=g), torch.randint(num_nodes, (E,), generator=g)
    if hubs:
        dst[torch.rand(E, generator=g) < 0.2] = 3          # one node with a fifth of all edges coming in
        src[torch.rand(E, generator=g) < 0.1] = 5
    lonely = num_nodes - 1                                  # a node without any edge
    src[src == lonely] = 0
    dst[dst == lonely] = 1
    edges = torch.stack([src, torch.randint(7, (E,), generator=g), dst], 1) if cols == 3 else torch.stack([src, dst], 1)
    return NO.MariusGraph.from_edges(edges, num_nodes), edges


def test_neighbor_oracle_properties():
    """oracle/neighbor_oracle.py is the reference's ATen op sequence (neighbor.cpp:9-105, 402-582, graph.cpp:16-44, 128-236, 290-398); the reference holds
    no known answers for it, so it is pinned by what the algorithm implies: ALL returns exactly the node's neighbour slice, in list order; UNIFORM
    returns min(degree, k) edges per node, each one an edge OF that node; the layered batch has unique ids, hop by hop, every hop's ids ascending and
    absent from the later hops; the mappings point at the right ids; the aggregation equals a dense float64 mean."""
    from oracle import neighbor_oracle as NO

    for cols in (2, 3):
        graph, edges = _nbr_graph(200, 3000, cols, seed=cols)
        ids = torch.tensor([3, 199, 5, 17, 17, 0])  # hub, lonely node, repeated id
        got, offs = NO.neighbors_for_node_ids(graph, ids, True)
        for i, v in enumerate(ids.tolist()):
            want = graph.dst_sorted_edges[graph.dst_sorted_edges[:, -1] == v]
            end = offs[i + 1] if i + 1 < len(ids) else got.size(0)
            assert torch.equal(got[offs[i]:end], want)
        k = 4
        num = graph.out_num_neighbors.index_select(0, ids)
        total = NO.uniform_total(num, k)
        rs = torch.randint(graph.max_out_num_neighbors, (total,), generator=torch.Generator().manual_seed(9))
        got, offs = NO.neighbors_for_node_ids(graph, ids, False, k, rs)
        assert got.size(0) == total == int(num.clamp(max=k).sum())
        for i, v in enumerate(ids.tolist()):
            end = offs[i + 1] if i + 1 < len(ids) else got.size(0)
            mine = got[offs[i]:end]
            assert mine.size(0) == min(int(num[i]), k) and bool((mine[:, 0] == v).all())
            allowed = {tuple(r) for r in graph.src_sorted_edges[graph.src_sorted_edges[:, 0] == v].tolist()}
            assert all(tuple(r) in allowed for r in mine.tolist())
            if int(num[i]) <= k:  # not capped: every neighbour, in list order
                assert torch.equal(mine, graph.src_sorted_edges[graph.src_sorted_edges[:, 0] == v])
        # layered, incoming + outgoing, three hops
        rg = torch.Generator().manual_seed(4)
        seeds = torch.tensor([7, 3, 150, 199])
        dg = NO.layered_neighbors(graph, seeds, [5, -1, 2], True, True, rand=lambda i, inc, t: torch.randint(1 << 30, (t,), generator=rg))
        assert dg.node_ids.unique().numel() == dg.node_ids.numel() and torch.equal(dg.node_ids[-4:], seeds)
        ho = dg.hop_offsets.tolist()
        assert ho[0] == 0 and ho[-1] == dg.node_ids.numel() and ho == sorted(ho) and len(ho) == 3 + 2
        for a, b in zip(ho[:-2], ho[1:-1]):  # every hop's new ids ascend (nonzero() of a bitmap)
            assert bool((dg.node_ids[a:b][1:] > dg.node_ids[a:b][:-1]).all())
        NO.perform_map(dg)
        assert torch.equal(dg.node_ids[dg.in_neighbors_mapping], dg.dst_sorted_edges[:, 0]) and torch.equal(dg.node_ids[dg.out_neighbors_mapping], dg.src_sorted_edges[:, -1])
        # the rows of in_offsets / out_offsets are the nodes of node_ids from hop 1 on, and each segment holds edges of exactly that node
        owners = dg.node_ids[ho[1]:]
        assert dg.in_offsets.numel() == owners.numel() == dg.out_offsets.numel()
        seg = NO.segment_ids_from_offsets(dg.in_offsets, dg.dst_sorted_edges.size(0))
        assert torch.equal(owners[seg], dg.dst_sorted_edges[:, -1])
        seg = NO.segment_ids_from_offsets(dg.out_offsets, dg.src_sorted_edges.size(0))
        assert torch.equal(owners[seg], dg.src_sorted_edges[:, 0])
        # aggregation against a dense float64 mean
        x = torch.randn(dg.node_ids.numel(), 12, generator=rg)
        a_i, self_rows = NO.graph_sage_aggregate(x, dg, "MEAN")
        n = owners.numel()
        want = torch.zeros(n, 12, dtype=torch.float64)
        cnt = torch.zeros(n, dtype=torch.float64)
        for m, offs_, T in ((dg.out_neighbors_mapping, dg.out_offsets, dg.src_sorted_edges.size(0)), (dg.in_neighbors_mapping, dg.in_offsets, dg.dst_sorted_edges.size(0))):
            s = NO.segment_ids_from_offsets(offs_, T)
            want.index_add_(0, s, x[m].double())
            cnt.index_add_(0, s, torch.ones(T, dtype=torch.float64))
        want = want / cnt.clamp(min=1).unsqueeze(-1)
        assert torch.allclose(a_i.double(), want, rtol=1e-5, atol=1e-6) and torch.equal(self_rows, x[ho[1]:])
        # next layer: the finished hop's nodes and their neighbour segments are dropped, ids shift
        before = dg.node_ids.clone()
        NO.prepare_for_next_layer(dg)
        assert torch.equal(dg.node_ids, before[ho[1]:]) and int(dg.hop_offsets[0]) == 0
        assert torch.equal(dg.node_ids[dg.in_neighbors_mapping], dg.dst_sorted_edges[:, 0])
