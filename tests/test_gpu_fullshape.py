"""GPU parity at the shapes the bench number is quoted on: the persistent score kernel walks many units per
workgroup, crosses row-tile and chunk-direction seams and ends each chunk in a partial (104-row) tile only at Bc > 128.

  * one decoder step at B=50,000 / C=50 / N=1000 / d=100 ComplEx + inverse edges (cfg2's batch) against the oracle;
  * B=4096 / C=4 / N=1000 (Bc = 1024: 8 row tiles, partial last column tile) for every score-kernel variant;
  * ComplEx-sized d=400 with 2-column edges (cfg5's row shape: d > 128 takes the K-chunked kernels);
  * one whole training step (sample -> unique -> gather -> forward/backward -> Adagrad scatter) at the bench shape on a
    10 M-row table against oracle/cpu_step.py: sampled ids and the unique map bit-exact, floats to tolerance.

Tolerance: `close_report` (scores, loss, tables) asserts the worst PURE relative error over every entry with
|want| >= FLOOR x max|want| (FLOOR = 1e-2) against rtol = 1e-4 and an absolute error of rtol x FLOOR x max|want| below that, and
prints the numbers so the claim is checkable; accumulated gradients are measured against the oracle evaluated in float64
(`grad_report`).
"""
import numpy as np
import pytest
import torch

from oracle import lp_oracle as O

pytestmark = pytest.mark.gpu

from tolerance import FLOOR, RTOL, close_report  # noqa: F401  (tests/tolerance.py)


@pytest.fixture(scope="module")
def H():
    from marius_amd import hip

    hip.lib()
    return hip


DEC = {"DISTMULT": (0, 0), "COMPLEX": (1, 0), "TRANSE": (2, 1)}


def make_batch(decoder, B, C, N, d, U, R, seed, edge_cols=3, scale=0.5):
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(U, d, generator=g) * scale
    cols = [torch.randint(U, (B,), generator=g)]
    if edge_cols == 3:
        cols.append(torch.randint(R, (B,), generator=g))
    cols.append(torch.randint(U, (B,), generator=g))
    edges = torch.stack(cols, 1)
    dst_neg = torch.randint(U, (C, N), generator=g)
    src_neg = torch.randint(U, (C, N), generator=g)
    rel = inv = None
    if edge_cols == 3:
        rel = O.init_relations(decoder, R, d) + 0.3 * torch.randn(R, d, generator=g)
        inv = O.init_relations(decoder, R, d) + 0.3 * torch.randn(R, d, generator=g)
    return emb, edges, dst_neg, src_neg, rel, inv


def run_hip(H, dev, decoder, emb, edges, dst_neg, src_neg, rel, inv, use_inverse):
    relop, cmp = DEC[decoder]
    if edges.size(1) == 2:
        relop = H.OP_NOOP
    B, (C, N), d = edges.size(0), dst_neg.shape, emb.size(1)
    W = H.LpWorkspace(relop, cmp, d, B, C, N, use_inverse, H.REDUCE_SUM, edges.size(1), True, dev)
    t = lambda x: None if x is None else x.to(dev)
    W.bind(t(emb), t(edges), t(dst_neg), t(src_neg), t(rel), t(inv) if use_inverse else None, None, None)
    W.forward()
    W.loss()
    W.backward()
    torch.cuda.synchronize()
    return W


def grad_report(got, want32, want64, what, rtol=RTOL):
    """Accumulated gradients are sums of hundreds of +- terms per entry, so small entries carry the rounding of the large terms
    that cancelled; the fp32 oracle itself (the reference's arithmetic) is off by a few 1e-5 of such an entry from the same
    oracle evaluated in float64, which is the yardstick here.  Asserted: worst pure-relative error <= 1e-4 over entries >= 0.1 max,
    <= 3e-4 over entries >= 0.01 max, absolute error <= 3e-6 max below that; the fp32 oracle's own numbers are printed next to ours."""
    got, w32, w64 = (t.detach().cpu().double().flatten() for t in (got, want32, want64))
    mx = max(w64.abs().max().item(), 1e-30)

    def stats(a):
        err = (a - w64).abs()
        out = []
        for floor in (0.1, 0.01):
            big = w64.abs() >= floor * mx
            out.append((err[big] / w64.abs()[big]).max().item() if bool(big.any()) else 0.0)
        small = w64.abs() < 0.01 * mx
        out.append((err[small].max().item() / mx) if bool(small.any()) else 0.0)
        return out

    h, r = stats(got), stats(w32)
    print("%-22s vs fp64 oracle  rel(>=.1max) %.2e [fp32 oracle %.2e]  rel(>=.01max) %.2e [%.2e]  abs/max below %.2e [%.2e]" % (
        what, h[0], r[0], h[1], r[1], h[2], r[2]))
    assert h[0] <= rtol and h[1] <= 3 * rtol and h[2] <= 3e-6, (what, h, r)


def check_against_oracle(H, dev, decoder, B, C, N, d, U, R, seed, use_inverse=True, edge_cols=3, scores=True):
    emb, edges, dst_neg, src_neg, rel, inv = make_batch(decoder, B, C, N, d, U, R, seed, edge_cols)
    want = O.train_batch(decoder, emb, torch.zeros(U, d), edges, dst_neg, src_neg, rel, inv if use_inverse else None)
    dbl = lambda t: None if t is None else t.double()
    want64 = O.train_batch(decoder, emb.double(), torch.zeros(U, d, dtype=torch.float64), edges, dst_neg, src_neg, dbl(rel),
                           dbl(inv) if use_inverse else None)
    W = run_hip(H, dev, decoder, emb, edges, dst_neg, src_neg, rel, inv, use_inverse)
    assert W.layout.Bp == want["pos"].numel()
    close_report(W.pos(0), want["pos"], "pos")
    if scores:
        close_report(W.neg(0), want["neg"], "neg")
    if use_inverse:
        close_report(W.pos(1), want["inv_pos"], "inv_pos")
        if scores:
            close_report(W.neg(1), want["inv_neg"], "inv_neg")
    close_report(W.loss_values()[0:1], want["loss"].reshape(1), "loss")
    occ_ids = torch.cat([edges[:, 0], edges[:, -1], src_neg.flatten(), dst_neg.flatten()])
    node_grad = torch.zeros(U, d, dtype=torch.float64).index_add_(0, occ_ids, W.gocc()[:, :d].cpu().double())
    grad_report(node_grad, want["node_grad"], want64["node_grad"], "node_grad")
    if edge_cols == 3:
        rg = torch.zeros(R, d, dtype=torch.float64).index_add_(0, edges[:, 1], W.grel(0)[:, :d].cpu().double())
        grad_report(rg, want["rel_grad"], want64["rel_grad"], "rel_grad")
        if use_inverse:
            ig = torch.zeros(R, d, dtype=torch.float64).index_add_(0, edges[:, 1], W.grel(1)[:, :d].cpu().double())
            grad_report(ig, want["inv_rel_grad"], want64["inv_rel_grad"], "inv_rel_grad")
    return W, want


def test_lp_bench_shape_matches_oracle(H, dev):
    """cfg2's batch through the default kernels: 12,800 score units over 768 persistent workgroups, 7 full + one 104-row tile
    per chunk, 16 x 16 tiles per chunk in the merged backward (comparators.cpp:7-20, decoder_methods.cpp:57-114)."""
    check_against_oracle(H, dev, "COMPLEX", 50000, 50, 1000, 100, 200000, 1000, seed=2024)


@pytest.mark.parametrize("variant", ["p", "r"])
@pytest.mark.parametrize("decoder", ["COMPLEX", "DISTMULT"])
def test_lp_multi_tile_shape_every_score_variant(H, dev, monkeypatch, variant, decoder):
    """Bc = 1024 (8 row tiles), N = 1000 (partial last column tile), several units per persistent workgroup."""
    monkeypatch.setenv("MARIUS_SCORES", variant)
    H.reload_env()
    check_against_oracle(H, dev, decoder, 4096, 4, 1000, 100, 9000, 17, seed=7)


def test_lp_multi_tile_shape_transe(H, dev):
    check_against_oracle(H, dev, "TRANSE", 4096, 4, 1000, 100, 9000, 17, seed=8)


@pytest.mark.parametrize("B,C,N", [(2048, 4, 512), (1000, 3, 700)])
def test_lp_d400_two_column_edges(H, dev, B, C, N):
    """cfg5's row shape (Twitter: ComplEx d=400, one relation type -> 2-column edges, no relation operator, one direction)."""
    check_against_oracle(H, dev, "COMPLEX", B, C, N, 400, 6000, 1, seed=11, use_inverse=False, edge_cols=2)


def test_lp_d400_three_column_edges(H, dev):
    check_against_oracle(H, dev, "COMPLEX", 1024, 4, 512, 400, 5000, 23, seed=12)


def test_train_step_bench_shape_matches_cpu_step(H, dev):
    """One whole fused step at B=50,000 / C=50 / N=1000 / d=100 on a 10 M-row table against oracle/cpu_step.py."""
    from marius_amd.lp_step import DeviceLinkPredictionStep
    from oracle.cpu_step import CpuLinkPredictionStep

    num_nodes, R, d, B, C, N, E, seed = 10_000_000, 14824, 100, 50000, 50, 1000, 200000, 42
    g = torch.Generator().manual_seed(3)
    table = (torch.rand(num_nodes, d, generator=g) - 0.5) * 0.4
    state = torch.rand(num_nodes, d, generator=g) * 0.01
    edges_all = torch.stack([torch.randint(num_nodes, (E,), generator=g), torch.randint(R, (E,), generator=g),
                             torch.randint(num_nodes, (E,), generator=g)], 1)
    t_d, s_d = table.to(dev), state.to(dev)
    cpu = CpuLinkPredictionStep("COMPLEX", table, state, R, B, C, N)   # updates table / state in place
    step = DeviceLinkPredictionStep("COMPLEX", num_nodes, R, d, B, C, N, seed=seed, device=dev, node_table=t_d, node_state=s_d)
    # Adagrad from an all-zero state moves a weight by lr * sign(g): discontinuous where a gradient component is rounding noise
    # around 0, so two correct fp32 implementations can differ by 2 lr there.  Start the relation sums slightly above zero on both
    # sides (the node state already is): the update is then a smooth function of g and comparable to tolerance.
    for t in (cpu.rel_sum, cpu.inv_rel_sum, step.rel_sum, step.inv_rel_sum):
        t.fill_(1e-3)
    torch.manual_seed(seed)
    perm_ref = torch.randperm(E)
    perm = step.gen.randperm_host(E)
    assert torch.equal(perm, perm_ref)
    e32 = edges_all.to(torch.int32).to(dev)
    touched = []
    for s in range(2):
        batch = edges_all[perm_ref[s * B:(s + 1) * B]]
        want = cpu.step(batch)
        edges = H.select_edges(e32, perm.to(dev), s * B, B)
        W = step.step(edges)
        torch.cuda.synchronize()
        assert torch.equal(step.last["src_neg"].cpu(), want["src_neg"])   # bit-exact sampled node indices
        assert torch.equal(step.last["dst_neg"].cpu(), want["dst_neg"])
        U = int(step.um.count.item())
        assert U == want["uniq"].numel() and torch.equal(step.um.uniq[:U].cpu(), want["uniq"])
        close_report(W.pos(0), want["pos"], "pos step %d" % s)
        close_report(W.neg(0), want["neg"], "neg step %d" % s)
        close_report(W.neg(1), want["inv_neg"], "inv_neg step %d" % s)
        close_report(W.loss_values()[0:1], want["loss"].reshape(1), "loss step %d" % s)
        touched.append(want["uniq"])
    rows = torch.unique(torch.cat(touched))
    # Updated parameters inherit the tolerance of the accumulated gradient they were stepped with (w' = w - lr g / (sqrt(s + g^2) + eps):
    # d w' / w' ~ d g / g while |lr g / sqrt(s)| >~ |w|, which holds in the first steps from a small init), i.e. the three-tier bound of
    # grad_report: 1e-4 over entries >= 0.1 max, 3e-4 over entries >= 0.01 max, 3e-6 max below.
    for got, want, what in ((t_d[rows.to(dev)], cpu.table[rows], "touched table rows"), (s_d[rows.to(dev)], cpu.state[rows], "touched state rows"),
                            (step.rel, cpu.rel, "relations"), (step.inv_rel, cpu.inv_rel, "inverse relations")):
        close_report(got, want, what, floor=0.1, atol_frac=1.0)
        close_report(got, want, what, rtol=3e-4, floor=0.01, atol_frac=3e-6)
    # untouched rows are untouched: spot-check a strided sample of the table bit for bit outside the touched set
    probe = torch.arange(0, num_nodes, 9973)
    mask = ~torch.isin(probe, rows)
    assert torch.equal(t_d[probe[mask].to(dev)].cpu(), cpu.table[probe[mask]])


def test_cpp_trainer_flash_pipeline_bench_shape_matches_cpu_step(H, dev):
    """the pipeline bench.py times, as a whole, at the bench shape — the C++ SynchronousTrainer with the flash decoder,
    node rows read in place from the table, the planned single-launch update, the loader stream a batch ahead and the permutation drawn
    ahead — 3 steps at B=50,000 / C=50 / N=1000 / d=100 on a 10 M-row table against oracle/cpu_step.py (trainer.cpp:106-138).
      * sampled ids and the unique map: bit-exact, read from an identical loader walking the same generator stream (exact_unique batches);
      * the trainer itself touched exactly the oracle's rows (rows whose Adagrad state changed == union of the oracle's unique ids);
      * loss of the last step, touched table / state rows, relation tables: close_report's three-tier bound."""
    import marius_amd
    from oracle.cpu_step import CpuLinkPredictionStep

    M = marius_amd.host()
    num_nodes, R, d, B, C, N, E, seed, steps = 10_000_000, 14824, 100, 50000, 50, 1000, 200000, 42, 3
    g = torch.Generator().manual_seed(3)
    table = (torch.rand(num_nodes, d, generator=g) - 0.5) * 0.4
    state = torch.rand(num_nodes, d, generator=g) * 0.01
    edges_all = torch.stack([torch.randint(num_nodes, (E,), generator=g), torch.randint(R, (E,), generator=g),
                             torch.randint(num_nodes, (E,), generator=g)], 1)
    t_d, s_d = table.to(dev), state.to(dev)
    s0_d = s_d.clone()
    e32 = edges_all.to(torch.int32).to(dev)

    def make(tab, st):
        gen = M.MariusGenerator(seed)
        emb, sta = M.InMemory(tab), M.InMemory(st)
        loader = M.DataLoader(M.InMemory(e32), emb, sta, M.CorruptNodeNegativeSampler(C, N, 0.0, False, M.LocalFilterMode.DEG, gen), gen, B, True)
        return loader

    cpu = CpuLinkPredictionStep("COMPLEX", table, state, R, B, C, N)   # updates table / state in place
    for t in (cpu.rel_sum, cpu.inv_rel_sum):  # see test_train_step_bench_shape_matches_cpu_step: Adagrad from an all-zero sum is discontinuous
        t.fill_(1e-3)
    torch.manual_seed(seed)
    perm_ref = torch.randperm(E)
    # ---- (1) ids through the C++ loader, exact-unique batches, no training
    ids_loader = make(t_d, s_d)
    ids_loader.initializeBatches(True)
    assert torch.equal(ids_loader.active_perm.cpu(), perm_ref)
    touched, want = [], None
    for s in range(steps):
        b = ids_loader.getBatch(True)
        want = cpu.step(edges_all[perm_ref[s * B:(s + 1) * B]])
        torch.cuda.synchronize()
        assert torch.equal(b.src_neg_indices.cpu(), want["src_neg"]) and torch.equal(b.dst_neg_indices.cpu(), want["dst_neg"])
        assert torch.equal(b.unique_node_indices.cpu(), want["uniq"])
        touched.append(want["uniq"])
    del ids_loader
    rows = torch.unique(torch.cat(touched))
    # ---- (2) the trainer as bench.py drives it
    loader = make(t_d, s_d)
    dec = M.ComplEx(R, d, dev, True, M.EdgeDecoderMethod.CORRUPT_NODE)
    model = M.Model(dec, M.SoftmaxCrossEntropy("sum"), M.LinkPredictionReporter(), dev)
    model.setup_optimizers(0.1)
    model.sparse_lr = 0.1
    ds = model.dense_state()
    assert len(ds) == 4 and all(tuple(t.shape) == (R, d) for t in ds)   # [relations, inverse relations, their Adagrad sums]
    for t in ds[2:]:
        assert float(t.abs().max()) == 0.0
        t.fill_(1e-3)
    trainer = M.SynchronousTrainer(loader, model)
    assert trainer.fused_update
    trainer.train_steps(steps)
    torch.cuda.synchronize()
    assert model.last_step_flash, "the bench pipeline is the flash path"
    assert model.ranges_valid and float(model.range_state[0]) >= float(t_d.abs().max()), "magnitude bound of the node table lost track"
    changed = (s_d != s0_d).any(1).nonzero().flatten().cpu()
    assert torch.equal(changed, rows), (changed.numel(), rows.numel())
    close_report(model.loss[0:1], want["loss"].reshape(1), "loss of step %d" % (steps - 1))
    for got, ref, what in ((t_d[rows.to(dev)], cpu.table[rows], "touched table rows"), (s_d[rows.to(dev)], cpu.state[rows], "touched state rows"),
                           (dec.relations, cpu.rel, "relations"), (dec.inverse_relations, cpu.inv_rel, "inverse relations")):
        # the same three tiers as the FP32-MFMA path (test_train_step_bench_shape_matches_cpu_step): the trainer tracks the tables' magnitude
        # bounds, so the flash contractions run on fp16 halves (22 significand bits per operand; measured here: 8.7e-5 at the 1e-2 floor,
        # 1.6e-6 of the maximum below it).  With bf16 halves (MARIUS_FLASH_F16=0) the second tier reads 5.6e-4 / 1.4e-5.
        close_report(got, ref, what, floor=0.1, atol_frac=1.0)
        close_report(got, ref, what, rtol=3e-4, floor=0.01, atol_frac=3e-6)
