"""The C++ ShardedTrainer with TWO ranks on the GPU box.  RCCL refuses two ranks on one device, so both processes share cuda:0 and every
collective (ids / rows / gradients all-to-all(v), count exchange, relation all-reduce) goes through gloo — the trainer only sees c10d
process groups by name, so the schedule, the split sizes, the owner-side merge of the per-sender runs and the cross-rank dedupe are
exactly the code that runs over RCCL.  Expected result: the synchronous union-batch update of the CPU oracle."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lp_oracle as O
from oracle.cpu_step import CpuLinkPredictionStep
from tolerance import tiers, well_conditioned

pytestmark = pytest.mark.gpu

CFG = dict(decoder="COMPLEX", num_nodes=1501, R=5, d=16, B=48, C=3, N=20, E=480, steps=5, seed=17, lr=0.1)
# the same on the flash decoder path (d in (16, 128]): the rows a rank scores come from every rank's shard, and the operand records must
# still carry fp16 halves (22 significand bits) — bounded by the gathered copy itself (Batch::row_bound_), not by any one rank's table
CFG_FLASH = dict(CFG, d=32, N=40)


def make_inputs(cfg, world=2):
    g = torch.Generator().manual_seed(3)
    table = (torch.rand(cfg["num_nodes"], cfg["d"], generator=g) - 0.5) * 0.8
    edges = [torch.stack([torch.randint(cfg["num_nodes"], (cfg["E"],), generator=g), torch.randint(cfg["R"], (cfg["E"],), generator=g),
                          torch.randint(cfg["num_nodes"], (cfg["E"],), generator=g)], 1) for _ in range(world)]
    return table, edges


def worker(rank, world, port, outdir, sync_interval, staleness, cfg=CFG, exchange="exact"):
    import marius_amd
    from marius_amd.sharded import shard_range

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", MARIUS_EXCHANGE=exchange)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    side = dist.new_group(backend="gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    M = marius_amd.host()
    table, edges = make_inputs(cfg, world)
    lo, hi = shard_range(cfg["num_nodes"], rank, world)
    tb, sb = table[lo:hi].clone().to(dev), torch.zeros(hi - lo, cfg["d"], device=dev)
    gen = M.MariusGenerator(cfg["seed"] + rank)
    sampler = M.CorruptNodeNegativeSampler(cfg["C"], cfg["N"], 0.0, False, M.LocalFilterMode.DEG, gen)
    stub = M.InMemory("", cfg["num_nodes"], cfg["d"], torch.float32, dev)
    loader = M.DataLoader(M.InMemory(edges[rank].to(torch.int32).to(dev)), stub, None, sampler, gen, cfg["B"], True)
    dec = M.ComplEx(cfg["R"], cfg["d"], dev, True, M.EdgeDecoderMethod.CORRUPT_NODE)
    model = M.Model(dec, M.SoftmaxCrossEntropy("sum"), M.LinkPredictionReporter(), dev)
    model.setup_optimizers(cfg["lr"])
    model.sparse_lr = cfg["lr"]
    tr = M.ShardedTrainer(loader, model, tb, sb, rank, world, cfg["num_nodes"], dist.group.WORLD.group_name, side.group_name, staleness, sync_interval)
    tr.train_steps(cfg["steps"])
    tr.finish()
    assert tr.fixed_capacity == (exchange == "fixed")
    torch.save({"shard": tb.cpu(), "state": sb.cpu(), "rel": dec.relations.cpu(), "inv_rel": dec.inverse_relations.cpu(), "flash": bool(model.last_step_flash),
                "records": model.last_step_records}, os.path.join(outdir, "r%d.pt" % rank))
    dist.destroy_process_group()


def simulate(cfg, world=2, staleness=0):
    """Every rank reads the same table state, gradients are summed per node over all ranks' batches, Adagrad is applied once; relation
    gradients are summed (all-reduce) before the dense step.  Rank r's generator: seed + r, first draw = the epoch permutation.
    staleness k: the rows of step s + k (of every rank) are read before the update of step s is applied."""
    table, edges = make_inputs(cfg, world)
    state = torch.zeros_like(table)
    steppers = []
    for r in range(world):
        st = CpuLinkPredictionStep(cfg["decoder"], table, state, cfg["R"], cfg["B"], cfg["C"], cfg["N"])
        st.num_nodes = cfg["num_nodes"]
        steppers.append(st)
    rel, inv = steppers[0].rel, steppers[0].inv_rel
    rel_sum, inv_sum = torch.zeros_like(rel), torch.zeros_like(inv)
    rng, perms = [], []
    for r in range(world):
        torch.manual_seed(cfg["seed"] + r)
        perms.append(torch.randperm(cfg["E"]))
        rng.append(torch.get_rng_state())

    def fetch(s):
        out = []
        for r in range(world):
            torch.set_rng_state(rng[r])
            batch = edges[r][perms[r][s * cfg["B"]:(s + 1) * cfg["B"]]]
            src_neg, _ = steppers[r].get_negatives(batch, True)
            dst_neg, _ = steppers[r].get_negatives(batch, False)
            rng[r] = torch.get_rng_state()
            uniq, mapped = O.map_tensors([batch[:, 0], batch[:, -1], src_neg.flatten(), dst_neg.flatten()])
            el = torch.stack([mapped[0], batch[:, 1], mapped[1]]).transpose(0, 1)
            out.append((uniq, el, mapped[3].reshape(dst_neg.shape), mapped[2].reshape(src_neg.shape), O.index_read(table, uniq)))
        return out

    queue = [fetch(k) for k in range(staleness + 1)] if staleness else None  # batches 0..staleness are read before the first update
    for s in range(cfg["steps"]):
        if staleness:
            cur = queue.pop(0)
        else:
            cur = fetch(s)
        ids, gs, rg, ig = [], [], torch.zeros_like(rel), torch.zeros_like(inv)
        for uniq, el, dst_map, src_map, emb in cur:
            out = O.train_batch(cfg["decoder"], emb, torch.zeros_like(emb), el, dst_map, src_map, rel, inv)
            ids.append(uniq)
            gs.append(out["node_grad"])
            rg += out["rel_grad"]
            ig += out["inv_rel_grad"]
        allid, allg = torch.cat(ids), torch.cat(gs)
        uniq, invx = torch.unique(allid, return_inverse=True)
        g = torch.zeros(uniq.numel(), allg.size(1)).index_add_(0, invx, allg)
        dw, ds = O.accumulate_gradients(g, O.index_read(state, uniq), cfg["lr"])
        O.index_add(table, uniq, dw)
        O.index_add(state, uniq, ds)
        O.dense_adagrad_step(rel, rg, rel_sum, cfg["lr"])
        O.dense_adagrad_step(inv, ig, inv_sum, cfg["lr"])
        if staleness:
            queue.append(fetch(s + staleness + 1))  # read after update s, before update s + 1 (the trainer runs `staleness` batches past the last step too)
    return table, state, rel, inv


@pytest.mark.parametrize("world,staleness,flash,exchange", [(2, 0, False, "exact"), (2, 1, False, "exact"), (2, 2, False, "exact"), (8, 1, False, "exact"),
                                                            (2, 0, True, "exact"), (2, 1, True, "exact"), (8, 1, True, "exact"),
                                                            (2, 0, False, "fixed"), (2, 1, True, "fixed"), (8, 1, True, "fixed"), (8, 0, False, "fixed")])
def test_cpp_sharded_trainer_ranks_equal_union_batch_update(world, staleness, flash, exchange):
    from marius_amd.sharded import shard_range

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = CFG_FLASH if flash else CFG
    port = 41000 + 2000 * staleness + 100 * world + 37 * int(flash) + 19 * int(exchange == "fixed") + os.getpid() % 1000
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(worker, args=(world, port, outdir, 1, staleness, cfg, exchange), nprocs=world, join=True)
        res = [torch.load(os.path.join(outdir, "r%d.pt" % r)) for r in range(world)]
    table, state, rel, inv = simulate(cfg, world, staleness)
    shared = 0
    for r in range(world):
        lo, hi = shard_range(cfg["num_nodes"], r, world)
        assert res[r]["flash"] == flash
        if flash:
            assert res[r]["records"] == "fp16", "the sharded step fell back to bf16 operand halves"
        # the tiers of the single-GPU training path (tests/tolerance.py): 1e-4 over entries >= 0.1 max, 3e-4 over >= 0.01 max, 3e-6 max below
        touched = (res[r]["state"] > 0).any(1)
        assert torch.equal(touched, (state[lo:hi] > 0).any(1)), "shard %d: the set of updated rows differs" % r
        assert torch.equal(res[r]["shard"][~touched], table[lo:hi][~touched])
        ok = well_conditioned(state[lo:hi])[touched]   # zero initial Adagrad state: elements whose gradients were rounding noise are left out
        assert float(ok.float().mean()) > 0.99
        tiers(res[r]["shard"][touched][ok], table[lo:hi][touched][ok], "shard %d rows" % r)
        tiers(res[r]["state"][touched][ok], state[lo:hi][touched][ok], "shard %d Adagrad state" % r)
        tiers(res[r]["rel"], rel, "relations (rank %d)" % r)
        tiers(res[r]["inv_rel"], inv, "inverse relations (rank %d)" % r)
        shared += int(touched.sum())
    assert all(torch.equal(res[0]["rel"], res[r]["rel"]) for r in range(1, world))  # replicas of the relation tables stay identical
    assert shared > 0
    if staleness:  # and the stale trajectory differs from the synchronous one
        assert not torch.allclose(simulate(cfg, world, 0)[0], table, rtol=1e-4, atol=1e-6)


def overflow_worker(rank, world, port, outdir):
    import marius_amd
    from marius_amd.sharded import shard_range

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo", MARIUS_EXCHANGE="fixed", MARIUS_EXCHANGE_SLACK="1.0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    side = dist.new_group(backend="gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    M = marius_amd.host()
    num_nodes, R, d, B, C, N, E = 100000, 5, 16, 600, 3, 300, 1800
    g = torch.Generator().manual_seed(11 + rank)
    table = (torch.rand(num_nodes, d, generator=torch.Generator().manual_seed(3)) - 0.5) * 0.8
    # rank 1's batches are ordinary; rank 0's edges have BOTH endpoints in shard 0: with the uniform negatives that is ~2100 of its 3000 ids for
    # one owner, more than the 1536 slots a (requester, owner) pair owns at slack 1.0
    hi_id = num_nodes if rank == 1 else num_nodes // 2
    edges = torch.stack([torch.randint(hi_id, (E,), generator=g), torch.randint(R, (E,), generator=g), torch.randint(hi_id, (E,), generator=g)], 1)
    lo, hi = shard_range(num_nodes, rank, world)
    tb, sb = table[lo:hi].clone().to(dev), torch.zeros(hi - lo, d, device=dev)
    gen = M.MariusGenerator(5 + rank)
    sampler = M.CorruptNodeNegativeSampler(C, N, 0.0, False, M.LocalFilterMode.DEG, gen)
    loader = M.DataLoader(M.InMemory(edges.to(torch.int32).to(dev)), M.InMemory("", num_nodes, d, torch.float32, dev), None, sampler, gen, B, True)
    dec = M.ComplEx(R, d, dev, True, M.EdgeDecoderMethod.CORRUPT_NODE)
    model = M.Model(dec, M.SoftmaxCrossEntropy("sum"), M.LinkPredictionReporter(), dev)
    model.setup_optimizers(0.1)
    model.sparse_lr = 0.1
    tr = M.ShardedTrainer(loader, model, tb, sb, rank, world, num_nodes, dist.group.WORLD.group_name, side.group_name, 1, 1)
    msg, again = "", ""
    try:
        tr.train_steps(3)
    except Exception as e:  # noqa: BLE001
        msg = str(e)
    try:
        tr.step()
    except Exception as e:  # noqa: BLE001
        again = str(e)
    torch.cuda.synchronize()
    torch.save({"msg": msg, "again": again, "steps": tr.steps, "cap": tr.pair_capacity, "shard_unchanged": bool(torch.equal(tb.cpu(), table[lo:hi])),
                "state_zero": bool((sb == 0).all())}, os.path.join(outdir, "r%d.pt" % rank))
    del tr
    dist.destroy_process_group()


def test_fixed_capacity_overflow_is_refused_by_every_rank_before_anything_is_applied():
    """a batch that asks one owner for more rows than a pair's capacity used to be flagged only on the overflowing rank, read
    RING steps later — after the rows beyond the capacity had been scored from slot 0 and their gradients applied to a real row — while the other
    rank blocked in the next all-to-all until the c10d watchdog fired.  Now the flag is max-reduced over the ranks on the preparation stream and
    travels in the published header: BOTH ranks raise at that batch, before any of its payloads is exchanged, and no shard row has changed."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    port = 43000 + os.getpid() % 1000
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(overflow_worker, args=(2, port, outdir), nprocs=2, join=True)
        res = [torch.load(os.path.join(outdir, "r%d.pt" % r)) for r in range(2)]
    for r in range(2):
        assert "planned maximum" in res[r]["msg"] and "batch 0" in res[r]["msg"], res[r]["msg"]  # rank 1 did not overflow itself: it raises all the same
        assert "earlier step failed" in res[r]["again"]
        assert res[r]["steps"] == 0 and res[r]["cap"] == 1536
        assert res[r]["shard_unchanged"] and res[r]["state_zero"]


def test_bench_launch_contract_two_ranks():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` as the driver launches it, with the collectives on gloo so
    that both ranks can share this box's single GPU (MARIUS_BENCH_BACKEND, testing only): one JSON line from rank 0 with the contract's keys."""
    import json
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MARIUS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(47000 + os.getpid() % 2000), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--cpu-seconds", "3",
           "--num-nodes", "2000000"]
    out = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["warmup"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert "C++ ShardedTrainer" in j["config"]["host"]
    # the N > 1 line is complete: bounded CPU leg, roofline of the dominant kernel, the communicator it ran on, bytes on the wire
    assert j["cpu_baseline"] and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] >= 1 and j["cpu_baseline"]["kind"] == "port"
    assert j["roofline"] and 0 < j["roofline"]["frac"] < 1 and j["roofline"]["bound"] == "mfma"
    assert j["ranks"] == 2 and j["collective_backend"] == "gloo" and j["rccl_ranks"] == 0  # no RCCL communicator in this emulation
    xb = j["exchange_bytes_per_step"]
    assert xb["ids"] > 0 and xb["rows"] > 0 and xb["gradients"] == xb["rows"] and xb["total"] == xb["ids"] + xb["rows"] + xb["gradients"]
    # strong scaling: the global batch stays B
    cmd2 = cmd[:-2] + ["--num-nodes", "2000000", "--strong", "--no-cpu-baseline"]
    cmd2[cmd2.index("--master-port") + 1] = str(49000 + os.getpid() % 2000)
    out2 = subprocess.run(cmd2, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert out2.returncode == 0, out2.stderr[-2000:]
    j2 = json.loads([ln for ln in out2.stdout.splitlines() if ln.startswith("{")][0])
    assert j2["scaling"] == "strong" and "B=25000 per GPU" in j2["config"]["workload"] and j2["cpu_baseline"] is None
