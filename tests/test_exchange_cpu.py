"""Fixed-capacity exchange of the sharded node table on CPU: the protocol (padded id payloads with -1 first, equal-split all-to-alls, owners
that merge / update from the payload alone) run by TWO gloo ranks over the numpy restatement of the kernels (oracle/exchange_oracle.py), against
the single-process update of the union batch.  The GPU tests hold the HIP kernels to the same restatement
(tests/test_gpu_parity.py::test_fixed_capacity_exchange_halves_against_numpy) and the C++ trainer to the same union-batch oracle
(tests/test_gpu_sharded2.py, MARIUS_EXCHANGE=fixed)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import exchange_oracle as X

CFG = dict(num_nodes=1001, d=8, L=96, steps=3, lr=0.1, slack=1.5)


def make_inputs(cfg, world):
    g = torch.Generator().manual_seed(5)
    table = ((torch.rand(cfg["num_nodes"], cfg["d"], generator=g) - 0.5) * 0.8).numpy()
    ids = [[torch.randint(cfg["num_nodes"], (cfg["L"],), generator=g).numpy() for _ in range(cfg["steps"])] for _ in range(world)]
    grads = [[((torch.rand(cfg["L"], cfg["d"], generator=g) - 0.5)).numpy().astype(np.float32) for _ in range(cfg["steps"])] for _ in range(world)]
    return table, ids, grads


def a2a(t):  # equal-split all-to-all: no split sizes anywhere
    out = torch.empty_like(t)
    dist.all_to_all_single(out, t)
    return out


def worker(rank, world, port, outdir, cfg):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table, ids, grads = make_inputs(cfg, world)
    S = (cfg["num_nodes"] + world - 1) // world
    lo, hi = rank * S, min((rank + 1) * S, cfg["num_nodes"])
    shard, state = table[lo:hi].copy(), np.zeros((hi - lo, cfg["d"]), dtype=np.float32)
    cap = X.capacity(cfg["L"], world, cfg["slack"])
    for s in range(cfg["steps"]):
        uniq, inverse = np.unique(ids[rank][s], return_inverse=True)   # map_tensors: ascending unique ids + per-occurrence index
        req, place, overflow = X.post(uniq, S, world, cap)
        assert not overflow
        req_recv = a2a(torch.from_numpy(req)).numpy()
        rows_send = np.zeros((world * cap, cfg["d"]), dtype=np.float32)
        ok = req_recv >= 0                                             # the gather skips padding slots
        rows_send[ok] = shard[req_recv[ok]]
        rows_recv = a2a(torch.from_numpy(rows_send)).numpy()
        assert np.array_equal(rows_recv[place], make_rows(table, shard, lo, hi, uniq, world, cfg, s, rank, outdir))   # the payload read in place through the slots
        # per-unique-row gradients straight into the owners' slot order (marius_segment_sum_rows_planned with out_rows = place)
        gsum = np.zeros((len(uniq), cfg["d"]), dtype=np.float32)
        for p in range(cfg["L"]):
            gsum[inverse[p]] = gsum[inverse[p]] + grads[rank][s][p]
        grad_send = np.full((world * cap, cfg["d"]), np.nan, dtype=np.float32)   # unused slots must never be read by an owner
        grad_send[place] = gsum
        grad_recv = a2a(torch.from_numpy(grad_send)).numpy()
        X.owner_update(shard, state, req_recv, grad_recv, cfg["lr"])
        dist.barrier()
    np.save(os.path.join(outdir, "shard%d.npy" % rank), shard)
    np.save(os.path.join(outdir, "state%d.npy" % rank), state)
    dist.destroy_process_group()


def make_rows(table, shard, lo, hi, uniq, world, cfg, s, rank, outdir):
    """what the requester must hold: the CURRENT rows of its unique ids — known to the test only through a side channel (every rank dumps its
    shard before the fetch); here: gathered again over gloo, the slow obvious way"""
    S = (cfg["num_nodes"] + world - 1) // world
    shards = [None] * world
    dist.all_gather_object(shards, shard)
    full = np.concatenate(shards, 0)
    return full[uniq]


def simulate(cfg, world):
    table, ids, grads = make_inputs(cfg, world)
    table, state = table.copy(), np.zeros_like(table)
    for s in range(cfg["steps"]):
        # union batch: per rank its per-unique-row sums (payload order = requester order), then one Adagrad step per node
        allid, allg = [], []
        for r in range(world):
            uniq, inverse = np.unique(ids[r][s], return_inverse=True)
            gsum = np.zeros((len(uniq), cfg["d"]), dtype=np.float32)
            for p in range(cfg["L"]):
                gsum[inverse[p]] = gsum[inverse[p]] + grads[r][s][p]
            allid.append(uniq)
            allg.append(gsum)
        X.owner_update(table, state, np.concatenate(allid), np.concatenate(allg, 0), cfg["lr"])
    return table, state


@pytest.mark.parametrize("world", [2])
def test_fixed_capacity_exchange_world2_gloo_equals_union_batch_update(world):
    port = 43000 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(worker, args=(world, port, outdir, CFG), nprocs=world, join=True)
        shards = [np.load(os.path.join(outdir, "shard%d.npy" % r)) for r in range(world)]
        states = [np.load(os.path.join(outdir, "state%d.npy" % r)) for r in range(world)]
    table, state = simulate(CFG, world)
    got_t, got_s = np.concatenate(shards, 0), np.concatenate(states, 0)
    assert (got_s > 0).any(1).sum() > 100
    assert np.array_equal((got_s > 0).any(1), (state > 0).any(1)), "the set of updated rows differs"
    np.testing.assert_allclose(got_t, table, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(got_s, state, rtol=1e-6, atol=1e-9)


def test_post_layout_and_capacity_rule():
    rng = np.random.default_rng(3)
    assert X.capacity(200000, 1) == 200000 and X.capacity(200000, 8, 1.5) == 37632 and X.capacity(96, 2, 1.5) == 96
    for world, L, nodes in [(1, 50, 40), (3, 400, 1000), (8, 1600, 5000)]:
        S = (nodes + world - 1) // world
        uniq = np.unique(rng.integers(0, nodes, L))
        cap = X.capacity(L, world, 1.5)
        req, place, over = X.post(uniq, S, world, cap)
        assert not over and len(req) == world * cap
        for q in range(world):
            blk = req[q * cap:(q + 1) * cap]
            assert np.all(np.diff(blk) >= 0), "a block must be a non-decreasing run (-1 padding first)"
            real = blk[blk >= 0]
            assert np.array_equal(real + q * S, uniq[(uniq >= q * S) & (uniq < (q + 1) * S)])
        assert np.array_equal(req[place] + (place // cap) * S, uniq)
        # the owner's merge of the runs: padding collapses into ONE leading segment with id -1
        u, inv, perm, seg = X.merge_runs(req[:cap] if world == 1 else np.concatenate([req[:cap], req[:cap]]))
        assert (u[0] == -1) == bool((req[:cap] < 0).any()) and np.all(u[1:] >= 0) and np.all(np.diff(u) > 0)
    # overflow is reported, never silent
    uniq = np.arange(0, 300, dtype=np.int64)  # all of them owned by shard 0 of 2
    req, place, over = X.post(uniq, 1000, 2, 256)
    assert over


def test_exchange_header_record_checksum_host_function_matches_restatement():
    """marius_a2a_record_checksum is a HOST function of libmarius_hip.so (no GPU needed): it must equal the numpy restatement of the record the
    publish kernel writes, and any single changed word — a stale split point, a torn read — must fail it."""
    import torch

    from marius_amd import hip as H
    from oracle import exchange_oracle as X

    L = H.lib()
    rng = np.random.default_rng(7)
    for world in (1, 2, 8, 64):
        assert L.marius_a2a_record_words(world) == 2 * world + 4
        cuts = np.sort(rng.integers(0, 200000, world - 1)) if world > 1 else np.zeros(0, dtype=np.int64)
        offs = np.concatenate([[0], cuts, [200000]]).astype(np.int64)
        for recv in (None, rng.integers(0, 50000, world)):
            rec = X.record(offs, recv, overflow=int(world == 8), stamp=1234567 + world)
            t = torch.from_numpy(rec.copy())
            want = int(np.array([rec[-1]], dtype=np.int64).view(np.uint64)[0])
            assert L.marius_a2a_record_checksum(t.data_ptr(), world) == want
            assert H.a2a_record_ok(t, world)
            for w in range(2 * world + 3):  # every word before the checksum is covered, the stamp included
                bad = t.clone()
                bad[w] += 1
                assert not H.a2a_record_ok(bad, world), w
