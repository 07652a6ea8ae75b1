"""One epoch of out-of-core training (PartitionBufferStorage: slab in HBM, swaps over PCIe) next to the same epoch with the table in
DEVICE_MEMORY, Freebase86m-shaped batches on a smaller table so that the files fit a scratch disk.

    python tools/bench_partition_train.py [--nodes 10000000] [--edges 5000000] [--partitions 8] [--capacity 4] [--dir /tmp]
Prints one JSON line: positive edges/s in both modes, number of swaps and the time spent in them.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import marius_amd  # noqa: E402


def bucket_sort_edges(cols, src, dst, ps, p, piece=1 << 24):
    """int32 edge list sorted (stably) by edge bucket (src // ps) * p + dst // ps, and the p * p bucket sizes.

    Sorted piece by piece and stitched per bucket: on this stack one `torch.sort(stable=True)` + row gather over >= 10^8 rows returned a list
    that was NOT bucket-sorted (tools/probe_torch_sort.py) — which is what the round-2 "stall" at 260 M edges was: edges whose endpoints
    were not in the buffer.  The result is verified before it is handed out."""
    E = src.numel()
    pieces, counts = [], []
    for lo in range(0, E, piece):
        hi = min(E, lo + piece)
        b = (src[lo:hi] // ps) * p + dst[lo:hi] // ps
        order = torch.sort(b, stable=True)[1]
        pieces.append(torch.stack([c[lo:hi] for c in cols], 1).index_select(0, order).to(torch.int32))
        counts.append(torch.bincount(b, minlength=p * p))
    counts_h = torch.stack(counts).cpu()                        # [pieces, p * p]
    starts = torch.cumsum(counts_h, 1) - counts_h
    parts = [pieces[i].narrow(0, int(starts[i, k]), int(counts_h[i, k])) for k in range(p * p) for i in range(len(pieces)) if counts_h[i, k] > 0]
    edges = torch.cat(parts)
    del pieces, parts
    sizes = counts_h.sum(0).tolist()
    eb = (edges[:, 0].long() // ps) * p + edges[:, -1].long() // ps
    if not bool((eb[1:] >= eb[:-1]).all()) or torch.bincount(eb, minlength=p * p).tolist() != sizes:
        raise RuntimeError("synthetic edge list is not bucket-sorted")
    return edges, sizes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--edges", type=int, default=5_000_000)
    ap.add_argument("--partitions", type=int, default=8)
    ap.add_argument("--capacity", type=int, default=4)
    ap.add_argument("--d", type=int, default=100)
    ap.add_argument("--relations", type=int, default=14824, help="1: a single relation type, 2-column edges, one direction (cfg5: Twitter-2010)")
    ap.add_argument("--dir", default="/tmp", help="/dev/shm keeps the partition files in host DRAM (cfg5's setting)")
    ap.add_argument("--skip-device-memory", action="store_true")
    ap.add_argument("--only-device-memory", action="store_true")
    ap.add_argument("--prefetch", type=int, default=1)
    ap.add_argument("--max-steps", type=int, default=0, help="stop the out-of-core epoch after this many steps (diagnosis runs)")
    a = ap.parse_args()
    M = marius_amd.host()
    dev = torch.device("cuda", 0)
    R, B, C, N, d, p = a.relations, 50000, 50, 1000, a.d, a.partitions
    g = torch.Generator(device=dev).manual_seed(1)
    src = torch.randint(a.nodes, (a.edges,), generator=g, device=dev)
    dst = torch.randint(a.nodes, (a.edges,), generator=g, device=dev)
    rel = torch.randint(R, (a.edges,), generator=g, device=dev)
    ps = -(-a.nodes // p)
    cols = [src, rel, dst] if R > 1 else [src, dst]            # io.cpp:42-45: one relation type -> (src, dst)
    edges, sizes = bucket_sort_edges(cols, src, dst, ps, p)    # torch_partitioner.py:12-46 on the device: stable sort by edge bucket
    del src, dst, rel, cols
    trace = bool(os.environ.get("PB_TRACE"))
    T0 = time.perf_counter()
    if trace:  # a run that stops making progress prints where every Python thread is
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("PB_TRACE_DUMP_AFTER", "45")), exit=False)

    def tr_print(*x):
        if trace:
            print("[%.1f s]" % (time.perf_counter() - T0), *x, file=sys.stderr, flush=True)

    tr_print("edges built")

    def model():
        dec = M.ComplEx(R, d, dev, R > 1, M.EdgeDecoderMethod.CORRUPT_NODE)
        m = M.Model(dec, M.SoftmaxCrossEntropy("sum"), M.LinkPredictionReporter(), dev)
        m.setup_optimizers(0.1)
        m.sparse_lr = 0.1
        return m

    out = {"nodes": a.nodes, "edges": a.edges, "d": d, "partitions": p, "capacity": a.capacity, "table_GB": round(a.nodes * d * 4 / 1e9, 2)}
    # ---- in memory
    if not a.skip_device_memory:
        gen = M.MariusGenerator(7)
        emb, st = M.InMemory(torch.zeros((a.nodes, d), device=dev).uniform_(-0.01, 0.01)), M.InMemory(torch.zeros((a.nodes, d), device=dev))
        loader = M.DataLoader(M.InMemory(edges), emb, st, M.CorruptNodeNegativeSampler(C, N, 0.0, False, M.LocalFilterMode.DEG, gen), gen, B, True)
        tr = M.SynchronousTrainer(loader, model())
        tr.train(1)
        out["device_memory_edges_per_s"] = round(tr.last_edges_per_second, 1)
        out["device_memory_epoch_s"] = round(tr.last_epoch_seconds, 2)
        del tr, loader, emb, st
        torch.cuda.empty_cache()
        tr_print("device-memory epoch done", out["device_memory_edges_per_s"])
    if a.only_device_memory:
        print(json.dumps(out))
        return
    # ---- partition buffer
    paths = [os.path.join(a.dir, n) for n in ("pb_bench_embeddings.bin", "pb_bench_state.bin")]
    rows = 1 << 20
    t_files = time.perf_counter()
    with open(paths[0], "wb") as fe, open(paths[1], "wb") as fs:
        for lo in range(0, a.nodes, rows):
            n = min(rows, a.nodes - lo)
            fe.write(torch.zeros((n, d), device=dev).uniform_(-0.01, 0.01).cpu().numpy().tobytes())
            fs.write(bytes(4 * d * n))
    out["file_init_s"] = round(time.perf_counter() - t_files, 1)
    tr_print("files written")
    o = M.PartitionBufferOptions()
    o.num_partitions, o.buffer_capacity, o.prefetching, o.fine_to_coarse_ratio = p, a.capacity, bool(a.prefetch), 1
    o.edge_bucket_ordering = M.EdgeBucketOrdering.NEW_BETA
    emb, st = M.PartitionBufferStorage(paths[0], a.nodes, d, o, dev), M.PartitionBufferStorage(paths[1], a.nodes, d, o, dev)
    gen = M.MariusGenerator(7)
    est = M.InMemory(edges)
    est.edge_bucket_sizes = sizes
    loader = M.DataLoader(est, emb, st, M.CorruptNodeNegativeSampler(C, N, 0.0, False, M.LocalFilterMode.DEG, gen), gen, B, True)
    pb_model = model()
    tr = M.SynchronousTrainer(loader, pb_model)
    # epoch by hand (== SynchronousTrainer::train(1)) so that the parts can be timed: per-state batch layout (host randperm of the state's
    # edges + global -> buffer-local id remap), the training steps, the swaps, and the final write-back of the resident partitions
    t0 = time.perf_counter()
    loader.loadStorage()
    t_load = time.perf_counter() - t0
    tr_print("loadStorage done")
    parts = {"load_first_state_s": round(t_load, 2)}
    tl = time.perf_counter()
    loader.initializeBatches(True)
    torch.cuda.synchronize()
    t_init0 = time.perf_counter() - tl
    tr_print("first initializeBatches done")
    te = time.perf_counter()
    steps = 0
    while loader.hasNextBatch():
        tr.train_one(True)
        steps += 1
        if trace and (steps % 100 == 0 or steps <= 3):
            torch.cuda.synchronize()
            tr_print("steps", steps, "swaps", emb.swaps, "ahead hits/misses", loader.shuffle_ahead_hits, loader.shuffle_ahead_misses)
        if a.max_steps and steps >= a.max_steps:
            break
    torch.cuda.synchronize()
    t_train = time.perf_counter() - te
    tw = time.perf_counter()
    loader.nextEpoch(True)
    t_unload = time.perf_counter() - tw
    wall = time.perf_counter() - t0
    tr_last = a.edges / (t_train + t_unload)
    parts.update({"first_state_batch_layout_s": round(t_init0, 2), "steps": steps, "train_loop_s_incl_swaps_and_layouts": round(t_train, 2),
                  "final_write_back_s": round(t_unload, 2), "swap_exchange_s_embeddings": round(emb.swap_seconds, 3), "swap_exchange_s_state": round(st.swap_seconds, 3),
                  "device_drain_at_swap_points_s": round(emb.drain_seconds + st.drain_seconds, 3)})
    out["partition_buffer_breakdown"] = parts
    out.update({"partition_buffer_edges_per_s": round(tr_last, 1), "partition_buffer_epoch_s": round(t_train + t_unload, 2),
                "epoch_wall_s_incl_load_and_writeback": round(wall, 2), "parameters_GB": round(2 * a.nodes * d * 4 / 1e9, 1),
                "resident_GB": round(2 * a.capacity * (-(-a.nodes // p)) * d * 4 / 1e9, 1), "swap_GB_each_way_per_swap": round(2 * (-(-a.nodes // p)) * d * 4 / 1e9, 2),
                "swaps": emb.swaps, "prefetch_hits": emb.prefetch_hits, "swap_seconds_embeddings": round(emb.swap_seconds, 3),
                "swap_seconds_state": round(st.swap_seconds, 3), "buffer_states": len(loader.buffer_states),
                # what the last step packed its operand records with: "fp16" = 22 significand bits per operand (slab bound), "bf16" = 16, "none" = FP32-MFMA path
                "operand_records": pb_model.last_step_records, "flash_path": bool(pb_model.last_step_flash),
                "dtype": "f32 (contractions: 2-way %s split x 3 products, f32 accumulate)" % pb_model.last_step_records if pb_model.last_step_flash else "f32"})
    for pth in paths:
        os.remove(pth)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
