"""Swap cost of the HBM partition buffer (partition_buffer.cpp): per swap one partition leaves (D2H) and one arrives (H2D) through
pinned staging; with prefetching the file IO runs on the IO thread between swaps.

    python tools/bench_partition_swap.py [--partition-mb 512] [--partitions 8] [--capacity 4] [--d 100] [--dir /tmp]
Prints one JSON line per mode (prefetching on / off): ms per swap and the PCIe rate the two copies reach.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import marius_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--partition-mb", type=int, default=512)
    ap.add_argument("--partitions", type=int, default=8)
    ap.add_argument("--capacity", type=int, default=4)
    ap.add_argument("--d", type=int, default=100)
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--think-ms", type=float, default=400.0, help="time spent 'training' on a buffer state (gather/scatter traffic on the slab)")
    a = ap.parse_args()
    M = marius_amd.host()
    dev = torch.device("cuda", 0)
    rows_pp = a.partition_mb * (1 << 20) // (4 * a.d)
    total = rows_pp * a.partitions
    path = os.path.join(a.dir, "swap_bench_embeddings.bin")
    chunk = torch.rand(rows_pp, a.d)
    with open(path, "wb") as f:
        for _ in range(a.partitions):
            f.write(chunk.numpy().tobytes())
    part_bytes = rows_pp * a.d * 4
    for prefetching in (True, False):
        o = M.PartitionBufferOptions()
        o.num_partitions, o.buffer_capacity, o.prefetching, o.fine_to_coarse_ratio = a.partitions, a.capacity, prefetching, 1
        st = M.PartitionBufferStorage(path, total, a.d, o, dev)
        states, _ = M.getEdgeBucketOrdering(M.EdgeBucketOrdering.OLD_BETA, a.partitions, a.capacity, 1, 0, False, M.MariusGenerator(1))
        st.setBufferOrdering(states)
        t0 = time.perf_counter()
        st.load()
        torch.cuda.synchronize()
        load_s = time.perf_counter() - t0
        ids = torch.randint(st.getNumInMemory(), (200000,), device=dev)
        uniq = torch.unique(ids)
        vals = torch.ones(uniq.numel(), a.d, device=dev)
        times = []
        while st.hasSwap():
            t_end = time.perf_counter() + a.think_ms * 1e-3
            while time.perf_counter() < t_end:  # the buffer state's batches
                st.indexRead(ids)
                st.indexAdd(uniq, vals)
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            st.performNextSwap()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        st.unload(True)
        unload_s = time.perf_counter() - t0
        times.sort()
        med = times[len(times) // 2]
        print(json.dumps({"prefetching": prefetching, "partition_MB": a.partition_mb, "swaps": len(times), "prefetch_hits": 0 if not prefetching else None,
                          "ms_per_swap_median": round(med * 1e3, 2), "ms_per_swap_max": round(times[-1] * 1e3, 2),
                          "pcie_GBps_median": round(2 * part_bytes / med / 1e9, 2), "load_s": round(load_s, 2), "unload_write_s": round(unload_s, 2)}))
    os.remove(path)


if __name__ == "__main__":
    main()
