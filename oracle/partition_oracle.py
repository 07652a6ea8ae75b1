"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the reference's out-of-core node storage —
the partition buffer, the edge-bucket orderings that drive it and the per-buffer-state epoch loop.

Reference: src/cpp/src/storage/buffer.cpp (PartitionedFile :64-120, PartitionBuffer :324-713), src/cpp/src/data/ordering.cpp
(getBetaOrderingHelper :78-129, greedyAssignEdgeBucketsToBuffers :131-152, randomlyAssignEdgeBucketsToBuffers :154-243,
getTwoLevelBetaOrdering :245-297), src/cpp/src/storage/graph_storage.cpp:334-470 (initializeInMemorySubGraph),
src/cpp/src/data/dataloader.cpp:120-183 (setActiveEdges), src/python/tools/preprocess/converters/partitioners/torch_partitioner.py:12-46.

Pinned by: the admit / evict sequences, the global->local map and the read / add / sync expectations of the reference's own
test/cpp/unit/test_buffer.cpp:241-318 (tests/test_partition_cpu.py holds them as literals), and for partition_edges by vectors generated
from the reference's Python function (tests/golden/partition_edges.json, made by tests/golden/make_partition_golden.py).
**The exact sequence of buffer states a seed produces is parity-unpinned**: ordering.cpp includes reporting/logger.h -> spdlog (an empty
submodule), so it cannot be compiled here; beta_ordering() issues the same randperm calls in the same order (so a seeded run would
agree) and is otherwise pinned by the invariants the algorithm exists for (every partition pair co-resident at least once, one
partition exchanged per step, every edge bucket assigned exactly once to a state holding both of its partitions).
"""
import os

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------------------- partition buffer
class PartitionBufferOracle:
    """buffer.cpp:324-713 with a numpy slab; ids passed to index_read / index_add are buffer-local, as in the reference."""

    def __init__(self, capacity, num_partitions, partition_size, embedding_size, total_embeddings, filename):
        self.capacity, self.num_partitions, self.partition_size = capacity, num_partitions, partition_size
        self.d, self.total, self.filename = embedding_size, total_embeddings, filename
        # partition table (buffer.cpp:345-362): the last partition may be short
        self.offsets = [min(i * partition_size, total_embeddings) for i in range(num_partitions)]
        self.sizes = [partition_size] * num_partitions
        self.sizes[-1] = total_embeddings - self.offsets[-1]
        self.buffer_idx = [-1] * num_partitions
        self.present = [False] * num_partitions
        self.states, self.cursor, self.state = [], 0, None
        self.slab = None

    def _file(self, mode):
        return np.memmap(self.filename, dtype=np.float32, mode=mode, shape=(self.total, self.d))

    def _read(self, p, slot):
        lo = slot * self.partition_size
        self.slab[lo:lo + self.partition_size] = 0.0                                   # readPartition memsets first (:92)
        self.slab[lo:lo + self.sizes[p]] = self._file("r")[self.offsets[p]:self.offsets[p] + self.sizes[p]]
        self.present[p], self.buffer_idx[p] = True, slot

    def _write(self, p):
        lo = self.buffer_idx[p] * self.partition_size
        f = self._file("r+")
        f[self.offsets[p]:self.offsets[p] + self.sizes[p]] = self.slab[lo:lo + self.sizes[p]]
        f.flush()
        self.slab[lo:lo + self.partition_size] = 0.0                                   # writePartition(clear_mem=true) (:115)

    def set_buffer_ordering(self, states):                                             # :488-497
        self.states = [[int(x) for x in s] for s in states]
        self.state, self.cursor = self.states[0], 1
        if self.slab is not None:
            self.unload(True)
            self.load()

    def load(self):                                                                    # :372-418
        self.slab = np.zeros((self.capacity * self.partition_size, self.d), dtype=np.float32)
        for slot, p in enumerate(self.state):
            self._read(p, slot)

    def has_swap(self):
        return self.cursor < len(self.states)

    def next_admit(self):                                                              # :549-567: in the order of the NEXT state
        if not self.has_swap():
            return []
        return [p for p in self.states[self.cursor] if p not in self.state]

    def next_evict(self):                                                              # :569-585: in the order of the CURRENT state
        return [p for p in self.state if p not in self.states[self.cursor]]

    def perform_next_swap(self):                                                       # :501-547
        if self.state is None or not self.has_swap():
            return
        evict, admit = self.next_evict(), self.next_admit()
        slots = [self.buffer_idx[p] for p in evict]
        self.state = self.states[self.cursor]
        self.cursor += 1
        for p in evict:                                                                # evict (:637-652); buffer_idx_ is left as is
            self._write(p)
            self.present[p] = False
        for p, slot in zip(admit, slots):                                              # admit into the freed slots, pairwise (:654-686)
            self._read(p, slot)

    def global_to_local_map(self, get_current=True):                                   # :587-635
        m = -np.ones(self.total, dtype=np.int64)
        if get_current:
            for p in self.state:
                b = self.buffer_idx[p] * self.partition_size
                m[self.offsets[p]:self.offsets[p] + self.sizes[p]] = np.arange(b, b + self.sizes[p])
            return m
        evict, admit = self.next_evict(), self.next_admit()
        for p in self.states[self.cursor]:
            if self.buffer_idx[p] != -1:   # note: an evicted-earlier partition keeps its stale slot until the admit loop below overwrites it
                b = self.buffer_idx[p] * self.partition_size
                m[self.offsets[p]:self.offsets[p] + self.sizes[p]] = np.arange(b, b + self.sizes[p])
        for a, e in zip(admit, evict):
            b = self.buffer_idx[e] * self.partition_size
            m[self.offsets[a]:self.offsets[a] + self.sizes[a]] = np.arange(b, b + self.sizes[a])
        return m

    def index_read(self, ids):                                                         # :434-448
        return self.slab[np.asarray(ids)]

    def index_add(self, ids, values):                                                  # :453-475 (ids unique)
        self.slab[np.asarray(ids)] += np.asarray(values, dtype=np.float32)

    def sync(self):                                                                    # :688-699
        for p in range(self.num_partitions):
            if self.present[p]:
                self._write(p)
                self.present[p], self.buffer_idx[p] = False, -1

    def unload(self, write):                                                           # :420-439
        if self.slab is not None:
            if write:
                self.sync()
            self.slab = None

    def num_in_memory(self):                                                           # buffer.h:189
        return self.capacity * self.partition_size


# ---------------------------------------------------------------------------------------------------------------- orderings
def beta_ordering(num_partitions, buffer_capacity, randperm=None):
    """ordering.cpp:78-129.  `randperm(n)` -> sequence of n ints (default: torch.randperm on the global CPU generator, the
    reference's source of randomness).  Returns the list of buffer states."""
    rp = (lambda n: torch.randperm(n).tolist()) if randperm is None else randperm
    allp = list(rp(num_partitions))
    in_buffer = allp[:buffer_capacity]
    on_disk = sorted(set(allp) - set(in_buffer))        # _unique2(sorted) of the ids that occur once in cat(all, in_buffer)
    states = [list(in_buffer)]
    while len(on_disk) >= 1:
        in_buffer = [in_buffer[i] for i in rp(len(in_buffer))]
        on_disk = [on_disk[i] for i in rp(len(on_disk))]
        for i in range(len(on_disk)):                   # cycle every on-disk partition through the last slot
            on_disk[i], in_buffer[-1] = in_buffer[-1], on_disk[i]
            states.append(list(in_buffer))
        on_disk = [on_disk[i] for i in rp(len(on_disk))]
        replaced = 0
        for i in range(buffer_capacity - 1):            # then refill the other slots from disk; those partitions are done
            if i >= len(on_disk):
                break
            replaced += 1
            in_buffer[i] = on_disk[i]
            states.append(list(in_buffer))
        on_disk = on_disk[replaced:]
    return states


def greedy_assign(states, num_partitions):
    """ordering.cpp:131-152: a bucket goes to the first state that holds both of its partitions."""
    seen = np.zeros((num_partitions, num_partitions), dtype=bool)
    out = []
    for st in states:
        cur = []
        for s in st:
            for t in st:
                if not seen[s, t]:
                    seen[s, t] = True
                    cur.append((s, t))
        out.append(cur)
    return out


def random_assign(states, num_partitions, choose):
    """ordering.cpp:154-243: every bucket picks uniformly among the states that hold both partitions (the reference draws with libc
    rand_r per OpenMP thread: not reproducible, so `choose(k)` -> int in [0, k) is injected); buckets are listed per state in
    ascending bucket-id order."""
    out = [[] for _ in states]
    for s in range(num_partitions):
        for t in range(num_partitions):
            options = [i for i, st in enumerate(states) if s in st and t in st]
            out[options[choose(len(options))]].append((s, t))
    return out


def two_level_beta_ordering(num_partitions, buffer_capacity, fine_to_coarse_ratio=1, num_cache_partitions=0, randomly_assign=False,
                            randperm=None, choose=None):
    """ordering.cpp:245-297 (COMET = BETA over coarse partitions, each a random group of `ratio` fine partitions)."""
    rp = (lambda n: torch.randperm(n).tolist()) if randperm is None else randperm
    cp = num_partitions // fine_to_coarse_ratio - num_cache_partitions
    cc = buffer_capacity // fine_to_coarse_ratio - num_cache_partitions
    coarse = beta_ordering(cp, cc, rp)
    cached_fine = num_cache_partitions * fine_to_coarse_ratio
    fine_map = list(range(cached_fine)) + [x + cached_fine for x in rp(num_partitions - cached_fine)]
    states = []
    for st in coarse:
        st = [x + num_cache_partitions for x in st] + list(range(num_cache_partitions))
        fine = []
        for c in st:
            fine += fine_map[c * fine_to_coarse_ratio:(c + 1) * fine_to_coarse_ratio]
        states.append(fine)
    buckets = random_assign(states, num_partitions, choose) if randomly_assign else greedy_assign(states, num_partitions)
    return states, buckets


# ---------------------------------------------------------------------------------------------------------------- edge buckets
def partition_edges(edges, num_nodes, num_partitions):
    """torch_partitioner.py:12-46: stable sort by (src partition, dst partition); offsets = bucket sizes, row-major [p*p]."""
    ps = int(np.ceil(num_nodes / num_partitions))
    src_p = torch.div(edges[:, 0], ps, rounding_mode="trunc")
    dst_p = torch.div(edges[:, -1], ps, rounding_mode="trunc")
    _, dst_args = torch.sort(dst_p, stable=True)
    _, src_args = torch.sort(src_p[dst_args], stable=True)
    order = dst_args[src_args]
    edges = edges[order]
    bucket = torch.div(edges[:, 0], ps, rounding_mode="trunc") * num_partitions + torch.div(edges[:, -1], ps, rounding_mode="trunc")
    sizes = torch.bincount(bucket, minlength=num_partitions * num_partitions)
    return edges, sizes.tolist()


def active_edges_for_state(edges, bucket_sizes, buckets, g2l, num_partitions):
    """dataloader.cpp:120-175 + graph_storage.cpp:334-440: the edges of the buckets assigned to this buffer state, in assignment
    order, with node ids mapped to buffer-local rows (relation column untouched)."""
    starts = np.concatenate([[0], np.cumsum(bucket_sizes)])
    parts = [edges[starts[s * num_partitions + t]:starts[s * num_partitions + t + 1]] for s, t in buckets]
    act = torch.cat(parts) if parts else edges[:0]
    m = torch.as_tensor(g2l)
    out = act.clone()
    out[:, 0] = m[act[:, 0]]
    out[:, -1] = m[act[:, -1]]
    return out


def write_table(path, table):
    np.asarray(table, dtype=np.float32).tofile(path)
    return os.path.getsize(path)
