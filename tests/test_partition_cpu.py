"""Out-of-core row, CPU side: the partition-buffer / ordering ORACLE against the expectations the reference's own
tests hold (test/cpp/unit/test_buffer.cpp:241-318, restated as literals) and against vectors generated from the reference's Python
edge partitioner (tests/golden/partition_edges.json)."""
import itertools
import json
import os

import numpy as np
import pytest
import torch

from oracle import partition_oracle as P

HERE = os.path.dirname(os.path.abspath(__file__))

# test_buffer.cpp:21-87 (PartitionBufferTest fixture): 45 rows in 5 partitions of 10, capacity 2, and this sequence of buffer states
REF_STATES = [[0, 1], [0, 2], [0, 3], [0, 4], [1, 4], [1, 3], [1, 2], [2, 3], [2, 4], [3, 4]]
# test_buffer.cpp:241-256: (admit, evict) per swap
REF_SWAPS = [([2], [1]), ([3], [2]), ([4], [3]), ([1], [0]), ([3], [4]), ([2], [3]), ([3], [1]), ([4], [3]), ([3], [2])]


def make_buffer(tmp_path, d=7, total=45):
    g = torch.Generator().manual_seed(5)
    table = torch.rand(total, d, generator=g).numpy()
    path = str(tmp_path / "embeddings.bin")
    P.write_table(path, table)
    pb = P.PartitionBufferOracle(2, 5, 10, d, total, path)
    pb.set_buffer_ordering(REF_STATES)
    pb.load()
    return pb, table, path


def test_swap_sequence_matches_reference_test(tmp_path):
    pb, _, _ = make_buffer(tmp_path)
    for admit, evict in REF_SWAPS:
        assert pb.has_swap()
        assert pb.next_admit() == admit and pb.next_evict() == evict
        pb.perform_next_swap()
    assert not pb.has_swap()


def test_global_map_matches_reference_test(tmp_path):  # test_buffer.cpp:309-318
    pb, _, _ = make_buffer(tmp_path)
    exp = -np.ones(45, dtype=np.int64)
    exp[0:20] = np.arange(20)
    assert np.array_equal(pb.global_to_local_map(True), exp)
    exp[10:20] = -1
    exp[20:30] = np.arange(10, 20)
    assert np.array_equal(pb.global_to_local_map(False), exp)


def test_read_add_sync_follow_index_select_semantics(tmp_path):  # test_buffer.cpp:270-307
    pb, table, path = make_buffer(tmp_path)
    ids = np.array([0, 3, 19, 10, 7])
    assert np.array_equal(pb.index_read(ids), table[ids])  # state [0, 1]: local == global for the first 20 rows
    vals = np.arange(5 * 7, dtype=np.float32).reshape(5, 7)
    pb.index_add(ids, vals)
    table[ids] += vals
    assert np.array_equal(pb.index_read(ids), table[ids])
    # swaps carry the update to the file and back; the short last partition leaves a zero tail in its slot
    for _ in range(3):
        pb.perform_next_swap()          # state [0, 4]: partition 4 (5 rows) in slot 1
    assert np.array_equal(pb.index_read(np.arange(10, 15)), table[40:45])
    assert not pb.index_read(np.arange(15, 20)).any()
    pb.unload(True)
    on_disk = np.fromfile(path, dtype=np.float32).reshape(45, 7)
    assert np.array_equal(on_disk, table)


@pytest.mark.parametrize("p,c", [(4, 2), (8, 4), (8, 3), (16, 8), (5, 2), (6, 5)])
def test_beta_ordering_invariants(p, c):
    torch.manual_seed(p * 100 + c)
    states = P.beta_ordering(p, c)
    assert all(len(set(s)) == c and all(0 <= x < p for x in s) for s in states)
    for a, b in zip(states, states[1:]):   # one partition exchanged per step (slots are re-shuffled between rounds: compare as sets)
        assert len(set(b) - set(a)) == 1
    together = {frozenset(q) for s in states for q in itertools.combinations(s, 2)}
    assert together == {frozenset(q) for q in itertools.combinations(range(p), 2)}
    buckets = P.greedy_assign(states, p)
    flat = [b for bs in buckets for b in bs]
    assert len(flat) == p * p and len(set(flat)) == p * p
    assert all(s in st and t in st for st, bs in zip(states, buckets) for s, t in bs)


def test_beta_ordering_consumes_the_generator_like_the_reference_call_sequence():
    """randperm(p), then per round randperm(|buffer|), randperm(|disk|), ..., randperm(|disk|): the same seed gives the same states
    whether the draws come from torch's global generator or are replayed in that order."""
    torch.manual_seed(11)
    a = P.beta_ordering(8, 3)
    torch.manual_seed(11)
    calls = []

    def rp(n):
        calls.append(n)
        return torch.randperm(n).tolist()

    assert P.beta_ordering(8, 3, rp) == a
    assert calls[:4] == [8, 3, 5, 5]


@pytest.mark.parametrize("random_assign", [False, True])
def test_comet_ordering_invariants(random_assign):
    torch.manual_seed(3)
    rng = np.random.default_rng(0)
    states, buckets = P.two_level_beta_ordering(8, 4, fine_to_coarse_ratio=2, randomly_assign=random_assign,
                                                choose=lambda k: int(rng.integers(k)))
    assert all(len(set(s)) == 4 for s in states)
    for a, b in zip(states, states[1:]):   # a coarse partition (two fine ones) is exchanged per step
        assert len(set(b) - set(a)) == 2
    flat = [b for bs in buckets for b in bs]
    assert len(flat) == 64 and len(set(flat)) == 64
    assert all(s in st and t in st for st, bs in zip(states, buckets) for s, t in bs)


def test_partition_edges_matches_reference_python():
    with open(os.path.join(HERE, "golden", "partition_edges.json")) as f:
        cases = json.load(f)
    for c in cases:
        edges, sizes = P.partition_edges(torch.tensor(c["edges"]), c["num_nodes"], c["num_partitions"])
        assert edges.tolist() == c["sorted_edges"]
        assert sizes == c["bucket_sizes"]


def test_active_edges_cover_every_edge_once_with_local_ids(tmp_path):
    torch.manual_seed(9)
    n, p, c = 43, 5, 2   # partition size ceil(43 / 5) = 9, last partition 7 rows
    e = torch.stack([torch.randint(n, (300,)), torch.randint(3, (300,)), torch.randint(n, (300,))], 1)
    edges, sizes = P.partition_edges(e, n, p)
    states = P.beta_ordering(p, c)
    buckets = P.greedy_assign(states, p)
    table = np.zeros((n, 2), dtype=np.float32)
    path = str(tmp_path / "t.bin")
    P.write_table(path, table)
    pb = P.PartitionBufferOracle(c, p, 9, 2, n, path)
    pb.set_buffer_ordering(states)
    pb.load()
    seen = 0
    for i, bs in enumerate(buckets):
        if i > 0:
            pb.perform_next_swap()
        act = P.active_edges_for_state(edges, sizes, bs, pb.global_to_local_map(True), p)
        assert (act[:, 0] >= 0).all() and (act[:, -1] >= 0).all() and (act[:, [0, -1]] < pb.num_in_memory()).all()
        seen += act.size(0)
    assert seen == 300 and not pb.has_swap()


@pytest.mark.parametrize("p,c,ratio,random_assign", [(8, 4, 1, False), (8, 4, 2, False), (16, 8, 2, True), (5, 2, 1, True), (12, 6, 3, True)])
def test_host_ordering_equals_oracle_under_the_same_seed(p, c, ratio, random_assign):
    """The C++ host function (partition_buffer.cpp, no GPU needed: pure host logic on the MT19937 restatement) and the oracle on
    torch's CPU generator produce the same buffer states and edge-bucket assignment from the same seed."""
    import marius_amd

    M = marius_amd.host()
    states, buckets = M.getEdgeBucketOrdering(M.EdgeBucketOrdering.COMET, p, c, ratio, 0, random_assign, M.MariusGenerator(123))
    torch.manual_seed(123)
    o_states, o_buckets = P.two_level_beta_ordering(p, c, ratio, 0, random_assign, choose=lambda k: int(torch.randperm(k)[0]))
    assert [s.tolist() for s in states] == o_states
    assert [[tuple(x) for x in b.tolist()] for b in buckets] == o_buckets
