"""TEST INFRASTRUCTURE (oracle/): numpy restatement of the fixed-capacity row exchange of the sharded node table — what
marius_a2a_capacity / marius_a2a_rows_post / the owner-side merge of the padded id runs compute (marius_amd/csrc/kernels/exchange.hip,
include/marius_hip.h).  Only tests/ import this; the product never does.

The reference has no counterpart to restate (its multi-GPU mode keeps the node table in one place and replicates the model,
src/cpp/src/pipeline/pipeline_gpu.cpp:23-80); what pins this file is (a) the partition rule it shards by — contiguous ranges of
ceil(num_nodes / world) rows, src/cpp/src/storage/storage.cpp:75 — and (b) the invariant every test holds it to: the rows a requester ends up
with are table[uniq], and the owners' updates equal the single-process update of the union batch (oracle/lp_oracle.py: accumulate_gradients,
src/cpp/src/data/batch.cpp:62-79)."""
import numpy as np


def capacity(max_rows, world, slack=1.5):
    """marius_a2a_capacity: slots per (requester, owner) pair"""
    if max_rows <= 0 or world <= 0:
        return 0
    if world == 1:
        return int(max_rows)
    slack = max(float(slack), 1.0)
    cap = int(max_rows / world * slack) + 1
    cap = (cap + 255) // 256 * 256
    return min(cap, int(max_rows))


def owner_offsets(uniq, shard_rows, world):
    """marius_owner_offsets: first position of every owner's ids in the ascending list `uniq`"""
    offs = np.searchsorted(uniq, np.arange(world + 1, dtype=np.int64) * shard_rows, side="left")
    offs[world] = len(uniq)
    return offs.astype(np.int64)


def post(uniq, shard_rows, world, cap):
    """marius_a2a_rows_post: (req_send [world * cap] — per owner: -1 padding FIRST, then its local ids ascending; place [U]; overflow)"""
    offs = owner_offsets(uniq, shard_rows, world)
    req = np.full(world * cap, -1, dtype=np.int64)
    place = np.zeros(len(uniq), dtype=np.int64)
    overflow = False
    for q in range(world):
        cnt = int(offs[q + 1] - offs[q])
        overflow |= cnt > cap
        c = min(cnt, cap)
        req[q * cap + cap - c:(q + 1) * cap] = uniq[offs[q]:offs[q] + c] - q * shard_rows
        place[offs[q]:offs[q] + c] = np.arange(q * cap + cap - c, (q + 1) * cap)
    return req, place, overflow


def merge_runs(recv_ids):
    """Owner side: stable sort of the received payload (what marius_merge_unique_runs computes from its `world` non-decreasing runs) ->
    (uniq incl. a leading -1 if any slot is padding, inverse, perm, seg_offsets)"""
    perm = np.argsort(recv_ids, kind="stable")
    srt = recv_ids[perm]
    head = np.ones(len(srt), dtype=bool)
    head[1:] = srt[1:] != srt[:-1]
    uniq = srt[head]
    seg = np.append(np.nonzero(head)[0], len(srt)).astype(np.int64)
    inverse = np.empty(len(srt), dtype=np.int64)
    inverse[perm] = np.cumsum(head) - 1
    return uniq, inverse, perm, seg


def owner_update(table, state, recv_ids, grads, lr, eps=1e-10):
    """The owner's segmented sum + sparse Adagrad over the received payload (negative ids = padding: neither read nor written), in place;
    float32, summation in payload order inside every segment (batch.cpp:62-79 for the rule)."""
    uniq, _, perm, seg = merge_runs(recv_ids)
    for u, rid in enumerate(uniq):
        if rid < 0:
            continue
        g = np.zeros(table.shape[1], dtype=np.float32)
        for p in perm[seg[u]:seg[u + 1]]:
            g = g + grads[p]
        s = state[rid] + g * g
        state[rid] = s
        table[rid] = table[rid] + (-lr * (g / (np.sqrt(s) + np.float32(eps)))).astype(np.float32)
    return uniq


# ---- the exchange header record (marius_a2a_publish / marius_a2a_record_checksum, exchange.hip) ---------------------------------------------
_M64 = (1 << 64) - 1


def _mix(h, v):
    h ^= (v + 0x9E3779B97F4A7C15 + ((h << 6) & _M64) + (h >> 2)) & _M64
    return (h * 0xFF51AFD7ED558CCD) & _M64


def record(offs, recv_counts=None, overflow=0, stamp=1):
    """int64 [2 world + 4]: stamp | offs[0..world] | recv counts (None: the send counts offs[q + 1] - offs[q]) | overflow | checksum"""
    offs = np.asarray(offs, dtype=np.int64)
    world = len(offs) - 1
    rc = np.diff(offs) if recv_counts is None else np.asarray(recv_counts, dtype=np.int64)
    words = np.concatenate([[stamp], offs, rc, [overflow]]).astype(np.int64)
    h = _mix(0x6D617269757361, int(stamp) & _M64)
    for v in words[1:]:
        h = _mix(h, int(v) & _M64)
    return np.concatenate([words, np.array([h], dtype=np.uint64).view(np.int64)])
