"""Node-embedding table sharded across the GPUs of one node, one process per GPU, RCCL over xGMI.

Partition axis = Marius's own: shard p of P owns the contiguous id range [p*S, min((p+1)*S, num_nodes)), S = ceil(num_nodes / P)
(reference src/cpp/src/storage/storage.cpp:75, storage/buffer.cpp:340-356,
src/python/tools/preprocess/converters/partitioners/torch_partitioner.py:13-16).  The reference never shards a DEVICE_MEMORY table
(storage.cpp:567-569 keeps it on one device; its multi-GPU mode replicates the dense model and all-reduces relation gradients,
nn/model.cpp:136-159) — this exchange is new design along that axis.

One synchronous data-parallel step on every rank (each rank trains its own batch):
  1. getBatch locally: edge slice, negatives (generator seeded random_seed + rank), sort/unique  -> ascending unique ids.
     Ascending order == grouped by owner, so the all-to-all split points are P+1 lower bounds (marius_owner_offsets).
  2. all-to-all(v) of the id lists, owners gather the requested rows from their shard, all-to-all(v) of the rows back.
  3. forward / loss / backward locally; per-unique-row gradient by atomic-free segmented sum.
  4. all-to-all(v) of the row gradients to the owners; the owner sort/uniques everything it received (a row may be requested
     by several ranks), sums per row and applies sparse Adagrad once:  exactly the single-GPU update for the union batch.
  5. relation tables are replicated.  sync_interval == 1: their gradients are all-reduced (sum) and every replica takes the same
     dense Adagrad step (model.cpp:136-159).  sync_interval == K > 1 (the reference's pipeline.gpu_sync_interval, default 16,
     with gpu_model_average: pipeline_gpu.cpp:52-80): every replica steps on its own gradient and every K-th step the tables and
     their Adagrad sums are averaged across ranks — 1/K of the all-reduce volume.
Only the exchange lives here; every local operation goes through a backend (HIP kernels in production, the oracle in the
CPU gloo tests) so the N>1 plumbing is testable without a GPU while the product path still has no CPU fallback.
"""
import math
import os

import torch
import torch.distributed as dist


def shard_rows(num_nodes, world):
    return (num_nodes + world - 1) // world


def shard_range(num_nodes, rank, world):
    S = shard_rows(num_nodes, world)
    lo = min(rank * S, num_nodes)
    return lo, min(lo + S, num_nodes)


def exchange_splits(send_offsets, group=None):
    """send_offsets: host int64 [P+1] split points of this rank's sorted unique ids. Returns (send_counts, recv_counts) lists."""
    send = (send_offsets[1:] - send_offsets[:-1]).to(torch.int64)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    s = send.to(dev)
    r = torch.empty_like(s)
    dist.all_to_all_single(r, s, group=group)
    return send.tolist(), r.cpu().tolist()


def a2a_rows(buf, send_counts, recv_counts, group=None, out=None):
    """all-to-all(v) along dim 0 of a [n, ...] tensor."""
    if out is None:
        out = buf.new_empty((sum(recv_counts),) + tuple(buf.shape[1:]))
    dist.all_to_all_single(out, buf.contiguous(), recv_counts, send_counts, group=group)
    return out


def sharded_step(backend, edges, rank, world, num_nodes, group=None, sync_interval=1, step_index=0):
    """One synchronous step. `backend` provides the local operations (see HipBackend below)."""
    S = shard_rows(num_nodes, world)
    lo, _ = shard_range(num_nodes, rank, world)
    ctx = backend.get_batch(edges)                              # sample + unique
    offs = backend.owner_offsets(ctx, S, world)                 # host [P+1] (the one host sync of the step)
    uniq = backend.unique_ids(ctx)                              # [U] ascending global ids (exact size)
    send_counts, recv_counts = exchange_splits(offs, group)
    req_ids = a2a_rows(uniq, send_counts, recv_counts, group)   # ids other ranks (and this one) need from my shard
    rows = backend.gather_local(req_ids - lo)                   # [sum(recv), d]
    emb = a2a_rows(rows, recv_counts, send_counts, group)       # my batch's rows, in unique-id order
    grad, rel_grads, loss = backend.compute(ctx, emb)           # grad [U, d]
    recv_grad = a2a_rows(grad, send_counts, recv_counts, group)
    backend.apply_local(req_ids - lo, recv_grad)                # dedupe across senders + Adagrad + scatter on my shard
    if sync_interval <= 1:
        for g in rel_grads:
            if g is not None:
                dist.all_reduce(g, group=group)
        backend.dense_step(rel_grads)
    else:
        backend.dense_step(rel_grads)  # local step; replicas drift for at most sync_interval steps
        if (step_index + 1) % sync_interval == 0:
            for t in backend.dense_state():
                dist.all_reduce(t, group=group)
                t.div_(world)
    return loss


class HipBackend:
    """Local operations on the MI355X through the C-ABI (no CPU fallback)."""

    def __init__(self, stepper, shard_table, shard_state):
        from . import hip as H

        self.H, self.s = H, stepper
        self.table, self.state = shard_table, shard_state
        self.um_recv = None
        self.carry_recv = None

    def get_batch(self, edges):
        s, H = self.s, self.H
        B, CN = s.B, s.C * s.N
        src_neg, dst_neg, sdeg, ddeg = s.sample(edges)
        ctx = {"edges": edges, "src_neg": src_neg, "dst_neg": dst_neg, "filters": (None, None)}
        if s.n_deg > 0:
            ctx["filters"] = (H.deg_filter(ddeg, B), H.deg_filter(sdeg, B))
        st = H.stream_ptr()
        H.check(H.lib().marius_assemble_ids(H.ptr(edges), B, s.edge_cols, H.ptr(src_neg), H.ptr(dst_neg), CN, H.ptr(s.all_ids), st), "assemble")
        s.um.run(s.all_ids, s.key_bits)
        H.check(H.lib().marius_remap_edges(H.ptr(edges), H.ptr(s.um.inverse), B, s.edge_cols, H.ptr(s.edges_local), st), "remap")
        return ctx

    def owner_offsets(self, ctx, S, world):
        offs = self.H.owner_offsets(self.s.um, S, world)  # host copy: the one sync per step
        ctx["U"] = int(offs[-1])
        return offs

    def unique_ids(self, ctx):
        if "U" not in ctx:
            ctx["U"] = int(self.s.um.count.item())
        return self.s.um.uniq[: ctx["U"]]

    def gather_local(self, local_ids):
        return self.H.gather_rows(self.table, local_ids)

    def compute(self, ctx, emb):
        s, H = self.s, self.H
        B, CN, d = s.B, s.C * s.N, s.d
        W = s.W
        src_map = s.um.inverse[2 * B: 2 * B + CN]
        dst_map = s.um.inverse[2 * B + CN: 2 * B + 2 * CN]
        W.bind(emb, s.edges_local, dst_map, src_map, s.rel, s.inv_rel, ctx["filters"][0], ctx["filters"][1])
        W.forward()
        W.loss()
        W.backward()
        grad = torch.empty((ctx["U"], d), dtype=torch.float32, device=emb.device)
        H.segment_sum_rows(W.gocc(), s.um, s.L, d, grad, carry=s.carry)
        s.rel_ids.copy_(ctx["edges"][:, 1])
        s.um_rel.run(s.rel_ids, s.rel_bits)
        s.rel_grad.zero_()
        H.segment_sum_rows(W.grel(0), s.um_rel, B, d, s.rel_grad, out_rows=s.um_rel.uniq, carry=s.carry_rel)
        inv = None
        if s.inverse:
            s.inv_rel_grad.zero_()
            H.segment_sum_rows(W.grel(1), s.um_rel, B, d, s.inv_rel_grad, out_rows=s.um_rel.uniq, carry=s.carry_rel)
            inv = s.inv_rel_grad
        return grad, [s.rel_grad, inv], W.loss_values()[0]

    def apply_local(self, local_ids, grads, recv_counts=None):
        H, s = self.H, self.s
        n = local_ids.numel()
        if n == 0:
            return
        if self.um_recv is None or self.um_recv.cap < n:
            cap = max(n, int(1.5 * s.L))
            self.um_recv = H.UniqueMap(cap, local_ids.device)
            self.carry_recv = H.segment_carry(cap, s.d, local_ids.device)
        bits = max(1, math.ceil(math.log2(self.table.size(0) + 1)))
        if recv_counts is not None and len(recv_counts) <= 64:
            # every sender's list is ascending and duplicate-free: merge the runs instead of radix-sorting them (bit-equal result)
            import ctypes

            offs = [0]
            for c in recv_counts:
                offs.append(offs[-1] + int(c))
            um, ids = self.um_recv, local_ids.contiguous()
            H.check(H.lib().marius_merge_unique_runs(H.ptr(ids), n, (ctypes.c_int64 * len(offs))(*offs), len(recv_counts), H.ptr(um.uniq), H.ptr(um.inverse),
                                                     H.ptr(um.perm), H.ptr(um.seg), H.ptr(um.count), H.ptr(um.ws), um.ws_bytes, H.stream_ptr()),
                    "merge_unique_runs")
        else:
            self.um_recv.run(local_ids.contiguous(), bits)
        H.segment_adagrad_scatter(grads, self.um_recv, n, s.d, self.table, self.state, s.sparse_lr, carry=self.carry_recv)

    def dense_state(self):
        s = self.s
        return [t for t in (s.rel, s.inv_rel, s.rel_sum, s.inv_rel_sum) if t is not None]

    def dense_step(self, rel_grads):
        s, H = self.s, self.H
        H.dense_adagrad_step(s.rel, s.rel_sum, rel_grads[0], s.dense_lr)
        if rel_grads[1] is not None:
            H.dense_adagrad_step(s.inv_rel, s.inv_rel_sum, rel_grads[1], s.dense_lr)


class _NullEvent:
    """Event stand-in for a backend without streams (the CPU test backend)."""

    def record(self, stream=None):
        pass

    def synchronize(self):
        pass


class PipelineSchedule:
    """Control flow of the pipelined / overlapped sharded step, independent of where the local work runs.

    Subclasses provide the local operations (`PipelinedShardedTrainer` below: HIP kernels and HIP streams; the CPU gloo tests: the
    oracle, no streams) — the exchange logic that only matters for world > 1 (split sizes, the order of the all-to-alls, slot reuse,
    staleness) is this class and is therefore covered by a world-size-2 test without a GPU.

    staleness = 0 (synchronous, the `sync` variant of SURVEY.md 8e): exactly sharded_step — every row a batch reads carries all
    earlier updates — but the batch preparation (edge slice, negatives, sort/unique, owner split points) runs one step ahead on a
    side stream: it never reads the table, so nothing changes, and the host read-back an all-to-all(v) needs (split points /
    receive counts) is served from work that finished a step ago instead of draining the main stream.

    staleness = 1 (the overlapped exchange of SURVEY.md 8e): additionally the row fetch of batch t+1 (ids all-to-all, owner gather,
    rows all-to-all) and the gradient return + owner update of batch t run on an exchange stream underneath the scoring of
    batch t / t+1.  Rows of batch t+1 are therefore read after the update of batch t-1 and before the update of batch t: a
    staleness of exactly one step, inside the bound the reference's own multi-GPU trainer runs with (its multi-GPU mode exists only
    as the asynchronous pipeline, pipeline_gpu.cpp:23-80, staleness_bound default 16, marius_config.py:673-676).
    Stream order on the exchange stream:  fetch(t+1) | grads(t), update(t) | fetch(t+2) | ...
    """

    RING = 4

    def __init__(self, rank, world, num_nodes, d, device, sync_interval=1, group=None, side_group=None, staleness=0):
        assert staleness in (0, 1)
        self.rank, self.world, self.num_nodes, self.d = rank, world, num_nodes, d
        self.S = shard_rows(num_nodes, world)
        self.lo, _ = shard_range(num_nodes, rank, world)
        self.sync_interval = sync_interval
        self.group, self.side_group = group, side_group
        self.staleness = staleness
        self.dev = device
        self.next_prepared = 0
        self.next_fetched = 0
        self.step_index = 0
        self._pool = {}
        self.slots = None  # the subclass creates RING slots

    # ---- hooks -----------------------------------------------------------------------------------------------------------------
    def _on(self, which):                      # context manager: run the enclosed work on stream "main" | "xchg"
        import contextlib
        return contextlib.nullcontext()

    def _wait(self, which, event):             # stream `which` waits for `event`
        pass

    def _record(self, event, which):
        event.record()

    def _retire_buffer(self, buf):             # a pooled buffer is being replaced while other streams may still read it
        pass

    def _prepare(self, t):                     # everything that does not read the table; ends by recording slot.ready
        raise NotImplementedError

    def _split_points(self, slot):             # host list [world + 1] of the slot's sorted unique ids (may block on slot.ready)
        raise NotImplementedError

    def _unique_ids(self, slot, U):
        raise NotImplementedError

    def _gather_local(self, local_ids, out):
        raise NotImplementedError

    def _compute(self, t):                     # forward / loss / backward of slot t: sets slot.grad [U, d], returns relation grads or None
        raise NotImplementedError

    def _apply_local(self, local_ids, grads, recv_counts=None):  # recv_counts: rows per sender (each sender's ids ascend)
        raise NotImplementedError

    def _dense_step(self, rel_grads):
        raise NotImplementedError

    def _dense_state(self):
        raise NotImplementedError

    def _loss(self):
        raise NotImplementedError

    # ---- generic part ----------------------------------------------------------------------------------------------------------
    def _buf(self, name, n, tail, dtype):
        """[n, *tail] view of a grow-only buffer: per-step sizes vary with the number of unique ids, and a fresh torch.empty per
        step keeps the caching allocator splitting/merging blocks (visible as multi-100-us jitter)."""
        b = self._pool.get(name)
        if b is None or b.size(0) < n:
            if b is not None:
                self._retire_buffer(b)
            b = torch.empty((max(n + n // 4, 1),) + tuple(tail), dtype=dtype, device=self.dev)
            self._pool[name] = b
        return b[:n]

    def _slot(self, t):
        return self.slots[t % self.RING]

    def _prepare_through(self, t):
        while self.next_prepared <= t:
            self._prepare(self.next_prepared)
            self.next_prepared += 1

    # stage 2 (host + exchange stream): split sizes, then ids -> owners, rows -> requesters
    def _fetch(self, t):
        slot = self._slot(t)
        offs = self._split_points(slot)  # of a batch prepared at least one step ago — no stream drains for this
        slot.send_counts = [offs[i + 1] - offs[i] for i in range(self.world)]
        if self.world > 1:
            # counts travel over the CPU (gloo) group: a second RCCL communicator on another stream could share a hardware queue with
            # the main one and order differently on different ranks; `world` integers over loopback cost less than that risk
            recv = torch.empty(self.world, dtype=torch.int64)
            dist.all_to_all_single(recv, torch.tensor(slot.send_counts, dtype=torch.int64), group=self.side_group)
            slot.recv_counts = recv.tolist()
        else:
            slot.recv_counts = list(slot.send_counts)
        U, nrecv, d = offs[-1], sum(slot.recv_counts), self.d
        slot.U, slot.nrecv = U, nrecv
        k = t % self.RING
        with self._on("xchg"):
            self._wait("xchg", slot.ready)
            req_ids = a2a_rows(self._unique_ids(slot, U), slot.send_counts, slot.recv_counts, self.group, out=self._buf("req", nrecv, (), torch.int64))
            slot.local_ids = torch.sub(req_ids, self.lo, out=self._buf("local%d" % k, nrecv, (), torch.int64))
            rows = self._gather_local(slot.local_ids, self._buf("rows", nrecv, (d,), torch.float32))
            slot.emb = a2a_rows(rows, slot.recv_counts, slot.send_counts, self.group, out=self._buf("emb%d" % k, U, (d,), torch.float32))
            self._record(slot.fetched, "xchg")

    def _fetch_through(self, t):
        while self.next_fetched <= t:
            self._prepare_through(self.next_fetched)
            self._fetch(self.next_fetched)
            self.next_fetched += 1

    # stage 4 (exchange stream): gradients -> owners, owners dedupe across senders + Adagrad + scatter
    def _update(self, t):
        slot = self._slot(t)
        with self._on("xchg"):
            self._wait("xchg", slot.computed)
            recv_grad = a2a_rows(slot.grad, slot.send_counts, slot.recv_counts, self.group, out=self._buf("recv_grad", slot.nrecv, (self.d,), torch.float32))
            self._apply_local(slot.local_ids, recv_grad, slot.recv_counts)
            self._record(slot.free, "xchg")

    def _dense(self, t, rel_grads):
        if self.sync_interval <= 1:
            for g in rel_grads:
                if g is not None:
                    dist.all_reduce(g, group=self.group)
            self._dense_step(rel_grads)
        else:
            if rel_grads is not None:
                self._dense_step(rel_grads)  # local step; replicas drift for at most sync_interval steps
            if (t + 1) % self.sync_interval == 0:
                for tt in self._dense_state():
                    dist.all_reduce(tt, group=self.group)
                    tt.div_(self.world)

    def step(self):
        t = self.step_index
        self._fetch_through(t)              # no-op except on the first step
        # the scoring of batch t is issued first: fetching the next batch may block the host on split points of a batch whose
        # preparation runs in the gaps the big kernels leave, and the compute stream must not run dry meanwhile
        rel_grads = self._compute(t)
        if self.staleness:
            self._prepare_through(t + 2)
            self._fetch_through(t + 1)      # rows of the next batch move while this one is scored; on the exchange stream this precedes update(t)
        else:
            self._prepare_through(t + 1)    # next batch's preparation overlaps with this batch's compute
        self._update(t)
        self._dense(t, rel_grads)
        self.step_index += 1
        return self._loss()


class _Slot:
    """Per-batch buffers of the pipelined trainer (a ring of four)."""

    def __init__(self, H, B, cols, L, world, dev):
        self.edges = torch.empty((B, cols), dtype=torch.int64, device=dev)
        self.edges_local = torch.empty((B, cols), dtype=torch.int64, device=dev)
        self.all_ids = torch.empty(L, dtype=torch.int64, device=dev)
        self.um = H.UniqueMap(L, dev)
        self.um_rel = H.UniqueMap(B, dev)   # relation ids of the batch, grouped (prepared ahead like the node map)
        self.rel_ids = torch.empty(B, dtype=torch.int64, device=dev)
        self.offs_dev = torch.empty(world + 1, dtype=torch.int64, device=dev)
        self.offs_host = torch.empty(world + 1, dtype=torch.int64).pin_memory()
        self.ready = torch.cuda.Event()     # prep stream: ids sorted, split points on their way to the host
        self.fetched = torch.cuda.Event()   # rows of this batch have arrived
        self.computed = torch.cuda.Event()  # per-row gradients of this batch are complete
        self.free = torch.cuda.Event()      # owners applied this batch's gradients: every buffer of the slot is reusable
        self.used = False
        self.src_neg = self.dst_neg = None
        self.filters = (None, None)
        self.send_counts = self.recv_counts = None
        self.U = self.nrecv = 0
        self.emb = self.grad = self.local_ids = None


class PipelinedShardedTrainer(PipelineSchedule):
    """PipelineSchedule over HIP streams: preparation on a prep stream, the exchange on an exchange stream (staleness 1) or the main
    stream (staleness 0), forward / loss / backward on the main stream.  Local work = the same kernels as the single-GPU step."""

    def __init__(self, stepper, shard_table, shard_state, edges_all, perm, rank, world, num_nodes, sync_interval=1, group=None,
                 side_group=None, staleness=0, trace=None):
        from . import hip as H

        super().__init__(rank, world, num_nodes, stepper.d, shard_table.device, sync_interval, group, side_group, staleness)
        self.H, self.s = H, stepper
        self.backend = HipBackend(stepper, shard_table, shard_state)
        self.edges_all, self.perm = edges_all, perm
        self.trace = trace
        dev = self.dev
        self.main_stream = torch.cuda.current_stream()
        self.prep_stream = torch.cuda.Stream(device=dev)
        self.xchg_stream = torch.cuda.Stream(device=dev) if staleness else self.main_stream
        self.slots = [_Slot(H, stepper.B, stepper.edge_cols, stepper.L, world, dev) for _ in range(self.RING)]
        self.nb = edges_all.size(0) // stepper.B
        self._prime_collectives()

    def _prime_collectives(self):
        """RCCL sets a collective up lazily on its first use per communicator (tens of milliseconds for the first all-reduce); the
        relation-table averaging only happens every sync_interval steps, so without this its one-time cost lands in the middle of a run."""
        s = self.s
        # the averaging step itself, on the real tensors: every replica starts from the same relation tables and zero sums, so
        # sum / world leaves them unchanged (exactly, for a power-of-two world)
        for tt in self.backend.dense_state():
            dist.all_reduce(tt, group=self.group)
            tt.div_(self.world)
        cnt = [1] * self.world
        a2a_rows(torch.zeros((self.world, s.d), dtype=torch.float32, device=self.dev), cnt, cnt, self.group)
        a2a_rows(torch.zeros(self.world, dtype=torch.int64, device=self.dev), cnt, cnt, self.group)
        if self.world > 1:
            dist.all_to_all_single(torch.zeros(self.world, dtype=torch.int64), torch.zeros(self.world, dtype=torch.int64), group=self.side_group)
        torch.cuda.synchronize(self.dev)

    # ---- hooks
    def _stream(self, which):
        return self.xchg_stream if which == "xchg" else self.main_stream

    def _on(self, which):
        return torch.cuda.stream(self._stream(which))

    def _wait(self, which, event):
        self._stream(which).wait_event(event)

    def _record(self, event, which):
        event.record(self._stream(which))

    def _retire_buffer(self, buf):  # the old block may still be read on either stream
        buf.record_stream(self.main_stream)
        buf.record_stream(self.xchg_stream)

    # stage 1 (prep stream): everything that does not read the table
    def _prepare(self, t):
        H, s = self.H, self.s
        slot = self._slot(t)
        B, CN = s.B, s.C * s.N
        with torch.cuda.stream(self.prep_stream):
            if slot.used:
                self.prep_stream.wait_event(slot.free)   # the batch that used this slot RING steps ago is fully retired
            else:
                self.prep_stream.wait_stream(self.main_stream)
            st = H.stream_ptr()
            L_ = H.lib()
            if self.perm is None:
                slot.edges.copy_(self.edges_all[(t % self.nb) * B: (t % self.nb) * B + B])
            else:
                H.check(L_.marius_select_edges(H.ptr(self.edges_all), 1 if self.edges_all.dtype == torch.int64 else 0, s.edge_cols, H.ptr(self.perm),
                                               (t % self.nb) * B, B, H.ptr(slot.edges), st), "select_edges")
            slot.src_neg, slot.dst_neg, sdeg, ddeg = s.sample(slot.edges)
            slot.filters = (H.deg_filter(ddeg, B), H.deg_filter(sdeg, B)) if s.n_deg > 0 else (None, None)
            H.check(L_.marius_assemble_ids(H.ptr(slot.edges), B, s.edge_cols, H.ptr(slot.src_neg), H.ptr(slot.dst_neg), CN, H.ptr(slot.all_ids), st), "assemble")
            slot.um.run(slot.all_ids, s.key_bits)
            H.check(L_.marius_remap_edges(H.ptr(slot.edges), H.ptr(slot.um.inverse), B, s.edge_cols, H.ptr(slot.edges_local), st), "remap")
            H.check(L_.marius_owner_offsets(H.ptr(slot.um.uniq), H.ptr(slot.um.count), self.S, self.world, H.ptr(slot.offs_dev), st), "owner_offsets")
            if s.edge_cols == 3:
                slot.rel_ids.copy_(slot.edges[:, 1])
                slot.um_rel.run(slot.rel_ids, s.rel_bits)
            slot.offs_host.copy_(slot.offs_dev, non_blocking=True)
            slot.ready.record(self.prep_stream)
            for tns in (slot.src_neg, slot.dst_neg) + tuple(f for f in slot.filters if f is not None):
                tns.record_stream(self.main_stream)
        slot.used = True

    def _split_points(self, slot):
        slot.ready.synchronize()
        return slot.offs_host.tolist()

    def _unique_ids(self, slot, U):
        return slot.um.uniq[:U]

    def _gather_local(self, local_ids, out):
        return self.H.gather_rows(self.backend.table, local_ids, out=out)

    # stage 3 (main stream): the same forward / loss / backward kernels as the single-GPU step
    def _compute(self, t):
        H, s = self.H, self.s
        slot = self._slot(t)
        B, CN, d, U = s.B, s.C * s.N, s.d, slot.U
        self.main_stream.wait_event(slot.fetched)
        W = s.W
        src_map = slot.um.inverse[2 * B: 2 * B + CN]
        dst_map = slot.um.inverse[2 * B + CN: 2 * B + 2 * CN]
        W.bind(slot.emb, slot.edges_local, dst_map, src_map, s.rel, s.inv_rel, slot.filters[0], slot.filters[1])
        W.forward()
        W.loss()
        W.backward()
        slot.grad = self._buf("grad%d" % (t % self.RING), U, (d,), torch.float32)
        H.segment_sum_rows(W.gocc(), slot.um, s.L, d, slot.grad, carry=s.carry)
        slot.computed.record(self.main_stream)
        if self.trace is not None:
            self.trace.append({"uniq": slot.um.uniq[:U].clone(), "emb": slot.emb.clone(), "grad": slot.grad.clone()})
        if self.sync_interval > 1:
            # replicas step on their own gradients between averaging points: update only the relation rows this batch touched
            # (a zero-gradient row is a fixed point of the dense Adagrad rule, so this equals the dense step bit for bit)
            H.segment_adagrad_scatter(W.grel(0), slot.um_rel, B, d, s.rel, s.rel_sum, s.dense_lr, carry=s.carry_rel)
            if s.inverse:
                H.segment_adagrad_scatter(W.grel(1), slot.um_rel, B, d, s.inv_rel, s.inv_rel_sum, s.dense_lr, carry=s.carry_rel)
            return None
        s.rel_grad.zero_()
        H.segment_sum_rows(W.grel(0), slot.um_rel, B, d, s.rel_grad, out_rows=slot.um_rel.uniq, carry=s.carry_rel)
        inv = None
        if s.inverse:
            s.inv_rel_grad.zero_()
            H.segment_sum_rows(W.grel(1), slot.um_rel, B, d, s.inv_rel_grad, out_rows=slot.um_rel.uniq, carry=s.carry_rel)
            inv = s.inv_rel_grad
        return [s.rel_grad, inv]

    def _apply_local(self, local_ids, grads, recv_counts=None):
        self.backend.apply_local(local_ids, grads, recv_counts)

    def _dense_step(self, rel_grads):
        self.backend.dense_step(rel_grads)

    def _dense_state(self):
        return self.backend.dense_state()

    def _loss(self):
        return self.s.W.loss_values()[0]

    def finish(self):
        """Retire everything in flight (prefetched batches are simply dropped: preparation and fetch have no side effects)."""
        torch.cuda.synchronize(self.dev)


def run_sharded_bench(a, cfg, rank, world, dev):
    """bench.py body for N > 1: weak scaling (every rank trains its own B-edge batches against the sharded table)."""
    import json
    import time

    from . import hip as H
    from .lp_step import DeviceLinkPredictionStep

    sys_path_fix = None  # noqa: F841
    num_nodes, R, d, B, C, N = cfg["num_nodes"], cfg["num_relations"], cfg["d"], cfg["B"], cfg["C"], cfg["N"]
    strong = bool(getattr(a, "strong", False))
    if strong:  # fixed global batch: every rank trains B / world positives per step (chunks stay whole: C must divide)
        if B % world or C % world:
            raise SystemExit("--strong needs the batch size and the chunk count to be multiples of the number of GPUs")
        B, C = B // world, C // world
    lo, hi = shard_range(num_nodes, rank, world)
    limit = math.sqrt(6.0 / (num_nodes + d))
    table = torch.empty((hi - lo, d), dtype=torch.float32, device=dev).uniform_(-limit, limit, generator=torch.Generator(device=dev).manual_seed(rank))
    state = torch.zeros((hi - lo, d), dtype=torch.float32, device=dev)
    import bench as bench_mod

    edges_all = bench_mod.synth_edges(num_nodes, R, cfg["num_edges"], a.edge_dist, dev, seed=1 + rank)
    sync_interval = int(os.environ.get("MARIUS_GPU_SYNC_INTERVAL", "16"))  # pipeline.gpu_sync_interval default (marius_config.py:672-685)
    pipelined = os.environ.get("MARIUS_SHARDED_PIPELINE", "1") != "0"
    staleness = 0
    driver = os.environ.get("MARIUS_SHARDED_DRIVER", "cpp")  # cpp: ShardedTrainer (csrc/host/sharded_trainer.cpp); py: the Python schedule below
    host_s = None
    cpp_trainer = None
    if pipelined:
        # per-step count exchange (world integers) stays on the CPU.  One node only: pin gloo to the loopback interface — its default
        # picks the interface by resolving the hostname, which containers do not always allow
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        side_group = dist.new_group(backend="gloo")
        # rows of batch t + s are fetched while batch t is scored.  s = 2 keeps the fetch off the score -> update -> fetch -> score cycle
        # (sharded_trainer.cpp: step()): no difference at world 1, one a2a round trip of slack per step once the exchange has wire time
        staleness = int(os.environ.get("MARIUS_SHARDED_STALENESS", "2" if world > 1 else "1"))
    if pipelined and driver == "cpp":
        import marius_amd

        M = marius_amd.host()
        gen = M.MariusGenerator(42 + rank)
        sampler = M.CorruptNodeNegativeSampler(C, N, 0.0, False, M.LocalFilterMode.DEG, gen)
        nodes = M.InMemory("", num_nodes, d, torch.float32, dev)  # never loaded: tells the sampler how many nodes exist
        loader = M.DataLoader(M.InMemory(edges_all), nodes, None, sampler, gen, B, True)
        dec = {"DISTMULT": M.DistMult, "COMPLEX": M.ComplEx, "TRANSE": M.TransE}[cfg["decoder"]](R, d, dev, True, M.EdgeDecoderMethod.CORRUPT_NODE)
        model = M.Model(dec, M.SoftmaxCrossEntropy("sum"), M.LinkPredictionReporter(), dev)
        model.setup_optimizers(0.1)
        model.sparse_lr = 0.1
        try:
            # the constructor already runs every collective the steps use (all-reduce, both all-to-all(v) forms, the gloo count exchange)
            cpp_trainer = M.ShardedTrainer(loader, model, table, state, rank, world, num_nodes, dist.group.WORLD.group_name, side_group.group_name,
                                           staleness, sync_interval)
            # per-stage device spans (which stage's stream is starved): fourteen event records + seven queries per step on the host, so only on request
            cpp_trainer.enable_spans(os.environ.get("MARIUS_SHARDED_SPANS", "0") == "1")
        except Exception as e:  # noqa: BLE001 — e.g. a c10d build without the C++ group registry: same schedule from Python
            import sys
            print("[rank %d] C++ ShardedTrainer unavailable (%s): falling back to the Python schedule" % (rank, e), file=sys.stderr)
            cpp_trainer = None
            driver = "py"
            staleness = min(staleness, 1)  # the Python schedule knows 0 and 1
    if cpp_trainer is not None:

        def run(k0, k):
            cpp_trainer.train_steps(k)
    elif pipelined:
        stepper = DeviceLinkPredictionStep(cfg["decoder"], num_nodes, R, d, B, C, N, seed=42 + rank, device=dev, node_table=None, node_state=None)
        perm = stepper.gen.randperm_host(edges_all.size(0)).to(dev)
        trainer = PipelinedShardedTrainer(stepper, table, state, edges_all, perm, rank, world, num_nodes, sync_interval=sync_interval, side_group=side_group,
                                          staleness=staleness)
        host_s = [0.0]

        def run(k0, k):
            for _ in range(k):
                t_ = time.perf_counter()
                trainer.step()
                host_s[0] += time.perf_counter() - t_   # time the host spends issuing one step (it runs ahead of the GPU unless it is the bottleneck)
    else:
        stepper = DeviceLinkPredictionStep(cfg["decoder"], num_nodes, R, d, B, C, N, seed=42 + rank, device=dev, node_table=None, node_state=None)
        perm = stepper.gen.randperm_host(edges_all.size(0)).to(dev)
        nb = edges_all.size(0) // B
        backend = HipBackend(stepper, table, state)

        def run(k0, k):
            for s in range(k0, k0 + k):
                edges = H.select_edges(edges_all, perm, (s % nb) * B, B)
                sharded_step(backend, edges, rank, world, num_nodes, sync_interval=sync_interval, step_index=s)

    run(0, a.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    if host_s is not None:
        host_s[0] = 0.0
    if cpp_trainer is not None:
        cpp_trainer.reset_counters()
    H.profile_reset()
    H.profile_enable(True, only="lp_grad_adj")  # HIP events around the dominant kernel only (one pair per step)
    t0 = time.perf_counter()
    run(a.warmup, a.steps)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    if cpp_trainer is not None:
        cpp_trainer.finish()  # (after the timed region is closed: raises if a batch exceeded the fixed exchange capacity)
    H.profile_enable(False)
    prof = H.profile_read()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    # wire bytes of the timed region, summed over ranks (ids to owners + rows served + gradients returned)
    xb = torch.tensor(list(cpp_trainer.exchange_bytes) if cpp_trainer is not None else [0, 0, 0], dtype=torch.float64, device=dev)
    dist.all_reduce(xb)
    if rank == 0:
        pos_eps = B * a.steps * world / dt
        flash = bench_mod.flash_selected(H, cfg, B, C, N)
        backend_name = cpp_trainer.backend() if cpp_trainer is not None else dist.get_backend()
        out = {
            "metric": "edges/sec scored (pos+neg)", "value": round(pos_eps * (2 + 2 * N), 1), "unit": "scored edges/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": ("f32 (contractions: 2-way %s split x 3 products, f32 accumulate)" % (
                "fp16 [22 significand bits per operand]" if (cpp_trainer is not None and model.last_step_records == "fp16") else "bf16 [16 significand bits per operand]")) if flash else "f32",
            "data": "synthetic", "arith_check": getattr(a, "arith_check", None),
            "config": {"workload": "%s %s d=%d, node table sharded by contiguous id range over %d GPUs, B=%d per GPU (%s), C=%d N=%d, %s edges" % (
                a.workload, cfg["decoder"], d, world, B, "global batch fixed" if strong else "fixed per GPU", C, N, a.edge_dist),
                "num_nodes": num_nodes, "num_relations": R,
                "parallelism": "dp%d + sharded node table, RCCL %s row fetch / gradient return, relation tables averaged every %d steps, %s" % (
                    world, ("fixed-capacity equal-split all-to-all (%d slots per rank pair, -1 padded: no split size on the host)" % cpp_trainer.pair_capacity)
                    if (cpp_trainer is not None and cpp_trainer.fixed_capacity) else "all-to-all(v)", sync_interval, ("row exchange overlapped with scoring on a second stream (staleness %d step%s; reference pipeline bound: 16)" % (staleness, "s" if staleness > 1 else "")
                                           if pipelined and staleness else "synchronous exchange"))},
            "positive_edges_per_s": round(pos_eps, 1), "roofline": None, "cpu_baseline": None,
            # the communicator the exchange ran on: `ranks` of `collective_backend` ("nccl" is RCCL on ROCm; "gloo" only in the single-GPU
            # emulation of tests/test_gpu_sharded2.py).  `rccl_ranks` counts ranks of an RCCL communicator only: 0 when there is none.
            "ranks": cpp_trainer.ranks() if cpp_trainer is not None else world, "collective_backend": backend_name,
            "rccl_ranks": (cpp_trainer.ranks() if cpp_trainer is not None else world) if backend_name == "nccl" else 0,
            "exchange_bytes_per_step": {"ids": round(float(xb[0]) / a.steps), "rows": round(float(xb[1]) / a.steps), "gradients": round(float(xb[2]) / a.steps),
                                        "total": round(float(xb.sum()) / a.steps),
                                        "note": ("all ranks, bytes that cross xGMI per step: the PADDED payloads of the fixed-capacity exchange (world - 1 blocks of pair_capacity slots per rank "
                                                 "and payload); the counts ride in the id payload as -1 padding") if (cpp_trainer is not None and cpp_trainer.fixed_capacity)
                                        else "all ranks, bytes that cross xGMI per step; counts ride a world-integer device all-to-all"},
        }
        if cpp_trainer is not None and cpp_trainer.fixed_capacity:
            out["exchange"] = {"form": "fixed capacity", "pair_capacity_rows": cpp_trainer.pair_capacity, "id_capacity_per_batch": 2 * B + 2 * C * N,
                               "slack": float(os.environ.get("MARIUS_EXCHANGE_SLACK", "1.5")) if world > 1 else 1.0}
        host_total = cpp_trainer.host_seconds if cpp_trainer is not None else (host_s[0] if host_s is not None else None)
        if host_total is not None:  # how long the host needs to issue a step: it must stay below ms_per_step or the host is the bottleneck
            out["host_issue_ms_per_step"] = round(host_total / a.steps * 1e3, 4)
            out["config"]["host"] = "C++ ShardedTrainer (libtorch, c10d)" if cpp_trainer is not None else "Python schedule (marius_amd/sharded.py)"
            if cpp_trainer is not None:
                if os.environ.get("MARIUS_SHARDED_SPANS", "0") == "1":  # (the instrumentation costs the host-bound step 2-3 %: off unless asked for)
                    out["device_span_ms"] = dict(zip(["prepare", "fetch", "compute", "update", "fetch_ids_a2a", "fetch_owner_gather", "fetch_rows_a2a"], [round(x, 4) for x in cpp_trainer.span_ms]))
                ph = [x / a.steps * 1e3 for x in cpp_trainer.phase_seconds]
                # `prepare` includes the slot-reuse check of the fixed-capacity form; `wait_for_device` is the time the loop stood still because the
                # DEVICE had not got that far (back-pressure: the host runs ahead of it) — the split points of the exact form, the slot-reuse stamp of
                # the fixed form.  host_busy = what the host needs to issue a step.
                out["host_phase_ms_per_step"] = dict(zip(["prepare", "wait_for_device", "fetch", "compute", "update", "dense"], [round(x, 4) for x in ph]))
                out["host_busy_ms_per_step"] = round(out["host_issue_ms_per_step"] - ph[1], 4)
                out["header_reads_repolled"] = int(cpp_trainer.torn_reads)  # exchange headers whose first host read failed the checksum (expected 0)
        ms, cnt = prof.get("lp_grad_adj", (0.0, 0))
        if cnt:  # rank 0's dominant kernel, same accounting as the N = 1 line
            out["roofline"] = bench_mod.dominant_roofline(ms / cnt, B, C, N, d, 2, flash, a.workload == "freebase86m" and not a.num_nodes and not strong)
            out["roofline"]["kernel"] += " (rank 0)"
        if not a.no_cpu_baseline:  # rank 0 only: the same bounded CPU leg as the N = 1 line (per-rank workload)
            out["cpu_baseline"] = bench_mod.cpu_baseline_leg(cfg, B, C, N, edges_all, a.cpu_seconds)
        bench_mod.emit_json(out)
    dist.destroy_process_group()
