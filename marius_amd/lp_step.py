"""One synchronous link-prediction training step on the device, sequenced through the C-ABI only.

Mirrors the body of SynchronousTrainer::train (reference src/cpp/src/pipeline/trainer.cpp:106-138) for a
DEVICE_MEMORY node table:  getBatch (edge slice -> negatives -> map_tensors) -> loadGPUParameters (gather) ->
Model::train_batch (forward_lp, loss, backward, dense optimizer step, accumulateGradients) -> updateEmbeddings.
All launches go to the current HIP stream; nothing here synchronises with the host, so steps pipeline back to back.
"""
import math

import torch

from . import hip as H

DECODERS = {"DISTMULT": (H.OP_HADAMARD, H.CMP_DOT), "COMPLEX": (H.OP_COMPLEX_HADAMARD, H.CMP_DOT), "TRANSE": (H.OP_TRANSLATION, H.CMP_L2)}


def init_relations(decoder, num_relations, d, device):
    """distmult.cpp:21-27 ones | complex.cpp:21-29 first d/2 columns one | transe.cpp:21-28 zeros."""
    if decoder == "DISTMULT":
        return torch.ones(num_relations, d, device=device)
    r = torch.zeros(num_relations, d, device=device)
    if decoder == "COMPLEX":
        r[:, : d // 2] = 1
    return r


class DeviceLinkPredictionStep:
    def __init__(self, decoder, num_nodes, num_relations, d, batch_size, num_chunks, num_negatives, degree_fraction=0.0,
                 inverse_edges=True, reduction="sum", sparse_lr=0.1, dense_lr=0.1, seed=42, device="cuda:0",
                 node_table=None, node_state=None, edge_cols=3):
        self.dev = torch.device(device)
        self.decoder, self.num_nodes, self.R, self.d = decoder, num_nodes, num_relations, d
        self.B, self.C, self.N, self.f = batch_size, num_chunks, num_negatives, degree_fraction
        self.n_deg = int(num_negatives * degree_fraction)
        self.inverse, self.sparse_lr, self.dense_lr = inverse_edges, sparse_lr, dense_lr
        self.edge_cols = edge_cols
        relop, cmp = DECODERS[decoder]
        self.table, self.state = node_table, node_state
        self.rel = init_relations(decoder, num_relations, d, self.dev)
        self.inv_rel = init_relations(decoder, num_relations, d, self.dev) if inverse_edges else None
        self.rel_sum = torch.zeros_like(self.rel)
        self.inv_rel_sum = torch.zeros_like(self.rel) if inverse_edges else None
        self.rel_grad = torch.zeros_like(self.rel)
        self.inv_rel_grad = torch.zeros_like(self.rel) if inverse_edges else None
        self.gen = H.Generator(seed, self.dev)
        CN = self.C * self.N
        self.L = 2 * self.B + 2 * CN
        self.words = H.negatives_raw_words(num_nodes, self.B, self.C, self.N, self.n_deg)
        # raw MT19937 words for one step (src negatives then dst negatives), double buffered: the single-workgroup generator
        # kernel runs on a side stream one step ahead of the consumer (same stream of numbers, just produced early)
        self.raw_bufs = [torch.empty(2 * self.words, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self.side = torch.cuda.Stream(device=self.dev)
        self.ev_ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.ev_free = [torch.cuda.Event(), torch.cuda.Event()]
        self.step_idx = 0
        self.prefetched = -1
        self.all_ids = torch.empty(self.L, dtype=torch.int64, device=self.dev)
        self.um = H.UniqueMap(self.L, self.dev)
        self.um_rel = H.UniqueMap(self.B, self.dev)
        self.edges_local = torch.empty((self.B, edge_cols), dtype=torch.int64, device=self.dev)
        self.emb = torch.empty((self.L, d), dtype=torch.float32, device=self.dev)
        self.W = H.LpWorkspace(relop, cmp, d, self.B, self.C, self.N, inverse_edges, H.REDUCE_SUM if reduction == "sum" else H.REDUCE_MEAN,
                               edge_cols, True, self.dev)
        self.carry = H.segment_carry(self.L, d, self.dev)
        self.carry_rel = H.segment_carry(self.B, d, self.dev)
        self.key_bits = max(1, math.ceil(math.log2(num_nodes + 1)))
        self.rel_bits = max(1, math.ceil(math.log2(num_relations + 1)))
        self.rel_ids = torch.empty(self.B, dtype=torch.int64, device=self.dev)
        self.last = {}

    # ---- getBatch: dataloader.cpp:360-471
    def sample(self, edges):
        """edges [B, cols] int64 global ids on device. Returns (src_neg, dst_neg, src_deg_pos, dst_deg_pos)."""
        slot = self.step_idx & 1
        main = torch.cuda.current_stream()
        if self.prefetched < self.step_idx:
            self._prefetch_raw(self.step_idx, main)
        main.wait_event(self.ev_ready[slot])
        raw = self.raw_bufs[slot]
        src_neg, sdeg = H.sample_negatives(raw[: self.words], edges, self.num_nodes, self.C, self.N, self.f, True)
        dst_neg, ddeg = H.sample_negatives(raw[self.words:], edges, self.num_nodes, self.C, self.N, self.f, False)
        self.ev_free[slot].record(main)
        self.step_idx += 1
        self._prefetch_raw(self.step_idx, main)
        return src_neg, dst_neg, sdeg, ddeg

    def _prefetch_raw(self, idx, main):
        slot = idx & 1
        with torch.cuda.stream(self.side):
            if idx >= 2:
                self.side.wait_event(self.ev_free[slot])
            else:
                self.side.wait_stream(main)
            self.gen.fill_device(2 * self.words, out=self.raw_bufs[slot])
            self.ev_ready[slot].record(self.side)
        self.prefetched = idx

    def step(self, edges, dst_filter=None, src_filter=None):
        B, CN, d = self.B, self.C * self.N, self.d
        src_neg, dst_neg, sdeg, ddeg = self.sample(edges)
        if self.n_deg > 0 and dst_filter is None:  # LocalFilterMode::DEG (training default, negative.cpp:21-39)
            src_filter, dst_filter = H.deg_filter(sdeg, B), H.deg_filter(ddeg, B)
        st = H.stream_ptr()
        L_ = H.lib()
        H.check(L_.marius_assemble_ids(H.ptr(edges), B, self.edge_cols, H.ptr(src_neg), H.ptr(dst_neg), CN, H.ptr(self.all_ids), st), "assemble")
        self.um.run(self.all_ids, self.key_bits)
        H.check(L_.marius_remap_edges(H.ptr(edges), H.ptr(self.um.inverse), B, self.edge_cols, H.ptr(self.edges_local), st), "remap")
        src_map = self.um.inverse[2 * B: 2 * B + CN]
        dst_map = self.um.inverse[2 * B + CN: 2 * B + 2 * CN]
        # ---- loadGPUParameters: capacity-sized gather (uniq tail reads row 0), no host sync on U
        H.gather_rows(self.table, self.um.uniq, out=self.emb)
        # ---- train_batch
        W = self.W
        W.bind(self.emb, self.edges_local, dst_map, src_map, self.rel if self.edge_cols == 3 else None, self.inv_rel, dst_filter, src_filter)
        W.forward()
        W.loss()
        W.backward()
        if self.edge_cols == 3:
            self.rel_ids.copy_(edges[:, 1])
            self.um_rel.run(self.rel_ids, self.rel_bits)
            self.rel_grad.zero_()
            H.segment_sum_rows(W.grel(0), self.um_rel, B, d, self.rel_grad, out_rows=self.um_rel.uniq, carry=self.carry_rel)
            H.dense_adagrad_step(self.rel, self.rel_sum, self.rel_grad, self.dense_lr)
            if self.inverse:
                self.inv_rel_grad.zero_()
                H.segment_sum_rows(W.grel(1), self.um_rel, B, d, self.inv_rel_grad, out_rows=self.um_rel.uniq, carry=self.carry_rel)
                H.dense_adagrad_step(self.inv_rel, self.inv_rel_sum, self.inv_rel_grad, self.dense_lr)
        # ---- accumulateGradients + updateEmbeddings, fused, atomic-free
        H.segment_adagrad_scatter(W.gocc(), self.um, self.L, d, self.table, self.state, self.sparse_lr, carry=self.carry)
        self.last = {"src_neg": src_neg, "dst_neg": dst_neg}
        return W
