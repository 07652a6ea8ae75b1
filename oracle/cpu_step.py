"""ORACLE (test infrastructure): the reference's synchronous CPU training step, restated with the same libtorch CPU ops
in the same order (reference src/cpp/src/pipeline/trainer.cpp:106-138 and callees).  Used by tests as the
end-to-end checker and by bench.py's cpu_baseline leg ("port": the reference sources cannot travel to the GPU box).
"""
import time

import torch

from . import lp_oracle as O


class CpuLinkPredictionStep:
    def __init__(self, decoder, table, state, num_relations, batch_size, num_chunks, num_negatives, degree_fraction=0.0,
                 inverse_edges=True, reduction="sum", sparse_lr=0.1, dense_lr=0.1):
        self.decoder, self.table, self.state = decoder, table, state
        self.num_nodes, self.d = table.shape
        self.B, self.C, self.N, self.f = batch_size, num_chunks, num_negatives, degree_fraction
        self.inverse, self.reduction, self.sparse_lr, self.dense_lr = inverse_edges, reduction, sparse_lr, dense_lr
        self.rel = O.init_relations(decoder, num_relations, self.d)
        self.inv_rel = O.init_relations(decoder, num_relations, self.d) if inverse_edges else None
        self.loss, self.margin = "SOFTMAX_CE", 0.1
        self.filtered_edges = None  # (src-sorted, dst-sorted) known edges: filtered sampler (negative.cpp:321-325, 354-356)
        self.rel_sum = torch.zeros_like(self.rel)
        self.inv_rel_sum = torch.zeros_like(self.rel) if inverse_edges else None
        # the embedding layer's post-hook (layer.cpp:9-16; default none): bias [d] is a parameter of the dense optimizer (model.cpp:175-183)
        self.enc_bias, self.enc_activation, self.enc_bias_sum = None, "NONE", None

    def get_negatives(self, edges, inverse):
        """negative.cpp:328-366 on the global torch CPU generator."""
        if self.filtered_edges is not None:  # every node is a negative, true edges masked; no random draw
            ids = torch.arange(self.num_nodes).unsqueeze(0)
            return ids, O.compute_filter_corruption_global(self.filtered_edges[0], self.filtered_edges[1], edges, inverse)
        n_deg = int(self.N * self.f)
        n_uni = self.N - n_deg
        rows, deg_rows = [], []
        for _ in range(self.C):
            ids = torch.randint(self.num_nodes, (n_uni,), dtype=torch.int64)
            if self.f > 0:
                pos = torch.randint(0, edges.size(0), (n_deg,), dtype=torch.int64)
                ids = torch.cat([edges.index_select(0, pos)[:, 0 if inverse else -1], ids])
                deg_rows.append(pos)
            rows.append(ids)
        deg = torch.stack(deg_rows) if deg_rows else None
        return torch.stack(rows), O.deg_negative_local_filter(deg, edges)

    def step(self, edges):
        """edges [B,3] int64 global ids. Returns dict of intermediates (global ids, scores, loss)."""
        src_neg, src_filter = self.get_negatives(edges, True)   # dataloader.cpp:498-503: inverse first
        dst_neg, dst_filter = self.get_negatives(edges, False)
        uniq, mapped = O.map_tensors([edges[:, 0], edges[:, -1], src_neg.flatten(), dst_neg.flatten()])
        edges_local = torch.stack([mapped[0], edges[:, 1], mapped[1]]).transpose(0, 1)
        src_map, dst_map = mapped[2].reshape(src_neg.shape), mapped[3].reshape(dst_neg.shape)
        emb = O.index_read(self.table, uniq)
        st = O.index_read(self.state, uniq)
        out = O.train_batch(self.decoder, emb, st, edges_local, dst_map, src_map, self.rel, self.inv_rel, dst_filter, src_filter,
                            self.reduction, self.sparse_lr, self.loss, self.margin, encoder_bias=self.enc_bias, encoder_activation=self.enc_activation)
        if self.enc_bias is not None:
            if self.enc_bias_sum is None:
                self.enc_bias_sum = torch.zeros_like(self.enc_bias)
            O.dense_adagrad_step(self.enc_bias, out["bias_grad"], self.enc_bias_sum, self.dense_lr)
        O.dense_adagrad_step(self.rel, out["rel_grad"], self.rel_sum, self.dense_lr)
        if self.inverse:
            O.dense_adagrad_step(self.inv_rel, out["inv_rel_grad"], self.inv_rel_sum, self.dense_lr)
        O.index_add(self.table, uniq, out["dw"])
        O.index_add(self.state, uniq, out["ds"])
        out.update({"uniq": uniq, "src_neg": src_neg, "dst_neg": dst_neg})
        return out


def time_cpu_baseline(decoder, num_nodes_proxy, num_nodes_ids, num_relations, d, B, C, N, edges, max_seconds=20.0, warmup=1, seed=42):
    """Times full steps on the host cores. Returns (positive edges / s, steps timed, threads)."""
    g = torch.Generator().manual_seed(0)
    table = (torch.rand(num_nodes_proxy, d, generator=g) * 2 - 1) * (6.0 / (num_nodes_ids + d)) ** 0.5
    state = torch.zeros(num_nodes_proxy, d)
    stepper = CpuLinkPredictionStep(decoder, table, state, num_relations, B, C, N)
    stepper.num_nodes = num_nodes_proxy
    torch.manual_seed(seed)
    n = 0
    t0 = None
    steps = 0
    while True:
        batch = edges[(n * B) % max(1, edges.size(0) - B):][:B].clone()
        batch[:, 0] %= num_nodes_proxy
        batch[:, 2] %= num_nodes_proxy
        if n == warmup:
            t0 = time.perf_counter()
        stepper.step(batch)
        n += 1
        if t0 is not None:
            steps = n - warmup
            if time.perf_counter() - t0 > max_seconds or steps >= 8:
                break
    dt = time.perf_counter() - t0
    return B * steps / dt, steps, torch.get_num_threads()
