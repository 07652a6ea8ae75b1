#!/usr/bin/env python3
"""Generates the committed golden fixtures (data only).  Run in the build container (needs /root/reference for oracle/_ref):

    make -C oracle && python tests/golden/make_golden.py

  ref_known_answers.json   literal inputs / expected outputs held by the reference's own tests
                           (test/python/bindings/integration/test_nn.py:15-25,148-160,197-207; test_data.py:34-47)
  rng_golden.json          torch.randint / torch.randperm streams of this container's torch (the generator the reference
                           draws from: negative.cpp:340-357, dataloader.cpp:176-182) for fixed seeds
  ref_scores_golden.npz    outputs of the REFERENCE's comparators.cpp / relation_operators.cpp (compiled where they lie into
                           oracle/_ref/libmarius_ref.so) on seeded inputs: operators, comparators, and one-direction
                           score + SoftmaxCE(SUM) forward/backward through the reference operators (libtorch autograd)
"""
import ctypes as C
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def known_answers():
    out = {
        "distmult_forward": {  # test_nn.py:15-25,148-160
            "node_embeddings": [[1.5, 2.5], [2.5, 3.5], [4.25, 1.0], [-1.0, 0.5]],
            "batch_edges": [[0, 0, 1], [2, 0, 3], [3, 1, 0]],
            "num_relations": 2, "embedding_dim": 2, "use_inverse_relations": False,
            "expected_scores": [12.5, -3.75, -0.25],
        },
        "accumulate_gradients": {  # test_data.py:34-47: state update == grad^2, gradients == -lr * grad / (sqrt(state) + 1e-10)
            "node_embeddings": [2.0, 4.0], "grad": [0.5, -1.0], "state": [0.0, 0.0], "learning_rate": 1.0,
            "expected_state_update": [0.25, 1.0],
        },
        "sgd_step": {  # test_nn.py:197-207: param (zeros) after step == -grad * lr
            "grad": [-1.0, -2.0], "learning_rate": 0.1, "expected_param": [0.1, 0.2],
        },
        "train_batch_shapes": {  # test_nn.py:162-176 (runs without throwing; shapes only)
            "dst_neg_indices_mapping": [[2, 0], [0, 1], [1, 0]],
        },
    }
    with open(os.path.join(HERE, "ref_known_answers.json"), "w") as f:
        json.dump(out, f, indent=1)


def rng():
    cases = []
    for seed, high, n in [(42, 86054151, 64), (42, 14541, 64), (7, 6, 40), (123456789012, 1000, 32), (42, 2 ** 28, 16),
                          (42, 2 ** 28 - 1, 16), (42, 2 ** 40, 16)]:
        torch.manual_seed(seed)
        cases.append({"kind": "randint", "seed": seed, "high": high, "n": n, "values": torch.randint(high, (n,)).tolist()})
    for seed, n in [(7, 10), (3, 1000), (99, 1), (42, 2)]:
        torch.manual_seed(seed)
        perm = torch.randperm(n).tolist()
        nxt = torch.randint(1000, (5,)).tolist()
        cases.append({"kind": "randperm", "seed": seed, "n": n, "values": perm, "next_randint_1000": nxt})
    # getNegatives call sequence (negative.cpp:328-366), inverse first then forward (dataloader.cpp:498-503)
    for seed, B, Cc, N, f, num_nodes in [(11, 6, 1, 5, 0.0, 6), (12, 6, 3, 5, 0.5, 6), (13, 40, 4, 10, 0.3, 1000)]:
        g = torch.Generator().manual_seed(seed)
        edges = torch.stack([torch.randint(num_nodes, (B,), generator=g), torch.randint(3, (B,), generator=g),
                             torch.randint(num_nodes, (B,), generator=g)], 1)
        torch.manual_seed(seed)
        n_deg = int(N * f)
        outs = []
        for inverse in (True, False):
            rows, degs = [], []
            for _ in range(Cc):
                ids = torch.randint(num_nodes, (N - n_deg,), dtype=torch.int64)
                if f > 0:
                    pos = torch.randint(0, B, (n_deg,), dtype=torch.int64)
                    ids = torch.cat([edges.index_select(0, pos)[:, 0 if inverse else -1], ids])
                    degs.append(pos.tolist())
                rows.append(ids.tolist())
            outs.append({"inverse": inverse, "ids": rows, "deg_pos": degs})
        cases.append({"kind": "get_negatives", "seed": seed, "B": B, "C": Cc, "N": N, "degree_fraction": f, "num_nodes": num_nodes,
                      "edges": edges.tolist(), "calls": outs})
    with open(os.path.join(HERE, "rng_golden.json"), "w") as f:
        json.dump({"torch_version": torch.__version__, "cases": cases}, f)


def ref_scores():
    so = os.path.join(ROOT, "oracle", "_ref", "libmarius_ref.so")
    L = C.CDLL(so)
    fp = lambda a: a.ctypes.data_as(C.c_void_p)
    out = {}
    rs = np.random.RandomState(0)
    B, d = 12, 10
    e, r, o = [rs.randn(B, d).astype(np.float32) for _ in range(3)]
    out["op_in_e"], out["op_in_r"], out["cmp_in_o"] = e, r, o
    for op, name in [(0, "hadamard"), (1, "complex_hadamard"), (2, "translation")]:
        res = np.zeros((B, d), np.float32)
        assert L.ref_relation_op(op, fp(e), fp(r), C.c_int64(B), C.c_int64(d), fp(res)) == 0
        out["op_" + name] = res
    for cmp, name in [(0, "dot"), (1, "l2"), (2, "cosine")]:
        res = np.zeros(B, np.float32)
        assert L.ref_compare_same(cmp, fp(e), fp(o), C.c_int64(B), C.c_int64(d), fp(res)) == 0
        out["cmp_same_" + name] = res
    for Bn, Cn, Nn in [(12, 3, 7), (10, 4, 5)]:  # second case: B % C != 0 -> zero padded rows (comparators.cpp:7-20)
        src = rs.randn(Bn, d).astype(np.float32)
        negs = rs.randn(Cn, Nn, d).astype(np.float32)
        Bp = Cn * int(np.ceil(np.float32(Bn) / Cn))
        out["neg_in_src_%d" % Bn], out["neg_in_negs_%d" % Bn] = src, negs
        for cmp, name in [(0, "dot"), (1, "l2"), (2, "cosine")]:
            res = np.zeros((Bp, Nn), np.float32)
            assert L.ref_compare_neg(cmp, fp(src), fp(negs), C.c_int64(Bn), C.c_int64(Cn), C.c_int64(Nn), C.c_int64(d), fp(res)) == 0
            out["cmp_neg_%s_%d" % (name, Bn)] = res
    for dec, (op, cmp) in {"DISTMULT": (0, 0), "COMPLEX": (1, 0), "TRANSE": (2, 1)}.items():
        Bn, Cn, Nn, dd = 12, 3, 6, 8
        src, rel, dst = [(rs.randn(Bn, dd) * 0.7).astype(np.float32) for _ in range(3)]
        negs = (rs.randn(Cn, Nn, dd) * 0.7).astype(np.float32)
        pos, neg, loss = np.zeros(Bn, np.float32), np.zeros((Bn, Nn), np.float32), np.zeros(1, np.float32)
        gs, gr, gd, gn = np.zeros_like(src), np.zeros_like(rel), np.zeros_like(dst), np.zeros_like(negs)
        rc = L.ref_score_fwd_bwd(op, cmp, fp(src), fp(rel), fp(dst), fp(negs), C.c_int64(Bn), C.c_int64(Cn), C.c_int64(Nn), C.c_int64(dd),
                                 fp(pos), fp(neg), fp(loss), fp(gs), fp(gr), fp(gd), fp(gn))
        assert rc == 0
        for k, v in dict(src=src, rel=rel, dst=dst, negs=negs, pos=pos, neg=neg, loss=loss, g_src=gs, g_rel=gr, g_dst=gd, g_negs=gn).items():
            out["fb_%s_%s" % (dec, k)] = v
    np.savez_compressed(os.path.join(HERE, "ref_scores_golden.npz"), **out)


if __name__ == "__main__":
    known_answers()
    rng()
    ref_scores()
    print("golden fixtures written to", HERE)
