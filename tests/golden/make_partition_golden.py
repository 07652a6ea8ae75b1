"""Generates tests/golden/partition_edges.json by importing the REFERENCE's Python edge partitioner (runs only in the build container:
/root/reference does not travel).  Inputs are seeded random edge lists; outputs are the reference's sorted edges and bucket sizes.

    python tests/golden/make_partition_golden.py
"""
import json
import os
import sys
import types

import torch

REF = "/root/reference/src/python"
# the reference installs src/python as the package `marius` (setup.cfg package_dir); alias it without installing anything
pkg = types.ModuleType("marius")
pkg.__path__ = [REF]
sys.modules["marius"] = pkg
from marius.tools.preprocess.converters.partitioners.torch_partitioner import partition_edges  # noqa: E402

cases = []
for seed, (n_nodes, n_rel, n_edges, parts, cols) in enumerate([(45, 3, 200, 5, 3), (100, 10, 1000, 8, 3), (64, 1, 300, 4, 2), (10, 2, 40, 3, 3)]):
    g = torch.Generator().manual_seed(100 + seed)
    src = torch.randint(n_nodes, (n_edges,), generator=g)
    dst = torch.randint(n_nodes, (n_edges,), generator=g)
    rel = torch.randint(n_rel, (n_edges,), generator=g)
    edges = torch.stack([src, rel, dst], 1) if cols == 3 else torch.stack([src, dst], 1)
    out_edges, offsets, _ = partition_edges(edges.clone(), n_nodes, parts)
    cases.append({"num_nodes": n_nodes, "num_partitions": parts, "edges": edges.tolist(), "sorted_edges": out_edges.tolist(),
                  "bucket_sizes": [int(x) for x in offsets]})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "partition_edges.json")
with open(path, "w") as f:
    json.dump(cases, f)
print("wrote", path, os.path.getsize(path), "bytes")
