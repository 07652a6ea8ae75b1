"""YAML configuration surface of `marius_train` for the link-prediction path.

Same keys and defaults as the reference's OmegaConf schema (src/python/tools/configuration/marius_config.py; defaults cited in
SURVEY.md §5.6) for everything the hot path reads, including storage.embeddings.type PARTITION_BUFFER (out-of-core node table,
options as datatypes.py:161-185); keys that configure subsystems outside the path (GNN layers, async pipeline) are accepted and
ignored with a warning.  Dataset statistics come from <dataset_dir>/dataset.yaml
(marius_config.py:470-493), model_dir defaults to <dataset_dir>/model_<i> (:47-56,562-565).
"""
import copy
import os
import random
import warnings

import yaml

DEFAULTS = {
    "model": {
        "random_seed": None,  # random if absent (marius_config.py:354-356)
        "learning_task": "LINK_PREDICTION",
        "encoder": {"layers": [[{"type": "EMBEDDING", "output_dim": 50}]]},
        "decoder": {"type": "DISTMULT", "options": {"inverse_edges": True, "edge_decoder_method": "CORRUPT_NODE"}},
        "loss": {"type": "SOFTMAX_CE", "options": {"reduction": "SUM"}},
        "dense_optimizer": {"type": "ADAGRAD", "options": {"learning_rate": 0.1}},
        "sparse_optimizer": {"type": "ADAGRAD", "options": {"learning_rate": 0.1}},
    },
    "storage": {
        "device_type": "cpu",
        "dataset": {"dataset_dir": None, "num_relations": 1},
        "edges": {"type": "DEVICE_MEMORY", "options": {"dtype": "int"}},
        "embeddings": {"type": "DEVICE_MEMORY", "options": {"dtype": "float"}},
        "save_model": True, "shuffle_input": True, "full_graph_evaluation": True, "model_dir": None,
    },
    "training": {
        "batch_size": 1000,
        "negative_sampling": {"num_chunks": 1, "negatives_per_positive": 1000, "degree_fraction": 0.0, "filtered": False, "local_filter_mode": "DEG"},
        "num_epochs": 10, "pipeline": {"sync": True}, "epochs_per_shuffle": 1, "logs_per_epoch": 10,
        # marius_config.py:649-652, 734-735
        "save_model": True, "checkpoint": {"save_best": False, "interval": -1, "save_state": False}, "resume_training": False, "resume_from_checkpoint": "",
    },
    "evaluation": {
        "batch_size": 1000,
        "negative_sampling": {"num_chunks": 1, "negatives_per_positive": 1000, "degree_fraction": 0.0, "filtered": False, "local_filter_mode": "DEG"},
        "pipeline": {"sync": True}, "epochs_per_eval": 1, "checkpoint_dir": "",  # marius_config.py:782
    },
}


# PartitionBufferOptions (src/python/tools/configuration/datatypes.py:161-169)
PARTITION_BUFFER_DEFAULTS = {"dtype": "float", "num_partitions": 16, "buffer_capacity": 8, "prefetching": True, "fine_to_coarse_ratio": 1,
                             "num_cache_partitions": 0, "edge_bucket_ordering": "COMET", "randomly_assign_edge_buckets": True}


def _merge(base, over, path=""):
    out = copy.deepcopy(base)
    for k, v in (over or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v, path + k + ".")
        else:
            out[k] = v
    return out


def load_config(path):
    """Returns the full configuration dict (defaults merged, dataset stats filled in)."""
    with open(path) as f:
        user = yaml.safe_load(f)
    cfg = _merge(DEFAULTS, user)
    base = os.path.dirname(os.path.abspath(path))
    ds = cfg["storage"]["dataset"]
    if not ds.get("dataset_dir"):
        raise ValueError("storage.dataset.dataset_dir is required")
    ddir = ds["dataset_dir"]
    if not os.path.isabs(ddir):
        ddir = os.path.normpath(os.path.join(base, ddir)) if not os.path.isdir(ddir) else os.path.abspath(ddir)
    ds["dataset_dir"] = ddir
    stats = os.path.join(ddir, "dataset.yaml")
    if os.path.exists(stats):
        with open(stats) as f:
            for k, v in (yaml.safe_load(f) or {}).items():
                if k != "dataset_dir":
                    ds[k] = v
    for k in ("num_nodes", "num_train"):
        if k not in ds:
            raise ValueError("dataset statistic %s missing (expected in %s)" % (k, stats))
    if cfg["model"]["random_seed"] is None:
        cfg["model"]["random_seed"] = random.randint(0, 2 ** 31 - 1)
    if cfg["model"]["learning_task"] != "LINK_PREDICTION":
        raise NotImplementedError("only LINK_PREDICTION is in scope (SURVEY.md §8)")
    layers = cfg["model"]["encoder"]["layers"]
    if len(layers) != 1 or len(layers[0]) != 1 or layers[0][0]["type"] != "EMBEDDING":
        raise NotImplementedError("only the embedding-only encoder is on the link-prediction hot path")
    # filtered (config.cpp:365-376): num_chunks = 1, negatives = -1 (every node), scores of true edges masked (negative.cpp:212-311);
    # marius_train sorts train + validation + test edges (GraphModelStorage::sortAllEdges) for every filtered sampler, training included
    for section in ("training", "evaluation"):
        ns = cfg[section]["negative_sampling"]
        if ns["filtered"]:
            ns.update({"num_chunks": 1, "degree_fraction": 0.0, "negatives_per_positive": -1, "local_filter_mode": "DEG"})
    if cfg["training"]["negative_sampling"]["filtered"] and cfg["storage"]["embeddings"]["type"] == "PARTITION_BUFFER":
        raise NotImplementedError("filtered training needs every node in memory: use DEVICE_MEMORY embeddings")
    emb = cfg["storage"]["embeddings"]
    if emb["type"] == "PARTITION_BUFFER":
        o = _merge(PARTITION_BUFFER_DEFAULTS, emb.get("options"))
        if o["num_partitions"] < 2:   # datatypes.py:171-179
            raise ValueError("There must be at least two partitions to use the partition buffer, got: %d" % o["num_partitions"])
        if o["buffer_capacity"] < 2:
            raise ValueError("The partition buffer must have capacity of at least 2, got: %d" % o["buffer_capacity"])
        o["buffer_capacity"] = min(o["buffer_capacity"], o["num_partitions"])  # :181-183
        emb["options"] = o
    elif emb["type"] not in ("DEVICE_MEMORY", "HOST_MEMORY"):
        raise NotImplementedError("storage.embeddings.type %s" % emb["type"])
    if not cfg["training"]["pipeline"].get("sync", True):
        warnings.warn("async pipeline is out of scope; running the synchronous trainer")
    if cfg["storage"]["model_dir"] is None:
        i = 0
        while os.path.exists(os.path.join(ddir, "model_%d" % i)):
            i += 1
        cfg["storage"]["model_dir"] = os.path.join(ddir, "model_%d" % i)
    return cfg


def embedding_dim(cfg):
    return int(cfg["model"]["encoder"]["layers"][0][0]["output_dim"])
