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
        "num_epochs": 10, "epochs_per_shuffle": 1, "logs_per_epoch": 10,
        # marius_config.py:664-682 (PipelineConfig): only sync / staleness_bound / gpu_sync_interval / gpu_model_average have a meaning on this path
        "pipeline": {"sync": True, "staleness_bound": 16, "gpu_sync_interval": 16, "gpu_model_average": True},
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


def get_model_dir_path(dataset_dir):
    """marius_config.py:47-56: the first free <dataset_dir>/model_<i>, i in 0..10; beyond that model_10 is overwritten."""
    p = None
    for i in range(11):
        p = os.path.join(dataset_dir, "model_%d" % i)
        if not os.path.exists(p):
            return p
    return p


def infer_model_dir(cfg, auto):
    """marius_config.py:875-896: a run that does not create a model directory (marius_eval, marius_predict, resume_training without
    resume_from_checkpoint) uses model_dir as given when it already holds model.pt, else — for the automatic model_<i> name — the
    latest existing model_<i-1>."""
    mdir = cfg["storage"]["model_dir"]
    if os.path.isdir(mdir) and os.path.exists(os.path.join(mdir, "model.pt")):
        return
    base = os.path.basename(os.path.normpath(mdir))
    if auto and base.startswith("model_") and base[6:].isdigit() and int(base[6:]) >= 1:
        cfg["storage"]["model_dir"] = os.path.join(os.path.dirname(os.path.normpath(mdir)), "model_%d" % (int(base[6:]) - 1))


def load_config(path, train=True):
    """Returns the full configuration dict (defaults merged, dataset stats filled in).  train=False: the caller evaluates (marius_eval):
    the model directory is looked up, never created (marius_config.py:920-940)."""
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
    if len(layers) != 1 or len(layers[0]) != 1 or str(layers[0][0].get("type", "")).upper() != "EMBEDDING":
        raise NotImplementedError("only the embedding-only encoder is on the link-prediction hot path")
    layers[0][0] = check_embedding_layer(layers[0][0])
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
    if int(cfg["training"]["pipeline"].get("staleness_bound", 16)) < 1:
        raise ValueError("training.pipeline.staleness_bound must be at least 1")
    auto = cfg["storage"]["model_dir"] is None
    if auto:
        cfg["storage"]["model_dir"] = get_model_dir_path(ddir)
    tr = cfg["training"]
    creates = train and (bool(tr.get("resume_from_checkpoint")) or not tr.get("resume_training", False))  # marius_config.py:923-929
    cfg["_creates_model_dir"] = creates
    if not creates:
        infer_model_dir(cfg, auto)
    return cfg


# LayerConfig (marius_config.py:190-199) of the one embedding layer.  Its post-hook (Layer::post_hook, layer.cpp:9-16: `+ bias`, then the
# activation) is marius_layer_post_hook / _backward behind GeneralEncoder (round 6); a model that uses it trains through the API-granular step.
INIT_TYPES = {"GLOROT_UNIFORM": {}, "GLOROT_NORMAL": {}, "UNIFORM": {"scale_factor": 1.0}, "NORMAL": {"mean": 0.0, "std": 1.0}, "ZEROS": {}, "ONES": {},
              "CONSTANT": {"constant": 0.0}}  # InitConfig / *InitOptions (marius_config.py:96-160), initialization.cpp:67-95


def check_init(init, where):
    init = dict(init or {"type": "GLOROT_UNIFORM"})
    kind = str(init.get("type", "GLOROT_UNIFORM")).upper()
    if kind not in INIT_TYPES:
        raise ValueError("%s: unknown init type %r (one of %s)" % (where, init.get("type"), ", ".join(sorted(INIT_TYPES))))
    opts = dict(init.get("options") or {})
    unknown = set(opts) - set(INIT_TYPES[kind])
    if unknown:
        raise ValueError("%s: init type %s takes no option(s) %s" % (where, kind, ", ".join(sorted(unknown))))
    return {"type": kind, "options": {**INIT_TYPES[kind], **{k: float(v) for k, v in opts.items()}}}


def check_embedding_layer(layer):
    known = {"type", "options", "input_dim", "output_dim", "init", "optimizer", "bias", "bias_init", "activation"}
    unknown = set(layer) - known
    if unknown:
        raise ValueError("model.encoder.layers[0][0]: unknown key(s) %s" % ", ".join(sorted(unknown)))
    out = dict(layer)
    out["type"] = "EMBEDDING"
    if int(out.get("output_dim", 50)) < 1:
        raise ValueError("model.encoder.layers[0][0].output_dim must be positive")
    out["output_dim"] = int(out.get("output_dim", 50))
    if int(out.get("input_dim", -1)) not in (-1, out["output_dim"]):
        raise ValueError("model.encoder.layers[0][0]: an embedding layer has no input (input_dim must be -1 or equal to output_dim)")
    if out.get("options"):
        raise NotImplementedError("model.encoder.layers[0][0].options: an EMBEDDING layer takes none (LayerOptions, marius_config.py:163-187)")
    out["bias"] = bool(out.get("bias", False))
    out["activation"] = str(out.get("activation", "NONE")).upper()
    if out["activation"] not in ("NONE", "RELU", "SIGMOID"):  # ActivationFunction (options.h) / apply_activation (activation.cpp:7-21)
        raise ValueError("model.encoder.layers[0][0].activation: %s (NONE, RELU or SIGMOID)" % out["activation"])
    out["bias_init"] = check_init(out.get("bias_init") or {"type": "ZEROS"}, "model.encoder.layers[0][0].bias_init")  # LayerConfig default: ZEROS
    opt = out.get("optimizer")
    if opt and str(opt.get("type", "DEFAULT")).upper() != "DEFAULT":
        raise NotImplementedError("model.encoder.layers[0][0].optimizer: node embeddings are trained by model.sparse_optimizer (ADAGRAD)")
    out["init"] = check_init(out.get("init"), "model.encoder.layers[0][0].init")
    return out


def initialize_rows(init, rows, d, fans, device, generator=None):
    """initialize_subtensor (initialization.cpp:98-119) for `rows` rows of the [fans[0], d] node table: the scale of the GLOROT forms comes from
    the FULL table shape (fan_in = num_nodes, fan_out = d)."""
    import math

    import torch

    kind, o = init["type"], init["options"]
    shape = (rows, d)
    kw = dict(dtype=torch.float32, device=device)
    if kind == "GLOROT_UNIFORM":
        limit = math.sqrt(6.0 / (fans[0] + fans[1]))
        return torch.empty(shape, **kw).uniform_(-limit, limit, generator=generator)
    if kind == "GLOROT_NORMAL":
        return torch.empty(shape, **kw).normal_(0.0, math.sqrt(2.0 / (fans[0] + fans[1])), generator=generator)
    if kind == "UNIFORM":
        return torch.empty(shape, **kw).uniform_(-o["scale_factor"], o["scale_factor"], generator=generator)
    if kind == "NORMAL":
        return torch.empty(shape, **kw).normal_(o["mean"], o["std"], generator=generator)
    if kind == "ZEROS":
        return torch.zeros(shape, **kw)
    if kind == "ONES":
        return torch.ones(shape, **kw)
    if kind == "CONSTANT":
        return torch.full(shape, o["constant"], **kw)
    raise ValueError(kind)


def embedding_init(cfg):
    return cfg["model"]["encoder"]["layers"][0][0]["init"]


def embedding_dim(cfg):
    return int(cfg["model"]["encoder"]["layers"][0][0]["output_dim"])
