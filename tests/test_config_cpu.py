"""YAML configuration surface (marius_amd/config.py) against the reference's schema defaults and validation rules
(src/python/tools/configuration/marius_config.py, datatypes.py; src/cpp/src/configuration/config.cpp:358-380).  No GPU needed."""
import os

import pytest
import yaml

from marius_amd import config as C


def write(tmp_path, user, stats=None):
    ddir = tmp_path / "ds"
    ddir.mkdir(exist_ok=True)
    yaml.safe_dump(stats or {"num_nodes": 100, "num_relations": 4, "num_train": 900, "num_valid": 50, "num_test": 50, "num_edges": 1000},
                   open(ddir / "dataset.yaml", "w"))
    user.setdefault("storage", {}).setdefault("dataset", {})["dataset_dir"] = str(ddir)
    user["storage"].setdefault("device_type", "cuda")
    path = tmp_path / "cfg.yaml"
    yaml.safe_dump(user, open(path, "w"))
    return str(path), str(ddir)


def test_defaults_match_reference_schema(tmp_path):
    path, ddir = write(tmp_path, {})
    cfg = C.load_config(path)
    assert cfg["model"]["decoder"] == {"type": "DISTMULT", "options": {"inverse_edges": True, "edge_decoder_method": "CORRUPT_NODE"}}
    assert cfg["model"]["loss"] == {"type": "SOFTMAX_CE", "options": {"reduction": "SUM"}}
    assert cfg["training"]["batch_size"] == 1000 and cfg["training"]["num_epochs"] == 10
    ns = cfg["training"]["negative_sampling"]
    assert (ns["num_chunks"], ns["negatives_per_positive"], ns["degree_fraction"], ns["filtered"], ns["local_filter_mode"]) == (1, 1000, 0.0, False, "DEG")
    assert cfg["storage"]["dataset"]["num_nodes"] == 100 and cfg["storage"]["dataset"]["num_train"] == 900  # dataset.yaml merged in
    assert cfg["storage"]["model_dir"] == os.path.join(ddir, "model_0")   # marius_config.py:47-56
    os.makedirs(os.path.join(ddir, "model_0"))
    assert C.load_config(path)["storage"]["model_dir"] == os.path.join(ddir, "model_1")
    assert isinstance(cfg["model"]["random_seed"], int)
    assert C.embedding_dim(cfg) == 50


def test_filtered_sampler_overrides(tmp_path):
    """config.cpp:365-376: filtered -> one chunk, every node (-1), no degree negatives; applies to training as well as evaluation."""
    path, _ = write(tmp_path, {"evaluation": {"negative_sampling": {"filtered": True, "num_chunks": 7, "negatives_per_positive": 33, "degree_fraction": 0.5}},
                              "training": {"negative_sampling": {"filtered": True}}})
    cfg = C.load_config(path)
    for section in ("training", "evaluation"):
        ns = cfg[section]["negative_sampling"]
        assert (ns["num_chunks"], ns["negatives_per_positive"], ns["degree_fraction"], ns["local_filter_mode"]) == (1, -1, 0.0, "DEG")


def test_partition_buffer_options(tmp_path):
    """datatypes.py:161-185: defaults, capacity clamped to the partition count, at least two partitions / capacity two."""
    path, _ = write(tmp_path, {"storage": {"embeddings": {"type": "PARTITION_BUFFER", "options": {"num_partitions": 4, "buffer_capacity": 9}}}})
    o = C.load_config(path)["storage"]["embeddings"]["options"]
    assert o["buffer_capacity"] == 4 and o["prefetching"] is True and o["edge_bucket_ordering"] == "COMET" and o["fine_to_coarse_ratio"] == 1
    assert o["randomly_assign_edge_buckets"] is True and o["num_cache_partitions"] == 0
    for bad in ({"num_partitions": 1}, {"buffer_capacity": 1}):
        path, _ = write(tmp_path, {"storage": {"embeddings": {"type": "PARTITION_BUFFER", "options": bad}}})
        with pytest.raises(ValueError):
            C.load_config(path)
    path, _ = write(tmp_path, {"storage": {"embeddings": {"type": "PARTITION_BUFFER"}}, "training": {"negative_sampling": {"filtered": True}}})
    with pytest.raises(NotImplementedError):
        C.load_config(path)


def test_out_of_scope_settings_are_refused_not_ignored(tmp_path):
    for user, exc in (({"model": {"learning_task": "NODE_CLASSIFICATION"}}, NotImplementedError),
                      ({"model": {"encoder": {"layers": [[{"type": "EMBEDDING", "output_dim": 8}], [{"type": "GNN", "output_dim": 8}]]}}}, NotImplementedError),
                      ({"storage": {"embeddings": {"type": "FLAT_FILE"}}}, NotImplementedError)):
        path, _ = write(tmp_path, user)
        with pytest.raises(exc):
            C.load_config(path)
    path, ddir = write(tmp_path, {})
    os.remove(os.path.join(ddir, "dataset.yaml"))
    with pytest.raises(ValueError):
        C.load_config(path)


def test_marius_train_refuses_cpu_device(tmp_path):
    """No CPU fallback: device_type cpu is an error, not a silent slow path."""
    from marius_amd.marius_train import marius_train

    path, _ = write(tmp_path, {"storage": {"device_type": "cpu"}})
    with pytest.raises(Exception) as e:
        marius_train(C.load_config(path), log=lambda *a: None)
    assert "no CPU path" in str(e.value) or "MI355X" in str(e.value) or "HIP" in str(e.value) or "cuda" in str(e.value).lower()


def test_model_dir_resolution_follows_the_reference(tmp_path):
    """marius_config.py:47-56 (get_model_dir_path) and :875-896 (infer_model_dir): training creates the first free model_<i> (at most
    model_10); marius_eval and resume_training without resume_from_checkpoint look the latest existing one up instead of creating one."""
    path, ddir = write(tmp_path, {})
    assert C.load_config(path, train=False)["storage"]["model_dir"] == os.path.join(ddir, "model_0")   # nothing to fall back to
    os.makedirs(os.path.join(ddir, "model_0"))
    os.makedirs(os.path.join(ddir, "model_1"))
    assert C.load_config(path)["storage"]["model_dir"] == os.path.join(ddir, "model_2")               # a training run: fresh directory
    cfg = C.load_config(path, train=False)                                                              # marius_eval: the latest existing
    assert cfg["storage"]["model_dir"] == os.path.join(ddir, "model_1") and cfg["_creates_model_dir"] is False
    rpath, _ = write(tmp_path, {"training": {"resume_training": True}})
    assert C.load_config(rpath)["storage"]["model_dir"] == os.path.join(ddir, "model_1")               # resume without a checkpoint dir
    cpath, _ = write(tmp_path, {"training": {"resume_training": True, "resume_from_checkpoint": str(tmp_path / "ckpt")}})
    assert C.load_config(cpath)["storage"]["model_dir"] == os.path.join(ddir, "model_2")               # resume_from_checkpoint: a new directory
    # an explicit model_dir holding model.pt is used as it is; the search stops at model_10
    mdir = tmp_path / "mine"
    mdir.mkdir()
    (mdir / "model.pt").write_bytes(b"x")
    upath, _ = write(tmp_path, {"storage": {"model_dir": str(mdir)}})
    assert C.load_config(upath, train=False)["storage"]["model_dir"] == str(mdir)
    for i in range(2, 11):
        os.makedirs(os.path.join(ddir, "model_%d" % i))
    path, _ = write(tmp_path, {})  # (write() reuses one file name)
    assert C.load_config(path)["storage"]["model_dir"] == os.path.join(ddir, "model_10")


def test_marius_eval_does_not_create_a_model_dir(tmp_path):
    """marius_eval on a dataset nobody trained on reports the missing directory instead of creating model_0 (which would shift the index of
    every later run)."""
    from marius_amd.marius_train import marius_train

    path, ddir = write(tmp_path, {})
    cfg = C.load_config(path, train=False)
    with pytest.raises(Exception):
        marius_train(cfg, log=lambda *a: None, train=False)
    assert not os.path.exists(cfg["storage"]["model_dir"])


def test_embedding_layer_options_are_honoured_or_refused(tmp_path):
    """LayerConfig of the embedding layer (marius_config.py:190-199).  `init` selects the node-table initialisation (initialization.cpp:67-119);
    `bias` / `bias_init` / `activation` (Layer::post_hook, layer.cpp:9-16) are honoured since round 6 (GeneralEncoder + marius_layer_post_hook); a
    per-layer optimizer is not implemented and is an ERROR — never accepted and ignored."""
    import torch

    def layer(**kw):
        return {"model": {"encoder": {"layers": [[dict({"type": "EMBEDDING", "output_dim": 16}, **kw)]]}}}

    path, _ = write(tmp_path, layer())
    lay = C.load_config(path)["model"]["encoder"]["layers"][0][0]
    assert lay["init"] == {"type": "GLOROT_UNIFORM", "options": {}} and lay["output_dim"] == 16
    # the reference's defaults spelled out are accepted
    path, _ = write(tmp_path, layer(bias=False, activation="none", input_dim=-1, optimizer={"type": "DEFAULT"}, init={"type": "uniform", "options": {"scale_factor": 0.25}}))
    cfg = C.load_config(path)
    assert C.embedding_init(cfg) == {"type": "UNIFORM", "options": {"scale_factor": 0.25}}
    path, _ = write(tmp_path, layer(bias=True, activation="relu", bias_init={"type": "CONSTANT", "options": {"constant": 0.5}}))
    lay = C.load_config(path)["model"]["encoder"]["layers"][0][0]
    assert lay["bias"] is True and lay["activation"] == "RELU" and lay["bias_init"] == {"type": "CONSTANT", "options": {"constant": 0.5}}
    path, _ = write(tmp_path, layer(activation="SIGMOID"))
    lay = C.load_config(path)["model"]["encoder"]["layers"][0][0]
    assert lay["bias"] is False and lay["activation"] == "SIGMOID" and lay["bias_init"]["type"] == "ZEROS"
    for bad, exc in ((dict(activation="TANH"), ValueError), (dict(bias=True, bias_init={"type": "XAVIER"}), ValueError), (dict(optimizer={"type": "ADAM"}), NotImplementedError),
                     (dict(options={"type": "GRAPH_SAGE"}), NotImplementedError), (dict(init={"type": "XAVIER"}), ValueError), (dict(input_dim=8), ValueError),
                     (dict(init={"type": "ZEROS", "options": {"constant": 1.0}}), ValueError), (dict(dropout=0.5), ValueError), (dict(output_dim=0), ValueError)):
        path, _ = write(tmp_path, layer(**bad))
        with pytest.raises(exc):
            C.load_config(path)
    # initialize_rows: every distribution of initialization.cpp:67-95; the GLOROT scale comes from the FULL table shape (initialize_subtensor)
    g = torch.Generator().manual_seed(0)
    rows, d, fans = 4000, 16, (1_000_000, 16)
    t = C.initialize_rows(C.check_init(None, "x"), rows, d, fans, "cpu", g)
    limit = (6.0 / (fans[0] + fans[1])) ** 0.5
    assert t.shape == (rows, d) and float(t.abs().max()) <= limit and float(t.abs().max()) > 0.9 * limit
    t = C.initialize_rows(C.check_init({"type": "GLOROT_NORMAL"}, "x"), rows, d, fans, "cpu", g)
    assert abs(float(t.std()) / (2.0 / (fans[0] + fans[1])) ** 0.5 - 1) < 0.05
    t = C.initialize_rows(C.check_init({"type": "NORMAL", "options": {"mean": 2.0, "std": 0.5}}, "x"), rows, d, fans, "cpu", g)
    assert abs(float(t.mean()) - 2.0) < 0.02 and abs(float(t.std()) - 0.5) < 0.02
    t = C.initialize_rows(C.check_init({"type": "UNIFORM", "options": {"scale_factor": 0.25}}, "x"), rows, d, fans, "cpu", g)
    assert 0.24 < float(t.abs().max()) <= 0.25
    assert float(C.initialize_rows(C.check_init({"type": "CONSTANT", "options": {"constant": 3.0}}, "x"), 5, d, fans, "cpu").min()) == 3.0
    assert float(C.initialize_rows(C.check_init({"type": "ONES"}, "x"), 5, d, fans, "cpu").max()) == 1.0
    assert float(C.initialize_rows(C.check_init({"type": "ZEROS"}, "x"), 5, d, fans, "cpu").abs().max()) == 0.0
