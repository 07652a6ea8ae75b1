"""`marius_train <config.yaml>` — drop-in entry point for link-prediction training on the MI355X.

Mirrors marius_init + marius_train of the reference (src/cpp/src/marius.cpp:38-163; console script
src/python/console_scripts/marius_train.py): load the configuration, seed the generator, build the decoder / loss / optimizers,
open the dataset's raw binaries (edges/train_edges.bin [E,3] int32, written by marius_preprocess) as DEVICE_MEMORY storage, create
<model_dir>/embeddings.bin + embeddings_state.bin, then per epoch: SynchronousTrainer::train(1), evaluation on the validation / test
edges, checkpoint of the node table.  All compute runs in the C++ host classes (marius_amd.host()) over the HIP kernels.
"""
import math
import os
import sys
import time

import torch
import yaml

from . import config as C


def _edge_file(ddir, split):
    return os.path.join(ddir, "edges", "%s_edges.bin" % split)


METRICS = ["MRR", "Mean Rank", "Hits@1", "Hits@3", "Hits@5", "Hits@10", "Hits@50", "Hits@100"]


def marius_train(cfg, log=print, train=True):
    """train=True: marius_train (marius.cpp:98-163).  train=False: marius_eval (marius.cpp:165-184) — load the model directory written by a
    training run (embeddings.bin, model.pt, metadata.csv) and evaluate the test edges once."""
    import marius_amd

    H = marius_amd.host()
    if cfg["storage"]["device_type"] == "cpu":
        raise RuntimeError("device_type cpu: this build has no CPU path (MI355X only); set storage.device_type: cuda")
    dev = torch.device("cuda", 0)
    ds = cfg["storage"]["dataset"]
    ddir, mdir = ds["dataset_dir"], cfg["storage"]["model_dir"]
    creates = cfg.pop("_creates_model_dir", train)  # run-mode state of load_config, not a key of the reference schema: never written to full_config.yaml
    if creates:
        os.makedirs(mdir, exist_ok=True)
    elif not os.path.isdir(mdir):  # marius_eval / resume_training: the directory of an earlier run is looked up, never created
        raise FileNotFoundError("model directory %s does not exist: nothing to %s (run marius_train first, or set storage.model_dir)" % (
            mdir, "resume" if train else "evaluate"))
    if train:
        with open(os.path.join(mdir, "full_config.yaml"), "w") as f:
            yaml.safe_dump(cfg, f)
    d = C.embedding_dim(cfg)
    num_nodes, R = int(ds["num_nodes"]), int(ds.get("num_relations", 1))
    cols = 3 if R > 1 else 2  # io.cpp:42-45
    edge_dtype = torch.int32 if cfg["storage"]["edges"]["options"]["dtype"] == "int" else torch.int64
    seed = int(cfg["model"]["random_seed"])
    gen = H.MariusGenerator(seed)  # torch::manual_seed(seed) stream (marius.cpp:47)
    torch.manual_seed(seed)

    # ---- model (initModelFromConfig, model.cpp:381-440)
    dec_cfg = cfg["model"]["decoder"]
    dec_cls = {"DISTMULT": H.DistMult, "COMPLEX": H.ComplEx, "TRANSE": H.TransE}[dec_cfg["type"]]
    method = getattr(H.EdgeDecoderMethod, dec_cfg["options"].get("edge_decoder_method", "CORRUPT_NODE"))
    decoder = dec_cls(R, d, dev, bool(dec_cfg["options"].get("inverse_edges", True)), method)
    lopt = cfg["model"]["loss"].get("options") or {}
    # getLossFunction (loss.cpp:189-209); RankingLossOptions.margin defaults to 0.1 (datatypes.py:41-43)
    loss = H.getLossFunction(str(cfg["model"]["loss"]["type"]).upper(), str(lopt.get("reduction", "SUM")), float(lopt.get("margin", 0.1)))
    model = H.Model(decoder, loss, H.LinkPredictionReporter(), dev)
    layer = cfg["model"]["encoder"]["layers"][0][0]
    if layer["bias"] or layer["activation"] != "NONE":  # Layer::post_hook of the embedding layer (layer.cpp:9-16); bias init: initialize_tensor(bias_init, {d})
        act = getattr(H.ActivationFunction, layer["activation"])
        if layer["bias"]:
            model.set_encoder(H.GeneralEncoder(d, True, act, dev, C.initialize_rows(layer["bias_init"], 1, d, (1, d), dev).reshape(d)))
        else:
            model.set_encoder(H.GeneralEncoder(d, False, act, dev))
    dopt = cfg["model"]["dense_optimizer"]
    o = dopt.get("options") or {}
    dtype_ = str(dopt.get("type", "ADAGRAD")).upper()
    # option defaults of the reference's config classes (src/python/tools/configuration/datatypes.py:54-80: Adagrad eps 1e-10; Adam eps 1e-8, betas 0.9 / 0.999)
    model.setup_optimizer(dtype_, float(o.get("learning_rate", 0.1)), float(o.get("eps", 1e-8 if dtype_ == "ADAM" else 1e-10)),
                          float(o.get("beta_1", 0.9)), float(o.get("beta_2", 0.999)), float(o.get("weight_decay", 0.0)), bool(o.get("amsgrad", False)))
    sp = cfg["model"].get("sparse_optimizer") or cfg["model"]["dense_optimizer"]
    model.sparse_lr = float(sp["options"]["learning_rate"])  # model.cpp:425-429: only the learning rate is read

    # ---- storage (initializeStorage, io.cpp:433-448)
    def edges(split, n):
        st = H.InMemory(_edge_file(ddir, split), int(n), cols, edge_dtype, dev)
        st.load()
        return st

    train_edges = edges("train", ds["num_train"])
    resume = (not train) or bool(cfg["training"].get("resume_training", False))
    # marius.cpp:59-90: training resumes from model_dir or from training.resume_from_checkpoint (whose files marius_config.py:932-934 copies
    # into the model directory), evaluation reads model_dir or evaluation.checkpoint_dir
    src_dir = (cfg["training"].get("resume_from_checkpoint") or "") if train else (cfg["evaluation"].get("checkpoint_dir") or "")
    if src_dir and os.path.abspath(src_dir) != os.path.abspath(mdir):
        import shutil

        for name in ("embeddings.bin", "embeddings_state.bin", "model.pt", "model_state.pt", "metadata.csv"):
            if os.path.exists(os.path.join(src_dir, name)):
                shutil.copyfile(os.path.join(src_dir, name), os.path.join(mdir, name))
        resume = True
    epochs_processed = 0
    if resume:  # Checkpointer::load needs these three (checkpointer.cpp:56-74): say which one is missing instead of a bare FileNotFoundError
        missing = [n for n in ("metadata.csv", "model.pt", "embeddings.bin") if not os.path.exists(os.path.join(mdir, n))]
        if missing:
            raise FileNotFoundError("%s: cannot %s — %s missing (was a training run with save_model: true written here?)" % (
                mdir, "resume training" if train else "evaluate", ", ".join(missing)))
    if resume and os.path.exists(os.path.join(mdir, "metadata.csv")):
        epochs_processed = int(open(os.path.join(mdir, "metadata.csv")).read().split("\n")[1])  # CheckpointMeta.num_epochs (marius.cpp:75)
    emb_cfg = cfg["storage"]["embeddings"]
    pb_cfg = emb_cfg["type"] == "PARTITION_BUFFER"
    partitioned = train and pb_cfg
    # storage.full_graph_evaluation: false (graph_storage.cpp:104-112): evaluate buffer state by buffer state instead of loading the whole table
    part_eval = pb_cfg and not bool(cfg["storage"].get("full_graph_evaluation", True))
    emb_path, state_path = os.path.join(mdir, "embeddings.bin"), os.path.join(mdir, "embeddings_state.bin")
    eval_emb = None
    pb_opts = None
    if pb_cfg:
        po = emb_cfg["options"]
        pb_opts = H.PartitionBufferOptions()
        pb_opts.num_partitions, pb_opts.buffer_capacity = int(po["num_partitions"]), int(po["buffer_capacity"])
        pb_opts.prefetching, pb_opts.fine_to_coarse_ratio = bool(po["prefetching"]), int(po["fine_to_coarse_ratio"])
        pb_opts.num_cache_partitions = int(po["num_cache_partitions"])
        pb_opts.edge_bucket_ordering = getattr(H.EdgeBucketOrdering, str(po["edge_bucket_ordering"]).upper())
        pb_opts.randomly_assign_edge_buckets = bool(po["randomly_assign_edge_buckets"])
    if partitioned:
        # out-of-core node table (io.cpp:300-330 -> PartitionBufferStorage over <model_dir>/embeddings.bin + embeddings_state.bin; the
        # train edges are sorted by edge bucket and edges/train_partition_offsets.txt lists the bucket sizes, io.cpp:110-121)
        opts = pb_opts
        train_edges.readPartitionSizes(os.path.join(ddir, "edges", "train_partition_offsets.txt"))  # io.cpp:110-121
        if not resume:
            rows = max(1, (256 << 20) // (4 * d))
            with open(emb_path, "wb") as fe, open(state_path, "wb") as fs:  # initialised in slabs: the table need not fit anywhere at once
                for lo in range(0, num_nodes, rows):
                    n = min(rows, num_nodes - lo)
                    fe.write(C.initialize_rows(C.embedding_init(cfg), n, d, (num_nodes, d), dev).cpu().numpy().tobytes())
                    fs.write(bytes(4 * d * n))
        else:
            meta = open(os.path.join(mdir, "metadata.csv")).read().split("\n")
            if not int(meta[6]):
                raise RuntimeError("checkpoint in %s has no model" % mdir)
            model.load(os.path.join(mdir, ""), train)
            if int(meta[4]) and not os.path.exists(state_path):
                raise FileNotFoundError("%s: metadata.csv says the optimizer state was saved, but embeddings_state.bin is missing" % mdir)
            if not os.path.exists(state_path):
                with open(state_path, "wb") as fs:
                    fs.write(bytes(4 * d * num_nodes))
        emb = H.PartitionBufferStorage(emb_path, num_nodes, d, opts, dev)
        state = H.PartitionBufferStorage(state_path, num_nodes, d, opts, dev)
        # full_graph_evaluation (graph_storage.cpp:104-112): evaluation reads the whole table from the file the epoch wrote back
        eval_emb = H.InMemory(emb_path, num_nodes, d, torch.float32, dev)
    elif resume:  # Checkpointer::load (checkpointer.cpp:56-74): the model directory of an earlier run
        meta = open(os.path.join(mdir, "metadata.csv")).read().split("\n")
        if not int(meta[6]):
            raise RuntimeError("checkpoint in %s has no model" % mdir)
        emb = H.InMemory(os.path.join(mdir, "embeddings.bin"), num_nodes, d, torch.float32, dev)
        emb.load()
        if train and int(meta[4]) and os.path.exists(os.path.join(mdir, "embeddings_state.bin")):
            state = H.InMemory(os.path.join(mdir, "embeddings_state.bin"), num_nodes, d, torch.float32, dev)
            state.load()
        else:
            state = H.InMemory(torch.zeros((num_nodes, d), dtype=torch.float32, device=dev))
            state.filename = os.path.join(mdir, "embeddings_state.bin")
        model.load(os.path.join(mdir, ""), train)
    else:
        # model.encoder.layers[0][0].init (default GLOROT_UNIFORM over the table shape, initialization.cpp:26-41, 98-119)
        table = C.initialize_rows(C.embedding_init(cfg), num_nodes, d, (num_nodes, d), dev)
        emb = H.InMemory(table)
        emb.filename = os.path.join(mdir, "embeddings.bin")
        state = H.InMemory(torch.zeros_like(table))
        state.filename = os.path.join(mdir, "embeddings_state.bin")

    tr, ev = cfg["training"], cfg["evaluation"]

    def sampler(ns):
        return H.CorruptNodeNegativeSampler(int(ns["num_chunks"]), int(ns["negatives_per_positive"]), float(ns["degree_fraction"]),
                                            bool(ns["filtered"]), getattr(H.LocalFilterMode, ns.get("local_filter_mode", "DEG")), gen)

    loader = H.DataLoader(train_edges, emb, state, sampler(tr["negative_sampling"]), gen, int(tr["batch_size"]), True)
    pipe = tr.get("pipeline", {})
    if pipe.get("sync", True):
        trainer = H.SynchronousTrainer(loader, model)
    else:
        # training.pipeline.sync: false (trainer.cpp:35-74): admission control with staleness_bound; parameters of host-resident tables are
        # read at admission (stale by up to staleness_bound - 1 updates), device-resident ones by the compute stage (no staleness)
        trainer = H.PipelineTrainer(loader, model, int(pipe.get("staleness_bound", 16)), emb_cfg["type"] == "HOST_MEMORY" and not partitioned)
    evals = {}
    eval_edges = {}
    for split, key in (("validation", "num_valid"), ("test", "num_test")):
        n = int(ds.get(key, -1))
        if n > 0 and os.path.exists(_edge_file(ddir, split)):
            eval_edges[split] = edges(split, n)
            table_for_eval = eval_emb or emb
            if part_eval:
                # bucket-sorted evaluation edges + edges/<split>_partition_offsets.txt (io.cpp:116-121); a buffer of its own over the same file
                # (the training buffer is unloaded between epochs)
                eval_edges[split].readPartitionSizes(os.path.join(ddir, "edges", "%s_partition_offsets.txt" % split))
                table_for_eval = H.PartitionBufferStorage(emb_path, num_nodes, d, pb_opts, dev)
            evals[split] = H.SynchronousEvaluator(H.DataLoader(eval_edges[split], table_for_eval, None, sampler(ev["negative_sampling"]), gen,
                                                               int(ev["batch_size"]), False), model)
    if evals and int(ev.get("epochs_per_eval", 1)) == 1 and not partitioned:
        # every epoch is followed by the evaluation passes, which draw from the same generator stream (their own permutation + their sampling):
        # tell the training loader, so that the permutation it draws ahead for the next epoch starts where the generator will really be
        # (a wrong count only costs the prediction: the loader verifies the generator state before adopting a permutation)
        loader.words_between_epochs = sum(e.dataloader.wordsPerEpoch(True) for e in evals.values())
    f_train, f_eval = bool(tr["negative_sampling"].get("filtered", False)), bool(ev["negative_sampling"].get("filtered", False)) and bool(evals)
    if f_train or f_eval:
        # GraphModelStorage::sortAllEdges (graph_storage.cpp:745-777; dataloader.cpp:592-598 calls it for any filtered sampler): train +
        # validation + test edges are the "true" edges that must not count as negatives
        all_edges = torch.cat([train_edges.data.to(torch.int64)] + [e.data.to(torch.int64) for e in eval_edges.values()])
        if f_eval:
            for e in evals.values():
                e.dataloader.graph.sortAllEdges(all_edges)
        if f_train:
            loader.graph.sortAllEdges(all_edges)

    def run_eval(split, rec):
        t0 = time.time()
        if eval_emb is not None and not part_eval:
            eval_emb.load()
        r = evals[split].evaluate()
        if eval_emb is not None and not part_eval:
            eval_emb.unload(False)
        log("%s evaluation (%.0f ms): %s" % (split, (time.time() - t0) * 1e3, ", ".join("%s: %.6f" % kv for kv in zip(METRICS, r))))
        rec[split] = dict(zip(METRICS, r))

    if not train:  # marius_eval: evaluator->evaluate(false) = the test edges
        rec = {}
        if "test" in evals:
            run_eval("test", rec)
        return [rec]

    def write_meta(directory, num_epochs):
        with open(os.path.join(directory, "metadata.csv"), "w") as f:
            # name, num_epochs, checkpoint_id, link_prediction, has_state, has_encoded, has_model  (CheckpointMeta, checkpointer.h:12-21)
            f.write("checkpoint\n%d\n-1\n1\n1\n%d\n1\n" % (num_epochs, 1 if cfg["storage"].get("export_encoded_nodes", False) else 0))

    def save_into(directory, num_epochs):
        """Checkpointer::save (checkpointer.cpp:39-54): node table + optimizer state as raw binaries, model.pt / model_state.pt as
        torch::serialize archives with the reference's keys (Model::save, model.cpp:82-106), metadata.csv (checkpointer.cpp:104-116)."""
        if not partitioned:  # the partition buffer wrote both files back at the end of the epoch
            emb.write()
            state.write()
        model.save(os.path.join(directory, ""))
        write_meta(directory, num_epochs)

    ck = tr.get("checkpoint") or {}
    interval = int(ck.get("interval", -1))
    save_model = cfg["storage"].get("save_model", True) and tr.get("save_model", True)
    results = []
    for epoch in range(1, int(tr["num_epochs"]) + 1):
        log("################ Starting training epoch %d ################" % (epochs_processed + epoch))
        trainer.train(1)
        log("Epoch Runtime: %dms" % int(trainer.last_epoch_seconds * 1e3))
        log("Edges per Second: %.2f" % trainer.last_edges_per_second)  # trainer.cpp:156-159
        rec = {"epoch": epoch, "edges_per_second": trainer.last_edges_per_second}
        if epoch % int(ev.get("epochs_per_eval", 1)) == 0:
            for split in evals:
                run_eval(split, rec)
        rec["shuffle_ahead"] = [int(loader.shuffle_ahead_hits), int(loader.shuffle_ahead_misses)]  # permutations adopted from the host thread / drawn serially
        results.append(rec)
        if save_model and interval > 0 and epoch % interval == 0 and epoch < int(tr["num_epochs"]):
            # Checkpointer::create_checkpoint (checkpointer.cpp:18-37): <model_dir>/checkpoint_<epochs>/ via a _tmp directory and a rename.  The
            # reference copies the embeddings file as it was LAST written and only then writes the current table to model_dir; here the
            # table is written first, so the checkpoint holds the state it is named after.
            import shutil

            done = epochs_processed + epoch
            tmp, final = os.path.join(mdir, "checkpoint_%d_tmp" % done), os.path.join(mdir, "checkpoint_%d" % done)
            os.makedirs(tmp, exist_ok=True)
            save_into(tmp, done)
            shutil.copyfile(emb_path, os.path.join(tmp, "embeddings.bin"))
            if bool(ck.get("save_state", False)):
                shutil.copyfile(state_path, os.path.join(tmp, "embeddings_state.bin"))
            if os.path.exists(final):
                shutil.rmtree(final)
            os.rename(tmp, final)
    if save_model:
        save_into(mdir, epochs_processed + int(tr["num_epochs"]))
    return results


def marius_eval(cfg, log=print):
    return marius_train(cfg, log=log, train=False)


def main(argv=None, train=True):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print("usage: %s <config.yaml>" % ("marius_train" if train else "marius_eval"))
        return 2
    marius_train(C.load_config(argv[0], train=train), train=train)
    return 0


if __name__ == "__main__":
    sys.exit(main())
