"""The GPU parity tests of the kernel files that have a CPU build (tests/emul/build_emul.py: rows.hip, rng.hip, encoder.hip, neighbor.hip compiled by
g++ against a shim that emulates HIP's execution model on host threads; since: segreduce.hip, sort_unique.hip, exchange.hip, lp_decoder.hip's generic level), run AS THEY ARE — the same test functions, imported from
tests/test_gpu_parity.py / tests/test_gpu_host.py — with the ctypes layer pointed at the emulated library for the duration of a test and "the device"
being the host.  Work-items are fibers of one thread (tests/emul/common.h), so most tests keep the shapes they have on the GPU; the few that exist only at
the bench's shape take minutes each and run with MARIUS_EMUL_FULL=1.  What this adds to the `-m gpu` runs: the kernels' logic is checked in the CPU suite of every round, and once more under
AddressSanitizer + UBSan (the GPU pool has no sanitizer builds).  What it does not replace: the hipcc build, the hardware, the timing.
Test infrastructure: marius_amd/ has no switch to reach the emulated library; the redirection is a monkeypatch that lives in this file."""
import ctypes as C
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emul"))
sys.path.insert(0, HERE)

CPU = torch.device("cpu")
SAN = os.environ.get("MARIUS_EMUL_SANITIZE") == "1"   # inside the sanitizer run: fewer and smaller cases
FULL = os.environ.get("MARIUS_EMUL_FULL") == "1"      # the bench-shape cases too (minutes each: profiles/r6_emulated_full_shapes.txt)


@pytest.fixture(scope="module")
def emulated_library():
    import build_emul
    from marius_amd import hip

    lib = C.CDLL(build_emul.build(sanitize=os.environ.get("MARIUS_EMUL_SANITIZE") == "1"))
    bound = []
    for name, (res, args) in hip.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype, getattr(lib, name).argtypes = res, args
            bound.append(name)
    assert {"marius_gather_rows", "marius_scatter_add_rows", "marius_mt19937_fill", "marius_sample_negatives", "marius_layer_post_hook", "marius_nbr_gather"} <= set(bound)
    return lib


@pytest.fixture()
def HE(emulated_library, monkeypatch):
    """marius_amd.hip with lib() -> the emulated library, host tensors accepted, no stream; Tensor.to always copies (a parity test's `x.to(dev)` must not
    alias its reference copy when the device IS the host)"""
    from marius_amd import hip

    monkeypatch.setattr(hip, "lib", lambda: emulated_library)
    monkeypatch.setattr(hip, "_dev", lambda t: t)
    monkeypatch.setattr(hip, "stream_ptr", lambda stream=None: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)

    class _Stream:  # the emulated library runs every launch to completion before it returns: streams and events have nothing to order
        cuda_stream = 0

        def __init__(self, *a, **k):
            pass

        def wait_event(self, e):
            pass

        def wait_stream(self, s):
            pass

        def record_event(self, e=None):
            return e

        def synchronize(self):
            pass

    class _Event:
        def __init__(self, *a, **k):
            pass

        def record(self, s=None):
            pass

        def wait(self, s=None):
            pass

        def synchronize(self):
            pass

    import contextlib

    monkeypatch.setattr(torch.cuda, "Stream", _Stream)
    monkeypatch.setattr(torch.cuda, "Event", _Event)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.Tensor, "is_pinned", lambda self, *a, **k: True)
    orig_to = torch.Tensor.to

    def to_copy(self, *a, **k):
        k.setdefault("copy", True)
        return orig_to(self, *a, **k)

    monkeypatch.setattr(torch.Tensor, "to", to_copy)
    return hip


def test_rows_kernels_on_the_cpu_build(HE):
    """rows.hip: InMemory::indexRead / indexAdd (storage.cpp:606-673), Batch::accumulateGradients' rule (batch.cpp:62-79), dense Adagrad / Adam (optim.cpp)"""
    import test_gpu_parity as TP

    for d in ((2, 100) if SAN else (2, 50, 100, 128, 400)):
        TP.test_gather_scatter_rows(HE, CPU, d)
    TP.test_gather_empty_and_errors(HE, CPU)
    TP.test_adagrad_rule_bit_exact(HE, CPU)
    TP.test_dense_adagrad_step(HE, CPU)
    for amsgrad, wd in ((False, 0.0), (True, 0.0), (False, 0.01)):
        TP.test_dense_adam_step_matches_reference_ops(HE, CPU, amsgrad, wd)


def test_sampler_kernels_on_the_cpu_build(HE):
    """rng.hip: ATen's MT19937 stream, CorruptNodeNegativeSampler::getNegatives (negative.cpp:328-366), RandomEdgeSampler::getEdges (edge.cpp:12-14)"""
    import test_gpu_parity as TP

    for n in ((1, 625, 3000) if SAN else (1, 623, 624, 625, 5000, 100000)):
        TP.test_mt19937_device_stream_bit_exact(HE, CPU, n)
    shapes = ((6, 1, 5, 0.0, 6), (6, 3, 5, 0.5, 6), (1000, 10, 500, 0.0, 14541), (1000, 10, 500, 0.5, 14541), (64, 2, 16, 0.25, 2 ** 28 + 5))
    shapes += ((5000, 50, 1000, 0.0, 86054151),)
    for B, Cn, N, f, num_nodes in (shapes[:2] + shapes[4:5] if SAN else shapes):
        TP.test_negative_sampler_bit_exact(HE, CPU, B, Cn, N, f, num_nodes)
    TP.test_select_edges(HE, CPU)


def test_post_hook_kernels_on_the_cpu_build(HE):
    """encoder.hip: Layer::post_hook (layer.cpp:9-16) and its backward"""
    import test_gpu_host as TH

    for activation in ("NONE", "RELU", "SIGMOID"):
        for with_bias in (True, False):
            for n, d, pad in (((901, 100, 28), (64, 7, 3)) if SAN else ((1, 2, 0), (777, 50, 0), (333, 400, 0), (901, 100, 28), (64, 7, 3))):
                TH.test_layer_post_hook_kernels_match_the_reference_ops(CPU, activation, with_bias, n, d, pad)


def test_unique_map_kernels_on_the_cpu_build(HE):
    """sort_unique.hip: map_tensors (util.cpp:180-205) — the hand-written radix sort and run-head scan, the merge of ascending runs, and the whole map
    chain of a batch as ONE persistent launch (marius_prepare_maps: the work-item queue is drained by the emulated workgroups one after another, which
    is the single-workgroup case its design promises to complete) — against torch.unique / the separate launches, bit for bit"""
    import test_gpu_parity as TP

    sizes = ((1, 5), (12, 6), (4097, 1 << 20), (5001, (1 << 36) - 5)) if SAN else ((1, 5), (12, 6), (12000, 14541), (4095, 1 << 20), (4096, 1 << 20), (4097, 1 << 20), (16385, 1),
                                                                                 (70001, (1 << 36) - 5), (70001, (1 << 40) + 3)) + (((200000, 86054151),) if FULL else ())
    for n, hi in sizes:
        TP.test_sort_unique_matches_map_tensors(HE, CPU, n, hi)
    TP.test_sort_unique_empty(HE, CPU)
    if FULL:
        TP.test_sort_unique_reuses_its_workspace_across_calls_and_sizes(HE, CPU)
    shapes = ((250, 5, 40, 4000, 11, 3, False), (7, 1, 3, 50, 2, 3, False), (1, 1, 1, 2, 1, 3, False), (2049, 3, 683, 99999, 5, 3, True))
    big = ((1000, 10, 500, 14541, 237, 3, False), (4096, 1, 4096, 1 << 20, 1, 2, False)) + (((50000, 50, 1000, 86054151, 14824, 3, True),) if FULL else ())
    for B, Cn, N, num_nodes, R, cols, hubs in (shapes[:3] if SAN else shapes + big):
        TP.test_prepare_maps_one_launch_equals_the_separate_launches(HE, CPU, B, Cn, N, num_nodes, R, cols, hubs)
    for runs in (([1], [0, 7, 0], [1000, 0, 0, 3]) if SAN else ([1], [5000], [0, 7, 0], [3000, 2500, 4000, 1], [25000] * 8, [1000, 0, 0, 3], [17] * 64)):
        TP.test_merge_unique_runs_equals_sort_unique(HE, CPU, runs)


def test_exchange_kernels_on_the_cpu_build(HE):
    """exchange.hip: the fixed-capacity halves of the sharded row exchange and the header record against the numpy restatement (oracle/exchange_oracle.py)"""
    import test_gpu_parity as TP

    for world, slack in (((2, 1.5),) if SAN else ((1, 1.0), (2, 1.5), (8, 1.5), (4, 1.0))):
        TP.test_fixed_capacity_exchange_halves_against_numpy(HE, CPU, world, slack)


def test_decoder_generic_kernels_on_the_cpu_build(HE, monkeypatch):
    """lp_decoder.hip at its generic level (the tuned levels live in lp_fast / lp_res / lp_flash .hip, which have no CPU build): select_relations,
    the relation operators, Dot / L2 scores on the emulated v_mfma_f32_32x32x2_f32 (a wave collective in the shim), pad_and_reshape, node_corrupt_forward,
    SoftmaxCE and the six other losses, the hand-derived backward, ranks — decoder_methods.cpp:57-114, relation_operators.cpp:7-47, comparators.cpp:7-73,
    loss.cpp:50-187, reporting.cpp:55-57 — through marius_lp_plan / _forward / _loss / _backward against the oracle, as the GPU suite does"""
    import test_gpu_parity as TP

    shapes = ((6, 3, 5, 2), (5, 4, 6, 8)) if SAN else ((6, 3, 5, 2), (100, 10, 50, 50), (5, 4, 6, 8), (2, 4, 64, 100), (1, 3, 33, 100))
    for decoder in ("DISTMULT", "COMPLEX", "TRANSE"):
        for use_inverse in (True, False):
            for B, Cn, N, d in (shapes if decoder == "COMPLEX" else shapes[:3]):   # (the whole-padding-chunk shapes: one decoder is enough here)
                TP.test_lp_forward_loss_backward(HE, CPU, decoder, use_inverse, B, Cn, N, d, "sum")
    if not SAN:
        TP.test_lp_forward_loss_backward(HE, CPU, "COMPLEX", True, 250, 7, 130, 100, "sum")   # B % C != 0, N not a multiple of the tile
        TP.test_lp_forward_loss_backward(HE, CPU, "TRANSE", False, 250, 7, 130, 100, "sum")
        TP.test_lp_forward_loss_backward(HE, CPU, "DISTMULT", True, 300, 4, 260, 200, "sum")  # rows wider than 128 columns
    for decoder in ("DISTMULT", "TRANSE"):
        TP.test_lp_mean_reduction_and_filter(HE, CPU, decoder)
    TP.test_lp_two_column_edges(HE, CPU)
    TP.test_lp_bad_edge_columns_raises(HE, CPU)
    TP.test_compute_ranks(HE, CPU)
    for loss in (TP.LOSSES[:2] if SAN else TP.LOSSES):   # (novlog: the generic backward with the loss-specific dL/dS — the form this build has)
        for cfg in ((("DISTMULT", False, 5, 4, 6, 8, "mean"),) if SAN else (("DISTMULT", False, 5, 4, 6, 8, "mean"),) +
                    ((("DISTMULT", True, 100, 10, 50, 50, "sum"),) if loss in ("RANKING", "BCE_WITH_LOGITS", "MSE") else ()) +
                    ((("COMPLEX", True, 250, 7, 130, 100, "mean"),) if loss == "SOFTPLUS" else ())):
            TP.test_lp_other_losses_forward_backward(HE, CPU, monkeypatch, True, loss, *cfg)


def test_whole_training_steps_on_the_cpu_build(HE):
    """SynchronousTrainer::train's step (trainer.cpp:106-138) through the API-granular device path — edge slice, MT19937 words, negatives (with and without
    the degree-based share and its filter), map_tensors, row gather, forward_lp, SoftmaxCE, backward, relation-table Adagrad, segmented node update —
    three consecutive steps against the CPU reference step: sampled ids and unique map bit-exact, scores / loss / tables within tolerance.  Nine of the
    library's kernel files in one test, none of them on a GPU."""
    import test_gpu_parity as TP

    # (TransE's third case stays with the GPU suite: its bound on the SMALL L2 distances — sqrt of a cancelling x^2 + y^2 - 2 x y — is calibrated on the
    # matrix pipe's internal summation order, which the emulated MFMA does not reproduce: 1.6e-4 x max here against the 1.0e-4 allowed)
    for decoder, f in ((("COMPLEX", 0.0),) if SAN else (("COMPLEX", 0.0), ("DISTMULT", 0.5))):
        TP.test_train_steps_match_cpu_reference_path(HE, CPU, decoder, f)


def test_true_edge_filter_kernels_on_the_cpu_build(HE):
    """eval_filter.hip: compute_filter_corruption's global branch (negative.cpp:50-293) through the C-ABI"""
    import test_gpu_zz_unverified_cfg4 as TZ

    for cols in (2, 3):
        for num_nodes, E, B in (((50, 400, 64), (7, 30, 9)) if SAN else ((50, 400, 64), (1000, 20000, 500), (7, 30, 9))):
            TZ.test_true_edge_filter_equals_the_reference_loop(HE, CPU, cols, num_nodes, E, B)


def test_cpu_build_under_address_and_undefined_behaviour_sanitizers():
    """every emulated test of this file (reduced shapes) and of tests/test_neighbor_emul_cpu.py again, in a python started under libasan with the kernel files built with
    -fsanitize=address,undefined: an out-of-bounds index, a misaligned or overflowing access in any emulated work-item aborts the run"""
    if os.environ.get("MARIUS_EMUL_SANITIZE") == "1":
        pytest.skip("already inside the sanitizer run")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan.so next to gcc")
    env = dict(os.environ, MARIUS_EMUL_SANITIZE="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # default: every kernel file is touched, the heaviest groups (decoder, segmented update, whole steps) by their cheapest cases only through the other
    # groups' dependencies; MARIUS_EMUL_SANITIZE_ALL=1: every emulated test of both files (run clean on the round's final tree: profiles/r6_sanitizer_pass.txt)
    select = ("not sanitizers" if os.environ.get("MARIUS_EMUL_SANITIZE_ALL") == "1" else
              "rows_kernels or sampler_kernels or post_hook_kernels or unique_map or exchange_kernels or true_edge or delta_ids or (one_hop and 3-2) or (dropout and 0.3) "
              "or (aggregation and GCN and 7) or (layered and fanouts1) or (three_layer and MEAN)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), os.path.join(HERE, "test_neighbor_emul_cpu.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", select], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=2400, cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-4000:]


def test_segmented_update_kernels_on_the_cpu_build(HE):
    """segreduce.hip: the sorted-unique, atomics-free sparse update — per-unique-row sums of the occurrence gradients, Batch::accumulateGradients' rule and
    InMemory::indexAdd on the unique rows (batch.cpp:62-79, storage.cpp:651-673), planned / tracked / grouped forms, the table magnitude scan"""
    import test_gpu_parity as TP

    for n, U, d in (((1, 1, 4), (1000, 900, 100)) if SAN else ((1, 1, 4), (1000, 900, 100), (4096, 3, 7)) + (((20000, 19000, 100),) if FULL else ())):
        TP.test_segment_sum_rows(HE, CPU, n, U, d)
    TP.test_segment_adagrad_scatter_matches_reference_update(HE, CPU)
    for n, num_nodes, power, d in ((33, 5, 1, 20), (1, 9, 1, 8), (700, 90, 2, 36)) + (((8000, 3000, 3, 100), (200000, 86054151, 1, 100)) if FULL else ()):
        TP.test_planned_segment_adagrad_scatter_is_bit_identical(HE, CPU, n, num_nodes, power, d)
    for n, rows, d, planned in ((33, 5, 20, False), (600, 200, 100, True)) + (() if SAN else ((8000, 3000, 100, True),)):
        TP.test_tracked_update_keeps_the_magnitude_bound(HE, CPU, n, rows, d, planned)
    for rows, d, ld in ((1000, 100, 100), (777, 50, 50), (513, 33, 33), (3, 7, 7), (4096, 100, 112)):
        TP.test_table_absmax_flat_strided_and_counted(HE, CPU, rows, d, ld)
    if FULL:  # the grouped-launch tests exist at the bench's shape only (200,000 occurrences x 100 columns): four minutes each on the emulator
        for planned in (True, False):
            TP.test_grouped_update_of_three_tables_equals_the_separate_updates(HE, CPU, planned)
        TP.test_group_with_a_reduce_only_job_equals_the_separate_calls(HE, CPU, True)
