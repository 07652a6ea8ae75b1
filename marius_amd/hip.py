"""ctypes binding of libmarius_hip.so (include/marius_hip.h).  Fails loudly when the library is missing:
there is NO CPU fallback anywhere in the product path.

The helpers take torch tensors only to obtain device pointers / the current HIP stream; the C-ABI itself sees plain
pointers and sizes.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmarius_hip.so")

OP_HADAMARD, OP_COMPLEX_HADAMARD, OP_TRANSLATION, OP_NOOP = 0, 1, 2, 3
CMP_DOT, CMP_L2, CMP_COSINE = 0, 1, 2
REDUCE_SUM, REDUCE_MEAN = 0, 1
LOSS = {"SOFTMAX_CE": 0, "RANKING": 1, "CROSS_ENTROPY": 2, "BCE_AFTER_SIGMOID": 3, "BCE_WITH_LOGITS": 4, "MSE": 5, "SOFTPLUS": 6}
LP_TRAIN_ONLY, LP_STORE_SCORES, LP_KEEP_DADJ = 1, 2, 4   # marius_lp_desc.flags
MT_STATE_WORDS = 625
ABI_VERSION = 11  # include/marius_hip.h MARIUS_HIP_ABI_VERSION


class MariusHipError(RuntimeError):
    pass


class LpDesc(C.Structure):
    _fields_ = [
        ("relop", C.c_int32), ("cmp", C.c_int32), ("d", C.c_int32), ("edge_cols", C.c_int32),
        ("B", C.c_int64), ("C", C.c_int32), ("N", C.c_int32), ("use_inverse", C.c_int32), ("reduction", C.c_int32),
        ("emb", C.c_void_p), ("emb_ld", C.c_int64), ("U", C.c_int64),
        ("edges", C.c_void_p), ("dst_neg", C.c_void_p), ("src_neg", C.c_void_p),
        ("rel", C.c_void_p), ("inv_rel", C.c_void_p), ("rel_ld", C.c_int64), ("R", C.c_int64),
        ("dst_filter", C.c_void_p), ("n_dst_filter", C.c_int64), ("src_filter", C.c_void_p), ("n_src_filter", C.c_int64),
        ("loss", C.c_int32), ("margin", C.c_float), ("flags", C.c_int32), ("free_cus", C.c_int32),
        ("absmax", C.c_void_p), ("absmax_rel", C.c_void_p),
        ("upd_occ_single", C.c_void_p), ("upd_state", C.c_void_p), ("upd_absmax", C.c_void_p), ("upd_lr", C.c_float), ("upd_eps", C.c_float),
    ]


class LpLayout(C.Structure):
    _fields_ = [
        ("Bp", C.c_int64), ("n_ld", C.c_int64), ("d_ld", C.c_int64), ("total_bytes", C.c_size_t),
        ("adj", C.c_size_t * 2), ("pos", C.c_size_t * 2), ("neg", C.c_size_t * 2), ("lse", C.c_size_t * 2),
        ("rowloss", C.c_size_t * 2), ("loss", C.c_size_t), ("dadj", C.c_size_t * 2), ("gocc", C.c_size_t),
        ("grel", C.c_size_t * 2), ("aux", C.c_size_t), ("lsepart", C.c_size_t), 
        ("dpos", C.c_size_t * 2), ("vlog", C.c_size_t), ("adjrec", C.c_size_t), ("negrec", C.c_size_t), ("fpart", C.c_size_t),
        ("flash", C.c_int32), ("flash_cfg", C.c_int32),
    ]


_i32, _i64, _f32, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class MapJob(C.Structure):  # marius_map_job
    _fields_ = [("ids_in", _vp), ("ids_out", _vp), ("edges", _vp), ("src_neg", _vp), ("dst_neg", _vp), ("B", _i64), ("CN", _i64), ("n", _i64),
                ("edge_cols", _i32), ("col", _i32), ("key_bits", _i32), ("reserved_", _i32), ("uniq", _vp), ("inverse", _vp), ("perm", _vp),
                ("seg_offsets", _vp), ("num_unique_dev", _vp), ("plan", _vp), ("edges_out", _vp), ("workspace", _vp), ("workspace_bytes", _sz)]


# name -> (restype, argtypes); mirrors include/marius_hip.h one to one
SIGNATURES = {
    "marius_hip_abi_version": (C.c_int, []),
    "marius_config_reload": (C.c_int, []),
    "marius_hip_struct_bytes": (C.c_int, [C.c_int]),
    "marius_hip_last_error": (C.c_char_p, []),
    "marius_gather_rows": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp]),
    "marius_gather_rows_counted": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp]),
    "marius_gather_rows2": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp]),
    "marius_scatter_add_rows": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp]),
    "marius_adagrad_rule": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _vp]),
    "marius_dense_adagrad_step": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _vp]),
    "marius_dense_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _vp]),
    "marius_mt19937_seed_host": (None, [_vp, C.c_uint64]),
    "marius_mt19937_fill_host": (None, [_vp, _vp, _i64]),
    "marius_mt19937_randperm_host": (C.c_int, [_vp, _vp, _i64]),
    "marius_mt19937_fill": (C.c_int, [_vp, _vp, _i64, _vp]),
    "marius_negatives_raw_words": (_i64, [_i64, _i64, _i32, _i32, _i32]),
    "marius_sample_negatives": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "marius_deg_filter": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp]),
    "marius_true_edge_filter_offsets": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp]),
    "marius_true_edge_filter_emit": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp]),
    "marius_select_edges": (C.c_int, [_vp, _i32, _i32, _vp, _i64, _i64, _vp, _vp]),
    "marius_assemble_ids": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp]),
    "marius_remap_edges": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "marius_debug_set_timeline": (C.c_int, [_vp]),
    "marius_profile_enable": (C.c_int, [C.c_int]),
    "marius_profile_reset": (C.c_int, []),
    "marius_profile_kernel_count": (C.c_int, []),
    "marius_profile_kernel_name": (C.c_char_p, [C.c_int]),
    "marius_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "marius_sort_unique_workspace_bytes": (_sz, [_i64]),
    "marius_sort_unique": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "marius_merge_unique_runs": (C.c_int, [_vp, _i64, C.POINTER(C.c_int64), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "marius_owner_offsets": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "marius_a2a_capacity": (_i64, [_i64, _i32, C.c_double]),
    "marius_a2a_rows_post": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "marius_a2a_rows_wait": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "marius_layer_post_hook": (C.c_int, [_vp, _i64, _vp, _i32, _i64, _i32, _vp, _i64, _vp]),
    "marius_layer_post_hook_workspace_bytes": (_sz, [_i64, _i32]),
    "marius_layer_post_hook_backward": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i64, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "marius_prepare_maps_supported": (C.c_int, [C.POINTER(MapJob), _i32]),
    "marius_prepare_maps_preferred": (C.c_int, []),
    "marius_nbr_workspace_bytes": (_sz, [_i64]),
    "marius_nbr_degrees": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "marius_nbr_gather": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "marius_nbr_dropout_offsets": (C.c_int, [_vp, _i64, _i64, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "marius_nbr_dropout_emit": (C.c_int, [_vp, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "marius_nbr_delta_ids": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "marius_nbr_positions": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i64, _vp, _vp, _vp]),
    "marius_segment_gather_sum": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp]),
    "marius_prepare_maps": (C.c_int, [C.POINTER(MapJob), _i32, _vp]),
    "marius_owner_offsets_counts": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "marius_a2a_record_words": (_i32, [_i32]),
    "marius_a2a_record_checksum": (C.c_uint64, [_vp, _i32]),
    "marius_a2a_publish": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp]),
    "marius_lp_plan": (C.c_int, [C.POINTER(LpDesc), C.POINTER(LpLayout)]),
    "marius_lp_forward": (C.c_int, [C.POINTER(LpDesc), C.POINTER(LpLayout), _vp, _vp]),
    "marius_lp_loss": (C.c_int, [C.POINTER(LpDesc), C.POINTER(LpLayout), _vp, _vp]),
    "marius_lp_backward": (C.c_int, [C.POINTER(LpDesc), C.POINTER(LpLayout), _vp, _vp]),
    "marius_softmax_ce": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _vp]),
    "marius_loss_scores": (C.c_int, [_i32, _f32, _vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp]),
    "marius_compute_ranks": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    "marius_segment_carry_bytes": (_sz, [_i64, _i32]),
    "marius_segment_sum_rows": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp]),
    "marius_segment_adagrad_scatter": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp]),
    "marius_segment_sum_rows_planned": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "marius_segment_plan_bytes": (C.c_size_t, [_i64]),
    "marius_segment_plan_occ_single": (C.c_void_p, [_vp, _i64]),
    "marius_lp_fuses_endpoint_update": (C.c_int, [_vp]),
    "marius_segment_plan": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "marius_segment_adagrad_scatter_planned": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp, _vp]),
    "marius_segment_adagrad_scatter_tracked": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp, _vp, _vp]),
    "marius_segment_adagrad_scatter_group": (C.c_int, [_vp, _i32, _vp]),
    "marius_table_absmax": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp]),
    "marius_table_absmax_counted": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp]),
}

_lib = None


def lib():
    """Load libmarius_hip.so; raise (never fall back) if it is absent."""
    global _lib
    if _lib is None:
        path = os.environ.get("MARIUS_HIP_LIB", LIB_PATH)  # override: experiment builds of the same C-ABI (tools/ablate_*.sh)
        if not os.path.exists(path):
            raise MariusHipError(
                "libmarius_hip.so not found at %s — build it with `python -m marius_amd.build`; there is no CPU fallback" % path)
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        # include/marius_hip.h MARIUS_HIP_ABI_VERSION: a stale library (or stale struct mirrors here) would read fields past the caller's struct
        got = L.marius_hip_abi_version()
        if got != ABI_VERSION:
            raise MariusHipError("libmarius_hip.so at %s has ABI version %d, this package was written against %d: rebuild with `python -m marius_amd.build`" % (path, got, ABI_VERSION))
        for which, mirror in ((0, LpDesc), (1, LpLayout), (2, SegmentUpdate)):
            if L.marius_hip_struct_bytes(which) != C.sizeof(mirror):
                raise MariusHipError("%s is %d bytes in libmarius_hip.so and %d in marius_amd/hip.py" % (mirror.__name__, L.marius_hip_struct_bytes(which), C.sizeof(mirror)))
        _lib = L
    return _lib


def reload_env():
    """re-read the MARIUS_* switches of the kernel library (it reads them once, at load): for tests that change one inside the process"""
    lib().marius_config_reload()


def check(rc, what=""):
    if rc != 0:
        msg = lib().marius_hip_last_error().decode()
        raise MariusHipError("%s failed (code %d): %s" % (what, rc, msg))


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _dev(t):
    if not t.is_cuda:
        raise MariusHipError("device tensor required (no CPU fallback)")
    return t


# ------------------------------------------------------------------------------------------------ thin wrappers
def gather_rows(table, ids, out=None):
    _dev(table)
    n, d = ids.numel(), table.size(1)
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=table.device)
    check(lib().marius_gather_rows(ptr(table), table.stride(0), ptr(ids), n, d, ptr(out), out.stride(0), stream_ptr()), "gather_rows")
    return out


def gather_rows_counted(table, ids, count_dev, out):
    """capacity-sized id list, valid length on the device: rows past the count are not touched"""
    _dev(table)
    check(lib().marius_gather_rows_counted(ptr(table), table.stride(0), ptr(ids), ids.numel(), ptr(count_dev), table.size(1), ptr(out), out.stride(0),
                                           stream_ptr()), "gather_rows_counted")
    return out


def gather_rows2(table_a, table_b, ids, out_a=None, out_b=None):
    _dev(table_a)
    n, d = ids.numel(), table_a.size(1)
    if out_a is None:
        out_a = torch.empty((n, d), dtype=torch.float32, device=table_a.device)
        out_b = torch.empty((n, d), dtype=torch.float32, device=table_a.device)
    assert table_a.stride(0) == table_b.stride(0) and out_a.stride(0) == out_b.stride(0)
    check(lib().marius_gather_rows2(ptr(table_a), ptr(table_b), table_a.stride(0), ptr(ids), n, d, ptr(out_a), ptr(out_b),
                                    out_a.stride(0), stream_ptr()), "gather_rows2")
    return out_a, out_b


def scatter_add_rows(table, ids, delta):
    _dev(table)
    check(lib().marius_scatter_add_rows(ptr(table), table.stride(0), ptr(ids), ids.numel(), table.size(1), ptr(delta), delta.stride(0),
                                        stream_ptr()), "scatter_add_rows")


def adagrad_rule(grad, state, lr, eps=1e-10):
    _dev(grad)
    dw, ds = torch.empty_like(grad), torch.empty_like(grad)
    check(lib().marius_adagrad_rule(ptr(grad), ptr(state), ptr(dw), ptr(ds), grad.numel(), lr, eps, stream_ptr()), "adagrad_rule")
    return dw, ds


def dense_adagrad_step(param, state_sum, grad, lr, eps=1e-10, weight_decay=0.0):
    _dev(param)
    check(lib().marius_dense_adagrad_step(ptr(param), ptr(state_sum), ptr(grad), param.numel(), lr, eps, weight_decay, stream_ptr()),
          "dense_adagrad_step")


def dense_adam_step(param, exp_avg, exp_avg_sq, grad, lr, num_steps, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, max_exp_avg_sq=None):
    _dev(param)
    check(lib().marius_dense_adam_step(ptr(param), ptr(exp_avg), ptr(exp_avg_sq), ptr(max_exp_avg_sq), ptr(grad), param.numel(), lr, beta1, beta2,
                                       eps, weight_decay, num_steps, stream_ptr()), "dense_adam_step")


class Generator:
    """ATen-CPU-compatible MT19937 stream that can live on the host or the device."""

    def __init__(self, seed, device=None):
        self.host = torch.zeros(MT_STATE_WORDS, dtype=torch.int32)
        lib().marius_mt19937_seed_host(ptr(self.host), seed)
        self.dev = None
        self.device = device

    def to_device(self, device=None):
        self.device = device or self.device
        self.dev = self.host.to(self.device)
        return self

    def to_host(self):
        if self.dev is not None:
            self.host = self.dev.cpu()
            self.dev = None
        return self

    def randperm_host(self, n):
        self.to_host()
        out = torch.empty(n, dtype=torch.int64)
        check(lib().marius_mt19937_randperm_host(ptr(self.host), ptr(out), n), "randperm")
        return out

    def fill_host(self, n):
        self.to_host()
        out = torch.empty(n, dtype=torch.int32)
        lib().marius_mt19937_fill_host(ptr(self.host), ptr(out), n)
        return out

    def fill_device(self, n, out=None):
        if self.dev is None:
            self.to_device()
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.dev.device)
        check(lib().marius_mt19937_fill(ptr(self.dev), ptr(out), n, stream_ptr()), "mt19937_fill")
        return out


def negatives_raw_words(num_nodes, B, C_, N, n_deg):
    return lib().marius_negatives_raw_words(num_nodes, B, C_, N, n_deg)


def sample_negatives(raw, edges, num_nodes, num_chunks, num_negatives, degree_fraction, inverse):
    """CorruptNodeNegativeSampler::getNegatives on the device. Returns (ids [C,N], deg_pos [C,n_deg] or None)."""
    _dev(raw)
    n_deg = int(num_negatives * degree_fraction)
    out = torch.empty((num_chunks, num_negatives), dtype=torch.int64, device=raw.device)
    deg = torch.empty((num_chunks, n_deg), dtype=torch.int64, device=raw.device) if n_deg > 0 else None
    B = edges.size(0)
    check(lib().marius_sample_negatives(ptr(raw), ptr(edges), B, edges.size(1), 1 if inverse else 0, num_nodes, num_chunks, num_negatives,
                                        n_deg, ptr(out), ptr(deg), stream_ptr()), "sample_negatives")
    return out, deg


def deg_filter(deg_pos, B):
    """Uncompacted DEG local filter [C*n_deg, 2]; rows of -1 are non-hits."""
    C_, n_deg = deg_pos.shape
    out = torch.empty((C_ * n_deg, 2), dtype=torch.int64, device=deg_pos.device)
    check(lib().marius_deg_filter(ptr(deg_pos), C_, n_deg, B, ptr(out), stream_ptr()), "deg_filter")
    return out


def select_edges(edges_all, perm, start, B):
    _dev(edges_all)
    out = torch.empty((B, edges_all.size(1)), dtype=torch.int64, device=edges_all.device)
    check(lib().marius_select_edges(ptr(edges_all), 1 if edges_all.dtype == torch.int64 else 0, edges_all.size(1), ptr(perm), start, B,
                                    ptr(out), stream_ptr()), "select_edges")
    return out


class UniqueMap:
    """Device buffers + workspace for marius_sort_unique over up to `capacity` ids."""

    def __init__(self, capacity, device):
        self.cap = capacity
        self.uniq = torch.empty(capacity, dtype=torch.int64, device=device)
        self.inverse = torch.empty(capacity, dtype=torch.int64, device=device)
        self.perm = torch.empty(capacity, dtype=torch.int32, device=device)
        self.seg = torch.empty(capacity + 1, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int64, device=device)
        self.ws_bytes = lib().marius_sort_unique_workspace_bytes(capacity)
        if self.ws_bytes == 0:
            raise MariusHipError("sort_unique workspace query failed")
        self.ws = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=device)  # zeroed once: the sort's control block (marius_hip.h)

    def run(self, ids, key_bits=63):
        n = ids.numel()
        assert n <= self.cap
        check(lib().marius_sort_unique(ptr(ids), n, key_bits, ptr(self.uniq), ptr(self.inverse), ptr(self.perm), ptr(self.seg),
                                       ptr(self.count), ptr(self.ws), self.ws_bytes, stream_ptr()), "sort_unique")
        return self


def map_job(um, key_bits, ids=None, edges=None, src_neg=None, dst_neg=None, col=-1, ids_out=None, plan=None, edges_out=None):
    """one marius_map_job writing into the UniqueMap `um`: ids given, or assembled from the batch's edges (+ negatives) by the launch"""
    j = MapJob()
    n = ids.numel() if ids is not None else (edges.size(0) if col >= 0 else 2 * edges.size(0) + sum(t.numel() for t in (src_neg, dst_neg) if t is not None))
    assert n <= um.cap
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    j.ids_in, j.ids_out, j.edges, j.src_neg, j.dst_neg = p(ids), p(ids_out), p(edges), p(src_neg), p(dst_neg)
    j.B = 0 if edges is None else edges.size(0)
    j.CN = max([t.numel() for t in (src_neg, dst_neg) if t is not None] + [0])
    j.n, j.edge_cols, j.col, j.key_bits = n, (0 if edges is None else edges.size(1)), col, key_bits
    j.uniq, j.inverse, j.perm, j.seg_offsets, j.num_unique_dev = p(um.uniq), p(um.inverse), p(um.perm), p(um.seg), p(um.count)
    j.plan, j.edges_out, j.workspace, j.workspace_bytes = p(plan), p(edges_out), p(um.ws), um.ws_bytes
    return j


def prepare_maps(jobs):
    """marius_prepare_maps: the whole map chain of up to two id lists in one persistent launch; False when a job is outside its range"""
    arr = (MapJob * len(jobs))(*jobs)
    if not lib().marius_prepare_maps_supported(arr, len(jobs)):
        return False
    check(lib().marius_prepare_maps(arr, len(jobs), stream_ptr()), "prepare_maps")
    return True


def owner_offsets(um, shard_rows, num_shards):
    """Split points of um.uniq[:U] by owning shard; returned on the HOST (one small D2H sync: all-to-all needs host sizes)."""
    out = torch.empty(num_shards + 1, dtype=torch.int64, device=um.uniq.device)
    check(lib().marius_owner_offsets(ptr(um.uniq), ptr(um.count), shard_rows, num_shards, ptr(out), stream_ptr()), "owner_offsets")
    return out.cpu()


def a2a_capacity(max_rows, world, slack=1.5):
    return int(lib().marius_a2a_capacity(max_rows, world, slack))


def a2a_rows_post(um, offs_dev, shard_rows, world, cap, inverse=None):
    """requester half of the fixed-capacity exchange (marius_a2a_rows_post): returns (req_send [world * cap], place [capacity of um],
    overflow flag int32[1], slot_of_occ [len(inverse)] or None)"""
    dev = um.uniq.device
    req = torch.empty(world * cap, dtype=torch.int64, device=dev)
    place = torch.full((um.cap,), -7, dtype=torch.int64, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    slot = None if inverse is None else torch.empty(inverse.numel(), dtype=torch.int64, device=dev)
    check(lib().marius_a2a_rows_post(ptr(um.uniq), ptr(offs_dev), shard_rows, world, cap, ptr(req), ptr(place), ptr(flag), ptr(inverse),
                                     0 if inverse is None else inverse.numel(), ptr(slot), stream_ptr()), "a2a_rows_post")
    return req, place, flag, slot


ACT = {"NONE": 0, "RELU": 1, "SIGMOID": 2}


def layer_post_hook(x, bias=None, activation="NONE", out=None):
    """Layer::post_hook (layer.cpp:9-16): act(x + bias) over the rows of x [n, d]"""
    _dev(x)
    out = torch.empty_like(x) if out is None else out
    check(lib().marius_layer_post_hook(ptr(x), x.stride(0), ptr(bias), ACT[activation], x.size(0), x.size(1), ptr(out), out.stride(0), stream_ptr()), "layer_post_hook")
    return out


def layer_post_hook_backward(gy, y, activation="NONE", with_bias=True):
    """(gx [n, d], bias_grad [d] or None) of layer_post_hook from the gradient of its output and the output itself"""
    _dev(gy)
    n, d = gy.shape
    gx = torch.empty_like(gy)
    bg = torch.empty(d, dtype=torch.float32, device=gy.device) if with_bias else None
    wsb = lib().marius_layer_post_hook_workspace_bytes(n, d)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=gy.device)
    check(lib().marius_layer_post_hook_backward(ptr(gy), gy.stride(0), ptr(y), 0 if y is None else y.stride(0), ACT[activation], n, d, ptr(gx), gx.stride(0), ptr(bg),
                                                ptr(ws), wsb, stream_ptr()), "layer_post_hook_backward")
    return gx, bg


def true_edge_filter(sorted_edges, edges, inverse):
    """compute_filter_corruption, global filter (src/data/samplers/negative.cpp:50-293): [F, 2] = (batch edge id, node whose score is masked) for every known
    edge that shares a batch edge's uncorrupted endpoint and relation.  sorted_edges: all known edges sorted by source (inverse False) / destination (True).
    One host read (F) between the two launches, as the reference sizes its output."""
    _dev(edges)
    B, cols = edges.shape
    counts = torch.empty(B, dtype=torch.int64, device=edges.device)
    offsets = torch.empty(B + 1, dtype=torch.int64, device=edges.device)
    check(lib().marius_true_edge_filter_offsets(ptr(sorted_edges), sorted_edges.size(0), cols, 1 if inverse else 0, ptr(edges), B, ptr(counts), ptr(offsets), stream_ptr()),
          "true_edge_filter_offsets")
    F = int(offsets[B].item())
    out = torch.empty((F, 2), dtype=torch.int64, device=edges.device)
    check(lib().marius_true_edge_filter_emit(ptr(sorted_edges), sorted_edges.size(0), cols, 1 if inverse else 0, ptr(edges), B, ptr(offsets), ptr(out), stream_ptr()),
          "true_edge_filter_emit")
    return out


def owner_offsets_counts(um, shard_rows, world):
    """marius_owner_offsets_counts: (offs [world + 1], send counts [world]) of a sorted unique map, one launch"""
    offs = torch.empty(world + 1, dtype=torch.int64, device=um.uniq.device)
    cnt = torch.empty(world, dtype=torch.int64, device=um.uniq.device)
    check(lib().marius_owner_offsets_counts(ptr(um.uniq), ptr(um.count), shard_rows, world, ptr(offs), ptr(cnt), stream_ptr()), "owner_offsets_counts")
    return offs, cnt


def a2a_publish(offs_dev, record_pinned, stamp, recv_counts=None, overflow=None):
    """marius_a2a_publish: the batch's exchange header -> ONE ordered record in pinned host memory (record_pinned: int64 [2 world + 4], pinned)"""
    world = offs_dev.numel() - 1
    assert record_pinned.is_pinned() and record_pinned.dtype == torch.int64 and record_pinned.numel() >= lib().marius_a2a_record_words(world)
    check(lib().marius_a2a_publish(ptr(offs_dev), ptr(recv_counts), ptr(overflow), world, int(stamp), record_pinned.data_ptr(), stream_ptr()), "a2a_publish")


def a2a_record_ok(record_host, world):
    """host check of a published record: checksum word == marius_a2a_record_checksum of the words before it"""
    r = record_host.contiguous()
    want = lib().marius_a2a_record_checksum(r.data_ptr(), world)
    return (int(r[2 * world + 3]) & 0xFFFFFFFFFFFFFFFF) == want


def a2a_rows_wait(rows_recv, absmax=None, place=None, count_dev=None, emb=None):
    """marius_a2a_rows_wait: bound of the received payload (absmax, device float[1], max'ed in) and — with emb — its copy in batch order"""
    check(lib().marius_a2a_rows_wait(ptr(rows_recv), rows_recv.stride(0), rows_recv.size(0), rows_recv.size(1), ptr(absmax), ptr(place), ptr(count_dev),
                                     0 if emb is None else emb.size(0), ptr(emb), 0 if emb is None else emb.stride(0), stream_ptr()), "a2a_rows_wait")
    return emb


class LpWorkspace:
    """marius_lp_desc + layout + workspace for one batch shape."""

    def __init__(self, relop, cmp, d, B, C_, N, use_inverse, reduction, edge_cols, has_src_neg, device, loss=0, margin=0.0, flags=0):
        self.desc = LpDesc()
        # this class serves tests and tools, which read dadj() back: ask the flash path to leave dL/dadj there (the C++ trainer does not)
        self.desc.loss, self.desc.margin, self.desc.flags = loss, margin, (flags | LP_KEEP_DADJ) if (flags & LP_TRAIN_ONLY) else flags
        self.desc.relop, self.desc.cmp, self.desc.d, self.desc.edge_cols = relop, cmp, d, edge_cols
        self.desc.B, self.desc.C, self.desc.N = B, C_, N
        self.desc.use_inverse, self.desc.reduction = int(use_inverse), reduction
        # plan needs src_neg / inv_rel presence for validation: use dummy non-null markers
        self.desc.src_neg = C.c_void_p(1) if has_src_neg else None
        self.desc.inv_rel = C.c_void_p(1) if use_inverse else None
        self.layout = LpLayout()
        check(lib().marius_lp_plan(C.byref(self.desc), C.byref(self.layout)), "lp_plan")
        self.ws = torch.empty(self.layout.total_bytes, dtype=torch.uint8, device=device)
        self.device = device
        self._keep = None

    def bind(self, emb, edges, dst_neg, src_neg, rel, inv_rel, dst_filter=None, src_filter=None, absmax=None, absmax_rel=None):
        """absmax: optional device float[2] (bound on |node rows|, bound on |relation rows|): fp16 operand records on the flash path;
        absmax_rel: optional device float[1], the relation bound kept apart from the node bound (absmax is then float[1])"""
        d = self.desc
        d.absmax = absmax.data_ptr() if absmax is not None else None
        d.absmax_rel = absmax_rel.data_ptr() if absmax_rel is not None else None
        self._absmax = (absmax, absmax_rel)
        d.emb, d.emb_ld, d.U = emb.data_ptr(), emb.stride(0), emb.size(0)
        d.edges, d.dst_neg = edges.data_ptr(), dst_neg.data_ptr()
        d.src_neg = src_neg.data_ptr() if src_neg is not None else None
        d.rel = rel.data_ptr() if rel is not None else None
        d.inv_rel = inv_rel.data_ptr() if inv_rel is not None else None
        d.rel_ld = rel.stride(0) if rel is not None else 0
        d.R = rel.size(0) if rel is not None else 0
        d.dst_filter = dst_filter.data_ptr() if dst_filter is not None and dst_filter.numel() else None
        d.n_dst_filter = dst_filter.size(0) if dst_filter is not None else 0
        d.src_filter = src_filter.data_ptr() if src_filter is not None and src_filter.numel() else None
        d.n_src_filter = src_filter.size(0) if src_filter is not None else 0
        self._keep = (emb, edges, dst_neg, src_neg, rel, inv_rel, dst_filter, src_filter)
        return self

    def fuse_endpoint_update(self, occ_single=None, state=None, lr=0.0, eps=1e-10, absmax=None):
        """marius_lp_desc.upd_*: the edge backward takes the Adagrad step of endpoint occurrences whose node occurs once (emb must be the
        table itself); returns whether this library build honours it for the bound shapes.  No arguments: off.
        Checked here because the C-ABI cannot: the bound `emb` must be the mutable table whose row pitch the state shares, and have
        at least as many rows.  The gocc rows of flagged endpoint occurrences are undefined after backward()."""
        d = self.desc
        if state is not None:
            emb = self._keep[0] if getattr(self, "_keep", None) else None
            if emb is None or state.dim() != 2 or state.stride(0) != d.emb_ld or state.size(0) != emb.size(0) or state.size(1) != emb.size(1) or state.data_ptr() == emb.data_ptr():
                raise MariusHipError("fuse_endpoint_update: bind() the node TABLE first; the state must be a distinct tensor of the table's shape and row pitch")
        d.upd_occ_single = occ_single.data_ptr() if occ_single is not None else None
        d.upd_state = state.data_ptr() if state is not None else None
        d.upd_absmax = absmax.data_ptr() if absmax is not None else None
        d.upd_lr, d.upd_eps = lr, eps
        self._upd = (occ_single, state, absmax)
        return bool(lib().marius_lp_fuses_endpoint_update(C.byref(d)))

    def forward(self):
        check(lib().marius_lp_forward(C.byref(self.desc), C.byref(self.layout), ptr(self.ws), stream_ptr()), "lp_forward")

    def loss(self):
        check(lib().marius_lp_loss(C.byref(self.desc), C.byref(self.layout), ptr(self.ws), stream_ptr()), "lp_loss")

    def backward(self):
        check(lib().marius_lp_backward(C.byref(self.desc), C.byref(self.layout), ptr(self.ws), stream_ptr()), "lp_backward")

    # typed views into the workspace
    def _view(self, off, shape):
        n = 1
        for s in shape:
            n *= s
        return self.ws[off:off + 4 * n].view(torch.float32).view(*shape)

    def pos(self, dir_):
        return self._view(self.layout.pos[dir_], (self.layout.Bp,))

    def neg(self, dir_):
        d = self.desc
        if self.layout.flash and d.d > 128 and not (d.flags & LP_STORE_SCORES):
            # rows wider than 128 columns: the flash path keeps its stored scores in tile order [chunk-direction][adj tile][negative block][128][32]
            Bc = -(-d.B // d.C)
            XT, YB = -(-Bc // 128), -(-d.N // 32)
            per_dir = d.C * XT * YB * 128 * 32
            t = self._view(self.layout.neg[0] + 4 * per_dir * dir_, (d.C, XT, YB, 128, 32))
            return t.permute(0, 1, 3, 2, 4).reshape(d.C, XT * 128, YB * 32)[:, :Bc, : d.N].reshape(d.C * Bc, d.N)
        return self._view(self.layout.neg[dir_], (self.layout.Bp, self.layout.n_ld))[:, : self.desc.N]

    def lse(self, dir_):
        return self._view(self.layout.lse[dir_], (self.layout.Bp,))

    def rowloss(self, dir_):
        return self._view(self.layout.rowloss[dir_], (self.layout.Bp,))

    def loss_values(self):
        return self._view(self.layout.loss, (4,))

    def adj(self, dir_):
        return self._view(self.layout.adj[dir_], (self.layout.Bp, self.layout.d_ld))[:, : self.desc.d]

    def dadj(self, dir_):
        return self._view(self.layout.dadj[dir_], (self.layout.Bp, self.layout.d_ld))[:, : self.desc.d]

    def num_occ(self):
        return 2 * self.desc.B + (2 if self.desc.src_neg else 1) * self.desc.C * self.desc.N

    def gocc(self):
        return self._view(self.layout.gocc, (self.num_occ(), self.layout.d_ld))

    def grel(self, dir_):
        return self._view(self.layout.grel[dir_], (self.desc.B, self.layout.d_ld))


def compute_ranks(pos, neg):
    _dev(pos)
    ranks = torch.empty(pos.numel(), dtype=torch.int64, device=pos.device)
    check(lib().marius_compute_ranks(ptr(pos), ptr(neg), pos.numel(), neg.size(1), neg.stride(0), ptr(ranks), stream_ptr()), "compute_ranks")
    return ranks


def segment_carry(n, d, device):
    return torch.empty(lib().marius_segment_carry_bytes(n, d), dtype=torch.uint8, device=device)


def segment_sum_rows(rows, um, n, d, out, out_rows=None, carry=None, plan=None):
    _dev(rows)
    if carry is None:
        carry = segment_carry(n, d, rows.device)
    if plan is not None:
        check(lib().marius_segment_sum_rows_planned(ptr(rows), rows.stride(0), ptr(um.perm), ptr(um.inverse), ptr(um.seg), n, d, ptr(out_rows), ptr(out),
                                                    out.stride(0), ptr(carry), ptr(plan), stream_ptr()), "segment_sum_rows_planned")
        return out
    check(lib().marius_segment_sum_rows(ptr(rows), rows.stride(0), ptr(um.perm), ptr(um.inverse), ptr(um.seg), n, d, ptr(out_rows), ptr(out),
                                        out.stride(0), ptr(carry), stream_ptr()), "segment_sum_rows")
    return out


def segment_plan(um, n):
    """marius_segment_plan of a unique map: the index work of the fused update, precomputable as soon as the ids are sorted."""
    plan = torch.empty(int(lib().marius_segment_plan_bytes(n)), dtype=torch.uint8, device=um.perm.device)
    check(lib().marius_segment_plan(ptr(um.perm), ptr(um.inverse), ptr(um.seg), ptr(um.uniq), n, ptr(plan), stream_ptr()), "segment_plan")
    return plan


def segment_plan_occ_single(plan, n):
    """uint8[n] view of the per-occurrence singleton flags inside a plan (marius_lp_desc.upd_occ_single)"""
    p = lib().marius_segment_plan_occ_single(ptr(plan), n)
    off = p - plan.data_ptr()
    return plan[off:off + n]


def table_absmax(*tables, out=None, count=None):
    """device float[1]: max |x| over the given [rows, d] tables (marius_table_absmax), max'ed into `out` when given; count: device int64
    holding the number of valid rows of a capacity-sized buffer (marius_table_absmax_counted)"""
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=tables[0].device)
    for t in tables:
        if count is not None:
            check(lib().marius_table_absmax_counted(ptr(t), t.size(0), ptr(count), t.stride(0), t.size(1), ptr(out), stream_ptr()), "table_absmax_counted")
        else:
            check(lib().marius_table_absmax(ptr(t), t.size(0), t.stride(0), t.size(1), ptr(out), stream_ptr()), "table_absmax")
    return out


def segment_adagrad_scatter(rows, um, n, d, table, state, lr, eps=1e-10, carry=None, plan=None, absmax=None):
    _dev(rows)
    if carry is None:
        carry = segment_carry(n, d, rows.device)
    if absmax is not None:
        check(lib().marius_segment_adagrad_scatter_tracked(ptr(rows), rows.stride(0), ptr(um.perm), ptr(um.inverse), ptr(um.seg), n, d, ptr(um.uniq),
                                                           ptr(table), ptr(state), table.stride(0), lr, eps, ptr(carry), ptr(plan), ptr(absmax), stream_ptr()),
              "segment_adagrad_scatter_tracked")
        return
    if plan is not None:
        check(lib().marius_segment_adagrad_scatter_planned(ptr(rows), rows.stride(0), ptr(um.perm), ptr(um.inverse), ptr(um.seg), n, d, ptr(um.uniq),
                                                           ptr(table), ptr(state), table.stride(0), lr, eps, ptr(carry), ptr(plan), stream_ptr()),
              "segment_adagrad_scatter_planned")
        return
    check(lib().marius_segment_adagrad_scatter(ptr(rows), rows.stride(0), ptr(um.perm), ptr(um.inverse), ptr(um.seg), n, d, ptr(um.uniq),
                                               ptr(table), ptr(state), table.stride(0), lr, eps, ptr(carry), stream_ptr()),
          "segment_adagrad_scatter")


class SegmentUpdate(C.Structure):
    """marius_segment_update (include/marius_hip.h): one table's job of marius_segment_adagrad_scatter_group"""
    _fields_ = [("rows", C.c_void_p), ("rows_ld", C.c_int64), ("perm", C.c_void_p), ("inverse", C.c_void_p), ("seg_offsets", C.c_void_p), ("n", C.c_int64),
                ("d", C.c_int32), ("uniq_ids", C.c_void_p), ("table", C.c_void_p), ("state", C.c_void_p), ("table_ld", C.c_int64), ("lr", C.c_float),
                ("eps", C.c_float), ("carry", C.c_void_p), ("plan", C.c_void_p), ("absmax", C.c_void_p), ("fused_below", C.c_int64),
                ("sum_out", C.c_void_p), ("sum_out_ld", C.c_int64), ("sum_out_rows", C.c_void_p)]


def segment_adagrad_scatter_group(jobs):
    """jobs: list of dicts with the keyword arguments of segment_adagrad_scatter (rows, um, n, d, table, state, lr [, eps, carry, plan, absmax]),
    or reduce-only jobs (rows, um, n, d, plan, sum_out [, sum_out_rows, carry]); all of them in one pair of launches"""
    arr = (SegmentUpdate * len(jobs))()
    keep = []
    for a, j in zip(arr, jobs):
        rows, um, n, d = j["rows"], j["um"], j["n"], j["d"]
        _dev(rows)
        carry = j.get("carry")
        if carry is None:
            carry = segment_carry(n, d, rows.device)
        keep.append(carry)
        a.rows, a.rows_ld, a.perm, a.inverse, a.seg_offsets, a.n, a.d = ptr(rows), rows.stride(0), ptr(um.perm), ptr(um.inverse), ptr(um.seg), n, d
        a.carry, a.plan = ptr(carry), ptr(j.get("plan"))
        if j.get("sum_out") is not None:  # a reduce-only job (marius_segment_update.sum_out)
            out = j["sum_out"]
            a.sum_out, a.sum_out_ld, a.sum_out_rows = ptr(out), out.stride(0), ptr(j.get("sum_out_rows"))
            continue
        a.uniq_ids, a.table, a.state, a.table_ld = ptr(um.uniq), ptr(j["table"]), ptr(j["state"]), j["table"].stride(0)
        a.lr, a.eps, a.absmax = j["lr"], j.get("eps", 1e-10), ptr(j.get("absmax"))
        a.fused_below = int(j.get("fused_below", 0))
    check(lib().marius_segment_adagrad_scatter_group(C.cast(arr, C.c_void_p), len(jobs), stream_ptr()), "segment_adagrad_scatter_group")


def profile_enable(on=True, only=None):
    """on: record HIP events around every instrumented kernel; only="name": around that kernel alone (cheap enough for a timed region)."""
    mode = 1 if on else 0
    if on and only is not None:
        names = [lib().marius_profile_kernel_name(i).decode() for i in range(lib().marius_profile_kernel_count())]
        mode = 2 + names.index(only)
    lib().marius_profile_enable(mode)


def profile_reset():
    lib().marius_profile_reset()


def profile_read():
    """{kernel name: (total_ms, launches)} measured with HIP events on the launch stream."""
    out = {}
    for i in range(lib().marius_profile_kernel_count()):
        ms, cnt = C.c_double(0), C.c_int64(0)
        check(lib().marius_profile_read(i, C.byref(ms), C.byref(cnt)), "profile_read")
        out[lib().marius_profile_kernel_name(i).decode()] = (ms.value, cnt.value)
    return out
