"""ORACLE (test infrastructure): ctypes wrapper around oracle/_build/libmt_oracle.so (oracle/mt19937_aten.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmt_oracle.so")


class _State(C.Structure):
    _fields_ = [("state", C.c_uint32 * 624), ("next", C.c_int32)]


def _lib():
    if not os.path.exists(_SO):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, os.path.join(_HERE, "mt19937_aten.c")])
    return C.CDLL(_SO)


class OracleGenerator:
    """at::mt19937 CPU generator stream (torch.manual_seed(seed))."""

    def __init__(self, seed):
        self.L = _lib()
        self.s = _State()
        self.L.mt_seed(C.byref(self.s), C.c_uint64(seed))

    def raw(self, n):
        out = np.zeros(n, dtype=np.uint32)
        self.L.mt_fill_raw(C.byref(self.s), out.ctypes.data_as(C.c_void_p), C.c_int64(n))
        return out

    def randint(self, high, n):
        out = np.zeros(n, dtype=np.int64)
        self.L.mt_randint(C.byref(self.s), out.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_uint64(high), C.c_int64(0))
        return out

    def randperm(self, n):
        out = np.zeros(n, dtype=np.int64)
        self.L.mt_randperm(C.byref(self.s), out.ctypes.data_as(C.c_void_p), C.c_int64(n))
        return out

    def get_negatives(self, edges, num_nodes, num_chunks, num_negatives, degree_fraction, inverse):
        """CorruptNodeNegativeSampler::getNegatives (negative.cpp:328-366). edges: int64 ndarray [B, 2|3]."""
        edges = np.ascontiguousarray(edges, dtype=np.int64)
        B, ncols = edges.shape
        n_deg = int(num_negatives * degree_fraction)
        ids = np.zeros((num_chunks, num_negatives), dtype=np.int64)
        deg = np.zeros((num_chunks, max(n_deg, 1)), dtype=np.int64)
        self.L.oracle_get_negatives(C.byref(self.s), edges.ctypes.data_as(C.c_void_p), C.c_int64(B), C.c_int(ncols), C.c_int(1 if inverse else 0),
                                    C.c_int64(num_nodes), C.c_int(num_chunks), C.c_int(num_negatives), C.c_float(degree_fraction),
                                    ids.ctypes.data_as(C.c_void_p), deg.ctypes.data_as(C.c_void_p))
        return ids, (deg[:, :n_deg] if n_deg > 0 else None)
