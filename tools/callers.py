"""ctypes face of tools/callers.c: T host threads, one item per call, closed loop, on one handle (bench.py `concurrent_callers`, tests/test_concurrent_gpu.py)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class Result(C.Structure):
    _fields_ = [("wall_s", C.c_double), ("p50_us", C.c_double), ("p99_us", C.c_double), ("mean_us", C.c_double), ("max_us", C.c_double),
                ("calls", C.c_uint64), ("mismatches", C.c_uint64), ("errors", C.c_uint64)]

    def as_dict(self, unit):
        return {unit + "_per_s": round(self.calls / self.wall_s, 1) if self.wall_s > 0 else 0.0, "calls": int(self.calls), "p50_us": round(self.p50_us, 1),
                "p99_us": round(self.p99_us, 1), "mean_us": round(self.mean_us, 1), "max_us": round(self.max_us, 1), "mismatches": int(self.mismatches), "errors": int(self.errors)}


def lib():
    global _lib
    if _lib is None:
        out = os.path.join(tempfile.mkdtemp(prefix="shodh_callers_"), "libshodh_callers.so")
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", os.path.join(HERE, "callers.c"), "-o", out])
        _lib = C.CDLL(out)
        _lib.callers_search.restype = C.c_int
        _lib.callers_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Result)]
        _lib.callers_encode.restype = C.c_int
        _lib.callers_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(Result)]
    return _lib


def _addr(fn):
    return C.cast(fn, C.c_void_p).value


def search(shodh_lib, handle, queries, k, threads, calls_per_thread, warmup=3, expect=None, sharded=False):
    """queries f32 [n][dim]; expect = (ids uint32 [n][k], dist f32 [n][k]) solo answers or None. Returns Result."""
    q = np.ascontiguousarray(queries, np.float32)
    e_ids = e_dist = None
    if expect is not None:
        e_ids = np.ascontiguousarray(expect[0], np.uint32); e_dist = np.ascontiguousarray(expect[1], np.float32)
        assert e_ids.shape == (q.shape[0], k) and e_dist.shape == (q.shape[0], k)
    r = Result()
    fn = shodh_lib.shodh_sharded_index_search if sharded else shodh_lib.shodh_index_search
    rc = lib().callers_search(_addr(fn), handle, q.ctypes.data, q.shape[0], q.shape[1], k, threads, calls_per_thread, warmup,
                              e_ids.ctypes.data if e_ids is not None else None, e_dist.ctypes.data if e_dist is not None else None, C.byref(r))
    assert rc == 0
    return r


def encode(shodh_lib, handle, ids, mask, hidden, threads, calls_per_thread, warmup=2, expect=None):
    ids = np.ascontiguousarray(ids, np.int32); mask = np.ascontiguousarray(mask, np.uint8)
    ex = None if expect is None else np.ascontiguousarray(expect, np.float32)
    r = Result()
    rc = lib().callers_encode(_addr(shodh_lib.shodh_embedder_encode_ids), handle, ids.ctypes.data, mask.ctypes.data, ids.shape[0], ids.shape[1], hidden, threads,
                              calls_per_thread, warmup, ex.ctypes.data if ex is not None else None, C.byref(r))
    assert rc == 0
    return r
