"""ctypes binding of the CPU oracle (oracle/_build/librjo.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys

sys.path.insert(0, os.path.join(ROOT, "radix-join_amd"))
from pyrj import pages as pg  # noqa: E402
from pyrj import plan as pl  # noqa: E402

_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "_build", "librjo.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.rjo_execute.argtypes = [C.POINTER(pl.rj_plan), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.rjo_execute.restype = C.c_int
        for f in ("rjo_result_num_rows", "rjo_result_num_cols"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_uint64
        L.rjo_result_col_type.argtypes = [C.c_void_p, C.c_uint64]
        L.rjo_result_col_type.restype = C.c_int32
        L.rjo_result_col_pages.argtypes = [C.c_void_p, C.c_uint64]
        L.rjo_result_col_pages.restype = C.c_uint64
        L.rjo_result_page.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.rjo_result_page.restype = C.c_void_p
        L.rjo_result_free.argtypes = [C.c_void_p]
        L.rjo_result_free.restype = None
        L.rjo_decode_fixed.argtypes = [C.POINTER(pl.rj_column), C.c_uint64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.rjo_decode_fixed.restype = C.c_int
        L.rjo_decode_varchar.argtypes = [C.POINTER(pl.rj_column), C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.rjo_decode_varchar.restype = C.c_int
        L.rjo_free.argtypes = [C.c_void_p]
        L.rjo_encode_fixed.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.rjo_encode_fixed.restype = C.c_int
        L.rjo_encode_varchar.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.rjo_encode_varchar.restype = C.c_int
        L.rjo_from_csv.restype = C.c_int
        L.rjo_hash_int.argtypes = [C.c_int64]
        L.rjo_hash_int.restype = C.c_uint64
        L.rjo_num_buckets.argtypes = [C.c_uint64, C.c_uint64]
        L.rjo_num_buckets.restype = C.c_uint64
        _LIB = L
    return _LIB


def _result_to_table(L, r) -> pl.ColumnarTable:
    t = pl.ColumnarTable()
    t.num_rows = L.rjo_result_num_rows(r)
    for c in range(L.rjo_result_num_cols(r)):
        n = L.rjo_result_col_pages(r, c)
        pages = np.zeros((n, pg.PAGE_SIZE), dtype=np.uint8)
        for i in range(n):
            C.memmove(pages[i].ctypes.data, L.rjo_result_page(r, c, i), pg.PAGE_SIZE)
        t.columns.append(pl.Column(L.rjo_result_col_type(r, c), pages))
    return t


def execute(plan: pl.Plan) -> pl.ColumnarTable:
    """The oracle's ``Contest::execute`` (reference src/execute.cpp:316-324)."""
    L = lib()
    cplan, keep = pl.plan_to_c(plan)
    out = C.c_void_p()
    err = C.create_string_buffer(256)
    rc = L.rjo_execute(C.byref(cplan), C.byref(out), err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    try:
        return _result_to_table(L, out)
    finally:
        L.rjo_result_free(out)
        del keep


def decode_fixed(col: pl.Column, num_rows: int):
    L = lib()
    keep = []
    t = pl.ColumnarTable(num_rows, [col])
    inp = pl.input_to_c(t, keep)
    vals = np.zeros(num_rows, dtype=pg.NP_DTYPE[col.type])
    valid = np.zeros(num_rows, dtype=np.uint8)
    err = C.create_string_buffer(256)
    rc = L.rjo_decode_fixed(inp.cols, num_rows, vals.ctypes.data, valid.ctypes.data, err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return vals, valid.astype(bool)


def decode_varchar(col: pl.Column, num_rows: int):
    L = lib()
    keep = []
    inp = pl.input_to_c(pl.ColumnarTable(num_rows, [col]), keep)
    offs = np.zeros(num_rows + 1, dtype=np.uint64)
    valid = np.zeros(num_rows, dtype=np.uint8)
    heap = C.c_void_p()
    err = C.create_string_buffer(256)
    rc = L.rjo_decode_varchar(inp.cols, num_rows, offs.ctypes.data, valid.ctypes.data, C.byref(heap), err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    total = int(offs[-1])
    buf = C.string_at(heap, total) if total else b""
    L.rjo_free(heap)
    return [buf[int(offs[i]) : int(offs[i + 1])] if valid[i] else None for i in range(num_rows)]


def encode_fixed(dtype: int, values, valid=None) -> pl.Column:
    L = lib()
    vals = np.ascontiguousarray(values, dtype=pg.NP_DTYPE[dtype])
    v8 = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
    out = C.c_void_p()
    rc = L.rjo_encode_fixed(dtype, vals.ctypes.data, None if v8 is None else v8.ctypes.data, vals.shape[0], C.byref(out))
    assert rc == 0
    try:
        return _result_to_table(L, out).columns[0]
    finally:
        L.rjo_result_free(out)


def encode_varchar(strings) -> pl.Column:
    L = lib()
    n = len(strings)
    valid = np.array([s is not None for s in strings], dtype=np.uint8)
    heap = b"".join(s for s in strings if s is not None)
    offs = np.zeros(n + 1, dtype=np.uint64)
    acc = 0
    for i, s in enumerate(strings):
        offs[i] = acc
        if s is not None:
            acc += len(s)
    offs[n] = acc
    out = C.c_void_p()
    rc = L.rjo_encode_varchar(offs.ctypes.data, heap, valid.ctypes.data, n, C.byref(out))
    assert rc == 0
    try:
        return _result_to_table(L, out).columns[0]
    finally:
        L.rjo_result_free(out)


def from_csv(text: bytes, types, filt=None) -> "pl.ColumnarTable":
    """The oracle's Table::from_csv (oracle/rjo_ingest.c): CSV text -> the filtered ColumnarTable,
    pages filled by ColumnInserter's rule.  Raises RuntimeError with the reference's message."""
    from pyrj import capi

    L = lib()
    n = len(types)
    ct = (C.c_int32 * n)(*types)
    ops, n_ops, keep = capi.filter_to_c(filt)
    out = C.c_void_p()
    err = C.create_string_buffer(256)
    L.rjo_from_csv.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(capi.rj_filter_op), C.c_uint64,
                               C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    rc = L.rjo_from_csv(text, len(text), n, ct, ops, n_ops, C.byref(out), err, 256)
    del keep
    if rc != 0:
        raise RuntimeError(err.value.decode())
    try:
        cols = []
        for c in range(n):
            npg = int(L.rjo_result_col_pages(out, c))
            pages = np.zeros((npg, pg.PAGE_SIZE), dtype=np.uint8)
            for k in range(npg):
                C.memmove(pages[k].ctypes.data, L.rjo_result_page(out, c, k), pg.PAGE_SIZE)
            cols.append(pl.Column(int(L.rjo_result_col_type(out, c)), pages))
        return pl.ColumnarTable(int(L.rjo_result_num_rows(out)), cols)
    finally:
        L.rjo_result_free(out)
