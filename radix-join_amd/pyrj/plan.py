"""Host-side mirror of the reference's plan/data model, for tests and bench.

Same names and argument meaning as the reference (include/plan.h):
  * ``Column`` / ``ColumnarTable``            — plan.h:60-105
  * ``Plan.new_scan_node/new_join_node/new_input`` — plan.h:118-148
  * ``ScanNode{base_table_id}`` / ``JoinNode{build_left,left,right,left_attr,right_attr}``
    and ``output_attrs = [(index, DataType)]``  — plan.h:32-52
so the parity tests read like reference tests/unit_tests.cpp.  Pages are numpy
``uint8[n_pages, 8192]`` arrays (one row per Page).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import pages as pg

INT32, INT64, FP64, VARCHAR = pg.INT32, pg.INT64, pg.FP64, pg.VARCHAR
TYPE_NAMES = {INT32: "INT32", INT64: "INT64", FP64: "FP64", VARCHAR: "VARCHAR"}
TYPE_IDS = {v: k for k, v in TYPE_NAMES.items()}


@dataclass
class Column:
    type: int
    pages: np.ndarray = field(default_factory=lambda: np.zeros((0, pg.PAGE_SIZE), np.uint8))


@dataclass
class ColumnarTable:
    num_rows: int = 0
    columns: list = field(default_factory=list)


@dataclass
class ScanNode:
    base_table_id: int


@dataclass
class JoinNode:
    build_left: bool
    left: int
    right: int
    left_attr: int
    right_attr: int


@dataclass
class PlanNode:
    data: object
    output_attrs: list  # [(index, DataType)]


class Plan:
    def __init__(self):
        self.nodes: list[PlanNode] = []
        self.inputs: list[ColumnarTable] = []
        self.root: int = 0

    def new_join_node(self, build_left, left, right, left_attr, right_attr, output_attrs):
        self.nodes.append(PlanNode(JoinNode(bool(build_left), left, right, left_attr, right_attr), list(output_attrs)))
        return len(self.nodes) - 1

    def new_scan_node(self, base_table_id, output_attrs):
        self.nodes.append(PlanNode(ScanNode(base_table_id), list(output_attrs)))
        return len(self.nodes) - 1

    def new_input(self, table: ColumnarTable):
        self.inputs.append(table)
        return len(self.inputs) - 1


# ------------------------------------------------------------ table helpers --
def make_table(columns) -> ColumnarTable:
    """columns: list of (dtype, values[, valid]) for fixed width, or
    (VARCHAR, list_of_bytes_or_None).  Pages follow the reference fill rule
    (what ``Table(...).to_columnar()`` produces in the reference tests)."""
    t = ColumnarTable()
    n = None
    for spec in columns:
        dt = spec[0]
        if dt == VARCHAR:
            vals = list(spec[1])
            cn = len(vals)
            t.columns.append(Column(dt, pg.pack_varchar(vals)))
        else:
            vals = np.asarray(spec[1], dtype=pg.NP_DTYPE[dt])
            valid = spec[2] if len(spec) > 2 else None
            cn = vals.shape[0]
            t.columns.append(Column(dt, pg.pack_fixed(vals, valid, dt)))
        if n is None:
            n = cn
        elif n != cn:
            raise ValueError("ragged columns")
    t.num_rows = n or 0
    return t


def table_from_rows(rows, types) -> ColumnarTable:
    """``Table(rows, types).to_columnar()`` of the reference tests: rows is a
    list of tuples with None for NULL."""
    cols = []
    for ci, dt in enumerate(types):
        col = [r[ci] for r in rows]
        if dt == VARCHAR:
            cols.append((dt, [None if v is None else (v.encode() if isinstance(v, str) else v) for v in col]))
        else:
            valid = np.array([v is not None for v in col], dtype=bool)
            vals = np.array([0 if v is None else v for v in col], dtype=pg.NP_DTYPE[dt])
            cols.append((dt, vals, valid))
    t = make_table(cols)
    t.num_rows = len(rows)
    return t


def decode_table(t: ColumnarTable):
    """-> list per column of (values, valid) or list[bytes|None] (VARCHAR)."""
    out = []
    for c in t.columns:
        if c.type == VARCHAR:
            out.append(pg.unpack_varchar(c.pages, t.num_rows))
        else:
            out.append(pg.unpack_fixed(c.pages, t.num_rows, c.type))
    return out


def table_rows(t: ColumnarTable):
    """``Table::from_columnar(t).table()`` as a list of tuples (None = NULL)."""
    cols = decode_table(t)
    rows = []
    for i in range(t.num_rows):
        r = []
        for c, col in zip(t.columns, cols):
            if c.type == VARCHAR:
                r.append(col[i])
            else:
                vals, valid = col
                r.append(vals[i].item() if valid[i] else None)
        rows.append(tuple(r))
    return rows


def _sort_key(row):
    # std::variant ordering is (index, value) with monostate last (index 4);
    # any total order works for multiset comparison
    return tuple((1, 0) if v is None else (0, v) for v in row)


def sorted_rows(t: ColumnarTable):
    return sorted(table_rows(t), key=_sort_key)


def canonical_rows(t: ColumnarTable):
    """sorted_rows with floats replaced by their bit patterns, so that NaN payloads (and
    -0.0 / +0.0) compare by identity of the stored value — for differential tests."""
    import struct

    def canon(v):
        return ("f64", struct.unpack("<q", struct.pack("<d", v))[0]) if isinstance(v, float) else v

    rows = [tuple(canon(v) for v in r) for r in table_rows(t)]
    return sorted(rows, key=lambda r: tuple((2, 0) if v is None else ((1, v[1]) if isinstance(v, tuple) else (0, v)) for v in r))


def table_digest(t: ColumnarTable):
    """Order-independent digest of a fixed-width table: (rows, sum, xor) of a
    64-bit row hash — for sizes where sorting rows in Python is too slow."""
    n = t.num_rows
    h = np.full(n, 0x9E3779B97F4A7C15, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for c in t.columns:
            if c.type == VARCHAR:
                raise ValueError("digest supports fixed-width columns only")
            vals, valid = pg.unpack_fixed(c.pages, n, c.type)
            if c.type == INT32:
                bits = vals.astype(np.int64).view(np.uint64)  # the declared type is mixed in below
            else:
                bits = np.ascontiguousarray(vals).view(np.uint64)
            x = np.where(valid, bits, np.uint64(0xDEADBEEFCAFEF00D)) + np.uint64(c.type + 1)
            x = x ^ (x >> np.uint64(33))
            x = x * np.uint64(0xFF51AFD7ED558CCD)
            x = x ^ (x >> np.uint64(33))
            h = (h ^ x) * np.uint64(0xC4CEB9FE1A85EC53)
            h = h ^ (h >> np.uint64(29))
        s = int(h.sum(dtype=np.uint64)) if n else 0
        x = int(np.bitwise_xor.reduce(h)) if n else 0
    return (n, s, x)


# ------------------------------------------------------------- C marshalling --
class rj_node(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("build_left", C.c_int32),
        ("base_table_id", C.c_uint64),
        ("left", C.c_uint64),
        ("right", C.c_uint64),
        ("left_attr", C.c_uint64),
        ("right_attr", C.c_uint64),
        ("n_out", C.c_uint64),
        ("out_idx", C.POINTER(C.c_uint64)),
        ("out_type", C.POINTER(C.c_int32)),
    ]


class rj_column(C.Structure):
    _fields_ = [("type", C.c_int32), ("n_pages", C.c_uint64), ("pages", C.POINTER(C.c_void_p))]


class rj_input(C.Structure):
    _fields_ = [("num_rows", C.c_uint64), ("n_cols", C.c_uint64), ("cols", C.POINTER(rj_column))]


class rj_plan(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint64),
        ("nodes", C.POINTER(rj_node)),
        ("n_inputs", C.c_uint64),
        ("inputs", C.POINTER(rj_input)),
        ("root", C.c_uint64),
    ]


def input_to_c(t: ColumnarTable, keep: list) -> rj_input:
    cols = (rj_column * max(1, len(t.columns)))()
    keep.append(cols)
    for ci, c in enumerate(t.columns):
        pages = np.ascontiguousarray(c.pages, dtype=np.uint8).reshape(-1, pg.PAGE_SIZE)
        keep.append(pages)
        n = pages.shape[0]
        ptrs = (C.c_void_p * max(1, n))()
        base = pages.ctypes.data
        for i in range(n):
            ptrs[i] = base + i * pg.PAGE_SIZE
        keep.append(ptrs)
        cols[ci].type = c.type
        cols[ci].n_pages = n
        cols[ci].pages = C.cast(ptrs, C.POINTER(C.c_void_p))
    inp = rj_input()
    inp.num_rows = t.num_rows
    inp.n_cols = len(t.columns)
    inp.cols = C.cast(cols, C.POINTER(rj_column))
    return inp


def plan_to_c(plan: Plan, with_inputs: bool = True):
    """-> (rj_plan, keepalive).  The flattening the C++ shim performs."""
    keep: list = []
    nodes = (rj_node * max(1, len(plan.nodes)))()
    keep.append(nodes)
    for i, n in enumerate(plan.nodes):
        nd = nodes[i]
        k = len(n.output_attrs)
        idx = (C.c_uint64 * max(1, k))(*[a[0] for a in n.output_attrs])
        typ = (C.c_int32 * max(1, k))(*[int(a[1]) for a in n.output_attrs])
        keep += [idx, typ]
        nd.n_out = k
        nd.out_idx = C.cast(idx, C.POINTER(C.c_uint64))
        nd.out_type = C.cast(typ, C.POINTER(C.c_int32))
        if isinstance(n.data, JoinNode):
            nd.kind = 1
            nd.build_left = 1 if n.data.build_left else 0
            nd.left, nd.right = n.data.left, n.data.right
            nd.left_attr, nd.right_attr = n.data.left_attr, n.data.right_attr
        else:
            nd.kind = 0
            nd.base_table_id = n.data.base_table_id
    p = rj_plan()
    p.n_nodes = len(plan.nodes)
    p.nodes = C.cast(nodes, C.POINTER(rj_node))
    p.root = plan.root
    if with_inputs:
        ins = (rj_input * max(1, len(plan.inputs)))()
        keep.append(ins)
        for i, t in enumerate(plan.inputs):
            ins[i] = input_to_c(t, keep)
        p.n_inputs = len(plan.inputs)
        p.inputs = C.cast(ins, C.POINTER(rj_input))
    else:
        p.n_inputs = len(plan.inputs)
        p.inputs = None
    return p, keep
