"""ctypes binding of librj.so — the C-ABI in include/rj.h.

This is the same call sequence the C++ shim (radix-join_amd/host/contest_execute.cpp)
performs behind ``Contest::execute``: flatten the Plan, rj_execute, copy the result
pages out.  There is NO CPU fallback here: if the HIP library is missing or there is
no GPU, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import pages as pg
from . import plan as pl

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "librj.so")

RJ_EXEC_KEEP_ON_DEVICE = 1


class rj_comm_id(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]


class rj_config(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("profile", C.c_int32),
        ("stream", C.c_void_p),
        ("radix_bits", C.c_int32),
        ("n_devices", C.c_int32),
        ("devices", C.POINTER(C.c_int32)),
        ("world_size", C.c_int32),
        ("rank_base", C.c_int32),
        ("comm_id", C.POINTER(rj_comm_id)),
        ("exchange", C.c_int32),
        ("flags", C.c_int32),
    ]


EXCHANGE_AUTO, EXCHANGE_P2P, EXCHANGE_RCCL = 0, 1, 2
CTX_PREWARM = 1


class rj_tuples(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("key", C.c_void_p),
        ("carry", C.c_void_p),
        ("hashed", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class rj_filter_op(C.Structure):
    _fields_ = [("op", C.c_int32), ("column", C.c_int32), ("ivalue", C.c_int64), ("bytes", C.c_void_p)]


# rj_filter_opcode (include/rj.h): a filter is a postfix program, e.g.
#   [("LT", 2, 1990), ("IS_NULL", 4), ("NOT",), ("AND",), ("BITMAP", np.packbits(mask, bitorder="little")), ("OR",)]
F_OPS = {"EQ": 0, "NEQ": 1, "LT": 2, "GT": 3, "LEQ": 4, "GEQ": 5, "IS_NULL": 6, "IS_NOT_NULL": 7, "BITMAP": 8, "AND": 9, "OR": 10, "NOT": 11, "LIKE": 12, "NOT_LIKE": 13}


def filter_to_c(prog):
    """-> (rj_filter_op array, n, keep-alive list)"""
    prog = list(prog or [])
    arr = (rj_filter_op * max(1, len(prog)))()
    keep = []
    for k, term in enumerate(prog):
        op = F_OPS[term[0]]
        arr[k].op = op
        if (op <= 5 or op >= 12) and isinstance(term[2], (bytes, bytearray)):  # string literal / LIKE pattern
            lit = np.frombuffer(bytes(term[2]) or b"\0", dtype=np.uint8).copy()
            keep.append(lit)
            arr[k].column, arr[k].ivalue, arr[k].bytes = int(term[1]), len(term[2]), lit.ctypes.data
        elif op <= 5 and isinstance(term[2], float):  # FP64 column: the literal's bits
            arr[k].column, arr[k].ivalue = int(term[1]), int(np.array([term[2]], dtype=np.float64).view(np.int64)[0])
        elif op <= 5:
            arr[k].column, arr[k].ivalue = int(term[1]), int(term[2])
        elif op in (6, 7):
            arr[k].column = int(term[1])
        elif op == 8:
            bm = np.ascontiguousarray(term[1], dtype=np.uint8)
            keep.append(bm)
            arr[k].bytes = bm.ctypes.data
    return arr, len(prog), keep


class rj_kernel_stat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class rj_device_info(C.Structure):
    _fields_ = [
        ("name", C.c_char * 128),
        ("arch", C.c_char * 64),
        ("compute_units", C.c_int32),
        ("wavefront", C.c_int32),
        ("hbm_bytes", C.c_uint64),
        ("lds_per_cu", C.c_uint64),
        ("device_count", C.c_int32),
        ("reserved", C.c_int32),
    ]


# every symbol include/rj.h declares (tests/test_capi_symbols.py checks the export list)
EXPORTS = [
    "rj_abi_version",
    "rj_context_create",
    "rj_context_destroy",
    "rj_last_error",
    "rj_context_n_devices",
    "rj_context_device",
    "rj_comm_id_create",
    "rj_execute_sharded",
    "rj_plan_shardable",
    "rj_table_upload",
    "rj_table_adopt_device",
    "rj_table_release",
    "rj_table_from_csv",
    "rj_debug_parse_fp64",
    "rj_table_num_rows",
    "rj_table_col_pages",
    "rj_table_copy_pages",
    "rj_execute",
    "rj_execute_resident",
    "rj_result_num_rows",
    "rj_result_num_cols",
    "rj_result_col_type",
    "rj_result_col_pages",
    "rj_result_copy_pages",
    "rj_result_device_pages",
    "rj_result_free",
    "rj_exchange_plan",
    "rj_shard_partition",
    "rj_join_tuples",
    "rj_profile_read",
    "rj_profile_reset",
    "rj_device_query",
]

_LIB = None


class RjError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librj error {code}: {msg}")
        self.code = code
        self.message = msg


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 /
    libhsa-runtime64 (same soname ``libamdhip64.so.7`` as /opt/rocm's); if librj.so pulled
    in /opt/rocm's copy first and torch then loaded its own, two HSA runtimes would fight
    over the device ("No HIP GPUs are available").  Loading torch's copy first — by path,
    without importing torch — makes librj.so bind to it through the shared soname, and a
    later ``import torch`` resolves to the very same file.  Without torch installed (e.g.
    the C++ shim) librj.so uses /opt/rocm's runtime through its DT_NEEDED entry."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    path = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def preload_torch_rccl():
    """librj dlopens RCCL by soname on first use.  In a process that runs on PyTorch's bundled
    HIP runtime (see above) the RCCL that goes with it is PyTorch's own copy: load that one
    first, so the soname lookup binds to it and not to /opt/rocm's build for another runtime."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def load():
    """Load librj.so (built in-tree by ``__graft_entry__.build()`` / csrc/Makefile)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    lib_path = os.environ.get("RJ_LIB_PATH", LIB_PATH)  # tuning variants: csrc/Makefile
    if not os.path.exists(lib_path):
        raise RuntimeError(
            f"{lib_path} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback."
        )
    _preload_torch_hip_runtime()
    L = C.CDLL(lib_path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
    L.rj_abi_version.restype = C.c_int
    L.rj_context_create.argtypes = [C.POINTER(vp), C.POINTER(rj_config)]
    L.rj_context_create.restype = C.c_int
    L.rj_context_destroy.argtypes = [vp]
    L.rj_context_destroy.restype = None
    L.rj_last_error.argtypes = [vp]
    L.rj_last_error.restype = C.c_char_p
    L.rj_context_n_devices.argtypes = [vp]
    L.rj_context_n_devices.restype = C.c_uint32
    L.rj_context_device.argtypes = [vp, C.c_uint32]
    L.rj_context_device.restype = vp
    L.rj_comm_id_create.argtypes = [C.POINTER(rj_comm_id)]
    L.rj_comm_id_create.restype = C.c_int
    L.rj_execute_sharded.argtypes = [vp, C.POINTER(pl.rj_plan), C.POINTER(vp), u64, i32, C.POINTER(vp)]
    L.rj_execute_sharded.restype = C.c_int
    L.rj_plan_shardable.argtypes = [C.POINTER(pl.rj_plan), C.c_char_p, C.c_size_t]
    L.rj_plan_shardable.restype = C.c_int
    L.rj_table_upload.argtypes = [vp, C.POINTER(pl.rj_input), C.POINTER(vp)]
    L.rj_table_upload.restype = C.c_int
    L.rj_table_adopt_device.argtypes = [vp, u64, u64, C.POINTER(i32), C.POINTER(vp), C.POINTER(u64), C.POINTER(vp)]
    L.rj_table_adopt_device.restype = C.c_int
    L.rj_table_release.argtypes = [vp, vp]
    L.rj_table_release.restype = None
    L.rj_table_num_rows.argtypes = [vp]
    L.rj_table_num_rows.restype = u64
    L.rj_execute.argtypes = [vp, C.POINTER(pl.rj_plan), C.POINTER(vp)]
    L.rj_execute.restype = C.c_int
    L.rj_execute_resident.argtypes = [vp, C.POINTER(pl.rj_plan), C.POINTER(vp), u64, i32, C.POINTER(vp)]
    L.rj_execute_resident.restype = C.c_int
    L.rj_result_num_rows.argtypes = [vp]
    L.rj_result_num_rows.restype = u64
    L.rj_result_num_cols.argtypes = [vp]
    L.rj_result_num_cols.restype = u64
    L.rj_result_col_type.argtypes = [vp, u64]
    L.rj_result_col_type.restype = i32
    L.rj_result_col_pages.argtypes = [vp, u64]
    L.rj_result_col_pages.restype = u64
    L.rj_result_copy_pages.argtypes = [vp, u64, C.POINTER(vp), u64]
    L.rj_result_copy_pages.restype = C.c_int
    L.rj_result_device_pages.argtypes = [vp, u64]
    L.rj_result_device_pages.restype = vp
    L.rj_result_free.argtypes = [vp]
    L.rj_result_free.restype = None
    L.rj_shard_partition.argtypes = [vp, vp, u64, u64, C.c_uint32, C.POINTER(rj_tuples), C.POINTER(u64)]
    L.rj_shard_partition.restype = C.c_int
    L.rj_join_tuples.argtypes = [vp, C.POINTER(rj_tuples), C.POINTER(rj_tuples), C.c_uint32, i32, C.POINTER(vp)]
    L.rj_join_tuples.restype = C.c_int
    L.rj_profile_read.argtypes = [vp, C.POINTER(rj_kernel_stat), u64, C.POINTER(u64)]
    L.rj_profile_read.restype = C.c_int
    L.rj_profile_reset.argtypes = [vp]
    L.rj_profile_reset.restype = None
    L.rj_device_query.argtypes = [vp, C.POINTER(rj_device_info)]
    L.rj_device_query.restype = C.c_int
    _LIB = L
    return L


class Result:
    """rj_result*: the ColumnarTable ``execute`` returns (pages still in HBM)."""

    def __init__(self, ctx: "Context", handle):
        self.ctx, self.h = ctx, handle

    @property
    def num_rows(self):
        return self.ctx.L.rj_result_num_rows(self.h)

    @property
    def num_cols(self):
        return self.ctx.L.rj_result_num_cols(self.h)

    def col_type(self, c):
        return self.ctx.L.rj_result_col_type(self.h, c)

    def col_pages(self, c):
        return self.ctx.L.rj_result_col_pages(self.h, c)

    def device_pages(self, c):
        return self.ctx.L.rj_result_device_pages(self.h, c)

    def to_table(self) -> pl.ColumnarTable:
        """Copy every column into host pages (what the shim does into ``new Page``s)."""
        L = self.ctx.L
        t = pl.ColumnarTable(self.num_rows, [])
        for c in range(self.num_cols):
            n = self.col_pages(c)
            pages = np.zeros((n, pg.PAGE_SIZE), dtype=np.uint8)
            if n:
                ptrs = (C.c_void_p * n)()
                base = pages.ctypes.data
                for i in range(n):
                    ptrs[i] = base + i * pg.PAGE_SIZE
                self.ctx._check(L.rj_result_copy_pages(self.h, c, ptrs, n))
            t.columns.append(pl.Column(self.col_type(c), pages))
        return t

    def free(self):
        # a result / table holds blocks of its context's HBM cache: once the context is gone
        # (destroy() before the object's finaliser ran) there is nothing left to give back
        if self.h and getattr(self.ctx, "h", None):
            self.ctx.L.rj_result_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Table:
    """rj_table*: a device-resident ColumnarTable."""

    def __init__(self, ctx: "Context", handle, keep=None):
        self.ctx, self.h, self.keep = ctx, handle, keep

    def release(self):
        if self.h and getattr(self.ctx, "h", None):
            self.ctx.L.rj_table_release(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Context:
    """rj_context*: ``Contest::build_context()`` / ``destroy_context()``."""

    def __init__(self, device=-1, profile=False, stream=None, radix_bits=0, devices=None, world_size=0,
                 rank_base=0, comm_id=None, exchange=EXCHANGE_AUTO, prewarm=False, _lane_of=None, _handle=None):
        """devices: HIP ordinals this context owns (one rank each; an ordinal may repeat =
        virtual ranks on one GPU); world_size/rank_base/comm_id: this process' place in a
        multi-process job (comm_id = bytes from ``make_comm_id()`` of one process)."""
        self.L = load()
        self._lanes = None
        if _lane_of is not None:  # a device of a group context: borrowed handle
            self.h, self.group = C.c_void_p(_handle), _lane_of
            return
        self.group = None
        cfg = rj_config()
        cfg.device, cfg.profile, cfg.stream, cfg.radix_bits = device, int(profile), stream, radix_bits  # 1: the hot kernels, 2: every launch
        if devices is not None:
            self._devs = (C.c_int32 * len(devices))(*devices)
            cfg.n_devices, cfg.devices = len(devices), self._devs
        cfg.world_size, cfg.rank_base, cfg.exchange = world_size, rank_base, exchange
        cfg.flags = CTX_PREWARM if prewarm else 0  # what Contest::build_context() sets
        if comm_id is not None or exchange == EXCHANGE_RCCL:
            preload_torch_rccl()
        if comm_id is not None:
            self._cid = rj_comm_id()
            if len(comm_id) != 128:
                raise ValueError("comm_id must be the 128 bytes of make_comm_id()")
            C.memmove(C.addressof(self._cid), bytes(comm_id), 128)  # (.bytes would be a copy)
            cfg.comm_id = C.pointer(self._cid)
        h = C.c_void_p()
        rc = self.L.rj_context_create(C.byref(h), C.byref(cfg))
        if rc != 0:
            raise RjError(rc, (self.L.rj_last_error(None) or b"").decode())
        self.h = h

    @property
    def n_devices(self):
        return int(self.L.rj_context_n_devices(self.h))

    def lane(self, i) -> "Context":
        """The single-device context of local device i (tables are uploaded / adopted per
        device); owned by this context."""
        if self._lanes is None:
            self._lanes = [Context(_lane_of=self, _handle=self.L.rj_context_device(self.h, k)) for k in range(self.n_devices)]
        return self._lanes[i]

    def execute_sharded(self, plan: pl.Plan, tables_per_device):
        """rj_execute_sharded: tables_per_device[d][i] = shard of input i on local device d.
        -> one Result per local device (its slice of the join result)."""
        cached = getattr(plan, "_c_resident", None)
        if cached is None:
            cached = pl.plan_to_c(plan, with_inputs=False)
            plan._c_resident = cached
        cplan, keep = cached
        nd = self.n_devices
        n_in = len(tables_per_device[0])
        flat = [t.h for d in range(nd) for t in tables_per_device[d]]
        hs = (C.c_void_p * max(1, len(flat)))(*flat)
        outs = (C.c_void_p * nd)()
        self._check(self.L.rj_execute_sharded(self.h, C.byref(cplan), hs, n_in, RJ_EXEC_KEEP_ON_DEVICE, outs))
        return [Result(self.lane(d), C.c_void_p(outs[d])) for d in range(nd)]

    def _check(self, rc):
        if rc != 0:
            raise RjError(rc, (self.L.rj_last_error(self.h) or b"").decode())

    def destroy(self):
        if getattr(self, "group", None) is not None:
            return  # a device of a group: the group owns it
        if getattr(self, "h", None):
            for ln in self._lanes or []:
                ln.h = None
            self.L.rj_context_destroy(self.h)
            self.h = None

    # -- tables
    def upload(self, t: pl.ColumnarTable) -> Table:
        keep: list = []
        inp = pl.input_to_c(t, keep)
        h = C.c_void_p()
        self._check(self.L.rj_table_upload(self.h, C.byref(inp), C.byref(h)))
        return Table(self, h)

    def from_csv(self, text: bytes, types, filt=None) -> Table:
        """rj_table_from_csv: ``Table::from_csv`` on the device — CSV text (the harness's dialect) ->
        a resident table of the rows that pass `filt` (a postfix program, see F_OPS)."""
        n = len(types)
        ct = (C.c_int32 * n)(*types)
        ops, n_ops, keep = filter_to_c(filt)
        h = C.c_void_p()
        self.L.rj_table_from_csv.restype = C.c_int
        self.L.rj_table_from_csv.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(rj_filter_op),
                                             C.c_uint64, C.POINTER(C.c_void_p)]
        self._check(self.L.rj_table_from_csv(self.h, text, len(text), n, ct, ops, n_ops, C.byref(h)))
        del keep
        t = Table(self, h)
        t.types = list(types)
        return t

    def table_to_host(self, t: Table, types=None) -> pl.ColumnarTable:
        """The pages of a resident table, copied out (rj_table_copy_pages)."""
        types = types or t.types
        L = self.L
        L.rj_table_num_rows.restype = C.c_uint64
        L.rj_table_num_rows.argtypes = [C.c_void_p]
        L.rj_table_col_pages.restype = C.c_uint64
        L.rj_table_col_pages.argtypes = [C.c_void_p, C.c_uint64]
        L.rj_table_copy_pages.restype = C.c_int
        L.rj_table_copy_pages.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_uint64]
        cols = []
        for c, ty in enumerate(types):
            npg = int(L.rj_table_col_pages(t.h, c))
            pages = np.zeros((npg, pg.PAGE_SIZE), dtype=np.uint8)
            ptrs = (C.c_void_p * max(1, npg))(*[pages[i].ctypes.data for i in range(npg)])
            self._check(L.rj_table_copy_pages(self.h, t.h, c, ptrs, npg))
            cols.append(pl.Column(ty, pages))
        return pl.ColumnarTable(int(L.rj_table_num_rows(t.h)), cols)

    def adopt_device(self, num_rows, types, dev_ptrs, n_pages, keep=None) -> Table:
        n = len(types)
        ct = (C.c_int32 * n)(*types)
        cp = (C.c_void_p * n)(*dev_ptrs)
        cn = (C.c_uint64 * n)(*n_pages)
        h = C.c_void_p()
        self._check(self.L.rj_table_adopt_device(self.h, num_rows, n, ct, cp, cn, C.byref(h)))
        return Table(self, h, keep)

    # -- execute
    def execute(self, plan: pl.Plan) -> pl.ColumnarTable:
        """``Contest::execute(plan, ctx)``: host pages in, host pages out."""
        cplan, keep = pl.plan_to_c(plan)
        out = C.c_void_p()
        self._check(self.L.rj_execute(self.h, C.byref(cplan), C.byref(out)))
        r = Result(self, out)
        try:
            return r.to_table()
        finally:
            r.free()
            del keep

    def execute_resident(self, plan: pl.Plan, tables, keep_on_device=True) -> Result:
        # the flattened plan is cached on the Plan object (callers that run one plan many
        # times, like bench.py, should not re-marshal it every step)
        cached = getattr(plan, "_c_resident", None)
        if cached is None:
            cached = pl.plan_to_c(plan, with_inputs=False)
            plan._c_resident = cached
        cplan, keep = cached
        hs = (C.c_void_p * max(1, len(tables)))(*[t.h for t in tables])
        out = C.c_void_p()
        flags = RJ_EXEC_KEEP_ON_DEVICE if keep_on_device else 0
        self._check(self.L.rj_execute_resident(self.h, C.byref(cplan), hs, len(tables), flags, C.byref(out)))
        return Result(self, out)

    # -- sharded path
    def shard_partition(self, table: Table, key_col, carry_col, n_ranks, key_ptr, carry_ptr):
        """Stage A into caller-owned device buffers (capacity = table rows)."""
        tup = rj_tuples(0, key_ptr, carry_ptr, 0, 0)
        counts = (C.c_uint64 * n_ranks)()
        self._check(self.L.rj_shard_partition(self.h, table.h, key_col, carry_col, n_ranks, C.byref(tup), counts))
        return int(tup.n), [int(c) for c in counts]

    def join_tuples(self, build, probe, skip_rank_bits=0, hashed=True) -> Result:
        """Stage B: build/probe = (n, key_ptr, carry_ptr) in HBM."""
        build = rj_tuples(build[0], build[1], build[2], 1 if hashed else 0, 0)
        probe = rj_tuples(probe[0], probe[1], probe[2], 1 if hashed else 0, 0)
        out = C.c_void_p()
        self._check(self.L.rj_join_tuples(self.h, C.byref(build), C.byref(probe), skip_rank_bits, RJ_EXEC_KEEP_ON_DEVICE, C.byref(out)))
        return Result(self, out)

    # -- profiling
    def profile(self):
        n = C.c_uint64()
        buf = (rj_kernel_stat * 64)()
        self._check(self.L.rj_profile_read(self.h, buf, 64, C.byref(n)))
        return [
            {"name": buf[i].name.decode(), "launches": int(buf[i].launches), "total_ms": float(buf[i].total_ms)}
            for i in range(min(n.value, 64))
        ]

    def profile_reset(self):
        self.L.rj_profile_reset(self.h)

    def device_info(self):
        d = rj_device_info()
        self._check(self.L.rj_device_query(self.h, C.byref(d)))
        return {
            "name": d.name.decode(),
            "arch": d.arch.decode(),
            "compute_units": d.compute_units,
            "wavefront": d.wavefront,
            "hbm_bytes": int(d.hbm_bytes),
            "lds_per_cu": int(d.lds_per_cu),
            "device_count": int(d.device_count),
        }


def plan_shardable(plan: pl.Plan):
    """rj_plan_shardable -> (bool, reason).  Needs no GPU."""
    L = load()
    cplan, keep = pl.plan_to_c(plan, with_inputs=False)
    buf = C.create_string_buffer(256)
    ok = L.rj_plan_shardable(C.byref(cplan), buf, 256)
    del keep
    return bool(ok), buf.value.decode()


def exchange_plan(world: int, subs: int, rank: int, counts) -> dict:
    """rj_exchange_plan: rank `rank`'s half of the exchange step of a sharded join, from the
    all-gathered count tensor counts[src, dst, sub] (tuples).  Host arithmetic only: needs no GPU."""
    L = load()
    cnt = np.ascontiguousarray(np.asarray(counts, dtype=np.uint64).reshape(world, world, subs))
    u64 = lambda n: np.zeros(n, dtype=np.uint64)  # noqa: E731
    u32 = lambda n: np.zeros(n, dtype=np.uint32)  # noqa: E731
    out = {
        "send_off": u64(world), "send_cnt": u64(world), "recv_off": u64(world), "recv_cnt": u64(world),
        "seg_begin": u32(subs * world), "seg_end": u32(subs * world), "part_off": u32(subs + 1),
    }
    n_recv = C.c_uint64(0)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    L.rj_exchange_plan.restype = C.c_int
    L.rj_exchange_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 8 + [C.POINTER(C.c_uint64)]
    rc = L.rj_exchange_plan(world, subs, rank, ptr(cnt), ptr(out["send_off"]), ptr(out["send_cnt"]), ptr(out["recv_off"]),
                            ptr(out["recv_cnt"]), ptr(out["seg_begin"]), ptr(out["seg_end"]), ptr(out["part_off"]),
                            C.byref(n_recv))
    if rc != 0:
        L.rj_last_error.restype = C.c_char_p
        raise RjError(rc, (L.rj_last_error(None) or b"").decode())
    out["n_recv"] = int(n_recv.value)
    return out


def parse_fp64(field: bytes):
    """rj_debug_parse_fp64: the ingest's FP64 field parser on the host.  -> (status, bits): 0 parsed,
    1 out of range, 2 left to std::from_chars."""
    L = load()
    L.rj_debug_parse_fp64.restype = C.c_int
    L.rj_debug_parse_fp64.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]
    bits = C.c_uint64(0)
    st = L.rj_debug_parse_fp64(field, len(field), C.byref(bits))
    return int(st), int(bits.value)


def make_comm_id() -> bytes:
    """rj_comm_id_create: 128 bytes to hand to every rank of a multi-process job."""
    L = load()
    preload_torch_rccl()
    cid = rj_comm_id()
    rc = L.rj_comm_id_create(C.byref(cid))
    if rc != 0:
        raise RjError(rc, (L.rj_last_error(None) or b"").decode())
    return C.string_at(C.addressof(cid), 128)  # (c_char arrays stop at the first NUL when read as bytes)


# ---------------------------------------------------------------- reference API --
def build_context(**kw):
    """``Contest::build_context()`` (reference src/execute.cpp:326-328)."""
    return Context(**kw)


def destroy_context(ctx: Context):
    """``Contest::destroy_context()`` (reference src/execute.cpp:330)."""
    ctx.destroy()


def execute(plan: pl.Plan, ctx: Context) -> pl.ColumnarTable:
    """``Contest::execute(const Plan&, void*)`` (reference src/execute.cpp:316-324)."""
    return ctx.execute(plan)
