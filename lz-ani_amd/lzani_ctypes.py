"""ctypes binding of the C-ABI in include/lzani.h (liblzani_hip.so).

Used by the test-suite, bench.py and __graft_entry__.py.  It is deliberately thin: the product
is the shared library; this module only marshals numpy arrays into the plain pointers and sizes
the ABI takes.  There is no CPU fallback: if the HIP library is missing or a call fails, it raises.

Interface mirror: `Engine` plays the role of the reference's CParser + the worker loop of
CLZMatcher::do_matching (/root/reference/src/parser.h:237-253, lz_matcher.cpp:172-277):
construct with the LZ parameters, hand over the sequences, run rows of (reference, queries).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("LZANI_LIB") or os.path.join(HERE, "liblzani_hip.so")   # LZANI_LIB: diagnostic builds
SRC = os.path.join(HERE, "csrc", "lzani_hip.hip")

PARAM_ORDER = ("mal", "msl", "mrd", "mqd", "reg", "aw", "am", "ar")
DEFAULT_PARAMS = dict(mal=11, msl=7, mrd=40, mqd=40, reg=35, aw=15, am=7, ar=3)

ERRORS = {-1: "LZANI_ERR_ARG", -2: "LZANI_ERR_PARAMS", -3: "LZANI_ERR_DEVICE",
          -4: "LZANI_ERR_STATE", -5: "LZANI_ERR_NOMEM"}

EXPORTS = ("lzani_default_params", "lzani_create", "lzani_destroy", "lzani_last_error",
           "lzani_set_genomes", "lzani_run_rows", "lzani_run_rows_device", "lzani_get_timing",
           "lzani_debug_get_index", "lzani_run_rows_regions", "lzani_get_layout",
           "lzani_row_costs", "lzani_partition_rows", "lzani_comm_unique_id", "lzani_comm_init", "lzani_comm_allgather",
           "lzani_comm_gatherv", "lzani_group_create", "lzani_group_destroy", "lzani_group_last_error",
           "lzani_group_set_genomes", "lzani_group_run_rows", "lzani_group_get_timing", "lzani_plan_gather", "lzani_get_rtc_info", "lzani_debug_rtc_compile",
           "lzani_debug_sort_segments")


class LzaniError(RuntimeError):
    pass


class Timing(C.Structure):
    _fields_ = [("index_ms", C.c_double), ("pairs_ms", C.c_double), ("pair_launches", C.c_uint32),
                ("index_launches", C.c_uint32), ("pairs", C.c_uint64), ("cand_ms", C.c_double), ("kmers_ms", C.c_double),
                ("cand_launches", C.c_uint32), ("reserved_", C.c_uint32)]


class LayoutInfo(C.Structure):
    _fields_ = [("key_bits", C.c_int32), ("dir_bits", C.c_int32), ("pos_bits", C.c_int32), ("tag_mask", C.c_uint32),
                ("kmer_words", C.c_int32), ("bucket_table", C.c_int32), ("tag_words", C.c_int32), ("n_free", C.c_int32),
                ("slots", C.c_uint32), ("batches_last_run", C.c_uint32), ("bytes_per_slot", C.c_uint64),
                ("bytes_genomes", C.c_uint64), ("join_lists", C.c_int32), ("block_launches", C.c_int32),
                ("bitmap_launches", C.c_int32), ("rtc_launches", C.c_int32),
                ("lpt_launches", C.c_int32), ("matrix_from_index", C.c_int32), ("split_launches", C.c_int32), ("reserved_", C.c_int32),
                ("split_segments", C.c_uint64)]


class RtcInfo(C.Structure):
    _fields_ = [("folded_ahead_of_time", C.c_int32), ("null_chain", C.c_int32), ("kernels_built", C.c_int32),
                ("kernels_from_cache", C.c_int32), ("kernels_failed", C.c_int32), ("reserved_", C.c_int32), ("build_ms", C.c_double)]


def build_library(force=False):
    """hipcc cross-compiles for gfx950 without a GPU present."""
    deps = [SRC] + [os.path.join(HERE, "csrc", h) for h in ("lzani_core.h", "lzani_layout.h", "lzani_kernels_index.h",
                                                             "lzani_kernels_cand.h", "lzani_kernels_pairs.h", "lzani_kernels_split.h", "lzani_multi.h", "lzani_sort.hip", "lzani_tables.h", "lzani_rtc.h")] + [os.path.join(ROOT, "include", "lzani.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wno-unused-value", "-I" + os.path.join(HERE, "csrc"), "-I" + os.path.join(ROOT, "include"),      # (-I: the .incbin of lzani_rtc.h)
           "-o", LIB_PATH, SRC, os.path.join(HERE, "csrc", "lzani_sort.hip"), "-lrccl", "-lhiprtc"]
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LzaniError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback for the HIP path)")
        lib = C.CDLL(LIB_PATH)
        lib.lzani_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.lzani_destroy.argtypes = [C.c_void_p]
        lib.lzani_destroy.restype = None
        lib.lzani_last_error.argtypes = [C.c_void_p]
        lib.lzani_last_error.restype = C.c_char_p
        lib.lzani_set_genomes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.lzani_run_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.lzani_run_rows_device.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.lzani_get_timing.argtypes = [C.c_void_p, C.c_void_p]
        lib.lzani_get_layout.argtypes = [C.c_void_p, C.c_void_p]
        lib.lzani_get_rtc_info.argtypes = [C.c_void_p, C.c_void_p]
        lib.lzani_debug_rtc_compile.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_uint64]
        lib.lzani_debug_rtc_compile.restype = C.c_int64
        lib.lzani_run_rows_regions.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_uint64, C.c_void_p]
        lib.lzani_debug_get_index.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
        lib.lzani_debug_sort_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int]
        lib.lzani_row_costs.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.lzani_partition_rows.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.lzani_comm_unique_id.argtypes = [C.c_void_p]
        lib.lzani_comm_init.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.lzani_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        lib.lzani_comm_gatherv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        lib.lzani_group_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]
        lib.lzani_group_destroy.argtypes = [C.c_void_p]
        lib.lzani_group_destroy.restype = None
        lib.lzani_group_last_error.argtypes = [C.c_void_p]
        lib.lzani_group_last_error.restype = C.c_char_p
        lib.lzani_group_set_genomes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.lzani_group_run_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.lzani_group_get_timing.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.lzani_plan_gather.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
        _lib = lib
    return _lib


def params_array(params=None):
    p = dict(DEFAULT_PARAMS)
    if params:
        p.update(params)
    return (C.c_int32 * 8)(*[int(p[k]) for k in PARAM_ORDER]), p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def dense_rows(n, rows=None):
    """CSR description of dense all2all rows (query_ids = None): ref_ids, row_off."""
    ref_ids = np.arange(n, dtype=np.uint32) if rows is None else np.asarray(rows, dtype=np.uint32)
    row_off = np.arange(len(ref_ids) + 1, dtype=np.uint64) * np.uint64(max(n - 1, 0))
    return ref_ids, row_off


def row_costs(ref_ids, row_off, query_ids, lens):
    """cost(row) = sum of query lengths + LZANI_ROW_COST_REF_WEIGHT * reference length (lzani_row_costs; no GPU needed)."""
    lib = load_library()
    ref_ids = np.ascontiguousarray(ref_ids, dtype=np.uint32)
    row_off = np.ascontiguousarray(row_off, dtype=np.uint64)
    q = None if query_ids is None else np.ascontiguousarray(query_ids, dtype=np.uint32)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    cost = np.zeros(len(ref_ids), dtype=np.uint64)
    rc = lib.lzani_row_costs(len(ref_ids), _ptr(ref_ids), _ptr(row_off), _ptr(q), len(lens), _ptr(lens), _ptr(cost))
    if rc != 0:
        raise LzaniError(f"lzani_row_costs: {ERRORS.get(rc, rc)}")
    return cost


def plan_gather(row_off, part_of_row, n_parts):
    """Shard bookkeeping of lzani_group_run_rows (lzani_plan_gather; no GPU needed): shard_base[n_parts + 1] and the
    scatter table src / dst / cnt / row_of_entry, one entry per row, shard by shard."""
    lib = load_library()
    row_off = np.ascontiguousarray(row_off, dtype=np.uint64)
    part = np.ascontiguousarray(part_of_row, dtype=np.uint32)
    n = len(part)
    base = np.zeros(n_parts + 1, dtype=np.uint64)
    src, dst, cnt = (np.zeros(n, dtype=np.uint64) for _ in range(3))
    row = np.zeros(n, dtype=np.uint32)
    rc = lib.lzani_plan_gather(n, _ptr(row_off), _ptr(part), n_parts, _ptr(base), _ptr(src), _ptr(dst), _ptr(cnt), _ptr(row))
    if rc != 0:
        raise LzaniError(f"lzani_plan_gather: {ERRORS.get(rc, rc)}")
    return base, src, dst, cnt, row


def partition_rows(n_rows, n_parts, row_cost=None):
    """Shard index of every row (lzani_partition_rows: cyclic without costs, greedy LPT with; no GPU needed)."""
    lib = load_library()
    cost = None if row_cost is None else np.ascontiguousarray(row_cost, dtype=np.uint64)
    part = np.zeros(n_rows, dtype=np.uint32)
    rc = lib.lzani_partition_rows(n_rows, _ptr(cost), n_parts, _ptr(part))
    if rc != 0:
        raise LzaniError(f"lzani_partition_rows: {ERRORS.get(rc, rc)}")
    return part


def rtc_compile(params=None, nfree=True, cand=2, arch="gfx950"):
    """Compile-only check of the run-time compiled pair kernel (no GPU needed): (code object bytes or negative code, log)."""
    lib = load_library()
    arr, _ = params_array(params)
    log = C.create_string_buffer(1 << 16)
    n = lib.lzani_debug_rtc_compile(arr, int(bool(nfree)), int(cand), arch.encode(), log, len(log))
    return int(n), log.value.decode(errors="replace")


def comm_unique_id():
    lib = load_library()
    buf = (C.c_uint8 * 128)()
    rc = lib.lzani_comm_unique_id(buf)
    if rc != 0:
        raise LzaniError(f"lzani_comm_unique_id: {ERRORS.get(rc, rc)}")
    return bytes(buf)


class Group:
    """lzani_group_*: one process, several GPUs (what `lz-ani --gpus n` uses)."""

    def __init__(self, params=None, devices=(0,)):
        self.lib = load_library()
        arr, self.params = params_array(params)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self.lib.lzani_group_create(arr, len(devices), devs, C.byref(h))
        if rc != 0:
            raise LzaniError(f"lzani_group_create failed: {ERRORS.get(rc, rc)}")
        self.h = h
        self.n_dev = len(devices)

    def close(self):
        if getattr(self, "h", None):
            self.lib.lzani_group_destroy(self.h)
            self.h = None

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.lzani_group_last_error(self.h)
            raise LzaniError(f"{what}: {ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")

    def set_genomes(self, seqs):
        seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        ptrs = (C.c_void_p * len(seqs))(*[s.ctypes.data for s in seqs])
        lens = np.array([len(s) for s in seqs], dtype=np.uint32)
        self._check(self.lib.lzani_group_set_genomes(self.h, len(seqs), ptrs, _ptr(lens)), "lzani_group_set_genomes")
        self.n = len(seqs)

    def run_rows(self, ref_ids, row_off, query_ids=None):
        ref_ids = np.ascontiguousarray(ref_ids, dtype=np.uint32)
        row_off = np.ascontiguousarray(row_off, dtype=np.uint64)
        q = None if query_ids is None else np.ascontiguousarray(query_ids, dtype=np.uint32)
        n_pairs = int(row_off[-1]) if len(row_off) else 0
        out = np.zeros((n_pairs, 3), dtype=np.int32)
        self._check(self.lib.lzani_group_run_rows(self.h, len(ref_ids), _ptr(ref_ids), _ptr(row_off), _ptr(q), _ptr(out)),
                    "lzani_group_run_rows")
        return out

    def timing(self, device_index=0):
        t = Timing()
        g = C.c_double(0)
        self._check(self.lib.lzani_group_get_timing(self.h, device_index, C.byref(t), C.byref(g)), "lzani_group_get_timing")
        return dict(index_ms=t.index_ms, pairs_ms=t.pairs_ms, pair_launches=t.pair_launches, pairs=t.pairs, gather_ms=g.value,
                    cand_ms=t.cand_ms, kmers_ms=t.kmers_ms)


class Engine:
    def __init__(self, params=None, device=0):
        self.lib = load_library()
        arr, self.params = params_array(params)
        h = C.c_void_p()
        rc = self.lib.lzani_create(arr, int(device), C.byref(h))
        if rc != 0:
            raise LzaniError(f"lzani_create failed: {ERRORS.get(rc, rc)}")
        self.h = h
        self.n = 0
        self.lens = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.lzani_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.lzani_last_error(self.h)
            raise LzaniError(f"{what}: {ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")

    def set_genomes(self, seqs):
        seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        ptrs = (C.c_void_p * len(seqs))(*[s.ctypes.data for s in seqs])
        lens = np.array([len(s) for s in seqs], dtype=np.uint32)
        self._check(self.lib.lzani_set_genomes(self.h, len(seqs), ptrs, _ptr(lens)), "lzani_set_genomes")
        self.n = len(seqs)
        self.lens = lens

    def run_rows(self, ref_ids, row_off, query_ids=None):
        ref_ids = np.ascontiguousarray(ref_ids, dtype=np.uint32)
        row_off = np.ascontiguousarray(row_off, dtype=np.uint64)
        q = None if query_ids is None else np.ascontiguousarray(query_ids, dtype=np.uint32)
        n_pairs = int(row_off[-1]) if len(row_off) else 0
        out = np.zeros((n_pairs, 3), dtype=np.int32)
        self._check(self.lib.lzani_run_rows(self.h, len(ref_ids), _ptr(ref_ids), _ptr(row_off), _ptr(q), _ptr(out)),
                    "lzani_run_rows")
        return out

    def run_rows_device(self, ref_ids, row_off, query_ids, d_out_ptr):
        """Results stay on the GPU at raw device pointer d_out_ptr (3 int32 per pair)."""
        ref_ids = np.ascontiguousarray(ref_ids, dtype=np.uint32)
        row_off = np.ascontiguousarray(row_off, dtype=np.uint64)
        q = None if query_ids is None else np.ascontiguousarray(query_ids, dtype=np.uint32)
        self._check(self.lib.lzani_run_rows_device(self.h, len(ref_ids), _ptr(ref_ids), _ptr(row_off), _ptr(q),
                                                   C.c_void_p(int(d_out_ptr))), "lzani_run_rows_device")

    REGION_DTYPE = np.dtype([("pair", np.uint64), ("ref_start", np.int32), ("ref_end", np.int32), ("seq_start", np.int32),
                             ("seq_end", np.int32), ("num_matches", np.int32), ("num_mismatches", np.int32)])

    def run_rows_regions(self, ref_ids, row_off, query_ids=None, capacity=1 << 16):
        """(results int32[n_pairs, 3], regions structured array sorted by (pair, length desc, seq_start))."""
        ref_ids = np.ascontiguousarray(ref_ids, dtype=np.uint32)
        row_off = np.ascontiguousarray(row_off, dtype=np.uint64)
        q = None if query_ids is None else np.ascontiguousarray(query_ids, dtype=np.uint32)
        n_pairs = int(row_off[-1]) if len(row_off) else 0
        while True:
            out = np.zeros((n_pairs, 3), dtype=np.int32)
            regs = np.zeros(capacity, dtype=self.REGION_DTYPE)
            cnt = C.c_uint64(0)
            self._check(self.lib.lzani_run_rows_regions(self.h, len(ref_ids), _ptr(ref_ids), _ptr(row_off), _ptr(q), _ptr(out),
                                                        _ptr(regs), capacity, C.byref(cnt)), "lzani_run_rows_regions")
            if cnt.value <= capacity:
                break
            capacity = int(cnt.value)
        regs = regs[:cnt.value]
        order = np.lexsort((regs["seq_start"], -(regs["seq_end"] - regs["seq_start"]), regs["pair"]))
        return out, regs[order]

    def all2all(self):
        """int32[n, n, 3]: out[r, q] = parse(query=q, ref=r), diagonal zero."""
        n = self.n
        ref_ids, row_off = dense_rows(n)
        flat = self.run_rows(ref_ids, row_off, None)
        out = np.zeros((n, n, 3), dtype=np.int32)
        out[~np.eye(n, dtype=bool)] = flat
        return out

    def comm_init(self, n_ranks, rank, unique_id):
        """RCCL communicator of this context (one process per GPU); unique_id = comm_unique_id() of rank 0."""
        buf = (C.c_uint8 * 128)(*unique_id)
        self._check(self.lib.lzani_comm_init(self.h, n_ranks, rank, buf), "lzani_comm_init")

    def comm_allgather(self, d_send_ptr, d_recv_ptr, n_results):
        self._check(self.lib.lzani_comm_allgather(self.h, C.c_void_p(int(d_send_ptr)), C.c_void_p(int(d_recv_ptr)), n_results),
                    "lzani_comm_allgather")

    def comm_gatherv(self, d_send_ptr, d_recv_ptr, counts, root=0):
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        self._check(self.lib.lzani_comm_gatherv(self.h, C.c_void_p(int(d_send_ptr)), C.c_void_p(int(d_recv_ptr or 0)),
                                                _ptr(counts), root), "lzani_comm_gatherv")

    def timing(self):
        t = Timing()
        self._check(self.lib.lzani_get_timing(self.h, C.byref(t)), "lzani_get_timing")
        return dict(index_ms=t.index_ms, pairs_ms=t.pairs_ms, pair_launches=t.pair_launches,
                    index_launches=t.index_launches, pairs=t.pairs, cand_ms=t.cand_ms, kmers_ms=t.kmers_ms,
                    cand_launches=t.cand_launches)

    def layout(self):
        o = LayoutInfo()
        self._check(self.lib.lzani_get_layout(self.h, C.byref(o)), "lzani_get_layout")
        return {k: getattr(o, k) for k, _ in LayoutInfo._fields_}

    def rtc_info(self):
        o = RtcInfo()
        self._check(self.lib.lzani_get_rtc_info(self.h, C.byref(o)), "lzani_get_rtc_info")
        return {k: getattr(o, k) for k, _ in RtcInfo._fields_ if k != "reserved_"}

    def debug_sort_segments(self, keys, seg_len, n_seg, begin_bit, end_bit):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        assert keys.size == seg_len * n_seg
        out = np.zeros_like(keys)
        self._check(self.lib.lzani_debug_sort_segments(self.h, _ptr(keys), _ptr(out), C.c_uint64(seg_len), C.c_uint32(n_seg),
                                                       C.c_int(begin_bit), C.c_int(end_bit)), "debug_sort_segments")
        return out

    def debug_index(self, gid):
        mrd = self.params["mrd"]
        T = 2 * int(self.lens[gid]) + 3 * mrd
        wn = (T + 63) // 64 + 2
        geom = np.zeros(4, dtype=np.uint32)
        self._check(self.lib.lzani_debug_get_index(self.h, gid, None, None, None, None, None, _ptr(geom)), "debug")
        nm = np.zeros(wn, dtype=np.uint64)
        t2 = np.zeros(2 * wn, dtype=np.uint64)
        dirz = np.zeros((1 << int(geom[1])) + 1, dtype=np.uint32)
        ent = np.zeros(T + 1, dtype=np.uint32)
        n_ent = C.c_uint32(0)
        self._check(self.lib.lzani_debug_get_index(self.h, gid, _ptr(t2), _ptr(nm), _ptr(dirz), _ptr(ent),
                                                   C.byref(n_ent), _ptr(geom)), "debug")
        return dict(t2=t2, nm=nm, dirz=dirz, ent=ent[:n_ent.value].copy(), geom=geom)
