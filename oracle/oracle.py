"""ctypes bindings for the parity oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It wraps

* ``liblzani_oracle.so``  -- the plain-C restatement (oracle/lzani_oracle.c), and
* ``_ref/libref_lzani.so`` -- the reference's own CParser behind oracle/ref_driver.cpp
  (present only if ``make -C oracle ref`` ran in a container that has /root/reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PARAMS = dict(mal=11, msl=7, mrd=40, mqd=40, reg=35, aw=15, am=7, ar=3)
PARAM_ORDER = ("mal", "msl", "mrd", "mqd", "reg", "aw", "am", "ar")

_CODE = np.full(256, 5, dtype=np.uint8)
for _i, _ch in enumerate("ACGT"):
    _CODE[ord(_ch)] = _i
    _CODE[ord(_ch.lower())] = _i


def encode(seq):
    """ASCII bases -> reservoir symbol codes (seq_reservoir.h:241-248): ACGT/acgt -> 0..3, else 5."""
    if isinstance(seq, str):
        seq = seq.encode()
    return _CODE[np.frombuffer(seq, dtype=np.uint8)]


def read_multifasta(path):
    """[(name, codes)] with load_multifasta's observable quirks (seq_reservoir.cpp:156-212):
    name cut at the first space, \\r stripped, a final line without newline dropped."""
    import gzip
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        data = f.read()
    lines = data.split(b"\n")
    lines = lines[:-1]  # whatever follows the last newline is never returned by getline
    out, name, chunks = [], None, []
    for ln in lines:
        ln = ln.replace(b"\r", b"")
        if not ln:
            continue
        if ln[:1] == b">":
            if name:
                out.append((name.split(b" ")[0].decode(), encode(b"".join(chunks))))
            name, chunks = ln[1:], []
        else:
            chunks.append(ln)
    if name:
        out.append((name.split(b" ")[0].decode(), encode(b"".join(chunks))))
    return out


def params_array(params=None):
    p = dict(DEFAULT_PARAMS)
    if params:
        p.update(params)
    return (C.c_int32 * 8)(*[int(p[k]) for k in PARAM_ORDER])


def build(ref=True):
    """Compile the restatement (always) and, where /root/reference exists, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_oracle = None
_ref = None


def lib_oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(HERE, "liblzani_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        lib = C.CDLL(path)
        lib.lzo_prepare_reference.restype = C.c_void_p
        lib.lzo_prepare_reference.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.lzo_free_reference.argtypes = [C.c_void_p]
        lib.lzo_query.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                  C.c_void_p, C.c_uint32, C.c_void_p,
                                  C.c_void_p, C.c_uint32, C.c_void_p]
        lib.lzo_pair.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.lzo_all2all.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _oracle = lib
    return _oracle


def lib_ref():
    """The reference-built library, or None when it has not been built (e.g. fresh checkout)."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libref_lzani.so")
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        lib.ref_pair.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.ref_rows.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        if hasattr(lib, "ref_format_real"):
            lib.ref_format_real.argtypes = [C.c_double, C.c_int, C.c_char_p]
        _ref = lib
    return _ref


def ref_expand_output_format(fmt):
    """Column names the reference derives from an --out-format string ("!token" if it rejects one)."""
    lib = lib_ref()
    assert lib is not None, "oracle/_ref not built"
    buf = C.create_string_buffer(4096)
    lib.ref_expand_output_format.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    n = lib.ref_expand_output_format(fmt.encode(), buf, 4096)
    return buf.raw[:n].decode()


def ref_format_real(v, prec):
    """refresh::real_to_pchar of the reference build (the TSV number formatting)."""
    lib = lib_ref()
    assert lib is not None, "oracle/_ref not built"
    buf = C.create_string_buffer(64)
    n = lib.ref_format_real(float(v), int(prec), buf)
    return buf.raw[:n].decode()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _seq_table(seqs):
    seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
    ptrs = (C.c_void_p * len(seqs))(*[s.ctypes.data for s in seqs])
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    return seqs, ptrs, lens


def oracle_pair(ref, qry, params=None, want_regions=False, want_factors=False):
    """(mat, lit, comp) of parse(query=qry, ref=ref) by the C restatement."""
    lib = lib_oracle()
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    p = params_array(params)
    h = lib.lzo_prepare_reference(_ptr(ref), len(ref), p)
    try:
        res = np.zeros(3, dtype=np.int32)
        nreg = C.c_uint32(0)
        nfac = C.c_uint32(0)
        regs = np.zeros((4096, 6), dtype=np.int32) if want_regions else None
        facs = np.zeros((1 << 18, 4), dtype=np.int32) if want_factors else None
        lib.lzo_query(h, _ptr(qry), len(qry), _ptr(res),
                      _ptr(regs) if want_regions else None, 4096, C.byref(nreg),
                      _ptr(facs) if want_factors else None, 1 << 18, C.byref(nfac))
    finally:
        lib.lzo_free_reference(h)
    out = [tuple(int(x) for x in res)]
    if want_regions:
        out.append(regs[:nreg.value].copy())
    if want_factors:
        out.append(facs[:nfac.value].copy())
    return out[0] if len(out) == 1 else tuple(out)


def oracle_all2all(seqs, params=None, threads=1):
    """Dense all2all by the restatement: int32[n, n, 3], out[r, q] = parse(query=q, ref=r)."""
    lib = lib_oracle()
    seqs, ptrs, lens = _seq_table(seqs)
    n = len(seqs)
    out = np.zeros((n, n, 3), dtype=np.int32)
    lib.lzo_all2all(n, ptrs, _ptr(lens), params_array(params), threads, _ptr(out))
    return out


def ref_pair(ref, qry, params=None, want_regions=False):
    lib = lib_ref()
    assert lib is not None, "oracle/_ref not built"
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    res = np.zeros(3, dtype=np.int32)
    regs = np.zeros((4096, 6), dtype=np.int32)
    nreg = C.c_uint32(0)
    lib.ref_pair(_ptr(ref), len(ref), _ptr(qry), len(qry), params_array(params), _ptr(res),
                 _ptr(regs), 4096, C.byref(nreg) if want_regions else None)
    r = tuple(int(x) for x in res)
    return (r, regs[:nreg.value].copy()) if want_regions else r


def rows_dense(n):
    """CSR rows of the dense all2all: row r = every q != r, ascending."""
    ref_ids = np.arange(n, dtype=np.uint32)
    row_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(max(n - 1, 0))
    q = np.tile(np.arange(n, dtype=np.uint32), (n, 1))
    query_ids = q[~np.eye(n, dtype=bool)].reshape(-1).astype(np.uint32)
    return ref_ids, row_off, query_ids


def ref_rows(seqs, ref_ids, row_off, query_ids, params=None, threads=1):
    """Pairs through the reference's CParser: int32[n_pairs, 3], CSR-aligned."""
    lib = lib_ref()
    assert lib is not None, "oracle/_ref not built"
    seqs, ptrs, lens = _seq_table(seqs)
    ref_ids = np.ascontiguousarray(ref_ids, dtype=np.uint32)
    row_off = np.ascontiguousarray(row_off, dtype=np.uint64)
    query_ids = np.ascontiguousarray(query_ids, dtype=np.uint32)
    out = np.zeros((len(query_ids), 3), dtype=np.int32)
    lib.ref_rows(len(seqs), ptrs, _ptr(lens), params_array(params), len(ref_ids),
                 _ptr(ref_ids), _ptr(row_off), _ptr(query_ids), threads, _ptr(out))
    return out


def ref_all2all(seqs, params=None, threads=1):
    n = len(seqs)
    ref_ids, row_off, query_ids = rows_dense(n)
    flat = ref_rows(seqs, ref_ids, row_off, query_ids, params, threads)
    out = np.zeros((n, n, 3), dtype=np.int32)
    out[~np.eye(n, dtype=bool)] = flat
    return out
