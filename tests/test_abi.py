"""CPU: the C-ABI shared library loads and exports every symbol include/lzani.h declares.
No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

import lzani_ctypes as L
import util as U


@pytest.fixture(scope="module")
def lib():
    L.build_library()
    return L.load_library()


def declared_symbols():
    hdr = open(os.path.join(U.ROOT, "include", "lzani.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lzani_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert set(names) == set(L.EXPORTS), names
    for n in names:
        assert getattr(lib, n) is not None


def test_default_params(lib):
    arr = (C.c_int32 * 8)()
    lib.lzani_default_params(arr)
    assert list(arr) == [11, 7, 40, 40, 35, 15, 7, 3]        # params.h:34-48


def test_create_fails_cleanly_without_gpu_or_bad_params(lib):
    h = C.c_void_p()
    bad = (C.c_int32 * 8)(11, 7, 40, 65, 35, 15, 7, 3)        # mqd outside the envelope
    assert lib.lzani_create(bad, 0, C.byref(h)) == -2
    assert lib.lzani_create(None, 0, C.byref(h)) == -1
    assert lib.lzani_last_error(None) == b"null context"


def test_python_binding_has_no_cpu_fallback(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(L, "_lib", None)
    with pytest.raises(L.LzaniError):
        L.load_library()


def test_run_time_compile_of_the_pair_kernel_without_a_gpu():
    """The pair kernel as hipRTC compiles it for a context's own parameter tuple (lzani_rtc.h): the embedded headers
    assemble into a translation unit that compiles for gfx950 -- inside the null chain's envelope (its constants become
    immediates of the hand-written loop: every one must be encodable), with msl 8, and outside the envelope."""
    import lzani_ctypes as L
    for prm, nfree, cand in ((dict(reg=36), True, 2), (dict(mal=12, msl=8, mrd=50, mqd=30, reg=200, aw=12, am=5, ar=2), False, 0),
                             (dict(aw=20), True, 1)):
        n, log = L.rtc_compile(prm, nfree=nfree, cand=cand)
        assert n > 20000, (prm, n, log[:2000])
    n, _ = L.rtc_compile(dict(mal=20))
    assert n == -2                      # LZANI_ERR_PARAMS: no k-mer words beyond 15, nothing to compile
