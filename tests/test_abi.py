"""CPU: libvtmhip.so loads without a GPU and exports every symbol include/vtmhip.h declares; the Python struct
mirrors have the library's sizeof(); host-only helpers (no device needed) agree with the golden data."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vtm_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "vtmhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vtmhip_[A-Za-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    L = lib.load()   # raises ImportError if the library is missing or a struct layout differs
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), "libvtmhip.so does not export %s" % s
    assert set(syms) == set(lib.exported_symbols()), set(syms) ^ set(lib.exported_symbols())
    assert L.vtmhip_abi_version() == 6


def test_no_device_is_reported_not_crashed():
    L = lib.load()
    n = C.c_int(-1)
    assert L.vtmhip_device_count(C.byref(n)) == lib.OK
    if n.value == 0:
        h = C.c_void_p()
        assert L.vtmhip_create(0, C.byref(h)) == lib.E_NODEVICE
        with pytest.raises(lib.VtmHipError):
            from vtm_amd.device import Context
            Context(0)   # the product path fails loudly without a GPU: there is no CPU fallback


def test_host_transform_matrices_match_reference_golden():
    L = lib.load()
    z = np.load(os.path.join(ROOT, "tests", "golden", "transform.npz"))
    for t in range(3):
        for n in (2, 4, 8, 16, 32, 64):
            m = np.zeros((n, n), np.int16)
            st = L.vtmhip_tr_matrix_host(t, n, m.ctypes.data)
            key = "m_%d_%d" % (t, n)
            assert (st == lib.OK) == (key in z.files)
            if st == lib.OK:
                assert np.array_equal(m, z[key]), key


def test_mts_select_thresholds():
    """TrQuant::transformNxN(..., trModes, maxCand) threshold rule (TrQuant.cpp:1005-1018) restated independently here."""
    L = lib.load()
    rng = np.random.default_rng(3)
    fac = [1.2, 1.3, 1.3, 1.4, 1.5]
    for _ in range(300):
        nc = int(rng.integers(1, 7))
        w, h = int(rng.choice([4, 8, 16, 32])), int(rng.choice([4, 8, 16, 32]))
        max_cand = int(rng.integers(0, 5))
        sums = rng.integers(0, 100000, nc).astype(np.int32)
        test = np.zeros(nc, np.uint8)
        assert L.vtmhip_mts_select(sums.ctypes.data, nc, w, h, max_cand, test.ctypes.data) == lib.OK
        thr = fac[max(0, int(np.log2(max(w, h))) - 2)] * float(sums[0])
        num, exp = 0, []
        for i in range(nc):
            t = bool(sums[i] <= (float(sums[0]) if i == 1 else thr)) and num <= max_cand
            exp.append(int(t))
            num += t
        assert list(test) == exp
