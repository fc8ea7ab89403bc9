import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    # VTM_TEST_SEED_OFFSET=N: every np.random.default_rng( seed ) of the tests draws from seed + N instead -- a soak over other job sets / sample data with the same
    # test code (the expected values come from the oracle inside each test; tests bound to golden files or to counts of a particular draw are left out by -k)
    off = int(os.environ.get("VTM_TEST_SEED_OFFSET", "0"))
    if off:
        import numpy as np
        orig = np.random.default_rng
        np.random.default_rng = lambda seed=None, _o=orig: _o(seed if seed is None or not isinstance(seed, (int, np.integer)) else int(seed) + off)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libvtmref.so (the real reference, built in the dev container)")


def _gpu_available():
    try:
        import ctypes
        from vtm_amd import lib
        L = lib.load()
        n = ctypes.c_int(0)
        L.vtmhip_device_count(ctypes.byref(n))
        return n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def ctx():
    from vtm_amd.device import Context
    c = Context(0)   # raises if the HIP library or the device is missing: GPU tests must not pass on a fallback
    yield c
    c.close()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.oracle()


@pytest.fixture(scope="session")
def reflib():
    import oracle_lib
    if not oracle_lib.have_ref():
        pytest.skip("oracle/_ref/libvtmref.so not built (only available in the dev container)")
    return oracle_lib.ref()
