"""Builds libvtmhip.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

    python -m vtm_amd.build [--force]

Each csrc/*.hip is compiled to _obj/<name>.o (in parallel, only when stale) and linked into vtm_amd/libvtmhip.so.
The shared object is git-ignored but travels to the GPU box with the repo snapshot."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libvtmhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off", "-Wall",
          "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers_mtime():
    hs = glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "vtmhip.h")]
    return max(os.path.getmtime(h) for h in hs)


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(src, hdr_t, force):
    o = _obj(src)
    return force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), hdr_t)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = _headers_mtime()
    todo = [s for s in sources() if _stale(s, hdr_t, force)]

    def cc(src):
        cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(cc, todo))
    objs = [_obj(s) for s in sources()]
    if todo or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
