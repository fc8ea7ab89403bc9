"""Encoder-level drop-in helper (TEST INFRASTRUCTURE): runs the real reference encoder of oracle/_ref/libvtmref.so on a small
synthetic clip in a child process -- plain, or with its dispatch tables routed to libvtmhip.so (oracle/ref_shim_enc.cpp)."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libvtmref.so")
REF_SP_SO = os.path.join(ROOT, "oracle", "_ref", "libvtmref_sp.so")      # the same sources with ENABLE_SPLIT_PARALLELISM (oracle/Makefile.ref SP=1): takes --NumSplitThreads
HIP_SO = os.path.join(ROOT, "vtm_amd", "libvtmhip.so")
CFG = os.path.join(ROOT, "tests", "data", "enc_ra_gop4.cfg")


class Stats(C.Structure):
    _fields_ = [("calls", C.c_uint64 * 4), ("device", C.c_uint64 * 4), ("mismatch", C.c_uint64 * 4), ("errors", C.c_uint64),
                ("firstMismatch", C.c_int32 * 8), ("firstError", C.c_char * 160),
                ("hookCalls", C.c_uint64 * 2), ("hookDevice", C.c_uint64 * 2), ("hookMismatch", C.c_uint64 * 2), ("hookUnsupported", C.c_uint64 * 2),
                ("hookFirstMismatch", C.c_int32 * 8), ("affineCalls", C.c_uint64), ("affineDevice", C.c_uint64), ("affineMismatch", C.c_uint64),
                ("affineUnsupported", C.c_uint64), ("lfnstCalls", C.c_uint64 * 2), ("lfnstDevice", C.c_uint64 * 2), ("lfnstMismatch", C.c_uint64 * 2),
                ("amvpCalls", C.c_uint64), ("amvpDevice", C.c_uint64), ("amvpMismatch", C.c_uint64), ("amvpUnsupported", C.c_uint64),
                ("smvdCalls", C.c_uint64 * 3), ("smvdDevice", C.c_uint64 * 3), ("smvdMismatch", C.c_uint64 * 3), ("smvdUnsupported", C.c_uint64),
                ("pisCalls", C.c_uint64), ("pisDevice", C.c_uint64), ("pisUnsupported", C.c_uint64), ("pisSkipped", C.c_uint64), ("pisReplayFallback", C.c_uint64),
                ("pisMismatch", C.c_uint64 * 6), ("pisFirstMismatch", C.c_int32 * 8), ("pisNs", C.c_uint64 * 4), ("affineNs", C.c_uint64 * 2),
                ("intraBatches", C.c_uint64 * 4), ("intraServed", C.c_uint64), ("intraMismatch", C.c_uint64), ("intraFirstMismatch", C.c_int32 * 8), ("hookThreads", C.c_uint64)]


def write_clip(path, w, h, frames, seed=77):
    sys.path.insert(0, ROOT)
    from vtm_amd import synth
    with open(path, "wb") as f:
        for y, u, v in synth.gen_frames(w, h, frames, seed=seed, chroma=True):
            for p in (y, u, v):
                f.write(np.ascontiguousarray(p).astype("<u2").tobytes())


def _child(argv_json):
    a = json.loads(argv_json)
    lib = C.CDLL(a.get("ref_so") or REF_SO)
    lib.ref_encode.restype = C.c_int
    lib.ref_encode.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_uint, C.c_uint64, C.c_uint64, C.POINTER(Stats)]
    args = [b"EncoderApp"] + [s.encode() for s in a["args"]]
    arr = (C.c_char_p * len(args))(*args)
    st = Stats()
    rc = lib.ref_encode(len(args), arr, a["hip"].encode() if a["hip"] else None, a["mask"], a["stride"], a["head"], C.byref(st))
    out = {"rc": rc, "calls": list(st.calls), "device": list(st.device), "mismatch": list(st.mismatch), "errors": st.errors,
           "firstMismatch": list(st.firstMismatch), "firstError": st.firstError.decode(errors="replace"),
           "hookCalls": list(st.hookCalls), "hookDevice": list(st.hookDevice), "hookMismatch": list(st.hookMismatch), "hookUnsupported": list(st.hookUnsupported),
           "hookFirstMismatch": list(st.hookFirstMismatch), "affine": [st.affineCalls, st.affineDevice, st.affineMismatch, st.affineUnsupported],
           "lfnst": [list(st.lfnstCalls), list(st.lfnstDevice), list(st.lfnstMismatch)],
           "amvp": [st.amvpCalls, st.amvpDevice, st.amvpMismatch, st.amvpUnsupported],
           "smvd": [list(st.smvdCalls), list(st.smvdDevice), list(st.smvdMismatch), st.smvdUnsupported],
           "pis": {"calls": st.pisCalls, "device": st.pisDevice, "unsupported": st.pisUnsupported, "skipped": st.pisSkipped, "replayFallback": st.pisReplayFallback,
                   "mismatch": list(st.pisMismatch), "firstMismatch": list(st.pisFirstMismatch), "seconds": [v / 1e9 for v in st.pisNs]},
           "affineSeconds": [v / 1e9 for v in st.affineNs],
           "hookThreads": st.hookThreads,
           "intra": {"batches": list(st.intraBatches), "served": st.intraServed, "mismatch": st.intraMismatch, "firstMismatch": list(st.intraFirstMismatch)}}
    sys.stdout.flush()
    os.write(2, ("\nDROPIN_RESULT " + json.dumps(out) + "\n").encode())


def encode(yuv, w, h, frames, qp, out_prefix, hip=False, mask=23, stride=1, head=0, extra=(), timeout=1500, env=None, cfg=None, ref_so=None):
    """Returns (stats dict, md5 of the bitstream, md5 of the reconstruction).  mask bit 2048: InterSearch::predInterSearch as one device call per CU
    (oracle/ref_shim_pis.hpp; env VTMREF_REPLACE=1: replace mode, VTMREF_PIS_DUMP=<file>: record mode without a device)."""
    bits, rec = out_prefix + ".bin", out_prefix + "_rec.yuv"
    args = ["-c", cfg or CFG, "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-f", str(frames), "-q", str(qp), "-b", bits, "-o", rec,
            "--SEIDecodedPictureHash=1", "--OutputBitDepth=10"] + list(extra)
    req = json.dumps({"args": args, "hip": HIP_SO if hip else "", "mask": mask, "stride": stride, "head": head, "ref_so": ref_so})
    p = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import enc_dropin; enc_dropin._child(sys.argv[1])"
                        % os.path.dirname(os.path.abspath(__file__)), req], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    err = p.stderr.decode(errors="replace")
    line = [l for l in err.splitlines() if l.startswith("DROPIN_RESULT ")]
    if p.returncode != 0 or not line:
        raise RuntimeError("reference encoder run failed (rc %d):\n%s\n%s" % (p.returncode, p.stdout.decode(errors="replace")[-1500:], err[-3000:]))
    st = json.loads(line[-1][len("DROPIN_RESULT "):])
    st["log_tail"] = p.stdout.decode(errors="replace")[-1200:]
    md5 = lambda f: hashlib.md5(open(f, "rb").read()).hexdigest()
    return st, md5(bits), md5(rec)
