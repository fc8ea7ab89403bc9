"""Golden records of InterSearch::predInterSearch taken INSIDE the real encoder (TEST INFRASTRUCTURE).

oracle/ref_shim_pis.hpp in record mode (VTMREF_PIS_DUMP) writes, for sampled CUs of a real encode, everything the translational part of the member read
(the PU's real AMVP lists, m_uniMvList, block-vector cache hits, search ranges, lambda ...) in the layout of the C ABI's own job tables, and what the reference's
members returned (chosen predictors, template costs, every row's search result, the bi rows, the SMVD block's states, what the member left in `pu`).
This module parses such a dump (gen_pis_golden.py -> tests/golden/pis_enc.npz) and rebuilds the tables for vtmhip_predInterSearch_batch_dev / the oracle."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from vtm_amd.lib import MAX_REF, MeJob, MeOut, PisPu, PisPuIn, PisRow, PredJob, SmvdJob   # noqa: E402

ROWS = 2 * MAX_REF
PIS_MAGIC, PLANE_MAGIC = 0x50495331, 0x50495332


class PisSlots(C.Structure):      # oracle/ref_shim_pis.hpp:PisSlots without the trailing original block
    _fields_ = [("uniJobs", MeJob * ROWS), ("uniOut", MeOut * ROWS), ("uniRows", PisRow * ROWS), ("distBiP", C.c_uint64 * ROWS), ("pus", PisPu * 1), ("puIn", PisPuIn * 1),
                ("predOther", PredJob * 1), ("biJobs", MeJob * MAX_REF), ("biOut", MeOut * MAX_REF), ("biRows", PisRow * MAX_REF), ("smvd", SmvdJob * 1), ("pos", C.c_int64 * 1)]


class PisFinal(C.Structure):
    _fields_ = [("ran", C.c_int32), ("affine", C.c_int32), ("interDir", C.c_int32), ("smvdMode", C.c_int32), ("refIdx", C.c_int32 * 2), ("mv", (C.c_int32 * 2) * 2),
                ("mvd", (C.c_int32 * 2) * 2), ("mvpIdx", C.c_int32 * 2), ("mvpNum", C.c_int32 * 2), ("refIdxBi", C.c_int32 * 2), ("biList", C.c_int32), ("smvdRan", C.c_int32),
                ("hevcCost", C.c_uint64)]


class PisHeader(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("bytes", C.c_uint32), ("poc", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("imv", C.c_int32),
                ("picW", C.c_int32), ("picH", C.c_int32), ("ctuSize", C.c_int32), ("bitDepth", C.c_int32), ("numRef", C.c_int32 * 2), ("smvdBit", C.c_int32),
                ("symRefIdx", C.c_int32 * 2), ("hasSmvd", C.c_int32), ("biRestricted", C.c_int32), ("mvdL1Zero", C.c_int32), ("fdm", C.c_int32), ("list1FromList0", C.c_int32 * MAX_REF),
                ("mbBits", C.c_uint32 * 3), ("bipredSearchRange", C.c_int32), ("useHadME", C.c_int32), ("fen13", C.c_int32), ("extendedSettings", C.c_int32),
                ("firstSearchStop", C.c_int32), ("uniMvListSize", C.c_int32), ("rowPlane", C.c_int32 * ROWS), ("rowOff", C.c_int64 * ROWS), ("rowCached", C.c_int32 * ROWS),
                ("rowCalls", C.c_int32 * ROWS),
                # appended in round 4 (records of round 3 end above: load_npz pads them with zeros)
                ("refPoc", (C.c_int32 * MAX_REF) * 2), ("curPoc", C.c_int32), ("givenRows", C.c_int32), ("bcwIdx", C.c_int32), ("bcwNoBi", C.c_int32)]


COST_UNKNOWN = 2 ** 64 - 2      # PisFinal.hevcCost when the member did not store its translational cost (oracle/ref_shim_pis.hpp: PIS_COST_UNKNOWN)


class PlaneHeader(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("bytes", C.c_uint32), ("poc", C.c_int32), ("stride", C.c_int32), ("margin", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("index", C.c_int32)]


def parse_dump(path):
    """-> (planes: [(PlaneHeader, int16 array)], records: [(PisHeader, PisSlots in, PisSlots out, org int16 [h, w], PisFinal)])"""
    buf = open(path, "rb").read()
    planes, recs, o = [], [], 0
    while o < len(buf):
        magic = int.from_bytes(buf[o:o + 4], "little")
        if magic == PLANE_MAGIC:
            ph = PlaneHeader.from_buffer_copy(buf[o:o + C.sizeof(PlaneHeader)])
            o += C.sizeof(PlaneHeader)
            planes.append((ph, np.frombuffer(buf[o:o + ph.bytes], np.int16).copy()))
            o += ph.bytes
        else:
            assert magic == PIS_MAGIC, hex(magic)
            hd = PisHeader.from_buffer_copy(buf[o:o + C.sizeof(PisHeader)])
            p = o + C.sizeof(PisHeader)
            sin = PisSlots.from_buffer_copy(buf[p:p + C.sizeof(PisSlots)]); p += C.sizeof(PisSlots)
            sout = PisSlots.from_buffer_copy(buf[p:p + C.sizeof(PisSlots)]); p += C.sizeof(PisSlots)
            org = np.frombuffer(buf[p:p + 2 * hd.w * hd.h], np.int16).reshape(hd.h, hd.w).copy(); p += 2 * hd.w * hd.h
            fin = PisFinal.from_buffer_copy(buf[p:p + C.sizeof(PisFinal)]); p += C.sizeof(PisFinal)
            assert p - o == hd.bytes, (p - o, hd.bytes)
            recs.append((hd, sin, sout, org, fin))
            o = p
    return planes, recs


def save_npz(path, planes, recs):
    """data only: plane samples + headers, and per record the raw bytes of the five parts"""
    d = {"plane_hdr": np.frombuffer(b"".join(bytes(p[0]) for p in planes), np.uint8), "num_planes": np.array([len(planes)]),
         "rec_hdr": np.frombuffer(b"".join(bytes(r[0]) for r in recs), np.uint8), "rec_in": np.frombuffer(b"".join(bytes(r[1]) for r in recs), np.uint8),
         "rec_out": np.frombuffer(b"".join(bytes(r[2]) for r in recs), np.uint8), "rec_fin": np.frombuffer(b"".join(bytes(r[4]) for r in recs), np.uint8),
         "org": np.concatenate([r[3].reshape(-1) for r in recs]), "num_recs": np.array([len(recs)])}
    for i, (ph, a) in enumerate(planes):
        d["plane%d" % i] = a
    np.savez_compressed(path, **d)


def load_npz(path):
    z = np.load(path)
    n, m = int(z["num_recs"][0]), int(z["num_planes"][0])
    planes = [(PlaneHeader.from_buffer_copy(z["plane_hdr"].tobytes()[i * C.sizeof(PlaneHeader):(i + 1) * C.sizeof(PlaneHeader)]), z["plane%d" % i]) for i in range(m)]
    recs, o = [], 0
    hb, ib, ob, fb, org = z["rec_hdr"].tobytes(), z["rec_in"].tobytes(), z["rec_out"].tobytes(), z["rec_fin"].tobytes(), z["org"]
    hsz = len(hb) // max(1, n)      # the header grew at its end in round 4: older files hold the shorter form
    assert hsz <= C.sizeof(PisHeader) and len(ib) == n * C.sizeof(PisSlots), "golden file of another ABI"
    for i in range(n):
        hd = PisHeader.from_buffer_copy(hb[i * hsz:(i + 1) * hsz].ljust(C.sizeof(PisHeader), b"\0"))
        sin = PisSlots.from_buffer_copy(ib[i * C.sizeof(PisSlots):(i + 1) * C.sizeof(PisSlots)])
        sout = PisSlots.from_buffer_copy(ob[i * C.sizeof(PisSlots):(i + 1) * C.sizeof(PisSlots)])
        fin = PisFinal.from_buffer_copy(fb[i * C.sizeof(PisFinal):(i + 1) * C.sizeof(PisFinal)])
        recs.append((hd, sin, sout, org[o:o + hd.w * hd.h].reshape(hd.h, hd.w), fin))
        o += hd.w * hd.h
    return planes, recs


PAD = 4096      # samples of slack around every plane of the rebuilt DPB (>= VTMHIP_PLANE_SLACK of include/vtmhip.h: the kernels fetch whole 16-byte groups / window rows)


def build_dpb(planes):
    """-> (int16 array holding all planes, [sample offset of each plane's first dumped sample])"""
    parts, bases, acc = [np.zeros(PAD, np.int16)], [], PAD
    for ph, a in planes:
        bases.append(acc)
        parts += [a, np.zeros(PAD, np.int16)]
        acc += a.size + PAD
    return np.concatenate(parts), bases
