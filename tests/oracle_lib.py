"""ctypes bindings for the CPU oracle (oracle/libvtmoracle.so) and, when it has been built in this
container, the real reference (oracle/_ref/libvtmref.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the vtm_amd product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libvtmoracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libvtmref.so")

DCT2, DCT8, DST7 = 0, 1, 2


def build_oracle(force=False):
    src = os.path.join(ROOT, "oracle", "vtm_oracle.c")
    hdr = os.path.join(ROOT, "oracle", "vtm_oracle.h")
    if (not force and os.path.exists(ORACLE_SO)
            and os.path.getmtime(ORACLE_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return ORACLE_SO
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", ORACLE_SO, src, "-lm"])
    return ORACLE_SO


def P(a):
    return a.ctypes.data_as(C.c_void_p)


class MvCost(C.Structure):
    _fields_ = [("motionLambda", C.c_double), ("predHor", C.c_int), ("predVer", C.c_int), ("costScale", C.c_int)]


class Range(C.Structure):
    _fields_ = [("left", C.c_int), ("right", C.c_int), ("top", C.c_int), ("bottom", C.c_int)]


class MeCtx(C.Structure):
    _fields_ = [("org", C.c_void_p), ("orgStride", C.c_int), ("ref", C.c_void_p), ("refStride", C.c_int),
                ("w", C.c_int), ("h", C.c_int), ("subShift", C.c_int), ("bitDepth", C.c_int), ("imvShift", C.c_uint),
                ("mv", MvCost), ("picW", C.c_int), ("picH", C.c_int), ("puX", C.c_int), ("puY", C.c_int),
                ("ctuSize", C.c_int)]


class TzJob(C.Structure):
    _fields_ = [("mvHor", C.c_int), ("mvVer", C.c_int), ("searchRange", C.c_int), ("extendedSettings", C.c_int),
                ("fastSettings", C.c_int), ("firstSearchStop", C.c_int), ("hasIntMv2Nx2NPred", C.c_int),
                ("intMv2Nx2NPredHor", C.c_int), ("intMv2Nx2NPredVer", C.c_int), ("numExtraStart", C.c_int),
                ("extraStart", (C.c_int * 2) * 16)]


class MeResult(C.Structure):
    _fields_ = [("mvX", C.c_int), ("mvY", C.c_int), ("cost", C.c_uint64), ("dist", C.c_uint64), ("nEval", C.c_uint64)]


class FracResult(C.Structure):
    _fields_ = [("halfX", C.c_int), ("halfY", C.c_int), ("qterX", C.c_int), ("qterY", C.c_int),
                ("costHalf", C.c_uint64), ("cost", C.c_uint64), ("candHalf", C.c_uint64 * 9),
                ("candQuarter", C.c_uint64 * 9)]


class MestCfg(C.Structure):
    _fields_ = [("bipredSearchRange", C.c_int), ("useHadME", C.c_int), ("fastInterSearchMode13", C.c_int), ("extendedSettings", C.c_int),
                ("firstSearchStop", C.c_int)]


class MestJob(C.Structure):
    _fields_ = [("org", C.c_void_p), ("orgStride", C.c_int), ("ref", C.c_void_p), ("refStride", C.c_int), ("otherPred", C.c_void_p),
                ("otherStride", C.c_int), ("w", C.c_int), ("h", C.c_int), ("puX", C.c_int), ("puY", C.c_int), ("picW", C.c_int),
                ("picH", C.c_int), ("ctuSize", C.c_int), ("bitDepth", C.c_int), ("bi", C.c_int), ("imv", C.c_int), ("mvpIdx", C.c_int),
                ("numAmvpCand", C.c_int), ("mvPredHor", C.c_int), ("mvPredVer", C.c_int), ("mvHor", C.c_int), ("mvVer", C.c_int),
                ("amvpCand", (C.c_int * 2) * 2), ("mvpIdxBits", C.c_uint * 2), ("bits", C.c_uint), ("searchRange", C.c_int),
                ("motionLambda", C.c_double), ("numExtraStart", C.c_int), ("extraStart", (C.c_int * 2) * 16), ("cachedIntMv", C.c_int), ("bcwWeight", C.c_int)]


class MestResult(C.Structure):
    _fields_ = [("mvHor", C.c_int), ("mvVer", C.c_int), ("mvPredHor", C.c_int), ("mvPredVer", C.c_int), ("mvpIdx", C.c_int),
                ("bits", C.c_uint), ("cost", C.c_uint64), ("intX", C.c_int), ("intY", C.c_int), ("intDist", C.c_uint64)]

    def key(self):
        return (self.mvHor, self.mvVer, self.mvPredHor, self.mvPredVer, self.mvpIdx, self.bits, self.cost)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        L = C.CDLL(build_oracle())
        for n in ("vo_sad", "vo_sse", "vo_satd", "vo_mv_cost", "vo_symmetric_cost"):
            getattr(L, n).restype = C.c_uint64
        L.vo_mv_bits.restype = C.c_uint
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_dist.restype = C.c_uint64
        L.ref_mv_cost.restype = C.c_uint64
        if hasattr(L, "ref_symmetric_cost"):
            L.ref_symmetric_cost.restype = C.c_uint64
        L.ref_mv_cost.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]
        _ref = L
    return _ref


# ---- thin numpy helpers (shared by the oracle tests and the GPU parity tests) -------------------------------
def i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


def o_dist(kind, org, cur, w, h, sub_shift=0, org_off=0, cur_off=0):
    """kind 0 SAD / 1 SATD / 2 SSE on 2-D int16 arrays (stride = row length)."""
    L = oracle()
    os_, cs = org.shape[1], cur.shape[1]
    po = org.ctypes.data + 2 * org_off
    pc = cur.ctypes.data + 2 * cur_off
    if kind == 0:
        return L.vo_sad(C.c_void_p(po), os_, C.c_void_p(pc), cs, w, h, sub_shift)
    if kind == 1:
        return L.vo_satd(C.c_void_p(po), os_, C.c_void_p(pc), cs, w, h)
    return L.vo_sse(C.c_void_p(po), os_, C.c_void_p(pc), cs, w, h)


def r_dist(kind, simd, org, cur, w, h, bit_depth=10, sub_shift=0, org_off=0, cur_off=0):
    L = ref()
    po = org.ctypes.data + 2 * org_off
    pc = cur.ctypes.data + 2 * cur_off
    return L.ref_dist(kind, simd, C.c_void_p(po), org.shape[1], C.c_void_p(pc), cur.shape[1], w, h, bit_depth, sub_shift)


class AffinePred(C.Structure):
    _fields_ = [("ref", C.c_void_p), ("refStride", C.c_int), ("w", C.c_int), ("h", C.c_int), ("puX", C.c_int), ("puY", C.c_int), ("picW", C.c_int),
                ("picH", C.c_int), ("ctuSize", C.c_int), ("bitDepth", C.c_int), ("sixParam", C.c_int), ("interDir", C.c_int), ("profAllowed", C.c_int),
                ("profNeedsLargeGrad", C.c_int), ("profIsBi", C.c_int)]


class AffineMeJob(C.Structure):
    _fields_ = [("pred", AffinePred), ("org", C.c_void_p), ("orgStride", C.c_int), ("otherPred", C.c_void_p), ("otherStride", C.c_int), ("bi", C.c_int),
                ("imv", C.c_int), ("useSatd", C.c_int), ("useAffineType", C.c_int), ("amvrEncOpt", C.c_int), ("lowDelayRounds", C.c_int),
                ("mvPred", (C.c_int * 2) * 3), ("mv", (C.c_int * 2) * 3), ("bits", C.c_uint), ("motionLambda", C.c_double), ("hevcCost", C.c_uint64)]


class SmvdJob(C.Structure):      # vo_smvd_job_t
    _fields_ = [("org", C.c_void_p), ("orgStride", C.c_int), ("ref", C.c_void_p * 2), ("refStride", C.c_int * 2), ("w", C.c_int), ("h", C.c_int), ("puX", C.c_int),
                ("puY", C.c_int), ("picW", C.c_int), ("picH", C.c_int), ("ctuSize", C.c_int), ("bitDepth", C.c_int), ("imv", C.c_int), ("useSatd", C.c_int),
                ("clipBiPred", C.c_int), ("bcwWeightTar", C.c_int), ("numCand", C.c_int * 2), ("cand", ((C.c_int * 2) * 2) * 2), ("mvpIdxBits", C.c_uint * 2),
                ("motionLambda", C.c_double)]


class SmvdResult(C.Structure):   # vo_smvd_result_t
    _fields_ = [("mvCur", C.c_int * 2), ("mvTar", C.c_int * 2), ("predSym", (C.c_int * 2) * 2), ("mvpIdxSym", C.c_int * 2), ("cost", C.c_uint64)]

    def key(self):
        return (tuple(self.mvCur), tuple(self.mvTar), tuple(self.predSym[0]), tuple(self.predSym[1]), tuple(self.mvpIdxSym), self.cost)


class AffineMeResult(C.Structure):
    _fields_ = [("mv", (C.c_int * 2) * 3), ("bits", C.c_uint), ("cost", C.c_uint64), ("iterations", C.c_int), ("refinements", C.c_int)]
