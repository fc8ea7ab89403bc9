"""ctypes binding of the C ABI in include/vtmhip.h (libvtmhip.so, built in-tree by vtm_amd.build).

The library is the product: if it is missing this module raises -- there is no CPU fallback."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvtmhip.so")

OK, E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
DIST_SAD, DIST_SATD, DIST_SSE = 0, 1, 2
DCT2, DCT8, DST7 = 0, 1, 2


class DistJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("curOff", C.c_int64), ("orgStride", C.c_int32), ("curStride", C.c_int32),
                ("width", C.c_int16), ("height", C.c_int16), ("subShift", C.c_int16), ("kind", C.c_int16)]


class PicParams(C.Structure):
    _fields_ = [("picW", C.c_int32), ("picH", C.c_int32), ("ctuSize", C.c_int32), ("bitDepth", C.c_int32),
                ("wavesPerJob", C.c_int32), ("maxSearchRange", C.c_int32)]


class TzJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("refOff", C.c_int64), ("orgStride", C.c_int32), ("refStride", C.c_int32),
                ("puX", C.c_int16), ("puY", C.c_int16), ("width", C.c_int16), ("height", C.c_int16),
                ("subShift", C.c_int16), ("imvShift", C.c_uint8), ("signedSamples", C.c_uint8), ("predHor", C.c_int32), ("predVer", C.c_int32),
                ("motionLambda", C.c_double), ("mvHor", C.c_int32), ("mvVer", C.c_int32), ("searchRange", C.c_int32),
                ("extendedSettings", C.c_uint8), ("fastSettings", C.c_uint8), ("firstSearchStop", C.c_uint8),
                ("hasIntMv2Nx2NPred", C.c_uint8), ("intMv2Nx2NPredHor", C.c_int32), ("intMv2Nx2NPredVer", C.c_int32),
                ("numExtraStart", C.c_int32), ("extraStart", (C.c_int32 * 2) * 15)]


class MeResult(C.Structure):
    _fields_ = [("mvX", C.c_int32), ("mvY", C.c_int32), ("nEval", C.c_uint32), ("reserved", C.c_uint32),
                ("cost", C.c_uint64), ("dist", C.c_uint64)]


class IfJob(C.Structure):
    _fields_ = [("srcOff", C.c_int64), ("dstOff", C.c_int64), ("srcStride", C.c_int32), ("dstStride", C.c_int32),
                ("width", C.c_int16), ("height", C.c_int16), ("vertical", C.c_uint8), ("taps", C.c_uint8),
                ("isFirst", C.c_uint8), ("isLast", C.c_uint8), ("coeff", C.c_int16 * 8), ("clipMin", C.c_int16),
                ("clipMax", C.c_int16), ("bitDepth", C.c_uint8), ("biMCForDMVR", C.c_uint8), ("pad0", C.c_uint8),
                ("pad1", C.c_uint8)]


class FracJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("refOff", C.c_int64), ("orgStride", C.c_int32), ("refStride", C.c_int32),
                ("width", C.c_int16), ("height", C.c_int16), ("intX", C.c_int16), ("intY", C.c_int16),
                ("predHor", C.c_int32), ("predVer", C.c_int32), ("motionLambda", C.c_double), ("useHad", C.c_uint8),
                ("useAltHpelIf", C.c_uint8), ("imvShift", C.c_uint8), ("bitDepth", C.c_uint8), ("wideOrg", C.c_int32)]


class FracResult(C.Structure):
    _fields_ = [("halfX", C.c_int16), ("halfY", C.c_int16), ("qterX", C.c_int16), ("qterY", C.c_int16),
                ("cost", C.c_uint64)]


class TrJob(C.Structure):
    _fields_ = [("srcOff", C.c_int64), ("dstOff", C.c_int64), ("srcStride", C.c_int32), ("dstStride", C.c_int32),
                ("width", C.c_int16), ("height", C.c_int16), ("typeHor", C.c_uint8), ("typeVer", C.c_uint8),
                ("bitDepth", C.c_uint8), ("pad", C.c_uint8)]


class QuantJob(C.Structure):
    _fields_ = [("srcOff", C.c_int64), ("dstOff", C.c_int64), ("width", C.c_int16), ("height", C.c_int16),
                ("qpPer", C.c_int16), ("qpRem", C.c_int16), ("bitDepth", C.c_uint8), ("isIRAP", C.c_uint8),
                ("isTransformSkip", C.c_uint8), ("pad", C.c_uint8), ("pad2", C.c_int32)]


class FullJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("refOff", C.c_int64), ("orgStride", C.c_int32), ("refStride", C.c_int32),
                ("puX", C.c_int16), ("puY", C.c_int16), ("width", C.c_int16), ("height", C.c_int16),
                ("subShift", C.c_int16), ("imvShift", C.c_uint8), ("signedSamples", C.c_uint8), ("predHor", C.c_int32),
                ("predVer", C.c_int32), ("motionLambda", C.c_double), ("centerHor", C.c_int32), ("centerVer", C.c_int32),
                ("searchRange", C.c_int32), ("pad", C.c_int32)]


class McJob(C.Structure):
    _fields_ = [("refOff", C.c_int64), ("dstOff", C.c_int64), ("refStride", C.c_int32), ("dstStride", C.c_int32),
                ("width", C.c_int16), ("height", C.c_int16), ("mvHor", C.c_int32), ("mvVer", C.c_int32), ("bi", C.c_uint8),
                ("bitDepth", C.c_uint8), ("useAltHpelIf", C.c_uint8), ("chroma", C.c_uint8)]


class PelOpJob(C.Structure):
    _fields_ = [("aOff", C.c_int64), ("bOff", C.c_int64), ("dstOff", C.c_int64), ("aStride", C.c_int32),
                ("bStride", C.c_int32), ("dstStride", C.c_int32), ("width", C.c_int16), ("height", C.c_int16),
                ("bitDepth", C.c_uint8), ("bcwWeight", C.c_int8), ("pad1", C.c_uint8), ("pad2", C.c_uint8), ("pad3", C.c_int32)]


class TuJob(C.Structure):
    _fields_ = [("resiOff", C.c_int64), ("outOff", C.c_int64), ("resiStride", C.c_int32), ("width", C.c_int16),
                ("height", C.c_int16), ("qpPer", C.c_int16), ("qpRem", C.c_int16), ("typeHor", C.c_uint8),
                ("typeVer", C.c_uint8), ("bitDepth", C.c_uint8), ("isIRAP", C.c_uint8), ("pad", C.c_int32)]


class TuResult(C.Structure):
    _fields_ = [("sse", C.c_uint64), ("sumAbs", C.c_int32), ("absSum", C.c_int32)]


class AffineJob(C.Structure):
    _fields_ = [("predOff", C.c_int64), ("resiOff", C.c_int64), ("derivHOff", C.c_int64), ("derivVOff", C.c_int64),
                ("predStride", C.c_int32), ("resiStride", C.c_int32), ("derivStride", C.c_int32), ("width", C.c_int16),
                ("height", C.c_int16), ("sixParam", C.c_uint8), ("pad", C.c_uint8 * 7)]


class MeCfg(C.Structure):
    _fields_ = [("bipredSearchRange", C.c_int32), ("useHadME", C.c_uint8), ("fastInterSearchMode13", C.c_uint8),
                ("extendedSettings", C.c_uint8), ("firstSearchStop", C.c_uint8), ("uniformImv", C.c_int32), ("uniformSquare", C.c_int32),
                ("uniformBi", C.c_int32), ("noUniMvList", C.c_uint8), ("biPatternGiven", C.c_uint8), ("pad0", C.c_uint8), ("pad1", C.c_uint8)]


class MeJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("refOff", C.c_int64), ("otherPredOff", C.c_int64), ("orgStride", C.c_int32),
                ("refStride", C.c_int32), ("otherPredStride", C.c_int32), ("puX", C.c_int16), ("puY", C.c_int16),
                ("width", C.c_int16), ("height", C.c_int16), ("bi", C.c_uint8), ("imv", C.c_uint8), ("mvpIdx", C.c_uint8),
                ("numAmvpCand", C.c_uint8), ("mvPredHor", C.c_int32), ("mvPredVer", C.c_int32), ("mvHor", C.c_int32),
                ("mvVer", C.c_int32), ("amvpCand", (C.c_int32 * 2) * 2), ("mvpIdxBits", C.c_uint32 * 2), ("bits", C.c_uint32),
                ("searchRange", C.c_int32), ("motionLambda", C.c_double), ("numExtraStart", C.c_int32),
                ("extraStart", (C.c_int32 * 2) * 15), ("flags", C.c_uint32)]


MEJ_CACHED_INT_MV = 1


class MeOut(C.Structure):
    _fields_ = [("mvHor", C.c_int32), ("mvVer", C.c_int32), ("mvPredHor", C.c_int32), ("mvPredVer", C.c_int32),
                ("mvpIdx", C.c_int32), ("bits", C.c_uint32), ("cost", C.c_uint64), ("intX", C.c_int32), ("intY", C.c_int32),
                ("intDist", C.c_uint64)]


class PredJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("refOff", C.c_int64 * 2), ("predOff", C.c_int64), ("outOff", C.c_int64), ("orgStride", C.c_int32),
                ("refStride", C.c_int32 * 2), ("predStride", C.c_int32), ("outStride", C.c_int32), ("mv", (C.c_int32 * 2) * 2),
                ("width", C.c_int16), ("height", C.c_int16), ("mode", C.c_uint8), ("epilogue", C.c_uint8), ("bitDepth", C.c_uint8),
                ("useAltHpelIf", C.c_uint8), ("chroma", C.c_uint8), ("route", C.c_uint8), ("bcwWeight", C.c_int16)]


class MaskedSadJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("curOff", C.c_int64), ("maskOff", C.c_int64), ("orgStride", C.c_int32), ("curStride", C.c_int32),
                ("maskStride", C.c_int32), ("maskStride2", C.c_int32), ("width", C.c_int16), ("height", C.c_int16), ("subShift", C.c_int16),
                ("stepX", C.c_int16)]


class GeoBlendJob(C.Structure):
    _fields_ = [("src0Off", C.c_int64), ("src1Off", C.c_int64), ("dstOff", C.c_int64), ("weightOff", C.c_int64), ("src0Stride", C.c_int32),
                ("src1Stride", C.c_int32), ("dstStride", C.c_int32), ("weightStride", C.c_int32), ("width", C.c_int16), ("height", C.c_int16),
                ("stepX", C.c_int16), ("pad", C.c_int16)]


class DmvrJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("refOff", C.c_int64 * 2), ("predOff", C.c_int64), ("outOff", C.c_int64), ("orgStride", C.c_int32),
                ("refStride", C.c_int32 * 2), ("predStride", C.c_int32), ("outStride", C.c_int32), ("mv", (C.c_int32 * 2) * 2), ("puX", C.c_int32),
                ("puY", C.c_int32), ("width", C.c_int16), ("height", C.c_int16), ("bioApplied", C.c_uint8), ("epilogue", C.c_uint8),
                ("bitDepth", C.c_uint8), ("pad0", C.c_uint8), ("mvdRow", C.c_int32)]


class LfnstJob(C.Structure):
    _fields_ = [("srcOff", C.c_int64), ("dstOff", C.c_int64), ("mode", C.c_uint8), ("index", C.c_uint8), ("size", C.c_uint8), ("zeroOutSize", C.c_uint8),
                ("inverse", C.c_uint8), ("pad0", C.c_uint8), ("pad1", C.c_uint8), ("pad2", C.c_uint8)]


MAX_REF = 4


class PisRow(C.Structure):
    _fields_ = [("mvHor", C.c_int32), ("mvVer", C.c_int32), ("mvPredHor", C.c_int32), ("mvPredVer", C.c_int32), ("mvpIdx", C.c_int32),
                ("bits", C.c_uint32), ("cost", C.c_uint64)]


class PisPu(C.Structure):
    _fields_ = [("cost", C.c_uint64 * 2), ("costBi", C.c_uint64), ("bits", C.c_uint32 * 3), ("refIdx", C.c_int32 * 2), ("mv", (C.c_int32 * 2) * 2),
                ("refIdxBi", C.c_int32 * 2), ("mvBi", (C.c_int32 * 2) * 2), ("refineList", C.c_int32), ("interDir", C.c_int32), ("smvdMode", C.c_int32),
                ("mvpIdxL1Zero", C.c_int32), ("pad", C.c_int32)]


class PisPuIn(C.Structure):
    _fields_ = [("noSmvd", C.c_uint8), ("uniMvInsert", C.c_uint8), ("uniMvSelfIsNew", C.c_uint8), ("bcwWeightL1", C.c_int8), ("uniMvSelfPos", C.c_int16),
                ("bcwIdxBits", C.c_uint8), ("bcwFastSkipPoc", C.c_uint8)]


class PisLevel(C.Structure):
    _fields_ = [("numPU", C.c_int32), ("numRef", C.c_int32 * 2), ("smvdBit", C.c_int32), ("mbBits", C.c_uint32 * 3), ("refStride", C.c_int32),
                ("refPlaneOff", (C.c_int64 * MAX_REF) * 2), ("uniJobs", C.c_void_p), ("uniOut", C.c_void_p), ("uniRows", C.c_void_p), ("pus", C.c_void_p),
                ("predOther", C.c_void_p), ("biJobs", C.c_void_p), ("biOut", C.c_void_p), ("predFinal", C.c_void_p), ("parentIdx", C.c_void_p),
                ("parentRows", C.c_void_p), ("parentNumPU", C.c_int32), ("pad", C.c_int32), ("pos", C.c_void_p), ("bdofEnabled", C.c_int32), ("curPoc", C.c_int32),
                ("refPoc", (C.c_int32 * MAX_REF) * 2), ("predFinalC", C.c_void_p), ("posC", C.c_void_p), ("refPlaneOffC", ((C.c_int64 * MAX_REF) * 2) * 2),
                ("affJobs", C.c_void_p), ("affLowDelay", C.c_int32), ("affCheckLDC", C.c_int32), ("smvdJobs", C.c_void_p), ("symRefIdx", C.c_int32 * 2),
                ("candsGiven", C.c_int32), ("biRestricted", C.c_int32), ("list1FromList0", C.c_int32 * MAX_REF), ("puIn", C.c_void_p), ("biRows", C.c_void_p),
                ("distBiP", C.c_void_p), ("mvdL1Zero", C.c_int32), ("fastMEForGenBLowDelay", C.c_int32), ("givenRows", C.c_int32), ("picW", C.c_int32), ("picH", C.c_int32), ("ctuSize", C.c_int32)]


class PisLevelRun(C.Structure):
    _fields_ = [("pis", PisLevel), ("pic", PicParams), ("picBi", PicParams), ("cfgUni", MeCfg), ("cfgBi", MeCfg), ("width", C.c_int32), ("height", C.c_int32),
                ("bdof", C.c_int32), ("pad0", C.c_int32), ("uniOut", C.c_void_p), ("biOut", C.c_void_p), ("tu", C.c_void_p), ("tuRes", C.c_void_p), ("qcoef", C.c_void_p),
                ("numTU", C.c_int32), ("numCands", C.c_int32), ("tuW", C.c_int32), ("tuH", C.c_int32), ("cand", C.c_uint8 * 8), ("tuC", C.c_void_p),
                ("tuResC", C.c_void_p), ("qcoefC", C.c_void_p), ("numTUC", C.c_int32), ("tuWC", C.c_int32), ("tuHC", C.c_int32), ("pad1", C.c_int32), ("affOut", C.c_void_p),
                ("mtsTest", C.c_void_p), ("mtsMaxCand", C.c_int32), ("pad2", C.c_int32)]


class PisBuffers(C.Structure):
    _fields_ = [("org", C.c_void_p), ("dpb", C.c_void_p), ("pred", C.c_void_p), ("resi", C.c_void_p), ("orgBi", C.c_void_p), ("predC", C.c_void_p), ("resiC", C.c_void_p)]


class AffineMeJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("refOff", C.c_int64), ("otherPredOff", C.c_int64), ("predOff", C.c_int64), ("orgStride", C.c_int32), ("refStride", C.c_int32),
                ("otherPredStride", C.c_int32), ("predStride", C.c_int32), ("puX", C.c_int16), ("puY", C.c_int16), ("width", C.c_int16), ("height", C.c_int16),
                ("sixParam", C.c_uint8), ("interDir", C.c_uint8), ("imv", C.c_uint8), ("bi", C.c_uint8), ("useSatd", C.c_uint8), ("useAffineType", C.c_uint8),
                ("amvrEncOpt", C.c_uint8), ("lowDelayRounds", C.c_uint8), ("profAllowed", C.c_uint8), ("profNeedsLargeGrad", C.c_uint8), ("profIsBi", C.c_uint8),
                ("bcwWeight", C.c_int8), ("mvPred", (C.c_int32 * 2) * 3), ("mv", (C.c_int32 * 2) * 3), ("bits", C.c_uint32), ("pad1", C.c_uint32),
                ("motionLambda", C.c_double), ("hevcCost", C.c_uint64), ("amvpCand", ((C.c_int32 * 2) * 3) * 2), ("mvpIdxBits", C.c_uint32 * 2), ("numAmvpCand", C.c_uint8),
                ("mvpIdx", C.c_uint8), ("pad2", C.c_uint8 * 6)]


class AffineMeOut(C.Structure):
    _fields_ = [("mv", (C.c_int32 * 2) * 3), ("bits", C.c_uint32), ("iterations", C.c_int32), ("refinements", C.c_int32), ("mvpIdx", C.c_int32), ("cost", C.c_uint64)]


class SmvdTrace(C.Structure):
    _fields_ = [("cost", C.c_uint64), ("mv", C.c_int32 * 2), ("idx", C.c_int32 * 2)]


class SmvdJob(C.Structure):
    _fields_ = [("orgOff", C.c_int64), ("refOff", C.c_int64 * 2), ("orgStride", C.c_int32), ("refStride", C.c_int32 * 2), ("puX", C.c_int16), ("puY", C.c_int16),
                ("width", C.c_int16), ("height", C.c_int16), ("imv", C.c_uint8), ("useSatd", C.c_uint8), ("clipBiPred", C.c_uint8), ("bcwWeightTar", C.c_int8),
                ("numCand", C.c_uint8 * 2), ("numStart", C.c_uint8), ("numFixed", C.c_uint8), ("skip", C.c_uint8), ("pad_", C.c_uint8 * 3),
                ("cand", ((C.c_int32 * 2) * 2) * 2), ("mvpIdxBits", C.c_uint32 * 2), ("modeBits", C.c_uint32), ("motionLambda", C.c_double),
                ("starts", (C.c_int32 * 2) * 18), ("mvCur", C.c_int32 * 2), ("mvTar", C.c_int32 * 2), ("predSym", (C.c_int32 * 2) * 2), ("mvpIdxSym", C.c_int32 * 2),
                ("cost", C.c_uint64), ("trace", SmvdTrace * 4)]


class LfnstTuJob(C.Structure):
    _fields_ = [("coefOff", C.c_int64), ("width", C.c_int16), ("height", C.c_int16), ("mode", C.c_uint8), ("index", C.c_uint8), ("transpose", C.c_uint8),
                ("inverse", C.c_uint8)]


_STRUCTS = [DistJob, TzJob, MeResult, PicParams, IfJob, FracJob, FracResult, TrJob, QuantJob, FullJob, McJob, PelOpJob,
            TuJob, TuResult, AffineJob, MeCfg, MeJob, MeOut, PredJob, MaskedSadJob, GeoBlendJob, DmvrJob, LfnstJob,
            PisRow, PisPu, PisLevel, AffineMeJob, AffineMeOut, LfnstTuJob, PisLevelRun, PisBuffers, SmvdJob, PisPuIn]   # order of vtmhip_struct_size(which)

# every symbol include/vtmhip.h declares (tests/test_abi.py checks the exports against the header text)
_PROTOS = {
    "vtmhip_abi_version": (C.c_int, []),
    "vtmhip_struct_size": (C.c_int, [C.c_int]),
    "vtmhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vtmhip_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vtmhip_destroy": (C.c_int, [C.c_void_p]),
    "vtmhip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vtmhip_use_own_stream": (C.c_int, [C.c_void_p]),
    "vtmhip_sync": (C.c_int, [C.c_void_p]),
    "vtmhip_last_error": (C.c_char_p, [C.c_void_p]),
    "vtmhip_status_string": (C.c_char_p, [C.c_int]),
    "vtmhip_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vtmhip_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vtmhip_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vtmhip_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vtmhip_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vtmhip_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vtmhip_timer_start": (C.c_int, [C.c_void_p]),
    "vtmhip_timer_stop_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "vtmhip_xGetSAD": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_uint64)]),
    "vtmhip_xGetHADs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_uint64)]),
    "vtmhip_xGetSADwMask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(C.c_uint64)]),
    "vtmhip_masked_sad_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vtmhip_remove_weight_high_freq_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_add_weighted_avg_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_lfnst_set_tables": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vtmhip_fwdLfnstNxN": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtmhip_invLfnstNxN": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtmhip_lfnst_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_bdof_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vtmhip_dmvr_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    "vtmhip_dmvr_chroma_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.c_int, C.c_void_p]),
    "vtmhip_weightedGeoBlk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtmhip_weightedGeoBlk_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtmhip_xGetSSE": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_uint64)]),
    "vtmhip_filterHor": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtmhip_filterVer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtmhip_filterCopy": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtmhip_if_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_frac_search_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_void_p]),
    "vtmhip_fastFwdTrans": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtmhip_fastInvTrans": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int32, C.c_int32]),
    "vtmhip_tr_matrix_host": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_mts_select": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_mts_select_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_mts_select2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_tu_ts_chain_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vtmhip_xT_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_xT_uniform_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vtmhip_xIT_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vtmhip_quant_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vtmhip_dequant_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_full_search_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p]),
    "vtmhip_motion_compensation_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                       C.c_int]),
    "vtmhip_mc_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vtmhip_full_search_uniform_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                       C.c_void_p]),
    "vtmhip_full_search_square_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                      C.c_void_p]),
    "vtmhip_mc_luma_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vtmhip_remove_high_freq_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_subtract_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_add_avg_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_tu_chain_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "vtmhip_affine_sobel_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_affine_equal_coeff_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vtmhip_dist_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vtmhip_dist_uniform_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_intra_cand_cost_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_satd8_grid_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    "vtmhip_xMotionEstimation_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.POINTER(MeCfg), C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_xEstimateMvPredAMVP_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                       C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_xAffineMotionEstimation_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                           C.c_void_p]),
    "vtmhip_xAffineMotionEstimation_bcw_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                               C.c_void_p]),
    "vtmhip_smvd_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtmhip_xGetSymmetricCost_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vtmhip_xSymmetricMotionEstimation_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vtmhip_symmvdCheckBestMvp_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vtmhip_xPredAffineBlk_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "vtmhip_lfnst_tu_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vtmhip_lfnst_scan_host": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "vtmhip_kernel_timing_read": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "vtmhip_merge_cand_satd_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                   C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vtmhip_pis_run_picture": (C.c_int, [C.c_void_p, C.POINTER(PisLevelRun), C.c_int, C.POINTER(PisBuffers), C.c_void_p, C.POINTER(C.c_void_p), C.c_int]),
    "vtmhip_pis_stage": (C.c_int, [C.c_void_p, C.POINTER(PisLevel), C.c_int]),
    "vtmhip_predInterSearch_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PisLevelRun), C.POINTER(PisBuffers)]),
    "vtmhip_is_uniform_shape": (C.c_int, [C.c_int, C.c_int]),
    "vtmhip_tz_search_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(PicParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_void_p]),
}

_lib = None


class VtmHipError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("libvtmhip status %d%s" % (status, (": " + detail) if detail else ""))


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm ships its own libamdhip64.so.  Two HIP runtimes in one process cannot both own the GPU (whichever
    initialises second sees no device), and streams / device pointers are only interchangeable inside ONE runtime.
    So when a PyTorch installation is present we map ITS runtime first (without importing torch); libvtmhip.so then
    binds to that already-loaded soname, whatever the later import order is."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass   # no torch: the system runtime libvtmhip.so was linked against is used


def load():
    """Loads libvtmhip.so, declares prototypes and checks the struct layouts against the library's sizeof()."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libvtmhip.so is missing (%s): build it with `python -m vtm_amd.build` -- the HIP "
                          "extension is the product, there is no CPU fallback" % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for i, s in enumerate(_STRUCTS):
        if lib.vtmhip_struct_size(i) != C.sizeof(s):
            raise ImportError("ABI mismatch: %s is %d bytes in Python, %d in libvtmhip.so"
                              % (s.__name__, C.sizeof(s), lib.vtmhip_struct_size(i)))
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_PROTOS)
