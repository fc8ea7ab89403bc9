"""Shared helpers for the motion-search tests: builds the same randomized jobs for the oracle (vo_*), the reference
shim (ref_*) and the HIP kernels."""
import ctypes as C

import numpy as np

import oracle_lib as ol
from vtm_amd import synth
from vtm_amd.lib import TzJob as HipTzJob

PU_W = [8, 16, 32, 64, 128, 4, 16, 8, 32, 64, 12, 24, 48]
PU_H = [8, 16, 32, 64, 128, 8, 4, 16]


class Scene:
    """Two frames of the synthetic clip: `cur` (original) and the border-extended reference plane."""

    def __init__(self, w=416, h=240, hard=True, t_ref=0, t_cur=2, margin=160):
        fr = (synth.gen_frames_hard if hard else synth.gen_frames)(w, h, t_cur + 1)
        self.W, self.H = w, h
        self.cur = np.ascontiguousarray(fr[t_cur])
        self.ref_buf, self.ref_off, self.ref_stride = synth.extend_plane(fr[t_ref], margin)
        self.margin = margin


def random_tz_jobs(scene, n, seed=5, ranges=(64, 96, 192, 384, 8), allow_ext=True, sizes=None):
    rng = np.random.default_rng(seed)
    jobs = []
    trial = 0
    while len(jobs) < n:
        trial += 1
        w = int(rng.choice(sizes[0] if sizes else PU_W))
        h = int(rng.choice(sizes[1] if sizes else PU_H))
        if w > scene.W or h > scene.H:
            continue
        x = int(rng.integers(0, (scene.W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (scene.H - h) // 4 + 1)) * 4
        j = dict(w=w, h=h, x=x, y=y, subShift=1 if (h > 8 and w <= 64) else 0,
                 lam=float(rng.uniform(1, 40)), predHor=int(rng.integers(-64, 64)), predVer=int(rng.integers(-64, 64)),
                 mvHor=int(rng.integers(-40 * 16, 40 * 16)), mvVer=int(rng.integers(-30 * 16, 30 * 16)),
                 searchRange=int(rng.choice(ranges)),
                 ext=int(allow_ext and trial % 5 == 0), fast=int(trial % 11 == 0), firstStop=int(trial % 3 != 0),
                 hasInt=int(trial % 4 == 0), intHor=int(rng.integers(-20, 20)), intVer=int(rng.integers(-20, 20)),
                 extra=[(int(rng.integers(-30 * 16, 30 * 16)), int(rng.integers(-30 * 16, 30 * 16)))
                        for _ in range(int(rng.integers(0, 6)))])
        if trial % 7 == 0:
            j["mvHor"] = j["mvVer"] = 0
        jobs.append(j)
    return jobs


def oracle_ctx(scene, j, org_block):
    c = ol.MeCtx()
    c.org = org_block.ctypes.data
    c.orgStride = org_block.shape[1]
    c.ref = scene.ref_buf.ctypes.data + 2 * (scene.ref_off + j["y"] * scene.ref_stride + j["x"])
    c.refStride = scene.ref_stride
    c.w, c.h, c.subShift, c.bitDepth, c.imvShift = j["w"], j["h"], j["subShift"], 10, 0
    c.mv = ol.MvCost(j["lam"], j["predHor"], j["predVer"], 2)
    c.picW, c.picH, c.puX, c.puY, c.ctuSize = scene.W, scene.H, j["x"], j["y"], 128
    return c


def oracle_tz_job(j):
    t = ol.TzJob()
    t.mvHor, t.mvVer, t.searchRange = j["mvHor"], j["mvVer"], j["searchRange"]
    t.extendedSettings, t.fastSettings, t.firstSearchStop = j["ext"], j["fast"], j["firstStop"]
    t.hasIntMv2Nx2NPred, t.intMv2Nx2NPredHor, t.intMv2Nx2NPredVer = j["hasInt"], j["intHor"], j["intVer"]
    t.numExtraStart = len(j["extra"])
    for i, (a, b) in enumerate(j["extra"]):
        t.extraStart[i][0], t.extraStart[i][1] = a, b
    return t


def hip_tz_jobs(scene, jobs, cur_stride):
    arr = (HipTzJob * len(jobs))()
    for k, j in enumerate(jobs):
        t = arr[k]
        t.orgOff = j["y"] * cur_stride + j["x"]
        t.refOff = scene.ref_off + j["y"] * scene.ref_stride + j["x"]
        t.orgStride, t.refStride = cur_stride, scene.ref_stride
        t.puX, t.puY, t.width, t.height = j["x"], j["y"], j["w"], j["h"]
        t.subShift, t.imvShift, t.signedSamples = j["subShift"], 0, j.get("signed", 0)
        t.predHor, t.predVer, t.motionLambda = j["predHor"], j["predVer"], j["lam"]
        t.mvHor, t.mvVer, t.searchRange = j["mvHor"], j["mvVer"], j["searchRange"]
        t.extendedSettings, t.fastSettings, t.firstSearchStop = j["ext"], j["fast"], j["firstStop"]
        t.hasIntMv2Nx2NPred, t.intMv2Nx2NPredHor, t.intMv2Nx2NPredVer = j["hasInt"], j["intHor"], j["intVer"]
        t.numExtraStart = len(j["extra"])
        for i, (a, b) in enumerate(j["extra"]):
            t.extraStart[i][0], t.extraStart[i][1] = a, b
    return arr


def run_oracle_tz(scene, jobs):
    L = ol.oracle()
    out = []
    for j in jobs:
        org = np.ascontiguousarray(scene.cur[j["y"]:j["y"] + j["h"], j["x"]:j["x"] + j["w"]])
        c = oracle_ctx(scene, j, org)
        t = oracle_tz_job(j)
        r = ol.MeResult()
        L.vo_tz_search(C.byref(c), C.byref(t), C.byref(r))
        out.append((r.mvX, r.mvY, r.cost, r.dist, r.nEval))
    return out


# ---- whole xMotionEstimation jobs ---------------------------------------------------------------------------------
AMVR_SHIFT = {0: 2, 1: 4, 2: 6, 3: 3}   # cu.imv -> right shift from internal precision (Mv::m_amvrPrecision)


def _round_amvr(v, imv):
    """Mv::roundTransPrecInternal2Amvr on one component."""
    rs = AMVR_SHIFT[imv]
    o = 1 << (rs - 1)
    return (((v + o - 1) >> rs) if v >= 0 else ((v + o) >> rs)) << rs


def random_mest_jobs(scene, n, seed=17, sizes=None):
    """(PU, list, refIdx) jobs of InterSearch::xMotionEstimation: uni / bi, every cu.imv mode, AMVP candidates rounded to the AMVR
    precision as the encoder's AMVP lists are, m_uniMvList entries with duplicates."""
    rng = np.random.default_rng(seed)
    jobs = []
    t = 0
    while len(jobs) < n:
        t += 1
        w = int(rng.choice(sizes[0] if sizes else PU_W))
        h = int(rng.choice(sizes[1] if sizes else PU_H))
        if w > scene.W or h > scene.H or (w == 4 and h == 4):
            continue
        x = int(rng.integers(0, (scene.W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (scene.H - h) // 4 + 1)) * 4
        imv = int(rng.choice([0, 0, 0, 1, 2, 3]))
        cands = [[_round_amvr(int(rng.integers(-24 * 16, 24 * 16)), imv), _round_amvr(int(rng.integers(-16 * 16, 16 * 16)), imv)] for _ in range(2)]
        if t % 9 == 0:
            cands[1] = list(cands[0])
        idx = int(rng.integers(0, 2))
        ncand = 2 if t % 6 else 1
        if ncand == 1:
            idx = 0
        extra = [(int(rng.integers(-30 * 16, 30 * 16)), int(rng.integers(-30 * 16, 30 * 16))) for _ in range(int(rng.integers(0, 7)))]
        if len(extra) > 2 and t % 2:
            extra.append(extra[0])   # duplicates exercise the de-duplication loop
        # bi only for widths a VVC PU can have: the reference's x86 removeHighFreq4 (BufferX86.h:873-892) touches just the first four
        # columns of a row, so for the (non-existent) 12-wide PU its scalar and SIMD builds disagree
        bi = int(t % 3 == 0 and (w & (w - 1)) == 0)
        jobs.append(dict(w=w, h=h, x=x, y=y, bi=bi, imv=imv, mvpIdx=idx, numCand=ncand, cands=cands,
                         mvPred=tuple(cands[idx]), mv=(int(rng.integers(-30, 30)) * 16 + int(rng.integers(0, 16)), int(rng.integers(-20, 20)) * 16 + int(rng.integers(0, 16))),
                         idxBits=(int(rng.integers(1, 3)), int(rng.integers(1, 3))), bits=int(rng.integers(3, 12)),
                         searchRange=int(rng.choice([64, 96, 192, 8])), lam=float(rng.uniform(1, 40)), extra=extra,
                         other_seed=int(rng.integers(0, 1 << 30))))
    return jobs


def other_pred(scene, j):
    """Prediction 'from the other list' for the bi-pred target: the co-located reference block displaced a little plus noise."""
    rng = np.random.default_rng(j["other_seed"])
    dx, dy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
    m = scene.margin
    plane = scene.ref_buf.reshape(-1, scene.ref_stride)
    blk = plane[m + j["y"] + dy:m + j["y"] + dy + j["h"], m + j["x"] + dx:m + j["x"] + dx + j["w"]].astype(np.int32)
    return np.ascontiguousarray(np.clip(blk + rng.integers(-6, 7, blk.shape), 0, 1023).astype(np.int16))


def oracle_mest_job(scene, j, keep):
    """ctypes job for vo_motion_estimation / ref_motion_estimation; `keep` collects the arrays the pointers refer to."""
    t = ol.MestJob()
    t.org = scene.cur.ctypes.data + 2 * (j["y"] * scene.W + j["x"])
    t.orgStride = scene.W
    t.ref = scene.ref_buf.ctypes.data + 2 * (scene.ref_off + j["y"] * scene.ref_stride + j["x"])
    t.refStride = scene.ref_stride
    if j["bi"]:
        o = other_pred(scene, j)
        keep.append(o)
        t.otherPred, t.otherStride = o.ctypes.data, j["w"]
    t.w, t.h, t.puX, t.puY, t.picW, t.picH, t.ctuSize, t.bitDepth = j["w"], j["h"], j["x"], j["y"], scene.W, scene.H, 128, 10
    t.bi, t.imv, t.mvpIdx, t.numAmvpCand = j["bi"], j["imv"], j["mvpIdx"], j["numCand"]
    t.mvPredHor, t.mvPredVer = j["mvPred"]
    t.mvHor, t.mvVer = j["mv"]
    for i in range(2):
        t.amvpCand[i][0], t.amvpCand[i][1] = j["cands"][i]
        t.mvpIdxBits[i] = j["idxBits"][i]
    t.bits, t.searchRange, t.motionLambda = j["bits"], j["searchRange"], j["lam"]
    t.numExtraStart = len(j["extra"])
    for i, (a, b) in enumerate(j["extra"]):
        t.extraStart[i][0], t.extraStart[i][1] = a, b
    t.bcwWeight = j.get("bcw", 0)      # bi: the searched list's CU-level BCW weight (0: the default pair)
    t.cachedIntMv = j.get("cached", 0)  # uni: rcMv = the block-vector cache's integer vector, xTZSearch with bFastSettings (InterSearch.cpp:3360-3368, :3434-3441)
    return t


def random_affine_jobs(scene, n, seed=5, sizes=(16, 32, 64, 128)):
    """Affine ME jobs on a scene: control-point vectors = a small rotation / zoom around a translation, predictors near them."""
    rng = np.random.default_rng(seed)
    jobs = []
    while len(jobs) < n:
        w, h = int(rng.choice(sizes)), int(rng.choice(sizes))
        if w > scene.W or h > scene.H:
            continue
        x = int(rng.integers(0, (scene.W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (scene.H - h) // 4 + 1)) * 4
        six = int(rng.integers(0, 2))
        imv = int(rng.choice([0, 0, 0, 1, 2]))
        tx, ty = int(rng.integers(-12 * 16, 12 * 16)), int(rng.integers(-8 * 16, 8 * 16))
        a, b = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))          # per-block vector differences (1/16 sample)
        mv = [[tx, ty], [tx + a, ty + b], [tx - b + int(rng.integers(-8, 9)) * six, ty + a + int(rng.integers(-8, 9)) * six]]
        if len(jobs) % 7 == 0:
            mv = [[tx, ty], [tx, ty], [tx, ty]]        # pure translation: PROF is off, the gradient iteration starts from a flat model
        if len(jobs) % 11 == 0:
            mv[1] = [tx + 900, ty - 700]              # a spread the sub-block vectors may not have (isSubblockVectorSpreadOverLimit)
        pred = [[v[0] + int(rng.integers(-3, 4)) * 4, v[1] + int(rng.integers(-3, 4)) * 4] for v in mv]
        sh = {0: 2, 1: 0, 2: 4}[imv]
        rnd = lambda v: (((v + (1 << sh >> 1) - (1 if v >= 0 else 0)) >> sh) << sh) if sh else v   # noqa: E731
        pred = [[rnd(p[0]), rnd(p[1])] for p in pred]
        jobs.append(dict(w=w, h=h, x=x, y=y, six=six, imv=imv, mv=mv, pred=pred, bi=int(rng.integers(0, 3) == 0), satd=int(rng.integers(0, 4) != 0),
                         affine_type=int(rng.integers(0, 5) != 0), enc_opt=int(imv != 2 and rng.integers(0, 2)), low_delay=int(rng.integers(0, 2)),
                         inter_dir=int(rng.choice([1, 2, 3])), prof=int(rng.integers(0, 4) != 0), prof_large=int(rng.integers(0, 2)), prof_bi=int(rng.integers(0, 2)),
                         bits=int(rng.integers(4, 14)), lam=float(rng.uniform(2, 30)), hevc_scale=float(rng.choice([0.5, 1.0, 4.0])),
                         other_seed=int(rng.integers(0, 1 << 30))))
    return jobs


def affine_pred_struct(scene, j):
    p = ol.AffinePred()
    p.ref = scene.ref_buf.ctypes.data + 2 * (scene.ref_off + j["y"] * scene.ref_stride + j["x"])
    p.refStride, p.w, p.h, p.puX, p.puY, p.picW, p.picH, p.ctuSize, p.bitDepth = scene.ref_stride, j["w"], j["h"], j["x"], j["y"], scene.W, scene.H, 128, 10
    p.sixParam, p.interDir, p.profAllowed, p.profNeedsLargeGrad, p.profIsBi = j["six"], j["inter_dir"], j["prof"], j["prof_large"], j["prof_bi"]
    return p


def affine_me_struct(scene, j, keep):
    t = ol.AffineMeJob()
    t.pred = affine_pred_struct(scene, j)
    t.org, t.orgStride = scene.cur.ctypes.data + 2 * (j["y"] * scene.W + j["x"]), scene.W
    if j["bi"]:
        o = other_pred(scene, j)
        keep.append(o)
        t.otherPred, t.otherStride = o.ctypes.data, j["w"]
    t.bi, t.imv, t.useSatd, t.useAffineType, t.amvrEncOpt, t.lowDelayRounds = j["bi"], j["imv"], j["satd"], j["affine_type"], j["enc_opt"], j["low_delay"]
    for i in range(3):
        t.mvPred[i][0], t.mvPred[i][1] = j["pred"][i]
        t.mv[i][0], t.mv[i][1] = j["mv"][i]
    t.bits, t.motionLambda = j["bits"], j["lam"]
    return t


# ---- SMVD (symmetric MVD search of predInterSearch) -------------------------------------------------------------------------------
class SmvdScene(Scene):
    """The original picture between two references: list 0 = an earlier frame (ref_buf), list 1 = a later one (ref_buf2)."""

    def __init__(self, w=416, h=240, hard=False, margin=160, bit_depth=10):
        fr = (synth.gen_frames_hard if hard else synth.gen_frames)(w, h, 5)
        if bit_depth != 10:      # the generator makes 10-bit samples
            fr = [(f.astype(np.int32) << (bit_depth - 10)).astype(np.int16) if bit_depth > 10 else (f >> (10 - bit_depth)).astype(np.int16) for f in fr]
        self.bd = bit_depth
        self.W, self.H = w, h
        self.cur = np.ascontiguousarray(fr[2])
        self.ref_buf, self.ref_off, self.ref_stride = synth.extend_plane(fr[0], margin)
        self.ref_buf2, _, _ = synth.extend_plane(fr[4], margin)
        self.margin = margin


SMVD_SIZES = [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 8), (8, 32), (16, 4), (4, 16), (64, 16), (4, 32), (128, 64)]


def random_smvd_jobs(scene, n, seed=3, sizes=None):
    """(PU, AMVP lists, start vectors) as predInterSearch hands them to the SMVD block; vectors in 1/16 sample, candidates at the AMVR precision."""
    rng = np.random.default_rng(seed)
    jobs = []
    while len(jobs) < n:
        w, h = (sizes or SMVD_SIZES)[int(rng.integers(0, len(sizes or SMVD_SIZES)))]
        if w > scene.W or h > scene.H:
            continue
        x = int(rng.integers(0, (scene.W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (scene.H - h) // 4 + 1)) * 4
        imv = int(rng.choice([0, 0, 0, 1, 2, 3]))
        base = (int(rng.integers(-10 * 16, 10 * 16)), int(rng.integers(-6 * 16, 6 * 16)))
        near = lambda c, r: (_round_amvr(c[0] + int(rng.integers(-r, r + 1)), imv), _round_amvr(c[1] + int(rng.integers(-r, r + 1)), imv))   # noqa: E731
        cands = [[near(base, 40), near(base, 40)], [near((-base[0], -base[1]), 40), near((-base[0], -base[1]), 40)]]
        for l in range(2):
            if rng.integers(0, 6) == 0:
                cands[l][1] = cands[l][0]              # equal candidates: the SMVD block shortens the list (:2668-2671)
        num = [int(rng.choice([1, 2, 2, 2])), int(rng.choice([1, 2, 2, 2]))]
        starts = [near(base, 60) if rng.integers(0, 4) else cands[0][int(rng.integers(0, 2))] for _ in range(int(rng.integers(2, 9)))]
        starts = [(s[0] + int(rng.integers(-2, 3)) * (k >= 3), s[1]) for k, s in enumerate(starts)]   # history entries need not sit on the AMVR grid
        jobs.append(dict(w=w, h=h, x=x, y=y, imv=imv, satd=int(rng.integers(0, 5) != 0), clip=int(rng.integers(0, 3) == 0),
                         bcw=int(rng.choice([4, 4, 4, 4, -2, 3, 5, 10])), num=num, cands=cands, idxBits=[int(rng.integers(1, 3)), int(rng.integers(1, 4))],
                         lam=float(rng.uniform(2, 40)), starts=starts, numFixed=int(rng.integers(2, 4)), modeBits=int(rng.integers(3, 9))))
    return jobs


def smvd_struct(scene, j):
    t = ol.SmvdJob()
    t.org, t.orgStride = scene.cur.ctypes.data + 2 * (j["y"] * scene.W + j["x"]), scene.W
    t.ref[0] = scene.ref_buf.ctypes.data + 2 * (scene.ref_off + j["y"] * scene.ref_stride + j["x"])
    t.ref[1] = scene.ref_buf2.ctypes.data + 2 * (scene.ref_off + j["y"] * scene.ref_stride + j["x"])
    t.refStride[0] = t.refStride[1] = scene.ref_stride
    t.w, t.h, t.puX, t.puY, t.picW, t.picH, t.ctuSize, t.bitDepth = j["w"], j["h"], j["x"], j["y"], scene.W, scene.H, 128, getattr(scene, "bd", 10)
    t.imv, t.useSatd, t.clipBiPred, t.bcwWeightTar = j["imv"], j["satd"], j["clip"], j["bcw"]
    for l in range(2):
        t.numCand[l] = j["num"][l]
        for i in range(2):
            t.cand[l][i][0], t.cand[l][i][1] = j["cands"][l][i]
        t.mvpIdxBits[l] = j["idxBits"][l]
    t.motionLambda = j["lam"]
    return t


def smvd_member_results(scene, j, lib, prefix):
    """(xGetSymmetricCost, xSymmetricMotionEstimation, symmvdCheckBestMvp) results of one job through `lib` (prefix "vo_": oracle, "ref_": the real members).
    Inputs derived from the job alone: start vector starts[0], its mirror around the first predictor pair, costs a little above the start cost."""
    I2 = C.c_int * 2
    t = smvd_struct(scene, j)
    pc, pt, start = j["cands"][0][0], j["cands"][1][0], j["starts"][0]
    pair = (pt[0] - (start[0] - pc[0]), pt[1] - (start[1] - pc[1]))
    fn = getattr(lib, prefix + "symmetric_cost")
    fn.restype = C.c_uint64
    c0 = fn(C.byref(t), I2(*start), I2(*pair))
    mc, mt, cost = I2(*start), I2(*pair), C.c_uint64(c0 + int(j["lam"] * 6))
    getattr(lib, prefix + "symmetric_me")(C.byref(t), I2(*pc), I2(*pt), mc, mt, C.byref(cost))
    me = (tuple(mc), tuple(mt), cost.value)
    pred, idx, cost = (I2 * 2)(I2(*pc), I2(*pt)), I2(0, 0), C.c_uint64(c0 + int(j["lam"] * 9))
    getattr(lib, prefix + "symmvd_check_best_mvp")(C.byref(t), I2(*start), j["x"] // 4 & 1, pred, idx, C.byref(cost))
    return c0, me, (tuple(pred[0]), tuple(pred[1]), tuple(idx), cost.value)


def smvd_search_oracle(scene, j, L):
    t = smvd_struct(scene, j)
    st = ((C.c_int * 2) * len(j["starts"]))(*[(C.c_int * 2)(*v) for v in j["starts"]])
    r = ol.SmvdResult()
    L.vo_smvd_search(C.byref(t), j["numFixed"], len(j["starts"]), st, j["modeBits"], C.byref(r))
    return r.key()
