"""Frame-level driver of the hot path: builds the job tables of one inter picture and runs the stages on the device.

This is the host-side orchestration that, inside the reference, is InterSearch::predInterSearch -> xMotionEstimation
(EncoderLib/InterSearch.cpp:2245-3065, 3299-3494) called from the CU recursion.  The reference visits one PU at a
time; here all PUs of one quadtree level x all reference pictures form one launch, and level L+1 takes its start
vector / MV predictor from the level-L result of the enclosing block (a stand-in for AMVP, which needs the CU
recursion that is out of scope -- SURVEY.md section 8a).  Job tables and results stay in HBM; torch is used only for
device memory, streams and the tiny gather that forms the child predictors.
"""
import ctypes as C

import os

import numpy as np

from . import lib as _lib
from .lib import MeResult, PicParams, TzJob

TZ_DT = np.dtype(TzJob)
RES_DT = np.dtype(MeResult)
# int32 word indices inside the structs (checked against ctypes offsets below)
_J_PRED_HOR, _J_PRED_VER, _J_MV_HOR, _J_MV_VER = 9, 10, 14, 15
assert TzJob.predHor.offset == 4 * _J_PRED_HOR and TzJob.mvHor.offset == 4 * _J_MV_HOR
assert C.sizeof(TzJob) % 4 == 0 and C.sizeof(MeResult) == 32


WAVES_PER_JOB = {128: 8, 64: 2, 32: 1, 16: 1, 8: 1}   # vtmhip_pic_params.wavesPerJob per PU size: measured with the levels' chains overlapping


FULL_WAVES_PER_JOB = {128: 16, 64: 8, 32: 4, 16: 1, 8: 1}   # 81-point exhaustive search: one candidate at a time per wave for big PUs


def subshift_mode2(w, h):
    """RdCost::setDistParam subShiftMode 2 (FEN=1 / FastSearch, RdCost.cpp:311-317)."""
    return 1 if (h > 8 and w <= 64) else 0


def ctu_bands(pic_w, pic_h, world, ctu=128, unit="ctu"):
    """Partition of a picture's CTUs (raster order) into `world` contiguous bands, one per rank -- the CTU-row sharding of one frame
    (the reference's row loop: EncSlice.cpp:1519-1720).  unit "row": whole CTU rows (17 rows of a 4K picture over 8 ranks
    = 3,2,2,... rows: the slowest rank caps the speed-up at 17 / (8 * 3) = 71 %); unit "ctu": the boundary row is cut at a CTU, i.e.
    raster-scan CTU ranges as VVC's raster-scan slices (510 CTUs over 8 ranks = 64 / 63 CTUs: 99.6 %).
    Returns [(first_ctu, last_ctu_exclusive)] in raster-scan CTU addresses."""
    cw, ch = -(-pic_w // ctu), -(-pic_h // ctu)
    n = ch if unit == "row" else cw * ch
    cuts = [(n * r) // world for r in range(world + 1)]
    return [(a * cw, b * cw) if unit == "row" else (a, b) for a, b in zip(cuts[:-1], cuts[1:])]


def band_filter(pic_w, band, ctu=128):
    """ctu_filter for quadtree_levels: keeps the PUs whose CTU address lies in band = (first, last_exclusive)"""
    cw = -(-pic_w // ctu)
    return lambda cy, cx: (cy * cw + cx >= band[0]) & (cy * cw + cx < band[1])


def quadtree_levels(pic_w, pic_h, sizes=(128, 64, 32, 16, 8), row_filter=None, ctu_filter=None, split_last=None):
    """PUs of every partition level that lie fully inside the picture.  sizes: one entry per level, coarse to fine, an int (square PUs: the quadtree)
    or (w, h) (the binary / ternary split shapes; a level's blocks nest inside the previous level's).  row_filter(ctu_row_array) -> bool array selects the
    CTU rows (128 luma rows each) this rank owns; ctu_filter(ctu_row_array, ctu_col_array) -> bool array selects single CTUs.
    split_last = K > 1: the LAST (finest, largest) level comes as K levels over K contiguous raster ranges of its PUs, all children of the level before it -- a level-order
    driver then runs the remaining stages of the first part beside the searches of the second (its tail after the dependent chain of searches is 1 / K as long).
    Returns [(size as given, xs, ys, parent_index_or_None)], coarse to fine; with split_last given (1: no split) every entry carries a fifth element, the index of its
    parent's entry in this list (or -1)."""
    levels = []
    prev = None
    for si, s in enumerate(sizes):
        bw, bh = (s, s) if np.isscalar(s) else s
        ys, xs = np.mgrid[0:pic_h - bh + 1:bh, 0:pic_w - bw + 1:bw]
        xs, ys = xs.ravel().astype(np.int64), ys.ravel().astype(np.int64)
        if row_filter is not None:
            keep = np.asarray(row_filter(ys // 128), dtype=bool)
            xs, ys = xs[keep], ys[keep]
        if ctu_filter is not None:
            keep = np.asarray(ctu_filter(ys // 128, xs // 128), dtype=bool)
            xs, ys = xs[keep], ys[keep]
        parent = None
        if prev is not None:
            (pw, ph), pxs, pys = prev
            lut = {(int(x), int(y)): i for i, (x, y) in enumerate(zip(pxs, pys))}
            parent = np.array([lut.get((int(x) // pw * pw, int(y) // ph * ph), -1) for x, y in zip(xs, ys)], dtype=np.int64)
        pl = len(levels) - 1 if prev is not None else -1
        if split_last is None:
            levels.append((s, xs, ys, parent))
        else:
            k = split_last if (si == len(sizes) - 1 and split_last > 1 and xs.size >= 64 * split_last) else 1
            for q in range(k):
                a, b = xs.size * q // k, xs.size * (q + 1) // k
                levels.append((s, xs[a:b], ys[a:b], None if parent is None else parent[a:b], pl))
        prev = ((bw, bh), xs, ys)
    return levels


# ======================================================================================================================
# Level-order InterSearch::predInterSearch (InterSearch.cpp:2245-3065, translational part) + residual coding of one picture
# ======================================================================================================================
from .lib import MAX_REF, AffineMeJob, AffineMeOut, FracJob, FracResult, MeCfg, MeJob, MeOut, PisBuffers, PisLevel, PisLevelRun, PisPu, PisRow, PredJob, SmvdJob, TuJob   # noqa: E402

AFF_DT, AFFOUT_DT = np.dtype(AffineMeJob), np.dtype(AffineMeOut)
SMVD_DT = np.dtype(SmvdJob)
MTS_INTER_MAX_CAND = 4      # cfg MTSInterMaxCand (encoder_randomaccess_vtm.cfg / encoder_lowdelay_P_vtm.cfg)

FRAC_DT, FRACRES_DT = np.dtype(FracJob), np.dtype(FracResult)
TU_DT = np.dtype(TuJob)
PRED_DT = np.dtype(PredJob)
ME_DT, MEOUT_DT, ROW_DT, PU_DT = np.dtype(MeJob), np.dtype(MeOut), np.dtype(PisRow), np.dtype(PisPu)


class _Tab:
    """A job table in HBM: uint8 [n, itemsize] torch tensor with typed column views for on-device patching."""

    def __init__(self, torch, device, arr):
        self.n = arr.size
        self.t = torch.from_numpy(arr.view(np.uint8).reshape(arr.size, arr.dtype.itemsize).copy()).to(device)
        self.dt = arr.dtype
        self.torch = torch

    def col(self, field):
        off = self.dt.fields[field][1]
        kind = self.dt.fields[field][0]
        return self._view(kind, off)

    def col2(self, field, k):
        """k-th scalar of an array field (row-major), e.g. col2("mv", 2 * list + component)."""
        kind, off = self.dt.fields[field][0], self.dt.fields[field][1]
        while kind.subdtype is not None:      # nested C arrays arrive as nested sub-array dtypes
            kind = kind.subdtype[0]
        return self._view(kind, off + k * kind.itemsize)

    def _view(self, kind, off):
        tt = {np.dtype(np.int64): self.torch.int64, np.dtype(np.int32): self.torch.int32, np.dtype(np.int16): self.torch.int16,
              np.dtype(np.uint8): self.torch.uint8}[kind]
        assert off % kind.itemsize == 0
        return self.t.view(tt)[:, off // kind.itemsize]

    @property
    def ptr(self):
        return self.t.data_ptr()


TRSKIP = 3
MTS_IDX_TYPES = {0: (0, 0), 1: (TRSKIP, TRSKIP), 2: (2, 2), 3: (1, 2), 4: (2, 1), 5: (1, 1)}


def asr_search_range(delta_poc, search_range=384, min_window=96):
    """EncSlice.cpp:1127 (ASR): Clip3( MinSearchWindow, SearchRange, (SearchRange * ADAPT_SR_SCALE * |dPOC| + 8) / 16 )"""
    return int(min(search_range, max(min_window, (search_range * abs(delta_poc) + 8) // 16)))


def chroma_qp(qp, q_in=(17, 22, 34, 42), q_out=(17, 23, 35 , 39)):
    """ChromaQpMappingTable::derivedChromaQPMappingTables (Slice.cpp:2851-2892) with the table of encoder_randomaccess_vtm.cfg:96-97
    (QpInValCb 17 22 34 42 -> QpOutValCb 17 23 35 39, same table for Cb / Cr, CbQpOffset = CrQpOffset = 0)"""
    tab = {q_in[0]: q_out[0]}
    for k in range(q_in[0] - 1, -13, -1):
        tab[k] = tab[k + 1] - 1
    for j in range(len(q_in) - 1):
        d = q_in[j + 1] - q_in[j]
        for m, k in enumerate(range(q_in[j] + 1, q_in[j + 1] + 1), 1):
            tab[k] = tab[q_in[j]] + ((q_out[j + 1] - q_out[j]) * m + (d >> 1)) // d
    for k in range(q_in[-1] + 1, 64):
        tab[k] = min(63, tab[k - 1] + 1)
    return tab[qp]


class FrameHotPath:
    """All stages of one inter picture, level by level over the quadtree of square PUs (128 .. 8):

      amvp    xEstimateMvPredAMVP per (PU, list, refIdx): template cost of the two AMVP candidates            (InterSearch.cpp:3088-3128)
      uni     xMotionEstimation per (PU, list, refIdx): xTZSearch from the chosen predictor, xPatternSearchFracDIF (SATD), rate re-weighting
              (:3299-3494); xCheckBestMVP; best reference picture per list                                     (:2354-2450)
      bi      B slices, FEN: the list with the larger cost is refined for EVERY reference picture against the other list's prediction
              (motionCompensation -> 2*org - pred, +-4 xPatternSearch, fractional search, fWeight 0.5); xCheckBestMVP (:2452-2640)
      decide  uiCostBi <= uiCost[0], uiCost[1] ? bi : the cheaper list                                         (:2846-2893)
      resi    chosen prediction (uni, or addAvg of two 14-bit predictions) -> residual -> per TU (<= 64x64) and transform candidate
              (DCT2, optionally transform skip, 4 MTS pairs up to 32x32): xT, Quant::quant, dequant, xIT, SSE    (:6637-6733)

    The reference visits one PU at a time inside the CU recursion; here every step is one launch over all PUs of a level x reference pictures
    (C ABI: vtmhip_xEstimateMvPredAMVP_batch_dev, vtmhip_xMotionEstimation_batch_dev, vtmhip_motion_compensation_batch_dev,
    vtmhip_tu_chain_batch_dev, glue: vtmhip_pis_stage), tables and decisions stay in HBM.  Stand-ins for what needs the CU recursion (out of
    scope): the AMVP candidates of a PU are its parent block's vector for the same (list, refIdx) and the zero vector, there
    is no m_uniMvList history, CABAC bit estimates and the mode decision between transform candidates stay with the host.

    refs: ([(ref_off, ref_stride)] list 0, [...] list 1) -- list 1 empty: P slice (uni-prediction only, as encoder_lowdelay_P_vtm.cfg).
    search_ranges: per list, per reference picture (m_aaiAdaptSR).
    pocs: (current POC, [list-0 POCs], [list-1 POCs]) -- enables BDOF in the final prediction of bi-predicted PUs where xPredInterBi applies it.
    chroma: dict(org_off=(Cb, Cr sample offsets of the original chroma planes behind the luma plane in the original buffer), org_stride=..., refs=([(Cb off, Cr off)]
    per list, per reference picture, inside the reference buffer), ref_stride=...) -- adds the 4:2:0 chroma planes to the final prediction, the residual and
    the TU chains (DCT2, chroma QP by the CTC mapping table).
    affine: adds the affine uni stage for PUs of at least 16x16: InterSearch::xAffineMotionEstimation (4-parameter model, uni-directional) per (PU, list, refIdx),
    started from and predicted by the row's translational result; low_delay: getIntraPeriod() == -1 / getCheckLDC() of the low-delay configurations.
    smvd: (refIdx in list 0, refIdx in list 1) = slice.getSymRefIdx() of a slice with getBiDirPred(): adds the symmetric-MVD block of predInterSearch (:2656-2790)
    between the bi refinement and the uni / bi decision (one vtmhip_smvd_batch_dev search per PU; one more bit on the bi rows, :2590-2593)."""

    def __init__(self, ctx, torch, device, pic_w, pic_h, org_stride, refs, search_ranges, motion_lambda=8.0, qp=32, sizes=(128, 64, 32, 16, 8),
                 ctu_filter=None, transform_skip=False, bit_depth=10, pocs=None, chroma=None, bdof=True, affine=False, low_delay=False, smvd=None, split_last=None):
        T, dev = torch, device
        self.ctx, self.torch, self.device = ctx, T, dev
        self.pic_w, self.pic_h, self.org_stride, self.lam = pic_w, pic_h, org_stride, motion_lambda
        self.refs = refs
        nref = [len(refs[0]), len(refs[1])]
        assert 1 <= nref[0] <= MAX_REF and nref[1] <= MAX_REF
        assert nref[1] == 0 or nref[1] == nref[0], "B slices: equal numbers of active reference pictures per list (as the CTC GOP tables have)"
        self.nref, self.is_b = nref, nref[1] > 0
        R = nref[0] + nref[1]
        rs = refs[0][0][1]
        assert all(r[1] == rs for l in refs for r in l)
        self.rs = rs
        base_qp = qp + 6 * (bit_depth - 8)
        self.qp_per, self.qp_rem = base_qp // 6, base_qp % 6
        self.bd = bit_depth
        wpj = dict(WAVES_PER_JOB)
        for kv in filter(None, os.environ.get("VTM_AMD_TZ_WPJ", "").split(",")):
            k, v = kv.split(":")
            wpj[int(k)] = int(v)
        cands = [0] + ([1] if transform_skip else []) + [2, 3, 4, 5]
        self.pocs, self.chroma = pocs, chroma
        self.affine = bool(affine)
        self.smvd = tuple(smvd) if (smvd is not None and nref[1] > 0) else None
        self.bdof = bool(bdof and pocs is not None and self.is_b)
        if chroma is not None:
            cqp = chroma_qp(qp) + 6 * (bit_depth - 8)
            self.cqp_per, self.cqp_rem = cqp // 6, cqp % 6
        self.levels = []
        sb = 0
        # VTM_AMD_SPLIT_LAST=K (default 1 = off): the finest level as K levels over raster ranges of its PUs -- part 1's remaining stages run beside part 2's searches, and the tail
        # behind the dependent chain of searches (the last level's bi refinement / SMVD / prediction / TU chains) would be 1 / K as long.  Measured (round 4, profiles/
        # r04_split_last.txt): 12.12 / 12.22 / 12.29 / 12.50 ms for K = 1 .. 4 -- the picture is bound by the kernels' total time on the machine, not by that tail; off by default
        split_last = int(os.environ.get("VTM_AMD_SPLIT_LAST", "1")) if split_last is None else int(split_last)
        built = {}      # index in quadtree_levels' list -> the level record (empty levels are not built)
        for qi, (s, xs, ys, parent, parent_q) in enumerate(quadtree_levels(pic_w, pic_h, sizes, None, ctu_filter, split_last)):
            n = xs.size
            prev = built.get(parent_q)
            if n == 0:
                continue
            w, h = (s, s) if np.isscalar(s) else s                   # a level of the quadtree, or of a binary / ternary split shape
            lvl = dict(size=s, w=w, h=h, npu=n, xs=xs, ys=ys, sb=sb, parent_level=next((i for i, l in enumerate(self.levels) if l is prev), -1))
            built[qi] = lvl
            blk = sb + np.arange(n, dtype=np.int64) * w * h          # compact per-PU slots of the level-wide sample buffers
            sb += n * w * h
            uj = np.zeros(R * n, ME_DT)
            for l in (0, 1):
                for r in range(nref[l]):
                    sl = slice(((nref[0] if l else 0) + r) * n, ((nref[0] if l else 0) + r + 1) * n)
                    uj["refOff"][sl] = refs[l][r][0] + ys * rs + xs
                    uj["searchRange"][sl] = search_ranges[l][r]
            uj["orgOff"], uj["orgStride"], uj["refStride"] = np.tile(ys * org_stride + xs, R), org_stride, rs
            uj["puX"], uj["puY"], uj["width"], uj["height"] = np.tile(xs, R), np.tile(ys, R), w, h
            uj["motionLambda"], uj["numAmvpCand"] = motion_lambda, 2
            lvl["uni_jobs"] = _Tab(T, dev, uj)
            lvl["uni_out"] = T.zeros((R * n, MEOUT_DT.itemsize), dtype=T.uint8, device=dev)
            lvl["uni_rows"] = T.zeros((R * n, ROW_DT.itemsize), dtype=T.uint8, device=dev)
            lvl["pus"] = T.zeros((n, PU_DT.itemsize), dtype=T.uint8, device=dev)
            pj = np.zeros(n, PRED_DT)
            pj["orgOff"], pj["orgStride"], pj["refStride"] = ys * org_stride + xs, org_stride, rs
            pj["predOff"], pj["outOff"], pj["predStride"], pj["outStride"] = blk, blk, w, w
            pj["width"], pj["height"], pj["bitDepth"] = w, h, bit_depth
            pf = pj.copy()
            pf["epilogue"] = 1                                         # residual = org - pred
            lvl["pred_final"] = _Tab(T, dev, pf)
            lvl["pos"] = T.from_numpy(ys * rs + xs).to(dev)
            if self.is_b:
                po = pj.copy()
                po["epilogue"] = 2                                     # bi-pred search pattern 2*org - pred (removeHighFreq)
                lvl["pred_other"] = _Tab(T, dev, po)
                nb = nref[0]
                bj = np.zeros(nb * n, ME_DT)
                bj["orgOff"], bj["orgStride"] = np.tile(ys * org_stride + xs, nb), org_stride
                bj["otherPredOff"], bj["otherPredStride"] = np.tile(blk, nb), w
                bj["puX"], bj["puY"], bj["width"], bj["height"] = np.tile(xs, nb), np.tile(ys, nb), w, h
                lvl["bi_jobs"] = _Tab(T, dev, bj)
                lvl["bi_out"] = T.zeros((nb * n, MEOUT_DT.itemsize), dtype=T.uint8, device=dev)
            par32 = None
            if parent is not None and prev is not None:
                par32 = T.from_numpy(parent.astype(np.int32)).to(dev)
            lvl["parent32"] = par32
            L = PisLevel()
            L.numPU, L.smvdBit, L.refStride = n, int(self.smvd is not None), rs
            L.numRef[0], L.numRef[1] = nref
            L.mbBits[0], L.mbBits[1], L.mbBits[2] = (3 if self.is_b else 1), 3, 5      # xGetBlkBits (:3164-3169)
            for l in (0, 1):
                for r in range(nref[l]):
                    L.refPlaneOff[l][r] = int(refs[l][r][0])
            L.uniJobs, L.uniOut, L.uniRows, L.pus = lvl["uni_jobs"].ptr, lvl["uni_out"].data_ptr(), lvl["uni_rows"].data_ptr(), lvl["pus"].data_ptr()
            L.predFinal, L.pos = lvl["pred_final"].ptr, lvl["pos"].data_ptr()
            if self.is_b:
                L.predOther, L.biJobs, L.biOut = lvl["pred_other"].ptr, lvl["bi_jobs"].ptr, lvl["bi_out"].data_ptr()
            if par32 is not None:
                L.parentIdx, L.parentRows, L.parentNumPU = par32.data_ptr(), prev["uni_rows"].data_ptr(), prev["npu"]
            if pocs is not None:
                L.bdofEnabled, L.curPoc = int(self.bdof), int(pocs[0])
                for l in (0, 1):
                    for r in range(nref[l]):
                        L.refPoc[l][r] = int(pocs[1 + l][r])
            if chroma is not None:
                # ---- the two 4:2:0 chroma planes of every PU: prediction + residual jobs (Cb jobs, then Cr jobs), compact slots in level-wide chroma buffers ----
                wc, hc, rsc, osc = w // 2, h // 2, chroma["ref_stride"], chroma["org_stride"]
                blk_c = (sb - n * w * h) // 4 + np.arange(n, dtype=np.int64) * wc * hc
                cj = np.zeros(2 * n, PRED_DT)
                for c in (0, 1):
                    sl = slice(c * n, (c + 1) * n)
                    cj["orgOff"][sl] = chroma["org_off"][c] + (ys // 2) * osc + xs // 2
                    cj["predOff"][sl] = cj["outOff"][sl] = blk_c      # + c * NSC (the Cr half of the chroma buffers), added below once NSC is known
                cj["orgStride"], cj["refStride"], cj["predStride"], cj["outStride"] = osc, rsc, wc, wc
                cj["width"], cj["height"], cj["bitDepth"], cj["chroma"], cj["epilogue"] = wc, hc, bit_depth, 1, 1
                lvl["blk_c"] = blk_c
                lvl["pred_final_c"] = _Tab(T, dev, cj)
                lvl["pos_c"] = T.from_numpy((ys // 2) * rsc + xs // 2).to(dev)
                L.predFinalC, L.posC = lvl["pred_final_c"].ptr, lvl["pos_c"].data_ptr()
                for c in (0, 1):
                    for l in (0, 1):
                        for r in range(nref[l]):
                            L.refPlaneOffC[c][l][r] = int(chroma["refs"][l][r][c])
                twc, thc = min(wc, 32), min(hc, 32)
                tu_src_c = np.stack([blk_c + qy * thc * wc + qx * twc for qy in range(hc // thc) for qx in range(wc // twc)], 1).reshape(-1)
                ntc = tu_src_c.size
                tc = np.zeros(2 * ntc, TU_DT)       # Cb TUs, then Cr TUs (the Cr plane's buffers start NSC samples further: added in _finish_chroma)
                tc["resiOff"], tc["resiStride"], tc["width"], tc["height"] = np.tile(tu_src_c, 2), wc, twc, thc
                tc["outOff"] = np.arange(2 * ntc, dtype=np.int64) * twc * thc
                tc["qpPer"], tc["qpRem"], tc["bitDepth"] = self.cqp_per, self.cqp_rem, bit_depth
                lvl.update(ntu_c=ntc, ts_c=twc, tw_c=twc, th_c=thc, tu_c_np=tc, tu_res_c=T.zeros((2 * ntc, 2), dtype=T.int64, device=dev),
                           qcoef_c=T.zeros(2 * ntc * twc * thc, dtype=T.int32, device=dev))
            if self.affine and min(w, h) >= 16:
                # ---- affine uni stage: one xAffineMotionEstimation job per row, written by stage 4 from the translational results ----
                lvl["aff_jobs"] = T.zeros((R * n, AFF_DT.itemsize), dtype=T.uint8, device=dev)
                lvl["aff_out"] = T.zeros((R * n, AFFOUT_DT.itemsize), dtype=T.uint8, device=dev)
                L.affJobs, L.affLowDelay, L.affCheckLDC = lvl["aff_jobs"].data_ptr(), int(low_delay), int(low_delay)
            if self.smvd is not None and w + h > 12:
                lvl["smvd_jobs"] = T.zeros((n, SMVD_DT.itemsize), dtype=T.uint8, device=dev)
                L.smvdJobs, L.symRefIdx[0], L.symRefIdx[1] = lvl["smvd_jobs"].data_ptr(), self.smvd[0], self.smvd[1]
            lvl["pis"] = L
            lvl["pic"] = PicParams(pic_w, pic_h, 128, bit_depth, wpj.get(max(w, h), 1), max(max(l) for l in search_ranges if len(l)))      # maxSearchRange: sizes the raster scans' LDS totals
            lvl["pic_bi"] = PicParams(pic_w, pic_h, 128, bit_depth, FULL_WAVES_PER_JOB.get(max(w, h), 1))
            lvl["cfg_uni"] = MeCfg(4, 1, 1, 0, 1, 0, 1, 1, 1, 0)           # BipredSearchRange 4, HadamardME, FEN, uniform: imv 0, square, all uni, no m_uniMvList
            lvl["cfg_bi"] = MeCfg(4, 1, 1, 0, 1, 0, 1, 2, 1, 1)            # all bi, the pattern 2*org - pred comes from the fused MC epilogue
            # ---- transform units: the PU itself up to 64x64, four 64x64 quadrants of a 128x128 PU (MaxTbSize 64) -------------------------
            tw, th = min(w, 64), min(h, 64)
            tu_src = np.stack([blk + qy * th * w + qx * tw for qy in range(h // th) for qx in range(w // tw)], 1).reshape(-1)
            ntu = tu_src.size
            cl = [c for c in cands if c == 0 or (max(tw, th) <= 32)]
            nc = len(cl)
            tj = np.zeros(ntu * nc, TU_DT)
            tj["resiOff"], tj["resiStride"], tj["width"], tj["height"] = np.tile(tu_src, nc), w, tw, th
            tj["outOff"] = np.arange(ntu * nc, dtype=np.int64) * tw * th
            tj["qpPer"], tj["qpRem"], tj["bitDepth"] = self.qp_per, self.qp_rem, bit_depth
            tj["typeHor"] = np.repeat([MTS_IDX_TYPES[c][0] for c in cl], ntu)
            tj["typeVer"] = np.repeat([MTS_IDX_TYPES[c][1] for c in cl], ntu)
            lvl.update(ntu=ntu, nc=nc, ts=tw, tw=tw, th=th, cands=cl, tu=_Tab(T, dev, tj), tu_res=T.zeros((ntu * nc, 2), dtype=T.int64, device=dev), mts_test=T.zeros(ntu * nc, dtype=T.uint8, device=dev),
                       qcoef=T.zeros(ntu * nc * tw * th, dtype=T.int32, device=dev))
            self.levels.append(lvl)
        self.NS = sb
        self.NP = sum(l["npu"] for l in self.levels)
        self.n_me_jobs = sum(l["npu"] * (R + (nref[0] if self.is_b else 0)) for l in self.levels)
        mk = lambda: T.zeros(max(1, sb), dtype=T.int16, device=dev)   # noqa: E731
        self.buf = dict(pred=mk(), resi=mk())
        if self.is_b:
            self.buf["org_bi"] = mk()
        if chroma is not None:
            nsc = sb // 4                                            # samples of ONE chroma plane over all levels; buffers hold Cb then Cr
            self.NSC = nsc
            self.buf["pred_c"] = T.zeros(max(1, 2 * nsc), dtype=T.int16, device=dev)
            self.buf["resi_c"] = T.zeros(max(1, 2 * nsc), dtype=T.int16, device=dev)
            for lvl in self.levels:
                n = lvl["npu"]
                t = lvl["pred_final_c"]
                for f in ("predOff", "outOff"):
                    t.col(f)[n:] += nsc
                tc = lvl.pop("tu_c_np")
                tc["resiOff"][lvl["ntu_c"]:] += nsc
                lvl["tu_c"] = _Tab(T, dev, tc)
        self.side_streams = [T.cuda.Stream(device=dev) for _ in range(int(os.environ.get("VTM_AMD_SIDE_STREAMS", "5")))] if dev.type == "cuda" else []
        self._marks = None
        # ---- the same chain as a table for the native loop (vtmhip_pis_run_picture: one library call per picture instead of ~130 from Python) ----
        self.run_levels = (PisLevelRun * max(1, len(self.levels)))()
        for k, lvl in enumerate(self.levels):
            r = self.run_levels[k]
            C.memmove(C.byref(r.pis), C.byref(lvl["pis"]), C.sizeof(PisLevel))
            r.pic, r.picBi, r.cfgUni, r.cfgBi = lvl["pic"], lvl["pic_bi"], lvl["cfg_uni"], lvl["cfg_bi"]
            r.width, r.height, r.bdof = lvl["w"], lvl["h"], int(self.bdof and lvl["w"] * lvl["h"] >= 128)
            r.uniOut = lvl["uni_out"].data_ptr()
            r.biOut = lvl["bi_out"].data_ptr() if self.is_b else None
            r.tu, r.tuRes, r.qcoef = lvl["tu"].ptr, lvl["tu_res"].data_ptr(), lvl["qcoef"].data_ptr()
            r.numTU, r.numCands, r.tuW, r.tuH = lvl["ntu"], lvl["nc"], lvl["tw"], lvl["th"]
            r.mtsTest, r.mtsMaxCand = lvl["mts_test"].data_ptr(), MTS_INTER_MAX_CAND
            for i, c in enumerate(lvl["cands"]):
                r.cand[i] = c
            if chroma is not None:
                r.tuC, r.tuResC, r.qcoefC = lvl["tu_c"].ptr, lvl["tu_res_c"].data_ptr(), lvl["qcoef_c"].data_ptr()
                r.numTUC, r.tuWC, r.tuHC = lvl["ntu_c"], lvl["tw_c"], lvl["th_c"]
            if "aff_out" in lvl:
                r.affOut = lvl["aff_out"].data_ptr()

    # ---- stage timing (HIP events on the launch stream; only when run(..., timing=True)) ----------------------------------
    def _mark(self, name):
        if self._marks is not None:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            self._marks.append((name, e))

    def stage_ms(self):
        acc = {}
        for (n0, e0), (n1, e1) in zip(self._marks[:-1], self._marks[1:]):
            acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
        acc.pop("start", None)
        return acc

    # ---- the steps of one level ---------------------------------------------------------------------------------------------
    def _uni(self, lvl, org_ptr, dpb_ptr):
        ctx, n, w, h = self.ctx, lvl["npu"], lvl["w"], lvl["h"]
        rows = (self.nref[0] + self.nref[1]) * n
        ctx.pis_stage(lvl["pis"], 0)
        self._mark("glue")
        ctx.estimate_mvp_amvp_batch(lvl["pic"], org_ptr, dpb_ptr, lvl["uni_jobs"].ptr, rows, w, h, uniform=True)
        self._mark("amvp")
        ctx.motion_estimation_batch(lvl["pic"], lvl["cfg_uni"], org_ptr, dpb_ptr, None, lvl["uni_jobs"].ptr, rows, w, h, lvl["uni_out"].data_ptr())
        self._mark("uni_me")
        ctx.pis_stage(lvl["pis"], 1)
        self._mark("glue")

    def _rest(self, lvl, org_ptr, dpb_ptr):
        ctx, n, w, h, buf = self.ctx, lvl["npu"], lvl["w"], lvl["h"], self.buf
        if self.is_b:
            ctx.pis_stage(lvl["pis"], 2)
            self._mark("glue")
            ctx.motion_compensation_batch(org_ptr, dpb_ptr, None, buf["org_bi"].data_ptr(), lvl["pred_other"].ptr, n, w, h)
            self._mark("mc")
            ctx.motion_estimation_batch(lvl["pic_bi"], lvl["cfg_bi"], org_ptr, dpb_ptr, buf["org_bi"].data_ptr(), lvl["bi_jobs"].ptr, self.nref[0] * n, w, h,
                                        lvl["bi_out"].data_ptr())
            self._mark("bi_me")
            ctx.pis_stage(lvl["pis"], 3)
            self._mark("glue")
            if "smvd_jobs" in lvl:
                ctx.smvd_batch(lvl["pic"], org_ptr, dpb_ptr, lvl["smvd_jobs"].data_ptr(), n, w, h, 3, uniform=True)
                self._mark("smvd")
                ctx.pis_stage(lvl["pis"], 5)
                self._mark("glue")
        ctx.motion_compensation_batch(org_ptr, dpb_ptr, buf["pred"].data_ptr(), buf["resi"].data_ptr(), lvl["pred_final"].ptr, n, w, h)
        if self.bdof and w * h >= 128:      # the PUs the final stage routed to BDOF (bi-prediction from opposite directions at equal POC distance; 8x8 never qualifies)
            ctx.bdof_batch(org_ptr, dpb_ptr, buf["pred"].data_ptr(), buf["resi"].data_ptr(), lvl["pred_final"].ptr, n, w, h)
        if self.chroma is not None:
            ctx.motion_compensation_batch(org_ptr, dpb_ptr, buf["pred_c"].data_ptr(), buf["resi_c"].data_ptr(), lvl["pred_final_c"].ptr, 2 * n, w // 2, h // 2)
        self._mark("mc")
        if "aff_jobs" in lvl:
            rows = (self.nref[0] + self.nref[1]) * n
            ctx.pis_stage(lvl["pis"], 4)
            ctx.affine_motion_estimation_batch(lvl["pic"], org_ptr, dpb_ptr, None, lvl["aff_jobs"].data_ptr(), rows, w, h, lvl["aff_out"].data_ptr())
            self._mark("affine")
        self._tu(lvl)
        if self.chroma is not None:
            twc, thc = lvl["tw_c"], lvl["th_c"]
            ctx.tu_chain_batch(buf["resi_c"].data_ptr(), lvl["tu_c"].ptr, 2 * lvl["ntu_c"], twc, thc, lvl["tu_res_c"].data_ptr(), lvl["qcoef_c"].data_ptr(), None,
                               uniform=True)      # 4x4 / 8x4 / 4x8: one lane per TU; from 8x8: the register-blocked kernel; 16x4-like shapes: the library's generic kernel
        self._mark("tu")

    def _tu(self, lvl):
        ctx, tw, th, ntu = self.ctx, lvl["tw"], lvl["th"], lvl["ntu"]
        tu_p, res_p, q_p = lvl["tu"].ptr, lvl["tu_res"].data_ptr(), lvl["qcoef"].data_ptr()
        cands, k = lvl["cands"], 0
        while k < len(cands):            # candidates are stored one after the other; transform skip goes to its own (elementwise) kernel
            run = 1
            if cands[k] != 1:
                while k + run < len(cands) and cands[k + run] != 1:
                    run += 1
            a, m = k * ntu, run * ntu
            if cands[k] == 1:
                ctx.tu_ts_chain_batch(self.buf["resi"].data_ptr(), tu_p + a * TU_DT.itemsize, m, tw, th, res_p + a * 16, q_p)
            else:
                ctx.tu_chain_batch(self.buf["resi"].data_ptr(), tu_p + a * TU_DT.itemsize, m, tw, th, res_p + a * 16, q_p, None, uniform=True)
            k += run
        ctx.mts_select_batch(res_p, ntu, cands, tw, th, self.bd, MTS_INTER_MAX_CAND, lvl["mts_test"].data_ptr())

    def run(self, org_ptr, dpb_ptr, timing=False):
        """timing=True (or no side streams / VTM_AMD_OVERLAP=0): every level's steps one after the other on one stream with an event after every
        step; otherwise level-major over the side streams: the uni searches form one dependent chain (a child's AMVP candidate is its parent's vector)
        on the caller's stream, everything after a level's uni stage runs on a side stream beside the next levels' searches."""
        T, ctx = self.torch, self.ctx
        overlapped = not timing and self.side_streams and os.environ.get("VTM_AMD_OVERLAP", "1") != "0"
        self._marks = [] if timing else None
        self._mark("start")
        if not overlapped:
            for lvl in self.levels:
                self._uni(lvl, org_ptr, dpb_ptr)
            for lvl in self.levels:
                self._rest(lvl, org_ptr, dpb_ptr)
            return
        main = T.cuda.current_stream()
        if os.environ.get("VTM_AMD_NATIVE_LOOP", "1") != "0":      # the native loop: the same calls in the same order on the same streams
            b = PisBuffers(org_ptr, dpb_ptr, self.buf["pred"].data_ptr(), self.buf["resi"].data_ptr(), self.buf["org_bi"].data_ptr() if self.is_b else None,
                           self.buf["pred_c"].data_ptr() if self.chroma is not None else None, self.buf["resi_c"].data_ptr() if self.chroma is not None else None)
            ctx.pis_run_picture(self.run_levels, len(self.levels), b, main.cuda_stream, [st.cuda_stream for st in self.side_streams])
            return
        for i, lvl in enumerate(self.levels):
            self._uni(lvl, org_ptr, dpb_ptr)
            ev = T.cuda.Event()
            ev.record(main)
            st = self.side_streams[i % len(self.side_streams)]
            st.wait_event(ev)
            ctx.set_stream(st.cuda_stream)
            self._rest(lvl, org_ptr, dpb_ptr)
            ctx.set_stream(main.cuda_stream)
        for st in self.side_streams:
            main.wait_stream(st)

    # ---- results ----------------------------------------------------------------------------------------------------------------
    def snapshot(self):
        """numpy copies of every decision, per level (tests/cpu_chain.py, a host encoder, the multi-GPU gather read these)"""
        out = []
        for lvl in self.levels:
            d = dict(size=lvl["size"], parent_level=lvl["parent_level"], w=lvl["w"], h=lvl["h"], tw=lvl["tw"], th=lvl["th"], npu=lvl["npu"], ntu=lvl["ntu"], nc=lvl["nc"], ts=lvl["ts"], cands=lvl["cands"], xs=lvl["xs"], ys=lvl["ys"],
                     uni_jobs=lvl["uni_jobs"].t.cpu().numpy().view(ME_DT).reshape(-1), uni_out=lvl["uni_out"].cpu().numpy().view(MEOUT_DT).reshape(-1),
                     uni_rows=lvl["uni_rows"].cpu().numpy().view(ROW_DT).reshape(-1), pus=lvl["pus"].cpu().numpy().view(PU_DT).reshape(-1),
                     tu_res=lvl["tu_res"].cpu().numpy(), mts_test=lvl["mts_test"].cpu().numpy())
            if self.is_b:
                d["bi_jobs"] = lvl["bi_jobs"].t.cpu().numpy().view(ME_DT).reshape(-1)
                d["bi_out"] = lvl["bi_out"].cpu().numpy().view(MEOUT_DT).reshape(-1)
            d["route"] = lvl["pred_final"].col("route").cpu().numpy()
            if "smvd_jobs" in lvl:
                d["smvd_jobs"] = lvl["smvd_jobs"].cpu().numpy().view(SMVD_DT).reshape(-1)
            if "aff_jobs" in lvl:
                d["aff_jobs"] = lvl["aff_jobs"].cpu().numpy().view(AFF_DT).reshape(-1)
                d["aff_out"] = lvl["aff_out"].cpu().numpy().view(AFFOUT_DT).reshape(-1)
            if self.chroma is not None:
                d.update(ntu_c=lvl["ntu_c"], ts_c=lvl["ts_c"], tw_c=lvl["tw_c"], th_c=lvl["th_c"], tu_res_c=lvl["tu_res_c"].cpu().numpy())
            out.append(d)
        return out

    def result_tensors(self):
        """the per-PU / per-TU result tensors a rank hands to rank 0 (bytes views), coarse to fine"""
        out = []
        for lvl in self.levels:
            out += [lvl["pus"].reshape(-1), lvl["tu_res"].view(self.torch.uint8).reshape(-1), lvl["mts_test"]]
            if self.chroma is not None:
                out.append(lvl["tu_res_c"].view(self.torch.uint8).reshape(-1))
            if "aff_out" in lvl:
                out.append(lvl["aff_out"].reshape(-1))
        return out

    def sad_candidates(self, org_ptr, dpb_ptr):
        """After run(): the integer-search candidates every level's uni searches evaluated (sum of vtmhip_me_result.nEval, the statistic the TZ kernel keeps): the same searches
        once more through vtmhip_tz_search_batch_dev on job records rebuilt from the rows (predictor as chosen by the AMVP stage, FEN row sub-sampling).  SURVEY.md 8(d) prices a
        SAD candidate at 4 * W * H >> subShift bytes: bench.py reports sum( candidates * that ) per picture beside the search kernels' time.  -> [(w, h, subShift, searches, candidates)]"""
        T, out = self.torch, []
        tz_dt, res_dt = np.dtype(TzJob), np.dtype(MeResult)
        for lvl in self.levels:
            uj = lvl["uni_jobs"].t.cpu().numpy().view(ME_DT).reshape(-1)
            n, w, h = uj.size, lvl["w"], lvl["h"]
            ss = subshift_mode2(w, h)
            tj = np.zeros(n, tz_dt)
            for f in ("orgOff", "refOff", "orgStride", "refStride", "puX", "puY", "width", "height", "motionLambda", "searchRange"):
                tj[f] = uj[f]
            down = lambda v: np.where(v >= 0, (v + 1) >> 2, (v + 2) >> 2)      # noqa: E731  Mv::changePrecision( INTERNAL -> QUARTER )
            tj["subShift"], tj["predHor"], tj["predVer"] = ss, down(uj["mvPredHor"].astype(np.int64)), down(uj["mvPredVer"].astype(np.int64))
            tj["mvHor"], tj["mvVer"], tj["firstSearchStop"] = uj["mvPredHor"], uj["mvPredVer"], 1
            d_j = T.from_numpy(tj.view(np.uint8)).to(self.device)
            d_r = T.zeros(n * res_dt.itemsize, dtype=T.uint8, device=self.device)
            self.ctx.tz_search_batch(lvl["pic"], org_ptr, dpb_ptr, d_j.data_ptr(), n, d_r.data_ptr())
            T.cuda.synchronize()
            r = d_r.cpu().numpy().view(res_dt).reshape(-1)
            out.append((w, h, ss, n, int(r["nEval"].astype(np.int64).sum())))
        return out

    def work_counts(self):
        R = self.nref[0] + self.nref[1]
        return dict(pus=self.NP, uni_searches=R * self.NP, bi_searches=(self.nref[0] if self.is_b else 0) * self.NP,
                    tu_chains=sum(l["ntu"] * l["nc"] for l in self.levels), affine_searches=sum(R * l["npu"] for l in self.levels if "aff_jobs" in l),
                    smvd_searches=sum(l["npu"] for l in self.levels if "smvd_jobs" in l))
