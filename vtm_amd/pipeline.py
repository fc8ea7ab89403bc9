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


def quadtree_levels(pic_w, pic_h, sizes=(128, 64, 32, 16, 8), row_filter=None, ctu_filter=None):
    """PUs of every partition level that lie fully inside the picture.  sizes: one entry per level, coarse to fine, an int (square PUs: the quadtree)
    or (w, h) (the binary / ternary split shapes; a level's blocks nest inside the previous level's).  row_filter(ctu_row_array) -> bool array selects the
    CTU rows (128 luma rows each) this rank owns; ctu_filter(ctu_row_array, ctu_col_array) -> bool array selects single CTUs.
    Returns [(size as given, xs, ys, parent_index_or_None)], coarse to fine."""
    levels = []
    prev = None
    for s in sizes:
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
        levels.append((s, xs, ys, parent))
        prev = ((bw, bh), xs, ys)
    return levels


def build_tz_jobs(size, xs, ys, org_stride, ref_off, ref_stride, search_range, motion_lambda, org_off=0):
    """TzJob table for n square PUs against one reference plane (zero start / zero predictor; the driver patches
    the predictor words from the parent level on the device)."""
    n = xs.size
    a = np.zeros(n, dtype=TZ_DT)
    a["orgOff"] = org_off + ys * org_stride + xs
    a["refOff"] = ref_off + ys * ref_stride + xs
    a["orgStride"], a["refStride"] = org_stride, ref_stride
    a["puX"], a["puY"], a["width"], a["height"] = xs, ys, size, size
    a["subShift"] = subshift_mode2(size, size)
    a["motionLambda"] = motion_lambda
    a["searchRange"] = search_range
    a["firstSearchStop"] = 1   # FastMEAssumingSmootherMVEnabled default (EncAppCfg.cpp:981)
    return a


class FrameME:
    """Integer ME of one picture: quadtree levels x reference pictures, device-resident.

    torch tensors: `org` int16 [H*orgStride], `dpb` int16 (all reference planes, border-extended, back to back)."""

    def __init__(self, ctx, torch, device, pic_w, pic_h, org_stride, refs, search_ranges, motion_lambda=8.0,
                 sizes=(128, 64, 32, 16, 8), row_filter=None, waves_per_job=None, ctu_filter=None):
        """refs: [(ref_off, ref_stride)] sample offsets of each reference plane's (0,0) inside `dpb`."""
        self.ctx, self.torch, self.device = ctx, torch, device
        self.pic_w, self.pic_h = pic_w, pic_h
        # waves that share one search: big PUs have long SADs and (at the top level, which has no predictor) raster scans
        self.wpj = dict(WAVES_PER_JOB)
        self.wpj.update(waves_per_job or {})
        for kv in filter(None, os.environ.get("VTM_AMD_TZ_WPJ", "").split(",")):   # tuning knob, e.g. VTM_AMD_TZ_WPJ=128:8,64:4
            k, v = kv.split(":")
            self.wpj[int(k)] = int(v)
        self.levels = []
        self.n_jobs = 0
        self.alg_bytes_per_eval = []
        for (s, xs, ys, parent) in quadtree_levels(pic_w, pic_h, sizes, row_filter, ctu_filter):
            if xs.size == 0:
                continue
            tabs, parents = [], []
            for (roff, rstride), sr in zip(refs, search_ranges):
                tabs.append(build_tz_jobs(s, xs, ys, org_stride, roff, rstride, sr, motion_lambda))
            nref = len(refs)
            jobs = np.concatenate(tabs)
            n = jobs.size
            par = None
            if parent is not None:
                npar = self.levels[-1]["n"] // nref
                par = np.concatenate([np.where(parent >= 0, parent + r * npar, -1) for r in range(nref)])
            lvl = dict(size=s, n=n, pic=PicParams(pic_w, pic_h, 128, 10, self.wpj.get(s, 1)),
                       jobs=torch.from_numpy(jobs.view(np.uint8).reshape(n, TZ_DT.itemsize).copy()).to(device),
                       res=torch.zeros((n, 8), dtype=torch.int32, device=device),
                       parent=None if par is None else torch.from_numpy(par).to(device),
                       parent32=None if par is None else torch.from_numpy(par.astype(np.int32)).to(device))
            self.levels.append(lvl)
            self.n_jobs += n
            self.alg_bytes_per_eval.append(4 * s * s >> subshift_mode2(s, s))

    def run(self, org_ptr, dpb_ptr):
        """Launches every level (coarse to fine) on the context's stream; no host synchronisation."""
        for i, lvl in enumerate(self.levels):
            if lvl["parent"] is not None:   # start vector and MV predictor = the parent's integer vector (device-side patch)
                self.ctx.frame_child_start(lvl["jobs"].data_ptr(), lvl["n"], lvl["parent32"].data_ptr(), self.levels[i - 1]["res"].data_ptr())
            self.ctx.tz_search_batch(lvl["pic"], org_ptr, dpb_ptr, lvl["jobs"].data_ptr(), lvl["n"], lvl["res"].data_ptr())

    def stats(self):
        """(total candidate evaluations, algorithmic bytes = sum over jobs of nEval * 4*W*H >> subShift)."""
        ev, by = 0, 0
        for lvl, b in zip(self.levels, self.alg_bytes_per_eval):
            ne = int(lvl["res"][:, 2].to(self.torch.int64).sum().item())
            ev += ne
            by += ne * b
        return ev, by

    def results_numpy(self):
        return [lvl["res"].cpu().numpy().view(RES_DT).reshape(-1) for lvl in self.levels]


# ======================================================================================================================
# Full hot path of one inter picture: integer ME -> fractional ME -> bi-predictive refinement -> residual coding.
# ======================================================================================================================
from .lib import DistJob, FracJob, FracResult, FrameTabs, FullJob, McJob, PelOpJob, PredJob, QuantJob, TrJob, TuJob   # noqa: E402

FRAC_DT, FRACRES_DT, MC_DT, FULL_DT = np.dtype(FracJob), np.dtype(FracResult), np.dtype(McJob), np.dtype(FullJob)
PEL_DT, TR_DT, Q_DT, DIST_DT = np.dtype(PelOpJob), np.dtype(TrJob), np.dtype(QuantJob), np.dtype(DistJob)
TU_DT = np.dtype(TuJob)
PRED_DT = np.dtype(PredJob)

# (typeHor, typeVer) of mtsIdx 0, 2, 3, 4, 5 (TrQuant::getTrTypes, TrQuant.cpp:695-772): DCT2 = 0, DCT8 = 1, DST7 = 2
MTS_CANDS = ((0, 0), (2, 2), (1, 2), (2, 1), (1, 1))


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


class FrameHotPathV1(FrameME):
    """All stages for one picture with two reference pictures (list 0 / list 1):

      tz      InterSearch::xTZSearch per (PU, list), level by level (children start from the parent's vector)   (InterSearch.cpp:3640-3976)
      frac    xPatternSearchFracDIF per (PU, list): half + quarter refinement, SATD                               (:4284-4339)
      bi      FEN bi-pred iteration (:2531-2680): refine the list with the LARGER uni cost: motion-compensate the other
              list, org' = 2*org - pred (removeHighFreq), +-4 exhaustive search (xPatternSearch), fractional search on org'
      resi    final prediction (bi via addAvg when cheaper, else best uni) -> residual -> per TU (<= 64x64) and transform
              candidate (DCT2 + 4 MTS candidates up to 32x32): xT -> Quant::quant -> dequant -> xIT -> SSE (one fused launch per level)
    Mode decision between the candidates, CABAC bit estimation and DepQuant stay on the host (out of scope);
    every MTS candidate is taken through the whole chain (the reference prunes with the sum|coef| threshold).

    STAGE-MAJOR execution: the integer search runs level by level (parent -> child dependence); every later stage is one
    launch per level over job tables that are concatenated across levels, so the on-device bookkeeping between stages
    (list choice, vector arithmetic, job patching -- a few dozen tiny tensor ops) happens once per picture, not per level.
    """

    def __init__(self, ctx, torch, device, pic_w, pic_h, org_stride, refs, search_ranges, motion_lambda=8.0, qp=32, fused_tu=True, **kw):
        assert len(refs) == 2 and refs[0][1] == refs[1][1]
        super().__init__(ctx, torch, device, pic_w, pic_h, org_stride, refs, search_ranges, motion_lambda, **kw)
        self.fused_tu = fused_tu
        self.refs, self.org_stride, self.lam = refs, org_stride, motion_lambda
        base_qp = qp + 12   # 10-bit: qpBdOffset = 12 (Quant.cpp:65-104)
        self.qp_per, self.qp_rem = base_qp // 6, base_qp % 6
        T, dev, rs = torch, device, refs[0][1]

        # ---- concatenate the per-level TZ tables so that every later stage sees one table ----------------------------------
        NP = sum(l["n"] // 2 for l in self.levels)
        self.NP = NP
        self.tz_jobs_all = T.cat([l["jobs"] for l in self.levels])            # [2*NP, 208] rows: per level [list0 PUs | list1 PUs]
        self.tz_res_all = T.zeros((2 * NP, 8), dtype=T.int32, device=dev)
        row0, row1, pos, blk, sizes = [], [], [], [], []
        pb, sb = 0, 0
        for lvl in self.levels:
            s, npu = lvl["size"], lvl["n"] // 2
            lvl["npu"], lvl["pb"], lvl["sb"] = npu, pb, sb
            lvl["jobs"] = self.tz_jobs_all[2 * pb:2 * pb + 2 * npu]           # views: launches and parent patching act on the big table
            lvl["res"] = self.tz_res_all[2 * pb:2 * pb + 2 * npu]
            lvl["pic_full"] = PicParams(pic_w, pic_h, 128, 10, FULL_WAVES_PER_JOB.get(s, 1))
            jn = lvl["jobs"].cpu().numpy().view(TZ_DT).reshape(-1)
            xs, ys = jn["puX"][:npu].astype(np.int64), jn["puY"][:npu].astype(np.int64)
            lvl["xs"], lvl["ys"] = xs, ys
            row0.append(2 * pb + np.arange(npu))
            row1.append(2 * pb + npu + np.arange(npu))
            pos.append(ys * rs + xs)
            blk.append(sb + np.arange(npu, dtype=np.int64) * s * s)
            sizes.append(np.full(npu, s))
            pb += npu
            sb += npu * s * s
        self.NS = sb                                                            # samples of one full-coverage buffer (all levels)
        row0, row1, pos, blk, sizes = (np.concatenate(a) for a in (row0, row1, pos, blk, sizes))
        xs_all = np.concatenate([l["xs"] for l in self.levels])
        ys_all = np.concatenate([l["ys"] for l in self.levels])
        self.row0, self.row1 = T.from_numpy(row0).to(dev), T.from_numpy(row1).to(dev)
        self.pos, self.blk_off = T.from_numpy(pos).to(dev), T.from_numpy(blk).to(dev)
        self.ref_base = T.tensor([r[0] for r in refs], dtype=T.int64, device=dev)
        tzj = self.tz_jobs_all.cpu().numpy().view(TZ_DT).reshape(-1)

        fj = np.zeros(2 * NP, FRAC_DT)
        fj["orgOff"], fj["refOff"] = tzj["orgOff"], tzj["refOff"]
        fj["orgStride"], fj["refStride"], fj["width"], fj["height"] = org_stride, rs, tzj["width"], tzj["height"]
        fj["motionLambda"], fj["useHad"], fj["bitDepth"] = motion_lambda, 1, 10
        self.frac = _Tab(T, dev, fj)
        self.frac_res = T.zeros((2 * NP, 16), dtype=T.uint8, device=dev)

        # motion compensation with the consumer fused in (vtmhip_motion_compensation_batch_dev):
        #   pred_other  other list's uni prediction -> bi-pred ME target 2*org - pred        (xMotionEstimation bBi branch, :3316-3329)
        #   pred_final  chosen uni / bi (addAvg) prediction -> prediction + residual org - pred
        qj0 = np.zeros(NP, PRED_DT)
        qj0["orgOff"], qj0["orgStride"] = ys_all * org_stride + xs_all, org_stride
        qj0["refStride"] = rs
        qj0["predOff"], qj0["outOff"], qj0["predStride"], qj0["outStride"] = blk, blk, sizes, sizes
        qj0["width"], qj0["height"], qj0["bitDepth"] = sizes, sizes, 10
        qo = qj0.copy()
        qo["epilogue"] = 2
        self.pred_other = _Tab(T, dev, qo)
        qf = qj0.copy()
        qf["epilogue"] = 1
        self.pred_final = _Tab(T, dev, qf)
        uj = np.zeros(NP, FULL_DT)
        uj["orgOff"], uj["orgStride"], uj["refStride"] = blk, sizes, rs
        uj["puX"], uj["puY"], uj["width"], uj["height"] = xs_all, ys_all, sizes, sizes
        uj["subShift"] = np.where((sizes > 8) & (sizes <= 64), 1, 0)
        uj["signedSamples"], uj["motionLambda"], uj["searchRange"] = 1, motion_lambda, 4
        self.full = _Tab(T, dev, uj)
        self.full_res = T.zeros((NP, 8), dtype=T.int32, device=dev)
        bj = np.zeros(NP, FRAC_DT)
        bj["orgOff"], bj["orgStride"], bj["refStride"], bj["width"], bj["height"] = blk, sizes, rs, sizes, sizes
        bj["motionLambda"], bj["useHad"], bj["bitDepth"] = motion_lambda, 1, 10
        self.frac_bi = _Tab(T, dev, bj)
        self.frac_bi_res = T.zeros((NP, 16), dtype=T.uint8, device=dev)

        # ---- device-side chaining (vtmhip_frame_*): static columns once, decisions in HBM -----------------------------------
        for tab in (self.pred_other, self.pred_final):
            for l in (0, 1):
                tab.col2("refOff", l).copy_(self.ref_base[l] + self.pos)
        self.row0_32, self.row1_32 = self.row0.to(T.int32), self.row1.to(T.int32)
        self.mvq = T.zeros((2 * NP, 2), dtype=T.int32, device=dev)
        self.refine = T.zeros(NP, dtype=T.int32, device=dev)
        self.bi_mv = T.zeros((NP, 2), dtype=T.int32, device=dev)
        self.cost_bi = T.zeros(NP, dtype=T.int64, device=dev)
        self.use_bi = T.zeros(NP, dtype=T.int32, device=dev)
        ft = FrameTabs()
        ft.numPU = NP
        ft.tz, ft.tzRes, ft.fracRes = self.tz_jobs_all.data_ptr(), self.tz_res_all.data_ptr(), self.frac_res.data_ptr()
        ft.row0, ft.row1, ft.pos = self.row0_32.data_ptr(), self.row1_32.data_ptr(), self.pos.data_ptr()
        ft.refBase[0], ft.refBase[1] = int(refs[0][0]), int(refs[1][0])
        ft.predOther, ft.full, ft.fracBi = self.pred_other.ptr, self.full.ptr, self.frac_bi.ptr
        ft.fullRes, ft.fracBiRes, ft.predFinal = self.full_res.data_ptr(), self.frac_bi_res.data_ptr(), self.pred_final.ptr
        ft.mvq, ft.refineList, ft.biMv = self.mvq.data_ptr(), self.refine.data_ptr(), self.bi_mv.data_ptr()
        ft.costBi, ft.useBi = self.cost_bi.data_ptr(), self.use_bi.data_ptr()
        self.frame_tabs = ft
        # per-level views of the same table set (the per-PU arrays advanced to the level's first PU): each level's chain of stages can then
        # run on its own stream as soon as that level's integer search is done
        for lvl in self.levels:
            pb, n = lvl["pb"], lvl["npu"]
            lt = FrameTabs()
            C.memmove(C.byref(lt), C.byref(ft), C.sizeof(FrameTabs))
            lt.numPU = n
            for name, isz in (("row0", 4), ("row1", 4), ("pos", 8), ("predOther", PRED_DT.itemsize), ("full", FULL_DT.itemsize), ("fracBi", FRAC_DT.itemsize),
                              ("fullRes", 32), ("fracBiRes", 16), ("predFinal", PRED_DT.itemsize), ("refineList", 4), ("biMv", 8), ("costBi", 8), ("useBi", 4)):
                setattr(lt, name, getattr(ft, name) + pb * isz)
            lvl["ftabs"] = lt
        self.side_streams = [T.cuda.Stream(device=dev) for _ in range(int(os.environ.get("VTM_AMD_SIDE_STREAMS", "5")))] if dev.type == "cuda" else []

        # ---- transform units: the PU itself up to 64x64, four 64x64 quadrants of a 128x128 PU (MaxTbSize 64) ---------------
        tu_tabs, legacy, tb, max_coef = [], [], 0, 0
        for lvl in self.levels:
            s, npu = lvl["size"], lvl["npu"]
            ts = min(s, 64)
            q = s // ts
            blk_l = lvl["sb"] + np.arange(npu, dtype=np.int64) * s * s
            tu_src = np.stack([blk_l + qy * ts * s + qx * ts for qy in range(q) for qx in range(q)], 1).reshape(-1)
            ntu = tu_src.size
            cands = MTS_CANDS if ts <= 32 else MTS_CANDS[:1]
            nc = len(cands)
            coef_off = np.arange(ntu * nc, dtype=np.int64) * ts * ts        # per-level arena (levels run one after the other)
            uj2 = np.zeros(ntu * nc, TU_DT)
            uj2["resiOff"], uj2["outOff"], uj2["resiStride"], uj2["width"], uj2["height"] = np.tile(tu_src, nc), coef_off, s, ts, ts
            uj2["qpPer"], uj2["qpRem"], uj2["bitDepth"] = self.qp_per, self.qp_rem, 10
            uj2["typeHor"] = np.repeat([c[0] for c in cands], ntu)
            uj2["typeVer"] = np.repeat([c[1] for c in cands], ntu)
            tu_tabs.append(uj2)
            lvl["ntu"], lvl["nc"], lvl["ts"], lvl["tb"] = ntu, nc, ts, tb
            tb += ntu * nc
            max_coef = max(max_coef, ntu * nc * ts * ts)
            if not fused_tu:   # the five separate kernels (kept for A/B and as the non-fused parity path)
                tj = np.zeros(ntu * nc, TR_DT)
                tj["srcOff"], tj["dstOff"], tj["srcStride"], tj["dstStride"] = np.tile(tu_src, nc), coef_off, s, ts
                tj["width"], tj["height"], tj["bitDepth"], tj["typeHor"], tj["typeVer"] = ts, ts, 10, uj2["typeHor"], uj2["typeVer"]
                ij = tj.copy()
                ij["srcOff"] = coef_off
                qj = np.zeros(ntu * nc, Q_DT)
                qj["srcOff"], qj["dstOff"], qj["width"], qj["height"] = coef_off, coef_off, ts, ts
                qj["qpPer"], qj["qpRem"], qj["bitDepth"] = self.qp_per, self.qp_rem, 10
                dj = np.zeros(ntu * nc, DIST_DT)
                dj["orgOff"], dj["curOff"], dj["orgStride"], dj["curStride"] = np.tile(tu_src, nc), coef_off, s, ts
                dj["width"], dj["height"], dj["kind"] = ts, ts, _lib.DIST_SSE
                lvl["xt"], lvl["xit"], lvl["quant"], lvl["sse"] = (_Tab(T, dev, x) for x in (tj, ij, qj, dj))
        self.tu = _Tab(T, dev, np.concatenate(tu_tabs))
        self.tu_res = T.zeros((tb, 2), dtype=T.int64, device=dev)   # vtmhip_tu_result {sse u64, sumAbs i32, absSum i32}
        for lvl in self.levels:
            n_l = lvl["ntu"] * lvl["nc"]
            r = self.tu_res[lvl["tb"]:lvl["tb"] + n_l]
            lvl["sse_out"], lvl["sum_abs"], lvl["abs_sum"] = r[:, 0], r.view(T.int32)[:, 2], r.view(T.int32)[:, 3]
            if not fused_tu:
                lvl["sum_abs"] = T.zeros(n_l, dtype=T.int32, device=dev)
                lvl["abs_sum"] = T.zeros(n_l, dtype=T.int32, device=dev)
                lvl["sse_out"] = T.zeros(n_l, dtype=T.int64, device=dev)
        mk = lambda n: T.zeros(n, dtype=T.int16, device=dev)   # noqa: E731
        self.buf = dict(org_bi=mk(sb), pred=mk(sb), resi=mk(sb))
        # quantised levels: one arena per level (the levels' chains may run concurrently)
        qoff = 0
        for lvl in self.levels:
            lvl["qoff"] = qoff
            qoff += lvl["ntu"] * lvl["nc"] * lvl["ts"] * lvl["ts"]
        self.qcoef = T.zeros(qoff, dtype=T.int32, device=dev)
        if not fused_tu:
            self.coef = T.zeros(max_coef, dtype=T.int32, device=dev)
            self.dqcoef = T.zeros(max_coef, dtype=T.int32, device=dev)
            self.rec_resi = T.zeros(max_coef, dtype=T.int16, device=dev)
        self._marks = None

    # ---- stage timing (HIP events on the launch stream; only when run(..., timing=True)) ----------------------------------
    def _mark(self, name):
        if self._marks is not None:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            self._marks.append((name, e))

    def stage_ms(self):
        """{stage: milliseconds} of the last timed run (call after a synchronize); 'glue' = on-device bookkeeping between stages."""
        acc = {}
        for (n0, e0), (n1, e1) in zip(self._marks[:-1], self._marks[1:]):
            acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
        acc.pop("start", None)
        return acc

    def alg_bytes(self):
        """Algorithmic bytes per kernel family for one picture (SURVEY.md 8d per-unit figures):
        tz    sum over searches of nEval * (4*W*H >> subShift)                       [SAD: 4*W*H >> subShift per candidate]
        frac  per search: 6 H + 18 V filter passes (4 B per output sample) + 18 SATDs (256 B per 8x8 tile = 4 B per sample)
        full  81 candidates * (4*W*H >> subShift)
        mc    4 B per output sample of every filter pass (H pass on H+7 rows, V pass) of the predictions formed (other-list uni, final
              uni or bi) + 6 B per sample of each fused epilogue (org read, prediction / target / residual written)
        tu    per sample: xT 6 + quant 8 + dequant 8 + xIT 6 + SSE 4 = 32 B"""
        b = dict(tz=self.stats()[1], frac=0, full=0, mc=0, tu=0)
        nbi = int(self._use_bi.sum().item()) if getattr(self, "_use_bi", None) is not None else 0
        frac_bi = nbi / max(1, self.NP)
        for lvl in self.levels:
            s, npu, nt, ts = lvl["size"], lvl["npu"], lvl["ntu"] * lvl["nc"], lvl["ts"]
            b["frac"] += 3 * npu * (24 * (s + 8) * s + 144 * s * s)
            b["full"] += npu * 81 * (4 * s * s >> subshift_mode2(s, s))
            b["mc"] += int(npu * (2 + frac_bi) * 4 * ((2 * s + 7) * s)) + 2 * npu * 6 * s * s
            b["tu"] += nt * ts * ts * 32
        return b

    def _per_level(self, fn):
        for lvl in self.levels:
            fn(lvl, lvl["pb"], lvl["npu"], lvl["size"])

    def _level_chain(self, lvl, org_ptr, dpb_ptr):
        """Stages (2)-(5) of ONE quadtree level, in order, on the context's current stream."""
        ctx, pb, n, s, lt = self.ctx, lvl["pb"], lvl["npu"], lvl["size"], lvl["ftabs"]
        fr, po, fu, fb, pf = self.frac, self.pred_other, self.full, self.frac_bi, self.pred_final
        ctx.frame_frac_jobs(fr.ptr + 2 * pb * FRAC_DT.itemsize, self.tz_jobs_all.data_ptr() + 2 * pb * TZ_DT.itemsize, self.tz_res_all.data_ptr() + 2 * pb * 32, 2 * n)
        ctx.frac_search_batch(org_ptr, dpb_ptr, fr.ptr + 2 * pb * FRAC_DT.itemsize, 2 * n, s, s, self.frac_res.data_ptr() + 2 * pb * 16, uniform_square=True)
        ctx.frame_stage(lt, 0)
        ctx.motion_compensation_batch(org_ptr, dpb_ptr, None, self.buf["org_bi"].data_ptr(), po.ptr + pb * PRED_DT.itemsize, n, s, s)
        ctx.full_search_batch(lvl["pic_full"], self.buf["org_bi"].data_ptr(), dpb_ptr, fu.ptr + pb * FULL_DT.itemsize, n, self.full_res.data_ptr() + pb * 32,
                              square=s if s <= 64 else 0)
        ctx.frame_stage(lt, 1)
        ctx.frac_search_batch(self.buf["org_bi"].data_ptr(), dpb_ptr, fb.ptr + pb * FRAC_DT.itemsize, n, s, s, self.frac_bi_res.data_ptr() + pb * 16, uniform_square=True)
        ctx.frame_stage(lt, 2)
        ctx.motion_compensation_batch(org_ptr, dpb_ptr, self.buf["pred"].data_ptr(), self.buf["resi"].data_ptr(), pf.ptr + pb * PRED_DT.itemsize, n, s, s)
        nt, ts = lvl["ntu"] * lvl["nc"], lvl["ts"]
        ctx.tu_chain_batch(self.buf["resi"].data_ptr(), self.tu.ptr + lvl["tb"] * TU_DT.itemsize, nt, ts, ts, self.tu_res.data_ptr() + lvl["tb"] * 16,
                           self.qcoef.data_ptr() + 4 * lvl["qoff"], None, uniform=True)

    def _run_overlapped(self, org_ptr, dpb_ptr):
        """Level-major order over several streams: the integer searches stay one dependent chain (child <- parent) on the caller's stream;
        every level's remaining stages start as soon as ITS integer search is done, on a side stream, so the low-parallelism launches of
        the large blocks (960 searches of 128x128) share the chip with the other levels' work.  Same launches, same tables, same results
        as the stage-major order."""
        T, ctx = self.torch, self.ctx
        main = T.cuda.current_stream()
        for i, lvl in enumerate(self.levels):
            if lvl["parent"] is not None:
                ctx.frame_child_start(lvl["jobs"].data_ptr(), lvl["n"], lvl["parent32"].data_ptr(), self.levels[i - 1]["res"].data_ptr())
            ctx.tz_search_batch(lvl["pic"], org_ptr, dpb_ptr, lvl["jobs"].data_ptr(), lvl["n"], lvl["res"].data_ptr())
            ev = T.cuda.Event()
            ev.record(main)
            st = self.side_streams[i % len(self.side_streams)]
            st.wait_event(ev)
            ctx.set_stream(st.cuda_stream)
            self._level_chain(lvl, org_ptr, dpb_ptr)
            ctx.set_stream(main.cuda_stream)
        for st in self.side_streams:
            main.wait_stream(st)
        self._use_bi = self.use_bi
        self._publish()

    def _publish(self):
        """per-level views of every decision (what tests/cpu_chain.py and a host encoder read back)"""
        T = self.torch
        cost_uni = self.frac_res.view(T.int64)[:, 1]
        for lvl in self.levels:
            pb, n = lvl["pb"], lvl["npu"]
            sl2, sl = slice(2 * pb, 2 * pb + 2 * n), slice(pb, pb + n)
            lvl["out"] = dict(mvq_x=self.mvq[sl2, 0], mvq_y=self.mvq[sl2, 1], cost_uni=cost_uni[sl2], rl=self.refine[sl], bi_x=self.bi_mv[sl, 0],
                              bi_y=self.bi_mv[sl, 1], cost_bi=self.cost_bi[sl], use_bi=self.use_bi[sl])

    def run(self, org_ptr, dpb_ptr, timing=False):
        """timing=True (or no side streams / VTM_AMD_OVERLAP=0): stage-major order on one stream with an event after every stage;
        otherwise the overlapped level-major order."""
        T, ctx, NP = self.torch, self.ctx, self.NP
        if not timing and self.fused_tu and self.side_streams and os.environ.get("VTM_AMD_OVERLAP", "1") != "0":
            self._marks = None
            return self._run_overlapped(org_ptr, dpb_ptr)
        self._marks = [] if timing else None
        self._mark("start")
        # (1) integer ME, coarse to fine: children start from / predict with the parent's vector
        for i, lvl in enumerate(self.levels):
            if lvl["parent"] is not None:
                ctx.frame_child_start(lvl["jobs"].data_ptr(), lvl["n"], lvl["parent32"].data_ptr(), self.levels[i - 1]["res"].data_ptr())
            ctx.tz_search_batch(lvl["pic"], org_ptr, dpb_ptr, lvl["jobs"].data_ptr(), lvl["n"], lvl["res"].data_ptr())
        self._mark("tz")

        # (2) fractional ME per (PU, list)
        fr, ft = self.frac, self.frame_tabs
        ctx.frame_frac_jobs(fr.ptr, self.tz_jobs_all.data_ptr(), self.tz_res_all.data_ptr(), 2 * NP)
        self._mark("glue")
        self._per_level(lambda l, pb, n, s: ctx.frac_search_batch(org_ptr, dpb_ptr, fr.ptr + 2 * pb * FRAC_DT.itemsize, 2 * n, s, s,
                                                                  self.frac_res.data_ptr() + 2 * pb * 16, uniform_square=True))
        self._mark("frac")

        # (3) bi-pred refinement of the list with the larger uni cost (FASTINTERSEARCH_MODE1: one iteration, :2544-2556):
        #     prediction of the other list fused with 2*org - pred, +-4 exhaustive search, fractional search on the new target
        po, fu, fb = self.pred_other, self.full, self.frac_bi
        ctx.frame_stage(ft, 0)
        self._mark("glue")
        self._per_level(lambda l, pb, n, s: ctx.motion_compensation_batch(org_ptr, dpb_ptr, None, self.buf["org_bi"].data_ptr(),
                                                                          po.ptr + pb * PRED_DT.itemsize, n, s, s))
        self._mark("mc")
        self._per_level(lambda l, pb, n, s: ctx.full_search_batch(l["pic_full"], self.buf["org_bi"].data_ptr(), dpb_ptr, fu.ptr + pb * FULL_DT.itemsize, n,
                                                                  self.full_res.data_ptr() + pb * 32, square=s if s <= 64 else 0))
        self._mark("full")
        ctx.frame_stage(ft, 1)
        self._mark("glue")
        self._per_level(lambda l, pb, n, s: ctx.frac_search_batch(self.buf["org_bi"].data_ptr(), dpb_ptr, fb.ptr + pb * FRAC_DT.itemsize, n, s, s,
                                                                  self.frac_bi_res.data_ptr() + pb * 16, uniform_square=True))
        self._mark("frac")

        # (4) final prediction and residual: bi-prediction (addAvg of the two 14-bit predictions) when cheaper, else the best uni list
        pf = self.pred_final
        ctx.frame_stage(ft, 2)
        self._use_bi = self.use_bi
        self._mark("glue")
        self._per_level(lambda l, pb, n, s: ctx.motion_compensation_batch(org_ptr, dpb_ptr, self.buf["pred"].data_ptr(), self.buf["resi"].data_ptr(),
                                                                          pf.ptr + pb * PRED_DT.itemsize, n, s, s))
        self._mark("mc")
        # (5) residual coding per TU and transform candidate
        for lvl in self.levels:
            nt, ts = lvl["ntu"] * lvl["nc"], lvl["ts"]
            if self.fused_tu:
                # levels go to the host for the CABAC estimate in the real encoder; the bench keeps them in HBM
                ctx.tu_chain_batch(self.buf["resi"].data_ptr(), self.tu.ptr + lvl["tb"] * TU_DT.itemsize, nt, ts, ts,
                                   self.tu_res.data_ptr() + lvl["tb"] * 16, self.qcoef.data_ptr() + 4 * lvl["qoff"], None, uniform=True)
            else:
                ctx.xT_batch(self.buf["resi"].data_ptr(), self.coef.data_ptr(), lvl["xt"].ptr, nt, ts, ts, lvl["sum_abs"].data_ptr())
                ctx.quant_batch(self.coef.data_ptr(), self.qcoef.data_ptr(), None, lvl["quant"].ptr, nt, lvl["abs_sum"].data_ptr())
                ctx.dequant_batch(self.qcoef.data_ptr(), self.dqcoef.data_ptr(), lvl["quant"].ptr, nt)
                ctx.xIT_batch(self.dqcoef.data_ptr(), self.rec_resi.data_ptr(), lvl["xit"].ptr, nt, ts, ts)
                ctx.dist_batch(self.buf["resi"].data_ptr(), self.rec_resi.data_ptr(), lvl["sse"].ptr, nt, lvl["sse_out"].data_ptr())
        self._mark("tu")
        self._publish()


# ======================================================================================================================
# Level-order InterSearch::predInterSearch (InterSearch.cpp:2245-3065, translational part) + residual coding of one picture
# ======================================================================================================================
from .lib import MAX_REF, MeCfg, MeJob, MeOut, PisLevel, PisPu, PisRow   # noqa: E402

ME_DT, MEOUT_DT, ROW_DT, PU_DT = np.dtype(MeJob), np.dtype(MeOut), np.dtype(PisRow), np.dtype(PisPu)
# (typeHor, typeVer) per tu.mtsIdx (TrQuant::getTrTypes): 0 DCT2/DCT2, 1 transform skip, 2..5 the DST7/DCT8 pairs
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
    the TU chains (DCT2, chroma QP by the CTC mapping table)."""

    def __init__(self, ctx, torch, device, pic_w, pic_h, org_stride, refs, search_ranges, motion_lambda=8.0, qp=32, sizes=(128, 64, 32, 16, 8),
                 ctu_filter=None, transform_skip=False, bit_depth=10, pocs=None, chroma=None, bdof=True):
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
        self.bdof = bool(bdof and pocs is not None and self.is_b)
        if chroma is not None:
            cqp = chroma_qp(qp) + 6 * (bit_depth - 8)
            self.cqp_per, self.cqp_rem = cqp // 6, cqp % 6
        self.levels, prev = [], None
        sb = 0
        for (s, xs, ys, parent) in quadtree_levels(pic_w, pic_h, sizes, None, ctu_filter):
            n = xs.size
            if n == 0:
                prev = None
                continue
            w, h = (s, s) if np.isscalar(s) else s                   # a level of the quadtree, or of a binary / ternary split shape
            lvl = dict(size=s, w=w, h=h, npu=n, xs=xs, ys=ys, sb=sb)
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
            L.numPU, L.smvdBit, L.refStride = n, 0, rs
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
            lvl["pis"] = L
            lvl["pic"] = PicParams(pic_w, pic_h, 128, bit_depth, wpj.get(max(w, h), 1))
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
            lvl.update(ntu=ntu, nc=nc, ts=tw, tw=tw, th=th, cands=cl, tu=_Tab(T, dev, tj), tu_res=T.zeros((ntu * nc, 2), dtype=T.int64, device=dev),
                       qcoef=T.zeros(ntu * nc * tw * th, dtype=T.int32, device=dev))
            self.levels.append(lvl)
            prev = lvl
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
        ctx.motion_compensation_batch(org_ptr, dpb_ptr, buf["pred"].data_ptr(), buf["resi"].data_ptr(), lvl["pred_final"].ptr, n, w, h)
        if self.bdof and w * h >= 128:      # the PUs the final stage routed to BDOF (bi-prediction from opposite directions at equal POC distance; 8x8 never qualifies)
            ctx.bdof_batch(org_ptr, dpb_ptr, buf["pred"].data_ptr(), buf["resi"].data_ptr(), lvl["pred_final"].ptr, n, w, h)
        if self.chroma is not None:
            ctx.motion_compensation_batch(org_ptr, dpb_ptr, buf["pred_c"].data_ptr(), buf["resi_c"].data_ptr(), lvl["pred_final_c"].ptr, 2 * n, w // 2, h // 2)
        self._mark("mc")
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
            d = dict(size=lvl["size"], w=lvl["w"], h=lvl["h"], tw=lvl["tw"], th=lvl["th"], npu=lvl["npu"], ntu=lvl["ntu"], nc=lvl["nc"], ts=lvl["ts"], cands=lvl["cands"], xs=lvl["xs"], ys=lvl["ys"],
                     uni_jobs=lvl["uni_jobs"].t.cpu().numpy().view(ME_DT).reshape(-1), uni_out=lvl["uni_out"].cpu().numpy().view(MEOUT_DT).reshape(-1),
                     uni_rows=lvl["uni_rows"].cpu().numpy().view(ROW_DT).reshape(-1), pus=lvl["pus"].cpu().numpy().view(PU_DT).reshape(-1),
                     tu_res=lvl["tu_res"].cpu().numpy())
            if self.is_b:
                d["bi_jobs"] = lvl["bi_jobs"].t.cpu().numpy().view(ME_DT).reshape(-1)
                d["bi_out"] = lvl["bi_out"].cpu().numpy().view(MEOUT_DT).reshape(-1)
            d["route"] = lvl["pred_final"].col("route").cpu().numpy()
            if self.chroma is not None:
                d.update(ntu_c=lvl["ntu_c"], ts_c=lvl["ts_c"], tw_c=lvl["tw_c"], th_c=lvl["th_c"], tu_res_c=lvl["tu_res_c"].cpu().numpy())
            out.append(d)
        return out

    def result_tensors(self):
        """the per-PU / per-TU result tensors a rank hands to rank 0 (bytes views), coarse to fine"""
        out = []
        for lvl in self.levels:
            out += [lvl["pus"].reshape(-1), lvl["tu_res"].view(self.torch.uint8).reshape(-1)]
            if self.chroma is not None:
                out.append(lvl["tu_res_c"].view(self.torch.uint8).reshape(-1))
        return out

    def work_counts(self):
        R = self.nref[0] + self.nref[1]
        return dict(pus=self.NP, uni_searches=R * self.NP, bi_searches=(self.nref[0] if self.is_b else 0) * self.NP,
                    tu_chains=sum(l["ntu"] * l["nc"] for l in self.levels))
