"""Frame-level driver of the hot path: builds the job tables of one inter picture and runs the stages on the device.

This is the host-side orchestration that, inside the reference, is InterSearch::predInterSearch -> xMotionEstimation
(EncoderLib/InterSearch.cpp:2245-3065, 3299-3494) called from the CU recursion.  The reference visits one PU at a
time; here all PUs of one quadtree level x all reference pictures form one launch, and level L+1 takes its start
vector / MV predictor from the level-L result of the enclosing block (a stand-in for AMVP, which needs the CU
recursion that is out of scope -- SURVEY.md section 8a).  Job tables and results stay in HBM; torch is used only for
device memory, streams and the tiny gather that forms the child predictors.
"""
import ctypes as C

import numpy as np

from . import lib as _lib
from .lib import MeResult, PicParams, TzJob

TZ_DT = np.dtype(TzJob)
RES_DT = np.dtype(MeResult)
# int32 word indices inside the structs (checked against ctypes offsets below)
_J_PRED_HOR, _J_PRED_VER, _J_MV_HOR, _J_MV_VER = 9, 10, 14, 15
assert TzJob.predHor.offset == 4 * _J_PRED_HOR and TzJob.mvHor.offset == 4 * _J_MV_HOR
assert C.sizeof(TzJob) % 4 == 0 and C.sizeof(MeResult) == 32


WAVES_PER_JOB = {128: 4, 64: 4, 32: 4, 16: 2, 8: 1}   # vtmhip_pic_params.wavesPerJob per PU size (measured)


FULL_WAVES_PER_JOB = {128: 16, 64: 8, 32: 4, 16: 1, 8: 1}   # 81-point exhaustive search: one candidate at a time per wave for big PUs


def subshift_mode2(w, h):
    """RdCost::setDistParam subShiftMode 2 (FEN=1 / FastSearch, RdCost.cpp:311-317)."""
    return 1 if (h > 8 and w <= 64) else 0


def quadtree_levels(pic_w, pic_h, sizes=(128, 64, 32, 16, 8), row_filter=None):
    """Square PUs of every quadtree level that lie fully inside the picture.  row_filter(ctu_row_array) -> bool array selects the
    CTU rows (128 luma rows each) this rank owns.  Returns [(size, xs, ys, parent_index_or_None)], coarse to fine."""
    levels = []
    prev = None
    for s in sizes:
        ys, xs = np.mgrid[0:pic_h - s + 1:s, 0:pic_w - s + 1:s]
        xs, ys = xs.ravel().astype(np.int64), ys.ravel().astype(np.int64)
        if row_filter is not None:
            keep = np.asarray(row_filter(ys // 128), dtype=bool)
            xs, ys = xs[keep], ys[keep]
        parent = None
        if prev is not None:
            ps, pxs, pys = prev
            lut = {(int(x), int(y)): i for i, (x, y) in enumerate(zip(pxs, pys))}
            parent = np.array([lut.get((int(x) // ps * ps, int(y) // ps * ps), -1) for x, y in zip(xs, ys)], dtype=np.int64)
        levels.append((s, xs, ys, parent))
        prev = (s, xs, ys)
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
                 sizes=(128, 64, 32, 16, 8), row_filter=None, waves_per_job=None):
        """refs: [(ref_off, ref_stride)] sample offsets of each reference plane's (0,0) inside `dpb`."""
        self.ctx, self.torch, self.device = ctx, torch, device
        self.pic_w, self.pic_h = pic_w, pic_h
        # waves that share one search: big PUs have long SADs and (at the top level, which has no predictor) raster scans
        self.wpj = dict(WAVES_PER_JOB)
        self.wpj.update(waves_per_job or {})
        self.levels = []
        self.n_jobs = 0
        self.alg_bytes_per_eval = []
        for (s, xs, ys, parent) in quadtree_levels(pic_w, pic_h, sizes, row_filter):
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
                       parent=None if par is None else torch.from_numpy(par).to(device))
            self.levels.append(lvl)
            self.n_jobs += n
            self.alg_bytes_per_eval.append(4 * s * s >> subshift_mode2(s, s))

    def run(self, org_ptr, dpb_ptr):
        """Launches every level (coarse to fine) on the context's stream; no host synchronisation."""
        torch = self.torch
        for i, lvl in enumerate(self.levels):
            if lvl["parent"] is not None:
                j32 = lvl["jobs"].view(torch.int32)
                pres = self.levels[i - 1]["res"]
                p = lvl["parent"].clamp(min=0)
                mvx = torch.where(lvl["parent"] >= 0, pres[p, 0], torch.zeros_like(pres[p, 0]))
                mvy = torch.where(lvl["parent"] >= 0, pres[p, 1], torch.zeros_like(pres[p, 1]))
                j32[:, _J_MV_HOR] = mvx << 4      # start vector, internal 1/16 precision
                j32[:, _J_MV_VER] = mvy << 4
                j32[:, _J_PRED_HOR] = mvx << 2    # MV predictor, quarter-sample units
                j32[:, _J_PRED_VER] = mvy << 2
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
from .lib import DistJob, FracJob, FracResult, FullJob, McJob, PelOpJob, QuantJob, TrJob, TuJob   # noqa: E402

FRAC_DT, FRACRES_DT, MC_DT, FULL_DT = np.dtype(FracJob), np.dtype(FracResult), np.dtype(McJob), np.dtype(FullJob)
PEL_DT, TR_DT, Q_DT, DIST_DT = np.dtype(PelOpJob), np.dtype(TrJob), np.dtype(QuantJob), np.dtype(DistJob)
TU_DT = np.dtype(TuJob)

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
        tt = {np.dtype(np.int64): self.torch.int64, np.dtype(np.int32): self.torch.int32, np.dtype(np.int16): self.torch.int16}[kind]
        assert off % kind.itemsize == 0
        return self.t.view(tt)[:, off // kind.itemsize]

    @property
    def ptr(self):
        return self.t.data_ptr()


class FrameHotPath(FrameME):
    """All stages for one picture with two reference pictures (list 0 / list 1), level by level:

      tz      InterSearch::xTZSearch per (PU, list)                                   (InterSearch.cpp:3640-3976)
      frac    xPatternSearchFracDIF per (PU, list): half + quarter refinement, SATD     (:4284-4339)
      bi      FEN bi-pred iteration (:2531-2680): refine the list with the LARGER uni cost: motion-compensate the other
              list, org' = 2*org - pred (removeHighFreq), +-4 exhaustive search (xPatternSearch), fractional search on org'
      resi    final prediction (bi via addAvg when cheaper, else best uni) -> residual -> per TU (<= 64x64):
              xT (DCT2 + 4 MTS candidates up to 32x32, with sum|coef| for the pre-selection) -> Quant::quant -> dequant -> xIT -> SSE
    Mode decision between the candidates, CABAC bit estimation and DepQuant stay on the host (out of scope);
    every MTS candidate is taken through the whole chain (the reference prunes with the sum|coef| threshold).
    """

    def __init__(self, ctx, torch, device, pic_w, pic_h, org_stride, refs, search_ranges, motion_lambda=8.0, qp=32, fused_tu=True, **kw):
        assert len(refs) == 2
        self.fused_tu = fused_tu
        super().__init__(ctx, torch, device, pic_w, pic_h, org_stride, refs, search_ranges, motion_lambda, **kw)
        self.refs, self.org_stride, self.lam = refs, org_stride, motion_lambda
        base_qp = qp + 12   # 10-bit: qpBdOffset = 12 (Quant.cpp:65-104)
        self.qp_per, self.qp_rem = base_qp // 6, base_qp % 6
        T, dev = torch, device
        max_samples = 0
        for lvl in self.levels:
            s, npu = lvl["size"], lvl["n"] // 2
            jobs = lvl["jobs"].cpu().numpy().view(TZ_DT).reshape(-1)
            xs, ys = jobs["puX"][:npu].astype(np.int64), jobs["puY"][:npu].astype(np.int64)
            blk = np.arange(npu, dtype=np.int64) * s * s
            lvl["npu"] = npu
            lvl["pic_full"] = PicParams(pic_w, pic_h, 128, 10, FULL_WAVES_PER_JOB.get(s, 1))
            lvl["ref_base"] = T.tensor([r[0] for r in refs], dtype=T.int64, device=dev)
            lvl["pos_off"] = [T.from_numpy(ys * r[1] + xs).to(dev) for r in refs]   # block offset inside each reference plane
            ref_strides = [r[1] for r in refs]
            assert ref_strides[0] == ref_strides[1]
            rs = ref_strides[0]

            fj = np.zeros(2 * npu, FRAC_DT)
            fj["orgOff"], fj["refOff"] = jobs["orgOff"], jobs["refOff"]
            fj["orgStride"], fj["refStride"], fj["width"], fj["height"] = org_stride, rs, s, s
            fj["motionLambda"], fj["useHad"], fj["bitDepth"] = motion_lambda, 1, 10
            lvl["frac"] = _Tab(T, dev, fj)
            lvl["frac_res"] = T.zeros((2 * npu, 16), dtype=T.uint8, device=dev)

            mj = np.zeros(npu, MC_DT)
            mj["dstOff"], mj["refStride"], mj["dstStride"], mj["width"], mj["height"], mj["bitDepth"] = blk, rs, s, s, s, 10
            lvl["mc_other"] = _Tab(T, dev, mj)

            pj = np.zeros(npu, PEL_DT)
            pj["aOff"], pj["aStride"] = ys * org_stride + xs, org_stride
            pj["bOff"], pj["bStride"], pj["dstOff"], pj["dstStride"] = blk, s, blk, s
            pj["width"], pj["height"], pj["bitDepth"] = s, s, 10
            lvl["rhf"] = _Tab(T, dev, pj)
            lvl["sub"] = _Tab(T, dev, pj.copy())   # bOff is patched to the chosen prediction
            aj = np.zeros(npu, PEL_DT)
            aj["aOff"], aj["bOff"], aj["dstOff"] = blk, blk, blk
            aj["aStride"], aj["bStride"], aj["dstStride"], aj["width"], aj["height"], aj["bitDepth"] = s, s, s, s, s, 10
            lvl["avg"] = _Tab(T, dev, aj)

            uj = np.zeros(npu, FULL_DT)
            uj["orgOff"], uj["orgStride"], uj["refStride"] = blk, s, rs
            uj["puX"], uj["puY"], uj["width"], uj["height"] = xs, ys, s, s
            uj["subShift"], uj["signedSamples"], uj["motionLambda"], uj["searchRange"] = subshift_mode2(s, s), 1, motion_lambda, 4
            lvl["full"] = _Tab(T, dev, uj)
            lvl["full_res"] = T.zeros((npu, 8), dtype=T.int32, device=dev)

            bj = np.zeros(npu, FRAC_DT)
            bj["orgOff"], bj["orgStride"], bj["refStride"], bj["width"], bj["height"] = blk, s, rs, s, s
            bj["motionLambda"], bj["useHad"], bj["bitDepth"] = motion_lambda, 1, 10
            lvl["frac_bi"] = _Tab(T, dev, bj)
            lvl["frac_bi_res"] = T.zeros((npu, 16), dtype=T.uint8, device=dev)

            for name, bi in (("mc_uni", 0), ("mc_b0", 1), ("mc_b1", 1)):
                m2 = mj.copy()
                m2["bi"] = bi
                lvl[name] = _Tab(T, dev, m2)

            # transform units: the PU itself up to 64x64, four 64x64 quadrants of a 128x128 PU (MaxTbSize 64)
            ts = min(s, 64)
            q = s // ts
            tu_src, tu_stride = [], s
            for qy in range(q):
                for qx in range(q):
                    tu_src.append(blk + qy * ts * s + qx * ts)
            tu_src = np.stack(tu_src, 1).reshape(-1)                     # [npu * q*q]
            ntu = tu_src.size
            cands = MTS_CANDS if ts <= 32 else MTS_CANDS[:1]
            nc = len(cands)
            tj = np.zeros(ntu * nc, TR_DT)
            coef_off = np.arange(ntu * nc, dtype=np.int64) * ts * ts
            tj["srcOff"] = np.tile(tu_src, nc)
            tj["dstOff"], tj["srcStride"], tj["dstStride"], tj["width"], tj["height"], tj["bitDepth"] = coef_off, tu_stride, ts, ts, ts, 10
            tj["typeHor"] = np.repeat([c[0] for c in cands], ntu)
            tj["typeVer"] = np.repeat([c[1] for c in cands], ntu)
            lvl["xt"] = _Tab(T, dev, tj)
            ij = tj.copy()
            ij["srcOff"], ij["dstOff"] = coef_off, coef_off              # dequantised coefficients -> reconstructed residual (contiguous per TU)
            lvl["xit"] = _Tab(T, dev, ij)
            qj = np.zeros(ntu * nc, Q_DT)
            qj["srcOff"], qj["dstOff"], qj["width"], qj["height"] = coef_off, coef_off, ts, ts
            qj["qpPer"], qj["qpRem"], qj["bitDepth"] = self.qp_per, self.qp_rem, 10
            lvl["quant"] = _Tab(T, dev, qj)
            dj = np.zeros(ntu * nc, DIST_DT)
            dj["orgOff"], dj["curOff"], dj["orgStride"], dj["curStride"] = np.tile(tu_src, nc), coef_off, tu_stride, ts
            dj["width"], dj["height"], dj["kind"] = ts, ts, _lib.DIST_SSE
            lvl["sse"] = _Tab(T, dev, dj)
            uj2 = np.zeros(ntu * nc, TU_DT)
            uj2["resiOff"], uj2["outOff"], uj2["resiStride"], uj2["width"], uj2["height"] = np.tile(tu_src, nc), coef_off, tu_stride, ts, ts
            uj2["qpPer"], uj2["qpRem"], uj2["bitDepth"] = self.qp_per, self.qp_rem, 10
            uj2["typeHor"], uj2["typeVer"] = tj["typeHor"], tj["typeVer"]
            lvl["tu"] = _Tab(T, dev, uj2)
            lvl["tu_res"] = T.zeros((ntu * nc, 2), dtype=T.int64, device=dev)   # vtmhip_tu_result: {sse u64, sumAbs i32, absSum i32}
            lvl["ntu"], lvl["nc"], lvl["ts"] = ntu, nc, ts
            lvl["sum_abs"] = T.zeros(ntu * nc, dtype=T.int32, device=dev)
            lvl["abs_sum"] = T.zeros(ntu * nc, dtype=T.int32, device=dev)
            lvl["sse_out"] = T.zeros(ntu * nc, dtype=T.int64, device=dev)
            lvl["blk_off"] = T.from_numpy(blk).to(dev)
            max_samples = max(max_samples, npu * s * s * nc)
        npx = max(l["npu"] * l["size"] ** 2 for l in self.levels)
        mk = lambda: T.zeros(npx, dtype=T.int16, device=dev)   # noqa: E731
        self.buf = dict(pred_other=mk(), org_bi=mk(), pred_uni=mk(), p0=mk(), p1=mk(), pred_bi=mk(), resi=mk())
        # one arena for the two selectable predictions so a job can address either with an offset
        self.pred_sel = T.zeros(2 * npx, dtype=T.int16, device=dev)
        self.npx = npx
        self.coef = T.zeros(max_samples, dtype=T.int32, device=dev)
        self.qcoef = T.zeros(max_samples, dtype=T.int32, device=dev)
        self.dqcoef = T.zeros(max_samples, dtype=T.int32, device=dev)
        self.rec_resi = T.zeros(max_samples, dtype=T.int16, device=dev)
        self.out = []
        self._marks = None

    # ---- stage timing (HIP events on the launch stream; only when run(..., timing=True)) ----------------------------------
    def _mark(self, name):
        if self._marks is not None:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            self._marks.append((name, e))

    def stage_ms(self):
        """{stage: milliseconds} of the last timed run (call after a synchronize)."""
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
        mc    4 B per output sample of every filter pass (H pass on H+7 rows, V pass)
        pelop 6 B per sample (two 2-byte reads, one 2-byte write)
        tu    per sample: xT 6 + quant 8 + dequant 8 + xIT 6 + SSE 4 = 32 B"""
        b = dict(tz=self.stats()[1], frac=0, full=0, mc=0, pelop=0, tu=0)
        for lvl in self.levels:
            s, npu, nt, ts = lvl["size"], lvl["npu"], lvl["ntu"] * lvl["nc"], lvl["ts"]
            b["frac"] += 3 * npu * (24 * (s + 8) * s + 144 * s * s)
            b["full"] += npu * 81 * (4 * s * s >> subshift_mode2(s, s))
            b["mc"] += 4 * npu * 4 * ((2 * s + 7) * s)
            b["pelop"] += 3 * npu * 6 * s * s
            b["tu"] += nt * ts * ts * 32
        return b

    # ---- per-level stages ---------------------------------------------------------------------------------------------
    def _level(self, i, org_ptr, dpb_ptr):
        T, ctx, lvl = self.torch, self.ctx, self.levels[i]
        s, npu = lvl["size"], lvl["npu"]
        # (1) integer ME (parent predictors patched as in FrameME.run)
        if lvl["parent"] is not None:
            j32 = lvl["jobs"].view(T.int32)
            pres = self.levels[i - 1]["res"]
            p = lvl["parent"].clamp(min=0)
            has = lvl["parent"] >= 0
            mvx = T.where(has, pres[p, 0], T.zeros_like(pres[p, 0]))
            mvy = T.where(has, pres[p, 1], T.zeros_like(pres[p, 1]))
            j32[:, _J_MV_HOR], j32[:, _J_MV_VER] = mvx << 4, mvy << 4
            j32[:, _J_PRED_HOR], j32[:, _J_PRED_VER] = mvx << 2, mvy << 2
        ctx.tz_search_batch(lvl["pic"], org_ptr, dpb_ptr, lvl["jobs"].data_ptr(), lvl["n"], lvl["res"].data_ptr())
        self._mark("tz")
        tz = lvl["res"]
        j32 = lvl["jobs"].view(T.int32)
        pred_h, pred_v = j32[:, _J_PRED_HOR], j32[:, _J_PRED_VER]

        # (2) fractional ME per (PU, list)
        fr = lvl["frac"]
        fr.col("intX").copy_(tz[:, 0].to(T.int16))
        fr.col("intY").copy_(tz[:, 1].to(T.int16))
        fr.col("predHor").copy_(pred_h)
        fr.col("predVer").copy_(pred_v)
        ctx.frac_search_batch(org_ptr, dpb_ptr, fr.ptr, 2 * npu, s, s, lvl["frac_res"].data_ptr(), uniform_square=True)
        self._mark("frac")
        fres16 = lvl["frac_res"].view(T.int16)
        cost_uni = lvl["frac_res"].view(T.int64)[:, 1]
        mvq_x = (tz[:, 0] << 2) + (fres16[:, 0].to(T.int32) << 1) + fres16[:, 2].to(T.int32)   # quarter-sample units
        mvq_y = (tz[:, 1] << 2) + (fres16[:, 1].to(T.int32) << 1) + fres16[:, 3].to(T.int32)

        # (3) bi-pred refinement of the list with the larger uni cost (FASTINTERSEARCH_MODE1: one iteration, :2544-2556)
        c0, c1 = cost_uni[:npu], cost_uni[npu:]
        rl = (c0 <= c1).to(T.int64)                    # list to refine
        ol_ = 1 - rl                                    # the other list supplies the fixed prediction
        idx = T.arange(npu, device=self.device)
        sel = lambda a, l: a[l * npu + idx]            # noqa: E731
        pos = T.stack(lvl["pos_off"])                  # [2, npu]
        ref_off = lambda l: lvl["ref_base"][l] + pos[l, idx]   # noqa: E731
        mo = lvl["mc_other"]
        mo.col("refOff").copy_(ref_off(ol_))
        mo.col("mvHor").copy_(sel(mvq_x, ol_) << 2)
        mo.col("mvVer").copy_(sel(mvq_y, ol_) << 2)
        ctx.mc_luma_batch(dpb_ptr, self.buf["pred_other"].data_ptr(), mo.ptr, npu, s, s)
        self._mark("mc")
        ctx.remove_high_freq_batch(org_ptr, self.buf["pred_other"].data_ptr(), self.buf["org_bi"].data_ptr(), lvl["rhf"].ptr, npu)
        self._mark("pelop")
        fu = lvl["full"]
        fu.col("refOff").copy_(ref_off(rl))
        fu.col("predHor").copy_(sel(pred_h, rl))
        fu.col("predVer").copy_(sel(pred_v, rl))
        fu.col("centerHor").copy_(sel(mvq_x, rl) << 2)
        fu.col("centerVer").copy_(sel(mvq_y, rl) << 2)
        ctx.full_search_batch(lvl["pic_full"], self.buf["org_bi"].data_ptr(), dpb_ptr, fu.ptr, npu, lvl["full_res"].data_ptr())
        self._mark("full")
        fb = lvl["frac_bi"]
        fb.col("refOff").copy_(ref_off(rl))
        fb.col("intX").copy_(lvl["full_res"][:, 0].to(T.int16))
        fb.col("intY").copy_(lvl["full_res"][:, 1].to(T.int16))
        fb.col("predHor").copy_(sel(pred_h, rl))
        fb.col("predVer").copy_(sel(pred_v, rl))
        ctx.frac_search_batch(self.buf["org_bi"].data_ptr(), dpb_ptr, fb.ptr, npu, s, s, lvl["frac_bi_res"].data_ptr(), uniform_square=True)
        self._mark("frac")
        b16 = lvl["frac_bi_res"].view(T.int16)
        cost_bi = lvl["frac_bi_res"].view(T.int64)[:, 1] >> 1   # the reference re-weights by 0.5 plus rate terms (:3483); mode decision is host work
        bi_x = (lvl["full_res"][:, 0] << 2) + (b16[:, 0].to(T.int32) << 1) + b16[:, 2].to(T.int32)
        bi_y = (lvl["full_res"][:, 1] << 2) + (b16[:, 1].to(T.int32) << 1) + b16[:, 3].to(T.int32)

        # (4) final prediction: best uni list, and the bi-prediction (addAvg of the two 14-bit MC outputs)
        best_l = (c1 < c0).to(T.int64)
        mu = lvl["mc_uni"]
        mu.col("refOff").copy_(ref_off(best_l))
        mu.col("mvHor").copy_(sel(mvq_x, best_l) << 2)
        mu.col("mvVer").copy_(sel(mvq_y, best_l) << 2)
        uni_ptr = self.pred_sel.data_ptr()
        bi_ptr = uni_ptr + 2 * self.npx
        ctx.mc_luma_batch(dpb_ptr, uni_ptr, mu.ptr, npu, s, s)
        mvx_l = [T.where(rl == l, bi_x, mvq_x[l * npu:(l + 1) * npu]) for l in (0, 1)]   # refined list takes the bi vector
        mvy_l = [T.where(rl == l, bi_y, mvq_y[l * npu:(l + 1) * npu]) for l in (0, 1)]
        for l, name, bufname in ((0, "mc_b0", "p0"), (1, "mc_b1", "p1")):
            mb = lvl[name]
            mb.col("refOff").copy_(lvl["ref_base"][l] + pos[l])
            mb.col("mvHor").copy_(mvx_l[l] << 2)
            mb.col("mvVer").copy_(mvy_l[l] << 2)
            ctx.mc_luma_batch(dpb_ptr, self.buf[bufname].data_ptr(), mb.ptr, npu, s, s)
        self._mark("mc")
        ctx.add_avg_batch(self.buf["p0"].data_ptr(), self.buf["p1"].data_ptr(), bi_ptr, lvl["avg"].ptr, npu)
        use_bi = cost_bi < T.minimum(c0, c1)
        sb = lvl["sub"]
        sb.col("bOff").copy_(lvl["blk_off"] + use_bi.to(T.int64) * self.npx)
        ctx.subtract_batch(org_ptr, uni_ptr, self.buf["resi"].data_ptr(), sb.ptr, npu)
        self._mark("pelop")

        # (5) residual coding per TU and transform candidate
        nt, ts = lvl["ntu"] * lvl["nc"], lvl["ts"]
        if self.fused_tu:
            # levels go to the host for the CABAC estimate in the real encoder; the bench keeps them in HBM
            ctx.tu_chain_batch(self.buf["resi"].data_ptr(), lvl["tu"].ptr, nt, ts, ts, lvl["tu_res"].data_ptr(), self.qcoef.data_ptr(), None, uniform=True)
            r32 = lvl["tu_res"].view(T.int32)
            lvl["sse_out"], lvl["sum_abs"], lvl["abs_sum"] = lvl["tu_res"][:, 0], r32[:, 2], r32[:, 3]
        else:
            ctx.xT_batch(self.buf["resi"].data_ptr(), self.coef.data_ptr(), lvl["xt"].ptr, nt, ts, ts, lvl["sum_abs"].data_ptr())
            ctx.quant_batch(self.coef.data_ptr(), self.qcoef.data_ptr(), None, lvl["quant"].ptr, nt, lvl["abs_sum"].data_ptr())
            ctx.dequant_batch(self.qcoef.data_ptr(), self.dqcoef.data_ptr(), lvl["quant"].ptr, nt)
            ctx.xIT_batch(self.dqcoef.data_ptr(), self.rec_resi.data_ptr(), lvl["xit"].ptr, nt, ts, ts)
            ctx.dist_batch(self.buf["resi"].data_ptr(), self.rec_resi.data_ptr(), lvl["sse"].ptr, nt, lvl["sse_out"].data_ptr())
        self._mark("tu")
        lvl["out"] = dict(mvq_x=mvq_x, mvq_y=mvq_y, cost_uni=cost_uni, rl=rl, bi_x=bi_x, bi_y=bi_y, cost_bi=cost_bi, use_bi=use_bi)

    def run(self, org_ptr, dpb_ptr, timing=False):
        self._marks = [] if timing else None
        self._mark("start")
        for i in range(len(self.levels)):
            self._level(i, org_ptr, dpb_ptr)
