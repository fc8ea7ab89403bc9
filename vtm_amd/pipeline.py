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
