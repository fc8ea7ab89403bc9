#!/usr/bin/env python3
"""Kernel micro-benchmarks of SURVEY.md section 8d on one MI355X (inputs resident in HBM, HIP-event timing on the context's stream):

  SATD-8x8   every 8-aligned 8x8 block of frame t x 81 displacements against frame t-1                 (vtmhip_satd8_grid_dev)
  SAD/SATD-PU PU sizes {8x8 .. 128x128, 16x8, 8x16, 32x8, 64x16}, candidate positions of a +-8 neighbourhood (vtmhip_dist_uniform_batch_dev)
  IF         half + quarter sample planes (one H pass first/!last, one V pass !first/last) of all 16x16 blocks (vtmhip_if_batch_dev)
  TR         forward 2-D transform of 4096 random residual blocks per (type, W, H), and the fused xT/quant/dequant/xIT/SSE chain
             for the square sizes                                                                       (vtmhip_xT_batch_dev, vtmhip_tu_chain_batch_dev)
  BDOF / GEO bi-predicted 16x16 and 64x64 luma PUs with and without BDOF; 32x32 GEO blends             (vtmhip_bdof_batch_dev, vtmhip_weightedGeoBlk_batch_dev)

Prints one JSON object per line: units per second and the algorithmic GB/s (SURVEY.md 8d per-unit bytes / time -- a descriptive figure: the data is re-used
out of LDS / L1 / L2, so it is NOT an HBM utilisation; the kernels are bounded by instruction issue).
usage (GPU box): python3 scripts/microbench.py [--width 3840 --height 2160] > gpurun_out/microbench.json"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vtm_amd import synth   # noqa: E402
from vtm_amd.device import Context   # noqa: E402
from vtm_amd.lib import DistJob, DmvrJob, FracJob, FullJob, GeoBlendJob, IfJob, PicParams, PredJob, TrJob, TuJob   # noqa: E402



def timed(ctx, fn, reps=10):
    for _ in range(2):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop_ms() / reps


def emit(name, units, unit_name, alg_bytes, ms, **kw):
    print(json.dumps(dict(kernel=name, units=units, unit=unit_name, ms=round(ms, 4), G_units_per_s=round(units / ms / 1e6, 3),
                          alg_GBps=round(alg_bytes / ms / 1e6, 1), **kw)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--only", default="", help="'shapes': only the per-shape rows of the uniform fast paths (squares vs the split rectangles)")
    a = ap.parse_args()
    W, H = a.width, a.height
    ctx = Context(0)
    fr = synth.gen_frames(W, H, 2)
    cur = np.ascontiguousarray(fr[1])
    ref, roff, rs = synth.extend_plane(fr[0], margin=160)
    d_cur, d_ref = ctx.to_device(cur), ctx.to_device(ref)
    rng = np.random.default_rng(2)

    # ---- the uniform fast paths per block shape: fractional search (tiled kernel), +-4 exhaustive search (lane per candidate), TU chain -----------
    # Per-sample rates of the split rectangles next to the squares of the same area class (within 20 %).
    shapes = ((8, 8), (16, 8), (8, 16), (16, 16), (32, 8), (8, 32), (32, 16), (16, 32), (32, 32), (64, 16), (16, 64), (64, 32), (32, 64), (64, 64))
    for (w, h) in shapes:
        n = min(60000, (W // w) * (H // h))
        py, px = np.divmod(np.arange(n), W // w)
        fj = np.zeros(n, np.dtype(FracJob))
        fj["orgOff"], fj["refOff"] = (py * h) * W + px * w, roff + (py * h) * rs + px * w
        fj["orgStride"], fj["refStride"], fj["width"], fj["height"] = W, rs, w, h
        fj["intX"], fj["intY"] = rng.integers(-8, 9, n), rng.integers(-8, 9, n)
        fj["motionLambda"], fj["useHad"], fj["bitDepth"] = 8.0, 1, 10
        d_fj, d_fr = ctx.to_device(fj.view(np.uint8)), ctx.alloc(16 * n)
        ms = timed(ctx, lambda: ctx.frac_search_batch(d_cur.ptr, d_ref.ptr, d_fj.ptr, n, w, h, d_fr.ptr, uniform_square=True), reps=5)
        emit("frac_search_uniform", n * w * h, "PU samples", n * (24 * (w + 8) * (h + 8) + 144 * w * h) // 8, ms, size="%dx%d" % (w, h), pus=n,
             ns_per_sample=round(ms * 1e6 / (n * w * h), 4))
        if w <= 64 and h <= 64:
            uj = np.zeros(n, np.dtype(FullJob))
            uj["orgOff"], uj["refOff"] = fj["orgOff"], fj["refOff"]
            uj["orgStride"], uj["refStride"], uj["width"], uj["height"], uj["puX"], uj["puY"] = W, rs, w, h, px * w, py * h
            uj["motionLambda"], uj["searchRange"] = 8.0, 4
            uj["centerHor"], uj["centerVer"] = rng.integers(-8, 9, n) * 16, rng.integers(-8, 9, n) * 16
            d_uj, d_ur = ctx.to_device(uj.view(np.uint8)), ctx.alloc(32 * n)
            pic = PicParams(W, H, 128, 10, 0)
            ms = timed(ctx, lambda: ctx.full_search_batch(pic, d_cur.ptr, d_ref.ptr, d_uj.ptr, n, d_ur.ptr, uniform=(w, h)), reps=5)
            emit("full_search_uniform", n * w * h, "PU samples", n * 2 * ((w + 8) * (h + 8) + w * h), ms, size="%dx%d" % (w, h), pus=n,
                 ns_per_sample=round(ms * 1e6 / (n * w * h), 4))
            resi = rng.integers(-512, 512, (n, h, w)).astype(np.int16)
            ju = np.zeros(n, np.dtype(TuJob))
            ju["resiOff"] = ju["outOff"] = np.arange(n) * w * h
            ju["resiStride"], ju["width"], ju["height"], ju["qpPer"], ju["qpRem"], ju["bitDepth"] = w, w, h, 7, 2, 10
            d_resi, d_ju, d_r, d_lv = ctx.to_device(resi), ctx.to_device(ju.view(np.uint8)), ctx.alloc(16 * n), ctx.alloc(4 * n * w * h)
            ms = timed(ctx, lambda: ctx.tu_chain_batch(d_resi.ptr, d_ju.ptr, n, w, h, d_r.ptr, d_lv.ptr, None, uniform=True), reps=5)
            emit("tu_chain_uniform", n * w * h, "samples", 32 * n * w * h, ms, size="%dx%d" % (w, h), tus=n, ns_per_sample=round(ms * 1e6 / (n * w * h), 4))
    if a.only == "shapes":
        ctx.close()
        return

    # ---- SATD 8x8 grid -------------------------------------------------------------------------------------------------------
    nb = (W // 8) * (H // 8)
    d_out = ctx.alloc(4 * nb * 81)
    ms = timed(ctx, lambda: ctx.satd8_grid(d_cur.ptr, W, d_ref.ptr + 2 * roff, rs, W, H, 4, d_out.ptr))
    emit("satd8_grid", nb * 81, "8x8 block pairs", nb * 81 * 256, ms, picture="%dx%d" % (W, H))

    # ---- SAD / SATD per PU size (uniform batches: several small blocks per wave) ----------------------------------------------------
    for kind, kname in ((0, "sad_pu"), (1, "satd_pu")):
        for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (16, 8), (8, 16), (32, 8), (64, 16)):
            ss = 1 if (kind == 0 and h > 8 and w <= 64) else 0            # FEN sub-sampling rule of the integer search (RdCost.cpp:289-323, mode 2)
            npu = min(20000, (W // w) * (H // h))
            per = 32                                                       # candidates per PU
            n = npu * per
            jobs = np.zeros(n, np.dtype(DistJob))
            px = rng.integers(0, W // w, npu) * w
            py = rng.integers(0, H // h, npu) * h
            dx, dy = rng.integers(-8, 9, n), rng.integers(-8, 9, n)
            jobs["orgOff"] = np.repeat(py * W + px, per)
            jobs["curOff"] = roff + (np.repeat(py, per) + dy) * rs + np.repeat(px, per) + dx
            jobs["orgStride"], jobs["curStride"], jobs["width"], jobs["height"], jobs["subShift"], jobs["kind"] = W, rs, w, h, ss, kind
            d_jobs, d_res = ctx.to_device(jobs.view(np.uint8)), ctx.alloc(8 * n)
            ms = timed(ctx, lambda: ctx.dist_uniform_batch(d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, kind, w, h, ss, d_res.ptr), reps=5)
            emit(kname, n, "candidates", n * (4 * w * h >> ss), ms, size="%dx%d" % (w, h), subShift=ss)

    # ---- interpolation planes of all 16x16 blocks ----------------------------------------------------------------------------
    bx, by = np.meshgrid(np.arange(W // 16) * 16, np.arange(H // 16) * 16)
    bx, by = bx.reshape(-1), by.reshape(-1)
    nblk = bx.size
    taps_h = np.array([-1, 4, -11, 40, 40, -11, 4, -1], np.int16)   # half-sample luma filter
    taps_q = np.array([-1, 4, -10, 58, 17, -5, 1, 0], np.int16)     # quarter-sample luma filter
    for name, taps in (("if_half", taps_h), ("if_quarter", taps_q)):
        jh = np.zeros(nblk, np.dtype(IfJob))
        jh["srcOff"] = roff + (by - 3) * rs + bx
        jh["dstOff"] = np.arange(nblk) * 16 * 23
        jh["srcStride"], jh["dstStride"], jh["width"], jh["height"] = rs, 16, 16, 23
        jh["vertical"], jh["taps"], jh["isFirst"], jh["isLast"], jh["coeff"], jh["clipMax"], jh["bitDepth"] = 0, 8, 1, 0, taps, 1023, 10
        jv = np.zeros(nblk, np.dtype(IfJob))
        jv["srcOff"] = np.arange(nblk) * 16 * 23 + 3 * 16
        jv["dstOff"] = np.arange(nblk) * 256
        jv["srcStride"], jv["dstStride"], jv["width"], jv["height"] = 16, 16, 16, 16
        jv["vertical"], jv["taps"], jv["isFirst"], jv["isLast"], jv["coeff"], jv["clipMax"], jv["bitDepth"] = 1, 8, 0, 1, taps, 1023, 10
        d_tmp, d_pl = ctx.alloc(2 * nblk * 16 * 23), ctx.alloc(2 * nblk * 256)
        d_jh, d_jv = ctx.to_device(jh.view(np.uint8)), ctx.to_device(jv.view(np.uint8))

        def both():
            ctx.if_batch(d_ref.ptr, d_tmp.ptr, d_jh.ptr, nblk)
            ctx.if_batch(d_tmp.ptr, d_pl.ptr, d_jv.ptr, nblk)
        ms = timed(ctx, both, reps=5)
        samples = nblk * (16 * 23 + 256)
        emit(name, samples, "output samples", 4 * samples, ms, blocks=nblk)

    # ---- transforms ----------------------------------------------------------------------------------------------------------
    nblk = 4096 * 16    # SURVEY.md 8d asks for 4096 blocks per combination; 16 such sets per launch take it out of the launch-latency regime
    for ty, tname in ((0, "DCT2"), (2, "DST7"), (1, "DCT8")):
        for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (16, 4), (32, 8), (8, 32), (64, 16)):
            if ty != 0 and max(w, h) > 32:
                continue
            resi = rng.integers(-512, 512, (nblk, h, w)).astype(np.int16)
            jt = np.zeros(nblk, np.dtype(TrJob))
            jt["srcOff"] = jt["dstOff"] = np.arange(nblk) * w * h
            jt["srcStride"], jt["dstStride"], jt["width"], jt["height"], jt["typeHor"], jt["typeVer"], jt["bitDepth"] = w, w, w, h, ty, ty, 10
            d_resi, d_coef, d_jt = ctx.to_device(resi), ctx.alloc(4 * nblk * w * h), ctx.to_device(jt.view(np.uint8))
            ms = timed(ctx, lambda: ctx.xT_batch(d_resi.ptr, d_coef.ptr, d_jt.ptr, nblk, w, h, None), reps=5)
            emit("xT", nblk * w * h, "samples", 6 * nblk * w * h, ms, type=tname, size="%dx%d" % (w, h))
    for s in (8, 16, 32, 64):   # the uniform forward transform (MTS candidate pre-selection) next to the generic one
        resi = rng.integers(-512, 512, (nblk, s, s)).astype(np.int16)
        ju = np.zeros(nblk, np.dtype(TuJob))
        ju["resiOff"] = ju["outOff"] = np.arange(nblk) * s * s
        ju["resiStride"], ju["width"], ju["height"], ju["qpPer"], ju["qpRem"], ju["bitDepth"] = s, s, s, 7, 2, 10
        d_resi, d_ju, d_r, d_cf = ctx.to_device(resi), ctx.to_device(ju.view(np.uint8)), ctx.alloc(16 * nblk), ctx.alloc(4 * nblk * s * s)
        ms = timed(ctx, lambda: ctx.xT_uniform_batch(d_resi.ptr, d_ju.ptr, nblk, s, s, d_cf.ptr, d_r.ptr), reps=5)
        emit("xT_uniform", nblk * s * s, "samples", 6 * nblk * s * s, ms, type="DCT2", size="%dx%d" % (s, s))
    for s in (8, 16, 32, 64):
        resi = rng.integers(-512, 512, (nblk, s, s)).astype(np.int16)
        ju = np.zeros(nblk, np.dtype(TuJob))
        ju["resiOff"] = ju["outOff"] = np.arange(nblk) * s * s
        ju["resiStride"], ju["width"], ju["height"], ju["qpPer"], ju["qpRem"], ju["bitDepth"] = s, s, s, 7, 2, 10
        d_resi, d_ju, d_r, d_lv = ctx.to_device(resi), ctx.to_device(ju.view(np.uint8)), ctx.alloc(16 * nblk), ctx.alloc(4 * nblk * s * s)
        ms = timed(ctx, lambda: ctx.tu_chain_batch(d_resi.ptr, d_ju.ptr, nblk, s, s, d_r.ptr, d_lv.ptr, None, uniform=True), reps=5)
        emit("tu_chain", nblk * s * s, "samples", 32 * nblk * s * s, ms, type="DCT2", size="%dx%d" % (s, s))
    # ---- BDOF (bi-predicted luma PUs, both vectors fractional) and GEO blending ---------------------------------------------------------
    ref2, roff2, rs2 = synth.extend_plane(fr[1], margin=160)
    d_refs = ctx.to_device(np.concatenate([ref.reshape(-1), ref2.reshape(-1)]))
    for s in (16, 64):
        nx, ny = W // s, H // s
        n = nx * ny
        jp = np.zeros(n, np.dtype(PredJob))
        py, px = np.divmod(np.arange(n), nx)
        base = (py * s) * rs + px * s
        jp["refOff"][:, 0], jp["refOff"][:, 1] = roff + base, ref.size + roff2 + base
        jp["refStride"][:, 0] = jp["refStride"][:, 1] = rs
        jp["mv"] = rng.integers(-64, 65, (n, 2, 2)) | 1
        jp["predOff"], jp["predStride"] = (py * s) * W + px * s, W
        jp["width"], jp["height"], jp["mode"], jp["bitDepth"] = s, s, 2, 10
        d_jp, d_pred = ctx.to_device(jp.view(np.uint8)), ctx.alloc(2 * W * H)
        ms = timed(ctx, lambda: ctx.bdof_batch(0, d_refs.ptr, d_pred.ptr, 0, d_jp.ptr, n, s, s), reps=5)
        emit("bdof", n * s * s, "samples", 6 * n * s * s, ms, size="%dx%d" % (s, s), pus=n)
        ms = timed(ctx, lambda: ctx.motion_compensation_batch(0, d_refs.ptr, d_pred.ptr, 0, d_jp.ptr, n, s, s), reps=5)
        emit("bi_pred", n * s * s, "samples", 6 * n * s * s, ms, size="%dx%d" % (s, s), pus=n)
    # DMVR (luma): every 16x16 / 64x64 PU of the picture with merge vectors a little off the clip's motion, BDOF on where the cost allows
    for s in (16, 64):
        nx, ny = W // s, H // s
        n = nx * ny
        jd = np.zeros(n, np.dtype(DmvrJob))
        py, px = np.divmod(np.arange(n), nx)
        base = (py * s) * rs + px * s
        jd["refOff"][:, 0], jd["refOff"][:, 1] = roff + base, ref.size + roff2 + base
        jd["refStride"][:, 0] = jd["refStride"][:, 1] = rs
        off = rng.integers(-24, 25, (n, 2))
        jd["mv"][:, 0, :], jd["mv"][:, 1, :] = off, -off + rng.integers(-20, 21, (n, 2))
        jd["predOff"], jd["predStride"], jd["puX"], jd["puY"] = (py * s) * W + px * s, W, px * s, py * s
        jd["width"], jd["height"], jd["bitDepth"], jd["bioApplied"] = s, s, 10, 1
        regions = ((s + 15) // 16) ** 2
        d_jd, d_pred, d_mvd = ctx.to_device(jd.view(np.uint8)), ctx.alloc(2 * W * H), ctx.alloc(8 * n * regions)
        pic = PicParams(W, H, 128, 10, 0)
        ms = timed(ctx, lambda: ctx.dmvr_batch(pic, 0, d_refs.ptr, d_pred.ptr, 0, d_jd.ptr, n, s, s, d_mvd.ptr), reps=5)
        emit("dmvr", n * s * s, "samples", 6 * n * s * s, ms, size="%dx%d" % (s, s), pus=n,
             moved_subpus=int(np.count_nonzero(d_mvd.to_host(np.int32).reshape(-1, 2).any(axis=1))))
    M = 112
    wplane = rng.integers(0, 9, (M, M)).astype(np.int16)
    s, per = 32, 8
    nx, ny = W // s, H // s
    n = nx * ny * per
    jg = np.zeros(n, np.dtype(GeoBlendJob))
    py, px = np.divmod(np.arange(n) // per, nx)
    jg["src0Off"] = (py * s) * W + px * s
    jg["src1Off"] = W * H + jg["src0Off"]
    jg["dstOff"] = np.arange(n) * s * s
    jg["src0Stride"] = jg["src1Stride"] = W
    jg["dstStride"], jg["width"], jg["height"] = s, s, s
    sx = np.where(np.arange(n) % 3 == 0, -1, 1)
    jg["stepX"], jg["weightStride"] = sx, np.where(np.arange(n) % 2 == 0, -M, M)
    jg["weightOff"] = (np.where(np.arange(n) % 2 == 0, M - 1 - rng.integers(0, M - s, n), rng.integers(0, M - s, n))) * M + \
        np.where(sx < 0, M - 1 - rng.integers(0, M - s, n), rng.integers(0, M - s, n))
    src = rng.integers(-8192, 8192, 2 * W * H).astype(np.int16)
    d_src, d_w, d_jg, d_dst = ctx.to_device(src), ctx.to_device(wplane), ctx.to_device(jg.view(np.uint8)), ctx.alloc(2 * n * s * s)
    ms = timed(ctx, lambda: ctx.weightedGeoBlk_batch(d_src.ptr, d_dst.ptr, d_w.ptr, d_jg.ptr, n), reps=5)
    emit("geo_blend", n * s * s, "samples", 6 * n * s * s, ms, size="32x32", blends=n)
    ctx.close()


if __name__ == "__main__":
    main()
