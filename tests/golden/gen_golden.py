#!/usr/bin/env python3
"""Records golden input/output vectors from the REAL reference (VTM 9.3), compiled in this container by
oracle/Makefile.ref into oracle/_ref/libvtmref.so.  Run here only (the reference does not exist on the GPU box):

    make -f oracle/Makefile.ref -j8 && python tests/golden/gen_golden.py

Outputs small .npz fixtures next to this script.  They are DATA: inputs (or the seeds of the synthetic clip they are
cut from) and the values the reference returned -- no reference source.  tests/test_oracle_golden.py replays them
against the plain-C oracle (CPU) and tests/test_gpu_golden.py against the HIP kernels (GPU).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import me_util   # noqa: E402
import oracle_lib as ol   # noqa: E402

R = ol.ref()
rng = np.random.default_rng(20260101)


def gen_dist():
    """DistParam::distFunc through RdCost::m_afpDistortFunc (AVX2 dispatch) AND the scalar members: SAD/SATD/SSE."""
    orgs, curs, meta, exp = [], [], [], []
    sizes = [(w, h) for w in (4, 8, 12, 16, 24, 32, 48, 64, 128) for h in (4, 8, 16, 32, 64, 128)]
    for (w, h) in sizes:
        for variant in range(3):
            if variant == 0:
                o = rng.integers(0, 1024, (h, w))
                c = rng.integers(0, 1024, (h, w))
            elif variant == 1:   # bi-pred ME target 2*org - pred, unclipped
                o = 2 * rng.integers(0, 1024, (h, w)) - rng.integers(0, 1024, (h, w))
                c = rng.integers(0, 1024, (h, w))
            else:                # edge cases: flat / checkerboard
                yy, xx = np.mgrid[0:h, 0:w]
                o = np.where((yy + xx) & 1, 1023, 0)
                c = np.where((yy + xx) & 1, 0, 1023) if (w + h) % 3 else np.full((h, w), 1023)
            o, c = ol.i16(o), ol.i16(c)
            for kind in (0, 1, 2):
                for ss in ((0, 1) if (kind == 0 and h >= 8) else (0,)):
                    simd = R.ref_dist(kind, 1, ol.P(o), w, ol.P(c), w, w, h, 10, ss)
                    scal = R.ref_dist(kind, 0, ol.P(o), w, ol.P(c), w, w, h, 10, ss)
                    assert simd == scal, (kind, w, h, ss, simd, scal)
                    meta.append((len(orgs), w, h, kind, ss))
                    exp.append(simd)
            orgs.append(o.reshape(-1))
            curs.append(c.reshape(-1))
    lens = np.array([a.size for a in orgs])
    np.savez_compressed(os.path.join(HERE, "dist.npz"), org=np.concatenate(orgs), cur=np.concatenate(curs),
                        starts=np.concatenate([[0], np.cumsum(lens)]), meta=np.array(meta, np.int32), exp=np.array(exp, np.uint64))
    print("dist:", len(exp), "cases")


def gen_mvcost():
    rows = []
    for _ in range(400):
        lam = float(rng.uniform(0.5, 90))
        ph, pv = int(rng.integers(-4000, 4000)), int(rng.integers(-4000, 4000))
        cs = int(rng.integers(0, 3))
        x, y = int(rng.integers(-600, 600)), int(rng.integers(-600, 600))
        imv = int(rng.choice([0, 0, 1, 2, 4]))
        rows.append((lam, ph, pv, cs, x, y, imv, R.ref_mv_cost(lam, ph, pv, cs, x, y, imv)))
    np.savez_compressed(os.path.join(HERE, "mvcost.npz"), rows=np.array(rows, np.float64))
    print("mvcost:", len(rows))


def gen_if():
    """InterpolationFilter::filterHor / filterVer (public entry points, SIMD dispatch == scalar checked here)."""
    src10 = ol.i16(rng.integers(0, 1024, (160, 160)))
    src14 = ol.i16(rng.integers(-8192, 8191, (160, 160)))
    meta, outs = [], []
    ss, off = 160, 8 * 160 + 8
    for (w, h) in ((4, 4), (4, 11), (8, 8), (16, 16), (17, 24), (64, 72), (129, 136)):
        for comp, fracs in ((0, (0, 1, 4, 8, 12, 15)), (1, (0, 1, 8, 16, 31))):
            for frac in fracs:
                for vertical in (0, 1):
                    for isFirst in ((1,) if not vertical else (0, 1)):
                        for isLast in (0, 1):
                            for alt in ((0, 1) if (comp == 0 and frac == 8) else (0,)):
                                src = src10 if isFirst else src14
                                d = [np.zeros((h, w), np.int16) for _ in range(2)]
                                for simd in (0, 1):
                                    if vertical:
                                        R.ref_if_ver(simd, comp, C.c_void_p(src.ctypes.data + 2 * off), ss, ol.P(d[simd]), w, w, h, frac,
                                                     isFirst, isLast, 10, 0, 0, alt)
                                    else:
                                        R.ref_if_hor(simd, comp, C.c_void_p(src.ctypes.data + 2 * off), ss, ol.P(d[simd]), w, w, h, frac,
                                                     isLast, 10, 0, 0, alt)
                                assert np.array_equal(d[0], d[1])
                                meta.append((w, h, comp, frac, vertical, isFirst, isLast, alt, sum(o.size for o in outs)))
                                outs.append(d[0].reshape(-1))
    np.savez_compressed(os.path.join(HERE, "interp.npz"), src10=src10, src14=src14, meta=np.array(meta, np.int32), out=np.concatenate(outs))
    print("interp:", len(meta), "cases")


def gen_tr():
    """fastFwdTrans / fastInvTrans table entries with the (shift, line, skip) combinations xT / xIT produce, plus the 14 core matrices."""
    mats = {}
    for t in range(3):
        for n in (2, 4, 8, 16, 32, 64):
            m = np.zeros((n, n), np.int16)
            if R.ref_tr_matrix(t, n, 0, ol.P(m)) == 0:
                mats["m_%d_%d" % (t, n)] = m
    meta, ins, outs = [], [], []
    for t in range(3):
        for n in (2, 4, 8, 16, 32, 64):
            if "m_%d_%d" % (t, n) not in mats:
                continue
            skip2 = 16 if (t != 0 and n == 32) else (n - 32 if n > 32 else 0)
            for line in (1, 4, 8, 32, 64):
                sk1 = 16 if line == 32 else (32 if line == 64 else 0)
                for (a, b) in {(0, 0), (sk1, skip2)}:
                    for shift, amp in ((2, 1024), (7, 32768), (10, 32768)):
                        src = rng.integers(-amp, amp, line * n).astype(np.int32)
                        for inv in (0, 1):
                            dst = np.zeros(line * n, np.int32)
                            if inv:
                                s2 = src.reshape(n, line).copy()
                                if b:
                                    s2[n - b:, :] = 0
                                if a:
                                    s2[:, line - a:] = 0
                                s2 = s2.reshape(-1)
                                R.ref_inv_trans(t, int(np.log2(n)) - 1, ol.P(s2), ol.P(dst), shift, line, a, b, -32768, 32767)
                                ins.append(s2)
                            else:
                                R.ref_fwd_trans(t, int(np.log2(n)) - 1, ol.P(src), ol.P(dst), shift, line, a, b)
                                ins.append(src)
                            meta.append((t, n, line, a, b, shift, inv, sum(o.size for o in outs)))
                            outs.append(dst)
    np.savez_compressed(os.path.join(HERE, "transform.npz"), meta=np.array(meta, np.int32), src=np.concatenate(ins), dst=np.concatenate(outs), **mats)
    print("transform:", len(meta), "cases,", len(mats), "matrices")


def gen_me():
    """InterSearch::xTZSearch / xPatternSearch / xPatternSearchFracDIF on the seeded synthetic clip (vtm_amd.synth)."""
    scene = me_util.Scene(416, 240, hard=True)
    jobs = me_util.random_tz_jobs(scene, 400, seed=77)
    res = []
    for j in jobs:
        org = np.ascontiguousarray(scene.cur[j["y"]:j["y"] + j["h"], j["x"]:j["x"] + j["w"]])
        c = me_util.oracle_ctx(scene, j, org)
        t = me_util.oracle_tz_job(j)
        r = ol.MeResult()
        R.ref_tz_search(C.byref(c), C.byref(t), C.byref(r))
        res.append((r.mvX, r.mvY, r.cost, r.dist))
    frac_jobs, frac_res, full_res = [], [], []
    frng = np.random.default_rng(78)
    for k in range(200):
        w = int(frng.choice([8, 16, 32, 64, 128, 4, 16, 8]))
        h = int(frng.choice([8, 16, 32, 64, 128, 8, 16]))
        x = int(frng.integers(0, (416 - w) // 4 + 1)) * 4
        y = int(frng.integers(0, (240 - h) // 4 + 1)) * 4
        j = dict(w=w, h=h, x=x, y=y, subShift=0, lam=float(frng.uniform(1, 40)), predHor=int(frng.integers(-64, 64)),
                 predVer=int(frng.integers(-64, 64)), intX=int(frng.integers(-12, 12)), intY=int(frng.integers(-12, 12)), useHad=int(k % 4 != 0))
        org = np.ascontiguousarray(scene.cur[y:y + h, x:x + w])
        c = me_util.oracle_ctx(scene, j, org)
        fr = ol.FracResult()
        R.ref_frac_search(C.byref(c), j["intX"], j["intY"], j["useHad"], 0, C.byref(fr))
        frac_jobs.append((w, h, x, y, j["lam"], j["predHor"], j["predVer"], j["intX"], j["intY"], j["useHad"]))
        frac_res.append((fr.halfX, fr.halfY, fr.qterX, fr.qterY, fr.cost))
        # bi-pred style exhaustive +-4 search around (intX, intY) (xSetSearchRange + xPatternSearch)
        rg = (C.c_int * 4)()
        R.ref_set_search_range(C.byref(c), j["intX"] * 16, j["intY"] * 16, 4, rg)
        c.subShift = 1 if (h > 8 and w <= 64) else 0
        r = ol.MeResult()
        R.ref_full_search(C.byref(c), rg, C.byref(r))
        full_res.append((rg[0], rg[1], rg[2], rg[3], r.mvX, r.mvY, r.cost, r.dist))
    import json
    np.savez_compressed(os.path.join(HERE, "me.npz"), tz_jobs=np.array([json.dumps(j) for j in jobs]), tz_res=np.array(res, np.int64),
                        frac_jobs=np.array(frac_jobs, np.float64), frac_res=np.array(frac_res, np.int64), full_res=np.array(full_res, np.int64))
    print("me:", len(jobs), "TZ,", len(frac_jobs), "frac/full")


def gen_misc():
    """Affine gradient (Sobel + normal equations) and the bi-pred buffer ops."""
    out = {}
    k = 0
    for (w, h) in ((16, 16), (32, 16), (64, 64), (128, 32)):
        pred = ol.i16(rng.integers(0, 1024, (h, w)))
        resi = ol.i16(rng.integers(-512, 512, (h, w)))
        gx, gy = np.zeros((h, w), np.int32), np.zeros((h, w), np.int32)
        for simd in (0, 1):
            g0, g1 = np.zeros((h, w), np.int32), np.zeros((h, w), np.int32)
            R.ref_sobel(simd, 0, ol.P(pred), w, ol.P(g0), w, w, h)
            R.ref_sobel(simd, 1, ol.P(pred), w, ol.P(g1), w, w, h)
            if simd:
                assert np.array_equal(g0, gx) and np.array_equal(g1, gy)
            gx, gy = g0, g1
        for six in (0, 1):
            eq = np.zeros((7, 7), np.int64)
            ptrs = (C.c_void_p * 2)(gx.ctypes.data, gy.ctypes.data)
            R.ref_equal_coeff(0, ol.P(resi), w, ptrs, w, ol.P(eq), w, h, six)
            eq2 = np.zeros((7, 7), np.int64)
            R.ref_equal_coeff(1, ol.P(resi), w, ptrs, w, ol.P(eq2), w, h, six)
            assert np.array_equal(eq, eq2)
            out["eq_%d_%d" % (k, six)] = eq
        out["pred_%d" % k], out["resi_%d" % k], out["gx_%d" % k], out["gy_%d" % k] = pred, resi, gx, gy
        # removeHighFreq / addAvg
        org = ol.i16(rng.integers(0, 1024, (h, w)))
        rh = org.copy()
        R.ref_remove_high_freq(ol.P(rh), w, ol.P(pred), w, w, h)
        a14 = ol.i16(rng.integers(-8192, 8191, (h, w)))
        b14 = ol.i16(rng.integers(-8192, 8191, (h, w)))
        av = np.zeros((h, w), np.int16)
        R.ref_add_avg(ol.P(a14), w, ol.P(b14), w, ol.P(av), w, w, h, 10)
        out["org_%d" % k], out["rhf_%d" % k], out["a14_%d" % k], out["b14_%d" % k], out["avg_%d" % k] = org, rh, a14, b14, av
        k += 1
    out["count"] = np.array([k])
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **out)
    print("misc:", k, "cases")


def gen_quant():
    """Quant::quant / Quant::dequant (flat scaling list) through the TransformUnit rig of oracle/ref_shim_me.cpp."""
    meta, coefs, qs, dqs, sums = [], [], [], [], []
    for w in (4, 8, 16, 32, 64):
        for h in (4, 8, 16, 32, 64):
            for qp in (22, 32, 37, 51):
                for irap in (0, 1):
                    c = rng.integers(-32768, 32768, w * h).astype(np.int32)
                    c[rng.random(w * h) < 0.6] //= 128
                    # blocks wider / taller than 32: the transform zeroes everything outside the top-left 32x32 (TrQuant.cpp:790-805) and
                    # Quant::quant walks a scan table that only covers that region (its loop bound is the full area, so it reads past the
                    # table for such blocks -- harmless there because those coefficients are zero).  Golden inputs respect the zero-out.
                    c2 = c.reshape(h, w)
                    c2[32:, :] = 0
                    c2[:, 32:] = 0
                    q, d, s = np.zeros(w * h, np.int32), np.zeros(w * h, np.int32), C.c_int32()
                    R.ref_quant_dequant(ol.P(c), w, h, 10, qp, irap, ol.P(q), C.byref(s), ol.P(d))
                    meta.append((w, h, qp, irap, sum(a.size for a in coefs), s.value))
                    coefs.append(c)
                    qs.append(q)
                    dqs.append(d)
    np.savez_compressed(os.path.join(HERE, "quant.npz"), meta=np.array(meta, np.int64), coef=np.concatenate(coefs), q=np.concatenate(qs),
                        dq=np.concatenate(dqs))
    print("quant:", len(meta), "cases")


def gen_mest():
    """InterSearch::xMotionEstimation through the rig of oracle/ref_shim_me.cpp (ref_motion_estimation): jobs as JSON (the clip is
    re-synthesised from its seed), results = rcMv, rcMvPred, riMVPIdx, ruiBits, ruiCost."""
    import json
    scene = me_util.Scene(416, 240, hard=True)
    jobs, res, cfgs = [], [], []
    for ci, cfgv in enumerate(((4, 1, 1, 0, 1), (4, 0, 1, 0, 0), (4, 1, 0, 1, 1))):
        cfg = ol.MestCfg(*cfgv)
        for j in me_util.random_mest_jobs(scene, 90, seed=900 + ci):
            keep = []
            t = me_util.oracle_mest_job(scene, j, keep)
            r = ol.MestResult()
            R.ref_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
            jobs.append(json.dumps(j))
            cfgs.append(cfgv)
            res.append(r.key())
    np.savez_compressed(os.path.join(HERE, "mest.npz"), jobs=np.array(jobs), cfg=np.array(cfgs, np.int32), res=np.array(res, np.int64))
    print("mest:", len(jobs), "jobs")


def gen_mc():
    """InterPrediction::xPredInterBlk through ref_pred_inter_blk (oracle/ref_shim_me.cpp): luma and 4:2:0 chroma blocks of the synthetic
    clip (re-synthesised from its seed), rows = (comp, x, y, w, h, mvHor, mvVer, bi, imv), outputs concatenated."""
    from vtm_amd import synth
    W, H, m = 416, 240, 64
    y, u, v = synth.gen_frames(W, H, 1, chroma=True)[0]
    yb, yo, ys = synth.extend_plane(y, m)
    ub, uo, us = synth.extend_plane(u, m // 2)
    g = np.random.default_rng(555)
    meta, outs = [], []
    t = 0
    while len(meta) < 240:
        t += 1
        w, h = int(g.choice([4, 8, 16, 32, 64, 128])), int(g.choice([4, 8, 16, 32, 64, 128]))
        if w == 4 and h == 4:
            continue
        x, yy = int(g.integers(0, (W - w) // 4 + 1)) * 4, int(g.integers(0, (H - h) // 4 + 1)) * 4
        mvh, mvv = int(g.integers(-20 * 16, 20 * 16)), int(g.integers(-20 * 16, 20 * 16))
        if t % 5 == 0:
            mvh &= ~15
        if t % 7 == 0:
            mvv &= ~15
        if t % 11 == 0:
            mvh = (mvh & ~15) | 8
        bi, imv, comp = t % 2, 3 if t % 3 == 0 else 0, (0, 1, 2)[t % 3]
        cw, ch = (w // 2, h // 2) if comp else (w, h)
        out = np.zeros((ch, cw), np.int16)
        R.ref_pred_inter_blk(comp, C.c_void_p(yb.ctypes.data + 2 * yo), ys, C.c_void_p(ub.ctypes.data + 2 * uo), us, W, H, x, yy, w, h, mvh, mvv,
                             bi, 10, imv, ol.P(out), cw)
        meta.append((comp, x, yy, w, h, mvh, mvv, bi, imv))
        outs.append(out.reshape(-1))
    np.savez_compressed(os.path.join(HERE, "mc.npz"), meta=np.array(meta, np.int32), out=np.concatenate(outs))
    print("mc:", len(meta), "blocks")


def gen_masked():
    """RdCost::xGetSADwMask through the table entry (x86) -- ref_sad_mask: blocks, the 112x112 weight plane, walk parameters, results."""
    R.ref_sad_mask.restype = C.c_uint64
    g = np.random.default_rng(808)
    M = 112
    plane = ol.i16(g.integers(0, 9, (M, M)))
    meta, orgs, curs, res = [], [], [], []
    for k in range(160):
        w, h = int(g.choice([8, 16, 32, 64])), int(g.choice([8, 16, 32, 64]))
        org, cur = ol.i16(g.integers(0, 1024, (h, w))), ol.i16(g.integers(0, 1024, (h, w)))
        sx, rd = (1 if k % 3 else -1), (1 if k % 2 else -1)
        x0 = int(g.integers(0, M - w)) + (w - 1 if sx < 0 else 0)
        y0 = int(g.integers(0, M - h)) + (h - 1 if rd < 0 else 0)
        off, ms, ms2 = y0 * M + x0, rd * M, -sx * w
        v = R.ref_sad_mask(1, ol.P(org), w, ol.P(cur), w, w, h, 10, C.c_void_p(plane.ctypes.data + 2 * off), ms, sx, ms2)
        meta.append((w, h, off, ms, sx, ms2))
        orgs.append(org.reshape(-1)); curs.append(cur.reshape(-1)); res.append(v)
    np.savez_compressed(os.path.join(HERE, "masked.npz"), plane=plane, meta=np.array(meta, np.int32), org=np.concatenate(orgs), cur=np.concatenate(curs),
                        res=np.array(res, np.uint64))
    print("masked:", len(meta), "cases")


def gen_geo():
    """InterpolationFilter::m_weightedGeoBlk (x86 entry) -- ref_weighted_geo_blk: the six prestored weight planes (data the reference builds in
    initGeoTemplate), per case the walk a trampoline derives (ref_geo_walk), both 14-bit predictions and the blended block."""
    M = 112
    planes = np.zeros((6, M, M), np.int16)
    for i in range(6):
        R.ref_geo_weights(i, ol.P(planes[i]))
    g = np.random.default_rng(909)
    meta, s0s, s1s, outs = [], [], [], []
    for k in range(128):
        split = k % 64
        lw, lh = int(g.choice([8, 16, 32, 64], p=[.3, .3, .25, .15])), int(g.choice([8, 16, 32, 64], p=[.3, .3, .25, .15]))
        comp = k % 3 if k >= 64 else 0
        w, h = (lw >> 1, lh >> 1) if comp else (lw, lh)
        s0, s1 = ol.i16(g.integers(-8192, 8192 + 1023 * 16, (h, w))), ol.i16(g.integers(-8192, 8192 + 1023 * 16, (h, w)))
        walk = (C.c_int * 4)()
        R.ref_geo_walk(split, comp, lw, lh, walk)
        dst = np.zeros((h, w), np.int16)
        R.ref_weighted_geo_blk(1, split, comp, lw, lh, ol.P(s0), w, ol.P(s1), w, ol.P(dst), w, 10)
        meta.append((split, comp, w, h) + tuple(walk))
        s0s.append(s0.reshape(-1)); s1s.append(s1.reshape(-1)); outs.append(dst.reshape(-1))
    np.savez_compressed(os.path.join(HERE, "geo.npz"), planes=planes, meta=np.array(meta, np.int32), src0=np.concatenate(s0s), src1=np.concatenate(s1s),
                        out=np.concatenate(outs))
    print("geo:", len(meta), "blends")


def gen_bdof():
    """BDOF of bi-predicted luma PUs through the reference's own xPredInterBlk(bioApplied) + applyBiOptFlow with the x86 buffer ops (ref_bdof_pu):
    two padded reference planes, per PU position / size / both vectors, and the refined prediction."""
    from vtm_amd import synth
    W, H, M = 160, 96, 40
    fr = list(synth.gen_frames(W, H, 3, seed=21))
    planes = [np.ascontiguousarray(np.pad(f.astype(np.int16), M, mode="edge")) for f in (fr[0], fr[2])]
    S = planes[0].shape[1]
    org = [C.c_void_p(p.ctypes.data + 2 * (M * S + M)) for p in planes]
    g = np.random.default_rng(1010)
    meta, outs = [], []
    sizes = [(8, 16), (16, 8), (16, 16), (32, 16), (16, 32), (32, 32), (64, 32), (64, 64), (128, 64), (8, 64)]
    for k in range(40):
        w, h = sizes[k % len(sizes)]
        x, y = int(g.integers(0, (W - w) // 4 + 1)) * 4, int(g.integers(0, (H - h) // 4 + 1)) * 4
        mv = [int(v) for v in g.integers(-400, 400, 4)]
        if k % 5 == 0:
            mv[0] &= ~15
        if k % 7 == 0:
            mv[3] &= ~15
        if k % 9 == 0:
            mv = [v & ~15 for v in mv]
        dst = np.zeros((h, w), np.int16)
        R.ref_bdof_pu(1, org[0], org[1], S, W, H, x, y, w, h, *mv, 10, ol.P(dst), w)
        meta.append((x, y, w, h, *mv))
        outs.append(dst.reshape(-1))
    np.savez_compressed(os.path.join(HERE, "bdof.npz"), planes=np.stack(planes), dims=np.array([W, H, M], np.int32), meta=np.array(meta, np.int32),
                        out=np.concatenate(outs))
    print("bdof:", len(meta), "PUs")


def gen_dmvr():
    """DMVR of bi-predicted 4:2:0 PUs through the reference's own InterPrediction::xProcessDMVR (ref_dmvr_pu420): the refined luma and chroma
    predictions and pu.mvdL0SubPu.  Luma planes: those of bdof.npz (same generator call); the chroma planes are stored here."""
    from vtm_amd import synth
    W, H, M = 160, 96, 40
    fr = list(synth.gen_frames(W, H, 3, seed=21, chroma=True))
    P = [[np.ascontiguousarray(np.pad(f[c].astype(np.int16), M if c == 0 else M // 2, mode="edge")) for c in range(3)] for f in (fr[0], fr[2])]
    assert np.array_equal(np.stack([P[0][0], P[1][0]]), np.load(os.path.join(HERE, "bdof.npz"))["planes"])
    SY, SC = P[0][0].shape[1], P[0][1].shape[1]
    planes = ((C.c_void_p * 3) * 2)()
    for l in range(2):
        for c in range(3):
            m, st = (M, SY) if c == 0 else (M // 2, SC)
            planes[l][c] = P[l][c].ctypes.data + 2 * (m * st + m)
    g = np.random.default_rng(1013)
    meta, outs, outc, mvds = [], [], [], []
    sizes = [(8, 16), (16, 8), (16, 16), (32, 16), (16, 32), (32, 32), (64, 32), (64, 64), (128, 64), (8, 64)]
    for k in range(40):
        w, h = sizes[k % len(sizes)]
        x, y = int(g.integers(0, (W - w) // 8 + 1)) * 8, int(g.integers(0, (H - h) // 8 + 1)) * 8
        base = np.array([48, 32]) + g.integers(-40, 41, 2)      # near the clip's true pan, so that the refinement has something to find
        mv = [int(-base[0]), int(-base[1]), int(base[0] + g.integers(-24, 25)), int(base[1] + g.integers(-24, 25))]
        if k % 5 == 0:
            mv[0] &= ~15
        if k % 7 == 0:
            mv[3] &= ~15
        if k % 9 == 0:
            mv = [v & ~15 for v in mv]
        bio = k % 2
        nsub = (w // min(w, 16)) * (h // min(h, 16))
        d = [np.zeros((h, w), np.int16), np.zeros((h // 2, w // 2), np.int16), np.zeros((h // 2, w // 2), np.int16)]
        dp = (C.c_void_p * 3)(*[a.ctypes.data for a in d])
        mvd = np.zeros(2 * nsub, np.int32)
        R.ref_dmvr_pu420(planes, SY, SC, W, H, 128, x, y, w, h, *mv, 10, bio, dp, w, w // 2, C.c_void_p(mvd.ctypes.data))
        meta.append((x, y, w, h, *mv, bio))
        outs.append(d[0].reshape(-1)); outc.append(np.concatenate([d[1].reshape(-1), d[2].reshape(-1)])); mvds.append(mvd)
    np.savez_compressed(os.path.join(HERE, "dmvr.npz"), dims=np.array([W, H, M], np.int32), meta=np.array(meta, np.int32), out=np.concatenate(outs),
                        outc=np.concatenate(outc), mvd=np.concatenate(mvds), planesC=np.stack([np.stack(P[0][1:]), np.stack(P[1][1:])]))
    print("dmvr:", len(meta), "PUs,", int(np.count_nonzero(np.concatenate(mvds))), "non-zero vector components")


def gen_lfnst():
    """TrQuant::fwdLfnstNxN / invLfnstNxN (ref_lfnst) on random coefficient vectors.  The trained core matrices are INPUT DATA of these functions
    (the integration passes the reference's arrays to the device library): stored with the vectors so that the GPU box can replay them."""
    m8, m4 = np.zeros((4, 2, 16, 48), np.int8), np.zeros((4, 2, 16, 16), np.int8)
    R.ref_lfnst_tables(C.c_void_p(m8.ctypes.data), C.c_void_p(m4.ctypes.data))
    g = np.random.default_rng(1020)
    meta, src, out = [], [], []
    for k in range(256):
        inverse, mode, index, size, zo = k & 1, int(g.integers(0, 4)), int(g.integers(0, 2)), int(g.choice([4, 8])), int(g.choice([8, 16]))
        n = 48 if size > 4 else 16
        s_ = np.zeros(48, np.int32)
        lim = 32768 if k % 9 == 0 else 2048
        s_[:(zo if inverse else n)] = g.integers(-lim, lim, zo if inverse else n)
        d = np.zeros(48, np.int32)
        R.ref_lfnst(inverse, ol.P(s_), ol.P(d), mode, index, size, zo)
        meta.append((inverse, mode, index, size, zo)); src.append(s_); out.append(d)
    np.savez_compressed(os.path.join(HERE, "lfnst.npz"), m8=m8, m4=m4, meta=np.array(meta, np.int32), src=np.stack(src), out=np.stack(out))
    print("lfnst:", len(meta), "vectors")


def gen_affine_me():
    """InterPrediction::xPredAffineBlk and InterSearch::xAffineMotionEstimation as the real members (oracle/ref_shim_me.cpp) on random affine jobs;
    m_hevcCost is set relative to the cost of the unrestricted run so that both outcomes of the refinement gate occur"""
    import json
    scene = me_util.Scene(416, 240, hard=False)
    jobs = me_util.random_affine_jobs(scene, 120, seed=777)
    mvs, bits, costs, hevc, preds = [], [], [], [], []
    for j in jobs:
        keep = []
        t = me_util.affine_me_struct(scene, j, keep)
        t.hevcCost = 1 << 62
        r0 = ol.AffineMeResult()
        R.ref_affine_motion_estimation(C.byref(t), C.byref(r0))
        t.hevcCost = int(r0.cost * j["hevc_scale"])
        r = ol.AffineMeResult()
        R.ref_affine_motion_estimation(C.byref(t), C.byref(r))
        hevc.append(t.hevcCost)
        mvs.append([list(v) for v in r.mv])
        bits.append(r.bits)
        costs.append(r.cost)
        p = me_util.affine_pred_struct(scene, j)
        mv = ((C.c_int * 2) * 3)(*[(C.c_int * 2)(*v) for v in j["mv"]])
        a = np.zeros((j["h"], j["w"]), np.int16)
        R.ref_pred_affine_blk(C.byref(p), mv, 0, ol.P(a), j["w"])
        preds.append(a.reshape(-1))
    np.savez_compressed(os.path.join(HERE, "affine_me.npz"), jobs=np.array([json.dumps(j) for j in jobs]), mv=np.array(mvs, np.int32), bits=np.array(bits, np.int64),
                        cost=np.array(costs, np.int64), hevc=np.array(hevc, np.int64), pred=np.concatenate(preds))
    print("affine_me:", len(jobs), "jobs")


def gen_smvd():
    """InterSearch::xGetSymmetricCost, xSymmetricMotionEstimation and symmvdCheckBestMvp as the real members (oracle/ref_shim_me.cpp) on random SMVD jobs"""
    import json
    scene = me_util.SmvdScene(416, 240)
    jobs = me_util.random_smvd_jobs(scene, 150, seed=4242)
    rows = []
    for j in jobs:
        c0, me, chk = me_util.smvd_member_results(scene, j, R, "ref_")
        rows.append([c0, *me[0], *me[1], me[2], *chk[0], *chk[1], *chk[2], chk[3]])
    np.savez_compressed(os.path.join(HERE, "smvd.npz"), jobs=np.array([json.dumps(j) for j in jobs]), out=np.array(rows, np.int64))
    print("smvd:", len(jobs), "jobs")


if __name__ == "__main__":
    if len(sys.argv) > 1:   # regenerate selected fixtures only: gen_golden.py mest quant ...
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_dist()
    gen_mvcost()
    gen_if()
    gen_tr()
    gen_me()
    gen_misc()
    gen_quant()
    gen_mest()
    gen_mc()
    gen_masked()
    gen_geo()
    gen_bdof()
    gen_dmvr()
    gen_lfnst()
    gen_affine_me()
    gen_smvd()
