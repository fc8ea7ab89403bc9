"""GPU parity: luma motion compensation, removeHighFreq / addAvg and the exhaustive (bi-pred refinement) search vs the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import me_util
import oracle_lib as ol
from vtm_amd.lib import FullJob, McJob, MeResult, PelOpJob, PicParams

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_mc_luma_matches_oracle(ctx):
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    rng = np.random.default_rng(51)
    n = 600
    jobs = (McJob * n)()
    dst_stride = 136
    exp = np.zeros((n * 128, dst_stride), np.int16)
    meta = []
    for k in range(n):
        w = int(rng.choice([4, 8, 16, 32, 64, 128, 8, 16]))
        h = int(rng.choice([4, 8, 16, 32, 64, 128, 8, 16]))
        x = int(rng.integers(0, (416 - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (240 - h) // 4 + 1)) * 4
        mvh, mvv = int(rng.integers(-40 * 16, 40 * 16)), int(rng.integers(-30 * 16, 30 * 16))
        if k % 5 == 0:
            mvh &= ~15
        if k % 7 == 0:
            mvv &= ~15
        if k % 11 == 0:
            mvh, mvv = (mvh & ~15) | 8, (mvv & ~15) | 8
        bi, alt = k % 3 == 0, k % 13 == 0
        j = jobs[k]
        j.refOff = scene.ref_off + y * scene.ref_stride + x
        j.dstOff, j.refStride, j.dstStride, j.width, j.height = k * 128 * dst_stride, scene.ref_stride, dst_stride, w, h
        j.mvHor, j.mvVer, j.bi, j.bitDepth, j.useAltHpelIf = mvh, mvv, bi, 10, alt
        e = np.zeros((h, dst_stride), np.int16)
        L.vo_mc_luma(C.c_void_p(scene.ref_buf.ctypes.data + 2 * j.refOff), scene.ref_stride, w, h, mvh, mvv, int(bi), 10, int(alt), ol.P(e), dst_stride)
        exp[k * 128:k * 128 + h, :w] = e[:, :w]
        meta.append((w, h, mvh, mvv, bi, alt))
    d_ref = ctx.to_device(scene.ref_buf)
    d_dst = ctx.to_device(np.zeros((n * 128, dst_stride), np.int16))
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    ctx.mc_luma_batch(d_ref.ptr, d_dst.ptr, d_jobs.ptr, n, 128, 128)
    got = d_dst.to_host().reshape(n * 128, dst_stride)
    for k, (w, h, *_rest) in enumerate(meta):
        assert np.array_equal(got[k * 128:k * 128 + h, :w], exp[k * 128:k * 128 + h, :w]), (k, meta[k])


def test_pel_ops_match_reference_golden(ctx):
    z = np.load(os.path.join(G, "misc.npz"))
    for k in range(int(z["count"][0])):
        org, pred, a14, b14 = z["org_%d" % k], z["pred_%d" % k], z["a14_%d" % k], z["b14_%d" % k]
        h, w = org.shape
        jobs = (PelOpJob * 1)()
        j = jobs[0]
        j.aOff = j.bOff = j.dstOff = 0
        j.aStride = j.bStride = j.dstStride = w
        j.width, j.height, j.bitDepth = w, h, 10
        d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
        d_a, d_b, d_o = ctx.to_device(org), ctx.to_device(pred), ctx.to_device(np.zeros((h, w), np.int16))
        ctx.remove_high_freq_batch(d_a.ptr, d_b.ptr, d_o.ptr, d_jobs.ptr, 1)
        assert np.array_equal(d_o.to_host(), z["rhf_%d" % k])
        d_a, d_b = ctx.to_device(a14), ctx.to_device(b14)
        ctx.add_avg_batch(d_a.ptr, d_b.ptr, d_o.ptr, d_jobs.ptr, 1)
        assert np.array_equal(d_o.to_host(), z["avg_%d" % k])


@pytest.mark.parametrize("wpj", [0, 2, 4, 8, 16])
def test_full_search_matches_oracle(ctx, wpj):
    """xSetSearchRange + xPatternSearch around the current vector on the unclipped bi-pred target 2*org - pred."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    rng = np.random.default_rng(61)
    tgt = np.ascontiguousarray((2 * scene.cur.astype(np.int32) - rng.integers(0, 1024, scene.cur.shape)).astype(np.int16))
    n = 500
    jobs = (FullJob * n)()
    exp = []
    for k in range(n):
        w = int(rng.choice([4, 8, 16, 32, 64, 128, 8, 16, 12, 24]))
        h = int(rng.choice([4, 8, 16, 32, 64, 128, 8, 16]))
        x = int(rng.integers(0, (416 - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (240 - h) // 4 + 1)) * 4
        if k % 6 == 0:
            x, y = (0 if k % 12 == 0 else 416 - w), (0 if k % 4 == 0 else 240 - h)
        ch, cv = int(rng.integers(-3000, 3000)), int(rng.integers(-2500, 2500))
        sr = int(rng.choice([4, 4, 4, 2, 7]))
        jd = dict(w=w, h=h, x=x, y=y, subShift=1 if (h > 8 and w <= 64) else 0, lam=float(rng.uniform(1, 40)),
                  predHor=int(rng.integers(-64, 64)), predVer=int(rng.integers(-64, 64)))
        org = np.ascontiguousarray(tgt[y:y + h, x:x + w])
        c = me_util.oracle_ctx(scene, jd, org)
        rg = ol.Range()
        L.vo_set_search_range(C.byref(c), ch, cv, sr, C.byref(rg))
        r = ol.MeResult()
        L.vo_full_search(C.byref(c), C.byref(rg), C.byref(r))
        exp.append((r.mvX, r.mvY, r.cost, r.dist, r.nEval))
        j = jobs[k]
        j.orgOff, j.refOff = y * 416 + x, scene.ref_off + y * scene.ref_stride + x
        j.orgStride, j.refStride, j.puX, j.puY, j.width, j.height = 416, scene.ref_stride, x, y, w, h
        j.subShift, j.imvShift, j.signedSamples = jd["subShift"], 0, 1
        j.predHor, j.predVer, j.motionLambda = jd["predHor"], jd["predVer"], jd["lam"]
        j.centerHor, j.centerVer, j.searchRange = ch, cv, sr
    d_org, d_ref = ctx.to_device(tgt), ctx.to_device(scene.ref_buf)
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_res = ctx.alloc(32 * n)
    ctx.full_search_batch(PicParams(416, 240, 128, 10, wpj), d_org.ptr, d_ref.ptr, d_jobs.ptr, n, d_res.ptr)
    res = (MeResult * n).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    got = [(r.mvX, r.mvY, r.cost, r.dist, r.nEval) for r in res]
    bad = [k for k in range(n) if got[k] != exp[k]]
    assert not bad, [(got[k], exp[k]) for k in bad[:5]]


def test_affine_gradient_matches_reference_golden(ctx):
    """Sobel planes and 4-/6-parameter normal equations vs vectors recorded from AffineGradientSearch (scalar == AVX2)."""
    from vtm_amd.lib import AffineJob
    z = np.load(os.path.join(G, "misc.npz"))
    for k in range(int(z["count"][0])):
        pred, resi = z["pred_%d" % k], z["resi_%d" % k]
        h, w = pred.shape
        for six in (0, 1):
            jobs = (AffineJob * 2)()   # two identical jobs: exercises the batch indexing
            for q in range(2):
                j = jobs[q]
                j.predOff, j.resiOff, j.derivHOff, j.derivVOff = 0, 0, q * 2 * w * h, q * 2 * w * h + w * h
                j.predStride, j.resiStride, j.derivStride, j.width, j.height, j.sixParam = w, w, w, w, h, six
            d_pred, d_resi = ctx.to_device(pred), ctx.to_device(resi)
            d_der = ctx.to_device(np.zeros(4 * w * h, np.int32))
            d_eq = ctx.to_device(np.zeros((2, 7, 7), np.int64))
            d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
            ctx.affine_sobel_batch(d_pred.ptr, d_der.ptr, d_jobs.ptr, 2)
            ctx.affine_equal_coeff_batch(d_resi.ptr, d_der.ptr, d_jobs.ptr, 2, d_eq.ptr)
            der = d_der.to_host().reshape(4, h, w)
            eq = d_eq.to_host().reshape(2, 7, 7)
            for q in range(2):
                assert np.array_equal(der[2 * q], z["gx_%d" % k]) and np.array_equal(der[2 * q + 1], z["gy_%d" % k]), (k, six, q)
                assert np.array_equal(eq[q], z["eq_%d_%d" % (k, six)]), (k, six, q)


def test_motion_compensation_fused_matches_oracle(ctx):
    """vtmhip_motion_compensation_batch_dev: uni / bi (two 14-bit predictions + addAvg) for luma and 4:2:0 chroma blocks, prediction
    output plus the fused residual (org - pred) or removeHighFreq (2*org - pred) epilogue, against the oracle composition
    vo_mc_block -> vo_add_avg -> subtract."""
    from vtm_amd import synth
    from vtm_amd.lib import PredJob
    L = ol.oracle()
    W, H, m = 416, 240, 64
    fr = synth.gen_frames(W, H, 3, chroma=True)
    (yb0, yo, ys), (ub0, uo, us) = synth.extend_plane(fr[0][0], m), synth.extend_plane(fr[0][1], m // 2)
    (yb1, _, _), (ub1, _, _) = synth.extend_plane(fr[1][0], m), synth.extend_plane(fr[1][1], m // 2)
    org_y, org_u = np.ascontiguousarray(fr[2][0]), np.ascontiguousarray(fr[2][1])
    # device layout: [Y ref0 | Y ref1 | U ref0 | U ref1], originals [Y | U]
    refs = np.concatenate([yb0, yb1, ub0, ub1])
    base = {(0, 0): 0, (0, 1): yb0.size, (1, 0): 2 * yb0.size, (1, 1): 2 * yb0.size + ub0.size}
    planes = {(0, 0): yb0, (0, 1): yb1, (1, 0): ub0, (1, 1): ub1}
    orgs = np.concatenate([org_y.reshape(-1), org_u.reshape(-1)])
    rng = np.random.default_rng(91)
    n = 500
    jobs = (PredJob * n)()
    exp_pred, exp_out, meta = [], [], []
    pos = 0
    for k in range(n):
        w, h = int(rng.choice([4, 8, 16, 32, 64, 128])), int(rng.choice([4, 8, 16, 32, 64, 128]))
        chroma, mode, epi, alt = int(k % 3 == 1), int(k % 4 if k % 4 < 3 else 2), int(k % 3), int(k % 7 == 0)
        if w == 4 and h == 4 and not chroma:
            w = 8
        x, y = int(rng.integers(0, (W - w) // 4 + 1)) * 4, int(rng.integers(0, (H - h) // 4 + 1)) * 4      # (drawn after the size is final: the block stays inside the picture)
        cw, ch, cx, cy = (w // 2, h // 2, x // 2, y // 2) if chroma else (w, h, x, y)
        st, o0 = (us, uo) if chroma else (ys, yo)
        mv = [[int(rng.integers(-20 * 16, 20 * 16)), int(rng.integers(-20 * 16, 20 * 16))] for _ in range(2)]
        if k % 5 == 0:
            mv[0][0] &= ~15
        if k % 9 == 0:
            mv[1][1] &= ~31
        j = jobs[k]
        j.orgOff = (org_y.size + cy * (W // 2) + cx) if chroma else (cy * W + cx)
        j.orgStride = W // 2 if chroma else W
        for l in (0, 1):
            j.refOff[l] = base[(chroma, l)] + o0 + cy * st + cx
            j.refStride[l] = st
            j.mv[l][0], j.mv[l][1] = mv[l]
        j.predOff = j.outOff = pos
        j.predStride = j.outStride = cw
        j.width, j.height, j.mode, j.epilogue, j.bitDepth, j.useAltHpelIf, j.chroma = cw, ch, mode, epi, 10, alt, chroma
        p = [np.zeros((ch, cw), np.int16), np.zeros((ch, cw), np.int16)]
        for l in ((0, 1) if mode == 2 else (mode,)):
            pl = planes[(chroma, l)]
            refp = pl.ctypes.data + 2 * (o0 + cy * st + cx)
            L.vo_mc_block(1 if chroma else 0, C.c_void_p(refp), st, cw, ch, mv[l][0], mv[l][1], int(mode == 2), 10, alt, ol.P(p[l]), cw)
        if mode == 2:
            pred = np.zeros((ch, cw), np.int16)
            L.vo_add_avg(ol.P(p[0]), cw, ol.P(p[1]), cw, ol.P(pred), cw, cw, ch, 10)
        else:
            pred = p[mode]
        ob = (org_u if chroma else org_y)[cy:cy + ch, cx:cx + cw].astype(np.int32)
        out = (ob - pred) if epi == 1 else (2 * ob - pred) if epi == 2 else np.zeros_like(ob)
        exp_pred.append(pred.reshape(-1))
        exp_out.append(out.astype(np.int16).reshape(-1))
        meta.append((cw, ch, chroma, mode, epi))
        pos += cw * ch
    d_org, d_ref = ctx.to_device(orgs), ctx.to_device(refs)
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_pred, d_out = ctx.to_device(np.zeros(pos, np.int16)), ctx.to_device(np.zeros(pos, np.int16))
    ctx.motion_compensation_batch(d_org.ptr, d_ref.ptr, d_pred.ptr, d_out.ptr, d_jobs.ptr, n, 128, 128)
    gp, go = d_pred.to_host(np.int16), d_out.to_host(np.int16)
    at = 0
    for k, (cw, ch, chroma, mode, epi) in enumerate(meta):
        sl = slice(at, at + cw * ch)
        assert np.array_equal(gp[sl], exp_pred[k]), ("pred", k, meta[k])
        assert np.array_equal(go[sl], exp_out[k]), ("out", k, meta[k])
        at += cw * ch
    # prediction-only and epilogue-only calls
    d_out2 = ctx.to_device(np.zeros(pos, np.int16))
    ctx.motion_compensation_batch(d_org.ptr, d_ref.ptr, None, d_out2.ptr, d_jobs.ptr, n, 128, 128)
    assert np.array_equal(d_out2.to_host(np.int16), go)


@pytest.mark.parametrize("size", [8, 16, 32, 64])
def test_full_search_square_kernel_matches_oracle(ctx, size):
    """vtmhip_full_search_square_batch_dev (one lane per candidate over an LDS window): uniform S x S jobs, +-4 and smaller / clipped
    ranges, picture-corner positions, signed bi-pred targets and plain pictures."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    rng = np.random.default_rng(62 + size)
    tgt = np.ascontiguousarray((2 * scene.cur.astype(np.int32) - rng.integers(0, 1024, scene.cur.shape)).astype(np.int16))
    for signed, plane in ((1, tgt), (0, scene.cur)):
        n = 700 if size <= 16 else 250
        jobs = (FullJob * n)()
        exp = []
        for k in range(n):
            w = h = size
            x = int(rng.integers(0, (416 - w) // 4 + 1)) * 4
            y = int(rng.integers(0, (240 - h) // 4 + 1)) * 4
            if k % 6 == 0:
                x, y = (0 if k % 12 == 0 else 416 - w), (0 if k % 4 == 0 else 240 - h)
            ch, cv = int(rng.integers(-3000, 3000)), int(rng.integers(-2500, 2500))
            if k % 9 == 0:
                ch, cv = -5000, 4000      # far outside: the clipped window collapses
            sr = int(rng.choice([4, 4, 4, 2, 1, 0, 3]))
            jd = dict(w=w, h=h, x=x, y=y, subShift=1 if (h > 8 and w <= 64) else 0, lam=float(rng.uniform(1, 40)),
                      predHor=int(rng.integers(-64, 64)), predVer=int(rng.integers(-64, 64)))
            org = np.ascontiguousarray(plane[y:y + h, x:x + w])
            c = me_util.oracle_ctx(scene, jd, org)
            rg = ol.Range()
            L.vo_set_search_range(C.byref(c), ch, cv, sr, C.byref(rg))
            r = ol.MeResult()
            L.vo_full_search(C.byref(c), C.byref(rg), C.byref(r))
            exp.append((r.mvX, r.mvY, r.cost, r.dist, r.nEval))
            j = jobs[k]
            j.orgOff, j.refOff = y * 416 + x, scene.ref_off + y * scene.ref_stride + x
            j.orgStride, j.refStride, j.puX, j.puY, j.width, j.height = 416, scene.ref_stride, x, y, w, h
            j.subShift, j.imvShift, j.signedSamples = jd["subShift"], 0, signed
            j.predHor, j.predVer, j.motionLambda = jd["predHor"], jd["predVer"], jd["lam"]
            j.centerHor, j.centerVer, j.searchRange = ch, cv, sr
        d_org, d_ref = ctx.to_device(plane), ctx.to_device(scene.ref_buf)
        d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
        d_res = ctx.alloc(32 * n)
        ctx.full_search_batch(PicParams(416, 240, 128, 10, 0), d_org.ptr, d_ref.ptr, d_jobs.ptr, n, d_res.ptr, square=size)
        res = (MeResult * n).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
        got = [(r.mvX, r.mvY, r.cost, r.dist, r.nEval) for r in res]
        bad = [k for k in range(n) if got[k] != exp[k]]
        assert not bad, (signed, [(got[k], exp[k]) for k in bad[:5]])


def test_geo_blend_matches_oracle(ctx):
    """vtmhip_weightedGeoBlk / _batch_dev vs the oracle on synthetic weight planes: all four walk directions, chroma step 2, odd widths and unaligned
    strides (scalar path of the kernel), 8- and 10-bit, a narrowed clip range."""
    from vtm_amd.lib import GeoBlendJob
    L = ol.oracle()
    rng = np.random.default_rng(911)
    M = 224
    plane = ol.i16(rng.integers(0, 9, (M, M))).reshape(-1)
    for bd, clip in ((10, (0, 1023)), (8, (0, 255)), (10, (64, 940))):
        n = 200
        jobs = (GeoBlendJob * n)()
        srcs, exp, pos, spos = [], [], 0, 0
        for k in range(n):
            w = int(rng.choice([4, 8, 16, 32, 64, 6, 2])); h = int(rng.choice([4, 8, 16, 32, 64]))
            pad0, pad1 = (0, 0) if k % 3 else (int(rng.integers(0, 5)), int(rng.integers(0, 5)))
            s0 = ol.i16(rng.integers(-8192, 8192 + ((1 << bd) - 1) * (1 << (14 - bd)), (h, w + pad0)))
            s1 = ol.i16(rng.integers(-8192, 8192 + ((1 << bd) - 1) * (1 << (14 - bd)), (h, w + pad1)))
            sx = int(rng.choice([1, -1, 2, -2])); rd = int(rng.choice([1, -1])) * (2 if abs(sx) == 2 else 1)
            x0 = (int(rng.integers(0, M - abs(sx) * w)) + (abs(sx) * (w - 1) if sx < 0 else 0))
            y0 = (int(rng.integers(0, M - abs(rd) * h)) + (abs(rd) * (h - 1) if rd < 0 else 0))
            off, ws = y0 * M + x0, rd * M
            e = np.zeros((h, w), np.int16)
            L.vo_weighted_geo_blk(ol.P(s0), w + pad0, ol.P(s1), w + pad1, ol.P(e), w, w, h, C.c_void_p(plane.ctypes.data + 2 * off), sx, ws, bd, clip[0], clip[1])
            exp.append(e.reshape(-1))
            if k < 40:
                got = ctx.weightedGeoBlk(s0, s1, w, h, plane, off, sx, ws, bd, clip)
                assert np.array_equal(got, e), (k, w, h, sx, rd, bd)
            j = jobs[k]
            j.src0Off, j.src0Stride = spos, w + pad0
            spos += s0.size
            j.src1Off, j.src1Stride = spos, w + pad1
            spos += s1.size
            j.dstOff, j.dstStride, j.weightOff, j.weightStride = pos, w, off, ws
            j.width, j.height, j.stepX = w, h, sx
            srcs += [s0.reshape(-1), s1.reshape(-1)]
            pos += w * h
        d_src, d_w = ctx.to_device(np.concatenate(srcs)), ctx.to_device(plane)
        d_jobs, d_dst = ctx.to_device(np.frombuffer(jobs, np.uint8)), ctx.alloc(2 * pos)
        ctx.weightedGeoBlk_batch(d_src.ptr, d_dst.ptr, d_w.ptr, d_jobs.ptr, n, bd, clip)
        assert np.array_equal(d_dst.to_host(np.int16), np.concatenate(exp)), bd


def test_bdof_matches_oracle(ctx):
    """vtmhip_bdof_batch_dev vs vo_bdof_pu: every PU size class the reference enables BDOF for, integer / half / mixed phases on either list,
    8- and 10-bit, prediction and both fused epilogues."""
    from vtm_amd import synth
    from vtm_amd.lib import PredJob
    L = ol.oracle()
    W, H, M = 256, 192, 48
    fr = list(synth.gen_frames(W, H, 3, seed=9))
    rng = np.random.default_rng(1012)
    for bd in (10, 8):
        planes = [np.ascontiguousarray(np.pad((f >> (10 - bd)).astype(np.int16), M, mode="edge")) for f in (fr[0], fr[2])]
        org = np.ascontiguousarray((fr[1] >> (10 - bd)).astype(np.int16))
        S, plane_sz = planes[0].shape[1], planes[0].size
        n = 150
        jobs = (PredJob * n)()
        exp_pred, exp_out, pos = [], [], 0
        for k in range(n):
            w, h = int(rng.choice([8, 16, 32, 64, 128])), int(rng.choice([8, 16, 32, 64, 128]))
            if w * h < 128:
                w = 16
            x, y = int(rng.integers(0, (W - w) // 4 + 1)) * 4, int(rng.integers(0, (H - h) // 4 + 1)) * 4
            mv = [int(v) for v in rng.integers(-500, 500, 4)]
            if k % 5 == 0:
                mv[k % 4] &= ~15
            if k % 13 == 0:
                mv = [v & ~15 for v in mv]
            if k % 17 == 0:
                mv = [(v & ~15) | 8 for v in mv]
            e = np.zeros((h, w), np.int16)
            at = [C.c_void_p(p.ctypes.data + 2 * ((y + M) * S + x + M)) for p in planes]
            L.vo_bdof_pu(at[0], S, at[1], S, w, h, *mv, bd, ol.P(e), w)
            j = jobs[k]
            for l in range(2):
                j.refOff[l], j.refStride[l] = l * plane_sz + (M + y) * S + M + x, S
            j.mv[0][0], j.mv[0][1], j.mv[1][0], j.mv[1][1] = mv
            j.orgOff, j.orgStride = y * W + x, W
            j.predOff = j.outOff = pos
            j.predStride = j.outStride = w
            j.width, j.height, j.mode, j.bitDepth, j.epilogue = w, h, 2, bd, 1 + k % 2
            o = org[y:y + h, x:x + w].astype(np.int32)
            exp_pred.append(e.reshape(-1))
            exp_out.append(((o if j.epilogue == 1 else 2 * o) - e).astype(np.int16).reshape(-1))
            pos += w * h
        d_ref = ctx.to_device(np.concatenate([p.reshape(-1) for p in planes]))
        d_org, d_jobs = ctx.to_device(org.reshape(-1)), ctx.to_device(np.frombuffer(jobs, np.uint8))
        d_pred, d_out = ctx.alloc(2 * pos), ctx.alloc(2 * pos)
        ctx.bdof_batch(d_org.ptr, d_ref.ptr, d_pred.ptr, d_out.ptr, d_jobs.ptr, n, 128, 128)
        assert np.array_equal(d_pred.to_host(np.int16), np.concatenate(exp_pred)), bd
        assert np.array_equal(d_out.to_host(np.int16), np.concatenate(exp_out)), bd
        d_pred2 = ctx.alloc(2 * pos)
        ctx.bdof_batch(0, d_ref.ptr, d_pred2.ptr, 0, d_jobs.ptr, n, 128, 128)   # prediction only
        assert np.array_equal(d_pred2.to_host(np.int16), np.concatenate(exp_pred)), bd


def test_dmvr_matches_oracle(ctx):
    """vtmhip_dmvr_batch_dev vs vo_dmvr_pu: all PU size classes, with / without BDOF, vectors around the clip's motion, integer phases, far
    out-of-picture vectors (clipMv), 8- and 10-bit, prediction + fused epilogues + vector differences."""
    from vtm_amd import synth
    from vtm_amd.lib import DmvrJob, PicParams
    L = ol.oracle()
    W, H, M = 256, 192, 160
    fr = list(synth.gen_frames(W, H, 3, seed=9))
    rng = np.random.default_rng(1015)
    for bd in (10, 8):
        planes = [np.ascontiguousarray(np.pad((f >> (10 - bd)).astype(np.int16), M, mode="edge")) for f in (fr[0], fr[2])]
        org = np.ascontiguousarray((fr[1] >> (10 - bd)).astype(np.int16))
        S, plane_sz = planes[0].shape[1], planes[0].size
        o = [C.c_void_p(p.ctypes.data + 2 * (M * S + M)) for p in planes]
        n, regions = 160, 64
        jobs = (DmvrJob * n)()
        exp_pred, exp_out, exp_mvd, pos = [], [], np.zeros((n, regions, 2), np.int32), 0
        for k in range(n):
            w, h = int(rng.choice([8, 16, 32, 64, 128])), int(rng.choice([8, 16, 32, 64, 128]))
            if w * h < 128:
                w = 16
            x, y = int(rng.integers(0, (W - w) // 4 + 1)) * 4, int(rng.integers(0, (H - h) // 4 + 1)) * 4
            base = np.array([48, 32]) + rng.integers(-40, 41, 2)
            mv = [int(-base[0]), int(-base[1]), int(base[0] + rng.integers(-24, 25)), int(base[1] + rng.integers(-24, 25))]
            if k % 9 == 0:
                mv = [int(v) for v in rng.integers(-4000, 4000, 4)]
            if k % 7 == 0:
                mv[k % 4] &= ~15
            if k % 11 == 0:
                mv = [v & ~15 for v in mv]
            bio = k % 2
            nsub = (w // min(w, 16)) * (h // min(h, 16))
            e, mvd = np.zeros((h, w), np.int16), np.zeros(2 * nsub, np.int32)
            L.vo_dmvr_pu(o[0], o[1], S, W, H, 128, x, y, w, h, *mv, bd, bio, ol.P(e), w, C.c_void_p(mvd.ctypes.data))
            exp_mvd[k, :nsub] = mvd.reshape(-1, 2)
            j = jobs[k]
            for l in range(2):
                j.refOff[l], j.refStride[l] = l * plane_sz + (M + y) * S + M + x, S
            j.mv[0][0], j.mv[0][1], j.mv[1][0], j.mv[1][1] = mv
            j.orgOff, j.orgStride, j.puX, j.puY = y * W + x, W, x, y
            j.predOff = j.outOff = pos
            j.predStride = j.outStride = w
            j.width, j.height, j.bitDepth, j.bioApplied, j.epilogue = w, h, bd, bio, 1 + k % 2
            ob = org[y:y + h, x:x + w].astype(np.int32)
            exp_pred.append(e.reshape(-1))
            exp_out.append(((ob if j.epilogue == 1 else 2 * ob) - e).astype(np.int16).reshape(-1))
            pos += w * h
        assert np.count_nonzero(exp_mvd) > 300
        pic = PicParams(W, H, 128, bd, 0)
        d_ref = ctx.to_device(np.concatenate([p.reshape(-1) for p in planes]))
        d_org, d_jobs = ctx.to_device(org.reshape(-1)), ctx.to_device(np.frombuffer(jobs, np.uint8))
        d_pred, d_out, d_mvd = ctx.alloc(2 * pos), ctx.alloc(2 * pos), ctx.alloc(4 * exp_mvd.size)
        ctx.dmvr_batch(pic, d_org.ptr, d_ref.ptr, d_pred.ptr, d_out.ptr, d_jobs.ptr, n, 128, 128, d_mvd.ptr)
        got_mvd = d_mvd.to_host(np.int32).reshape(n, regions, 2)
        for k in range(n):
            nsub = (jobs[k].width // min(jobs[k].width, 16)) * (jobs[k].height // min(jobs[k].height, 16))
            assert np.array_equal(got_mvd[k, :nsub], exp_mvd[k, :nsub]), (bd, k)
        assert np.array_equal(d_pred.to_host(np.int16), np.concatenate(exp_pred)), bd
        assert np.array_equal(d_out.to_host(np.int16), np.concatenate(exp_out)), bd


def test_dmvr_chroma_matches_oracle(ctx):
    """vtmhip_dmvr_chroma_batch_dev vs vo_dmvr_chroma on vector differences taken from vo_dmvr_pu: moved and unmoved sub-PUs, clipped vectors, 8- and
    10-bit, both fused epilogues."""
    from vtm_amd import synth
    from vtm_amd.lib import DmvrJob, PicParams
    L = ol.oracle()
    W, H, M = 256, 192, 160
    fr = list(synth.gen_frames(W, H, 3, seed=9, chroma=True))
    rng = np.random.default_rng(1017)
    for bd in (10, 8):
        P = [[np.ascontiguousarray(np.pad((f[c] >> (10 - bd)).astype(np.int16), M if c == 0 else M // 2, mode="edge")) for c in range(3)] for f in (fr[0], fr[2])]
        orgc = np.ascontiguousarray((fr[1][1] >> (10 - bd)).astype(np.int16))
        SY, SC, csz = P[0][0].shape[1], P[0][1].shape[1], P[0][1].size
        oy = [C.c_void_p(P[l][0].ctypes.data + 2 * (M * SY + M)) for l in range(2)]
        n, regions = 120, 64
        jobs = (DmvrJob * n)()
        mvd_all = np.zeros((n, regions, 2), np.int32)
        exp_pred, exp_out, pos = [], [], 0
        for k in range(n):
            w, h = int(rng.choice([8, 16, 32, 64, 128])), int(rng.choice([8, 16, 32, 64, 128]))
            if w * h < 128:
                w = 16
            x, y = int(rng.integers(0, (W - w) // 8 + 1)) * 8, int(rng.integers(0, (H - h) // 8 + 1)) * 8
            base = np.array([48, 32]) + rng.integers(-40, 41, 2)
            mv = [int(-base[0]), int(-base[1]), int(base[0] + rng.integers(-24, 25)), int(base[1] + rng.integers(-24, 25))]
            if k % 9 == 0:
                mv = [int(v) for v in rng.integers(-4000, 4000, 4)]
            comp = 1 + k % 2
            nsub = (w // min(w, 16)) * (h // min(h, 16))
            lum, mvd = np.zeros((h, w), np.int16), np.zeros(2 * nsub, np.int32)
            L.vo_dmvr_pu(oy[0], oy[1], SY, W, H, 128, x, y, w, h, *mv, bd, 0, ol.P(lum), w, C.c_void_p(mvd.ctypes.data))
            mvd_all[k, :nsub] = mvd.reshape(-1, 2)
            oc = [C.c_void_p(P[l][comp].ctypes.data + 2 * ((M // 2) * SC + M // 2)) for l in range(2)]
            e = np.zeros((h // 2, w // 2), np.int16)
            L.vo_dmvr_chroma(oc[0], oc[1], SC, W, H, 128, x, y, w, h, *mv, C.c_void_p(mvd.ctypes.data), bd, ol.P(e), w // 2)
            j = jobs[k]
            for l in range(2):
                j.refOff[l], j.refStride[l] = (2 * l + comp - 1) * csz + (M // 2 + y // 2) * SC + M // 2 + x // 2, SC
            j.mv[0][0], j.mv[0][1], j.mv[1][0], j.mv[1][1] = mv
            j.orgOff, j.orgStride, j.puX, j.puY, j.mvdRow = (y // 2) * (W // 2) + x // 2, W // 2, x, y, k
            j.predOff = j.outOff = pos
            j.predStride = j.outStride = w // 2
            j.width, j.height, j.bitDepth, j.epilogue = w, h, bd, 1 + (k // 2) % 2
            ob = orgc[y // 2:(y + h) // 2, x // 2:(x + w) // 2].astype(np.int32)
            exp_pred.append(e.reshape(-1))
            exp_out.append(((ob if j.epilogue == 1 else 2 * ob) - e).astype(np.int16).reshape(-1))
            pos += w * h // 4
        pic = PicParams(W, H, 128, bd, 0)
        d_ref = ctx.to_device(np.concatenate([P[l][c].reshape(-1) for l in range(2) for c in (1, 2)]))
        d_org, d_jobs, d_mvd = ctx.to_device(orgc.reshape(-1)), ctx.to_device(np.frombuffer(jobs, np.uint8)), ctx.to_device(mvd_all.reshape(-1))
        d_pred, d_out = ctx.alloc(2 * pos), ctx.alloc(2 * pos)
        ctx.dmvr_chroma_batch(pic, d_org.ptr, d_ref.ptr, d_pred.ptr, d_out.ptr, d_jobs.ptr, n, 128, 128, d_mvd.ptr)
        assert np.array_equal(d_pred.to_host(np.int16), np.concatenate(exp_pred)), bd
        assert np.array_equal(d_out.to_host(np.int16), np.concatenate(exp_out)), bd


def test_prediction_entry_points_empty_and_invalid(ctx):
    """Error behaviour of the GEO / BDOF / DMVR entry points: an empty batch is a no-op, wrong arguments return VTMHIP_E_INVALID (a C status, never a launch)."""
    from vtm_amd.lib import PicParams, VtmHipError
    buf = ctx.alloc(4096)
    pic = PicParams(64, 64, 128, 10, 0)
    ctx.bdof_batch(0, buf.ptr, buf.ptr, 0, buf.ptr, 0, 16, 16)                       # n == 0
    ctx.dmvr_batch(pic, 0, buf.ptr, buf.ptr, 0, buf.ptr, 0, 16, 16, buf.ptr)
    ctx.dmvr_chroma_batch(pic, 0, buf.ptr, buf.ptr, 0, buf.ptr, 0, 16, 16, buf.ptr)
    ctx.weightedGeoBlk_batch(buf.ptr, buf.ptr, buf.ptr, buf.ptr, 0)
    for bad in (lambda: ctx.bdof_batch(0, buf.ptr, buf.ptr, 0, buf.ptr, 1, 4, 16),      # BDOF needs w >= 8
                lambda: ctx.bdof_batch(0, buf.ptr, buf.ptr, buf.ptr, buf.ptr, 1, 16, 16),   # epilogue output without the original plane
                lambda: ctx.bdof_batch(0, 0, buf.ptr, 0, buf.ptr, 1, 16, 16),           # null reference plane
                lambda: ctx.dmvr_batch(PicParams(64, 64, 128, 14, 0), 0, buf.ptr, buf.ptr, 0, buf.ptr, 1, 16, 16, 0),   # bit depth
                lambda: ctx.dmvr_batch(pic, 0, buf.ptr, 0, 0, buf.ptr, 1, 16, 16, 0),   # nothing to write
                lambda: ctx.dmvr_chroma_batch(pic, 0, buf.ptr, buf.ptr, 0, buf.ptr, 1, 16, 16, 0),   # chroma needs the vector differences
                lambda: ctx.weightedGeoBlk_batch(buf.ptr, buf.ptr, buf.ptr, buf.ptr, 1, 10, (0, 2000)),   # clip range beyond the bit depth
                lambda: ctx.weightedGeoBlk(np.zeros((8, 8), np.int16), np.zeros((8, 8), np.int16), 8, 8, np.zeros(64, np.int16), 0, 3, 8)):   # stepX
        with pytest.raises(VtmHipError):
            bad()


def test_bcw_ops_match_oracle(ctx):
    """vtmhip_remove_weight_high_freq_batch_dev / vtmhip_add_weighted_avg_batch_dev vs the oracle: the five BCW weights (and their list-0 complements)."""
    L = ol.oracle()
    rng = np.random.default_rng(1019)
    n = 120
    jobs = (PelOpJob * n)()
    A, B, E0, E1, pos = [], [], [], [], 0
    for k in range(n):
        w, h = int(rng.choice([4, 8, 16, 32, 64, 128])), int(rng.choice([4, 8, 16, 32, 64]))
        bw = [-2, 3, 4, 5, 10][k % 5]
        lw = bw if k % 2 else 8 - bw
        bd = 8 if k % 7 == 0 else 10
        hi = 8192 + ((1 << bd) - 1) * (1 << (14 - bd))
        a, b = ol.i16(rng.integers(0, 1 << bd, (h, w))), ol.i16(rng.integers(0, 1 << bd, (h, w)))
        e0 = a.copy()
        L.vo_remove_weight_high_freq(ol.P(e0), w, ol.P(b), w, w, h, lw)
        a14, b14 = ol.i16(rng.integers(-8192, hi, (h, w))), ol.i16(rng.integers(-8192, hi, (h, w)))
        e1 = np.zeros((h, w), np.int16)
        L.vo_add_weighted_avg(ol.P(a14), w, ol.P(b14), w, ol.P(e1), w, w, h, bd, lw)
        j = jobs[k]
        j.aOff = j.bOff = j.dstOff = pos
        j.aStride = j.bStride = j.dstStride = w
        j.width, j.height, j.bitDepth, j.bcwWeight = w, h, bd, lw
        A.append(np.concatenate([a.reshape(-1), a14.reshape(-1)])); B.append(np.concatenate([b.reshape(-1), b14.reshape(-1)]))
        E0.append(e0.reshape(-1)); E1.append(e1.reshape(-1))
        pos += w * h
    a0 = np.concatenate([x[:x.size // 2] for x in A]); a1 = np.concatenate([x[x.size // 2:] for x in A])
    b0 = np.concatenate([x[:x.size // 2] for x in B]); b1 = np.concatenate([x[x.size // 2:] for x in B])
    d_jobs, d_dst = ctx.to_device(np.frombuffer(jobs, np.uint8)), ctx.alloc(2 * pos)
    ctx.remove_weight_high_freq_batch(ctx.to_device(a0).ptr, ctx.to_device(b0).ptr, d_dst.ptr, d_jobs.ptr, n)
    assert np.array_equal(d_dst.to_host(np.int16), np.concatenate(E0))
    ctx.add_weighted_avg_batch(ctx.to_device(a1).ptr, ctx.to_device(b1).ptr, d_dst.ptr, d_jobs.ptr, n)
    assert np.array_equal(d_dst.to_host(np.int16), np.concatenate(E1))


def test_merge_candidate_satd_batch(ctx):
    """Hook B10 (EncCu::xCheckRDCostMerge2Nx2N :2399-2440): one call = the luma prediction of every merge candidate (plain uni / bi, BDOF, DMVR (+ BDOF) by
    the candidate's motion) + the Hadamard distortion against the original block; predictions stay in the buffer (acMergeBuffer).  vs the oracle."""
    from vtm_amd import synth
    from vtm_amd.lib import DmvrJob, PicParams, PredJob
    L = ol.oracle()
    W, H, M, bd = 256, 192, 160, 10
    fr = list(synth.gen_frames(W, H, 3, seed=9))
    rng = np.random.default_rng(2024)
    planes = [np.ascontiguousarray(np.pad(f.astype(np.int16), M, mode="edge")) for f in (fr[0], fr[2])]
    org = np.ascontiguousarray(fr[1].astype(np.int16))
    S, plane_sz = planes[0].shape[1], planes[0].size
    o = [C.c_void_p(p.ctypes.data + 2 * (M * S + M)) for p in planes]
    for (w, h) in ((16, 16), (32, 16), (8, 32), (64, 64)):
        n_each = 20
        plain, bdof, dmvr = (PredJob * n_each)(), (PredJob * n_each)(), (DmvrJob * n_each)()
        exp, preds, pos = [], [], 0
        for kind, tab in ((0, plain), (1, bdof), (2, dmvr)):
            for k in range(n_each):
                x, y = int(rng.integers(0, (W - w) // 4 + 1)) * 4, int(rng.integers(0, (H - h) // 4 + 1)) * 4
                base = np.array([48, 32]) + rng.integers(-40, 41, 2)
                mv = [int(-base[0]), int(-base[1]), int(base[0] + rng.integers(-24, 25)), int(base[1] + rng.integers(-24, 25))]
                e = np.zeros((h, w), np.int16)
                at = [C.c_void_p(p.ctypes.data + 2 * ((y + M) * S + x + M)) for p in planes]
                j = tab[k]
                mode = 2
                if kind == 0:
                    mode = k % 3                      # list 0, list 1, bi (addAvg)
                    if mode == 2:
                        p0, p1 = np.zeros((h, w), np.int16), np.zeros((h, w), np.int16)
                        L.vo_mc_luma(at[0], S, w, h, mv[0], mv[1], 1, bd, 0, ol.P(p0), w)
                        L.vo_mc_luma(at[1], S, w, h, mv[2], mv[3], 1, bd, 0, ol.P(p1), w)
                        L.vo_add_avg(ol.P(p0), w, ol.P(p1), w, ol.P(e), w, w, h, bd)
                    else:
                        L.vo_mc_luma(at[mode], S, w, h, mv[2 * mode], mv[2 * mode + 1], 0, bd, 0, ol.P(e), w)
                elif kind == 1:
                    L.vo_bdof_pu(at[0], S, at[1], S, w, h, *mv, bd, ol.P(e), w)
                else:
                    mvd = np.zeros(2 * 64, np.int32)
                    L.vo_dmvr_pu(o[0], o[1], S, W, H, 128, x, y, w, h, *mv, bd, k % 2, ol.P(e), w, C.c_void_p(mvd.ctypes.data))
                    j.puX, j.puY, j.bioApplied = x, y, k % 2
                for l in range(2):
                    j.refOff[l], j.refStride[l] = l * plane_sz + (M + y) * S + M + x, S
                j.mv[0][0], j.mv[0][1], j.mv[1][0], j.mv[1][1] = mv
                j.orgOff, j.orgStride, j.predOff, j.predStride = y * W + x, W, pos, w
                j.width, j.height, j.bitDepth, j.epilogue = w, h, bd, 0
                if kind < 2:
                    j.mode = mode
                ob = np.ascontiguousarray(org[y:y + h, x:x + w])
                exp.append(ol.o_dist(1, ob, e, w, h))
                preds.append(e.reshape(-1))
                pos += w * h
        pic = PicParams(W, H, 128, bd, 0)
        d_ref = ctx.to_device(np.concatenate([p.reshape(-1) for p in planes]))
        d_org = ctx.to_device(org.reshape(-1))
        d_p, d_b, d_d = (ctx.to_device(np.frombuffer(t, np.uint8)) for t in (plain, bdof, dmvr))
        d_pred, d_dist, d_mvd = ctx.alloc(2 * pos), ctx.alloc(8 * 3 * n_each), ctx.alloc(4 * n_each * 64 * 2)
        for uniform in (False, True):
            ctx.merge_cand_satd_batch(pic, d_org.ptr, d_ref.ptr, d_pred.ptr, d_p.ptr, n_each, d_b.ptr, n_each, d_d.ptr, n_each, d_mvd.ptr, w, h, d_dist.ptr, uniform=uniform)
            assert d_dist.to_host(np.uint64).tolist() == exp, (w, h, uniform)
            assert np.array_equal(d_pred.to_host(np.int16), np.concatenate(preds)), (w, h, uniform)
