"""GPU: the HIP kernels against the golden vectors recorded from the real reference (tests/golden/*.npz)."""
import json
import os

import numpy as np
import pytest

import me_util
from vtm_amd.lib import FracJob, FracResult, MeResult, PicParams

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_dist_golden(ctx):
    z = np.load(os.path.join(G, "dist.npz"))
    org, cur, st = z["org"], z["cur"], z["starts"]
    for (bi, w, h, kind, ss), e in zip(z["meta"], z["exp"]):
        o = np.ascontiguousarray(org[st[bi]:st[bi + 1]])
        c = np.ascontiguousarray(cur[st[bi]:st[bi + 1]])
        w, h = int(w), int(h)
        got = (ctx.xGetSAD(o, w, c, w, w, h, int(ss)) if kind == 0 else ctx.xGetHADs(o, w, c, w, w, h) if kind == 1 else ctx.xGetSSE(o, w, c, w, w, h))
        assert got == int(e), (w, h, kind, ss)


def test_transform_golden(ctx):
    z = np.load(os.path.join(G, "transform.npz"))
    src, dst = z["src"], z["dst"]
    for (t, n, line, a, b, shift, inv, o0) in z["meta"]:
        s = src[o0:o0 + n * line]
        got = (ctx.fastInvTrans if inv else ctx.fastFwdTrans)(int(t), int(n), s, int(shift), int(line), int(a), int(b))
        assert np.array_equal(got, dst[o0:o0 + n * line]), (t, n, line, a, b, shift, inv)


def test_motion_search_golden(ctx):
    z = np.load(os.path.join(G, "me.npz"))
    scene = me_util.Scene(416, 240, hard=True)
    jobs = [json.loads(s) for s in z["tz_jobs"]]
    arr = me_util.hip_tz_jobs(scene, jobs, scene.W)
    d_cur, d_ref = ctx.to_device(scene.cur), ctx.to_device(scene.ref_buf)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_res = ctx.alloc(len(jobs) * 32)
    ctx.tz_search_batch(PicParams(scene.W, scene.H, 128, 10), d_cur.ptr, d_ref.ptr, d_jobs.ptr, len(jobs), d_res.ptr)
    res = (MeResult * len(jobs)).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    got = np.array([(r.mvX, r.mvY, r.cost, r.dist) for r in res], np.int64)
    assert np.array_equal(got, z["tz_res"])

    fj = z["frac_jobs"]
    farr = (FracJob * len(fj))()
    for k, (w, h, x, y, lam, ph, pv, ix, iy, had) in enumerate(fj):
        t = farr[k]
        t.orgOff, t.refOff = int(y) * scene.W + int(x), scene.ref_off + int(y) * scene.ref_stride + int(x)
        t.orgStride, t.refStride, t.width, t.height = scene.W, scene.ref_stride, int(w), int(h)
        t.intX, t.intY, t.predHor, t.predVer, t.motionLambda = int(ix), int(iy), int(ph), int(pv), float(lam)
        t.useHad, t.useAltHpelIf, t.imvShift, t.bitDepth = int(had), 0, 0, 10
    d_fj = ctx.to_device(np.frombuffer(farr, np.uint8))
    d_fr = ctx.alloc(16 * len(fj))
    ctx.frac_search_batch(d_cur.ptr, d_ref.ptr, d_fj.ptr, len(fj), 128, 128, d_fr.ptr)
    fres = (FracResult * len(fj)).from_buffer_copy(d_fr.to_host(np.uint8).tobytes())
    gotf = np.array([(r.halfX, r.halfY, r.qterX, r.qterY, r.cost) for r in fres], np.int64)
    assert np.array_equal(gotf, z["frac_res"])


def test_quant_golden(ctx):
    """Quant::quant / dequant kernels vs vectors recorded from the real Quant (incl. 64-wide blocks: only the 32x32 region is scanned)."""
    from vtm_amd.lib import QuantJob
    z = np.load(os.path.join(G, "quant.npz"))
    meta = z["meta"]
    n = len(meta)
    jobs = (QuantJob * n)()
    for k, (w, h, qp, irap, o0, asum) in enumerate(meta):
        bq = int(qp) + 12
        j = jobs[k]
        j.srcOff, j.dstOff, j.width, j.height, j.qpPer, j.qpRem, j.bitDepth, j.isIRAP = int(o0), int(o0), int(w), int(h), bq // 6, bq % 6, 10, int(irap)
    d_c = ctx.to_device(z["coef"])
    d_q, d_dq = ctx.to_device(np.zeros_like(z["q"])), ctx.to_device(np.zeros_like(z["dq"]))
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_sum = ctx.alloc(4 * n, np.int32)
    ctx.quant_batch(d_c.ptr, d_q.ptr, None, d_jobs.ptr, n, d_sum.ptr)
    ctx.dequant_batch(d_q.ptr, d_dq.ptr, d_jobs.ptr, n)
    assert np.array_equal(d_q.to_host(), z["q"]) and np.array_equal(d_dq.to_host(), z["dq"])
    assert np.array_equal(d_sum.to_host().astype(np.int64), meta[:, 5])


def test_mc_luma_chroma_golden(ctx):
    """vtmhip_mc_batch_dev (luma + 4:2:0 chroma jobs in one batch) vs xPredInterBlk outputs recorded from the real reference."""
    from vtm_amd import synth
    from vtm_amd.lib import McJob
    z = np.load(os.path.join(G, "mc.npz"))
    W, H, m = 416, 240, 64
    y, u, v = synth.gen_frames(W, H, 1, chroma=True)[0]
    yb, yo, ys = synth.extend_plane(y, m)
    ub, uo, us = synth.extend_plane(u, m // 2)
    planes = np.concatenate([yb, ub])          # one device buffer: luma plane, then the chroma plane
    meta = z["meta"].tolist()
    arr = (McJob * len(meta))()
    pos = 0
    for k, (comp, x, yy, w, h, mvh, mvv, bi, imv) in enumerate(meta):
        cw, ch = (w // 2, h // 2) if comp else (w, h)
        t = arr[k]
        t.refOff = (yb.size + uo + (yy // 2) * us + x // 2) if comp else (yo + yy * ys + x)
        t.refStride = us if comp else ys
        t.dstOff, t.dstStride, t.width, t.height = pos, cw, cw, ch
        t.mvHor, t.mvVer, t.bi, t.bitDepth, t.useAltHpelIf, t.chroma = mvh, mvv, bi, 10, int(imv == 3), int(comp != 0)
        pos += cw * ch
    d_ref = ctx.to_device(planes)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_dst = ctx.alloc(2 * pos)
    ctx.mc_batch(d_ref.ptr, d_dst.ptr, d_jobs.ptr, len(meta), 128, 128)
    got = d_dst.to_host(np.int16)
    assert np.array_equal(got, z["out"])


def test_masked_sad_golden(ctx):
    """vtmhip_xGetSADwMask vs the values recorded from the reference's DF_SAD_WITH_MASK table entry."""
    z = np.load(os.path.join(G, "masked.npz"))
    plane = np.ascontiguousarray(z["plane"]).reshape(-1)
    pos = 0
    for (w, h, off, ms, sx, ms2), exp in zip(z["meta"].tolist(), z["res"].tolist()):
        org = np.ascontiguousarray(z["org"][pos:pos + w * h]).reshape(h, w)
        cur = np.ascontiguousarray(z["cur"][pos:pos + w * h]).reshape(h, w)
        pos += w * h
        assert ctx.xGetSADwMask(org, w, cur, w, w, h, plane, off, ms, sx, ms2) == exp, (w, h, off, ms, sx, ms2)


def test_geo_blend_golden(ctx):
    """vtmhip_weightedGeoBlk (pointer surface) and the batched form vs the blocks recorded from the reference's m_weightedGeoBlk entry."""
    from vtm_amd.lib import GeoBlendJob
    z = np.load(os.path.join(G, "geo.npz"))
    planes = np.ascontiguousarray(z["planes"]).reshape(-1)
    meta = z["meta"].tolist()
    jobs = (GeoBlendJob * len(meta))()
    pos = 0
    for k, (split, comp, w, h, mi, off, sx, ws) in enumerate(meta):
        s0, s1 = (np.ascontiguousarray(z[n][pos:pos + w * h]).reshape(h, w) for n in ("src0", "src1"))
        if k % 4 == 0:
            got = ctx.weightedGeoBlk(s0, s1, w, h, planes, mi * 112 * 112 + off, sx, ws)
            assert np.array_equal(got.reshape(-1), z["out"][pos:pos + w * h]), (split, comp, w, h)
        j = jobs[k]
        j.src0Off, j.src1Off, j.dstOff, j.weightOff = pos, len(z["src0"]) + pos, pos, mi * 112 * 112 + off
        j.src0Stride = j.src1Stride = j.dstStride = w
        j.weightStride, j.width, j.height, j.stepX = ws, w, h, sx
        pos += w * h
    d_src = ctx.to_device(np.concatenate([z["src0"], z["src1"]]))
    d_w, d_jobs, d_dst = ctx.to_device(planes), ctx.to_device(np.frombuffer(jobs, np.uint8)), ctx.alloc(2 * pos)
    ctx.weightedGeoBlk_batch(d_src.ptr, d_dst.ptr, d_w.ptr, d_jobs.ptr, len(meta))
    assert np.array_equal(d_dst.to_host(np.int16), z["out"])


def test_bdof_golden(ctx):
    """vtmhip_bdof_batch_dev vs the BDOF predictions recorded from the reference."""
    from vtm_amd.lib import PredJob
    z = np.load(os.path.join(G, "bdof.npz"))
    planes = np.ascontiguousarray(z["planes"])
    W, H, M = z["dims"].tolist()
    S, plane_sz = planes.shape[2], planes.shape[1] * planes.shape[2]
    meta = z["meta"].tolist()
    jobs = (PredJob * len(meta))()
    pos = 0
    for k, (x, y, w, h, a, b, c, d) in enumerate(meta):
        j = jobs[k]
        for l in range(2):
            j.refOff[l], j.refStride[l] = l * plane_sz + (M + y) * S + M + x, S
        j.mv[0][0], j.mv[0][1], j.mv[1][0], j.mv[1][1] = a, b, c, d
        j.predOff, j.predStride, j.width, j.height, j.mode, j.bitDepth = pos, w, w, h, 2, 10
        pos += w * h
    d_ref, d_jobs, d_pred = ctx.to_device(planes.reshape(-1)), ctx.to_device(np.frombuffer(jobs, np.uint8)), ctx.alloc(2 * pos)
    ctx.bdof_batch(0, d_ref.ptr, d_pred.ptr, 0, d_jobs.ptr, len(meta), 128, 128)
    assert np.array_equal(d_pred.to_host(np.int16), z["out"])


def test_dmvr_golden(ctx):
    """vtmhip_dmvr_batch_dev vs the predictions and pu.mvdL0SubPu recorded from the reference's xProcessDMVR."""
    from vtm_amd.lib import DmvrJob, PicParams
    z, zb = np.load(os.path.join(G, "dmvr.npz")), np.load(os.path.join(G, "bdof.npz"))
    planes = np.ascontiguousarray(zb["planes"])
    W, H, M = z["dims"].tolist()
    S, plane_sz = planes.shape[2], planes.shape[1] * planes.shape[2]
    meta = z["meta"].tolist()
    jobs = (DmvrJob * len(meta))()
    pos = 0
    for k, (x, y, w, h, a, b, c, d, bio) in enumerate(meta):
        j = jobs[k]
        for l in range(2):
            j.refOff[l], j.refStride[l] = l * plane_sz + (M + y) * S + M + x, S
        j.mv[0][0], j.mv[0][1], j.mv[1][0], j.mv[1][1] = a, b, c, d
        j.predOff, j.predStride, j.width, j.height, j.bitDepth, j.bioApplied, j.puX, j.puY = pos, w, w, h, 10, bio, x, y
        pos += w * h
    pic = PicParams(W, H, 128, 10, 0)
    regions = 64
    d_ref, d_jobs, d_pred = ctx.to_device(planes.reshape(-1)), ctx.to_device(np.frombuffer(jobs, np.uint8)), ctx.alloc(2 * pos)
    d_mvd = ctx.alloc(4 * 2 * regions * len(meta))
    ctx.dmvr_batch(pic, 0, d_ref.ptr, d_pred.ptr, 0, d_jobs.ptr, len(meta), 128, 128, d_mvd.ptr)
    mvd = d_mvd.to_host(np.int32).reshape(len(meta), regions, 2)
    mpos = 0
    for k, (x, y, w, h, *_rest) in enumerate(meta):
        nsub = (w // min(w, 16)) * (h // min(h, 16))
        assert np.array_equal(mvd[k, :nsub].reshape(-1), z["mvd"][mpos:mpos + 2 * nsub]), (k, x, y, w, h)
        mpos += 2 * nsub
    assert np.array_equal(d_pred.to_host(np.int16), z["out"])
    # the chroma planes of the same PUs from the device-resident vector differences
    planesC = np.ascontiguousarray(z["planesC"])   # [picture][Cb, Cr][rows][cols]
    SC, csz = planesC.shape[3], planesC.shape[2] * planesC.shape[3]
    d_refc = ctx.to_device(planesC.reshape(-1))
    cjobs = (DmvrJob * (2 * len(meta)))()
    cpos = 0
    for k, (x, y, w, h, a, b, c, d, bio) in enumerate(meta):
        for comp in range(2):
            j = cjobs[2 * k + comp]
            for l in range(2):
                j.refOff[l], j.refStride[l] = (2 * l + comp) * csz + (M // 2 + y // 2) * SC + M // 2 + x // 2, SC
            j.mv[0][0], j.mv[0][1], j.mv[1][0], j.mv[1][1] = a, b, c, d
            j.predOff, j.predStride, j.width, j.height, j.bitDepth, j.puX, j.puY, j.mvdRow = cpos, w // 2, w, h, 10, x, y, k
            cpos += w * h // 4
    d_cj, d_cpred = ctx.to_device(np.frombuffer(cjobs, np.uint8)), ctx.alloc(2 * cpos)
    ctx.dmvr_chroma_batch(pic, 0, d_refc.ptr, d_cpred.ptr, 0, d_cj.ptr, 2 * len(meta), 128, 128, d_mvd.ptr)
    assert np.array_equal(d_cpred.to_host(np.int16), z["outc"])


def test_lfnst_golden(ctx):
    """vtmhip_fwdLfnstNxN / vtmhip_invLfnstNxN and the batched call vs the vectors recorded from the reference; matrices uploaded as caller data."""
    from vtm_amd.lib import LfnstJob
    z = np.load(os.path.join(G, "lfnst.npz"))
    ctx.lfnst_set_tables(z["m8"], z["m4"])
    meta = z["meta"].tolist()
    jobs = (LfnstJob * len(meta))()
    for k, (inverse, mode, index, size, zo) in enumerate(meta):
        n = 48 if size > 4 else 16
        if k % 8 == 0:
            got = ctx.lfnst(inverse, z["src"][k], mode, index, size, zo)
            assert np.array_equal(got[:n], z["out"][k][:n]), (k, inverse, mode, index, size, zo)
        j = jobs[k]
        j.srcOff = j.dstOff = 48 * k
        j.mode, j.index, j.size, j.zeroOutSize, j.inverse = mode, index, size, zo, inverse
    d_src, d_jobs, d_dst = ctx.to_device(np.ascontiguousarray(z["src"]).reshape(-1)), ctx.to_device(np.frombuffer(jobs, np.uint8)), ctx.alloc(4 * 48 * len(meta))
    ctx.lfnst_batch(d_src.ptr, d_dst.ptr, d_jobs.ptr, len(meta))
    got = d_dst.to_host(np.int32).reshape(len(meta), 48)
    for k, (inverse, mode, index, size, zo) in enumerate(meta):
        n = 48 if size > 4 else 16
        assert np.array_equal(got[k, :n], z["out"][k][:n]), (k, inverse, mode, index, size, zo)
