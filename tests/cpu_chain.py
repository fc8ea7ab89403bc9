"""The hot-path chain of ONE prediction unit on the CPU: the checker for vtm_amd.pipeline.FrameHotPath (tests) and the
cpu_baseline of bench.py.  TEST INFRASTRUCTURE: uses the oracle (and, when `ref` is given, the real reference's own
xTZSearch / xPatternSearchFracDIF / xPatternSearch / filterHor / filterVer / fastFwdTrans / fastInvTrans)."""
import ctypes as C

import numpy as np

import oracle_lib as ol
from vtm_amd.pipeline import MTS_CANDS


def _mc(L, R, ref_ptr, rs, s, mvx, mvy, bi, dst):
    """xPredInterBlk luma (InterPrediction.cpp:660-815) through the reference's filterHor/filterVer when available."""
    if R is None:
        L.vo_mc_luma(C.c_void_p(ref_ptr), rs, s, s, mvx, mvy, bi, 10, 0, ol.P(dst), s)
        return
    xf, yf, rnd = mvx & 15, mvy & 15, 0 if bi else 1
    src = ref_ptr + 2 * ((mvy >> 4) * rs + (mvx >> 4))
    if yf == 0:
        R.ref_if_hor(1, 0, C.c_void_p(src), rs, ol.P(dst), s, s, s, xf, rnd, 10, 0, 0, 0)
    elif xf == 0:
        R.ref_if_ver(1, 0, C.c_void_p(src), rs, ol.P(dst), s, s, s, yf, 1, rnd, 10, 0, 0, 0)
    else:
        tmp = np.zeros((s + 7, s), np.int16)
        R.ref_if_hor(1, 0, C.c_void_p(src - 2 * 3 * rs), rs, ol.P(tmp), s, s, s + 7, xf, 0, 10, 0, 0, 0)
        R.ref_if_ver(1, 0, C.c_void_p(tmp.ctypes.data + 2 * 3 * s), s, ol.P(dst), s, s, s, yf, 0, rnd, 10, 0, 0, 0)


def _fwd2d(L, R, r_tu, ts, th, tv, coef):
    if R is None:
        assert L.vo_fwd_2d(ol.P(r_tu), ts, ts, ts, 10, th, tv, ol.P(coef)) == 0
        return
    # TrQuant::xT (TrQuant.cpp:776-851) on the reference's own fastFwdTrans entries
    lg = int(np.log2(ts))
    skip = lambda t: 16 if (t != 0 and ts == 32) else (ts - 32 if ts > 32 else 0)   # noqa: E731
    blk = r_tu.astype(np.int32).reshape(-1)
    tmp = np.zeros(ts * ts, np.int32)
    R.ref_fwd_trans(th, lg - 1, ol.P(blk), ol.P(tmp), lg + 10 + 6 - 15, ts, 0, skip(th))
    R.ref_fwd_trans(tv, lg - 1, ol.P(tmp), ol.P(coef), lg + 6, ts, skip(th), skip(tv))


def _inv2d(L, R, dq, ts, th, tv, rec):
    if R is None:
        assert L.vo_inv_2d(ol.P(dq), ts, ts, 10, th, tv, ol.P(rec), ts) == 0
        return
    lg = int(np.log2(ts))
    skip = lambda t: 16 if (t != 0 and ts == 32) else (ts - 32 if ts > 32 else 0)   # noqa: E731
    tmp, blk = np.zeros(ts * ts, np.int32), np.zeros(ts * ts, np.int32)
    R.ref_inv_trans(tv, lg - 1, ol.P(dq), ol.P(tmp), 7, ts, skip(th), skip(tv), -32768, 32767)
    R.ref_inv_trans(th, lg - 1, ol.P(tmp), ol.P(blk), 20 - 10, ts, 0, skip(th), -32768, 32767)
    rec[:] = blk.reshape(ts, ts).astype(np.int16)


def run_pu(cur_np, dpb_ptr, rs, W, H, s, x, y, jobs2, lam, qp_per, qp_rem, ref=None):
    """jobs2: the two vtmhip_tz_job records (list 0 / list 1) of this PU as the device driver used them.
    Returns a dict with every intermediate decision the device pipeline exposes."""
    L, R = ol.oracle(), ref
    org = np.ascontiguousarray(cur_np[y:y + s, x:x + s])
    out = dict(tz=[], mvq=[], cost_uni=[])
    ctxs = []
    for l in (0, 1):
        j = jobs2[l]
        c = ol.MeCtx()
        c.org, c.orgStride = org.ctypes.data, s
        c.ref, c.refStride = dpb_ptr + 2 * int(j["refOff"]), rs
        c.w, c.h, c.subShift, c.bitDepth, c.imvShift = s, s, int(j["subShift"]), 10, 0
        c.mv = ol.MvCost(lam, int(j["predHor"]), int(j["predVer"]), 2)
        c.picW, c.picH, c.puX, c.puY, c.ctuSize = W, H, x, y, 128
        t = ol.TzJob()
        t.mvHor, t.mvVer, t.searchRange, t.firstSearchStop = int(j["mvHor"]), int(j["mvVer"]), int(j["searchRange"]), 1
        r = ol.MeResult()
        (R.ref_tz_search if R else L.vo_tz_search)(C.byref(c), C.byref(t), C.byref(r))
        out["tz"].append((r.mvX, r.mvY, r.cost, r.dist))
        c.subShift = 0
        f = ol.FracResult()
        (R.ref_frac_search if R else L.vo_frac_search)(C.byref(c), r.mvX, r.mvY, 1, 0, C.byref(f))
        out["mvq"].append(((r.mvX << 2) + (f.halfX << 1) + f.qterX, (r.mvY << 2) + (f.halfY << 1) + f.qterY))
        out["cost_uni"].append(f.cost)
        ctxs.append(c)
    mvq, cost = out["mvq"], out["cost_uni"]
    rl = 1 if cost[0] <= cost[1] else 0
    o = 1 - rl
    pred_o = np.zeros((s, s), np.int16)
    _mc(L, R, ctxs[o].ref, rs, s, mvq[o][0] << 2, mvq[o][1] << 2, 0, pred_o)
    org_bi = org.copy()
    if R:
        R.ref_remove_high_freq(ol.P(org_bi), s, ol.P(pred_o), s, s, s)
    else:
        L.vo_remove_high_freq(ol.P(org_bi), s, ol.P(pred_o), s, s, s)
    c = ctxs[rl]
    c.org, c.subShift = org_bi.ctypes.data, 1 if (s > 8 and s <= 64) else 0
    m = ol.MeResult()
    if R:
        rg = (C.c_int * 4)()
        R.ref_set_search_range(C.byref(c), mvq[rl][0] << 2, mvq[rl][1] << 2, 4, rg)
        R.ref_full_search(C.byref(c), rg, C.byref(m))
    else:
        sr = ol.Range()
        L.vo_set_search_range(C.byref(c), mvq[rl][0] << 2, mvq[rl][1] << 2, 4, C.byref(sr))
        L.vo_full_search(C.byref(c), C.byref(sr), C.byref(m))
    c.subShift = 0
    f = ol.FracResult()
    (R.ref_frac_search if R else L.vo_frac_search)(C.byref(c), m.mvX, m.mvY, 1, 0, C.byref(f))
    bx, by = (m.mvX << 2) + (f.halfX << 1) + f.qterX, (m.mvY << 2) + (f.halfY << 1) + f.qterY
    out.update(rl=rl, bi=(bx, by), cost_bi=f.cost >> 1)
    use_bi = (f.cost >> 1) < min(cost)
    out["use_bi"] = use_bi
    pred = np.zeros((s, s), np.int16)
    if use_bi:
        mv = [mvq[0], mvq[1]]
        mv[rl] = (bx, by)
        p = [np.zeros((s, s), np.int16) for _ in range(2)]
        for l in (0, 1):
            _mc(L, R, ctxs[l].ref, rs, s, mv[l][0] << 2, mv[l][1] << 2, 1, p[l])
        if R:
            R.ref_add_avg(ol.P(p[0]), s, ol.P(p[1]), s, ol.P(pred), s, s, s, 10)
        else:
            L.vo_add_avg(ol.P(p[0]), s, ol.P(p[1]), s, ol.P(pred), s, s, s, 10)
    else:
        bl = 1 if cost[1] < cost[0] else 0
        _mc(L, R, ctxs[bl].ref, rs, s, mvq[bl][0] << 2, mvq[bl][1] << 2, 0, pred)
    resi = (org.astype(np.int32) - pred).astype(np.int16)
    ts = min(s, 64)
    q = s // ts
    cands = MTS_CANDS if ts <= 32 else MTS_CANDS[:1]
    out["tus"] = {}
    for qy in range(q):
        for qx in range(q):
            r_tu = np.ascontiguousarray(resi[qy * ts:(qy + 1) * ts, qx * ts:(qx + 1) * ts])
            for ci, (th, tv) in enumerate(cands):
                coef = np.zeros(ts * ts, np.int32)
                _fwd2d(L, R, r_tu, ts, th, tv, coef)
                qc, dq, asum = np.zeros(ts * ts, np.int32), np.zeros(ts * ts, np.int32), C.c_int32()
                L.vo_quant(ol.P(coef), ts, ts, 10, qp_per, qp_rem, 0, 0, ol.P(qc), None, C.byref(asum))
                L.vo_dequant(ol.P(qc), ts, ts, 10, qp_per, qp_rem, 0, ol.P(dq))
                rec = np.zeros((ts, ts), np.int16)
                _inv2d(L, R, dq, ts, th, tv, rec)
                sse = ol.r_dist(2, 0, r_tu, rec, ts, ts) if R else ol.o_dist(2, r_tu, rec, ts, ts)
                out["tus"][(qy * q + qx, ci)] = (int(np.abs(coef.astype(np.int64)).sum()), asum.value, int(sse))
    return out


def compare_with_device(hp_level, i, out):
    """Asserts that PU i of a FrameHotPath level (numpy snapshots in hp_level) equals the CPU chain result `out`."""
    npu, ntu, s, ts = hp_level["npu"], hp_level["ntu"], hp_level["size"], hp_level["ts"]
    q = s // ts
    o, tz = hp_level["out_np"], hp_level["tz_np"]
    for l in (0, 1):
        g = tz[l * npu + i]
        assert out["tz"][l] == (int(g["mvX"]), int(g["mvY"]), int(g["cost"]), int(g["dist"])), ("tz", s, i, l)
        assert out["mvq"][l] == (int(o["mvq_x"][l * npu + i]), int(o["mvq_y"][l * npu + i])), ("frac mv", s, i, l)
        assert out["cost_uni"][l] == int(o["cost_uni"][l * npu + i]), ("frac cost", s, i, l)
    assert out["rl"] == int(o["rl"][i]) and out["bi"] == (int(o["bi_x"][i]), int(o["bi_y"][i])), ("bi", s, i)
    assert out["cost_bi"] == int(o["cost_bi"][i]) and out["use_bi"] == bool(o["use_bi"][i]), ("bi cost", s, i)
    for (tu, ci), (sa, asum, sse) in out["tus"].items():
        k = ci * ntu + i * q * q + tu
        assert (sa, asum, sse) == (int(hp_level["sum_abs_np"][k]), int(hp_level["abs_sum_np"][k]), int(hp_level["sse_np"][k])), ("tu", s, i, tu, ci)


def snapshot(hp):
    """Copies everything compare_with_device needs from the device (numpy), per level."""
    from vtm_amd.pipeline import RES_DT, TZ_DT
    snaps = []
    for lvl in hp.levels:
        snaps.append(dict(size=lvl["size"], npu=lvl["npu"], ntu=lvl["ntu"], ts=lvl["ts"],
                          jobs_np=lvl["jobs"].cpu().numpy().view(TZ_DT).reshape(-1), tz_np=lvl["res"].cpu().numpy().view(RES_DT).reshape(-1),
                          out_np={k: v.cpu().numpy() for k, v in lvl["out"].items()}, sum_abs_np=lvl["sum_abs"].cpu().numpy(),
                          abs_sum_np=lvl["abs_sum"].cpu().numpy(), sse_np=lvl["sse_out"].cpu().numpy()))
    return snaps
