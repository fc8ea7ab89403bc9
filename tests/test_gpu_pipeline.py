"""GPU end-to-end parity: the frame-level hot path (TZ -> frac -> bi-pred refinement -> residual coding) against the same
chain driven through the CPU oracle, PU by PU, on a small picture."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from vtm_amd import synth
from vtm_amd.pipeline import MTS_CANDS, RES_DT, TZ_DT, FrameHotPath

pytestmark = pytest.mark.gpu


def test_frame_hot_path_matches_oracle_chain():
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    L = ol.oracle()
    W, H = 256, 128
    frames = synth.gen_frames_hard(W, H, 5)
    cur_np = np.ascontiguousarray(frames[2])
    planes, refs, acc = [], [], 0
    for t in (0, 4):
        buf, off, stride = synth.extend_plane(frames[t], margin=160)
        refs.append((acc + off, stride))
        planes.append(buf)
        acc += buf.size
    dpb_np = np.concatenate(planes)
    dev = torch.device("cuda", 0)
    cur, dpb = torch.from_numpy(cur_np).to(dev), torch.from_numpy(dpb_np).to(dev)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lam, qp = 8.0, 32
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, [96, 96], motion_lambda=lam, qp=qp, sizes=(128, 64, 32, 16, 8))
    hp.run(cur.data_ptr(), dpb.data_ptr())
    torch.cuda.synchronize()
    rs = refs[0][1]
    cur_base, dpb_base = cur_np.ctypes.data, dpb_np.ctypes.data
    qp_per, qp_rem = (qp + 12) // 6, (qp + 12) % 6
    checked = 0
    for lvl in hp.levels:
        s, npu = lvl["size"], lvl["npu"]
        jobs = lvl["jobs"].cpu().numpy().view(TZ_DT).reshape(-1)
        tz = lvl["res"].cpu().numpy().view(RES_DT).reshape(-1)
        out = {k: v.cpu().numpy() for k, v in lvl["out"].items()}
        sum_abs, abs_sum, sse = lvl["sum_abs"].cpu().numpy(), lvl["abs_sum"].cpu().numpy(), lvl["sse_out"].cpu().numpy()
        step = max(1, npu // 40)   # sample PUs (the oracle chain is slow); every level, both lists
        for i in range(0, npu, step):
            x, y = int(jobs["puX"][i]), int(jobs["puY"][i])
            org = np.ascontiguousarray(cur_np[y:y + s, x:x + s])
            mvq, cost = [], []
            ctxs = []
            for l in (0, 1):
                j = jobs[l * npu + i]
                c = ol.MeCtx()
                c.org, c.orgStride = org.ctypes.data, s
                c.ref, c.refStride = dpb_base + 2 * int(j["refOff"]), rs
                c.w, c.h, c.subShift, c.bitDepth, c.imvShift = s, s, int(j["subShift"]), 10, 0
                c.mv = ol.MvCost(lam, int(j["predHor"]), int(j["predVer"]), 2)
                c.picW, c.picH, c.puX, c.puY, c.ctuSize = W, H, x, y, 128
                t = ol.TzJob()
                t.mvHor, t.mvVer, t.searchRange, t.firstSearchStop = int(j["mvHor"]), int(j["mvVer"]), 96, 1
                r = ol.MeResult()
                L.vo_tz_search(C.byref(c), C.byref(t), C.byref(r))
                g = tz[l * npu + i]
                assert (r.mvX, r.mvY, r.cost, r.dist) == (int(g["mvX"]), int(g["mvY"]), int(g["cost"]), int(g["dist"]))
                c.subShift = 0
                f = ol.FracResult()
                L.vo_frac_search(C.byref(c), r.mvX, r.mvY, 1, 0, C.byref(f))
                qx, qy = (r.mvX << 2) + (f.halfX << 1) + f.qterX, (r.mvY << 2) + (f.halfY << 1) + f.qterY
                assert (qx, qy, f.cost) == (int(out["mvq_x"][l * npu + i]), int(out["mvq_y"][l * npu + i]), int(out["cost_uni"][l * npu + i]))
                mvq.append((qx, qy))
                cost.append(f.cost)
                ctxs.append(c)
            rl = 1 if cost[0] <= cost[1] else 0
            o = 1 - rl
            assert rl == int(out["rl"][i])
            pred_o = np.zeros((s, s), np.int16)
            L.vo_mc_luma(C.c_void_p(ctxs[o].ref), rs, s, s, mvq[o][0] << 2, mvq[o][1] << 2, 0, 10, 0, ol.P(pred_o), s)
            org_bi = org.copy()
            L.vo_remove_high_freq(ol.P(org_bi), s, ol.P(pred_o), s, s, s)
            c = ctxs[rl]
            c.org, c.subShift = org_bi.ctypes.data, 1 if (s > 8 and s <= 64) else 0
            sr = ol.Range()
            L.vo_set_search_range(C.byref(c), mvq[rl][0] << 2, mvq[rl][1] << 2, 4, C.byref(sr))
            m = ol.MeResult()
            L.vo_full_search(C.byref(c), C.byref(sr), C.byref(m))
            c.subShift = 0
            f = ol.FracResult()
            L.vo_frac_search(C.byref(c), m.mvX, m.mvY, 1, 0, C.byref(f))
            bx, by = (m.mvX << 2) + (f.halfX << 1) + f.qterX, (m.mvY << 2) + (f.halfY << 1) + f.qterY
            assert (bx, by, f.cost >> 1) == (int(out["bi_x"][i]), int(out["bi_y"][i]), int(out["cost_bi"][i]))
            use_bi = (f.cost >> 1) < min(cost)
            assert use_bi == bool(out["use_bi"][i])
            pred = np.zeros((s, s), np.int16)
            if use_bi:
                mv = [mvq[0], mvq[1]]
                mv[rl] = (bx, by)
                p = [np.zeros((s, s), np.int16) for _ in range(2)]
                for l in (0, 1):
                    L.vo_mc_luma(C.c_void_p(ctxs[l].ref), rs, s, s, mv[l][0] << 2, mv[l][1] << 2, 1, 10, 0, ol.P(p[l]), s)
                L.vo_add_avg(ol.P(p[0]), s, ol.P(p[1]), s, ol.P(pred), s, s, s, 10)
            else:
                bl = 1 if cost[1] < cost[0] else 0
                L.vo_mc_luma(C.c_void_p(ctxs[bl].ref), rs, s, s, mvq[bl][0] << 2, mvq[bl][1] << 2, 0, 10, 0, ol.P(pred), s)
            resi = (org.astype(np.int32) - pred).astype(np.int16)
            ts = lvl["ts"]
            q = s // ts
            cands = MTS_CANDS if ts <= 32 else MTS_CANDS[:1]
            for qy in range(q):
                for qx in range(q):
                    tu = i * q * q + qy * q + qx
                    r_tu = np.ascontiguousarray(resi[qy * ts:(qy + 1) * ts, qx * ts:(qx + 1) * ts])
                    for ci, (th, tv) in enumerate(cands):
                        k = ci * lvl["ntu"] + tu
                        coef = np.zeros(ts * ts, np.int32)
                        assert L.vo_fwd_2d(ol.P(r_tu), ts, ts, ts, 10, th, tv, ol.P(coef)) == 0
                        assert int(np.abs(coef.astype(np.int64)).sum()) == int(sum_abs[k]), (s, i, ci)
                        qc, dq, asum = np.zeros(ts * ts, np.int32), np.zeros(ts * ts, np.int32), C.c_int32()
                        L.vo_quant(ol.P(coef), ts, ts, 10, qp_per, qp_rem, 0, 0, ol.P(qc), None, C.byref(asum))
                        assert asum.value == int(abs_sum[k])
                        L.vo_dequant(ol.P(qc), ts, ts, 10, qp_per, qp_rem, 0, ol.P(dq))
                        rec = np.zeros((ts, ts), np.int16)
                        assert L.vo_inv_2d(ol.P(dq), ts, ts, 10, th, tv, ol.P(rec), ts) == 0
                        assert ol.o_dist(2, r_tu, rec, ts, ts) == int(sse[k]), (s, i, ci)
            checked += 1
    assert checked >= 100
    ctx.close()
