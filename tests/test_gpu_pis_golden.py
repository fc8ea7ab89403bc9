"""GPU replay of tests/golden/pis_enc.npz: 353 predInterSearch calls recorded INSIDE the real reference encoder (oracle/ref_shim_pis.hpp in record mode, gen_pis_golden.py) --
the PU's real AMVP lists, m_uniMvList, block-vector cache hits, FastMEForGenBLowDelay copies, MvdL1Zero pictures, AMVR modes, and what the reference's own members returned --
through ONE vtmhip_predInterSearch_batch_dev call each, compared field by field with the recorded reference results.  The reference itself is not needed here."""
import ctypes as C
import os

import numpy as np
import pytest

import pis_golden as G
from vtm_amd.lib import MeCfg, PicParams, PisBuffers, PisLevelRun

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# ra: the random-access clip (353 records); ldp / ldb: BASELINE config 2's structures (147 / 148 records: P slices with four references; low-delay B with the same four
# pictures in both lists -- every list-1 row a FastMEForGenBLowDelay copy, mvd_l1_zero)
# bcw: the random-access clip with BCW + BcwFast + AffineAmvr (350 records, ~250 of them at a CU-level weight other than the default: given uni rows, weighted bi targets, the
# distortion weight |w| / 8, BcwFast's same-POC skip, the enforced bi mode, the weighted SMVD block)
NPZS = {"ra": os.path.join(GOLDEN, "pis_enc.npz"), "ldp": os.path.join(GOLDEN, "pis_enc_ldp.npz"), "ldb": os.path.join(GOLDEN, "pis_enc_ldb.npz"), "bcw": os.path.join(GOLDEN, "pis_enc_bcw.npz")}
GIVEN_UNI = 2          # VTMHIP_MEJ_GIVEN_UNI
SKIPPED = 2 ** 64 - 1  # biRows[].cost of a row BcwFast skipped


def rate(lam, bits):
    return int(lam * bits)


def same(a, b, fields):
    return all(getattr(a, f) == getattr(b, f) for f in fields)


@pytest.mark.parametrize("structure", ["ra", "ldp", "ldb", "bcw"])
def test_predInterSearch_one_call_per_pu_matches_the_real_encoder(structure):
    from vtm_amd.device import Context
    planes, recs = G.load_npz(NPZS[structure])
    ctx = Context(0)
    dpb_np, bases = G.build_dpb(planes)
    d_dpb = ctx.to_device(dpb_np)
    d_slots = ctx.alloc(C.sizeof(G.PisSlots) + 128 * 128 * 2)
    d_orgbi = ctx.alloc(128 * 128 * 2)
    off = {f: getattr(G.PisSlots, f).offset for f, _ in G.PisSlots._fields_}
    stats = dict(recs=0, rows=0, copies=0, cached=0, bi=0, smvd=0, smvd_won=0, mvdl1zero=0, imv=[0, 0, 0, 0], affine_won=0, dirs=[0, 0, 0, 0], given=0, weighted=0, skipped=0,
                 weighted_bi_won=0, cost_unknown=0, weights={})
    for hd, sin, sout, org, fin in recs:
        w, h, n0, n1 = hd.w, hd.h, hd.numRef[0], hd.numRef[1]
        rows = n0 + n1
        S = G.PisSlots.from_buffer_copy(bytes(sin))
        R = PisLevelRun()
        L = R.pis
        stride = planes[hd.rowPlane[0]][0].stride
        for l in (0, 1):
            for r in range((n0, n1)[l]):
                row = (n0 if l else 0) + r
                pl = planes[hd.rowPlane[row]][0]
                assert pl.stride == stride
                zero = bases[hd.rowPlane[row]] + pl.margin * pl.stride + pl.margin      # the plane's sample (0, 0) inside the rebuilt DPB
                L.refPlaneOff[l][r] = zero
                S.uniJobs[row].refOff = zero + hd.y * pl.stride + hd.x
                assert bases[hd.rowPlane[row]] + hd.rowOff[row] == S.uniJobs[row].refOff
        S.pos[0] = hd.y * stride + hd.x
        base = d_slots.ptr
        L.numPU, L.smvdBit, L.refStride, L.candsGiven, L.biRestricted, L.mvdL1Zero, L.fastMEForGenBLowDelay = 1, hd.smvdBit, stride, 1, hd.biRestricted, hd.mvdL1Zero, hd.fdm
        L.picW, L.picH, L.ctuSize = hd.picW, hd.picH, hd.ctuSize
        L.numRef[0], L.numRef[1] = n0, n1
        for i in range(3):
            L.mbBits[i] = hd.mbBits[i]
        for r in range(4):
            L.list1FromList0[r] = hd.list1FromList0[r]
        L.symRefIdx[0], L.symRefIdx[1] = hd.symRefIdx[0], hd.symRefIdx[1]
        L.givenRows, L.curPoc = hd.givenRows, hd.curPoc
        for l in (0, 1):
            for r in range(4):
                L.refPoc[l][r] = hd.refPoc[l][r]
        L.uniJobs, L.uniOut, L.uniRows, L.distBiP = base + off["uniJobs"], base + off["uniOut"], base + off["uniRows"], base + off["distBiP"]
        L.pus, L.puIn, L.predOther, L.biJobs, L.biOut, L.biRows, L.pos = (base + off[k] for k in ("pus", "puIn", "predOther", "biJobs", "biOut", "biRows", "pos"))
        L.smvdJobs = base + off["smvd"] if hd.hasSmvd else None
        R.uniOut, R.biOut, R.width, R.height = L.uniOut, L.biOut, w, h
        big = max(w, h)
        R.pic = PicParams(hd.picW, hd.picH, hd.ctuSize, hd.bitDepth, 8 if big >= 128 else 2 if big >= 64 else 1)
        R.picBi = PicParams(hd.picW, hd.picH, hd.ctuSize, hd.bitDepth, 16 if big >= 128 else 8 if big >= 64 else 4 if big >= 32 else 1)
        uni_shape = int(ctx.is_uniform_shape(w, h))
        ins = int(S.puIn[0].uniMvInsert)
        R.cfgUni = MeCfg(hd.bipredSearchRange, hd.useHadME, hd.fen13, hd.extendedSettings, hd.firstSearchStop, hd.imv, uni_shape, 1, int(hd.uniMvListSize == 0), 0)
        R.cfgBi = MeCfg(hd.bipredSearchRange, hd.useHadME, hd.fen13, hd.extendedSettings, hd.firstSearchStop, hd.imv, uni_shape, 2, int(hd.uniMvListSize == 0 and not ins), 1)
        B = PisBuffers(base + C.sizeof(G.PisSlots), d_dpb.ptr, None, None, d_orgbi.ptr, None, None)
        d_slots.upload(np.concatenate([np.frombuffer(bytes(S), np.uint8), np.ascontiguousarray(org).view(np.uint8).reshape(-1)]))
        ctx.pred_inter_search_batch(R, B)
        D = G.PisSlots.from_buffer_copy(d_slots.to_host(np.uint8)[:C.sizeof(G.PisSlots)].tobytes())
        E, lam = sout, S.uniJobs[0].motionLambda
        tag = (hd.poc, hd.x, hd.y, w, h, hd.imv)
        # ---- every row: the AMVP stage; the searched rows: xMotionEstimation ----
        for row in range(rows):
            assert same(D.uniJobs[row], E.uniJobs[row], ("mvpIdx", "mvPredHor", "mvPredVer", "bits")) and D.distBiP[row] == E.distBiP[row], ("amvp", tag, row)
            if hd.rowCalls[row]:
                # (a GIVEN row -- xReadBufferedUniMv under a non-default BCW weight -- went through the member too: it returned the buffered vector, re-priced)
                assert E.uniOut[row].intDist == 1, ("the reference did not search a row it should have", tag, row)
                assert same(D.uniOut[row], E.uniOut[row], ("mvHor", "mvVer", "mvPredHor", "mvPredVer", "mvpIdx", "bits", "cost")), ("uni", tag, row, hd.rowCached[row])
                stats["given"] += int(bool(S.uniJobs[row].flags & GIVEN_UNI))
                assert bool(S.uniJobs[row].flags & GIVEN_UNI) == bool((hd.givenRows >> row) & 1)
            else:
                assert E.uniOut[row].intDist == 0
                stats["copies"] += 1
            stats["rows"] += 1
            stats["cached"] += hd.rowCached[row]
        P = D.pus[0]
        # ---- the bi iteration ----
        if fin.biList >= 0:
            assert P.refineList == fin.biList, ("refined list", tag)
            for r in range((n0, n1)[fin.biList]):
                if E.biOut[r].intDist == 0:      # the reference never searched this row: BcwFast's same-POC skip (InterSearch.cpp:2588-2593)
                    assert S.puIn[0].bcwWeightL1 not in (0, 4) and D.biRows[r].cost == SKIPPED, ("bi row the reference skipped", tag, r)
                    stats["skipped"] += 1
                    continue
                assert D.biRows[r].cost != SKIPPED, ("bi row the device skipped", tag, r)
                assert same(D.biJobs[r], E.biJobs[r], ("mvPredHor", "mvPredVer", "mvHor", "mvVer", "mvpIdx", "bits")), ("bi entry", tag, r)
                assert same(D.biOut[r], E.biOut[r], ("mvHor", "mvVer", "mvPredHor", "mvPredVer", "mvpIdx", "bits", "cost")), ("bi", tag, r)
            stats["bi"] += 1
        elif S.puIn[0].bcwWeightL1 not in (0, 4) and n1 and not hd.biRestricted:
            # every row of the refined list skipped: no served bi call told the record mode which list it was
            assert all(D.biRows[r].cost == SKIPPED for r in range((n0, n1)[P.refineList])), ("bi stage without a searched row", tag)
        # ---- the SMVD block ----
        if fin.smvdRan:
            assert hd.hasSmvd
            j, e = D.smvd[0], E.smvd[0]
            mvp = rate(lam, j.mvpIdxBits[j.trace[1].idx[0]] + j.mvpIdxBits[j.trace[1].idx[1]])
            assert tuple(j.trace[1].mv) == tuple(e.trace[1].mv) and j.trace[1].cost - mvp == e.trace[1].cost, ("smvd start", tag)
            assert [j.cand[0][j.trace[1].idx[0]][c] for c in (0, 1)] == list(e.predSym[0]) and [j.cand[1][j.trace[1].idx[1]][c] for c in (0, 1)] == list(e.predSym[1]), ("smvd predictors", tag)
            assert tuple(j.trace[2].mv) == tuple(e.trace[2].mv) and j.trace[2].cost == e.trace[2].cost, ("smvd me", tag)
            if e.trace[3].cost:      # the final predictor check ran in the reference (the search moved the vector)
                assert j.trace[3].cost == e.trace[3].cost and tuple(j.trace[3].idx) == tuple(e.trace[3].idx), ("smvd final", tag)
            stats["smvd"] += 1
        # ---- what the member left in pu ----
        bi = P.interDir == 3
        dev_cost = P.costBi if bi else P.cost[1 if P.interDir == 2 else 0]
        if fin.hevcCost == G.COST_UNKNOWN:      # the member did not store its translational cost (a non-default weight with both affine models buffered, :3054-3057)
            stats["cost_unknown"] += 1
        else:
            assert dev_cost == fin.hevcCost, ("best translational cost", tag, dev_cost, fin.hevcCost)
        wl1 = int(S.puIn[0].bcwWeightL1)
        if wl1 not in (0, 4):
            stats["weighted"] += 1
            stats["weights"][wl1] = stats["weights"].get(wl1, 0) + 1
            stats["weighted_bi_won"] += int(bi and not fin.affine)
        if not fin.affine:
            assert P.interDir == fin.interDir and (not bi or bool(P.smvdMode) == bool(fin.smvdMode)), ("decision", tag)
            for l in (0, 1):
                if not P.interDir & (1 << l):
                    continue
                ref = P.refIdxBi[l] if bi else P.refIdx[l]
                mv = tuple(P.mvBi[l]) if bi else tuple(P.mv[l])
                if bi and P.smvdMode:
                    idx, pred = D.smvd[0].mvpIdxSym[l], tuple(D.smvd[0].predSym[l])
                elif bi and l == P.refineList:
                    idx, pred = D.biRows[ref].mvpIdx, (D.biRows[ref].mvPredHor, D.biRows[ref].mvPredVer)
                elif bi and hd.mvdL1Zero:
                    idx, pred = P.mvpIdxL1Zero, mv
                else:
                    rw = D.uniRows[(n0 if l else 0) + ref]
                    idx, pred = rw.mvpIdx, (rw.mvPredHor, rw.mvPredVer)
                assert ref == fin.refIdx[l] and mv == tuple(fin.mv[l]) and (mv[0] - pred[0], mv[1] - pred[1]) == tuple(fin.mvd[l]) and idx == fin.mvpIdx[l], ("pu", tag, l)
            stats["dirs"][P.interDir] += 1
            stats["smvd_won"] += int(bi and P.smvdMode != 0)
        else:
            stats["affine_won"] += 1
        stats["recs"] += 1
        stats["imv"][hd.imv] += 1
        stats["mvdl1zero"] += hd.mvdL1Zero
    print("pis golden:", structure, stats)
    if structure == "ra":
        assert stats["recs"] >= 300 and stats["copies"] >= 100 and stats["cached"] >= 100 and stats["bi"] >= 200 and stats["smvd"] >= 30 and stats["mvdl1zero"] >= 100
        assert min(stats["imv"][:3]) >= 20 and stats["dirs"][3] >= 50 and stats["dirs"][1] >= 20
    elif structure == "bcw":
        assert stats["recs"] >= 300 and stats["weighted"] >= 200 and stats["given"] >= 300 and stats["skipped"] >= 1 and stats["weighted_bi_won"] >= 50 and len(stats["weights"]) == 4
        assert stats["smvd"] >= 30 and min(stats["imv"][:3]) >= 10
    elif structure == "ldp":
        assert stats["recs"] >= 100 and stats["bi"] == 0 and stats["dirs"][1] >= 50 and stats["dirs"][2] == 0 and stats["dirs"][3] == 0
    else:
        assert stats["recs"] >= 100 and stats["copies"] >= 100 and stats["mvdl1zero"] >= 50 and stats["smvd"] == 0
    ctx.close()


def _level(ctx, planes, bases, hd, n, base_ptr, off_of):
    """a PisLevelRun for n PUs of one slice / shape whose tables start at base_ptr + off_of(name) on the device"""
    R = PisLevelRun()
    L = R.pis
    n0, n1 = hd.numRef[0], hd.numRef[1]
    stride = planes[hd.rowPlane[0]][0].stride
    for l in (0, 1):
        for r in range((n0, n1)[l]):
            pl = planes[hd.rowPlane[(n0 if l else 0) + r]][0]
            L.refPlaneOff[l][r] = bases[hd.rowPlane[(n0 if l else 0) + r]] + pl.margin * pl.stride + pl.margin
    L.numPU, L.smvdBit, L.refStride, L.candsGiven, L.biRestricted, L.mvdL1Zero, L.fastMEForGenBLowDelay = n, hd.smvdBit, stride, 1, hd.biRestricted, hd.mvdL1Zero, hd.fdm
    L.picW, L.picH, L.ctuSize = hd.picW, hd.picH, hd.ctuSize
    L.numRef[0], L.numRef[1] = n0, n1
    for i in range(3):
        L.mbBits[i] = hd.mbBits[i]
    for r in range(4):
        L.list1FromList0[r] = hd.list1FromList0[r]
    L.symRefIdx[0], L.symRefIdx[1] = hd.symRefIdx[0], hd.symRefIdx[1]
    L.givenRows, L.curPoc = hd.givenRows, hd.curPoc
    for l in (0, 1):
        for r in range(4):
            L.refPoc[l][r] = hd.refPoc[l][r]
    for name in ("uniJobs", "uniOut", "uniRows", "distBiP", "pus", "puIn", "predOther", "biJobs", "biOut", "biRows", "pos"):
        setattr(L, name, base_ptr + off_of(name))
    L.smvdJobs = base_ptr + off_of("smvd") if hd.hasSmvd else None
    R.uniOut, R.biOut, R.width, R.height = L.uniOut, L.biOut, hd.w, hd.h
    big = max(hd.w, hd.h)
    R.pic = PicParams(hd.picW, hd.picH, hd.ctuSize, hd.bitDepth, 8 if big >= 128 else 2 if big >= 64 else 1)
    R.picBi = PicParams(hd.picW, hd.picH, hd.ctuSize, hd.bitDepth, 16 if big >= 128 else 8 if big >= 64 else 4 if big >= 32 else 1)
    return R, stride


@pytest.mark.parametrize("structure", ["ra", "ldp", "ldb", "bcw"])
def test_predInterSearch_batches_of_several_pus_equal_the_single_calls(structure):
    """The same records in BATCHES: the PUs of one slice, shape and AMVR mode (up to 17 in the golden file) in one vtmhip_predInterSearch_batch_dev call -- tables in
    the level-order layout (rows (list, refIdx)-major, PU-minor), per-PU m_uniMvList state in vtmhip_pis_pu_in -- must give every PU exactly what its own call gives."""
    from vtm_amd.device import Context
    from vtm_amd.lib import MeJob, MeOut, PisPu, PisPuIn, PisRow, PredJob, SmvdJob
    planes, recs = G.load_npz(NPZS[structure])
    ctx = Context(0)
    dpb_np, bases = G.build_dpb(planes)
    d_dpb = ctx.to_device(dpb_np)
    groups = {}
    for rec in recs:
        hd, sin = rec[0], rec[1]
        # (one batch = one level record: PUs that share the slice, the shape, the AMVR mode and the level-wide switches -- under BCW also the set of given rows; the CU-level
        # weight itself is per PU (vtmhip_pis_pu_in), so PUs of different weights may share a batch when their given rows agree)
        key = (hd.poc, hd.w, hd.h, hd.imv, hd.hasSmvd, hd.biRestricted, hd.mvdL1Zero, hd.numRef[0], hd.numRef[1], hd.givenRows, hd.uniMvListSize == 0, int(sin.puIn[0].uniMvInsert))
        groups.setdefault(key, []).append(rec)
    types = [("uniJobs", MeJob, 8), ("uniOut", MeOut, 8), ("uniRows", PisRow, 8), ("distBiP", C.c_uint64, 8), ("pus", PisPu, 1), ("puIn", PisPuIn, 1), ("predOther", PredJob, 1),
             ("biJobs", MeJob, 4), ("biOut", MeOut, 4), ("biRows", PisRow, 4), ("smvd", SmvdJob, 1), ("pos", C.c_int64, 1)]
    checked = batches = 0
    for key, rs in groups.items():
        if len(rs) < 2:
            continue
        n, hd = len(rs), rs[0][0]
        w, h, n0, n1 = hd.w, hd.h, hd.numRef[0], hd.numRef[1]
        rows = n0 + n1
        # ---- table offsets of the n-PU layout inside one device block ----
        offs, acc = {}, 0
        for name, T, per in types:
            offs[name] = acc
            acc += (C.sizeof(T) * per * n + 255) & ~255
        org_off = acc
        total = org_off + 2 * n * w * h
        host = bytearray(total)

        def put(name, T, idx, obj):
            o = offs[name] + idx * C.sizeof(T)
            host[o:o + C.sizeof(T)] = bytes(obj)
        stride = planes[hd.rowPlane[0]][0].stride
        for p, (hp, sin, sout, org, fin) in enumerate(rs):
            for row in range(rows):
                j = MeJob.from_buffer_copy(bytes(sin.uniJobs[row]))
                pl = planes[hp.rowPlane[row]][0]
                j.refOff = bases[hp.rowPlane[row]] + pl.margin * pl.stride + pl.margin + hp.y * pl.stride + hp.x
                j.orgOff = p * w * h
                put("uniJobs", MeJob, row * n + p, j)
                put("uniOut", MeOut, row * n + p, sin.uniOut[row])      # (input of a GIVEN row: the buffered vector and its distortion)
            put("puIn", PisPuIn, p, sin.puIn[0])
            po = PredJob.from_buffer_copy(bytes(sin.predOther[0]))
            po.orgOff = po.predOff = po.outOff = p * w * h
            put("predOther", PredJob, p, po)
            for r in range(n0 if n1 else 0):
                b = MeJob.from_buffer_copy(bytes(sin.biJobs[r]))
                b.orgOff = b.otherPredOff = p * w * h
                put("biJobs", MeJob, r * n + p, b)
            put("pos", C.c_int64, p, C.c_int64(hp.y * stride + hp.x))
            host[org_off + 2 * p * w * h:org_off + 2 * (p + 1) * w * h] = np.ascontiguousarray(org).tobytes()
        d = ctx.alloc(total)
        d_bi = ctx.alloc(2 * n * w * h)
        R, _ = _level(ctx, planes, bases, hd, n, d.ptr, lambda name: offs[name])
        uni_shape = int(ctx.is_uniform_shape(w, h))
        ins = key[-1]
        R.cfgUni = MeCfg(hd.bipredSearchRange, hd.useHadME, hd.fen13, hd.extendedSettings, hd.firstSearchStop, hd.imv, uni_shape, 1, int(key[-2]), 0)
        R.cfgBi = MeCfg(hd.bipredSearchRange, hd.useHadME, hd.fen13, hd.extendedSettings, hd.firstSearchStop, hd.imv, uni_shape, 2, int(key[-2] and not ins), 1)
        B = PisBuffers(d.ptr + org_off, d_dpb.ptr, None, None, d_bi.ptr, None, None)
        d.upload(np.frombuffer(bytes(host), np.uint8))
        ctx.pred_inter_search_batch(R, B)
        got = d.to_host(np.uint8).tobytes()

        def get(name, T, idx):
            o = offs[name] + idx * C.sizeof(T)
            return T.from_buffer_copy(got[o:o + C.sizeof(T)])
        # ---- every PU on its own ----
        d1 = ctx.alloc(C.sizeof(G.PisSlots) + 2 * w * h)
        off1 = {f: getattr(G.PisSlots, f).offset for f, _ in G.PisSlots._fields_}
        for p, (hp, sin, sout, org, fin) in enumerate(rs):
            S = G.PisSlots.from_buffer_copy(bytes(sin))
            for row in range(rows):
                pl = planes[hp.rowPlane[row]][0]
                S.uniJobs[row].refOff = bases[hp.rowPlane[row]] + pl.margin * pl.stride + pl.margin + hp.y * pl.stride + hp.x
            S.pos[0] = hp.y * stride + hp.x
            R1, _ = _level(ctx, planes, bases, hp, 1, d1.ptr, lambda name: off1[name])
            R1.cfgUni, R1.cfgBi = R.cfgUni, R.cfgBi
            B1 = PisBuffers(d1.ptr + C.sizeof(G.PisSlots), d_dpb.ptr, None, None, d_bi.ptr, None, None)
            d1.upload(np.concatenate([np.frombuffer(bytes(S), np.uint8), np.ascontiguousarray(org).view(np.uint8).reshape(-1)]))
            ctx.pred_inter_search_batch(R1, B1)
            D = G.PisSlots.from_buffer_copy(d1.to_host(np.uint8)[:C.sizeof(G.PisSlots)].tobytes())
            tag = (key, p)
            assert bytes(get("pus", PisPu, p)) == bytes(D.pus[0]), ("pus", tag)
            for row in range(rows):
                assert bytes(get("uniOut", MeOut, row * n + p)) == bytes(D.uniOut[row]) and bytes(get("uniRows", PisRow, row * n + p)) == bytes(D.uniRows[row]), ("uni", tag, row)
                assert get("distBiP", C.c_uint64, row * n + p).value == D.distBiP[row], ("distBiP", tag, row)
            if n1 and not hd.biRestricted:
                for r in range(n0):
                    assert bytes(get("biOut", MeOut, r * n + p)) == bytes(D.biOut[r]) and bytes(get("biRows", PisRow, r * n + p)) == bytes(D.biRows[r]), ("bi", tag, r)
            if hd.hasSmvd:
                a, b = get("smvd", SmvdJob, p), D.smvd[0]
                assert (a.cost, tuple(a.mvCur), tuple(a.mvTar), tuple(a.mvpIdxSym), [tuple(x) for x in a.predSym]) == (b.cost, tuple(b.mvCur), tuple(b.mvTar), tuple(b.mvpIdxSym), [tuple(x) for x in b.predSym]), ("smvd", tag)
                assert [(t.cost, tuple(t.mv), tuple(t.idx)) for t in a.trace] == [(t.cost, tuple(t.mv), tuple(t.idx)) for t in b.trace], ("smvd trace", tag)
            checked += 1
        for buf in (d, d_bi, d1):
            buf.free()
        batches += 1
    print("pis golden batches:", structure, batches, "PUs:", checked)
    assert (batches >= 40 and checked >= 200) if structure == "ra" else (batches >= 10 and checked >= 40)
    ctx.close()
