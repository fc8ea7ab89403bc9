"""GPU parity: 1-D transform pointer surface, batched xT / xIT, scalar quant / dequant vs the CPU oracle.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from vtm_amd.lib import QuantJob, TrJob

pytestmark = pytest.mark.gpu

DCT2, DCT8, DST7 = 0, 1, 2


def _skip(t, n):
    return 16 if (t != DCT2 and n == 32) else (n - 32 if n > 32 else 0)


def _legal_types(n):
    return (DCT2, DCT8, DST7) if 4 <= n <= 32 else (DCT2,)


def test_fast_trans_pointer_surface_matches_oracle(ctx):
    L = ol.oracle()
    rng = np.random.default_rng(31)
    for n in (2, 4, 8, 16, 32, 64):
        for t in _legal_types(n):
            for line in (1, 2, 4, 8, 16, 32, 64):
                sk1s = {0} | ({16} if line == 32 else set()) | ({32} if line == 64 else set())
                for sk1 in sk1s:
                    for sk2 in {0, _skip(t, n)}:
                        for amp, shift in ((512, 1), (32767, 7), (1 << 20, 10)):   # the last one wraps in 32 bits, like the reference
                            src = rng.integers(-amp, amp, line * n).astype(np.int32)
                            exp = np.zeros(line * n, np.int32)
                            assert L.vo_fwd_trans(t, n, ol.P(src), ol.P(exp), shift, line, sk1, sk2) == 0
                            got = ctx.fastFwdTrans(t, n, src, shift, line, sk1, sk2)
                            assert np.array_equal(got, exp), ("fwd", t, n, line, sk1, sk2, amp)
                            assert L.vo_inv_trans(t, n, ol.P(src), ol.P(exp), shift, line, sk1, sk2, -32768, 32767) == 0
                            got = ctx.fastInvTrans(t, n, src, shift, line, sk1, sk2)
                            assert np.array_equal(got, exp), ("inv", t, n, line, sk1, sk2, amp)


def test_null_table_slots_are_rejected(ctx):
    from vtm_amd.lib import VtmHipError
    z = np.zeros(64 * 64, np.int32)
    for (t, n) in ((DST7, 2), (DCT8, 64), (DST7, 64), (DCT8, 2), (DCT2, 128), (DCT2, 3)):
        with pytest.raises(VtmHipError):
            ctx.fastFwdTrans(t, n, z, 1, 1, 0, 0)


def _tu_list():
    tus = []
    for w in (1, 2, 4, 8, 16, 32, 64):
        for h in (1, 2, 4, 8, 16, 32, 64):
            if w == 1 and h == 1:
                continue
            for th in _legal_types(w) if w > 1 else (DCT2,):
                for tv in _legal_types(h) if h > 1 else (DCT2,):
                    tus.append((w, h, th, tv))
    return tus


@pytest.mark.parametrize("bd", [10, 8])
def test_xT_xIT_batch_matches_oracle(ctx, bd):
    L = ol.oracle()
    rng = np.random.default_rng(32 + bd)
    tus = _tu_list() * 2
    n = len(tus)
    stride = 80
    resi = np.zeros((n * 64, stride), np.int16)
    jobs = (TrJob * n)()
    exp_coef = np.zeros((n, 64 * 64), np.int32)
    amp = 1 << bd
    for k, (w, h, th, tv) in enumerate(tus):
        blk = rng.integers(-amp + 1, amp, (h, w)).astype(np.int16)
        if k % 7 == 0:
            blk[:] = amp - 1 if k % 14 == 0 else -(amp - 1)   # dc extremes
        resi[k * 64:k * 64 + h, 3:3 + w] = blk
        j = jobs[k]
        j.srcOff, j.dstOff, j.srcStride, j.dstStride = k * 64 * stride + 3, k * 4096, stride, stride
        j.width, j.height, j.typeHor, j.typeVer, j.bitDepth = w, h, th, tv, bd
        e = np.zeros(w * h, np.int32)
        assert L.vo_fwd_2d(C.c_void_p(resi.ctypes.data + 2 * j.srcOff), stride, w, h, bd, th, tv, ol.P(e)) == 0
        exp_coef[k, :w * h] = e
    d_resi = ctx.to_device(resi)
    d_coef = ctx.to_device(np.zeros((n, 4096), np.int32))
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_sum = ctx.alloc(4 * n, np.int32)
    ctx.xT_batch(d_resi.ptr, d_coef.ptr, d_jobs.ptr, n, 64, 64, d_sum.ptr)
    got = d_coef.to_host().reshape(n, 4096)
    sums = d_sum.to_host()
    for k, (w, h, th, tv) in enumerate(tus):
        assert np.array_equal(got[k, :w * h], exp_coef[k, :w * h]), ("xT", w, h, th, tv)
        assert sums[k] == np.abs(exp_coef[k, :w * h].astype(np.int64)).sum(), ("sumAbs", w, h)

    # inverse of (perturbed) coefficients: coefficient range as dequant produces it (16-bit clipped)
    coefs = np.zeros((n, 4096), np.int32)
    exp_resi = np.zeros((n * 64, stride), np.int16)
    for k, (w, h, th, tv) in enumerate(tus):
        c = (exp_coef[k, :w * h] // 3) * 3
        c2 = c.reshape(h, w).copy()
        sw, sh = _skip(th, w), _skip(tv, h)
        if sw:
            c2[:, w - sw:] = 0
        if sh:
            c2[h - sh:, :] = 0
        coefs[k, :w * h] = c2.reshape(-1)
        jobs[k].srcOff, jobs[k].dstOff = k * 4096, k * 64 * stride + 5
        e = np.zeros((h, stride), np.int16)
        assert L.vo_inv_2d(ol.P(np.ascontiguousarray(coefs[k, :w * h])), w, h, bd, th, tv, ol.P(e), stride) == 0
        exp_resi[k * 64:k * 64 + h, 5:5 + w] = e[:, :w]
    d_coef2 = ctx.to_device(coefs)
    d_out = ctx.to_device(np.zeros((n * 64, stride), np.int16))
    d_jobs2 = ctx.to_device(np.frombuffer(jobs, np.uint8))
    ctx.xIT_batch(d_coef2.ptr, d_out.ptr, d_jobs2.ptr, n, 64, 64)
    got_r = d_out.to_host().reshape(n * 64, stride)
    assert np.array_equal(got_r, exp_resi)


def test_transform_roundtrip_property(ctx):
    """Size-independent property: xIT(xT(r)) reproduces r within +-6 (integer-transform rounding; the plain-C oracle
    shows at most 5 at 32x32) for every DCT2 size up to 32, where nothing is zeroed out."""
    rng = np.random.default_rng(5)
    sizes = [(w, h) for w in (4, 8, 16, 32) for h in (4, 8, 16, 32)]
    n = len(sizes)
    resi = np.zeros((n * 32, 32), np.int16)
    jobs = (TrJob * n)()
    for k, (w, h) in enumerate(sizes):
        resi[k * 32:k * 32 + h, :w] = rng.integers(-300, 300, (h, w))
        j = jobs[k]
        j.srcOff, j.dstOff, j.srcStride, j.dstStride, j.width, j.height, j.bitDepth = k * 1024, k * 1024, 32, 32, w, h, 10
    d_resi = ctx.to_device(resi)
    d_coef = ctx.alloc(4 * n * 1024, np.int32)
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_back = ctx.to_device(np.zeros((n * 32, 32), np.int16))
    ctx.xT_batch(d_resi.ptr, d_coef.ptr, d_jobs.ptr, n, 32, 32, None)
    ctx.xIT_batch(d_coef.ptr, d_back.ptr, d_jobs.ptr, n, 32, 32)
    back = d_back.to_host().reshape(n * 32, 32)
    assert np.abs(back.astype(int) - resi).max() <= 6


def test_quant_dequant_batch_matches_oracle(ctx):
    L = ol.oracle()
    rng = np.random.default_rng(41)
    cases = [(w, h, qp, irap, ts) for w in (4, 8, 16, 32, 64) for h in (4, 8, 16, 32, 64) for qp in (22, 27, 32, 37, 51)
             for irap in (0, 1) for ts in (0, 1) if not (ts and max(w, h) > 32)]
    n = len(cases)
    coef = np.zeros((n, 4096), np.int32)
    jobs = (QuantJob * n)()
    exp_q, exp_du, exp_dq = (np.zeros((n, 4096), np.int32) for _ in range(3))
    exp_sum = np.zeros(n, np.int32)
    for k, (w, h, qp, irap, ts) in enumerate(cases):
        c = rng.integers(-32768, 32768, w * h).astype(np.int32)
        c[rng.random(w * h) < 0.5] //= 64
        coef[k, :w * h] = c
        base_qp = qp + 12   # 10-bit: qpBdOffset 12 (Quant.cpp:65-104)
        j = jobs[k]
        j.srcOff, j.dstOff, j.width, j.height, j.qpPer, j.qpRem = k * 4096, k * 4096, w, h, base_qp // 6, base_qp % 6
        j.bitDepth, j.isIRAP, j.isTransformSkip = 10, irap, ts
        s = C.c_int32()
        L.vo_quant(ol.P(np.ascontiguousarray(c)), w, h, 10, j.qpPer, j.qpRem, irap, ts, ol.P(exp_q[k]), ol.P(exp_du[k]), C.byref(s))
        exp_sum[k] = s.value
        L.vo_dequant(ol.P(exp_q[k]), w, h, 10, j.qpPer, j.qpRem, ts, ol.P(exp_dq[k]))
    d_coef = ctx.to_device(coef)
    d_q = ctx.to_device(np.zeros((n, 4096), np.int32))
    d_du = ctx.to_device(np.zeros((n, 4096), np.int32))
    d_dq = ctx.to_device(np.zeros((n, 4096), np.int32))
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_sum = ctx.alloc(4 * n, np.int32)
    ctx.quant_batch(d_coef.ptr, d_q.ptr, d_du.ptr, d_jobs.ptr, n, d_sum.ptr)
    ctx.dequant_batch(d_q.ptr, d_dq.ptr, d_jobs.ptr, n)
    q, du, dq = (d.to_host().reshape(n, 4096) for d in (d_q, d_du, d_dq))
    sm = d_sum.to_host()
    for k, (w, h, qp, irap, ts) in enumerate(cases):
        assert np.array_equal(q[k, :w * h], exp_q[k, :w * h]), ("q", cases[k])
        assert np.array_equal(du[k, :w * h], exp_du[k, :w * h]), ("deltaU", cases[k])
        assert np.array_equal(dq[k, :w * h], exp_dq[k, :w * h]), ("dequant", cases[k])
        assert sm[k] == exp_sum[k], ("absSum", cases[k])


@pytest.mark.parametrize("big,uniform", [(False, False), (True, False), (False, True), (True, True)])
def test_tu_chain_matches_oracle(ctx, big, uniform):
    """Fused xT -> quant -> dequant -> xIT -> SSE (one launch) vs the oracle's separate steps; all 2-D sizes and type pairs."""
    from vtm_amd.lib import TuJob, TuResult
    L = ol.oracle()
    rng = np.random.default_rng(77 + big)
    tus = [t for t in _tu_list() if t[0] > 1 and t[1] > 1 and ((t[0] * t[1] > 256) == big)]
    n = len(tus)
    stride = 72
    resi = np.zeros((n * 64, stride), np.int16)
    jobs = (TuJob * n)()
    exp, exp_lv, exp_rec = [], np.zeros((n, 4096), np.int32), np.zeros((n, 4096), np.int16)
    for k, (w, h, th, tv) in enumerate(tus):
        amp = int(rng.choice([8, 60, 400, 1023]))
        blk = rng.integers(-amp, amp + 1, (h, w)).astype(np.int16)
        resi[k * 64:k * 64 + h, 4:4 + w] = blk
        qp = int(rng.choice([22, 27, 32, 37])) + 12
        irap = int(rng.integers(0, 2))
        j = jobs[k]
        j.resiOff, j.outOff, j.resiStride, j.width, j.height = k * 64 * stride + 4, k * 4096, stride, w, h
        j.qpPer, j.qpRem, j.typeHor, j.typeVer, j.bitDepth, j.isIRAP = qp // 6, qp % 6, th, tv, 10, irap
        coef, qc, dq = np.zeros(w * h, np.int32), np.zeros(w * h, np.int32), np.zeros(w * h, np.int32)
        blk_c = np.ascontiguousarray(blk)
        assert L.vo_fwd_2d(ol.P(blk_c), w, w, h, 10, th, tv, ol.P(coef)) == 0
        s = C.c_int32()
        L.vo_quant(ol.P(coef), w, h, 10, j.qpPer, j.qpRem, irap, 0, ol.P(qc), None, C.byref(s))
        L.vo_dequant(ol.P(qc), w, h, 10, j.qpPer, j.qpRem, 0, ol.P(dq))
        rec = np.zeros((h, w), np.int16)
        assert L.vo_inv_2d(ol.P(dq), w, h, 10, th, tv, ol.P(rec), w) == 0
        exp.append((ol.o_dist(2, blk_c, rec, w, h), int(np.abs(coef.astype(np.int64)).sum()), s.value))
        exp_lv[k, :w * h], exp_rec[k, :w * h] = qc, rec.reshape(-1)
    d_resi = ctx.to_device(resi)
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_res = ctx.alloc(16 * n)
    d_lv, d_rec = ctx.to_device(np.zeros((n, 4096), np.int32)), ctx.to_device(np.zeros((n, 4096), np.int16))
    if big and not uniform:
        ctx.tu_chain_batch(d_resi.ptr, d_jobs.ptr, n, 64, 64, d_res.ptr, d_lv.ptr, d_rec.ptr)
    else:
        # w*h <= 256 covers shapes from 64x4 to 4x64; one launch per shape (maxWidth x maxHeight sizes the LDS and picks 64 threads per TU)
        acc = np.zeros((n, 16), np.uint8)
        for shape in sorted({(t[0], t[1]) for t in tus}):
            idx = [k for k, t in enumerate(tus) if (t[0], t[1]) == shape]
            sub = (TuJob * len(idx))()
            for i, k in enumerate(idx):
                C.memmove(C.byref(sub[i]), C.byref(jobs[k]), C.sizeof(TuJob))
            d_sub = ctx.to_device(np.frombuffer(sub, np.uint8))
            d_r = ctx.alloc(16 * len(idx))
            ctx.tu_chain_batch(d_resi.ptr, d_sub.ptr, len(idx), shape[0], shape[1], d_r.ptr, d_lv.ptr, d_rec.ptr, uniform=uniform)
            acc[idx] = d_r.to_host(np.uint8).reshape(len(idx), 16)
        d_res = ctx.to_device(acc)
    res = (TuResult * n).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    got = [(r.sse, r.sumAbs, r.absSum) for r in res]
    lv, rc = d_lv.to_host().reshape(n, 4096), d_rec.to_host().reshape(n, 4096)
    for k, (w, h, th, tv) in enumerate(tus):
        assert got[k] == exp[k], (tus[k], got[k], exp[k])
        assert np.array_equal(lv[k, :w * h], exp_lv[k, :w * h]) and np.array_equal(rc[k, :w * h], exp_rec[k, :w * h]), tus[k]


@pytest.mark.parametrize("size", [(8, 8), (16, 16), (32, 32), (64, 64), (32, 8), (8, 32), (16, 64)])
def test_xT_uniform_batch_matches_oracle(ctx, size):
    """vtmhip_xT_uniform_batch_dev: forward transforms of the MTS candidates of one TU size (TuJob table), coefficients and sum|coef|."""
    from vtm_amd.lib import TuJob, TuResult
    L = ol.oracle()
    w, h = size
    rng = np.random.default_rng(700 + w + h)
    cands = [(0, 0)] if max(w, h) > 32 else [(0, 0), (2, 2), (1, 2), (2, 1), (1, 1)]
    nblk = 37
    n = nblk * len(cands)
    stride = w + 6
    resi = rng.integers(-1023, 1024, (nblk, h, stride)).astype(np.int16)
    jobs = (TuJob * n)()
    exp = np.zeros((n, h * w), np.int32)
    for k in range(n):
        b, (th, tv) = k % nblk, cands[k // nblk]
        j = jobs[k]
        j.resiOff, j.outOff, j.resiStride, j.width, j.height = b * h * stride, k * w * h, stride, w, h
        j.typeHor, j.typeVer, j.bitDepth, j.qpPer, j.qpRem = th, tv, 10, 7, 2
        assert L.vo_fwd_2d(C.c_void_p(resi.ctypes.data + 2 * j.resiOff), stride, w, h, 10, th, tv, ol.P(exp[k])) == 0
    d_resi, d_jobs = ctx.to_device(resi), ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_coef, d_res = ctx.alloc(4 * n * w * h), ctx.alloc(C.sizeof(TuResult) * n)
    ctx.xT_uniform_batch(d_resi.ptr, d_jobs.ptr, n, w, h, d_coef.ptr, d_res.ptr)
    got = d_coef.to_host(np.int32).reshape(n, w * h)
    assert np.array_equal(got, exp)
    res = (TuResult * n).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    assert [r.sumAbs for r in res] == [int(np.abs(exp[k]).sum()) for k in range(n)]


@pytest.mark.parametrize("uniform", [False, True])
def test_transform_skip_candidate_of_the_fused_chain(ctx, uniform):
    """The MTS_SKIP candidate (TuJob.typeHor == 3): xTransformSkip -> Quant::quant(TS) -> dequant(TS) -> xITransformSkip -> SSE, mixed with
    DCT2 jobs in the generic launch and alone in the uniform one; sumAbs = sum |residual| (vtmhip_mts_select2 scales it)."""
    from vtm_amd.lib import TuJob, TuResult
    L = ol.oracle()
    rng = np.random.default_rng(991 + uniform)
    shapes = [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (4, 16), (32, 8), (16, 32)]
    for (w, h) in shapes:
        n = 24
        stride = w + 5
        resi = rng.integers(-700, 701, (n, h, stride)).astype(np.int16)
        resi[rng.random(resi.shape) < 0.5] //= 40
        jobs = (TuJob * n)()
        exp = []
        exp_lv, exp_rec = np.zeros((n, w * h), np.int32), np.zeros((n, w * h), np.int16)
        for k in range(n):
            ts = uniform or (k % 2 == 0)
            qp = int(rng.choice([22, 27, 32, 37])) + 12
            irap = int(rng.integers(0, 2))
            j = jobs[k]
            j.resiOff, j.outOff, j.resiStride, j.width, j.height = k * h * stride, k * w * h, stride, w, h
            j.qpPer, j.qpRem, j.bitDepth, j.isIRAP = qp // 6, qp % 6, 10, irap
            j.typeHor = j.typeVer = 3 if ts else 0
            blk = np.ascontiguousarray(resi[k, :, :w])
            coef, qc, dq = np.zeros(w * h, np.int32), np.zeros(w * h, np.int32), np.zeros(w * h, np.int32)
            s = C.c_int32()
            rec = np.zeros((h, w), np.int16)
            if ts:
                coef[:] = blk.reshape(-1)
                L.vo_quant(ol.P(coef), w, h, 10, j.qpPer, j.qpRem, irap, 1, ol.P(qc), None, C.byref(s))
                L.vo_dequant(ol.P(qc), w, h, 10, j.qpPer, j.qpRem, 1, ol.P(dq))
                rec[:] = dq.reshape(h, w).astype(np.int16)   # xITransformSkip: Pel( coefficient )
            else:
                assert L.vo_fwd_2d(ol.P(blk), w, w, h, 10, 0, 0, ol.P(coef)) == 0
                L.vo_quant(ol.P(coef), w, h, 10, j.qpPer, j.qpRem, irap, 0, ol.P(qc), None, C.byref(s))
                L.vo_dequant(ol.P(qc), w, h, 10, j.qpPer, j.qpRem, 0, ol.P(dq))
                assert L.vo_inv_2d(ol.P(dq), w, h, 10, 0, 0, ol.P(rec), w) == 0
            exp.append((ol.o_dist(2, blk, rec, w, h), int(np.abs(coef.astype(np.int64)).sum()), s.value))
            exp_lv[k], exp_rec[k] = qc, rec.reshape(-1)
        d_resi, d_jobs = ctx.to_device(resi), ctx.to_device(np.frombuffer(jobs, np.uint8))
        d_res = ctx.alloc(16 * n)
        d_lv, d_rec = ctx.to_device(np.zeros((n, w * h), np.int32)), ctx.to_device(np.zeros((n, w * h), np.int16))
        if uniform:
            ctx.tu_ts_chain_batch(d_resi.ptr, d_jobs.ptr, n, w, h, d_res.ptr, d_lv.ptr, d_rec.ptr)
        else:
            ctx.tu_chain_batch(d_resi.ptr, d_jobs.ptr, n, w, h, d_res.ptr, d_lv.ptr, d_rec.ptr)
        res = (TuResult * n).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
        assert [(r.sse, r.sumAbs, r.absSum) for r in res] == exp, (w, h)
        assert np.array_equal(d_lv.to_host().reshape(n, w * h), exp_lv) and np.array_equal(d_rec.to_host().reshape(n, w * h), exp_rec), (w, h)


def test_lfnst_tu_matches_oracle(ctx):
    """vtmhip_lfnst_tu_batch_dev = TrQuant::xFwdLfnst / xInvLfnst on whole TUs (gather, core multiply, scatter along the scan) vs the oracle restatement;
    the core matrices are the golden copy of the reference's tables (tests/golden/lfnst.npz)."""
    import os
    from vtm_amd.lib import LfnstTuJob
    L = ol.oracle()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lfnst.npz"))
    m8, m4 = np.ascontiguousarray(z["m8"]), np.ascontiguousarray(z["m4"])
    ctx.lfnst_set_tables(m8, m4)
    rng = np.random.default_rng(4711)
    shapes = [(4, 4), (8, 8), (4, 16), (16, 4), (8, 16), (16, 16), (32, 8), (64, 64), (4, 8)]
    n = 400
    jobs = (LfnstTuJob * n)()
    coef = np.zeros(n * 4096, np.int32)
    exp = coef.copy()
    for k in range(n):
        w, h = shapes[k % len(shapes)]
        mode, index, tr, inv = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        blk = rng.integers(-2000, 2001, w * h).astype(np.int32)
        coef[k * 4096:k * 4096 + w * h] = blk
        j = jobs[k]
        j.coefOff, j.width, j.height, j.mode, j.index, j.transpose, j.inverse = k * 4096, w, h, mode, index, tr, inv
        M = np.ascontiguousarray((m8 if (w >= 8 and h >= 8) else m4)[mode, index])
        e = blk.copy()
        L.vo_lfnst_tu(ol.P(e), w, h, ol.P(M), tr, inv)
        exp[k * 4096:k * 4096 + w * h] = e
    d_coef, d_jobs = ctx.to_device(coef), ctx.to_device(np.frombuffer(jobs, np.uint8))
    ctx.lfnst_tu_batch(d_coef.ptr, d_jobs.ptr, n)
    got = d_coef.to_host(np.int32)
    assert np.array_equal(got, exp)
    assert not np.array_equal(got, coef)


def test_mixed_tu_batch_is_bucketed_by_shape(ctx):
    """A mixed batch (>= 256 TUs of every shape from 4x4 to 64x64, DCT2 / MTS pairs / transform skip) in ONE call without the uniform promise: the library
    buckets the table by shape on the device (bucket.hpp), runs the register-blocked kernel per {8..64} x {8..64} class and the generic kernel on the
    rest, and returns every result at its own index = the oracle chain TU by TU."""
    import cpu_pis
    from vtm_amd.lib import TuJob, TuResult
    from vtm_amd.pipeline import MTS_IDX_TYPES
    L = ol.oracle()
    rng = np.random.default_rng(4242)
    n = 700
    stride = 72
    resi = np.zeros((n * 64, stride), np.int16)
    jobs = (TuJob * n)()
    exp = []
    shapes = set()
    for k in range(n):
        w, h = int(rng.choice([4, 8, 16, 32, 64])), int(rng.choice([4, 8, 16, 32, 64]))
        mts = int(rng.integers(0, 6)) if max(w, h) <= 32 else 0
        shapes.add((w, h, mts == 1))
        amp = int(rng.choice([8, 60, 400, 1023]))
        blk = np.ascontiguousarray(rng.integers(-amp, amp + 1, (h, w)).astype(np.int16))
        resi[k * 64:k * 64 + h, 4:4 + w] = blk
        qp = int(rng.choice([22, 27, 32, 37])) + 12
        j = jobs[k]
        j.resiOff, j.outOff, j.resiStride, j.width, j.height = k * 64 * stride + 4, k * 4096, stride, w, h
        j.qpPer, j.qpRem, j.bitDepth, j.isIRAP = qp // 6, qp % 6, 10, 0
        j.typeHor, j.typeVer = (3, 3) if mts == 1 else MTS_IDX_TYPES[mts]
        if w == h:
            exp.append(cpu_pis._tu_chain(L, None, blk, w, mts, qp // 6, qp % 6, 10))
        else:   # the oracle steps for a rectangle
            coef, qc, dq, asum = np.zeros(w * h, np.int32), np.zeros(w * h, np.int32), np.zeros(w * h, np.int32), C.c_int32()
            rec = np.zeros((h, w), np.int16)
            if mts == 1:
                coef[:] = blk.reshape(-1)
                L.vo_quant(ol.P(coef), w, h, 10, qp // 6, qp % 6, 0, 1, ol.P(qc), None, C.byref(asum))
                L.vo_dequant(ol.P(qc), w, h, 10, qp // 6, qp % 6, 1, ol.P(dq))
                rec[:] = dq.reshape(h, w).astype(np.int16)
            else:
                th, tv = MTS_IDX_TYPES[mts]
                assert L.vo_fwd_2d(ol.P(blk), w, w, h, 10, th, tv, ol.P(coef)) == 0
                L.vo_quant(ol.P(coef), w, h, 10, qp // 6, qp % 6, 0, 0, ol.P(qc), None, C.byref(asum))
                L.vo_dequant(ol.P(qc), w, h, 10, qp // 6, qp % 6, 0, ol.P(dq))
                assert L.vo_inv_2d(ol.P(dq), w, h, 10, th, tv, ol.P(rec), w) == 0
            exp.append((int(ol.o_dist(2, blk, rec, w, h)), int(np.abs(coef.astype(np.int64)).sum()), asum.value))
    assert len(shapes) >= 30
    d_resi = ctx.to_device(resi)
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_res = ctx.alloc(16 * n)
    ctx.tu_chain_batch(d_resi.ptr, d_jobs.ptr, n, 64, 64, d_res.ptr, None, None)
    res = (TuResult * n).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    got = [(int(r.sse), int(r.sumAbs), int(r.absSum)) for r in res]
    bad = [k for k in range(n) if got[k] != exp[k]]
    assert not bad, [(jobs[k].width, jobs[k].height, jobs[k].typeHor, got[k], exp[k]) for k in bad[:6]]


def test_mts_select_batch_matches_oracle(ctx):
    """vtmhip_mts_select_batch_dev (the pre-selection of TrQuant::transformNxN( trModes, maxCand ) for every TU of a level, from the sum |coef| of its candidates)
    vs vo_mts_select (pinned against the real member in tests/test_oracle_vs_ref.py): candidate lists with and without the transform-skip entry, every TU
    shape class of the threshold table, maxCand 0 .. 4, costs that tie with the threshold."""
    from vtm_amd.lib import TuResult
    L = ol.oracle()
    rng = np.random.default_rng(4711)
    for it in range(24):
        w, h = int(rng.choice([4, 8, 16, 32])), int(rng.choice([4, 8, 16, 32]))
        modes = [0] + ([1] if it % 3 else []) + [2, 3, 4, 5][:int(rng.integers(0, 5))]
        max_cand = int(rng.integers(0, 5))
        ntu, nc = 700, len(modes)
        base = rng.integers(0, 200000, ntu)
        sums = np.zeros((nc, ntu), np.int32)
        for c in range(nc):
            f = rng.choice([0.5, 0.9, 1.0, 1.2, 1.3, 1.4, 1.5, 2.0], ntu)
            sums[c] = (base * f).astype(np.int32) if c else base
            if modes[c] == 1:
                sums[c] = (sums[c] * rng.choice([0.25, 1.0, 4.0])).astype(np.int32)     # sum |residual| of the transform-skip candidate: another scale
        res = np.zeros(nc * ntu, np.dtype(TuResult))
        res["sumAbs"] = sums.reshape(-1)
        d_res = ctx.to_device(res.view(np.uint8))
        d_test = ctx.alloc(nc * ntu)
        ctx.mts_select_batch(d_res.ptr, ntu, modes, w, h, 10, max_cand, d_test.ptr)
        got = d_test.to_host(np.uint8).reshape(nc, ntu)
        marr = np.array(modes, np.uint8)
        pruned = 0
        for t in range(0, ntu, 7):
            col = np.ascontiguousarray(sums[:, t])
            exp = np.zeros(nc, np.uint8)
            L.vo_mts_select(ol.P(col), ol.P(marr), nc, w, h, 10, 15, max_cand, ol.P(exp))
            assert list(got[:, t]) == list(exp), (w, h, modes, max_cand, list(col), list(got[:, t]), list(exp))
            pruned += int(nc - exp.sum())
        assert nc == 1 or pruned > 0
