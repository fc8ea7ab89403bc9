"""GPU parity: interpolation filters (pointer surface + batch) and the fused fractional search vs the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import me_util
import oracle_lib as ol
from vtm_amd.lib import FracJob, FracResult, IfJob

pytestmark = pytest.mark.gpu


def _taps(name, n, nt):
    L = ol.oracle()
    arr = (C.c_int16 * (n * nt)).in_dll(L, name)
    return np.array(arr, dtype=np.int16).reshape(n, nt)


def test_filter_pointer_surface_matches_oracle(ctx):
    L = ol.oracle()
    rng = np.random.default_rng(21)
    luma, chroma = _taps("vo_luma_filter", 16, 8), _taps("vo_chroma_filter", 32, 4)
    for (w, h) in ((4, 4), (8, 8), (16, 16), (17, 24), (64, 72), (129, 136), (5, 3)):
        for bd in (8, 10):
            pel = ol.i16(rng.integers(0, 1 << bd, (h + 16, w + 16)))
            mid = ol.i16(rng.integers(-8192, 8191, (h + 16, w + 16)))   # 14-bit intermediates
            ss = w + 16
            off = 8 * ss + 8
            cmax = (1 << bd) - 1
            for vertical in (0, 1):
                for (taps, tab) in ((8, luma), (4, chroma), (2, np.array([[16 - f, f] for f in range(16)], np.int16))):
                    for frac in (1, 4, 8, 15) if taps != 4 else (1, 8, 16, 31):
                        for (first, last) in ((1, 0), (0, 1), (1, 1), (0, 0)):
                            src = pel if first else mid
                            exp = np.zeros((h, w), np.int16)
                            L.vo_if_filter(vertical, taps, first, last, C.c_void_p(src.ctypes.data + 2 * off), ss, ol.P(exp), w, w, h,
                                           ol.P(np.ascontiguousarray(tab[frac])), bd, 0, cmax, 0)
                            got = ctx.filter(vertical, taps, first, last, src, off, ss, w, h, tab[frac], bd)
                            assert np.array_equal(got, exp), (w, h, bd, vertical, taps, frac, first, last)
            for (first, last) in ((1, 0), (0, 1), (1, 1), (0, 0)):
                src = pel if first else mid
                exp = np.zeros((h, w), np.int16)
                L.vo_if_copy(first, last, C.c_void_p(src.ctypes.data + 2 * off), ss, ol.P(exp), w, w, h, bd, 0, cmax, 0)
                got = ctx.filter_copy(first, last, src, off, ss, w, h, bd)
                assert np.array_equal(got, exp), ("copy", w, h, bd, first, last)


def test_if_batch_matches_oracle(ctx):
    """A picture-sized batch: H pass (first,!last) of every 16x16 block's (16 x 23) region + V pass (!first,last)."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    luma = _taps("vo_luma_filter", 16, 8)
    rng = np.random.default_rng(4)
    jobs, exp_blocks = [], []
    n = 300
    arr = (IfJob * n)()
    dst_stride = 32
    dst = np.zeros((n * 24, dst_stride), np.int16)
    ref2d = scene.ref_buf.reshape(-1, scene.ref_stride)
    for k in range(n):
        x, y = int(rng.integers(0, 400)), int(rng.integers(0, 224))
        frac = int(rng.integers(1, 16))
        vertical = k & 1
        j = arr[k]
        j.srcOff = scene.ref_off + y * scene.ref_stride + x
        j.dstOff = k * 24 * dst_stride
        j.srcStride, j.dstStride, j.width, j.height = scene.ref_stride, dst_stride, 16, 23
        j.vertical, j.taps, j.isFirst, j.isLast = vertical, 8, 1, k % 3 == 0
        for t in range(8):
            j.coeff[t] = int(luma[frac][t])
        j.clipMin, j.clipMax, j.bitDepth = 0, 1023, 10
        e = np.zeros((23, 16), np.int16)
        L.vo_if_filter(vertical, 8, 1, int(j.isLast), C.c_void_p(scene.ref_buf.ctypes.data + 2 * j.srcOff), scene.ref_stride, ol.P(e), 16,
                       16, 23, ol.P(np.ascontiguousarray(luma[frac])), 10, 0, 1023, 0)
        exp_blocks.append(e)
    d_src = ctx.to_device(scene.ref_buf)
    d_dst = ctx.to_device(dst)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    ctx.if_batch(d_src.ptr, d_dst.ptr, d_jobs.ptr, n)
    got = d_dst.to_host().reshape(n, 24, dst_stride)
    for k in range(n):
        assert np.array_equal(got[k, :23, :16], exp_blocks[k]), k


def _frac_jobs(scene, n, seed):
    rng = np.random.default_rng(seed)
    jobs = []
    while len(jobs) < n:
        w = int(rng.choice([8, 16, 32, 64, 128, 4, 16, 8, 32, 64]))
        h = int(rng.choice([8, 16, 32, 64, 128, 8, 4, 16]))
        if w == 4 and h == 4:
            continue
        x = int(rng.integers(0, (scene.W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (scene.H - h) // 4 + 1)) * 4
        jobs.append(dict(w=w, h=h, x=x, y=y, subShift=0, lam=float(rng.uniform(1, 40)), predHor=int(rng.integers(-64, 64)),
                         predVer=int(rng.integers(-64, 64)), intX=int(rng.integers(-12, 12)), intY=int(rng.integers(-12, 12)),
                         useHad=int(len(jobs) % 4 != 0), alt=int(len(jobs) % 9 == 0)))
    return jobs


@pytest.mark.parametrize("seed", [6, 7])
def test_frac_search_matches_oracle(ctx, seed):
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    jobs = _frac_jobs(scene, 500, seed)
    exp = []
    for j in jobs:
        org = np.ascontiguousarray(scene.cur[j["y"]:j["y"] + j["h"], j["x"]:j["x"] + j["w"]])
        c = me_util.oracle_ctx(scene, j, org)
        fr = ol.FracResult()
        L.vo_frac_search(C.byref(c), j["intX"], j["intY"], j["useHad"], 0, C.byref(fr))
        exp.append((fr.halfX, fr.halfY, fr.qterX, fr.qterY, fr.cost))
    arr = (FracJob * len(jobs))()
    for k, j in enumerate(jobs):
        t = arr[k]
        t.orgOff = j["y"] * scene.W + j["x"]
        t.refOff = scene.ref_off + j["y"] * scene.ref_stride + j["x"]
        t.orgStride, t.refStride, t.width, t.height = scene.W, scene.ref_stride, j["w"], j["h"]
        t.intX, t.intY, t.predHor, t.predVer, t.motionLambda = j["intX"], j["intY"], j["predHor"], j["predVer"], j["lam"]
        t.useHad, t.useAltHpelIf, t.imvShift, t.bitDepth = j["useHad"], 0, 0, 10
    d_cur, d_ref = ctx.to_device(scene.cur), ctx.to_device(scene.ref_buf)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_res = ctx.alloc(16 * len(jobs))
    ctx.frac_search_batch(d_cur.ptr, d_ref.ptr, d_jobs.ptr, len(jobs), 128, 128, d_res.ptr)
    res = (FracResult * len(jobs)).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    got = [(r.halfX, r.halfY, r.qterX, r.qterY, r.cost) for r in res]
    bad = [k for k in range(len(jobs)) if got[k] != exp[k]]
    assert not bad, [(jobs[k], got[k], exp[k]) for k in bad[:5]]


def test_frac_search_half_pel_only_alt_filter(ctx):
    """cu.imv == IMV_HPEL: half-sample refinement only, with the alternative half-sample filter (InterSearch.cpp:4327)."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=False)
    jobs = _frac_jobs(scene, 120, 12)
    arr = (FracJob * len(jobs))()
    exp = []
    for k, j in enumerate(jobs):
        org = np.ascontiguousarray(scene.cur[j["y"]:j["y"] + j["h"], j["x"]:j["x"] + j["w"]])
        c = me_util.oracle_ctx(scene, j, org)
        fr = ol.FracResult()
        L.vo_frac_search(C.byref(c), j["intX"], j["intY"], 1, 1, C.byref(fr))
        exp.append((fr.halfX, fr.halfY, fr.costHalf))
        t = arr[k]
        t.orgOff = j["y"] * scene.W + j["x"]
        t.refOff = scene.ref_off + j["y"] * scene.ref_stride + j["x"]
        t.orgStride, t.refStride, t.width, t.height = scene.W, scene.ref_stride, j["w"], j["h"]
        t.intX, t.intY, t.predHor, t.predVer, t.motionLambda = j["intX"], j["intY"], j["predHor"], j["predVer"], j["lam"]
        t.useHad, t.useAltHpelIf, t.imvShift, t.bitDepth = 1, 1, 1, 10
    d_cur, d_ref = ctx.to_device(scene.cur), ctx.to_device(scene.ref_buf)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_res = ctx.alloc(16 * len(jobs))
    ctx.frac_search_batch(d_cur.ptr, d_ref.ptr, d_jobs.ptr, len(jobs), 128, 128, d_res.ptr)
    res = (FracResult * len(jobs)).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    got = [(r.halfX, r.halfY, r.cost) for r in res]
    assert got == exp


@pytest.mark.parametrize("size", [8, 16, 32, 64, 128])
@pytest.mark.parametrize("use_had,signed", [(1, 0), (0, 0), (1, 1)])
def test_frac_search_tiled_square_path(ctx, size, use_had, signed):
    """uniformSquare fast path (one lane per (PU, candidate, 8x8 tile)) vs the oracle, incl. the signed bi-pred target."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    rng = np.random.default_rng(100 + size)
    cur = scene.cur
    if signed:
        cur = np.ascontiguousarray((2 * cur.astype(np.int32) - rng.integers(0, 1024, cur.shape)).astype(np.int16))
    n = {8: 300, 16: 150, 32: 60, 64: 24, 128: 6}[size]
    arr = (FracJob * n)()
    exp = []
    for k in range(n):
        x = int(rng.integers(0, (416 - size) // 4 + 1)) * 4
        y = int(rng.integers(0, (240 - size) // 4 + 1)) * 4
        j = dict(w=size, h=size, x=x, y=y, subShift=0, lam=float(rng.uniform(1, 40)), predHor=int(rng.integers(-64, 64)),
                 predVer=int(rng.integers(-64, 64)))
        ix, iy = int(rng.integers(-12, 12)), int(rng.integers(-12, 12))
        org = np.ascontiguousarray(cur[y:y + size, x:x + size])
        c = me_util.oracle_ctx(scene, j, org)
        fr = ol.FracResult()
        L.vo_frac_search(C.byref(c), ix, iy, use_had, 0, C.byref(fr))
        exp.append((fr.halfX, fr.halfY, fr.qterX, fr.qterY, fr.cost))
        t = arr[k]
        t.orgOff, t.refOff = y * 416 + x, scene.ref_off + y * scene.ref_stride + x
        t.orgStride, t.refStride, t.width, t.height = 416, scene.ref_stride, size, size
        t.intX, t.intY, t.predHor, t.predVer, t.motionLambda = ix, iy, j["predHor"], j["predVer"], j["lam"]
        t.useHad, t.useAltHpelIf, t.imvShift, t.bitDepth = use_had, 0, 0, 10
    d_cur, d_ref = ctx.to_device(cur), ctx.to_device(scene.ref_buf)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_res = ctx.alloc(16 * n)
    ctx.frac_search_batch(d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, size, size, d_res.ptr, uniform_square=True)
    res = (FracResult * n).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    got = [(r.halfX, r.halfY, r.qterX, r.qterY, r.cost) for r in res]
    bad = [k for k in range(n) if got[k] != exp[k]]
    assert not bad, [(got[k], exp[k]) for k in bad[:5]]


@pytest.mark.parametrize("size", [8, 16, 32, 64, 128])
@pytest.mark.parametrize("mode", ["bd8", "bd12", "mixed"])
def test_frac_search_tiled_bit_depths_and_mixed_batches(ctx, size, mode):
    """The tiled path picks its arithmetic per WORKGROUP: packed 16-bit (every PU of the workgroup asks for the Hadamard cost at bitDepth <= 10; with the identity
    shortcuts for phase 0) or the general 32-bit path.  bd8: 8-bit samples through the packed path (other shifts and offsets than at 10 bits); bd12: the general
    path; mixed: a batch whose jobs alternate between SATD / SAD and between 8 / 10 / 12 bits per job in runs of 5, so that workgroups hold both kinds."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    rng = np.random.default_rng(500 + size)
    n = {8: 300, 16: 150, 32: 60, 64: 24, 128: 6}[size]
    # three sample ranges of the same pictures: 8, 10, 12 bits
    curs = {8: np.ascontiguousarray(scene.cur >> 2), 10: scene.cur, 12: np.ascontiguousarray(scene.cur << 2)}
    refs = {8: np.ascontiguousarray(scene.ref_buf >> 2), 10: scene.ref_buf, 12: np.ascontiguousarray(scene.ref_buf << 2)}
    plane = curs[10].size
    cur_all = np.ascontiguousarray(np.concatenate([curs[8].ravel(), curs[10].ravel(), curs[12].ravel()]))
    ref_all = np.ascontiguousarray(np.concatenate([refs[8].ravel(), refs[10].ravel(), refs[12].ravel()]))
    arr = (FracJob * n)()
    exp = []
    for k in range(n):
        bd = {"bd8": 8, "bd12": 12}.get(mode) or (8, 10, 12, 10)[(k // 5) % 4]
        use_had = 1 if mode != "mixed" else (1, 1, 1, 0, 1)[(k // 5) % 5]
        slot = {8: 0, 10: 1, 12: 2}[bd]
        x = int(rng.integers(0, (416 - size) // 4 + 1)) * 4
        y = int(rng.integers(0, (240 - size) // 4 + 1)) * 4
        j = dict(w=size, h=size, x=x, y=y, subShift=0, lam=float(rng.uniform(1, 40)), predHor=int(rng.integers(-64, 64)), predVer=int(rng.integers(-64, 64)))
        ix, iy = int(rng.integers(-12, 12)), int(rng.integers(-12, 12))
        org = np.ascontiguousarray(curs[bd][y:y + size, x:x + size])
        c = me_util.oracle_ctx(scene, j, org)
        c.ref = refs[bd].ctypes.data + 2 * (scene.ref_off + y * scene.ref_stride + x)
        c.bitDepth = bd
        fr = ol.FracResult()
        L.vo_frac_search(C.byref(c), ix, iy, use_had, 0, C.byref(fr))
        exp.append((fr.halfX, fr.halfY, fr.qterX, fr.qterY, fr.cost))
        t = arr[k]
        t.orgOff, t.refOff = slot * plane + y * 416 + x, slot * refs[10].size + scene.ref_off + y * scene.ref_stride + x
        t.orgStride, t.refStride, t.width, t.height = 416, scene.ref_stride, size, size
        t.intX, t.intY, t.predHor, t.predVer, t.motionLambda = ix, iy, j["predHor"], j["predVer"], j["lam"]
        t.useHad, t.useAltHpelIf, t.imvShift, t.bitDepth = use_had, 0, 0, bd
    d_cur, d_ref = ctx.to_device(cur_all), ctx.to_device(ref_all)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_res = ctx.alloc(16 * n)
    ctx.frac_search_batch(d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, size, size, d_res.ptr, uniform_square=True)
    res = (FracResult * n).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    got = [(r.halfX, r.halfY, r.qterX, r.qterY, r.cost) for r in res]
    bad = [k for k in range(n) if got[k] != exp[k]]
    assert not bad, [(k, got[k], exp[k]) for k in bad[:5]]


RECT_SHAPES = [(16, 8), (8, 16), (32, 8), (8, 32), (32, 16), (16, 32), (64, 16), (16, 64), (64, 32), (32, 64)]


@pytest.mark.parametrize("w,h", RECT_SHAPES)
@pytest.mark.parametrize("use_had,signed,bd", [(1, 0, 10), (0, 0, 10), (1, 1, 10), (1, 0, 12)])
def test_frac_search_tiled_rect_path(ctx, w, h, use_had, signed, bd):
    """The tiled fast path on the binary / ternary split shapes: their SATD tiles are the reference's 16x8 / 8x16 Hadamards (RdCost.cpp:2837-2931),
    formed from two 8x8 lane items through a DPP exchange -- packed 10-bit path, packed 12-bit-difference path (signed bi-pred target) and the
    32-bit path (bitDepth 12) -- vs the oracle."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    rng = np.random.default_rng(1000 + 64 * w + h)
    cur = scene.cur
    if signed:
        cur = np.ascontiguousarray((2 * cur.astype(np.int32) - rng.integers(0, 1024, cur.shape)).astype(np.int16))
    n = max(10, 9000 // (w * h))
    arr = (FracJob * n)()
    exp = []
    for k in range(n):
        x = int(rng.integers(0, (416 - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (240 - h) // 4 + 1)) * 4
        j = dict(w=w, h=h, x=x, y=y, subShift=0, lam=float(rng.uniform(1, 40)), predHor=int(rng.integers(-64, 64)), predVer=int(rng.integers(-64, 64)))
        ix, iy = int(rng.integers(-12, 12)), int(rng.integers(-12, 12))
        org = np.ascontiguousarray(cur[y:y + h, x:x + w])
        c = me_util.oracle_ctx(scene, j, org)
        c.bitDepth = bd
        fr = ol.FracResult()
        L.vo_frac_search(C.byref(c), ix, iy, use_had, 0, C.byref(fr))
        exp.append((fr.halfX, fr.halfY, fr.qterX, fr.qterY, fr.cost))
        t = arr[k]
        t.orgOff, t.refOff = y * 416 + x, scene.ref_off + y * scene.ref_stride + x
        t.orgStride, t.refStride, t.width, t.height = 416, scene.ref_stride, w, h
        t.intX, t.intY, t.predHor, t.predVer, t.motionLambda = ix, iy, j["predHor"], j["predVer"], j["lam"]
        t.useHad, t.useAltHpelIf, t.imvShift, t.bitDepth = use_had, 0, 0, bd
    d_cur, d_ref = ctx.to_device(cur), ctx.to_device(scene.ref_buf)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_res = ctx.alloc(16 * n)
    ctx.frac_search_batch(d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, w, h, d_res.ptr, uniform_square=True)
    res = (FracResult * n).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    got = [(r.halfX, r.halfY, r.qterX, r.qterY, r.cost) for r in res]
    bad = [k for k in range(n) if got[k] != exp[k]]
    assert not bad, [(got[k], exp[k]) for k in bad[:5]]
