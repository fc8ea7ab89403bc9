"""GPU parity: distortion kernels (SAD / SATD / SSE) through the C ABI vs the CPU oracle.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from vtm_amd import synth
from vtm_amd.lib import DistJob

pytestmark = pytest.mark.gpu

SIZES = [(w, h) for w in (4, 8, 12, 16, 24, 32, 48, 64, 128) for h in (4, 8, 16, 32, 64, 128)]
SIZES += [(2, 2), (2, 4), (2, 8), (2, 32), (6, 4), (4, 2), (8, 2), (16, 2), (64, 2), (10, 6)]   # chroma / odd shapes the scalar reference accepts


def test_pointer_surface_matches_oracle(ctx):
    rng = np.random.default_rng(11)
    for (w, h) in SIZES:
        org = ol.i16(rng.integers(-1023, 2047, (h, w + 7)))   # bi-pred ME target range
        cur = ol.i16(rng.integers(0, 1024, (h, w + 3)))
        for ss in (0, 1, 2):
            if h >> ss < 1 or h % (1 << ss):
                continue
            assert ctx.xGetSAD(org, w + 7, cur, w + 3, w, h, ss) == ol.o_dist(0, org, cur, w, h, ss), (w, h, ss)
        assert ctx.xGetHADs(org, w + 7, cur, w + 3, w, h) == ol.o_dist(1, org, cur, w, h), (w, h)
        assert ctx.xGetSSE(org, w + 7, cur, w + 3, w, h) == ol.o_dist(2, org, cur, w, h), (w, h)


def test_pointer_surface_extreme_values(ctx):
    """all-0, all-max, +-32767 checkerboards: full int16 range must not overflow (diffs are 17-bit)."""
    for (w, h) in ((8, 8), (16, 8), (8, 16), (64, 64), (128, 128), (4, 8), (8, 4)):
        yy, xx = np.mgrid[0:h, 0:w]
        chk = np.where((yy + xx) & 1, 32767, -32768).astype(np.int16)
        for org, cur in ((chk, -1 - chk), (np.zeros((h, w), np.int16), np.zeros((h, w), np.int16)),
                         (np.full((h, w), 1023, np.int16), np.zeros((h, w), np.int16)), (chk, chk)):
            org = np.ascontiguousarray(org)
            cur = np.ascontiguousarray(cur.astype(np.int16))
            assert ctx.xGetSAD(org, w, cur, w, w, h, 0) == ol.o_dist(0, org, cur, w, h, 0)
            assert ctx.xGetHADs(org, w, cur, w, w, h) == ol.o_dist(1, org, cur, w, h)
            assert ctx.xGetSSE(org, w, cur, w, w, h) == ol.o_dist(2, org, cur, w, h)


def test_invalid_arguments_return_status(ctx):
    from vtm_amd.lib import VtmHipError
    a = np.zeros((8, 8), np.int16)
    with pytest.raises(VtmHipError):
        ctx.xGetSAD(a, 8, a, 8, 0, 8)
    with pytest.raises(VtmHipError):
        ctx.xGetHADs(a, 8, a, 8, 5, 8)       # odd size: the reference THROWs "Invalid size" (RdCost.cpp:2925-2931)
    with pytest.raises(VtmHipError):
        ctx.xGetHADs(a, 8, a, 8, 256, 8)


def test_dist_batch_matches_oracle(ctx):
    W, H = 416, 240
    fr = synth.gen_frames_hard(W, H, 2)
    cur = np.ascontiguousarray(fr[1])
    ref, off, stride = synth.extend_plane(fr[0])
    rng = np.random.default_rng(3)
    n = 4000
    jobs = (DistJob * n)()
    exp = np.zeros(n, np.uint64)
    for k in range(n):
        w, h = SIZES[int(rng.integers(len(SIZES)))]
        x = int(rng.integers(0, W - w + 1))
        y = int(rng.integers(0, H - h + 1))
        dx, dy = int(rng.integers(-64, 65)), int(rng.integers(-64, 65))
        kind = int(rng.integers(0, 3))
        ss = int(rng.integers(0, 2)) if (kind == 0 and h >= 8) else 0
        j = jobs[k]
        j.orgOff, j.curOff = y * W + x, off + (y + dy) * stride + x + dx
        j.orgStride, j.curStride, j.width, j.height, j.subShift, j.kind = W, stride, w, h, ss, kind
        exp[k] = ol.o_dist(kind, cur, ref.reshape(-1, stride), w, h, ss, org_off=j.orgOff, cur_off=j.curOff)
    d_cur, d_ref = ctx.to_device(cur), ctx.to_device(ref)
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_out = ctx.alloc(8 * n, np.uint64)
    ctx.dist_batch(d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, d_out.ptr)
    got = d_out.to_host()
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (bad[:10], got[bad[:10]], exp[bad[:10]])


@pytest.mark.parametrize("size,r", [((416, 240), 4), ((200, 104), 2), ((64, 32), 0), ((72, 40), 7)])
def test_satd8_grid_matches_oracle(ctx, size, r):
    W, H = size
    fr = synth.gen_frames_hard(max(W, 128), max(H, 128), 2)
    cur = np.ascontiguousarray(fr[1][:H, :W])
    W8 = (W + 7) // 8 * 8
    curp = np.zeros((H, W8), np.int16)
    curp[:, :W] = cur
    ref, off, stride = synth.extend_plane(np.ascontiguousarray(fr[0][:H, :W]), margin=32)
    nd2 = (2 * r + 1) ** 2
    nb = (W // 8) * (H // 8)
    exp = np.zeros(nb * nd2, np.uint64)
    ol.oracle().vo_satd8_grid(ol.P(curp), W8, C.c_void_p(ref.ctypes.data + 2 * off), stride, W, H, r, ol.P(exp))
    d_cur, d_ref = ctx.to_device(curp), ctx.to_device(ref)
    d_out = ctx.alloc(4 * nb * nd2, np.uint32)
    ctx.satd8_grid(d_cur.ptr, W8, d_ref.ptr + 2 * off, stride, W, H, r, d_out.ptr)
    got = d_out.to_host().astype(np.uint64)
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:10]


def test_masked_sad_pointer_surface_and_batch(ctx):
    """DF_SAD_WITH_MASK (GEO merge estimation): pointer surface and batched call vs the oracle; mask walked forwards / backwards in x and y."""
    from vtm_amd.lib import MaskedSadJob
    L = ol.oracle()
    L.vo_sad_mask.restype = C.c_uint64
    rng = np.random.default_rng(405)
    M = 112
    plane = ol.i16(rng.integers(0, 9, (M, M))).reshape(-1)
    n = 400
    jobs = (MaskedSadJob * n)()
    orgs, curs, exp = [], [], []
    pos = 0
    for k in range(n):
        w, h = int(rng.choice([4, 8, 16, 32, 64, 128 if k % 50 == 0 else 8])), int(rng.choice([4, 8, 16, 32, 64]))
        w = min(w, 64)
        org, cur = ol.i16(rng.integers(-1023, 2047, (h, w))), ol.i16(rng.integers(0, 1024, (h, w)))
        sx, rd = (1 if k % 3 else -1), (1 if k % 2 else -1)
        x0 = int(rng.integers(0, M - w)) + (w - 1 if sx < 0 else 0)
        y0 = int(rng.integers(0, M - h)) + (h - 1 if rd < 0 else 0)
        off, ms, ms2, ss = y0 * M + x0, rd * M, -sx * w, (1 if (k % 5 == 0 and h >= 8) else 0)
        e = L.vo_sad_mask(ol.P(org), w, ol.P(cur), w, w, h, ss, C.c_void_p(plane.ctypes.data + 2 * off), ms, sx, ms2)
        exp.append(e)
        if k < 60:
            assert ctx.xGetSADwMask(org, w, cur, w, w, h, plane, off, ms, sx, ms2, ss) == e, (k, w, h, sx, rd, ss)
        j = jobs[k]
        j.orgOff = j.curOff = pos
        j.maskOff, j.orgStride, j.curStride, j.maskStride, j.maskStride2 = off, w, w, ms, ms2
        j.width, j.height, j.subShift, j.stepX = w, h, ss, sx
        orgs.append(org.reshape(-1)); curs.append(cur.reshape(-1))
        pos += w * h
    d_org, d_cur, d_mask = ctx.to_device(np.concatenate(orgs)), ctx.to_device(np.concatenate(curs)), ctx.to_device(plane)
    d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
    d_out = ctx.alloc(8 * n)
    ctx.masked_sad_batch(d_org.ptr, d_cur.ptr, d_mask.ptr, d_jobs.ptr, n, d_out.ptr)
    assert list(d_out.to_host(np.uint64)) == exp


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_dist_uniform_batch_matches_oracle(ctx, kind):
    """vtmhip_dist_uniform_batch_dev: one block size per launch, several jobs per wave; every size class incl. the fall-back for narrow widths,
    signed (bi-pred target) samples, a job count that leaves the last wave partly empty."""
    W, H = 416, 240
    fr = synth.gen_frames_hard(W, H, 2)
    rng = np.random.default_rng(33 + kind)
    cur = np.ascontiguousarray((2 * fr[1].astype(np.int32) - rng.integers(0, 1024, fr[1].shape)).astype(np.int16))
    ref, off, stride = synth.extend_plane(fr[0])
    d_org, d_ref = ctx.to_device(cur), ctx.to_device(ref)
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (16, 8), (8, 16), (32, 8), (8, 4), (4, 8), (4, 4), (64, 16), (24, 8), (12, 16)):
        if kind == 1 and (w % 4 or h % 4):
            continue
        ss = 1 if (kind == 0 and h > 8 and w <= 64) else 0
        n = 333
        jobs = (DistJob * n)()
        exp = []
        for k in range(n):
            x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
            dx, dy = int(rng.integers(-20, 21)), int(rng.integers(-20, 21))
            j = jobs[k]
            j.orgOff, j.curOff, j.orgStride, j.curStride = y * W + x, off + (y + dy) * stride + x + dx, W, stride
            j.width, j.height, j.subShift, j.kind = w, h, ss, kind
            exp.append(ol.o_dist(kind, cur, ref.reshape(-1, stride), w, h, ss, org_off=j.orgOff, cur_off=j.curOff))
        d_jobs = ctx.to_device(np.frombuffer(jobs, np.uint8))
        d_out = ctx.alloc(8 * n)
        ctx.dist_uniform_batch(d_org.ptr, d_ref.ptr, d_jobs.ptr, n, kind, w, h, ss, d_out.ptr)
        assert list(d_out.to_host(np.uint64)) == exp, (kind, w, h)


@pytest.mark.parametrize("rng_kind", ["10bit", "12bit", "signed"])
def test_satd8_grid_sample_ranges(ctx, rng_kind):
    """The grid kernel picks its arithmetic per workgroup from the staged samples: five packed 16-bit butterfly levels for [0, 1023],
    three for [0, 4095], 32-bit otherwise -- all must equal the oracle."""
    L = ol.oracle()
    W, H, r = 128, 64, 4
    rng = np.random.default_rng(len(rng_kind))
    lo, hi = {"10bit": (0, 1024), "12bit": (0, 4096), "signed": (-20000, 20000)}[rng_kind]
    org = ol.i16(rng.integers(lo, hi, (H, W)))
    refp = ol.i16(rng.integers(lo, hi, (H + 2 * r, W + 2 * r + 8)))
    if rng_kind == "12bit":
        org[0:8, 0:8] = 4095   # extreme block: 8 * 4095 after three levels
        refp[r:r + 8, r:r + 8] = 0
    rs = refp.shape[1]
    nb = (W // 8) * (H // 8)
    exp = np.zeros(nb * 81, np.uint64)
    L.vo_satd8_grid(ol.P(org), W, C.c_void_p(refp.ctypes.data + 2 * (r * rs + r)), rs, W, H, r, ol.P(exp))
    d_org, d_ref = ctx.to_device(org), ctx.to_device(refp)
    d_out = ctx.alloc(4 * nb * 81)
    ctx.satd8_grid(d_org.ptr, W, d_ref.ptr + 2 * (r * rs + r), rs, W, H, r, d_out.ptr)
    assert np.array_equal(d_out.to_host(np.uint32).astype(np.uint64), exp)
