"""GPU parity: vtmhip_xMotionEstimation_batch_dev (whole InterSearch::xMotionEstimation per job) vs the oracle composition and vs
the golden vectors recorded from the real member function.  Bit-exact: vectors, predictor choice, bits, cost."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import me_util
import oracle_lib as ol
from vtm_amd.lib import MeCfg, MeJob, MeOut, PicParams

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hip_jobs(scene, jobs, others):
    """others: flat int16 array holding every bi job's other-list prediction; returns the ctypes job array."""
    arr = (MeJob * len(jobs))()
    off = 0
    for k, j in enumerate(jobs):
        t = arr[k]
        t.orgOff = j["y"] * scene.W + j["x"]
        t.refOff = scene.ref_off + j["y"] * scene.ref_stride + j["x"]
        t.orgStride, t.refStride = scene.W, scene.ref_stride
        if j["bi"]:
            o = me_util.other_pred(scene, j)
            others[off:off + o.size] = o.reshape(-1)
            t.otherPredOff, t.otherPredStride = off, j["w"]
            off += o.size
        t.puX, t.puY, t.width, t.height = j["x"], j["y"], j["w"], j["h"]
        t.bi, t.imv, t.mvpIdx, t.numAmvpCand = j["bi"], j["imv"], j["mvpIdx"], j["numCand"]
        t.mvPredHor, t.mvPredVer = j["mvPred"]
        t.mvHor, t.mvVer = j["mv"]
        for i in range(2):
            t.amvpCand[i][0], t.amvpCand[i][1] = j["cands"][i]
            t.mvpIdxBits[i] = j["idxBits"][i]
        t.bits, t.searchRange, t.motionLambda = j["bits"], j["searchRange"], j["lam"]
        t.numExtraStart = len(j["extra"])
        for i, (a, b) in enumerate(j["extra"]):
            t.extraStart[i][0], t.extraStart[i][1] = a, b
        t.flags = ((j.get("bcw", 0) & 0xff) << 8) | (1 if j.get("cached") else 0)      # VTMHIP_MEJ_BCW_FLAGS: the searched list's CU-level BCW weight of a bi job; VTMHIP_MEJ_CACHED_INT_MV
    return arr


def run_device(ctx, scene, jobs, cfgv, uniform_imv=-1, uniform_square=0, max_wh=(128, 128), uniform_bi=0, no_uni_mv_list=0, pattern_given=0):
    others = np.zeros(max(1, sum(j["w"] * j["h"] for j in jobs if j["bi"])), np.int16)
    arr = hip_jobs(scene, jobs, others)
    if pattern_given:   # the caller hands over 2*org - otherPred itself (what a fused motion-compensation epilogue writes)
        for k, j in enumerate(jobs):
            o, n = arr[k].otherPredOff, j["w"] * j["h"]
            org = scene.cur[j["y"]:j["y"] + j["h"], j["x"]:j["x"] + j["w"]].astype(np.int32)
            others[o:o + n] = (2 * org - others[o:o + n].reshape(j["h"], j["w"])).astype(np.int16).reshape(-1)
    cfg = MeCfg(cfgv[0], cfgv[1], cfgv[2], cfgv[3], cfgv[4], uniform_imv, uniform_square, uniform_bi, no_uni_mv_list, pattern_given)
    pic = PicParams(scene.W, scene.H, 128, 10, 0)
    d_cur, d_ref, d_oth = ctx.to_device(scene.cur), ctx.to_device(scene.ref_buf), ctx.to_device(others)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_res = ctx.alloc(C.sizeof(MeOut) * len(jobs))
    ctx.motion_estimation_batch(pic, cfg, d_cur.ptr, d_ref.ptr, d_oth.ptr, d_jobs.ptr, len(jobs), max_wh[0], max_wh[1], d_res.ptr)
    res = (MeOut * len(jobs)).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    return [(r.mvHor, r.mvVer, r.mvPredHor, r.mvPredVer, r.mvpIdx, r.bits, r.cost) for r in res], res


@pytest.mark.parametrize("weight,uniform", [(-2, 0), (3, 0), (5, 0), (10, 0), (-2, 1), (3, 1)])
def test_bi_rows_under_a_bcw_weight(ctx, weight, uniform):
    """Bi rows under a CU-level BCW weight (flags bits 8..15): the weighted target of removeWeightHighFreq, the distortion weight |w| / 8 -- a mixed batch (pattern copies made
    by the library) and uniform 16x16 batches through the fused exhaustive / fractional / AMVR kernels (the -2 weight leaves the packed 16-bit Hadamard range: wideOrg)."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    sizes = ([16], [16]) if uniform else None
    jobs = [j for j in me_util.random_mest_jobs(scene, 300, seed=950 + weight + uniform, **({"sizes": sizes} if sizes else {})) if j["bi"]]
    cfgv = (4, 1, 1, 0, 1)
    cfg = ol.MestCfg(*cfgv)
    for imv in ((0, 1, 2, 3) if uniform else (None,)):
        js = [dict(j) for j in jobs]
        for j in js:
            j["bcw"] = weight
            if imv is not None:
                j["imv"] = imv
                j["cands"] = [[me_util._round_amvr(v, imv) for v in c] for c in j["cands"]]
                j["mvPred"] = tuple(j["cands"][j["mvpIdx"]])
        exp = []
        for j in js:
            keep = []
            t = me_util.oracle_mest_job(scene, j, keep)
            r = ol.MestResult()
            L.vo_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
            exp.append(r.key())
        if uniform:
            got, _ = run_device(ctx, scene, js, cfgv, uniform_imv=imv, uniform_square=1, max_wh=(16, 16), uniform_bi=2, pattern_given=0)
        else:
            got, _ = run_device(ctx, scene, js, cfgv)
        assert got == exp, (weight, uniform, imv)


def test_matches_golden_from_reference(ctx):
    z = np.load(os.path.join(G, "mest.npz"))
    scene = me_util.Scene(416, 240, hard=True)
    jobs = [json.loads(str(s)) for s in z["jobs"]]
    cfgs = [tuple(int(v) for v in c) for c in z["cfg"]]
    exp = [tuple(int(v) for v in e) for e in z["res"]]
    for cfgv in sorted(set(cfgs)):
        idx = [i for i, c in enumerate(cfgs) if c == cfgv]
        got, _ = run_device(ctx, scene, [jobs[i] for i in idx], cfgv)
        for g, i in zip(got, idx):
            assert g == exp[i], (jobs[i], g, exp[i])


@pytest.mark.parametrize("cfgv", [(4, 1, 1, 0, 1), (4, 0, 0, 1, 0), (8, 1, 1, 1, 1)])
def test_matches_oracle_mixed_batch(ctx, cfgv):
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    jobs = me_util.random_mest_jobs(scene, 400, seed=40 + cfgv[0] + cfgv[1])
    cfg = ol.MestCfg(*cfgv)
    exp, exp_int = [], []
    for j in jobs:
        keep = []
        t = me_util.oracle_mest_job(scene, j, keep)
        r = ol.MestResult()
        L.vo_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
        exp.append(r.key())
        exp_int.append((r.intX, r.intY, r.intDist))
    got, res = run_device(ctx, scene, jobs, cfgv)
    for k, (g, e) in enumerate(zip(got, exp)):
        assert g == e, (jobs[k], g, e)
    assert [(r.intX, r.intY, r.intDist) for r in res] == exp_int


@pytest.mark.parametrize("size,imv", [(8, 0), (16, 0), (32, 3), (64, 0), (16, 1), (32, 2)])
def test_uniform_batches_use_fast_paths(ctx, size, imv):
    """uniformSquare + uniformImv: tiled fractional kernel / skipped stages; same results as the oracle."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    jobs = me_util.random_mest_jobs(scene, 200, seed=70 + size + imv, sizes=([size], [size]))
    for j in jobs:
        j["imv"] = imv
        j["cands"] = [[me_util._round_amvr(v, imv) for v in c] for c in j["cands"]]
        j["mvPred"] = tuple(j["cands"][j["mvpIdx"]])
    cfgv = (4, 1, 1, 0, 1)
    cfg = ol.MestCfg(*cfgv)
    exp = []
    for j in jobs:
        keep = []
        t = me_util.oracle_mest_job(scene, j, keep)
        r = ol.MestResult()
        L.vo_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
        exp.append(r.key())
    got, _ = run_device(ctx, scene, jobs, cfgv, uniform_imv=imv, uniform_square=1, max_wh=(size, size))
    assert got == exp


@pytest.mark.parametrize("size,bi,opts", [(8, 0, {}), (32, 0, {}), (16, 1, {}), (16, 1, dict(no_uni_mv_list=1)), (64, 1, dict(no_uni_mv_list=1, pattern_given=1)),
                                          (8, 1, dict(pattern_given=1)), (128, 0, {}), (128, 1, dict(no_uni_mv_list=1, pattern_given=1))])
def test_uniform_bi_fast_paths(ctx, size, bi, opts):
    """uniformBi 1 (all uni: the searches read the original plane, no pattern copies) / 2 (all bi: no TZ stage, lane-per-candidate exhaustive
    kernel; optionally no start-candidate SADs and a caller-made 2*org - pred pattern): same results as the oracle's xMotionEstimation."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    jobs = me_util.random_mest_jobs(scene, 160, seed=300 + size + bi, sizes=([size], [size]))
    for j in jobs:
        j["imv"], j["bi"] = 0, bi
        j["cands"] = [[me_util._round_amvr(v, 0) for v in c] for c in j["cands"]]
        j["mvPred"] = tuple(j["cands"][j["mvpIdx"]])
        if opts.get("no_uni_mv_list"):
            j["extra"] = []
    cfgv = (4, 1, 1, 0, 1)
    cfg = ol.MestCfg(*cfgv)
    exp = []
    for j in jobs:
        keep = []
        t = me_util.oracle_mest_job(scene, j, keep)
        r = ol.MestResult()
        L.vo_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
        exp.append(r.key())
    got, _ = run_device(ctx, scene, jobs, cfgv, uniform_imv=0, uniform_square=1, max_wh=(size, size), uniform_bi=1 + bi, **opts)
    assert got == exp


@pytest.mark.parametrize("w,h", [(8, 8), (16, 16), (16, 8), (8, 16), (8, 32), (32, 16), (16, 32), (32, 8)])
def test_four_searches_per_wave_integer_kernel(ctx, w, h):
    """tz_group_kernel (uniform all-uni batches of blocks up to 32 segments: a DPP row per search, a lane per candidate) on what its state machine has to get right:
    0 .. 15 m_uniMvList entries with duplicates (the 15th in a round of its own, up to 4 start candidates beside the speculative distance-1 / -2 points, more without them),
    cached integer vectors (xTZSearch's fast settings: no zero candidate, halved range, raster distance 8, the star loop's early stop), search ranges 1 .. 192 (empty and
    one-round loops, raster scans listed for the column kernel), predictors far from the true motion (two-point steps and star refinements)."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    jobs = me_util.random_mest_jobs(scene, 336, seed=7000 + 64 * w + h, sizes=([w], [h]))
    rng = np.random.default_rng(9 + w + h)
    for k, j in enumerate(jobs):
        j["imv"], j["bi"] = 0, 0
        j["cands"] = [[me_util._round_amvr(v, 0) for v in c] for c in j["cands"]]
        j["mvPred"] = tuple(j["cands"][j["mvpIdx"]])
        j["searchRange"] = [1, 2, 4, 8, 64, 96, 192][k % 7]
        n = k % 16
        extra = [(int(rng.integers(-12 * 16, 12 * 16)), int(rng.integers(-12 * 16, 12 * 16))) for _ in range(n)]
        if n >= 3 and k % 2:
            extra[n - 1] = extra[0]                    # a duplicate at the end (with 15 entries: no second round)
        if n >= 4 and k % 3 == 0:
            extra[2] = extra[1]
        j["extra"] = extra
        j["cached"] = int(k % 5 == 0)
        if k % 4 == 0:                                 # a predictor next to the zero vector: the zero candidate wins or ties the start round
            j["cands"] = [[0, 0], [16, 0]]
            j["mvPred"] = tuple(j["cands"][j["mvpIdx"]])
    cfgv = (4, 1, 1, 0, 1)
    cfg = ol.MestCfg(*cfgv)
    exp = []
    for j in jobs:
        keep = []
        t = me_util.oracle_mest_job(scene, j, keep)
        r = ol.MestResult()
        L.vo_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
        exp.append(r.key())
    got, _ = run_device(ctx, scene, jobs, cfgv, uniform_imv=0, uniform_square=1, max_wh=(w, h), uniform_bi=1)
    bad = [k for k in range(len(jobs)) if got[k] != exp[k]]
    assert not bad, (bad[:10], [(got[k], exp[k], jobs[k]["searchRange"], len(jobs[k]["extra"]), jobs[k]["cached"]) for k in bad[:3]])


def test_integer_search_in_one_launch(ctx):
    """VTMHIP_TZ_SPLIT=0 (read once per process: a child process): the raster scans run inside the search kernels instead of tz_raster_cols_kernel between two launches --
    tz_group_kernel's lane-strided scan and tz_search_kernel's -- with the same results."""
    if os.environ.get("VTMHIP_TEST_CHILD"):
        pytest.skip("the child itself")
    import subprocess
    import sys
    env = dict(os.environ, VTMHIP_TZ_SPLIT="0", VTMHIP_TEST_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "four_searches_per_wave or uniform_bi_fast_paths"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]


@pytest.mark.parametrize("w,h", [(16, 8), (8, 16), (32, 8), (8, 32), (32, 16), (16, 32), (64, 16), (16, 64), (64, 32), (32, 64)])
@pytest.mark.parametrize("bi,opts", [(0, {}), (1, {}), (1, dict(no_uni_mv_list=1, pattern_given=1))])
def test_uniform_rect_fast_paths(ctx, w, h, bi, opts):
    """The uniform fast paths on the binary / ternary split shapes: all-uni batches (TZ on the original plane + the tiled
    fractional kernel with 16x8 / 8x16 Hadamard tiles) and all-bi batches (lane-per-candidate exhaustive kernel on W x H) = the oracle's
    xMotionEstimation."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    jobs = me_util.random_mest_jobs(scene, 96, seed=500 + 64 * w + h + bi, sizes=([w], [h]))
    for j in jobs:
        j["imv"], j["bi"] = 0, bi
        j["cands"] = [[me_util._round_amvr(v, 0) for v in c] for c in j["cands"]]
        j["mvPred"] = tuple(j["cands"][j["mvpIdx"]])
        if opts.get("no_uni_mv_list"):
            j["extra"] = []
    cfgv = (4, 1, 1, 0, 1)
    cfg = ol.MestCfg(*cfgv)
    exp = []
    for j in jobs:
        keep = []
        t = me_util.oracle_mest_job(scene, j, keep)
        r = ol.MestResult()
        L.vo_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
        exp.append(r.key())
    got, _ = run_device(ctx, scene, jobs, cfgv, uniform_imv=0, uniform_square=1, max_wh=(w, h), uniform_bi=1 + bi, **opts)
    assert got == exp


@pytest.mark.parametrize("bi", [0, 1])
def test_mixed_shape_batch_is_bucketed_by_shape(ctx, bi):
    """One call with blocks of every shape (squares, split rectangles, odd sizes) and no uniform-size promise: the library buckets the job table by shape
    on the device and runs the uniform chains per class (the tiled fractional kernel needs uniformImv 0); results come back at the jobs' own indices."""
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=True)
    ws = [8, 16, 32, 64, 16, 8, 32, 8, 32, 16, 64, 16, 64, 32, 24, 4, 128]
    hs = [8, 16, 32, 64, 8, 16, 8, 32, 16, 32, 16, 64, 32, 64, 16, 8, 128]
    rng = np.random.default_rng(77 + bi)
    jobs = []
    for k in range(340):
        i = int(rng.integers(0, len(ws)))
        jobs += me_util.random_mest_jobs(scene, 1, seed=9000 + 10 * k + bi, sizes=([ws[i]], [hs[i]]))
    for j in jobs:
        j["imv"], j["bi"] = 0, bi
        j["cands"] = [[me_util._round_amvr(v, 0) for v in c] for c in j["cands"]]
        j["mvPred"] = tuple(j["cands"][j["mvpIdx"]])
    assert len({(j["w"], j["h"]) for j in jobs}) >= 15
    cfgv = (4, 1, 1, 0, 1)
    cfg = ol.MestCfg(*cfgv)
    exp = []
    for j in jobs:
        keep = []
        t = me_util.oracle_mest_job(scene, j, keep)
        r = ol.MestResult()
        L.vo_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
        exp.append(r.key())
    got, _ = run_device(ctx, scene, jobs, cfgv, uniform_imv=0, uniform_square=0, max_wh=(128, 128), uniform_bi=1 + bi)
    bad = [k for k in range(len(jobs)) if got[k] != exp[k]]
    assert not bad, [(jobs[k]["w"], jobs[k]["h"], got[k], exp[k]) for k in bad[:5]]
