"""GPU parity: the symmetric-MVD search -- vtmhip_xGetSymmetricCost_batch_dev, vtmhip_xSymmetricMotionEstimation_batch_dev,
vtmhip_symmvdCheckBestMvp_batch_dev and the whole block (vtmhip_smvd_batch_dev, op VTMHIP_SMVD_SEARCH) vs the oracle (pinned against the real members,
tests/test_oracle_vs_ref.py) and vs golden vectors recorded from the real members.  Bit-exact."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import me_util
import oracle_lib as ol
from test_oracle_golden import smvd_flat
from vtm_amd.lib import PicParams, SmvdJob

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hip_jobs(scene, jobs):
    arr = (SmvdJob * len(jobs))()
    for k, j in enumerate(jobs):
        t = arr[k]
        t.orgOff, t.orgStride = j["y"] * scene.W + j["x"], scene.W
        pos = scene.ref_off + j["y"] * scene.ref_stride + j["x"]
        t.refOff[0], t.refOff[1] = pos, scene.ref_buf.size + pos            # the two planes live back to back in one device buffer
        t.refStride[0] = t.refStride[1] = scene.ref_stride
        t.puX, t.puY, t.width, t.height = j["x"], j["y"], j["w"], j["h"]
        t.imv, t.useSatd, t.clipBiPred, t.bcwWeightTar = j["imv"], j["satd"], j["clip"], j["bcw"]
        for l in range(2):
            t.numCand[l] = j["num"][l]
            for i in range(2):
                t.cand[l][i][0], t.cand[l][i][1] = j["cands"][l][i]
            t.mvpIdxBits[l] = j["idxBits"][l]
        t.numStart, t.numFixed, t.modeBits, t.motionLambda = len(j["starts"]), j["numFixed"], j["modeBits"], j["lam"]
        for i, v in enumerate(j["starts"]):
            t.starts[i][0], t.starts[i][1] = v
    return arr


def run_op(ctx, scene, d_cur, d_ref, arr, n, op, max_w=128, max_h=128, uniform=False):
    pic = PicParams(scene.W, scene.H, 128, getattr(scene, "bd", 10), 0)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    ctx.smvd_batch(pic, d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, max_w, max_h, op, uniform=uniform)
    return (SmvdJob * n).from_buffer_copy(d_jobs.to_host(np.uint8).tobytes())


def device_member_results(ctx, scene, jobs, size=None):
    """the same three calls as me_util.smvd_member_results, each as one batch (size: all jobs have that shape -> the uniform form of the call)"""
    kw = dict(max_w=size[0], max_h=size[1], uniform=True) if size else {}
    d_cur = ctx.to_device(scene.cur)
    d_ref = ctx.to_device(np.concatenate([scene.ref_buf.reshape(-1), scene.ref_buf2.reshape(-1)]))
    arr = hip_jobs(scene, jobs)
    n = len(jobs)
    for t, j in zip(arr, jobs):
        pc, pt, start = j["cands"][0][0], j["cands"][1][0], j["starts"][0]
        t.mvCur[0], t.mvCur[1] = start
        t.mvTar[0], t.mvTar[1] = pt[0] - (start[0] - pc[0]), pt[1] - (start[1] - pc[1])
        t.predSym[0][0], t.predSym[0][1] = pc
        t.predSym[1][0], t.predSym[1][1] = pt
    c0 = [r.cost for r in run_op(ctx, scene, d_cur, d_ref, arr, n, 0, **kw)]
    for t, j, c in zip(arr, jobs, c0):
        t.cost = c + int(j["lam"] * 6)
    me = [(tuple(r.mvCur), tuple(r.mvTar), r.cost) for r in run_op(ctx, scene, d_cur, d_ref, arr, n, 1, **kw)]
    for t, j, c in zip(arr, jobs, c0):
        t.cost, t.skip = c + int(j["lam"] * 9), j["x"] // 4 & 1
        t.mvpIdxSym[0] = t.mvpIdxSym[1] = 0
    chk = [(tuple(r.predSym[0]), tuple(r.predSym[1]), tuple(r.mvpIdxSym), r.cost) for r in run_op(ctx, scene, d_cur, d_ref, arr, n, 2, **kw)]
    full = [(tuple(r.mvCur), tuple(r.mvTar), tuple(r.predSym[0]), tuple(r.predSym[1]), tuple(r.mvpIdxSym), r.cost) for r in run_op(ctx, scene, d_cur, d_ref, arr, n, 3, **kw)]
    return list(zip(c0, me, chk)), full


@pytest.mark.parametrize("hard", [False, True])
def test_smvd_matches_oracle(ctx, hard):
    L = ol.oracle()
    scene = me_util.SmvdScene(416, 240, hard=hard)
    jobs = me_util.random_smvd_jobs(scene, 400, seed=77 + hard)
    got, full = device_member_results(ctx, scene, jobs)
    moved = 0
    for k, j in enumerate(jobs):
        exp = me_util.smvd_member_results(scene, j, L, "vo_")
        assert got[k] == exp, ("members", k, j, got[k], exp)
        e = me_util.smvd_search_oracle(scene, j, L)
        assert full[k] == e, ("whole block", k, j, full[k], e)
        moved += e[0] != e[2]
    assert moved > 100, moved


@pytest.mark.parametrize("hard", [False, True])
def test_smvd_uniform_batches(ctx, hard):
    """every size alone as a uniform batch: the lane-per-tile kernel (8x8, 16x8, 8x16, 16x16: up to eight candidates of a PU per pass, 2 .. 8 PUs per wave),
    one wave per PU up to 32x32 and the four-wave form above -- all four ops"""
    L = ol.oracle()
    scene = me_util.SmvdScene(416, 240, hard=hard)
    for size in me_util.SMVD_SIZES:
        tile = max(size) <= 16 and min(size) >= 8
        jobs = me_util.random_smvd_jobs(scene, 150 if tile else 20, seed=size[0] * 131 + size[1] + hard, sizes=[size])
        got, full = device_member_results(ctx, scene, jobs, size)
        for k, j in enumerate(jobs):
            exp = me_util.smvd_member_results(scene, j, L, "vo_")
            assert got[k] == exp, ("members", size, k, j, got[k], exp)
            e = me_util.smvd_search_oracle(scene, j, L)
            assert full[k] == e, ("whole block", size, k, j, full[k], e)


@pytest.mark.parametrize("bd", [8, 12])
def test_smvd_other_bit_depths(ctx, bd):
    """8-bit samples through the tile kernel (no head-room shift in the first filter pass) and 12-bit samples (differences beyond the packed Hadamard's range:
    the block-wide kernel with the 32-bit Hadamard), mixed and uniform batches"""
    L = ol.oracle()
    scene = me_util.SmvdScene(416, 240, hard=True, bit_depth=bd)
    jobs = me_util.random_smvd_jobs(scene, 160, seed=500 + bd)
    got, full = device_member_results(ctx, scene, jobs)
    for k, j in enumerate(jobs):
        assert got[k] == me_util.smvd_member_results(scene, j, L, "vo_"), ("members", bd, k, j)
        assert full[k] == me_util.smvd_search_oracle(scene, j, L), ("whole block", bd, k, j)
    for size in ((8, 8), (16, 16), (32, 16), (64, 64)):
        jobs = me_util.random_smvd_jobs(scene, 60, seed=size[0] + bd, sizes=[size])
        got, full = device_member_results(ctx, scene, jobs, size)
        for k, j in enumerate(jobs):
            assert got[k] == me_util.smvd_member_results(scene, j, L, "vo_"), ("members", bd, size, k, j)
            assert full[k] == me_util.smvd_search_oracle(scene, j, L), ("whole block", bd, size, k, j)


def test_smvd_bcw_weight_minus2(ctx):
    """BCW weight -2 makes the search target -4 org + 5 predA: differences to predB beyond the 4095 the packed 16-bit Hadamard levels can take (10-bit samples:
    up to 6138) -- the tile kernel switches to two packed levels + 32-bit ones.  Hard scene, every tile-kernel shape, unclipped and clipped targets."""
    L = ol.oracle()
    scene = me_util.SmvdScene(416, 240, hard=True)
    for size in ((8, 8), (16, 8), (8, 16), (16, 16), (32, 32), (64, 32), (64, 64), (128, 128)):
        jobs = me_util.random_smvd_jobs(scene, 90, seed=9000 + size[0] * 5 + size[1], sizes=[size])
        for k, j in enumerate(jobs):
            j["bcw"], j["satd"], j["clip"] = -2, 1, int(k % 5 == 0)
        got, full = device_member_results(ctx, scene, jobs, size)
        for k, j in enumerate(jobs):
            assert got[k] == me_util.smvd_member_results(scene, j, L, "vo_"), ("members", size, k, j)
            assert full[k] == me_util.smvd_search_oracle(scene, j, L), ("whole block", size, k, j)


def test_smvd_matches_golden_from_reference(ctx):
    z = np.load(os.path.join(G, "smvd.npz"))
    scene = me_util.SmvdScene(416, 240)
    jobs = [json.loads(str(s)) for s in z["jobs"]]
    got, _ = device_member_results(ctx, scene, jobs)
    for k, j in enumerate(jobs):
        assert smvd_flat(got[k]) == z["out"][k].tolist(), ("golden", k, j)
