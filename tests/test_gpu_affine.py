"""GPU parity: affine motion estimation -- vtmhip_xPredAffineBlk_batch_dev (xPredAffineBlk incl. PROF) and vtmhip_xAffineMotionEstimation_batch_dev
(the whole InterSearch::xAffineMotionEstimation: gradient iterations with the fp64 solver on the device, control-point refinement) vs the oracle
(itself pinned against the real members, tests/test_oracle_vs_ref.py) and vs golden vectors recorded from the real members.  Bit-exact."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import me_util
import oracle_lib as ol
from vtm_amd.lib import AffineMeJob, AffineMeOut, PicParams

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hip_jobs(scene, jobs, others, hevc):
    arr = (AffineMeJob * len(jobs))()
    off = poff = 0
    for k, j in enumerate(jobs):
        t = arr[k]
        t.orgOff, t.orgStride = j["y"] * scene.W + j["x"], scene.W
        t.refOff, t.refStride = scene.ref_off + j["y"] * scene.ref_stride + j["x"], scene.ref_stride
        if j["bi"]:
            o = me_util.other_pred(scene, j)
            others[off:off + o.size] = o.reshape(-1)
            t.otherPredOff, t.otherPredStride = off, j["w"]
            off += o.size
        t.predOff, t.predStride = poff, j["w"]
        poff += j["w"] * j["h"]
        t.puX, t.puY, t.width, t.height = j["x"], j["y"], j["w"], j["h"]
        t.sixParam, t.interDir, t.imv, t.bi, t.useSatd, t.useAffineType = j["six"], j["inter_dir"], j["imv"], j["bi"], j["satd"], j["affine_type"]
        t.amvrEncOpt, t.lowDelayRounds, t.profAllowed, t.profNeedsLargeGrad, t.profIsBi = j["enc_opt"], j["low_delay"], j["prof"], j["prof_large"], j["prof_bi"]
        for i in range(3):
            t.mvPred[i][0], t.mvPred[i][1] = j["pred"][i]
            t.mv[i][0], t.mv[i][1] = j["mv"][i]
        t.bits, t.motionLambda, t.hevcCost = j["bits"], j["lam"], hevc[k]
    return arr, poff


def oracle_results(scene, jobs, L):
    exp, hevc, preds = [], [], []
    for j in jobs:
        keep = []
        t = me_util.affine_me_struct(scene, j, keep)
        t.hevcCost = 1 << 62
        r0 = ol.AffineMeResult()
        L.vo_affine_motion_estimation(C.byref(t), C.byref(r0))
        t.hevcCost = int(r0.cost * j["hevc_scale"])
        r = ol.AffineMeResult()
        L.vo_affine_motion_estimation(C.byref(t), C.byref(r))
        hevc.append(t.hevcCost)
        exp.append(([tuple(v) for v in r.mv][:3 if j["six"] else 2], r.bits, r.cost, r.iterations, r.refinements))
        p = me_util.affine_pred_struct(scene, j)
        mv = ((C.c_int * 2) * 3)(*[(C.c_int * 2)(*v) for v in j["mv"]])
        a = np.zeros((j["h"], j["w"]), np.int16)
        L.vo_pred_affine_blk(C.byref(p), mv, 0, ol.P(a), j["w"])
        preds.append(a)
    return exp, hevc, preds


def run_device(ctx, scene, jobs, hevc):
    others = np.zeros(max(1, sum(j["w"] * j["h"] for j in jobs if j["bi"])), np.int16)
    arr, npred = hip_jobs(scene, jobs, others, hevc)
    pic = PicParams(scene.W, scene.H, 128, 10, 0)
    d_cur, d_ref, d_oth = ctx.to_device(scene.cur), ctx.to_device(scene.ref_buf), ctx.to_device(others)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_res = ctx.alloc(C.sizeof(AffineMeOut) * len(jobs))
    d_pred = ctx.alloc(2 * npred)
    ctx.pred_affine_blk_batch(pic, d_ref.ptr, d_pred.ptr, d_jobs.ptr, len(jobs), 128, 128)
    ctx.affine_motion_estimation_batch(pic, d_cur.ptr, d_ref.ptr, d_oth.ptr, d_jobs.ptr, len(jobs), 128, 128, d_res.ptr)
    res = (AffineMeOut * len(jobs)).from_buffer_copy(d_res.to_host(np.uint8).tobytes())
    pred = d_pred.to_host(np.int16)
    got = [([tuple(v) for v in r.mv][:3 if j["six"] else 2], r.bits, r.cost, r.iterations, r.refinements) for r, j in zip(res, jobs)]
    preds, off = [], 0
    for j in jobs:
        preds.append(pred[off:off + j["w"] * j["h"]].reshape(j["h"], j["w"]))
        off += j["w"] * j["h"]
    return got, preds


@pytest.mark.parametrize("hard", [False, True])
def test_affine_me_matches_oracle(ctx, hard):
    L = ol.oracle()
    scene = me_util.Scene(416, 240, hard=hard)
    jobs = me_util.random_affine_jobs(scene, 300, seed=41 + hard)
    exp, hevc, exp_pred = oracle_results(scene, jobs, L)
    got, preds = run_device(ctx, scene, jobs, hevc)
    for k, j in enumerate(jobs):
        assert np.array_equal(preds[k], exp_pred[k]), ("xPredAffineBlk", k, j)
        assert got[k] == exp[k], ("xAffineMotionEstimation", k, j, got[k], exp[k])
    assert sum(e[3] for e in exp) > 300 and sum(e[4] for e in exp) > 2000


def test_affine_me_matches_golden_from_reference(ctx):
    z = np.load(os.path.join(G, "affine_me.npz"))
    scene = me_util.Scene(416, 240, hard=False)
    jobs = [json.loads(str(s)) for s in z["jobs"]]
    hevc = [int(v) for v in z["hevc"]]
    got, preds = run_device(ctx, scene, jobs, hevc)
    off = 0
    for k, j in enumerate(jobs):
        n = 3 if j["six"] else 2
        assert [list(v) for v in got[k][0]] == z["mv"][k][:n].tolist() and got[k][1] == int(z["bits"][k]) and got[k][2] == int(z["cost"][k]), ("golden ME", k, j)
        assert np.array_equal(preds[k].reshape(-1), z["pred"][off:off + j["w"] * j["h"]]), ("golden prediction", k)
        off += j["w"] * j["h"]
