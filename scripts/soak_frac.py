"""GPU soak: uniform batches of the whole xMotionEstimation (the tiled fractional-search kernels and the uniform integer / exhaustive searches, square and rectangular
shapes, uni and bi) against the oracle on job sets other than the test-suite seeds."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import me_util                      # noqa: E402
import oracle_lib as ol             # noqa: E402
import test_gpu_mest as TS          # noqa: E402
from vtm_amd.device import Context  # noqa: E402


def main():
    ctx = Context(0)
    L = ol.oracle()
    bad = tot = 0
    cfgv = (4, 1, 1, 0, 1)
    cfg = ol.MestCfg(*cfgv)
    for seed in (4000, 4001):
        scene = me_util.Scene(416, 240, hard=True)
        for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (16, 8), (32, 16), (64, 32), (8, 16), (64, 16)):
            for bi, opts in ((0, {}), (1, dict(no_uni_mv_list=1, pattern_given=1))):
                jobs = me_util.random_mest_jobs(scene, 120 if w * h <= 1024 else 50, seed=seed + w * 7 + h + bi, sizes=([w], [h]))
                for j in jobs:
                    j["imv"], j["bi"] = 0, bi
                    j["cands"] = [[me_util._round_amvr(v, 0) for v in c] for c in j["cands"]]
                    j["mvPred"] = tuple(j["cands"][j["mvpIdx"]])
                    if opts.get("no_uni_mv_list"):
                        j["extra"] = []
                exp = []
                for j in jobs:
                    keep = []
                    t = me_util.oracle_mest_job(scene, j, keep)
                    r = ol.MestResult()
                    L.vo_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
                    exp.append(r.key())
                got, _ = TS.run_device(ctx, scene, jobs, cfgv, uniform_imv=0, uniform_square=1, max_wh=(w, h), uniform_bi=1 + bi, **opts)
                for k, (g, e) in enumerate(zip(got, exp)):
                    tot += 1
                    if g != e:
                        bad += 1
                        print("MISMATCH", seed, (w, h), bi, jobs[k], g, e, flush=True)
        print("seed", seed, "done:", tot, "jobs,", bad, "mismatches", flush=True)
    print("soak: jobs", tot, "mismatches", bad)


if __name__ == "__main__":
    main()
