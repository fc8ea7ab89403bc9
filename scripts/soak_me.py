"""GPU soak: integer search (TZ: batches small enough for the shared raster scan and large ones, every waves-per-job setting), the whole xMotionEstimation (mixed and
uniform batches, uni / bi) against the oracle on job sets other than the test-suite seeds."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import me_util                      # noqa: E402
import oracle_lib as ol             # noqa: E402
import test_gpu_me as TM            # noqa: E402
import test_gpu_mest as TS          # noqa: E402
from vtm_amd.device import Context  # noqa: E402


def main():
    ctx = Context(0)
    L = ol.oracle()
    bad = tot = 0
    for seed in range(2000, 2004):
        for hard in (True, False):
            scene = me_util.Scene(416, 240, hard=hard)
            for n, sizes, wpj in ((300, None, 0), (1800, None, 0), (200, ([128], [128]), 8), (500, ([64], [64]), 2), (250, ([128], [128]), 16), (400, ([32, 16], [32, 16]), 0),
                                  (150, ([128, 64], [128, 64]), 4)):
                jobs = me_util.random_tz_jobs(scene, n, seed=seed * 13 + n, sizes=sizes)
                exp = me_util.run_oracle_tz(scene, jobs)
                got = TM._run_hip(ctx, scene, jobs, wpj)
                for k in range(n):
                    tot += 1
                    if got[k] != exp[k]:
                        bad += 1
                        print("TZ MISMATCH", seed, hard, n, sizes, wpj, jobs[k], got[k], exp[k], flush=True)
            for cfgv in ((4, 1, 1, 0, 1), (4, 0, 0, 1, 0)):
                jobs = me_util.random_mest_jobs(scene, 250, seed=seed * 3 + cfgv[1])
                cfg = ol.MestCfg(*cfgv)
                exp = []
                for j in jobs:
                    keep = []
                    t = me_util.oracle_mest_job(scene, j, keep)
                    r = ol.MestResult()
                    L.vo_motion_estimation(C.byref(cfg), C.byref(t), C.byref(r))
                    exp.append(r.key())
                got, _ = TS.run_device(ctx, scene, jobs, cfgv)
                for k, (g, e) in enumerate(zip(got, exp)):
                    tot += 1
                    if g != e:
                        bad += 1
                        print("MEST MISMATCH", seed, hard, cfgv, jobs[k], g, e, flush=True)
        print("seed", seed, "done: jobs", tot, "mismatches", bad, flush=True)
    print("soak: jobs", tot, "mismatches", bad)


if __name__ == "__main__":
    main()
