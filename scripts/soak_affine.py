"""GPU soak: xPredAffineBlk and the whole xAffineMotionEstimation against the oracle on job sets other than the test-suite seeds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                  # noqa: E402
import me_util                      # noqa: E402
import oracle_lib as ol             # noqa: E402
import test_gpu_affine as TA        # noqa: E402
from vtm_amd.device import Context  # noqa: E402


def main():
    ctx = Context(0)
    L = ol.oracle()
    bad = tot = 0
    for seed in range(3000, 3004):
        for hard in (False, True):
            scene = me_util.Scene(416, 240, hard=hard)
            jobs = me_util.random_affine_jobs(scene, 180, seed=seed)
            exp, hevc, exp_pred = TA.oracle_results(scene, jobs, L)
            got, preds = TA.run_device(ctx, scene, jobs, hevc)
            for k, j in enumerate(jobs):
                tot += 1
                if got[k] != exp[k] or not np.array_equal(preds[k], exp_pred[k]):
                    bad += 1
                    print("AFFINE MISMATCH", seed, hard, j, got[k], exp[k], flush=True)
        print("seed", seed, "done:", tot, "jobs,", bad, "mismatches", flush=True)
    print("soak: jobs", tot, "mismatches", bad)


if __name__ == "__main__":
    main()
