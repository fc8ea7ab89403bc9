"""GPU soak: the SMVD kernels (every op, mixed and uniform batches, all shapes) against the oracle on job sets other than the test-suite seeds."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import me_util, oracle_lib as ol
from vtm_amd.device import Context
import test_gpu_smvd as T
ctx = Context(0)
L = ol.oracle()
bad = 0; tot = 0
for seed in range(1000, 1006):
    for hard in (False, True):
        scene = me_util.SmvdScene(416, 240, hard=hard)
        for size in [None, (8, 8), (16, 16), (16, 8), (8, 16), (32, 32), (64, 32), (128, 128), (32, 8), (64, 64)]:
            jobs = me_util.random_smvd_jobs(scene, 120 if size is None else 60, seed=seed * 7 + (0 if size is None else size[0] * 3 + size[1]), sizes=None if size is None else [size])
            got, full = T.device_member_results(ctx, scene, jobs, size)
            for k, j in enumerate(jobs):
                tot += 1
                if got[k] != me_util.smvd_member_results(scene, j, L, "vo_") or full[k] != me_util.smvd_search_oracle(scene, j, L):
                    bad += 1
                    print("MISMATCH", seed, hard, size, k, j)
print("soak: jobs", tot, "mismatches", bad)
