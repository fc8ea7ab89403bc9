#!/usr/bin/env python3
"""TZ search micro-benchmark on HARD jobs (random start vectors / predictors on the noisy clip: the star refinement and the raster
scan run often, unlike the well-predicted searches of bench.py).  usage (GPU box): python3 scripts/tz_micro.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import me_util   # noqa: E402
from vtm_amd.device import Context   # noqa: E402
from vtm_amd.lib import PicParams   # noqa: E402

ctx = Context(0)
scene = me_util.Scene(1920, 1080, hard=True)
d_cur, d_ref = ctx.to_device(scene.cur), ctx.to_device(scene.ref_buf)
for size, wpj, n in ((8, 1, 60000), (16, 2, 30000), (32, 4, 12000), (64, 4, 4000)):
    jobs = me_util.random_tz_jobs(scene, n, seed=size, ranges=(96,), allow_ext=False, sizes=([size], [size]))
    for j in jobs:
        j["fast"] = 0
        j["hasInt"] = 0
    arr = me_util.hip_tz_jobs(scene, jobs, scene.W)
    d_jobs = ctx.to_device(np.frombuffer(arr, np.uint8))
    d_res = ctx.alloc(32 * n)
    pic = PicParams(scene.W, scene.H, 128, 10, wpj)
    for _ in range(2):
        ctx.tz_search_batch(pic, d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, d_res.ptr)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.tz_search_batch(pic, d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, d_res.ptr)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    res = d_res.to_host(np.uint8).view(np.uint32).reshape(n, 8)
    print("tz hard %3dx%-3d jobs %6d  %.3f ms  %.1f ns/job  mean nEval %.1f" % (size, size, n, dt * 1e3, dt * 1e9 / n, res[:, 2].mean()))
