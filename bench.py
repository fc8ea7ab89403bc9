#!/usr/bin/env python3
"""bench.py -- hot-path throughput on MI355X.

One STEP = one pass of the implemented hot-path stages over ONE 3840x2160 10-bit inter picture of the
random-access configuration (synthetic YUV, SURVEY.md 8d generator):
  stage "tz"   integer TZ search (InterSearch::xTZSearch) for every square PU of the quadtree levels 128..8
               against 2 reference pictures (L0/L1 at |dPOC| = 2 -> ASR search range 96), SAD with the FEN
               sub-sampling rule; level L+1 starts from / predicts with the level-L vector of the enclosing block.
Inputs (original picture, border-extended reference planes, job tables) are resident in HBM before the timed region.
With --gpus N each rank owns the 17 CTU rows of its own picture (N pictures in flight, weak scaling); the reference
planes are re-broadcast from rank 0 over RCCL inside every step (the reconstructed-picture exchange of SURVEY.md 8e).

Prints ONE JSON line (rank 0).  `value` = pictures/s of the stages listed in config.workload -- NOT a full encode.
Extra keys: satd_gblocks_per_s (SURVEY.md 8d SATD-8x8 grid micro-benchmark, 81 displacements), roofline, cpu_baseline.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(fme, cur_np, ref_np, refs, W, H, budget_s):
    """Times the SAME TZ jobs (with the predictors the device run used) on the host, on a bounded sample.
    kind "reference": the real VTM 9.3 xTZSearch (x86 SIMD distFunc) from oracle/_ref/libvtmref.so when it travelled
    with the repo; otherwise kind "port": the plain-C oracle.  Also cross-checks the sampled results against the GPU's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from vtm_amd.pipeline import RES_DT, TZ_DT
    use_ref = ol.have_ref()
    try:
        L = ol.ref() if use_ref else ol.oracle()
    except OSError:
        use_ref, L = False, ol.oracle()
    fn = L.ref_tz_search if use_ref else L.vo_tz_search
    jobs = np.concatenate([lvl["jobs"].cpu().numpy().view(TZ_DT).reshape(-1) for lvl in fme.levels])
    res = np.concatenate([lvl["res"].cpu().numpy().view(RES_DT).reshape(-1) for lvl in fme.levels])
    n = jobs.size
    rng = np.random.default_rng(7)
    order = rng.permutation(n)
    done, mism, t0 = 0, 0, time.perf_counter()
    cur_base, ref_base = cur_np.ctypes.data, ref_np.ctypes.data
    for k in order:
        j = jobs[k]
        c = ol.MeCtx()
        c.org, c.orgStride = cur_base + 2 * int(j["orgOff"]), int(j["orgStride"])
        c.ref, c.refStride = ref_base + 2 * int(j["refOff"]), int(j["refStride"])
        c.w, c.h, c.subShift, c.bitDepth, c.imvShift = int(j["width"]), int(j["height"]), int(j["subShift"]), 10, 0
        c.mv = ol.MvCost(float(j["motionLambda"]), int(j["predHor"]), int(j["predVer"]), 2)
        c.picW, c.picH, c.puX, c.puY, c.ctuSize = W, H, int(j["puX"]), int(j["puY"]), 128
        t = ol.TzJob()
        t.mvHor, t.mvVer, t.searchRange, t.firstSearchStop = int(j["mvHor"]), int(j["mvVer"]), int(j["searchRange"]), 1
        r = ol.MeResult()
        fn(C.byref(c), C.byref(t), C.byref(r))
        g = res[k]
        if (r.mvX, r.mvY, r.cost, r.dist) != (int(g["mvX"]), int(g["mvY"]), int(g["cost"]), int(g["dist"])):
            mism += 1
        done += 1
        if done % 64 == 0 and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt / n, "unit": "pictures/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": "%d of %d TZ-search jobs of the same picture (uniform random sample, %.1f s); %d mismatches vs GPU"
                      % (done, n, dt, mism),
            "jobs_per_s": done / dt, "mismatches_vs_gpu": mism}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from vtm_amd import synth
    from vtm_amd.device import Context
    from vtm_amd.pipeline import FrameME

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 or world > 1:
        assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)
    W, H = a.width, a.height

    # ---- synthetic picture set: cur = frame 2, refs = frames 0 and 4 (|dPOC| = 2) ---------------------------------
    frames = synth.gen_frames(W, H, 5)
    cur_np = np.ascontiguousarray(frames[2])
    planes, refs, off_acc = [], [], 0
    for t in (0, 4):
        buf, off, stride = synth.extend_plane(frames[t], margin=160)
        refs.append((off_acc + off, stride))
        planes.append(buf.reshape(-1))
        off_acc += buf.size
    dpb_np = np.concatenate(planes)
    cur = torch.from_numpy(cur_np).to(dev)
    dpb = torch.from_numpy(dpb_np).to(dev)

    ctx = Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    sr = 96   # Clip3(MinSearchWindow 96, 384, (384*|dPOC| + 8)/16) for |dPOC| = 2 (EncSlice.cpp:1127)
    fme = FrameME(ctx, torch, dev, W, H, W, refs, [sr, sr], motion_lambda=8.0)

    tz_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def step(k=None):
        if world > 1:
            dist.broadcast(dpb.view(torch.uint8), src=0)   # reconstructed reference pictures -> every GPU (xGMI); bytes: int16 is not a collective dtype
        if k is not None:
            tz_ev[k][0].record()
        fme.run(cur.data_ptr(), dpb.data_ptr())
        if k is not None:
            tz_ev[k][1].record()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    evals, alg_bytes = fme.stats()
    tz_ms = sum(s.elapsed_time(e) for s, e in tz_ev) / a.steps   # includes the (tiny) predictor gathers between levels

    # ---- SATD 8x8 grid micro-benchmark (extra key; outside the timed steps) --------------------------------------
    nb = (W // 8) * (H // 8)
    satd_out = torch.empty(nb * 81, dtype=torch.int32, device=dev)
    ref0_ptr = dpb.data_ptr() + 2 * refs[0][0]
    for _ in range(2):
        ctx.satd8_grid(cur.data_ptr(), W, ref0_ptr, refs[0][1], W, H, 4, satd_out.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.satd8_grid(cur.data_ptr(), W, ref0_ptr, refs[0][1], W, H, 4, satd_out.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    satd_ms = e0.elapsed_time(e1) / 10
    satd_g = torch.tensor([nb * 81 / satd_ms / 1e6], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(satd_g)

    if rank == 0:
        out = {
            "metric": "hot-path pictures/sec (3840x2160 randomaccess QP32; stages: integer TZ-search SAD; not a full encode) + SATD Gblocks/s",
            "value": world * a.steps / dt, "unit": "pictures/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 samples, int32 accumulation (fp64 MV-rate multiply)", "data": "synthetic",
            "config": {"workload": "%dx%d 10-bit, encoder_randomaccess_vtm.cfg operating point (QP32, SR 96 via ASR, FEN subsampling): "
                                   "TZ search of quadtree PUs 128..8 x 2 refs = %d searches/picture" % (W, H, fme.n_jobs),
                       "stages": ["tz_search"], "pictures_in_flight": world, "parallelism": "ctu-rows: 1 picture (17 CTU rows) per GPU"},
            "satd_gblocks_per_s": float(satd_g.item()),
            "tz_candidates_per_picture": evals,
            "roofline": {"bound": "hbm", "kernel": "tz_search_kernel", "achieved": alg_bytes / tz_ms / 1e6, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg_bytes / tz_ms / 1e6 / 8000.0, "traffic": None,
                         "note": "algorithmic bytes = sum over searches of nEval * (4*W*H >> subShift); %d launches/step, %.3f ms/step"
                                 % (len(fme.levels), tz_ms)},
            "satd_roofline": {"bound": "hbm", "kernel": "satd8_grid_kernel", "achieved": nb * 81 * 256 / satd_ms / 1e6, "peak": 8000.0,
                              "unit": "GB/s", "frac": nb * 81 * 256 / satd_ms / 1e6 / 8000.0, "ms": satd_ms},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fme, cur_np, dpb_np, refs, W, H, a.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
