#!/usr/bin/env python3
"""bench.py -- hot-path throughput on MI355X.

One STEP = one pass of the implemented hot-path stages over ONE 3840x2160 10-bit inter picture of the
random-access configuration (synthetic YUV, SURVEY.md 8d generator):
  tz         integer TZ search (InterSearch::xTZSearch) for every square PU of the quadtree levels 128..8 against
             2 reference pictures (L0/L1 at |dPOC| = 2 -> ASR search range 96), SAD with the FEN sub-sampling rule;
             level L+1 starts from / predicts with the level-L vector of the enclosing block
  frac       half + quarter sample refinement (xPatternSearchFracDIF, SATD) per (PU, list)
  bi_search  FEN bi-pred iteration: MC of the other list, 2*org - pred, +-4 exhaustive search, fractional refinement
  mc         final prediction (best uni list; bi via two 14-bit MCs + addAvg) and residual
  resi       per TU (<= 64x64) and transform candidate (DCT2 + 4 MTS up to 32x32): xT, Quant::quant, dequant, xIT, SSE
Inputs (original picture, border-extended reference planes, job tables) are resident in HBM before the timed region.
With --gpus N each rank owns the 17 CTU rows of its own picture (N pictures in flight, weak scaling); the reference
planes are re-broadcast from rank 0 over RCCL inside every step (the reconstructed-picture exchange of SURVEY.md 8e).

Prints ONE JSON line (rank 0).  `value` = pictures/s of the stages listed in config.workload -- NOT a full encode.
Extra keys: satd_gblocks_per_s (SURVEY.md 8d SATD-8x8 grid micro-benchmark, 81 displacements), roofline, cpu_baseline.
"""
import argparse
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
    ap.add_argument("--serial", action="store_true", help="stage-major order on one stream (no overlap of the levels' chains): clean per-kernel times for profiling")
    ap.add_argument("--graph", action="store_true", help="capture the picture's launches (fork/join over the side streams) into one hipGraph and replay it per "
                    "step; measured slower than eager multi-stream launches (5.47 vs 5.13 ms per 4K picture), so off by default")
    return ap.parse_args()


def cpu_baseline(hp, cur_np, dpb_np, refs, W, H, lam, qp, budget_s):
    """Runs the SAME chain (tests/cpu_chain.py) for a bounded random sample of PUs of every level on one host core and
    extrapolates to the picture.  kind "reference": the real VTM 9.3 xTZSearch / xPatternSearchFracDIF / xPatternSearch /
    filterHor / filterVer / fastFwdTrans / fastInvTrans / distFunc (x86 SIMD) from oracle/_ref/libvtmref.so when it
    travelled with the repo (quant/dequant: oracle port); otherwise kind "port": the plain-C oracle throughout.
    Every sampled PU is also compared with the GPU result."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_chain
    import oracle_lib as ol
    R = None
    if ol.have_ref():
        try:
            R = ol.ref()
        except OSError:
            R = None
    snaps = cpu_chain.snapshot(hp)
    rng = np.random.default_rng(7)
    total_s, done, mism, detail = 0.0, 0, 0, []
    per_level = budget_s / len(snaps)
    for lvl in snaps:
        s, npu, jobs = lvl["size"], lvl["npu"], lvl["jobs_np"]
        order = rng.permutation(npu)
        t_lvl, n_lvl = 0.0, 0
        for i in order:
            t0 = time.perf_counter()
            out = cpu_chain.run_pu(cur_np, dpb_np.ctypes.data, refs[0][1], W, H, s, int(jobs["puX"][i]), int(jobs["puY"][i]),
                                   (jobs[i], jobs[npu + i]), lam, (qp + 12) // 6, (qp + 12) % 6, ref=R)
            t_lvl += time.perf_counter() - t0
            n_lvl += 1
            try:
                cpu_chain.compare_with_device(lvl, int(i), out)
            except AssertionError:
                mism += 1
            if t_lvl > per_level and n_lvl >= 4:
                break
        total_s += t_lvl / n_lvl * npu
        done += n_lvl
        detail.append("%dx%d:%d" % (s, s, n_lvl))
    return {"value": 1.0 / total_s, "unit": "pictures/s", "cores": 1, "kind": "reference" if R is not None else "port",
            "sample": "%d PUs (%s) of the same picture through the whole chain, per-level time extrapolated to all %d PUs; %d mismatches vs GPU"
                      % (done, " ".join(detail), sum(l["npu"] for l in snaps), mism),
            "seconds_per_picture": total_s, "mismatches_vs_gpu": mism}


def main():
    # The contract is ONE JSON line on stdout.  Libraries print banners there (RCCL writes its version block to stdout when the process
    # group comes up), so fd 1 is pointed at stderr for the whole run and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    a = parse()
    import torch
    import torch.distributed as dist
    from vtm_amd import synth
    from vtm_amd.device import Context
    from vtm_amd.pipeline import FrameHotPath

    if a.serial:
        os.environ["VTM_AMD_OVERLAP"] = "0"
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # VTM_BENCH_FORCE_DIST=1: run the N > 1 code path (process group, reference-plane broadcast) with a single rank -- a functional check
    # of that path on a one-GPU box
    force_dist = world == 1 and a.gpus == 1 and os.environ.get("VTM_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if a.gpus > 1 or world > 1 or force_dist:
        assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    use_dist = world > 1 or force_dist
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
    lam, qp = 8.0, 32
    fme = FrameHotPath(ctx, torch, dev, W, H, W, refs, [sr, sr], motion_lambda=lam, qp=qp)

    # --graph: the launches of one picture are captured ONCE into a hipGraph and replayed per step: same kernels, same dependencies, same
    # buffers -- only the launch path changes (the 43 launches are not launch-bound; eager launches over the side streams are the default).
    graph = None
    if a.graph and not use_dist:   # with RCCL initialised its watchdog thread may touch the runtime during a capture: eager there
        try:
            fme.run(cur.data_ptr(), dpb.data_ptr())          # allocations / lazy initialisation happen outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                fme.run(cur.data_ptr(), dpb.data_ptr())
            graph = g
        except Exception as e:   # capture unsupported on this runtime: eager launches (identical work)
            sys.stderr.write("hipGraph capture failed (%r); launching eagerly\n" % (e,))
            graph = None
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()

    # N > 1: the reconstructed reference planes go from rank 0 to every GPU inside every step (RCCL broadcast over xGMI; bytes view: int16 is
    # not a collective dtype).  Double-buffered: the planes of step k + 1 travel on RCCL's stream while step k computes on the planes that
    # arrived before (the asynchronous collective starts after the launches already queued on the compute stream, i.e. after the step that
    # last read its target buffer); a step's compute waits -- on the stream -- for its own planes.
    from vtm_amd.exchange import PlaneExchange
    xchg = PlaneExchange([dpb, dpb.clone()], src=0) if use_dist else None

    def step(k=None):
        if xchg is not None:
            planes = xchg.next()
            fme.run(cur.data_ptr(), planes.data_ptr())
        elif graph is not None:
            graph.replay()
        else:
            fme.run(cur.data_ptr(), dpb.data_ptr())

    def drain():
        if xchg is not None:
            xchg.drain()

    for _ in range(a.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-stage kernel time: a separate, event-instrumented pass after the timed region (events between the launches of one step)
    stage_acc = {}
    for _ in range(3):
        fme.run(cur.data_ptr(), dpb.data_ptr(), timing=True)
        torch.cuda.synchronize()
        for k2, v in fme.stage_ms().items():
            stage_acc[k2] = stage_acc.get(k2, 0.0) + v / 3
    evals = fme.stats()[0]
    alg = fme.alg_bytes()
    stages = {k2: ({"ms": stage_acc[k2], "alg_GBps": alg[k2] / stage_acc[k2] / 1e6} if k2 in alg else {"ms": stage_acc[k2]}) for k2 in stage_acc}
    dom = max((k2 for k2 in stage_acc if k2 in alg), key=stage_acc.get)
    dom_kernel = {"tz": "tz_search_kernel", "frac": "frac_search_sq_kernel", "full": "full_search_kernel", "mc": "motion_comp_kernel",
                  "tu": "tu_chain"}[dom]
    launches = {"tz": 1, "frac": 2, "full": 1, "mc": 2, "tu": 1}[dom] * len(fme.levels)
    # HBM-side traffic of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null when the file is absent
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic_per_launch_KB.json")))
        tot = sum((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 * v.get("launches_per_step", 1) for k2, v in pmc.items()
                  if k2.startswith(dom_kernel))
        traffic = tot / launches if tot else None
    except Exception:
        pass

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
    if use_dist:
        dist.all_reduce(satd_g)

    if rank == 0:
        out = {
            "metric": "hot-path pictures/sec (3840x2160 randomaccess QP32; stages: TZ integer ME, fractional ME, bi-pred refinement, MC, residual xT/quant/xIT/SSE; not a full encode) + SATD Gblocks/s",
            "value": world * a.steps / dt, "unit": "pictures/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 samples, int32 accumulation (fp64 MV-rate multiply)", "data": "synthetic",
            "config": {"workload": "%dx%d 10-bit, encoder_randomaccess_vtm.cfg operating point (QP32, SR 96 via ASR, FEN subsampling): "
                                   "quadtree PUs 128..8 x 2 refs = %d integer searches + %d fractional + %d bi-pred refinements/picture, %d TU x transform-candidate chains"
                                   % (W, H, fme.n_jobs, fme.n_jobs, fme.n_jobs // 2, sum(l["ntu"] * l["nc"] for l in fme.levels)),
                       "stages": ["tz_search", "frac_search", "bi-pred refinement (mc+removeHighFreq fused, full_search, frac_search)", "final uni/bi prediction + residual (fused)",
                                  "tu_chain (xT, quant, dequant, xIT, SSE)"], "launch": "hipGraph replay" if graph is not None else "eager",
                       "order": "stage-major, one stream" if a.serial else "level-major over 5 side streams (each level's stages start when its integer search is done)",
                       "pictures_in_flight": world, "parallelism": "ctu-rows: 1 picture (17 CTU rows) per GPU"},
            "satd_gblocks_per_s": float(satd_g.item()),
            "tz_candidates_per_picture": evals,
            "stages": stages,
            "roofline": {"bound": "hbm", "kernel": dom_kernel, "stage": dom, "achieved": stages[dom]["alg_GBps"], "peak": 8000.0, "unit": "GB/s",
                         "frac": stages[dom]["alg_GBps"] / 8000.0, "traffic": traffic, "launches_per_step": launches,
                         "achieved_per_launch_bytes": alg[dom] / launches, "ms_per_launch": stage_acc[dom] / launches,
                         "note": "dominant kernel family of the step by time; achieved = algorithmic bytes (DESIGN.md section 5) / kernel time, both per "
                                 "launch averaged over its %d launches/step, timed in the stage-major pass (one stream, no overlap between the levels' chains): "
                                 "%.3f ms of %.3f ms there; traffic = HBM-side bytes per launch from the committed PMC passes (profiles/)"
                                 % (launches, stage_acc[dom], sum(stage_acc.values()))},
            "satd_roofline": {"bound": "hbm", "kernel": "satd8_grid_kernel", "achieved": nb * 81 * 256 / satd_ms / 1e6, "peak": 8000.0,
                              "unit": "GB/s", "frac": nb * 81 * 256 / satd_ms / 1e6 / 8000.0, "ms": satd_ms},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fme, cur_np, dpb_np, refs, W, H, lam, qp, a.cpu_seconds)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
