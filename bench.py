#!/usr/bin/env python3
"""bench.py -- hot-path throughput on MI355X.

One STEP = one pass of the hot path over ONE 3840x2160 10-bit inter picture at the operating point of encoder_randomaccess_vtm.cfg
(synthetic YUV, SURVEY.md 8d generator; B slice, 2 + 2 reference pictures at dPOC -2 -4 / +2 +4 -> ASR search range 96, FEN, QP 32; SMVD, Affine and
TransformSkip ON as in the cfg (cfg/encoder_randomaccess_vtm.cfg:71,120,142) -- `--lite` switches the three off: the round-2 default), as the
level-order form of InterSearch::predInterSearch + xEstimateInterResidualQT over the quadtree of square PUs 128..8 (vtm_amd/pipeline.py):
  amvp     xEstimateMvPredAMVP (template cost of the AMVP candidates) per (PU, list, refIdx)
  uni_me   xMotionEstimation per (PU, list, refIdx): xTZSearch (SAD, FEN sub-sampling) + xPatternSearchFracDIF (SATD) + rate re-weighting,
           xCheckBestMVP, best reference picture per list
  bi_me    the list with the larger cost refined for every refIdx against the other list's prediction (MC -> 2*org - pred, +-4 xPatternSearch,
           fractional search), xCheckBestMVP, uni / bi decision
  mc       chosen prediction (uni, or two 14-bit predictions + addAvg) and the residual
  smvd     the symmetric-MVD block of predInterSearch per PU (between the bi refinement and the decision)
  affine   xAffineMotionEstimation (4-parameter, uni) per (PU >= 16x16, list, refIdx)
  tu       per TU (<= 64x64) and transform candidate (DCT2, transform skip, 4 MTS pairs up to 32x32): xT, Quant::quant, dequant, xIT, SSE
Inputs (original picture, border-extended reference planes, job tables) are resident in HBM before the timed region.

--gpus N: the CTUs of the ONE picture are sharded over the N ranks (raster-scan CTU ranges: whole CTU rows plus one row cut at a CTU, so that
510 CTUs split 64 / 63 per rank -- whole rows would give 3 / 2 rows and cap the speed-up at 71 %); the reference planes travel from rank 0 to
every rank inside every step (RCCL broadcast over xGMI, double-buffered) and every rank's result records travel back to rank 0 (gather).
STRONG scaling: `value` = pictures/s of the whole job.

Prints ONE JSON line (rank 0).  `value` = pictures/s of the stages above -- NOT a full encode (the CU recursion, CABAC and DepQuant are host work
the path does not contain).  Extra keys: satd_gblocks_per_s (SURVEY.md 8d SATD-8x8 grid micro-benchmark, 81 displacements), roofline, cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SIMDS, CLOCK_HZ = 1024, 2.4e9          # 256 CUs x 4 SIMDs, peak shader clock (MI355X_MICROARCH.md)
GUIDE_CYCLES = 2.0                     # MI355X_MICROARCH.md: one wave64 VALU instruction per 2 cycles per SIMD (>= 2 waves) -- the machine's full-rate issue peak
MIX_CLOCK_HZ = 2.3e9                   # the clock the issue micro-benchmark actually held on mixed streams (implied_MHz 2 280 .. 2 360, profiles/r02_valu_issue.jsonl)
SALU_CYCLES = 4.03   # measured: cycles per scalar instruction per SIMD (profiles/r02_valu_issue.jsonl, s_add_u32, 1 .. 8 waves)
KERNELS = ("tz_search_kernel", "tz_group_kernel", "tz_raster_cols_kernel", "frac_search_sq_kernel", "full_search_sq_kernel", "full_search_kernel", "motion_comp_kernel", "tu_chain_uni_kernel",
           "dist_uniform_kernel", "tu_ts_kernel", "bdof_kernel", "tu_chain_lane_kernel", "affine_me_kernel", "smvd_tile_kernel", "smvd_kernel")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--config", choices=["ra", "ldp"], default="ra", help="ra: B slices, --refs + --refs reference pictures, ASR search ranges; "
                    "ldp: P slices, 4 list-0 pictures, SearchRange 64 (encoder_lowdelay_P_vtm.cfg)")
    ap.add_argument("--refs", type=int, default=2, help="ra: active reference pictures per list (1 = the round-1 operating point)")
    ap.add_argument("--qp", type=int, default=32, help="slice QP: quantiser of the TU chains and, through the reference's lambda formula, the motion lambda of the searches")
    ap.add_argument("--lambda-motion", type=float, default=0.0, help="override the motion lambda (rounds 1-2 used 8.0 at every QP)")
    ap.add_argument("--lite", action="store_true", help="the lighter tool set of rounds 1-2: no SMVD block, no affine stage, no transform-skip candidate (each can be added back "
                    "with --smvd / --affine / --transform-skip).  Default: all three ON, as encoder_randomaccess_vtm.cfg has them (SMVD:1 Affine:1 TransformSkip:1)")
    ap.add_argument("--dpoc", type=str, default="", help="ra: |dPOC| of the reference pictures of EACH list, nearest first, e.g. 8,16 = the top layers of the RA GOP "
                    "(ASR search ranges 192 / 384); default 2,4,.. (--refs of them: search range 96)")
    ap.add_argument("--transform-skip", action="store_true", help="add the MTS_SKIP candidate to the TU chains")
    ap.add_argument("--shard", choices=["ctu", "row"], default="ctu", help="--gpus N: raster-scan CTU ranges (balanced) or whole CTU rows")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encoder-level", action="store_true", help="skip cpu_baseline.encoder_level (the real reference encoder, plain vs predInterSearch on the device, on a small clip: ~40 s)")
    ap.add_argument("--partition", choices=["qt", "btt"], default="qt", help="qt: the five quadtree levels 128 .. 8 (the headline workload); btt: a binary / ternary "
                    "split mix -- 128x128, 64x64, 64x32, 32x32, 32x16, 16x16, 16x8, 8x8 -- through the rectangular fast paths")
    ap.add_argument("--smvd", action="store_true", help="add the symmetric-MVD block of predInterSearch (one search per PU between the bi refinement and the uni / bi decision; ra only)")
    ap.add_argument("--affine", action="store_true", help="add the affine uni stage: xAffineMotionEstimation (4-parameter) per (PU >= 16x16, list, refIdx) -- BASELINE config 5's tool set")
    ap.add_argument("--luma-only", action="store_true", help="no chroma planes and no BDOF in the final prediction (the round-2 mid-round operating point)")
    ap.add_argument("--graph", choices=["on", "off"], default="off", help="replay the picture's launches from a hipGraph (captured once per reference buffer).  Measured: no gain -- "
                    "9.83 vs 9.89 ms on one GPU, 2.08 vs 2.09 ms for a rank's share of an 8-GPU run: the step is bound by its dependent chain of kernels, not by launches")
    ap.add_argument("--inflight", type=int, default=1, help="pictures in flight: step k + 1 starts on a second stream set while step k's tail still runs (each has its own tables)")
    ap.add_argument("--serial", action="store_true", help="one stream, no overlap of the levels' chains: clean per-kernel times for profiling")
    ap.add_argument("--launch-check", action="store_true", help="every rank prints {rank, world} as one JSON line and exits before any GPU work (the CPU test of the --gpus N self-launch)")
    a = ap.parse_args()
    if not a.lite:
        a.smvd, a.affine, a.transform_skip = True, True, True
    a.dpoc_list = [int(x) for x in a.dpoc.split(",") if x] if a.dpoc else [2 * (k + 1) for k in range(a.refs)]
    assert a.dpoc_list == sorted(a.dpoc_list) and len(set(a.dpoc_list)) == len(a.dpoc_list) and 1 <= len(a.dpoc_list) <= 4 and a.dpoc_list[0] >= 1
    return a


KERNEL_SRC = {"tz_search_kernel": "me.hip", "tz_group_kernel": "me.hip", "tz_raster_cols_kernel": "me.hip", "full_search_sq_kernel": "me.hip", "full_search_kernel": "me.hip", "frac_search_sq_kernel": "interp.hip",
              "motion_comp_kernel": "mc.hip", "bdof_kernel": "mc.hip", "tu_chain_uni_kernel": "transform.hip", "tu_ts_kernel": "transform.hip", "tu_chain_lane_kernel": "transform.hip",
              "dist_uniform_kernel": "dist.hip", "satd8_grid_kernel": "dist.hip", "affine_me_kernel": "affine.hip", "smvd_tile_kernel": "smvd.hip", "smvd_kernel": "smvd.hip"}


def workload_key(a, world):
    """what the PMC instruction counters of a run depend on: the argument vector that shapes the work + the kernel sources (one hash per .hip file and one over the
    headers: a kernel's counters stay valid while ITS file and the headers are unchanged -- the other stages' results are bit-exact by the parity tests, so its job set
    does not depend on how they are implemented).  The committed counters (profiles/pmc_insts_per_launch.json, written from a run of THIS script) carry the key of their
    run; a roofline fraction is only formed when it matches (counters_for)."""
    import hashlib
    d, hdr = os.path.join(ROOT, "vtm_amd", "csrc"), hashlib.sha1()
    src = {}
    for f in sorted(os.listdir(d)):
        b = open(os.path.join(d, f), "rb").read()
        if f.endswith(".hpp"):
            hdr.update(b)
        elif f.endswith(".hip"):
            src[f] = hashlib.sha1(b).hexdigest()
    src["headers"] = hdr.hexdigest()
    args = dict(width=a.width, height=a.height, config=a.config, dpoc=a.dpoc_list if a.config == "ra" else None, qp=a.qp, ts=bool(a.transform_skip), smvd=bool(a.smvd and a.config == "ra"),
                affine=bool(a.affine), partition=a.partition, luma_only=bool(a.luma_only), shard=a.shard if world > 1 else None, world=world, lam=a.lambda_motion,
                sim=int(os.environ.get("VTM_BENCH_SIMULATE_WORLD", "0")), lambda_bit_depth=10)
    return {"args": args, "src_sha1": src}


def counters_for(kernel, meta, wkey, same_args=True):
    """True when the committed counters (their _meta key) were collected with this kernel's source file and the headers as they are now (and, same_args, from this command)."""
    if not meta or "src_sha1" not in meta:
        return False
    f = KERNEL_SRC.get(kernel)
    m, w = meta["src_sha1"], wkey["src_sha1"]
    return f is not None and m.get(f) == w.get(f) and m.get("headers") == w.get("headers") and (not same_args or meta.get("args") == wkey["args"])


def cpu_baseline(hp, cur_np, dpb_np, refs, sr, W, H, lam, qp, budget_s, pocs=None, chroma=None, affine=False, low_delay=False, smvd=None):
    """The SAME chain (tests/cpu_pis.py) for a bounded random sample of PUs of every level on one host core, extrapolated per level to the
    picture.  kind "reference": every step through the real VTM 9.3 members compiled in place (xEstimateMvPredAMVP, xMotionEstimation,
    xCheckBestMVP, filterHor / filterVer, removeHighFreq / addAvg, TrQuant::xT / xIT, distFunc with the x86 SIMD tables; quant / dequant: the
    oracle port) from oracle/_ref/libvtmref.so when it travelled with the repo; otherwise kind "port": the plain-C oracle throughout.
    Every sampled PU is also compared with the GPU result."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_pis
    import oracle_lib as ol
    R = None
    if ol.have_ref():
        try:
            R = ol.ref()
        except OSError:
            R = None
    snaps = hp.snapshot()
    rng = np.random.default_rng(7)
    total_s, done, mism, detail = 0.0, 0, 0, []
    per_level = budget_s / len(snaps)
    for li, lvl in enumerate(snaps):
        s, npu = lvl["size"], lvl["npu"]
        parent = snaps[lvl["parent_level"]] if lvl["parent_level"] >= 0 else None
        t_lvl, n_lvl = 0.0, 0
        for i in rng.permutation(npu):
            i = int(i)
            cands = cpu_pis.cands_of(lvl, hp.nref, i)
            t0 = time.perf_counter()
            out = cpu_pis.run_pu(cur_np, dpb_np.ctypes.data, refs, sr, W, H, s, int(lvl["xs"][i]), int(lvl["ys"][i]), cands, lam, (qp + 12) // 6, (qp + 12) % 6,
                                 lvl["cands"], ref=R, pocs=pocs, chroma=chroma, affine=affine, low_delay=low_delay, smvd=smvd)
            t_lvl += time.perf_counter() - t0
            n_lvl += 1
            try:
                cpu_pis.compare_with_device(lvl, parent, hp.nref, i, out)
            except AssertionError:
                mism += 1
            if t_lvl > per_level and n_lvl >= 4:
                break
        total_s += t_lvl / n_lvl * npu
        done += n_lvl
        detail.append("%dx%d:%d" % (lvl["w"], lvl["h"], n_lvl))
    return {"value": 1.0 / total_s, "unit": "pictures/s", "cores": 1, "kind": "reference" if R is not None else "port",
            "sample": "%d PUs (%s) of the same picture through the whole chain, per-level time extrapolated to all %d PUs; %d mismatches vs GPU"
                      % (done, " ".join(detail), sum(l["npu"] for l in snaps), mism),
            "seconds_per_picture": total_s, "mismatches_vs_gpu": mism}


def encoder_level():
    """BASELINE metric (1), encoded frames/s, on a BOUNDED sample: the real VTM 9.3 encoder (oracle/_ref/libvtmref.so, compiled in place from /root/reference; present only when
    it travelled with the repo) encodes a 192x128 x 5-picture synthetic random-access clip twice in child processes -- plain on one host core, and with
    InterSearch::predInterSearch routed to the device as one vtmhip_predInterSearch_batch_dev call per CU (+ xAffineMotionEstimation; the reference's own glue replayed over the device
    tables: oracle/ref_shim_pis.hpp, test infrastructure).  Reported beside the hot-path metric, never part of `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tempfile
    import enc_dropin
    if not (os.path.exists(enc_dropin.REF_SO) and os.path.exists(enc_dropin.HIP_SO)):
        return None
    W, H, N, qp = 192, 128, 5, 32
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "clip.yuv")
        enc_dropin.write_clip(yuv, W, H, N)
        t = time.time()
        st0, bits0, rec0 = enc_dropin.encode(yuv, W, H, N, qp, os.path.join(tmp, "plain"), False, 2048 | 8, 1, 0, timeout=600)
        plain_s = time.time() - t
        t = time.time()
        st1, bits1, rec1 = enc_dropin.encode(yuv, W, H, N, qp, os.path.join(tmp, "rep"), True, 2048 | 128, 1, 0, env={"VTMREF_REPLACE": "1"}, timeout=600)
        rep_s = time.time() - t
    in_member = sum(st0["pis"]["seconds"])
    return {"metric": "encoded frames/s (BASELINE metric 1), one CU per device call", "clip": "%dx%d synthetic, %d pictures, QP %d, tests/data/enc_ra_gop4.cfg" % (W, H, N, qp),
            "encoded_fps_plain_1_core": N / plain_s, "encoded_fps_predInterSearch_on_device": N / rep_s, "speedup": plain_s / rep_s,
            "identical_bitstream_and_reconstruction": bits0 == bits1 and rec0 == rec1, "predInterSearch_calls": st1["pis"]["calls"], "calls_on_device": st1["pis"]["device"],
            "mismatches": sum(st1["pis"]["mismatch"]), "predInterSearch_share_of_plain_run": in_member / plain_s,
            "amdahl_bound_if_predInterSearch_were_free": 1.0 / (1.0 - in_member / plain_s),
            "note": "process start, clip input and the library's start-up are inside both times; profiles/r04_encoder_replace_*.json hold the 416x240 (reference RA cfg) and 1920x1080 runs"}


def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def main():
    # The contract is ONE JSON line on stdout.  Libraries print banners there (RCCL writes its version block to stdout when the process
    # group comes up), so fd 1 is pointed at stderr for the whole run and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: start the N ranks as a CHILD process (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) before this process
        # imports torch or touches the GPU, relay the ranks' stdout (rank 0's JSON line) and leave with the child's exit code.  No exec: this process stays the parent.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))
        for line in child.stdout:
            os.write(json_fd, line)
        sys.exit(child.wait())
    if a.launch_check:
        os.write(json_fd, (json.dumps({"launch_check": True, "rank": int(os.environ.get("RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1")), "gpus": a.gpus}) + "\n").encode())
        return
    import torch
    import torch.distributed as dist
    from vtm_amd import pipeline, synth
    from vtm_amd.device import Context
    from vtm_amd.pipeline import FrameHotPath

    if a.serial:
        os.environ["VTM_AMD_OVERLAP"] = "0"
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # VTM_BENCH_FORCE_DIST=1: the N > 1 code path (process group, plane broadcast, result gather) with a single rank -- a functional check on a one-GPU box
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

    # ---- synthetic picture set and reference lists ------------------------------------------------------------------------------
    if a.config == "ra":
        cur_poc = max(a.dpoc_list)
        pocs = ([cur_poc - d for d in a.dpoc_list], [cur_poc + d for d in a.dpoc_list])
        sr = tuple([pipeline.asr_search_range(p - cur_poc) for p in l] for l in pocs)   # Clip3(96, 384, (384 |dPOC| + 8) / 16), EncSlice.cpp:1127
    else:
        cur_poc = 4
        pocs = ([3, 2, 1, 0], [])
        sr = ([64] * 4, [])
    with_c = not a.luma_only
    frames = synth.gen_frames(W, H, max(pocs[0] + pocs[1] + [cur_poc]) + 1, chroma=with_c)
    planes, refs, refs_c, off_acc, seen, rsc = [], ([], []), ([], []), 0, {}, 0
    for l in (0, 1):
        for p in pocs[l]:
            if p not in seen:      # one reference picture = its extended luma plane, then (4:2:0) its extended Cb and Cr planes
                if not seen:
                    newest = [off_acc, 0]      # the first picture of list 0 = the picture reconstructed last: [first element, elements] inside the DPB buffer
                buf, off, stride = synth.extend_plane(frames[p][0] if with_c else frames[p], margin=160)
                ent = [(off_acc + off, stride), None]
                planes.append(buf.reshape(-1))
                off_acc += buf.size
                if with_c:
                    offs = []
                    for c in (1, 2):
                        buf, off, rsc = synth.extend_plane(frames[p][c], margin=80)
                        offs.append(off_acc + off)
                        planes.append(buf.reshape(-1))
                        off_acc += buf.size
                    ent[1] = tuple(offs)
                seen[p] = ent
                if len(seen) == 1:
                    newest[1] = off_acc - newest[0]
            refs[l].append(seen[p][0])
            refs_c[l].append(seen[p][1])
    dpb_np = np.concatenate(planes)
    if with_c:
        cur_np, cu, cv = (np.ascontiguousarray(x) for x in frames[cur_poc])
        cur = torch.from_numpy(np.concatenate([cur_np.reshape(-1), cu.reshape(-1), cv.reshape(-1)])).to(dev)      # the original picture: Y | Cb | Cr
        ch_dev = dict(org_off=(W * H, W * H + (W // 2) * (H // 2)), org_stride=W // 2, refs=refs_c, ref_stride=rsc)
        cqp = pipeline.chroma_qp(a.qp) + 12
        ch_cpu = dict(cur=(cu, cv), refs=refs_c, ref_stride=rsc, qp_per=cqp // 6, qp_rem=cqp % 6)
        poc_arg = (cur_poc, pocs[0], pocs[1])
    else:
        cur_np = np.ascontiguousarray(frames[cur_poc])
        cur = torch.from_numpy(cur_np).to(dev)
        ch_dev = ch_cpu = poc_arg = None
    dpb = torch.from_numpy(dpb_np).to(dev)

    ctx = Context(local)
    # the caller's stream carries the dependent chain of the uni searches (the picture's critical path): VTM_BENCH_MAIN_PRIORITY=-1 gives it a high-priority queue, so that
    # its workgroups are dispatched ahead of the side streams' (default priority) when both have work
    if os.environ.get("VTM_BENCH_MAIN_PRIORITY", "0") != "0":
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["VTM_BENCH_MAIN_PRIORITY"])))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    # lambda_motion from the slice QP as the reference derives it for this cfg (LambdaFromQpEnable, DepQuant): lambda = 0.57 * 2^((QP + bitDepthShift) / 3) * 2^(0.25 / 3) with
    # bitDepthShift = 6 * (bitDepth - 8 - DISTORTION_PRECISION_ADJUSTMENT) - 12 = 0 for the 10-bit pictures of this bench (FULL_NBIT: the adjustment is 0; EncSlice.cpp:699-786,
    # TypeDef.h:228-233), m_dLambdaMotionSAD = sqrt( lambda ) (RdCost.cpp:79-84): QP 32 -> 31.3, QP 27 -> 17.6, QP 22 -> 9.9 (rounds 1-3 used the 8-bit shift of -12: 7.83 at QP 32,
    # four times too small for a 10-bit picture -- ADVICE r3; the encoder's own records hold 37 .. 117 at QP 30 + the GOP's QP offsets)
    qp = a.qp
    BIT_DEPTH = 10
    lam = a.lambda_motion if a.lambda_motion > 0 else (0.57 * 2.0 ** ((qp + 6 * (BIT_DEPTH - 8) - 12) / 3.0) * 2.0 ** (0.25 / 3.0)) ** 0.5
    bands = pipeline.ctu_bands(W, H, world, unit=a.shard)
    ctu_filter = pipeline.band_filter(W, bands[rank]) if world > 1 else None
    sim = int(os.environ.get("VTM_BENCH_SIMULATE_WORLD", "0"))      # one GPU computing rank 0's share of an N-GPU run (no exchange): what a rank's step costs
    if sim > 1 and world == 1:
        ctu_filter = pipeline.band_filter(W, pipeline.ctu_bands(W, H, sim, unit=a.shard)[0])
    smvd = (0, 0) if (a.smvd and a.config == "ra") else None      # the nearest picture of either list: equal POC distance, opposite directions (Slice::checkBiDirPred)
    fme_sizes = (128, 64, 32, 16, 8) if a.partition == "qt" else (128, (64, 64), (64, 32), (32, 32), (32, 16), (16, 16), (16, 8), (8, 8))
    fme = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=lam, qp=qp, ctu_filter=ctu_filter, transform_skip=a.transform_skip, pocs=poc_arg, chroma=ch_dev,
                       sizes=fme_sizes, affine=a.affine, low_delay=a.config == "ldp", smvd=smvd)

    # N > 1: the planes of the picture reconstructed last go from rank 0 to every GPU inside every step (RCCL broadcast over xGMI; bytes view: int16 is
    # not a collective dtype).  Every rank keeps ONE resident DPB; the new picture lands in a slot the running step does not reference (a ring of two slots: the
    # newest picture's own slot and a spare one behind the DPB), so the planes of step k + 1 travel while step k computes and every other picture of the DPB
    # stays where it is on every rank.  One table set per slot (the newest picture's offsets differ); the ranks' result records go back to rank 0 (gather),
    # also asynchronously.  The timed region ends only after the last transfer of either kind has landed.
    from vtm_amd.exchange import PlaneExchange, ResultGather
    xchg, gath, fme_slot, res_slot = None, None, [fme], None
    if use_dist:
        spare = int(dpb.numel())
        dpb = torch.cat([dpb, dpb[newest[0]:newest[0] + newest[1]]])          # slot 1 of the ring starts as a copy of the newest picture
        shift = spare - newest[0]
        in_new = lambda off: newest[0] <= off < newest[0] + newest[1]          # noqa: E731
        refs_b = tuple([(o + shift if in_new(o) else o, st) for (o, st) in l] for l in refs)
        ch_b = None
        if ch_dev is not None:
            ch_b = dict(ch_dev, refs=tuple([tuple(c + shift if in_new(c) else c for c in e) for e in l] for l in refs_c))
        fme_slot.append(FrameHotPath(ctx, torch, dev, W, H, W, refs_b, sr, motion_lambda=lam, qp=qp, ctu_filter=ctu_filter, transform_skip=a.transform_skip, pocs=poc_arg,
                                     chroma=ch_b, sizes=fme_sizes, affine=a.affine, low_delay=a.config == "ldp", smvd=smvd))
        xchg = PlaneExchange(dpb, [(newest[0], newest[1]), (spare, newest[1])], src=0)    # one broadcast per reconstructed picture
        xchg.sync_all()                                                                     # set-up: every rank starts from rank 0's pictures
        res_slot = [f.result_tensors() for f in fme_slot]
        gath = ResultGather(sum(t.numel() for t in res_slot[0]), dev, dst=0)

    # --inflight 2: a second set of tables and streams; consecutive steps alternate between the two, so the latency-bound head of picture k + 1
    # (the 128x128 level's searches) runs under the tail of picture k.  Every picture is still computed completely inside the timed region.
    fmes, lanes, turn = [fme], [torch.cuda.current_stream()], [0]
    for _ in range(1, max(1, a.inflight) if not use_dist else 1):
        fmes.append(FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=lam, qp=qp, ctu_filter=ctu_filter, transform_skip=a.transform_skip, pocs=poc_arg, chroma=ch_dev,
                                 sizes=fme_sizes, affine=a.affine, low_delay=a.config == "ldp", smvd=smvd))
        lanes.append(torch.cuda.Stream(device=dev))

    use_graph = a.graph == "on" and not a.serial
    graphs = {}

    def run_picture(f, dpb_ptr):
        """one picture on the current stream: eager launches, or the replay of the graph captured for this table set"""
        if not use_graph:
            f.run(cur.data_ptr(), dpb_ptr)
            return
        g = graphs.get(id(f))
        if g is None:
            outer = torch.cuda.current_stream()
            if "stream" not in graphs:
                graphs["stream"] = torch.cuda.Stream(device=dev)
            cap = graphs["stream"]
            cap.wait_stream(outer)
            with torch.cuda.stream(cap):   # one eager pass on the capture stream first: the library sizes its per-stream workspaces outside the capture
                ctx.set_stream(cap.cuda_stream)
                f.run(cur.data_ptr(), dpb_ptr)
            cap.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):      # the side streams of FrameHotPath.run join the capture through its fork / join events
                ctx.set_stream(cap.cuda_stream)
                f.run(cur.data_ptr(), dpb_ptr)
            ctx.set_stream(outer.cuda_stream)
            graphs[id(f)] = g
        g.replay()

    def step():
        if xchg is not None:
            slot = xchg.next()
            run_picture(fme_slot[slot], dpb.data_ptr())
            gath.submit(res_slot[slot])
        elif len(fmes) == 1:
            run_picture(fme, dpb.data_ptr())
        else:
            k = turn[0] % len(fmes)
            turn[0] += 1
            with torch.cuda.stream(lanes[k]):
                ctx.set_stream(lanes[k].cuda_stream)
                fmes[k].run(cur.data_ptr(), dpb.data_ptr())
            ctx.set_stream(lanes[0].cuda_stream)

    def drain():
        if xchg is not None:
            xchg.drain()
            gath.drain()

    for _ in range(a.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
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

    # ---- per-stage and per-kernel time: a separate, event-instrumented pass after the timed region ---------------------------------
    # stage_ms: events between the steps of the serial order; kernel times: HIP events around every launch of the main kernels, recorded by the
    # library on the stream the kernel is launched on (vtmhip_kernel_timing)
    stage_acc, reps = {}, 3
    ctx.kernel_timing(True)
    for _ in range(reps):
        fme.run(cur.data_ptr(), dpb.data_ptr(), timing=True)
        torch.cuda.synchronize()
        for k2, v in fme.stage_ms().items():
            stage_acc[k2] = stage_acc.get(k2, 0.0) + v / reps
    kern = {}
    for k in KERNELS:
        ms, n = ctx.kernel_timing_read(k)
        if n:
            kern[k] = {"ms_per_step": ms / reps, "launches_per_step": n // reps}
    ctx.kernel_timing(False)

    # ---- SATD 8x8 grid micro-benchmark (extra key; outside the timed steps): each rank takes its band of 8-sample block rows ---------------
    rows8 = H // 8
    r0, r1 = (rows8 * rank) // world, (rows8 * (rank + 1)) // world
    nb = (W // 8) * (r1 - r0)
    satd_out = torch.empty(max(1, nb) * 81, dtype=torch.int32, device=dev)
    ref0_ptr = dpb.data_ptr() + 2 * (refs[0][0][0] + r0 * 8 * refs[0][0][1])
    cur_ptr = cur.data_ptr() + 2 * r0 * 8 * W
    # ONE number: 2 warm-up launches, then SATD_REPS launches back to back between two events on the launch stream (torch's current stream IS the context's stream here); the
    # per-launch event bracket of vtmhip_kernel_timing is NOT used for this kernel (it added ~80 us to a 144 us launch in round 3 and gave the line two different rates)
    SATD_REPS = 20

    def satd_rate(c_ptr, c_stride, r_ptr, r_stride, w, h, out_ptr):
        for _ in range(2):
            ctx.satd8_grid(c_ptr, c_stride, r_ptr, r_stride, w, h, 4, out_ptr)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(SATD_REPS):
            ctx.satd8_grid(c_ptr, c_stride, r_ptr, r_stride, w, h, 4, out_ptr)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / SATD_REPS

    satd_ms = satd_rate(cur_ptr, W, ref0_ptr, refs[0][0][1], W, (r1 - r0) * 8, satd_out.data_ptr())
    # the other two picture sizes SURVEY.md 8(d) lists for this micro-benchmark (rank 0, N = 1 only; uniform random 10-bit samples: the kernel's arithmetic path -- packed
    # 16-bit levels for samples inside [0, 1023] -- is the one the synthetic picture takes)
    satd_sizes = {}
    if world == 1 and not a.no_cpu_baseline:
        for (sw, sh) in ((1920, 1080), (7680, 4320)):
            if (sw, sh) == (W, H):
                continue
            g = torch.Generator(device=dev)
            g.manual_seed(sw)
            sc = torch.randint(0, 1024, (sh, sw), dtype=torch.int16, device=dev, generator=g)
            sr_ = torch.randint(0, 1024, (sh + 16, sw + 16), dtype=torch.int16, device=dev, generator=g)
            so = torch.empty((sw // 8) * (sh // 8) * 81, dtype=torch.int32, device=dev)
            ms_ = satd_rate(sc.data_ptr(), sw, sr_.data_ptr() + 2 * (8 * (sw + 16) + 8), sw + 16, sw, (sh // 8) * 8, so.data_ptr())
            satd_sizes["%dx%d" % (sw, sh)] = {"pairs_per_launch": (sw // 8) * (sh // 8) * 81, "ms_per_launch": ms_, "gblocks_per_s": (sw // 8) * (sh // 8) * 81 / ms_ / 1e6}
            del sc, sr_, so
    satd_g = torch.tensor([nb * 81 / satd_ms / 1e6], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(satd_g)

    # SAD candidates of the integer searches (SURVEY.md 8d prices one at 4 * W * H >> subShift algorithmic bytes): counted after the timed region by running the levels' uni
    # searches once more through the public TZ call (the kernel keeps the count per search) -- rank 0, one GPU
    sad_cands = None
    if rank == 0 and world == 1 and sim <= 1:
        try:
            sad_cands = fme.sad_candidates(cur.data_ptr(), dpb.data_ptr())
        except Exception as e:      # a statistic: never fails the line
            print("sad_candidates:", e, file=sys.stderr)
    if rank == 0:
        wc = fme.work_counts()
        # ---- roofline of the dominant kernel: vector-ALU ISSUE (these kernels are integer / packed-16-bit instruction streams; their HBM traffic is
        # a few per cent of the 8 TB/s: DESIGN.md section 4).  insts = SQ_INSTS_VALU per step from the committed PMC pass of this workload
        # (profiles/pmc_insts_per_launch.json), priced with the kernel's measured issue cost per instruction (scripts/valu_issue.hip micro-benchmark
        # x static instruction mix, profiles/isa_mix.json); peak = what 1024 SIMDs issue at 2.4 GHz; the kernel time is measured live (HIP events).
        mix, pmc_i, pmc_b = load_json("isa_mix.json") or {}, load_json("pmc_insts_per_launch.json") or {}, load_json("pmc_hbm_traffic_per_launch_KB.json") or {}
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None

        wkey = workload_key(a, world)
        pmc_meta = pmc_i.get("_meta")

        def issue_roofline(name, ms, launches, in_step=True):
            # in_step: the kernel runs inside the picture step -> counters summed over its launches of one step; otherwise (the SATD micro-benchmark) ONE launch
            cpi = next((v["cycles_per_valu_inst"] for k, v in mix.items() if isinstance(v, dict) and k.startswith(name)), None)
            lps = (lambda v: v.get("launches_per_step", 1)) if in_step else (lambda v: 1)
            # the committed counters were collected from THIS command (in-step kernels) on THIS kernel's sources; the SATD micro-benchmark only depends on the picture size
            ok = counters_for(name, pmc_meta, wkey, same_args=in_step) and (in_step or ((W, H) == (3840, 2160) and world == 1))
            sel = [v for k, v in pmc_i.items() if k.startswith(name) and isinstance(v, dict)] if ok else []
            insts = sum(v.get("SQ_INSTS_VALU", 0) * lps(v) for v in sel) or None
            salu = sum(v.get("SQ_INSTS_SALU", 0) * lps(v) for v in sel) or None
            traffic = (sum((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 * lps(v) for k, v in pmc_b.items() if k.startswith(name) and isinstance(v, dict)) or None) if ok else None
            r = {"bound": "valu_issue", "kernel": name, "unit": "G wave-instructions/s", "ms_per_step": ms, "launches_per_step": launches,
                 "ms_per_launch": ms / max(1, launches), "insts": insts, "traffic": (traffic / launches) if traffic else None,
                 "peak": SIMDS * CLOCK_HZ / GUIDE_CYCLES / 1e9, "achieved": None, "frac": None, "counters_match_this_run": bool(ok)}
            if insts:
                r["achieved"] = insts / (ms * 1e-3) / 1e9
                r["frac"] = r["achieved"] / r["peak"]
                if cpi:      # the same achieved rate against what this kernel's instruction MIX can issue at the clock the micro-benchmark held
                    r["mixed_issue"] = {"cycles_per_inst": cpi, "clock_GHz": MIX_CLOCK_HZ / 1e9, "peak": SIMDS * MIX_CLOCK_HZ / cpi / 1e9,
                                        "frac": r["achieved"] / (SIMDS * MIX_CLOCK_HZ / cpi / 1e9)}
            if salu:
                # the scalar port of a SIMD issues one instruction per 4.03 cycles whatever the number of waves, beside the vector port (s_add_u32 rows and the
                # vector + scalar pair row of profiles/r02_valu_issue.jsonl): a kernel is bounded by the busier of the two
                r["salu_insts"] = salu
                r["salu_frac"] = salu * SALU_CYCLES / (SIMDS * MIX_CLOCK_HZ * ms * 1e-3)
            if traffic:
                r["hbm_GBps"] = traffic / (ms * 1e-3) / 1e9
                r["hbm_frac"] = r["hbm_GBps"] / 8000.0
            r["note"] = ("insts = SQ_INSTS_VALU per step from the committed rocprofv3 --pmc pass (profiles/pmc_insts_per_launch.json); used only when that pass ran this command on these "
                         "kernel sources (its _meta key), else achieved / frac are null.  peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md: "
                         "the machine's full-rate issue peak).  mixed_issue: the same rate against the measured issue cost of this kernel's static mix of half-rate opcodes "
                         "(packed-16 / sad / dot / perm / min-max / mul / cmp: 3.5 .. 4.0 cycles, profiles/r02_valu_issue.jsonl x profiles/isa_mix.json) at the 2.3 GHz that "
                         "benchmark held.  Kernel time measured live with HIP events on the launch stream; traffic = HBM-side bytes per launch (PMC FETCH_SIZE x 2 + WRITE_SIZE)")
            return r

        out = {
            "metric": "hot-path pictures/sec (%dx%d %s QP%d; level-order predInterSearch: AMVP estimation, TZ + fractional ME per (list, refIdx), bi refinement, "
                      "MC, residual xT/quant/xIT/SSE; not a full encode) + SATD Gblocks/s" % (W, H, "randomaccess" if a.config == "ra" else "lowdelay_P", qp),
            "value": a.steps / dt, "unit": "pictures/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int16 samples, int32 accumulation (fp64 MV-rate multiply)", "data": "synthetic",
            "config": {"workload": "%dx%d 10-bit, %s operating point (QP%d, lambda_motion %.2f, %s, FEN): %s = %d PUs x (%d + %d) reference pictures = %d uni searches + %d bi searches "
                                   "per picture, %d TU x transform-candidate chains"
                                   % (W, H, "encoder_randomaccess_vtm.cfg" if a.config == "ra" else "encoder_lowdelay_P_vtm.cfg", qp, lam,
                                      ("SR %s via ASR" % "/".join(str(x) for x in sorted({v for l in sr for v in l}))) if a.config == "ra" else "SR 64", "quadtree PUs 128..8" if a.partition == "qt" else "split-shape PU levels 128x128 64x64 64x32 32x32 32x16 16x16 16x8 8x8",
                                      wc["pus"] * world if world > 1 else wc["pus"], len(refs[0]), len(refs[1]),
                                      wc["uni_searches"], wc["bi_searches"], wc["tu_chains"])
                                   + (", %d affine uni searches (4-parameter xAffineMotionEstimation, PUs >= 16x16)" % wc["affine_searches"] if a.affine else "")
                                   + (", %d SMVD searches (the symmetric-MVD block of predInterSearch)" % wc["smvd_searches"] if smvd else "")
                                   + (" (this rank's share)" if world > 1 else ""),
                       "stages": ["xEstimateMvPredAMVP", "xMotionEstimation uni (TZ + frac)", "xCheckBestMVP / best reference", "bi refinement (MC + removeHighFreq fused, xPatternSearch, frac)",
                                  "uni/bi decision", "final prediction + residual (fused)" + ("" if a.luma_only else "; BDOF where xPredInterBi applies it; Cb / Cr prediction + residual"),
                                  "tu_chain (xT, quant, dequant, xIT, SSE)" + ("" if a.luma_only else ", luma MTS candidates + chroma DCT2 at the mapped chroma QP")],
                       "order": ("one stream" if a.serial else "level-major over 5 side streams (each level's later stages run beside the next levels' searches)")
                                + (", replayed from a hipGraph" if use_graph else ""),
                       "parallelism": ("1 GPU" if world == 1 else "one picture, CTUs sharded over %d GPUs (%s): bands %s; the planes of the newest reference picture (Y, Cb, Cr) broadcast from rank 0 and the results gathered to rank 0 every step"
                                       % (world, "raster-scan CTU ranges" if a.shard == "ctu" else "whole CTU rows", [b[1] - b[0] for b in bands]))},
            "satd_gblocks_per_s": float(satd_g.item()),
            "stages_ms": stage_acc, "kernels": kern, "workload_key": wkey,
        }
        if sad_cands:
            tz_ms = sum(kern[k]["ms_per_step"] for k in ("tz_search_kernel", "tz_group_kernel", "tz_raster_cols_kernel") if k in kern)
            alg = sum(c * ((4 * w * h) >> ss) for (w, h, ss, n, c) in sad_cands)
            out["integer_search"] = {"levels": [{"pu": "%dx%d" % (w, h), "subShift": ss, "searches": n, "candidates": c, "candidates_per_search": c / max(1, n),
                                                 "algorithmic_bytes": c * ((4 * w * h) >> ss)} for (w, h, ss, n, c) in sad_cands],
                                     "candidates_per_picture": sum(c for (_, _, _, _, c) in sad_cands), "algorithmic_GB_per_picture": alg / 1e9,
                                     "kernels_ms_per_picture": tz_ms, "algorithmic_GBps": alg / 1e9 / (tz_ms * 1e-3) if tz_ms else None,
                                     "note": "SURVEY.md 8(d): a SAD candidate = 4 * W * H >> subShift algorithmic bytes (org + reference samples); candidates = sum of the "
                                             "TZ kernel's per-search count (start points, diamond rounds, raster scan, star refinement) of every uni search of the picture; the "
                                             "rate is against tz_search_kernel + tz_group_kernel + tz_raster_cols_kernel time.  The samples are re-used out of L1 / L2 (counter traffic: roofline.traffic), "
                                             "so this is a work rate, not an HBM fraction"}
        if dom:
            out["roofline"] = issue_roofline(dom, kern[dom]["ms_per_step"], kern[dom]["launches_per_step"])
            # the next two kernel families by time, same definition (the three largest are within 6 % of each other at the default operating point)
            out["roofline_next"] = []
            for k in sorted(kern, key=lambda k: -kern[k]["ms_per_step"])[1:3]:
                r = issue_roofline(k, kern[k]["ms_per_step"], kern[k]["launches_per_step"])
                r.pop("note", None)
                out["roofline_next"].append(r)
        r = issue_roofline("satd8_grid_kernel", satd_ms, 1, in_step=False)      # the same 20 back-to-back launches as satd_gblocks_per_s: one rate per line
        r["pairs_per_s_G"] = nb * 81 / satd_ms / 1e6
        r["pairs_per_launch"] = nb * 81
        out["satd_roofline"] = r
        satd_sizes["%dx%d" % (W, H)] = {"pairs_per_launch": nb * 81, "ms_per_launch": satd_ms, "gblocks_per_s": nb * 81 / satd_ms / 1e6}
        out["satd_grid_sizes"] = satd_sizes
        if world == 1:
            # what a host encoder would add per picture if the inputs were NOT resident: the original picture and the picture reconstructed last (the other
            # reference pictures are already on the device) from pinned host memory, measured here, not overlapped -- never part of `value`
            n_up = int(cur.numel() + newest[1])
            host = torch.empty(n_up, dtype=torch.int16).pin_memory()
            stage = torch.empty(n_up, dtype=torch.int16, device=dev)
            stage.copy_(host, non_blocking=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                stage.copy_(host, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            up_ms = e0.elapsed_time(e1) / 5
            out["pcie"] = {"h2d_bytes_per_picture": 2 * n_up, "h2d_ms_per_picture": up_ms, "h2d_GBps": 2 * n_up / up_ms / 1e6,
                           "pictures_per_s_if_not_overlapped": 1000.0 / (1e3 * dt / a.steps + up_ms)}
            del host, stage
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fme, cur_np, dpb_np, refs, sr, W, H, lam, qp, a.cpu_seconds, pocs=poc_arg, chroma=ch_cpu, affine=a.affine, low_delay=a.config == "ldp", smvd=smvd)
            if not a.no_encoder_level:
                try:
                    out["cpu_baseline"]["encoder_level"] = encoder_level()
                except Exception as e:      # a reported side figure: never fails the line
                    out["cpu_baseline"]["encoder_level"] = {"error": str(e)[-400:]}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
