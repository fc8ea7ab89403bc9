"""GPU end-to-end parity of the level-order predInterSearch driver (vtm_amd.pipeline.FrameHotPath: AMVP estimation -> uni ME per (list, refIdx)
-> xCheckBestMVP -> bi refinement -> decision -> prediction / residual -> TU chains) against the same chain through the CPU oracle and
through the REAL reference members (tests/cpu_pis.py), PU by PU, at the operating points of BASELINE.json's configurations:
  random access B slices (1 + 1 and 2 + 2 reference pictures, SR via ASR), low-delay P slices (4 list-0 pictures, SR 64), QP 22 / 27 / 32,
and the CTU sharding of one frame (the union of the ranks' tables is the unsharded result)."""
import numpy as np
import pytest

import cpu_pis
import oracle_lib as ol
from vtm_amd import pipeline, synth
from vtm_amd.pipeline import FrameHotPath

pytestmark = pytest.mark.gpu


def make_scene(torch, dev, W, H, pocs0, pocs1, cur_poc, hard=True, chroma=False):
    """-> (cur_np, dpb_np, refs, sr, cur, dpb[, chroma_dev, chroma_cpu]): with chroma=True the original buffer is Y | Cb | Cr, every reference picture's
    extended Cb / Cr planes follow its luma plane in the reference buffer, and the two dicts are FrameHotPath's / cpu_pis.run_pu's `chroma` arguments"""
    nfr = max(pocs0 + pocs1 + [cur_poc]) + 1
    frames = (synth.gen_frames_hard if hard else synth.gen_frames)(W, H, nfr, chroma=chroma)
    planes, refs, acc = [], ([], []), 0
    refs_c = ([], [])
    cache = {}
    rsc = 0
    for l, pocs in enumerate((pocs0, pocs1)):
        for p in pocs:
            if p not in cache:
                buf, off, stride = synth.extend_plane(frames[p][0] if chroma else frames[p], margin=160)
                ent = [(acc + off, stride)]
                planes.append(buf.reshape(-1))
                acc += buf.size
                if chroma:
                    offs = []
                    for c in (1, 2):
                        buf, off, rsc = synth.extend_plane(frames[p][c], margin=80)
                        offs.append(acc + off)
                        planes.append(buf.reshape(-1))
                        acc += buf.size
                    ent.append(tuple(offs))
                cache[p] = ent
            refs[l].append(cache[p][0])
            if chroma:
                refs_c[l].append(cache[p][1])
    dpb_np = np.concatenate(planes)
    sr = ([pipeline.asr_search_range(p - cur_poc) for p in pocs0], [pipeline.asr_search_range(p - cur_poc) for p in pocs1])
    if not chroma:
        cur_np = np.ascontiguousarray(frames[cur_poc])
        return cur_np, dpb_np, refs, sr, torch.from_numpy(cur_np).to(dev), torch.from_numpy(dpb_np).to(dev)
    y, u, v = (np.ascontiguousarray(a) for a in frames[cur_poc])
    cur_all = np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)])
    ch_dev = dict(org_off=(W * H, W * H + (W // 2) * (H // 2)), org_stride=W // 2, refs=refs_c, ref_stride=rsc)
    ch_cpu = dict(cur=(u, v), refs=refs_c, ref_stride=rsc)
    return y, dpb_np, refs, sr, torch.from_numpy(cur_all).to(dev), torch.from_numpy(dpb_np).to(dev), ch_dev, ch_cpu


def check(hp, cur_np, dpb_np, refs, sr, W, H, lam, qp, R, per_level=24, min_checked=60, pocs=None, chroma=None, stats=None, affine=False, low_delay=False, smvd=None):
    snaps = hp.snapshot()
    nref = hp.nref
    checked = 0
    dirs = set()
    if chroma is not None:
        cqp = pipeline.chroma_qp(qp) + 12
        chroma = dict(chroma, qp_per=cqp // 6, qp_rem=cqp % 6)
    for li, lvl in enumerate(snaps):
        parent = snaps[lvl["parent_level"]] if lvl["parent_level"] >= 0 else None
        s, npu = lvl["size"], lvl["npu"]
        for i in range(0, npu, max(1, npu // per_level)):
            out = cpu_pis.run_pu(cur_np, dpb_np.ctypes.data, refs, sr, W, H, s, int(lvl["xs"][i]), int(lvl["ys"][i]), cpu_pis.cands_of(lvl, nref, i), lam,
                                 (qp + 12) // 6, (qp + 12) % 6, lvl["cands"], ref=R, pocs=pocs, chroma=chroma, affine=affine, low_delay=low_delay, smvd=smvd)
            cpu_pis.compare_with_device(lvl, parent, nref, i, out)
            if stats is not None and "smvd" in out:
                stats["smvd"] = stats.get("smvd", 0) + 1
                stats["smvd_won"] = stats.get("smvd_won", 0) + out["smvd_mode"]
                stats["smvd_moved"] = stats.get("smvd_moved", 0) + int(out["smvd"][0] != out["smvd"][2])
            if stats is not None and "aff" in out:
                stats["aff"] = stats.get("aff", 0) + len(out["aff"])
                stats["aff_moved"] = stats.get("aff_moved", 0) + sum(1 for (mv, _, _) in out["aff"].values() if len(set(mv)) > 1)
            dirs.add(out["inter_dir"])
            if stats is not None:
                stats["bio"] = stats.get("bio", 0) + int(out["bio"])
                stats["chroma_nz"] = stats.get("chroma_nz", 0) + sum(1 for v in out.get("tus_c", {}).values() if v[2])
                stats["mts_pruned"] = stats.get("mts_pruned", 0) + sum(f.count(0) for f in out.get("mts", {}).values())
                stats["mts_kept"] = stats.get("mts_kept", 0) + sum(f.count(1) for f in out.get("mts", {}).values())
            checked += 1
    assert checked >= min_checked
    return dirs


@pytest.mark.parametrize("use_ref", [False, True])
@pytest.mark.parametrize("name,pocs0,pocs1,cur,qp,ts", [("ra_1+1_qp32", [0], [4], 2, 32, False), ("ra_2+2_qp27_ts", [2, 0], [4, 6], 3, 27, True),
                                                       ("ldp_4_qp32", [3, 2, 1, 0], [], 4, 32, False), ("ra_1+1_qp22", [1], [3], 2, 22, False)])
def test_frame_hot_path_matches_cpu_chain(use_ref, name, pocs0, pocs1, cur, qp, ts):
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    if use_ref and not ol.have_ref():
        pytest.skip("oracle/_ref/libvtmref.so not present")
    W, H = 256, 128
    dev = torch.device("cuda", 0)
    cur_np, dpb_np, refs, sr, cur, dpb = make_scene(torch, dev, W, H, pocs0, pocs1, cur)
    if not pocs1:
        sr = ([64] * len(pocs0), [])          # encoder_lowdelay_P_vtm.cfg: SearchRange 64, no ASR
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lam = 8.0
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=lam, qp=qp, transform_skip=ts)
    for overlapped in (True, False):       # level-major over the side streams, then one stream: same tables, same results
        hp.run(cur.data_ptr(), dpb.data_ptr(), timing=not overlapped)
        torch.cuda.synchronize()
        dirs = check(hp, cur_np, dpb_np, refs, sr, W, H, lam, qp, ol.ref() if use_ref else None, per_level=24 if overlapped else 6, min_checked=60 if overlapped else 15)
    if pocs1:
        assert 3 in dirs and len(dirs) >= 2      # bi-prediction and uni-prediction both occur
    ctx.close()


@pytest.mark.parametrize("use_ref", [False, True])
@pytest.mark.parametrize("name,pocs0,pocs1,cur,qp", [("ra_1+1_qp32", [0], [4], 2, 32), ("ra_2+2_qp27", [2, 0], [4, 6], 3, 27), ("ldp_2_qp32", [1, 0], [], 2, 32)])
def test_frame_hot_path_with_bdof_and_chroma(use_ref, name, pocs0, pocs1, cur, qp):
    """The full final prediction of the driver: BDOF on the bi-predicted PUs xPredInterBi gives it to (opposite directions, equal POC distance,
    size rule), plain weighted average on the rest, the two 4:2:0 chroma planes through the 4-tap chroma filter, chroma residual and the DCT2
    TU chain at the mapped chroma QP -- PU by PU against the oracle and against the reference's own xPredInterBlk / applyBiOptFlow / xT / xIT."""
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    if use_ref and not ol.have_ref():
        pytest.skip("oracle/_ref/libvtmref.so not present")
    W, H = 256, 128
    dev = torch.device("cuda", 0)
    cur_np, dpb_np, refs, sr, cur_d, dpb, ch_dev, ch_cpu = make_scene(torch, dev, W, H, pocs0, pocs1, cur, chroma=True)
    if not pocs1:
        sr = ([64] * len(pocs0), [])
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lam = 8.0
    pocs = (cur, pocs0, pocs1)
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=lam, qp=qp, pocs=pocs, chroma=ch_dev)
    hp.run(cur_d.data_ptr(), dpb.data_ptr())
    torch.cuda.synchronize()
    stats = {}
    check(hp, cur_np, dpb_np, refs, sr, W, H, lam, qp, ol.ref() if use_ref else None, per_level=20, min_checked=50, pocs=pocs, chroma=ch_cpu, stats=stats)
    if pocs1:
        assert stats["bio"] >= 5, stats         # BDOF really ran on some of the checked PUs
    assert stats["chroma_nz"] >= 5, stats        # and chroma TUs with non-zero levels were compared
    ctx.close()


@pytest.mark.parametrize("use_ref", [False, True])
@pytest.mark.parametrize("name,sizes", [("bt", (128, (64, 64), (64, 32), (32, 32), (16, 32), (16, 16), (16, 8), (8, 8))),
                                        ("tt", ((64, 64), (64, 16), (32, 16), (32, 8), (8, 8))), ("tt_ver", ((64, 64), (16, 64), (16, 32), (8, 32), (8, 16)))])
def test_frame_hot_path_on_split_shapes(use_ref, name, sizes):
    """The driver on binary / ternary split shapes: every level is one W x H shape whose blocks nest inside the previous level's; the
    uniform rectangular fast paths (TZ, tiled fractional search with 16x8 / 8x16 Hadamard tiles, lane-per-candidate refinement, rectangular TU chains,
    BDOF, chroma) against the CPU chain through the oracle and through the reference's own members, PU by PU."""
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    if use_ref and not ol.have_ref():
        pytest.skip("oracle/_ref/libvtmref.so not present")
    W, H = 256, 128
    dev = torch.device("cuda", 0)
    pocs0, pocs1, cur = [0], [4], 2
    cur_np, dpb_np, refs, sr, cur_d, dpb, ch_dev, ch_cpu = make_scene(torch, dev, W, H, pocs0, pocs1, cur, chroma=True)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lam, qp = 8.0, 32
    pocs = (cur, pocs0, pocs1)
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=lam, qp=qp, sizes=sizes, pocs=pocs, chroma=ch_dev)
    hp.run(cur_d.data_ptr(), dpb.data_ptr())
    torch.cuda.synchronize()
    stats = {}
    dirs = check(hp, cur_np, dpb_np, refs, sr, W, H, lam, qp, ol.ref() if use_ref else None, per_level=12, min_checked=50, pocs=pocs, chroma=ch_cpu, stats=stats)
    assert 3 in dirs and stats["bio"] >= 3 and stats["chroma_nz"] >= 5, (dirs, stats)
    assert stats["mts_pruned"] >= 5 and stats["mts_kept"] > stats["mts_pruned"], stats     # the pre-selection of transformNxN( trModes ) really prunes here
    print("mts:", name, stats["mts_kept"], "kept,", stats["mts_pruned"], "pruned")
    ctx.close()


@pytest.mark.parametrize("use_ref", [False, True])
@pytest.mark.parametrize("name,pocs0,pocs1,cur,low_delay", [("ra_1+1", [0], [4], 2, False), ("ldp_2", [1, 0], [], 2, True)])
def test_frame_hot_path_affine_uni_stage(use_ref, name, pocs0, pocs1, cur, low_delay):
    """The affine uni stage of the driver: InterSearch::xAffineMotionEstimation (4-parameter, PROF, gradient iterations + control-point refinement) for every
    (PU >= 16x16, list, refIdx) row, started from the row's translational result -- vector triple, bits and cost of every sampled row against the oracle and
    against the reference's own member."""
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    if use_ref and not ol.have_ref():
        pytest.skip("oracle/_ref/libvtmref.so not present")
    W, H = 256, 128
    dev = torch.device("cuda", 0)
    cur_np, dpb_np, refs, sr, cur_d, dpb = make_scene(torch, dev, W, H, pocs0, pocs1, cur, hard=False)
    if not pocs1:
        sr = ([64] * len(pocs0), [])
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lam, qp = 8.0, 32
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=lam, qp=qp, sizes=(64, 32, (32, 16), 16), affine=True, low_delay=low_delay)
    hp.run(cur_d.data_ptr(), dpb.data_ptr())
    torch.cuda.synchronize()
    stats = {}
    check(hp, cur_np, dpb_np, refs, sr, W, H, lam, qp, ol.ref() if use_ref else None, per_level=8, min_checked=24, stats=stats, affine=True, low_delay=low_delay)
    assert stats["aff"] >= 40 and stats["aff_moved"] >= 3, stats      # rows compared, and rows whose model left pure translation
    ctx.close()


@pytest.mark.parametrize("use_ref", [False, True])
@pytest.mark.parametrize("name,pocs0,pocs1,cur,sym,hard", [("ra_1+1", [0], [4], 2, (0, 0), True), ("ra_2+2", [2, 0], [4, 6], 3, (0, 0), False), ("ra_2+2_far", [2, 0], [6, 4], 3, (1, 0), True)])
def test_frame_hot_path_smvd_stage(use_ref, name, pocs0, pocs1, cur, sym, hard):
    """The SMVD block of predInterSearch in the driver (between the bi refinement and the uni / bi decision): predictor pair, start vectors, symmvdCheckBestMvp,
    xSymmetricMotionEstimation, the final predictor check and the symCost < uiCostBi replacement -- every sampled PU against the chain built from the oracle's
    members and from the reference's own three members; BDOF is off for the PUs the pair wins, the final prediction and the TU chains follow."""
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    if use_ref and not ol.have_ref():
        pytest.skip("oracle/_ref/libvtmref.so not present")
    W, H = 256, 128
    dev = torch.device("cuda", 0)
    cur_np, dpb_np, refs, sr, cur_d, dpb = make_scene(torch, dev, W, H, pocs0, pocs1, cur, hard=hard)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lam, qp = 8.0, 32
    pocs = (cur, pocs0, pocs1)
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=lam, qp=qp, sizes=(128, 64, 32, (32, 16), 16, 8), pocs=pocs, smvd=sym)
    for native in (True, False):
        hp.run(cur_d.data_ptr(), dpb.data_ptr(), timing=not native)
        torch.cuda.synchronize()
        stats = {}
        check(hp, cur_np, dpb_np, refs, sr, W, H, lam, qp, ol.ref() if use_ref else None, per_level=16 if native else 5, min_checked=40 if native else 12, pocs=pocs,
              stats=stats, smvd=sym)
        assert stats["smvd"] >= (40 if native else 12), stats
    print("smvd:", name, stats)
    assert stats["smvd_moved"] >= 1, stats
    ctx.close()


def test_full_size_picture_is_deterministic_and_order_independent():
    """BASELINE's picture size (3840x2160, 2 + 2 references, chroma + BDOF) through size-independent properties: the result tables of the native
    level-major loop over six streams, of the step-by-step Python loop on one stream, and of a second run are identical byte for byte (no race between the
    levels' chains, no dependence on atomics' order), and a spot check of PUs of every level against the oracle chain."""
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    W, H = 3840, 2160
    dev = torch.device("cuda", 0)
    pocs0, pocs1, cur = [2, 0], [6, 8], 4
    cur_np, dpb_np, refs, sr, cur_d, dpb, ch_dev, ch_cpu = make_scene(torch, dev, W, H, pocs0, pocs1, cur, hard=False, chroma=True)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lam, qp, pocs = 8.0, 32, (cur, pocs0, pocs1)
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=lam, qp=qp, pocs=pocs, chroma=ch_dev)

    def tables():
        torch.cuda.synchronize()
        return [t.clone() for t in hp.result_tensors()] + [lvl["uni_rows"].clone() for lvl in hp.levels]
    hp.run(cur_d.data_ptr(), dpb.data_ptr())                     # native loop, levels overlapping
    a = tables()
    hp.run(cur_d.data_ptr(), dpb.data_ptr())                     # again
    b = tables()
    hp.run(cur_d.data_ptr(), dpb.data_ptr(), timing=True)        # Python loop, one stream, stage by stage
    c = tables()
    for k, (x, y, z) in enumerate(zip(a, b, c)):
        assert torch.equal(x, y), ("second run differs", k)
        assert torch.equal(x, z), ("serial order differs", k)
    assert sum(int(l["npu"]) for l in hp.snapshot()) == 172500
    check(hp, cur_np, dpb_np, refs, sr, W, H, lam, qp, None, per_level=3, min_checked=10, pocs=pocs, chroma=ch_cpu)
    ctx.close()


def test_ctu_sharding_union_equals_unsharded():
    """north_star: CTU rows of a frame shard across the GPUs.  Two ranks' tables (raster-scan CTU ranges; the boundary row is cut at a CTU) run
    one after the other on this GPU: the union of their per-PU / per-TU results is bit-identical to the unsharded picture."""
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    W, H = 640, 384                      # 5 x 3 CTUs: the two bands are 8 and 7 CTUs, the cut falls inside CTU row 1
    dev = torch.device("cuda", 0)
    cur_np, dpb_np, refs, sr, cur, dpb = make_scene(torch, dev, W, H, [0], [4], 2)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    full = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr)
    full.run(cur.data_ptr(), dpb.data_ptr())
    torch.cuda.synchronize()
    whole = {lv["size"]: lv for lv in full.snapshot()}
    seen = {s: 0 for s in whole}
    for unit in ("ctu", "row"):
        bands = pipeline.ctu_bands(W, H, 2, unit=unit)
        assert bands[0][1] == bands[1][0] and bands[0][0] == 0 and bands[1][1] == 15
        seen = {s: 0 for s in whole}
        for band in bands:
            part = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, ctu_filter=pipeline.band_filter(W, band))
            part.run(cur.data_ptr(), dpb.data_ptr())
            torch.cuda.synchronize()
            for lv in part.snapshot():
                w = whole[lv["size"]]
                lut = {(int(x), int(y)): k for k, (x, y) in enumerate(zip(w["xs"], w["ys"]))}
                idx = np.array([lut[(int(x), int(y))] for x, y in zip(lv["xs"], lv["ys"])])
                assert np.array_equal(lv["pus"], w["pus"][idx]), ("pus", unit, band, lv["size"])
                R = full.nref[0] + full.nref[1]
                for lr in range(R):
                    assert np.array_equal(lv["uni_rows"][lr * lv["npu"]:(lr + 1) * lv["npu"]], w["uni_rows"][lr * w["npu"] + idx]), ("rows", unit, band, lv["size"])
                q2 = (lv["w"] // lv["tw"]) * (lv["h"] // lv["th"])
                for ci in range(lv["nc"]):
                    a = lv["tu_res"][ci * lv["ntu"]:(ci + 1) * lv["ntu"]].reshape(lv["npu"], q2, 2)
                    b = w["tu_res"][ci * w["ntu"]:(ci + 1) * w["ntu"]].reshape(w["npu"], q2, 2)[idx]
                    assert np.array_equal(a, b), ("tu", unit, band, lv["size"], ci)
                seen[lv["size"]] += lv["npu"]
        assert seen == {s: lv["npu"] for s, lv in whole.items()}
    ctx.close()


@pytest.mark.parametrize("use_ref", [False, True])
def test_frame_hot_path_config5_tool_set(use_ref):
    """BASELINE config 5's tool set as ONE operating point: QP 22, 2 + 2 reference pictures, affine uni stage + SMVD block + MTS candidates + the
    transform-skip candidate + chroma + BDOF together -- every sampled PU of every level against the oracle chain and against the reference's own members."""
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    if use_ref and not ol.have_ref():
        pytest.skip("oracle/_ref/libvtmref.so not present")
    W, H = 256, 128
    dev = torch.device("cuda", 0)
    pocs0, pocs1, cur, sym = [2, 0], [6, 8], 4, (0, 0)
    cur_np, dpb_np, refs, sr, cur_d, dpb, ch_dev, ch_cpu = make_scene(torch, dev, W, H, pocs0, pocs1, cur, hard=False, chroma=True)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lam, qp, pocs = 8.0, 22, (cur, pocs0, pocs1)
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=lam, qp=qp, sizes=(128, 64, 32, 16, 8), transform_skip=True, pocs=pocs, chroma=ch_dev, affine=True, smvd=sym)
    hp.run(cur_d.data_ptr(), dpb.data_ptr())
    torch.cuda.synchronize()
    stats = {}
    check(hp, cur_np, dpb_np, refs, sr, W, H, lam, qp, ol.ref() if use_ref else None, per_level=10, min_checked=30, pocs=pocs, chroma=ch_cpu, stats=stats, affine=True, smvd=sym)
    assert stats["aff"] >= 40 and stats["smvd"] >= 30 and stats["chroma_nz"] >= 5 and stats["mts_kept"] >= 20, stats
    assert any(lv["cands"][:2] == [0, 1] for lv in hp.snapshot())      # the transform-skip candidate is in the TU chains
    print("config5:", stats)
    ctx.close()


def test_kernel_timing_does_not_disturb_the_picture_loop():
    """vtmhip_kernel_timing used to destroy the fork / join events of vtmhip_pis_run_picture and leave the stale handles in the pool.  Timing on,
    timing off, then an overlapped picture: the tables must equal the serial run's, and the timed kernels must report launches."""
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    W, H = 256, 128
    dev = torch.device("cuda", 0)
    cur_np, dpb_np, refs, sr, cur, dpb = make_scene(torch, dev, W, H, [0], [4], 2)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr)

    def tables():
        torch.cuda.synchronize()
        return [t.clone() for t in hp.result_tensors()] + [lvl["uni_rows"].clone() for lvl in hp.levels]
    hp.run(cur.data_ptr(), dpb.data_ptr(), timing=True)          # one stream
    serial = tables()
    hp.run(cur.data_ptr(), dpb.data_ptr())                       # overlapped: fills the context's fork / join event pool
    torch.cuda.synchronize()
    for _ in range(2):
        ctx.kernel_timing(True)
        hp.run(cur.data_ptr(), dpb.data_ptr())
        torch.cuda.synchronize()
        ms, n = ctx.kernel_timing_read("tz_search_kernel")
        assert n > 0 and ms > 0
        ctx.kernel_timing(False)
        for t in hp.result_tensors():
            t.zero_()
        hp.run(cur.data_ptr(), dpb.data_ptr())                   # overlapped again, on the (reused) events
        for k, (x, y) in enumerate(zip(serial, tables())):
            assert torch.equal(x, y), ("overlapped run after kernel_timing differs from the serial run", k)
    ctx.close()
