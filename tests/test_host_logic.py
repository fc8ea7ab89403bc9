"""CPU: host-side logic of the frame pipeline (job tables, quadtree parents, CTU-row ownership) and the 2-rank
sharding path on the gloo backend."""
import os
import subprocess
import sys

import numpy as np

from vtm_amd import pipeline, synth
from vtm_amd.lib import TzJob  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_subshift_rule_matches_oracle(oracle):
    for w in (4, 8, 12, 16, 24, 32, 48, 64, 128):
        for h in (4, 8, 16, 32, 64, 128):
            assert pipeline.subshift_mode2(w, h) == oracle.vo_subshift_for_mode(w, h, 2)


def test_quadtree_levels_cover_picture_and_parents_enclose_children():
    lv = pipeline.quadtree_levels(416, 240)
    assert [l[0] for l in lv] == [128, 64, 32, 16, 8]
    for (s, xs, ys, parent), (ps, pxs, pys, _) in zip(lv[1:], lv[:-1]):
        ok = parent >= 0
        assert ((xs[ok] // ps * ps == pxs[parent[ok]]) & (ys[ok] // ps * ps == pys[parent[ok]])).all()
        # children without a parent only exist where the coarser block would cross the picture edge
        assert (((xs[~ok] // ps * ps + ps > 416) | (ys[~ok] // ps * ps + ps > 240))).all()
    s, xs, ys, _ = lv[-1]
    assert xs.size == (416 // 8) * (240 // 8)


def test_row_filter_partitions_the_picture():
    all_lv = pipeline.quadtree_levels(3840, 2160, sizes=(64,))
    parts = [pipeline.quadtree_levels(3840, 2160, sizes=(64,), row_filter=lambda r, k=k: r % 4 == k) for k in range(4)]
    assert sum(p[0][1].size for p in parts) == all_lv[0][1].size
    seen = set()
    for p in parts:
        for x, y in zip(p[0][1], p[0][2]):
            assert (x, y) not in seen
            seen.add((x, y))


def test_driver_job_tables_layout():
    """FrameHotPath builds its job tables on the host (numpy -> HBM): the uni-ME rows of a split-shape level, their transform units and the chroma jobs
    carry the positions, sizes and strides the C structs expect (checked through the ctypes mirrors on a CPU-resident table)."""
    import torch
    from vtm_amd.lib import MeJob, PredJob, TuJob
    W, H, rs = 256, 128, 640
    refs = ([(1000, rs)], [(500000, rs)])
    chroma = dict(org_off=(W * H, W * H + W * H // 4), org_stride=W // 2, refs=([(900000, 910000)], [(920000, 930000)]), ref_stride=320)
    hp = pipeline.FrameHotPath(None, torch, torch.device("cpu"), W, H, W, refs, ([96], [96]), sizes=((64, 64), (64, 32), (16, 8)), pocs=(2, [0], [4]), chroma=chroma)
    lv = {l["size"]: l for l in hp.levels}
    l = lv[(64, 32)]
    assert l["npu"] == (W // 64) * (H // 32) and (l["w"], l["h"], l["tw"], l["th"]) == (64, 32, 64, 32)
    raw = l["uni_jobs"].t.numpy()
    n = l["npu"]
    j = MeJob.from_buffer_copy(raw[n + 5].tobytes())          # list 1, PU 5 = (x 64, y 32)
    assert (j.puX, j.puY, j.width, j.height, j.orgOff, j.refOff, j.searchRange) == (64, 32, 64, 32, 32 * W + 64, 500000 + 32 * rs + 64, 96)
    t = TuJob.from_buffer_copy(l["tu"].t.numpy()[5].tobytes())
    assert (t.width, t.height, t.resiStride, t.resiOff) == (64, 32, 64, l["sb"] + 5 * 64 * 32)
    c = PredJob.from_buffer_copy(l["pred_final_c"].t.numpy()[n + 5].tobytes())   # the Cr job of PU 5
    assert (c.width, c.height, c.chroma, c.orgOff, c.predStride) == (32, 16, 1, chroma["org_off"][1] + 16 * (W // 2) + 32, 32)
    s8 = lv[(16, 8)]
    assert (s8["tw_c"], s8["th_c"]) == (8, 4) and s8["parent32"] is not None    # 16x8 PUs nest in the 64x32 level; their chroma TUs are 8x4
    par = s8["parent32"].numpy()
    k = int(np.nonzero((s8["xs"] == 80) & (s8["ys"] == 40))[0][0])
    assert (l["xs"][par[k]], l["ys"][par[k]]) == (64, 32)


def test_extend_plane_replicates_border():
    p = np.arange(12, dtype=np.int16).reshape(3, 4)
    buf, off, stride = synth.extend_plane(p, margin=5, align=8)
    ext = buf.reshape(-1, stride)
    assert ext[5, 5] == 0 and ext[0, 0] == 0 and ext[7, 8] == 11 and ext[12, 13] == 11 and buf[off + 1 * stride + 2] == 6


def test_two_rank_gloo_sharding():
    """world_size 2 on CPU/gloo: rank 0 broadcasts the reference planes, each rank builds the job table of its own CTU
    rows, and the union over ranks is the single-rank table."""
    script = os.path.join(ROOT, "tests", "gloo_worker.py")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", script], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GLOO_OK" in out.stdout


def test_two_rank_plane_exchange():
    """world_size 2 on CPU/gloo: the double-buffered asynchronous reference-plane broadcast bench.py uses for N > 1 (vtm_amd/exchange.py):
    every rank computes step k on picture k while picture k + 1 is already travelling."""
    script = os.path.join(ROOT, "tests", "gloo_exchange_worker.py")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29519", script], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "EXCHANGE_OK" in out.stdout


def test_ctu_bands_partition_and_balance():
    """raster-scan CTU ranges: contiguous, complete, balanced to one CTU; whole-row bands for comparison (17 rows over 8 ranks: 3 / 2 rows)"""
    b = pipeline.ctu_bands(3840, 2160, 8, unit="ctu")
    assert b[0][0] == 0 and b[-1][1] == 30 * 17 and all(x[1] == y[0] for x, y in zip(b[:-1], b[1:]))
    assert sorted({x[1] - x[0] for x in b}) == [63, 64]
    r = pipeline.ctu_bands(3840, 2160, 8, unit="row")
    assert sorted({(x[1] - x[0]) // 30 for x in r}) == [2, 3] and sum(x[1] - x[0] for x in r) == 510
    parts = [pipeline.quadtree_levels(3840, 2160, sizes=(128, 8), ctu_filter=pipeline.band_filter(3840, x)) for x in b]
    full = pipeline.quadtree_levels(3840, 2160, sizes=(128, 8))
    for li in (0, 1):
        assert sum(p[li][1].size for p in parts) == full[li][1].size
    # parents stay inside the band: a child's parent index refers to the band's own 128-level table
    for p in parts:
        (s0, xs0, ys0, _), (s1, xs1, ys1, par) = p
        ok = par >= 0
        assert ((xs1[ok] // 128 * 128 == xs0[par[ok]]) & (ys1[ok] // 128 * 128 == ys0[par[ok]])).all()


def test_two_rank_ctu_shards_and_result_gather():
    """world_size 2 on CPU/gloo: CTU bands per rank + the double-buffered result gather of bench.py --gpus N (vtm_amd/exchange.py:ResultGather)"""
    script = os.path.join(ROOT, "tests", "gloo_gather_worker.py")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29521", script], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GATHER_OK" in out.stdout
