#!/usr/bin/env python3
"""SATD-8x8 grid micro-benchmark for one workgroup shape (VTMHIP_SATD_VARIANT, read once per process): prints G pairs/s."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtm_amd import synth   # noqa: E402
from vtm_amd.device import Context   # noqa: E402

W, H = 3840, 2160
ctx = Context(0)
fr = synth.gen_frames(W, H, 2)
cur = np.ascontiguousarray(fr[1])
ref, roff, rs = synth.extend_plane(fr[0], margin=160)
d_cur, d_ref = ctx.to_device(cur), ctx.to_device(ref)
nb = (W // 8) * (H // 8)
d_out = ctx.alloc(4 * nb * 81)
for _ in range(3):
    ctx.satd8_grid(d_cur.ptr, W, d_ref.ptr + 2 * roff, rs, W, H, 4, d_out.ptr)
ctx.sync()
ctx.timer_start()
for _ in range(20):
    ctx.satd8_grid(d_cur.ptr, W, d_ref.ptr + 2 * roff, rs, W, H, 4, d_out.ptr)
ms = ctx.timer_stop_ms() / 20
chk = int(d_out.to_host(np.uint32).astype(np.uint64).sum())
print("variant", os.environ.get("VTMHIP_SATD_VARIANT", "0"), "ms %.4f" % ms, "G pairs/s %.2f" % (nb * 81 / ms / 1e6), "checksum", chk)
