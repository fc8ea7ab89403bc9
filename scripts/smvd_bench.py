"""Per-size timing of the SMVD search (vtmhip_smvd_batch_dev, op VTMHIP_SMVD_SEARCH): every PU of a tiling of the picture, AMVP lists = a vector near the scene's
motion and zero, two start vectors.  Prints ms per launch, PUs/s and candidate evaluations are data dependent (the search stops when no direction improves)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import me_util
    from vtm_amd.device import Context
    from vtm_amd.lib import PicParams, SmvdJob
    W, H = (int(v) for v in os.environ.get("SMVD_BENCH_SIZE", "3840x2160").split("x"))
    scene = me_util.SmvdScene(W, H)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_cur = ctx.to_device(scene.cur)
    d_ref = ctx.to_device(np.concatenate([scene.ref_buf.reshape(-1), scene.ref_buf2.reshape(-1)]))
    pic = PicParams(W, H, 128, 10, 0)
    out = {}
    for (w, h) in [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (16, 8), (32, 16), (64, 32)]:
        xs, ys = np.meshgrid(np.arange(0, W - w + 1, w), np.arange(0, H - h + 1, h))
        xs, ys = xs.reshape(-1), ys.reshape(-1)
        n = xs.size
        dt = np.dtype(SmvdJob)
        j = np.zeros(n, dt)
        j["orgOff"], j["orgStride"] = ys * W + xs, W
        pos = scene.ref_off + ys * scene.ref_stride + xs
        j["refOff"][:, 0], j["refOff"][:, 1] = pos, scene.ref_buf.size + pos
        j["refStride"][:] = scene.ref_stride
        j["puX"], j["puY"], j["width"], j["height"] = xs, ys, w, h
        j["imv"], j["useSatd"], j["clipBiPred"], j["bcwWeightTar"] = 0, 1, 0, 4
        j["numCand"][:] = 2
        j["cand"][:, 0, 0] = (-8, 4)
        j["cand"][:, 1, 0] = (8, -4)
        j["mvpIdxBits"][:] = 1
        j["numStart"], j["numFixed"], j["modeBits"], j["motionLambda"] = 2, 2, 6, 8.0
        j["starts"][:, 0] = (-6, 2)
        j["starts"][:, 1] = (-16, 0)
        d_jobs = ctx.to_device(j.view(np.uint8))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ctx.smvd_batch(pic, d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, w, h, 3, uniform=True)
        torch.cuda.synchronize()
        reps = 3
        ev[0].record()
        for _ in range(reps):
            ctx.smvd_batch(pic, d_cur.ptr, d_ref.ptr, d_jobs.ptr, n, w, h, 3, uniform=True)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        res = np.frombuffer(d_jobs.to_host(np.uint8).tobytes(), dt)
        out["%dx%d" % (w, h)] = dict(pus=int(n), ms=round(ms, 3), mpus_per_s=round(n / ms / 1e3, 2), checksum=int(res["cost"].sum() % (1 << 31)),
                                    moved=int(np.count_nonzero((res["mvCur"] != res["predSym"][:, 0]).any(1))))
        print("%dx%d" % (w, h), out["%dx%d" % (w, h)], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
