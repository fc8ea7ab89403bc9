"""GPU soak: the level-order driver with every stage switched on (chroma, BDOF, SMVD, affine, transform skip, split shapes) on pictures and reference structures other
than the test suite's, PU by PU against the CPU chain through the oracle (and through the real reference members when oracle/_ref is present)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                            # noqa: E402
import oracle_lib as ol                 # noqa: E402
import test_gpu_pis as TP               # noqa: E402
from vtm_amd.device import Context      # noqa: E402
from vtm_amd.pipeline import FrameHotPath   # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    R = ol.ref() if ol.have_ref() else None
    total = 0
    for (W, H, pocs0, pocs1, cur, qp, sizes, smvd, affine, ts, hard) in (
            (384, 256, [1], [5], 3, 27, (128, 64, 32, 16, 8), (0, 0), True, True, True),
            (320, 192, [4, 2], [6, 8], 5, 37, (64, (64, 32), 32, (32, 16), 16, (16, 8), 8), (0, 0), False, False, True),
            (256, 256, [2, 0], [4, 6], 3, 22, (128, 64, 32, 16, 8), (0, 0), True, False, False),
            (448, 128, [3, 2, 1, 0], [], 4, 32, (64, 32, 16, 8), None, True, True, True)):
        cur_np, dpb_np, refs, sr, cur_d, dpb, ch_dev, ch_cpu = TP.make_scene(torch, dev, W, H, pocs0, pocs1, cur, chroma=True, hard=hard)
        if not pocs1:
            sr = ([64] * len(pocs0), [])
        pocs = (cur, pocs0, pocs1)
        hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, sr, motion_lambda=8.0, qp=qp, sizes=sizes, pocs=pocs, chroma=ch_dev, transform_skip=ts, affine=affine,
                          low_delay=not pocs1, smvd=smvd)
        hp.run(cur_d.data_ptr(), dpb.data_ptr())
        torch.cuda.synchronize()
        stats = {}
        TP.check(hp, cur_np, dpb_np, refs, sr, W, H, 8.0, qp, R, per_level=40, min_checked=40, pocs=pocs, chroma=ch_cpu, stats=stats, affine=affine, low_delay=not pocs1,
                 smvd=hp.smvd)
        total += sum(min(40, l["npu"]) for l in hp.levels)
        print("picture %dx%d refs %s+%s qp %d: ok" % (W, H, pocs0, pocs1, qp), stats, flush=True)
    print("soak: ~%d PUs compared, 0 mismatches (an assertion stops the run otherwise)" % total)


if __name__ == "__main__":
    main()
