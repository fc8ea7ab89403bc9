#!/usr/bin/env python3
"""GPU box: the encoder-level drop-in with a denser sampling of the hooked calls than the test uses (every `stride`-th call per table slot goes to the
device; default 7 instead of 41).  Prints the call / device / mismatch counts and whether bitstream and reconstruction equal the plain run's.
usage: python3 scripts/gpu_dropin_dense.py [stride]"""
import concurrent.futures as cf
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import enc_dropin   # noqa: E402

stride = int(sys.argv[1]) if len(sys.argv) > 1 else 7
W, H, FRAMES, QP = 192, 128, 5, 30
with tempfile.TemporaryDirectory() as d:
    yuv = os.path.join(d, "clip.yuv")
    enc_dropin.write_clip(yuv, W, H, FRAMES)
    with cf.ThreadPoolExecutor(2) as ex:
        plain = ex.submit(enc_dropin.encode, yuv, W, H, FRAMES, QP, os.path.join(d, "plain"))
        hooked = ex.submit(enc_dropin.encode, yuv, W, H, FRAMES, QP, os.path.join(d, "hip"), True, 23, stride, 256, (), 1100)
        st0, bits0, rec0 = plain.result()
        st1, bits1, rec1 = hooked.result()
print({"stride": stride, "calls": st1["calls"], "device": st1["device"], "mismatch": st1["mismatch"], "errors": st1["errors"],
       "bitstream_equal": bits1 == bits0, "reconstruction_equal": rec1 == rec0})
