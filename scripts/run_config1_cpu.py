#!/usr/bin/env python3
"""BASELINE.json config 1 (plumbing, no GPU): the reference encoder compiled in place (oracle/_ref/libvtmref.so) encodes the 416x240 synthetic clip with
encoder_intra_vtm.cfg, QP 37, 8 frames (-ts 1: the cfg's TemporalSubsampleRatio 8 would otherwise code one frame) on the host CPU.  Runs only where
/root/reference exists (this container); prints a JSON summary: wall time, frames/s, bitstream / reconstruction MD5 and the encoder's own summary lines.
usage: python3 scripts/run_config1_cpu.py > profiles/rNN_config1_intra_cpu.json"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import enc_dropin   # noqa: E402


def main():
    cfg = "/root/reference/cfg/encoder_intra_vtm.cfg"
    if not os.path.exists(cfg) or not os.path.exists(enc_dropin.REF_SO):
        sys.exit("needs /root/reference and oracle/_ref/libvtmref.so (python -c 'import __graft_entry__ as g; g.build()')")
    w, h, frames, qp = 416, 240, 8, 37
    with tempfile.TemporaryDirectory() as d:
        yuv = os.path.join(d, "clip.yuv")
        enc_dropin.write_clip(yuv, w, h, frames, seed=1234)
        enc_dropin.CFG = cfg
        t0 = time.perf_counter()
        st, md5_bits, md5_rec = enc_dropin.encode(yuv, w, h, frames, qp, os.path.join(d, "out"), hip=False, extra=["-ts", "1", "--InputBitDepth=10", "-fr", "30"], timeout=3000)
        dt = time.perf_counter() - t0
        size = os.path.getsize(os.path.join(d, "out.bin"))
    tail = [l for l in st["log_tail"].splitlines() if l.strip()]
    print(json.dumps({"config": "encoder_intra_vtm.cfg 416x240 10-bit QP37, 8 frames, reference encoder on one host core (AVX2), no GPU", "seconds": round(dt, 2),
                      "frames_per_s": round(frames / dt, 4), "bitstream_bytes": size, "bitstream_md5": md5_bits, "reconstruction_md5": md5_rec,
                      "dispatch_table_calls": st["calls"], "encoder_log_tail": tail[-8:]}, indent=1))


if __name__ == "__main__":
    main()
