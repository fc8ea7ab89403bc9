"""Makes tests/golden/pis_enc.npz (DATA ONLY: sample planes and job / result records): runs the REAL reference encoder of oracle/_ref/libvtmref.so on the small
random-access clip of the encoder tests with the predInterSearch hook in RECORD mode (oracle/ref_shim_pis.hpp, no device) and keeps every STRIDE-th CU's record.

    python tests/golden/gen_pis_golden.py [stride = 45] [ra | ldp | ldb | bcw]

ra: tests/data/enc_ra_gop4.cfg, 5 pictures -> pis_enc.npz; ldp / ldb: the low-delay structures (tests/data/enc_ld{p,b}_gop4.cfg, 6 pictures, four reference pictures,
SearchRange 64 without ASR) -> pis_enc_ldp.npz / pis_enc_ldb.npz; bcw: the random-access cfg with --BCW=1 --BcwFast=1 --AffineAmvr=1 --AffineAmvrEncOpt=1 as
cfg/encoder_randomaccess_vtm.cfg has them -> pis_enc_bcw.npz: every 4 * stride-th call at the default weight (their bi costs carry the weight-index bits) and every
(stride / 2)-th call at another CU-level weight (given uni rows, weighted targets, BcwFast's same-POC skip, the enforced bi mode).

Needs /root/reference (the reference is compiled in place by oracle/Makefile.ref); the .npz travels, the reference does not."""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import enc_dropin      # noqa: E402
import pis_golden      # noqa: E402


def main():
    stride = int(sys.argv[1]) if len(sys.argv) > 1 else 45
    structure = sys.argv[2] if len(sys.argv) > 2 else "ra"
    frames = 5 if structure in ("ra", "bcw") else 6
    cfg = None if structure in ("ra", "bcw") else os.path.join(os.path.dirname(HERE), "data", "enc_%s_gop4.cfg" % structure)
    extra = ("--BCW=1", "--BcwFast=1", "--AffineAmvr=1", "--AffineAmvrEncOpt=1") if structure == "bcw" else ()
    env = {"VTMREF_PIS_DUMP_STRIDE": str(stride)}
    if structure == "bcw":
        env = {"VTMREF_PIS_DUMP_STRIDE": str(4 * stride), "VTMREF_PIS_DUMP_BCW_STRIDE": str(max(1, stride // 2))}
    with tempfile.TemporaryDirectory() as tmp:
        yuv, dump = os.path.join(tmp, "clip.yuv"), os.path.join(tmp, "pis.bin")
        enc_dropin.write_clip(yuv, 192, 128, frames)
        st, bits, rec = enc_dropin.encode(yuv, 192, 128, frames, 30, os.path.join(tmp, "rec"), False, 2048 | 8, 1, 0, env=dict(env, VTMREF_PIS_DUMP=dump), cfg=cfg, extra=extra)
        assert st["rc"] == 0, st
        planes, recs = pis_golden.parse_dump(dump)
    out = os.path.join(HERE, "pis_enc.npz" if structure == "ra" else "pis_enc_%s.npz" % structure)
    pis_golden.save_npz(out, planes, recs)
    print("%d planes, %d records (of %d predInterSearch calls) -> %s (%d bytes); bitstream md5 %s" % (len(planes), len(recs), st["pis"]["calls"], out, os.path.getsize(out), bits))


if __name__ == "__main__":
    main()
