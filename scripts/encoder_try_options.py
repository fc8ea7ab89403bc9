"""GPU box: the predInterSearch hook (compare mode) on the small random-access clip with extra encoder options: counts of device / unsupported calls and mismatches.\n    python scripts/encoder_try_options.py --AffineAmvr=1 --CIIP=1 ..."""
import sys, os, json
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import enc_dropin
extra = tuple(sys.argv[1:])
yuv = "/tmp/try_clip.yuv"
enc_dropin.write_clip(yuv, 192, 128, 5)
st0, b0, r0 = enc_dropin.encode(yuv, 192, 128, 5, 30, "/tmp/try_plain", False, 2048 | 8, 1, 0, extra=extra)
st1, b1, r1 = enc_dropin.encode(yuv, 192, 128, 5, 30, "/tmp/try_cmp", True, 2048 | 128, 1, 0, extra=extra)
print("extra", extra)
print("plain pis", st0["pis"]["calls"], "compare pis", {k: st1["pis"][k] for k in ("calls", "device", "unsupported", "skipped", "replayFallback", "mismatch", "firstMismatch")}, "affine", st1["affine"], "errors", st1["errors"], st1.get("firstError"))
print("identical", b0 == b1 and r0 == r1)
