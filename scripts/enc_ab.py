import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import enc_dropin
yuv = "/tmp/ab_clip.yuv"
enc_dropin.write_clip(yuv, 192, 128, 5)
for fuse in ("0", "1"):
    st, b, r = enc_dropin.encode(yuv, 192, 128, 5, 30, "/tmp/ab_%s" % fuse, True, 2048 | 128, 1, 0, env={"VTMHIP_MEST_FUSE": fuse})
    print("fuse", fuse, st["pis"]["mismatch"], st["pis"]["replayFallback"], st["pis"]["firstMismatch"], b, flush=True)
