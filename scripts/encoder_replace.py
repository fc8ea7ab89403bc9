"""GPU box: the real reference encoder (oracle/_ref/libvtmref.so) plain vs REPLACE mode -- InterSearch::predInterSearch as one vtmhip_predInterSearch_batch_dev call per CU
(the reference's own glue replayed over the device tables, its members' bodies not run) and xAffineMotionEstimation on the device -- on a synthetic random-access clip.

    python scripts/encoder_replace.py WIDTH HEIGHT FRAMES [compare]      -> gpurun_out/encoder_replace_<W>x<H>.json

BASELINE metric (1), "encoded frames/sec": this is its honest number for a CU-at-a-time integration (hard parts 1 and 3)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import enc_dropin      # noqa: E402


def main():
    W, H, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    qp, tmp = 32, os.environ.get("TMPDIR", "/tmp")
    yuv = os.path.join(tmp, "replace_%dx%d.yuv" % (W, H))
    enc_dropin.write_clip(yuv, W, H, N)
    mask = 2048 | 128
    runs, t = {}, time.time()
    st0, bits0, rec0 = enc_dropin.encode(yuv, W, H, N, qp, os.path.join(tmp, "plain"), False, 2048 | 8, 1, 0, timeout=3000)      # the hook only times the member
    runs["plain_s"] = time.time() - t
    print("plain %.1f s" % runs["plain_s"], flush=True)
    t = time.time()
    st2, bits2, rec2 = enc_dropin.encode(yuv, W, H, N, qp, os.path.join(tmp, "rep"), True, mask, 1, 0, env={"VTMREF_REPLACE": "1"}, timeout=3000)
    runs["replace_s"] = time.time() - t
    print("replace %.1f s" % runs["replace_s"], st2["pis"], flush=True)
    out = {"clip": "%dx%d synthetic (vtm_amd/synth.py), %d pictures, QP %d, tests/data/enc_ra_gop4.cfg" % (W, H, N, qp), "bitstream_md5_plain": bits0, "bitstream_md5_replace": bits2,
           "identical_bitstream": bits0 == bits2 and rec0 == rec2, **runs, "encoded_fps_plain": N / runs["plain_s"], "encoded_fps_replace": N / runs["replace_s"],
           "speedup_replace_vs_plain": runs["plain_s"] / runs["replace_s"], "plain": st0["pis"], "replace": st2["pis"], "affine_replace": st2["affine"],
           "affine_seconds_replace": st2["affineSeconds"], "errors": st2["errors"]}
    in_member = sum(st0["pis"]["seconds"])
    out["predInterSearch_share_of_plain_run"] = in_member / runs["plain_s"]
    out["amdahl_bound_if_predInterSearch_were_free"] = 1.0 / (1.0 - in_member / runs["plain_s"])
    out["per_hook_seconds_replace"] = {"gather_and_final_compare": st2["pis"]["seconds"][0], "upload_device_download": st2["pis"]["seconds"][1],
                                       "reference_glue_over_the_tables_incl_affine_search_and_final_mc": st2["pis"]["seconds"][2], "xAffineMotionEstimation_device_calls": st2["affineSeconds"][1]}
    if "compare" in sys.argv:
        t = time.time()
        st1, bits1, rec1 = enc_dropin.encode(yuv, W, H, N, qp, os.path.join(tmp, "cmp"), True, mask, 1, 0, timeout=3000)
        out.update(compare_s=time.time() - t, compare=st1["pis"], affine_compare=st1["affine"], identical_bitstream_compare=bits1 == bits0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "encoder_replace_%dx%d.json" % (W, H)), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("identical_bitstream", "plain_s", "replace_s", "speedup_replace_vs_plain", "predInterSearch_share_of_plain_run")}))
    assert out["identical_bitstream"] and st2["pis"]["mismatch"] == [0] * 6 and st2["pis"]["replayFallback"] == 0


if __name__ == "__main__":
    main()
