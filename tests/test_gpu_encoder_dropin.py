"""Encoder-level drop-in: the REAL reference encoder (oracle/_ref/libvtmref.so, VTM 9.3 EncApp/EncLib) encodes a small
random-access clip with its own dispatch tables -- RdCost::m_afpDistortFunc (SAD/HAD/SSE), InterpolationFilter::m_filterHor/
m_filterVer/m_filterCopy, fastFwdTrans/fastInvTrans, g_pelBufOP.addAvg/removeHighFreq, the affine Sobel / normal-equation
pointers -- routed through the C ABI of libvtmhip.so (oracle/ref_shim_enc.cpp, the
trampolines of INTEGRATION.md section 2).  Every routed call is compared with the reference's own function on the same
arguments and its device result replaces the reference's; the bitstream and the reconstruction must equal the plain run's."""
import concurrent.futures as cf
import os

import pytest

import enc_dropin

pytestmark = pytest.mark.gpu

W, H, FRAMES, QP = 192, 128, 5, 30


@pytest.mark.skipif(not os.path.exists(enc_dropin.REF_SO), reason="oracle/_ref/libvtmref.so not built (needs /root/reference)")
def test_reference_encoder_with_device_dispatch(tmp_path):
    yuv = str(tmp_path / "clip.yuv")
    enc_dropin.write_clip(yuv, W, H, FRAMES)
    with cf.ThreadPoolExecutor(2) as ex:
        plain = ex.submit(enc_dropin.encode, yuv, W, H, FRAMES, QP, str(tmp_path / "plain"))
        hooked = ex.submit(enc_dropin.encode, yuv, W, H, FRAMES, QP, str(tmp_path / "hip"), True, 23, 41, 256)
        st0, bits0, rec0 = plain.result()
        st1, bits1, rec1 = hooked.result()
    print("dropin:", {k: st1[k] for k in ("calls", "device", "mismatch", "errors")})
    assert st0["rc"] == 0 and st1["rc"] == 0
    assert st1["errors"] == 0, st1
    assert st1["mismatch"] == [0, 0, 0, 0], st1
    # every family really went to the device: distortion, interpolation, transforms > 10^5 calls each, buffer ops + affine gradients > 5000
    assert min(st1["device"][:3]) > 100000 and st1["device"][3] > 5000, st1
    assert bits1 == bits0 and rec1 == rec0


@pytest.mark.skipif(not os.path.exists(enc_dropin.REF_SO), reason="oracle/_ref/libvtmref.so not built (needs /root/reference)")
def test_reference_encoder_with_batched_hooks(tmp_path, monkeypatch):
    """SURVEY.md Appendix B inside the real encoder: InterSearch::xMotionEstimation replaced by ONE vtmhip_xMotionEstimation_batch_dev call per
    invocation (start candidates, TZ / exhaustive search, fractional or AMVR refinement, rate re-weighting: hooks B1-B6) and the MTS candidate loop of
    TrQuant::transformNxN( trModes ) by one batch of forward transforms + vtmhip_mts_select2 (hook B8); InterSearch::xEstimateMvPredAMVP's template costs and
    selection by vtmhip_xEstimateMvPredAMVP_batch_dev (hook B7); the three SMVD members (xGetSymmetricCost, xSymmetricMotionEstimation, symmvdCheckBestMvp) by the
    ops of vtmhip_smvd_batch_dev.  The members are intercepted at link level
    (oracle/Makefile.ref weakens the two reference symbols; oracle/ref_shim_enc.cpp holds the strong definitions).  Every 3rd supported call goes to
    the device, its results replace the reference's and are compared with them; bitstream and reconstruction must equal the plain run's."""
    import time
    yuv = str(tmp_path / "clip.yuv")
    enc_dropin.write_clip(yuv, W, H, FRAMES)
    monkeypatch.setenv("VTMREF_HOOK_STRIDE", "3")
    t0 = time.time()
    st0, bits0, rec0 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "plain"), extra=("--LFNST=1",))
    t1 = time.time()
    st1, bits1, rec1 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "hooks"), True, 32 | 64 | 128 | 256 | 512 | 1024, 1000000, 64, extra=("--LFNST=1",))
    t2 = time.time()
    print("batched hooks:", {k: st1[k] for k in ("hookCalls", "hookDevice", "hookMismatch", "hookUnsupported", "affine", "lfnst", "amvp", "smvd", "errors")}, "plain %.1f s, hooked %.1f s" % (t1 - t0, t2 - t1))
    assert st0["rc"] == 0 and st1["rc"] == 0
    assert st1["errors"] == 0, st1
    assert st1["hookMismatch"] == [0, 0], st1
    assert st1["hookDevice"][0] > 5000 and st1["hookDevice"][1] > 50000, st1
    assert st1["affine"][2] == 0 and st1["affine"][1] > 500, st1          # xAffineMotionEstimation: [calls, on the device, mismatches, unsupported]
    assert st1["lfnst"][2] == [0, 0] and min(st1["lfnst"][1]) > 1000, st1  # xFwdLfnst / xInvLfnst: [[calls], [on the device], [mismatches]]
    assert st1["amvp"][2] == 0 and st1["amvp"][1] > 5000, st1               # xEstimateMvPredAMVP: [calls, on the device, mismatches, unsupported]
    # xGetSymmetricCost / xSymmetricMotionEstimation / symmvdCheckBestMvp: [[calls], [on the device], [mismatches], unsupported]
    assert st1["smvd"][2] == [0, 0, 0] and min(st1["smvd"][1]) > 50, st1
    assert bits1 == bits0 and rec1 == rec0
    out = os.path.join(enc_dropin.ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "encoder_batched_hooks.txt"), "w") as f:
            f.write("clip %dx%d, %d pictures, QP %d; plain run %.1f s, hooked run %.1f s (every 3rd supported call of xMotionEstimation / xEstimateMvPredAMVP / the SMVD members / transformNxN(trModes) / xAffineMotionEstimation / xFwdLfnst / xInvLfnst on the device)\n%r\nbitstream md5 %s (plain %s)\n"
                    % (W, H, FRAMES, QP, t1 - t0, t2 - t1, {k: st1[k] for k in ("hookCalls", "hookDevice", "hookMismatch", "hookUnsupported", "affine", "lfnst", "amvp", "smvd", "errors")}, bits1, bits0))


@pytest.mark.skipif(not os.path.exists(enc_dropin.REF_SO), reason="oracle/_ref/libvtmref.so not built (needs /root/reference)")
def test_reference_encoder_predInterSearch_one_call_per_cu(tmp_path):
    """InterSearch::predInterSearch inside the real encoder as ONE vtmhip_predInterSearch_batch_dev call per CU (oracle/ref_shim_pis.hpp): the PU's real AMVP lists,
    m_uniMvList, block-vector cache hits and FastMEForGenBLowDelay copies go in, every (list, refIdx) search, the bi refinement, the SMVD block and the decision come back behind
    one synchronisation, and the reference's own predInterSearch runs over the downloaded tables (its member calls are served from them; every served call checks that the
    reference's arguments are the ones the device's glue derived).  Compare mode: the served members also run the reference's code (results compared), and what the member
    leaves in pu -- interDir, mv, mvd, mvpIdx, refIdx, smvdMode, best translational cost -- is compared with the device's own decision record.  Replace mode
    (VTMREF_REPLACE=1): the members' bodies do not run at all (xAffineMotionEstimation included).  Both: bitstream and reconstruction equal the plain run's."""
    import json
    import time
    yuv = str(tmp_path / "clip.yuv")
    enc_dropin.write_clip(yuv, W, H, FRAMES)
    mask = 2048 | 128
    t0 = time.time()
    st0, bits0, rec0 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "plain"), False, 2048 | 8, 1, 0)      # plain run; the hook only times the member
    t1 = time.time()
    st1, bits1, rec1 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "cmp"), True, mask, 1, 0)
    t2 = time.time()
    st2, bits2, rec2 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "rep"), True, mask, 1, 0, env={"VTMREF_REPLACE": "1"})
    t3 = time.time()
    print("predInterSearch compare:", st1["pis"], st1["affine"], "replace:", st2["pis"], st2["affine"], "plain %.1f s compare %.1f s replace %.1f s" % (t1 - t0, t2 - t1, t3 - t2))
    for st in (st0, st1, st2):
        assert st["rc"] == 0 and st["errors"] == 0, st
    for st in (st1, st2):
        assert st["pis"]["device"] >= 5000, st["pis"]
        assert st["pis"]["mismatch"] == [0] * 6 and st["pis"]["replayFallback"] == 0, st["pis"]
        assert st["affine"][2] == 0 and st["affine"][1] > 1000, st["affine"]
    assert bits1 == bits0 and rec1 == rec0
    assert bits2 == bits0 and rec2 == rec0
    out = os.path.join(enc_dropin.ROOT, "gpurun_out")
    if os.path.isdir(out):
        in_member = sum(st0["pis"]["seconds"])
        p = in_member / (t1 - t0)
        json.dump({"clip": "%dx%d, %d pictures, QP %d, tests/data/enc_ra_gop4.cfg" % (W, H, FRAMES, QP), "bitstream_md5": bits0, "identical_bitstream": True,
                   "plain_s": t1 - t0, "compare_s": t2 - t1, "replace_s": t3 - t2, "speedup_replace_vs_plain": (t1 - t0) / (t3 - t2),
                   "predInterSearch_share_of_plain_run": p, "amdahl_bound_if_predInterSearch_were_free": 1.0 / (1.0 - p),
                   "plain": st0["pis"], "compare": st1["pis"], "replace": st2["pis"], "affine_compare": st1["affine"], "affine_replace": st2["affine"],
                   "affine_seconds_compare": st1["affineSeconds"], "affine_seconds_replace": st2["affineSeconds"],
                   "device_us_per_cu_replace": 1e6 * st2["pis"]["seconds"][1] / max(1, st2["pis"]["device"]),
                   "host_us_per_cu_in_the_plain_run": 1e6 * in_member / max(1, st0["pis"]["calls"])},
                  open(os.path.join(out, "encoder_replace_192x128.json"), "w"), indent=1)


@pytest.mark.skipif(not os.path.exists(enc_dropin.REF_SO), reason="oracle/_ref/libvtmref.so not built (needs /root/reference)")
@pytest.mark.parametrize("structure", ["ldp", "ldb"])
def test_reference_encoder_predInterSearch_low_delay(tmp_path, structure):
    """The same one-call-per-CU hook on BASELINE config 2's coding structures (tests/data/enc_ld{p,b}_gop4.cfg, SearchRange 64 without ASR, four reference pictures):
    low-delay P -- P slices, list 0 only (InterSearch.cpp:2363 numRefDir 1, the uni decision without a bi stage); low-delay B -- both lists hold the SAME four pictures:
    every list-1 row is a FastMEForGenBLowDelay copy (:2391-2404), mvd_l1_zero pictures (:2477-2522), no SMVD (needs references on both sides).  Replace mode, bitstream
    and reconstruction equal the plain run's."""
    frames = 6
    yuv = str(tmp_path / "clip.yuv")
    enc_dropin.write_clip(yuv, W, H, frames)
    cfg = os.path.join(enc_dropin.ROOT, "tests", "data", "enc_%s_gop4.cfg" % structure)
    st0, bits0, rec0 = enc_dropin.encode(yuv, W, H, frames, QP, str(tmp_path / "plain"), False, 2048 | 8, 1, 0, cfg=cfg)
    st2, bits2, rec2 = enc_dropin.encode(yuv, W, H, frames, QP, str(tmp_path / "rep"), True, 2048 | 128, 1, 0, env={"VTMREF_REPLACE": "1"}, cfg=cfg)
    print(structure, "plain:", st0["pis"], "replace:", st2["pis"], st2["affine"])
    assert st0["rc"] == 0 and st2["rc"] == 0 and st2["errors"] == 0, st2
    assert st2["pis"]["calls"] == st0["pis"]["calls"] and st2["pis"]["device"] >= 5000, st2["pis"]
    assert st2["pis"]["unsupported"] == 0 and st2["pis"]["mismatch"] == [0] * 6 and st2["pis"]["replayFallback"] == 0, st2["pis"]
    assert st2["affine"][2] == 0, st2["affine"]
    assert bits2 == bits0 and rec2 == rec0
    out = os.path.join(enc_dropin.ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        json.dump({"clip": "%dx%d, %d pictures, QP %d, tests/data/enc_%s_gop4.cfg" % (W, H, frames, QP, structure), "bitstream_md5": bits0, "identical_bitstream": True,
                   "plain": st0["pis"], "replace": st2["pis"], "affine_replace": st2["affine"]}, open(os.path.join(out, "encoder_replace_%s_192x128.json" % structure), "w"), indent=1)


@pytest.mark.skipif(not os.path.exists(enc_dropin.REF_SO), reason="oracle/_ref/libvtmref.so not built (needs /root/reference)")
def test_reference_encoder_predInterSearch_8bit_internal(tmp_path):
    """The same hook with InternalBitDepth 8 (the random-access cfg otherwise): the interpolation shifts and offsets, the packed SATD paths and the clipping ranges of
    8-bit samples inside the real encoder.  Replace mode; bitstream and reconstruction equal the plain run's."""
    yuv = str(tmp_path / "clip.yuv")
    enc_dropin.write_clip(yuv, W, H, FRAMES)
    extra = ("--InternalBitDepth=8",)
    st0, bits0, rec0 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "plain"), False, 2048 | 8, 1, 0, extra=extra)
    st2, bits2, rec2 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "rep"), True, 2048 | 128, 1, 0, extra=extra, env={"VTMREF_REPLACE": "1"})
    print("8-bit plain:", st0["pis"], "replace:", st2["pis"], st2["affine"])
    assert st0["rc"] == 0 and st2["rc"] == 0 and st2["errors"] == 0, st2
    assert st2["pis"]["calls"] == st0["pis"]["calls"] and st2["pis"]["device"] >= 5000, st2["pis"]
    assert st2["pis"]["unsupported"] == 0 and st2["pis"]["mismatch"] == [0] * 6 and st2["pis"]["replayFallback"] == 0, st2["pis"]
    assert st2["affine"][2] == 0, st2["affine"]
    assert bits2 == bits0 and rec2 == rec0


@pytest.mark.skipif(not os.path.exists(enc_dropin.REF_SO), reason="oracle/_ref/libvtmref.so not built (needs /root/reference)")
def test_reference_encoder_predInterSearch_with_bcw_enabled(tmp_path):
    """BCW : 1, BcwFast : 1, AffineAmvr : 1 as in cfg/encoder_randomaccess_vtm.cfg: the encoder then calls predInterSearch once per CU-level weight.  Every call goes to the
    device (round 4): the default-weight calls with the weight-index bits in their bi costs (InterSearch.cpp:2594, 2622, 2780), the calls at another weight with their uni rows
    GIVEN (xReadBufferedUniMv :7677-7697), the list with the smaller weight refined against the weighted target (removeWeightHighFreq, :2556-2559, 3320-3326), the distortion
    weight |w| / 8 (:7666-7676), BcwFast's same-POC skip (:2588-2593), the enforced bi mode (:2843-2847) and the weighted SMVD block; xAffineMotionEstimation's bi calls under a
    weight run on the device too.  Compare mode (every served member also runs the reference's code) and replace mode; bitstream and reconstruction equal the plain run's."""
    yuv = str(tmp_path / "clip.yuv")
    enc_dropin.write_clip(yuv, W, H, FRAMES)
    extra = ("--BCW=1", "--BcwFast=1", "--AffineAmvr=1", "--AffineAmvrEncOpt=1")
    st0, bits0, rec0 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "plain"), False, 2048 | 8, 1, 0, extra=extra)
    st1, bits1, rec1 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "cmp"), True, 2048 | 128, 1, 0, extra=extra)
    st2, bits2, rec2 = enc_dropin.encode(yuv, W, H, FRAMES, QP, str(tmp_path / "rep"), True, 2048 | 128, 1, 0, extra=extra, env={"VTMREF_REPLACE": "1"})
    print("BCW plain:", st0["pis"], "compare:", st1["pis"], st1["affine"], "replace:", st2["pis"], st2["affine"])
    assert st0["rc"] == 0
    for st in (st1, st2):
        assert st["rc"] == 0 and st["errors"] == 0, st
        assert st["pis"]["calls"] == st0["pis"]["calls"] and st["pis"]["device"] >= 5000 and st["pis"]["unsupported"] == 0, st["pis"]
        assert st["pis"]["device"] + st["pis"]["skipped"] == st["pis"]["calls"], st["pis"]
        assert st["pis"]["mismatch"] == [0] * 6 and st["pis"]["replayFallback"] == 0, st["pis"]
        assert st["affine"][2] == 0 and st["affine"][3] == 0, st["affine"]
    assert bits1 == bits0 and rec1 == rec0
    assert bits2 == bits0 and rec2 == rec0
    out = os.path.join(enc_dropin.ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        json.dump({"clip": "%dx%d, %d pictures, QP %d, tests/data/enc_ra_gop4.cfg + %s" % (W, H, FRAMES, QP, " ".join(extra)), "bitstream_md5": bits0, "identical_bitstream": True,
                   "plain": st0["pis"], "compare": st1["pis"], "replace": st2["pis"], "affine_compare": st1["affine"], "affine_replace": st2["affine"]},
                  open(os.path.join(out, "encoder_replace_bcw_192x128.json"), "w"), indent=1)


FULL_CFG = os.path.join(enc_dropin.ROOT, "tests", "data", "enc_ra_full.cfg")


@pytest.mark.skipif(not os.path.exists(enc_dropin.REF_SO), reason="oracle/_ref/libvtmref.so not built (needs /root/reference)")
def test_reference_encoder_predInterSearch_full_random_access_cfg(tmp_path):
    """The hook under the reference's OWN random-access operating point (tests/data/enc_ra_full.cfg = the values of cfg/encoder_randomaccess_vtm.cfg: GOP 16 with its reference
    lists, SearchRange 384 / MinSearchWindow 96 with ASR, MaxMTTHierarchyDepth 3, BCW + BcwFast, AffineAmvr + AffineAmvrEncOpt, CIIP, LFNST, ISP, MIP, LMCS, JointCbCr, MMVD ... all
    on): 416x240, 17 pictures (one whole GOP + the intra picture), QP 32, replace mode -- every predInterSearch call with a translational part and every
    xAffineMotionEstimation call on the MI355X, none left to the host, bitstream and reconstruction equal the plain run's."""
    import json
    import time
    w, h, frames, qp = 416, 240, 17, 32
    yuv = str(tmp_path / "clip.yuv")
    enc_dropin.write_clip(yuv, w, h, frames)
    t0 = time.time()
    st0, bits0, rec0 = enc_dropin.encode(yuv, w, h, frames, qp, str(tmp_path / "plain"), False, 2048 | 8, 1, 0, cfg=FULL_CFG)
    t1 = time.time()
    st2, bits2, rec2 = enc_dropin.encode(yuv, w, h, frames, qp, str(tmp_path / "rep"), True, 2048 | 128, 1, 0, cfg=FULL_CFG, env={"VTMREF_REPLACE": "1"})
    t2 = time.time()
    print("full RA cfg plain %.1f s:" % (t1 - t0), st0["pis"], "replace %.1f s:" % (t2 - t1), st2["pis"], st2["affine"])
    assert st0["rc"] == 0 and st2["rc"] == 0 and st2["errors"] == 0, st2
    assert st2["pis"]["calls"] == st0["pis"]["calls"] and st2["pis"]["device"] >= 100000, st2["pis"]
    assert st2["pis"]["unsupported"] == 0 and st2["pis"]["device"] + st2["pis"]["skipped"] == st2["pis"]["calls"], st2["pis"]
    assert st2["pis"]["mismatch"] == [0] * 6 and st2["pis"]["replayFallback"] == 0, st2["pis"]
    assert st2["affine"][2] == 0 and st2["affine"][3] == 0 and st2["affine"][1] > 10000, st2["affine"]
    assert bits2 == bits0 and rec2 == rec0
    out = os.path.join(enc_dropin.ROOT, "gpurun_out")
    if os.path.isdir(out):
        in_member = sum(st0["pis"]["seconds"])
        json.dump({"clip": "%dx%d, %d pictures, QP %d, tests/data/enc_ra_full.cfg (the values of cfg/encoder_randomaccess_vtm.cfg)" % (w, h, frames, qp), "bitstream_md5": bits0,
                   "identical_bitstream": True, "plain_s": t1 - t0, "replace_s": t2 - t1, "speedup_replace_vs_plain": (t1 - t0) / (t2 - t1),
                   "predInterSearch_share_of_plain_run": in_member / (t1 - t0), "plain": st0["pis"], "replace": st2["pis"], "affine_replace": st2["affine"],
                   "affine_seconds_replace": st2["affineSeconds"], "device_us_per_cu_replace": 1e6 * st2["pis"]["seconds"][1] / max(1, st2["pis"]["device"]),
                   "host_us_per_cu_in_the_plain_run": 1e6 * in_member / max(1, st0["pis"]["calls"])}, open(os.path.join(out, "encoder_replace_full_cfg_416x240.json"), "w"), indent=1)


@pytest.mark.skipif(not os.path.exists(enc_dropin.REF_SO), reason="oracle/_ref/libvtmref.so not built (needs /root/reference)")
def test_reference_encoder_predInterSearch_1080p_replace(tmp_path):
    """BASELINE metric (1) at the size of configs 2-3: 1920x1080, 3 pictures (I + two B), QP 32, tests/data/enc_ra_gop4.cfg, the hook in replace mode against the plain
    encoder -- every predInterSearch call and every xAffineMotionEstimation call on the MI355X, identical bitstream and reconstruction; wall times, the per-hook split and the
    Amdahl bound go to gpurun_out/encoder_replace_1920x1080.json (profiles/r04_encoder_replace_1920x1080.json)."""
    import json
    import time
    w, h, frames, qp = 1920, 1080, 3, 32
    yuv = str(tmp_path / "clip.yuv")
    enc_dropin.write_clip(yuv, w, h, frames)
    t0 = time.time()
    st0, bits0, rec0 = enc_dropin.encode(yuv, w, h, frames, qp, str(tmp_path / "plain"), False, 2048 | 8, 1, 0, timeout=3000)
    t1 = time.time()
    st2, bits2, rec2 = enc_dropin.encode(yuv, w, h, frames, qp, str(tmp_path / "rep"), True, 2048 | 128, 1, 0, env={"VTMREF_REPLACE": "1"}, timeout=3000)
    t2 = time.time()
    print("1080p plain %.1f s:" % (t1 - t0), st0["pis"], "replace %.1f s:" % (t2 - t1), st2["pis"], st2["affine"])
    assert st0["rc"] == 0 and st2["rc"] == 0 and st2["errors"] == 0, st2
    assert st2["pis"]["calls"] == st0["pis"]["calls"] and st2["pis"]["device"] >= 10000, st2["pis"]
    assert st2["pis"]["unsupported"] == 0 and st2["pis"]["mismatch"] == [0] * 6 and st2["pis"]["replayFallback"] == 0, st2["pis"]
    assert st2["affine"][2] == 0 and st2["affine"][3] == 0, st2["affine"]
    assert bits2 == bits0 and rec2 == rec0
    out = os.path.join(enc_dropin.ROOT, "gpurun_out")
    if os.path.isdir(out):
        in_member = sum(st0["pis"]["seconds"])
        dev = st2["pis"]["seconds"][1]
        json.dump({"clip": "%dx%d synthetic (vtm_amd/synth.py), %d pictures, QP %d, tests/data/enc_ra_gop4.cfg" % (w, h, frames, qp), "bitstream_md5": bits0, "identical_bitstream": True,
                   "plain_s": t1 - t0, "replace_s": t2 - t1, "encoded_fps_plain": frames / (t1 - t0), "encoded_fps_replace": frames / (t2 - t1),
                   "speedup_replace_vs_plain": (t1 - t0) / (t2 - t1), "predInterSearch_share_of_plain_run": in_member / (t1 - t0),
                   "amdahl_bound_if_predInterSearch_were_free": 1.0 / (1.0 - in_member / (t1 - t0)), "plain": st0["pis"], "replace": st2["pis"], "affine_replace": st2["affine"],
                   "per_hook_seconds_replace": {"gather_and_final_compare": st2["pis"]["seconds"][0], "upload_device_download": dev,
                                                "reference_glue_over_the_tables_incl_affine_search_and_final_mc": st2["pis"]["seconds"][2],
                                                "xAffineMotionEstimation_device_calls": st2["affineSeconds"][1]},
                   "device_us_per_cu": 1e6 * dev / max(1, st2["pis"]["device"]), "host_us_per_cu_in_the_plain_run": 1e6 * in_member / max(1, st0["pis"]["calls"])},
                  open(os.path.join(out, "encoder_replace_1920x1080.json"), "w"), indent=1)


@pytest.mark.skipif(not os.path.exists(enc_dropin.REF_SO), reason="oracle/_ref/libvtmref.so not built (needs /root/reference)")
def test_reference_encoder_intra_preselection_batched(tmp_path):
    """SURVEY.md 8(f) row 4, first part: the SATD pre-selection of IntraSearch::estIntraPredLumaQT (IntraSearch.cpp:549-592) as ONE vtmhip_intra_cand_cost_batch_dev call per CU
    inside the real encoder (oracle/ref_shim_intra.hpp): the 35 first-round predictors (formed by the reference's own predIntraAng) against the original block -> 35 SADs + 35
    SATDs; the distFunc calls of the member's loop are served from the batch (compare mode: and checked against the reference's functions; replace mode: served only)."""
    yuv = str(tmp_path / "clip.yuv")
    enc_dropin.write_clip(yuv, W, H, 3)
    st0, bits0, rec0 = enc_dropin.encode(yuv, W, H, 3, QP, str(tmp_path / "plain"))
    st1, bits1, rec1 = enc_dropin.encode(yuv, W, H, 3, QP, str(tmp_path / "cmp"), True, 1 | 4096, 10 ** 12, 0)
    st2, bits2, rec2 = enc_dropin.encode(yuv, W, H, 3, QP, str(tmp_path / "rep"), True, 1 | 4096, 10 ** 12, 0, env={"VTMREF_REPLACE": "1"})
    print("intra pre-selection:", st1["intra"], st2["intra"])
    for st in (st1, st2):
        assert st["rc"] == 0 and st["errors"] == 0, st
        assert st["intra"]["batches"][1] >= 2000 and st["intra"]["served"] >= 100000 and st["intra"]["mismatch"] == 0 and st["intra"]["batches"][3] == 0, st["intra"]
    assert bits1 == bits0 and rec1 == rec0 and bits2 == bits0 and rec2 == rec0
