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
