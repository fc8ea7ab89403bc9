"""GPU end-to-end parity: the frame-level hot path (TZ -> frac -> bi-pred refinement -> residual coding) against the same
chain driven through the CPU oracle (tests/cpu_chain.py), PU by PU, on a small picture."""
import numpy as np
import pytest

import cpu_chain
import oracle_lib as ol
from vtm_amd import synth
from vtm_amd.pipeline import FrameHotPathV1 as FrameHotPath

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_ref", [False, True])
def test_frame_hot_path_matches_cpu_chain(use_ref):
    torch = pytest.importorskip("torch")
    from vtm_amd.device import Context
    if use_ref and not ol.have_ref():
        pytest.skip("oracle/_ref/libvtmref.so not present")
    W, H = 256, 128
    frames = synth.gen_frames_hard(W, H, 5)
    cur_np = np.ascontiguousarray(frames[2])
    planes, refs, acc = [], [], 0
    for t in (0, 4):
        buf, off, stride = synth.extend_plane(frames[t], margin=160)
        refs.append((acc + off, stride))
        planes.append(buf)
        acc += buf.size
    dpb_np = np.concatenate(planes)
    dev = torch.device("cuda", 0)
    cur, dpb = torch.from_numpy(cur_np).to(dev), torch.from_numpy(dpb_np).to(dev)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lam, qp = 8.0, 32
    hp = FrameHotPath(ctx, torch, dev, W, H, W, refs, [96, 96], motion_lambda=lam, qp=qp)
    hp.run(cur.data_ptr(), dpb.data_ptr())
    torch.cuda.synchronize()
    checked = 0
    R = ol.ref() if use_ref else None
    for lvl in cpu_chain.snapshot(hp):
        s, npu, jobs = lvl["size"], lvl["npu"], lvl["jobs_np"]
        for i in range(0, npu, max(1, npu // 40)):   # a sample of every level (the CPU chain is slow)
            out = cpu_chain.run_pu(cur_np, dpb_np.ctypes.data, refs[0][1], W, H, s, int(jobs["puX"][i]), int(jobs["puY"][i]),
                                   (jobs[i], jobs[npu + i]), lam, (qp + 12) // 6, (qp + 12) % 6, ref=R)
            cpu_chain.compare_with_device(lvl, i, out)
            checked += 1
    assert checked >= 100
    ctx.close()
