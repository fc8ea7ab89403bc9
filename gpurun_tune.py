import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from vtm_amd import synth
from vtm_amd.device import Context
from vtm_amd.pipeline import FrameME
W,H=3840,2160
frames=synth.gen_frames(W,H,5); cur_np=np.ascontiguousarray(frames[2])
planes,refs,acc=[],[],0
for t in (0,4):
    buf,off,stride=synth.extend_plane(frames[t],160); refs.append((acc+off,stride)); planes.append(buf); acc+=buf.size
dev=torch.device('cuda',0)
cur=torch.from_numpy(cur_np).to(dev); dpb=torch.from_numpy(np.concatenate(planes)).to(dev)
ctx=Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
from vtm_amd.lib import PicParams
fme=FrameME(ctx,torch,dev,W,H,W,refs,[96,96])
fme.run(cur.data_ptr(),dpb.data_ptr()); torch.cuda.synchronize()
for lvl in fme.levels:
    for wpj in (1,2,4,8,16):
        pic=PicParams(W,H,128,10,wpj)
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        ctx.tz_search_batch(pic,cur.data_ptr(),dpb.data_ptr(),lvl['jobs'].data_ptr(),lvl['n'],lvl['res'].data_ptr())
        e0.record()
        for _ in range(3): ctx.tz_search_batch(pic,cur.data_ptr(),dpb.data_ptr(),lvl['jobs'].data_ptr(),lvl['n'],lvl['res'].data_ptr())
        e1.record(); torch.cuda.synchronize()
        print(lvl['size'], wpj, '%.3f ms'%(e0.elapsed_time(e1)/3), flush=True)
