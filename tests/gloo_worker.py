"""Worker of tests/test_host_logic.py::test_two_rank_gloo_sharding (launched by torch.distributed.run, gloo, CPU)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtm_amd import pipeline, synth   # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, H = 416, 240
    # rank 0 owns the "reconstructed" reference picture; everyone else receives it (the RCCL broadcast on GPUs)
    if rank == 0:
        buf, off, stride = synth.extend_plane(synth.gen_frames(W, H, 1)[0], margin=160)
        plane = torch.from_numpy(buf.reshape(-1).copy())
    else:
        stride = synth.padded_stride(W, 160)
        plane = torch.zeros((H + 320) * stride, dtype=torch.int16)
    dist.broadcast(plane.view(torch.uint8), src=0)   # int16 is not a collective dtype (gloo nor RCCL): move bytes
    ref = synth.extend_plane(synth.gen_frames(W, H, 1)[0], margin=160)[0].reshape(-1)
    assert np.array_equal(plane.numpy(), ref)
    # CTU-row ownership: row r belongs to rank r % world
    mine = pipeline.quadtree_levels(W, H, sizes=(64, 32), row_filter=lambda r: r % world == rank)
    counts = torch.tensor([lv[1].size for lv in mine], dtype=torch.int64)
    dist.all_reduce(counts)
    full = pipeline.quadtree_levels(W, H, sizes=(64, 32))
    assert counts.tolist() == [lv[1].size for lv in full], (counts.tolist(), [lv[1].size for lv in full])
    for (s, xs, ys, _) in mine:
        assert ((ys // 128) % world == rank).all()
    dist.barrier()
    if rank == 0:
        print("GLOO_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
