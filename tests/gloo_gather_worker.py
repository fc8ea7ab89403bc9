"""Worker of tests/test_host_logic.py::test_two_rank_ctu_shards_and_result_gather (torch.distributed.run, gloo, CPU): the N > 1 path of bench.py --
raster-scan CTU bands per rank (vtm_amd.pipeline.ctu_bands / band_filter) and the double-buffered result gather to rank 0 (vtm_amd.exchange.ResultGather).
Every rank 'computes' a record per PU of its band (a hash of the PU position and the step); rank 0 must end up with every PU of the picture exactly once."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtm_amd import pipeline   # noqa: E402
from vtm_amd.exchange import ResultGather   # noqa: E402


def records(levels, step):
    out = []
    for (s, xs, ys, _) in levels:
        out.append(np.stack([np.full(xs.size, s), xs, ys, (xs * 31 + ys * 17 + s + step) % 9973], 1).astype(np.int32))
    return np.concatenate(out) if out else np.zeros((0, 4), np.int32)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, H = 1920, 1080          # 15 x 9 CTUs (the last row is partial): 135 CTUs over 2 ranks = 67 / 68, the cut falls inside a row
    for unit in ("ctu", "row"):
        bands = pipeline.ctu_bands(W, H, world, unit=unit)
        assert bands[0][0] == 0 and bands[-1][1] == 15 * 9 and all(a[1] == b[0] for a, b in zip(bands[:-1], bands[1:]))
        if unit == "ctu":
            assert max(b[1] - b[0] for b in bands) - min(b[1] - b[0] for b in bands) <= 1
        mine = pipeline.quadtree_levels(W, H, sizes=(128, 64, 32), ctu_filter=pipeline.band_filter(W, bands[rank]))
        full = pipeline.quadtree_levels(W, H, sizes=(128, 64, 32))
        g = ResultGather(records(mine, 0).nbytes, torch.device("cpu"), dst=0)
        for step in range(5):
            rec = torch.from_numpy(records(mine, step).view(np.uint8).reshape(-1).copy())
            g.submit([rec[:rec.numel() // 2], rec[rec.numel() // 2:]])      # several tensors per rank, packed in order
        g.drain()
        if rank == 0:
            got = np.concatenate([t.numpy().view(np.int32).reshape(-1, 4) for t in g.last()])
            exp = records(full, 4)
            key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]   # noqa: E731
            assert got.shape == exp.shape and np.array_equal(key(got), key(exp)), (unit, got.shape, exp.shape)
    dist.barrier()
    if rank == 0:
        print("GATHER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
