"""Worker of tests/test_host_logic.py::test_two_rank_plane_exchange (torch.distributed.run, gloo, CPU): the double-buffered reference-plane
broadcast of bench.py --gpus N.  Rank 0 'reconstructs' a new picture before every transfer; every rank must see picture k in step k."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtm_amd.exchange import PlaneExchange   # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    bufs = [torch.full((70001,), -1, dtype=torch.int16), torch.full((70001,), -1, dtype=torch.int16)]
    sent = []
    x = PlaneExchange(bufs, src=0, produce=lambda b, k: (b.fill_(100 + k), sent.append(k)))
    seen = []
    for k in range(7):
        planes = x.next()
        assert planes.data_ptr() == bufs[k & 1].data_ptr()
        assert bool((planes == 100 + k).all()), (rank, k, int(planes[0]))   # picture k, complete, in step k
        seen.append(int(planes[12345]))
    x.drain()
    assert seen == [100 + k for k in range(7)]
    if rank == 0:
        assert sent == list(range(8))   # one transfer ahead
    # window: only the newest picture's planes travel; the rest of the buffer (older pictures, here: a marker each rank wrote itself) stays untouched
    bufs = [torch.full((5000,), 7 + rank, dtype=torch.int16), torch.full((5000,), 7 + rank, dtype=torch.int16)]
    x = PlaneExchange(bufs, src=0, produce=lambda b, k: b[1000:3000].fill_(200 + k), window=(1000, 2000))
    for k in range(4):
        planes = x.next()
        assert bool((planes[1000:3000] == 200 + k).all()) and bool((planes[:1000] == 7 + rank).all()) and bool((planes[3000:] == 7 + rank).all()), (rank, k)
    x.drain()
    dist.barrier()
    if rank == 0:
        print("EXCHANGE_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
