"""Worker of tests/test_host_logic.py::test_two_rank_plane_exchange (torch.distributed.run, gloo, CPU): the reference-plane broadcast of
bench.py --gpus N into a ring of slots of ONE resident decoded-picture buffer.  Rank 0 'reconstructs' a new picture into a different slot before every
transfer; in step k every rank must see picture k in its slot, the pictures of the earlier steps still in theirs, and the rest of the buffer as rank 0 had
it at start-up."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtm_amd.exchange import PlaneExchange   # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    # a DPB of 9001 samples: static pictures in [0, 3000), a ring of three 2000-sample slots behind them; every rank starts with its own garbage
    dpb = torch.full((9001,), -1 - rank, dtype=torch.int16)
    if rank == 0:
        dpb[:3000] = torch.arange(3000, dtype=torch.int16)
        dpb[3000:] = 55
    slots = [(3000, 2000), (5000, 2000), (7000, 2000)]
    sent = []

    def produce(buf, slot, k):
        a, n = slots[slot]
        buf[a:a + n].fill_(100 + k)
        sent.append((slot, k))

    x = PlaneExchange(dpb, slots, src=0, produce=produce)
    x.sync_all()
    assert bool((dpb[:3000] == torch.arange(3000, dtype=torch.int16)).all()) and int(dpb[9000]) == 55
    for k in range(8):
        i = x.next()
        assert i == k % 3
        a, n = slots[i]
        assert bool((dpb[a:a + n] == 100 + k).all()), (rank, k, int(dpb[a]))          # picture k, complete, in step k
        if k >= 1:                                                                       # the previous picture is still resident on EVERY rank (ring of 3: one
            a1, n1 = slots[(k - 1) % 3]                                                  # slot is being written for step k + 1, two hold pictures k and k - 1)
            assert bool((dpb[a1:a1 + n1] == 100 + k - 1).all()), (rank, k, int(dpb[a1]))
        assert bool((dpb[:3000] == torch.arange(3000, dtype=torch.int16)).all()) and int(dpb[9000]) == 55   # nothing outside the ring moved
    x.drain()
    if rank == 0:
        assert sent == [(k % 3, k) for k in range(9)]   # one transfer ahead
    # a ring of two: the slot of step k + 1 is the one step k - 1 used; step k's own slot is never touched while it runs
    dpb2 = torch.full((5000,), 7 + rank, dtype=torch.int16)
    ring = [(1000, 1000), (3000, 1000)]
    x = PlaneExchange(dpb2, ring, src=0, produce=lambda b, s, k: b[ring[s][0]:ring[s][0] + 1000].fill_(200 + k))
    for k in range(5):
        i = x.next()
        a = ring[i][0]
        assert bool((dpb2[a:a + 1000] == 200 + k).all()), (rank, k)
        assert bool((dpb2[:1000] == 7 + rank).all()) and bool((dpb2[2000:3000] == 7 + rank).all()) and bool((dpb2[4000:] == 7 + rank).all()), (rank, k)
    x.drain()
    dist.barrier()
    if rank == 0:
        print("EXCHANGE_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
