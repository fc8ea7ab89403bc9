"""Reference-plane exchange between the GPUs of a node: rank `src` owns the reconstructed picture, every rank needs it before
its searches start.  Every rank keeps ONE resident decoded-picture buffer; a newly reconstructed picture travels (asynchronous broadcast on
torch.distributed: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests) into a SLOT of that buffer which the running step does
not reference, exactly as an encoder writes a reconstruction into a free DPB slot.  All other slots stay where they are on every rank, so a step
always sees the same set of pictures on every rank -- also when the DPB really changes from step to step.

Hazards and how they are excluded:
  * the picture of step k must have landed before step k computes    -> `Work.wait()` of that broadcast (a stream wait on GPUs)
  * the slot receiving the picture of step k + 1 must not be read by step k -> the caller passes a ring of >= 2 slots and step k's tables address
                                                                         slot k % len(slots) as the newest picture (bench.py keeps one table set per slot)
  * slot (k + 1) % len(slots) was read by step k + 1 - len(slots)     -> the broadcast is issued after that step's launches are queued; an asynchronous
                                                                         collective starts behind the work already on the caller's stream (gloo: host order)
A step may therefore reference the pictures of the last len(slots) - 1 steps out of the ring (plus anything outside the ring, which never changes after
sync_all()); a longer reference history needs a longer ring.
"""
import torch
import torch.distributed as dist


class PlaneExchange:
    def __init__(self, dpb, slots, src=0, produce=None):
        """dpb: the rank's resident decoded-picture buffer (one tensor; the collective moves bytes: int16 is not a collective dtype);
        slots: ring of (first element, number of elements) windows of `dpb`, all of one size -- the picture of step k lands in slots[k % len(slots)];
        produce(dpb, slot, k): optional, called on rank `src` before the picture of step k is sent (the encoder writing its reconstruction there)."""
        assert len(slots) >= 2 and len({int(s[1]) for s in slots}) == 1, "a ring of at least two equally sized slots"
        flat = dpb.reshape(-1)
        for a, n in slots:
            assert 0 <= a and a + n <= flat.numel()
        for i, (a, n) in enumerate(slots):
            for b, m in slots[i + 1:]:
                assert a + n <= b or b + m <= a, "slots overlap"
        self.dpb, self.slots, self.src, self.produce = dpb, [(int(a), int(n)) for a, n in slots], src, produce
        self.n, self.pending = 0, None

    def sync_all(self):
        """One blocking broadcast of the WHOLE buffer (setup, outside any timed region): every rank starts from rank `src`'s pictures."""
        dist.broadcast(self.dpb.reshape(-1).view(torch.uint8), src=self.src)

    def _send(self, k):
        i = k % len(self.slots)
        if self.produce is not None and dist.get_rank() == self.src:
            self.produce(self.dpb, i, k)
        a, n = self.slots[i]
        return dist.broadcast(self.dpb.reshape(-1)[a:a + n].view(torch.uint8), src=self.src, async_op=True)

    def next(self):
        """Returns the index of the slot that holds the picture of this step (landed; usable on the current stream) and starts the next step's transfer
        into the following slot of the ring."""
        if self.pending is None:
            self.pending = self._send(self.n)
        self.pending.wait()
        i = self.n % len(self.slots)
        self.pending = self._send(self.n + 1)
        self.n += 1
        return i

    def drain(self):
        """Waits for the transfer that is still in flight (call before stopping a clock or destroying the process group)."""
        if self.pending is not None:
            self.pending.wait()
            self.pending = None


class ResultGather:
    """The way back: every rank's per-PU / per-TU result records of a step travel to rank `dst` (where the host encoder would continue with the
    mode decision and the entropy coder).  The ranks' shares differ by a few CTUs, so each sends a buffer padded to the largest share; double-
    buffered and asynchronous like PlaneExchange: the gather of step k runs while step k + 1 computes, a buffer is reused only after its gather
    has completed."""

    def __init__(self, nbytes_local, device, dst=0):
        self.rank, self.world, self.dst = dist.get_rank(), dist.get_world_size(), dst
        sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(self.world)]
        dist.all_gather(sizes, torch.tensor([nbytes_local], dtype=torch.int64, device=device))
        self.sizes = [int(s.item()) for s in sizes]
        self.max = max(self.sizes)
        self.send = [torch.zeros(self.max, dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [[torch.zeros(self.max, dtype=torch.uint8, device=device) for _ in range(self.world)] for _ in range(2)] if self.rank == dst else None
        self.pending, self.n = [None, None], 0

    def submit(self, tensors):
        """tensors: this rank's result tensors (uint8 views) of the step just queued on the current stream"""
        i = self.n & 1
        if self.pending[i] is not None:
            self.pending[i].wait()
        total = sum(t.numel() for t in tensors)
        assert total == self.sizes[self.rank]
        torch.cat(tensors, out=self.send[i][:total])      # one copy kernel for all result tables of the step
        self.pending[i] = dist.gather(self.send[i], self.recv[i] if self.rank == self.dst else None, dst=self.dst, async_op=True)
        self.n += 1

    def drain(self):
        for i in (0, 1):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None

    def last(self):
        """rank dst, after drain(): the byte tensors of the last submitted step, one per rank, trimmed to that rank's share"""
        i = (self.n - 1) & 1
        return [self.recv[i][r][:self.sizes[r]] for r in range(self.world)]
