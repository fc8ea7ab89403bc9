"""Reference-plane exchange between the GPUs of a node: rank `src` owns the reconstructed picture, every rank needs it before
its searches start.  Double-buffered asynchronous broadcast on torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs; "gloo" in the
CPU tests): while step k computes on buffer k % 2, the planes of step k + 1 travel into the other buffer.

Hazards and how they are excluded:
  * the planes of step k must have landed before step k computes      -> `Work.wait()` of that broadcast (a stream wait on GPUs)
  * buffer (k + 1) % 2 was read by step k - 1                         -> the next broadcast is issued after step k - 1's launches are queued;
                                                                         an asynchronous collective starts behind the work already on the
                                                                         caller's stream (with gloo the calls are host-ordered anyway)
"""
import torch
import torch.distributed as dist


class PlaneExchange:
    def __init__(self, buffers, src=0, produce=None, window=None):
        """buffers: two equally sized tensors (the collective moves their bytes: int16 is not a collective dtype);
        produce(buf, k): optional, called on rank `src` before the planes of step k are sent (the encoder writing its reconstruction);
        window: (first element, number of elements) of the buffers that a step sends -- the planes of the picture reconstructed last; the older
        pictures of the decoded-picture buffer arrived with earlier steps and stay resident (None: the whole buffer)."""
        assert len(buffers) == 2 and buffers[0].numel() == buffers[1].numel()
        self.bufs, self.src, self.produce = buffers, src, produce
        self.window = window
        self.n, self.pending = 0, None

    def _send(self, i, k):
        if self.produce is not None and dist.get_rank() == self.src:
            self.produce(self.bufs[i], k)
        flat = self.bufs[i].reshape(-1)
        if self.window is not None:
            flat = flat[self.window[0]:self.window[0] + self.window[1]]
        return dist.broadcast(flat.view(torch.uint8), src=self.src, async_op=True)

    def next(self):
        """Returns the buffer holding the planes of this step (ready for use on the current stream) and starts the next step's transfer."""
        i = self.n & 1
        if self.pending is None:
            self.pending = self._send(i, self.n)
        self.pending.wait()
        self.pending = self._send(1 - i, self.n + 1)
        self.n += 1
        return self.bufs[i]

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
