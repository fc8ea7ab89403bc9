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
    def __init__(self, buffers, src=0, produce=None):
        """buffers: two equally sized tensors (the collective moves their bytes: int16 is not a collective dtype);
        produce(buf, k): optional, called on rank `src` before the planes of step k are sent (the encoder writing its reconstruction)."""
        assert len(buffers) == 2 and buffers[0].numel() == buffers[1].numel()
        self.bufs, self.src, self.produce = buffers, src, produce
        self.n, self.pending = 0, None

    def _send(self, i, k):
        if self.produce is not None and dist.get_rank() == self.src:
            self.produce(self.bufs[i], k)
        return dist.broadcast(self.bufs[i].reshape(-1).view(torch.uint8), src=self.src, async_op=True)

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
