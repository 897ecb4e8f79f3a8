"""Batch data parallelism: one process per GPU, ONE RCCL all-reduce of a flat gradient bucket.

The reference has no distributed code at all (single device, model/main.py:140-141).  Sequences
are independent through the whole forward/backward, so the batch is
sharded across ranks, every rank keeps a full replica, and the only exchange per step is the
sum of the gradients: 1.45 M fp32 values = 5.8 MB, a latency-class message on xGMI.  All
parameter gradients are views into one contiguous buffer, so the exchange is a single
`all_reduce` (RCCL picks its one-shot/direct algorithm at this size) with no flatten/unflatten
copies; clipping and Adam then run on the reduced gradients, as train.py:471-473 orders them.
"""
import torch
import torch.distributed as dist


class GradBucket:
    """Flat fp32 gradient bucket of a module.

    Autograd writes `p.grad` as usual (no extra accumulate kernels).  `all_reduce()` packs the
    gradients into ONE contiguous buffer with a single `torch.cat`, all-reduces that buffer, and
    re-points every `p.grad` at its slice (views, no unflatten copies).  Parameters that never
    receive a gradient (the unused dynamics cores 1-2, reference stove.py:698-699) are left out,
    identically on every rank.  With a single rank nothing is done at all.
    """

    def __init__(self, module, world_size=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.world_size = world_size
        self.flat = None

    def rebind(self):
        """Kept for API symmetry: nothing to do, gradients are packed lazily in all_reduce()."""

    def zero(self):
        for p in self.params:
            p.grad = None

    def pack(self):
        live = [p for p in self.params if p.grad is not None]
        self.flat = torch.cat([p.grad.reshape(-1) for p in live])
        off = 0
        for p in live:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        return self.flat

    def all_reduce(self):
        """Average the gradients over the ranks (no-op for a single rank)."""
        if self.world_size <= 1:
            return
        flat = self.pack()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(self.world_size)


def shard_batch(tensor, rank, world_size):
    """Contiguous shard of the leading (sequence) dimension owned by `rank`."""
    per = tensor.shape[0] // world_size
    return tensor[rank * per:(rank + 1) * per]
