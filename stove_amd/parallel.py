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
    """Makes every `p.grad` of `module` a view into one flat fp32 buffer and all-reduces that buffer."""

    def __init__(self, module, world_size=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.world_size = world_size
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def rebind(self):
        """Re-attach the views if something replaced p.grad (e.g. zero_grad(set_to_none=True))."""
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += n

    def zero(self):
        self.flat.zero_()

    def all_reduce(self):
        """Average the gradients over the ranks (no-op for a single rank)."""
        if self.world_size <= 1:
            return
        self.rebind()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(self.world_size)


def shard_batch(tensor, rank, world_size):
    """Contiguous shard of the leading (sequence) dimension owned by `rank`."""
    per = tensor.shape[0] // world_size
    return tensor[rank * per:(rank + 1) * per]
