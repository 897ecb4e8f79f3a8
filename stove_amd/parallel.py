"""Batch data parallelism: one process per GPU, ONE RCCL all-reduce of a flat gradient bucket.

The reference has no distributed code at all (single device, model/main.py:140-141).  Sequences
are independent through the whole forward/backward, so the batch is
sharded across ranks, every rank keeps a full replica, and the only exchange per step is the
sum of the gradients: 1.45 M fp32 values = 5.8 MB, a latency-class message on xGMI.  All
parameter gradients are views into one contiguous buffer (stove_amd/arena.py: ParamArena), so the exchange is a single
`all_reduce` (RCCL picks its one-shot/direct algorithm at this size) with no flatten/unflatten
copies; clipping and Adam then run on the reduced gradients, as train.py:471-473 orders them.
"""
import torch
import torch.distributed as dist


def shard_batch(tensor, rank, world_size):
    """Contiguous shard of the leading (sequence) dimension owned by `rank`."""
    per = tensor.shape[0] // world_size
    return tensor[rank * per:(rank + 1) * per]


# ---------------------------------------------------------------------------------------------
# Replica agreement.  The reference is single-process (model/main.py:140-141), so none of this
# exists there; SURVEY.md section 8(e) fixes the contract: identical `random_seed` (the SPN region
# graphs are drawn from it, supair.py:37,42) and identical parameters on every rank, disjoint clip
# shards, per-rank reparameterisation noise.
# ---------------------------------------------------------------------------------------------
def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def _comm_device():
    """Tensors handed to collectives must live where the backend works: the GPU for nccl (= RCCL), anywhere for gloo."""
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def broadcast_int(value, src=0):
    """Rank `src`'s integer on every rank (one 8-byte broadcast; a no-op without a process group)."""
    if world() <= 1:
        return int(value)
    t = torch.tensor([int(value) if rank() == src else 0], dtype=torch.int64, device=_comm_device())
    dist.broadcast(t, src)
    return int(t.item())


def agree_on_seed(seed, high=1000, src=0):
    """`config.random_seed` as every rank must see it: rank 0's value, drawn there if it is None
    (reference main.py:166-168 draws `np.random.randint(0, 1000)` per process -- per rank that would build a
    different SPN structure on every GPU)."""
    import numpy as np
    if rank() == src and seed is None:
        seed = int(np.random.randint(low=0, high=high))
    return broadcast_int(-1 if seed is None else seed, src)


def broadcast_tensors(tensors, src=0):
    """In-place broadcast of rank `src`'s values (parameters, optimiser moments)."""
    if world() <= 1:
        return
    for t in tensors:
        dist.broadcast(t, src)


def shard_order(order, rank_, world_, batch_size):
    """Clips of one epoch owned by `rank_`: the SHARED shuffled order dealt round-robin, cut to a whole number of
    per-rank batches that is the same on every rank (so every rank enters the same number of all-reduces)."""
    mine = order[rank_::world_]
    n_batches = (len(order) // world_) // batch_size
    return mine[:n_batches * batch_size]
