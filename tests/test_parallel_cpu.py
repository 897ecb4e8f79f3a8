"""Data-parallel path on CPU: world_size-2 gloo processes (127.0.0.1).  The HIP model itself needs a
GPU, so the exchange is exercised with a small torch module: sharded batches + GradBucket.all_reduce
must reproduce the single-process gradient of the concatenated batch, with one flat buffer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from stove_amd.parallel import GradBucket, shard_batch


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(6, 8), nn.Tanh(), nn.Linear(8, 3)).double()


def _loss(model, x):
    # a mean over the batch, like the ELBO (stove.py:748), so rank-averaging the gradients is exact
    return (model(x) ** 2).sum(1).mean()


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = _make_model()
    bucket = GradBucket(model, world)
    torch.manual_seed(1)
    x = torch.randn(8, 6, dtype=torch.float64)
    xs = shard_batch(x, rank, world)
    assert xs.shape[0] == 4
    for _ in range(2):                      # two steps: the bucket is re-packed every step
        bucket.zero()
        _loss(model, xs).backward()
        bucket.all_reduce()
        off = 0
        for p in bucket.params:             # after the exchange every grad is a view into the one flat buffer
            assert p.grad.data_ptr() == bucket.flat[off:].data_ptr()
            off += p.numel()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)       # after the reduce, as train.py:471-472
    if rank == 0:
        torch.save(bucket.flat.clone(), out)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_allreduce_matches_single_process(tmp_path):
    out = str(tmp_path / 'flat.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    model = _make_model()
    torch.manual_seed(1)
    x = torch.randn(8, 6, dtype=torch.float64)
    _loss(model, x).backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(got, ref, rtol=1e-12, atol=1e-14)


def test_bucket_is_one_contiguous_buffer_single_rank():
    model = _make_model()
    b = GradBucket(model, 1)
    _loss(model, torch.randn(4, 6, dtype=torch.float64)).backward()
    b.all_reduce()        # no-op for a single rank
    flat = b.pack()
    assert flat.numel() == sum(p.numel() for p in model.parameters())
    off = 0
    for p in model.parameters():
        assert p.grad.data_ptr() == flat[off:].data_ptr()
        off += p.numel()
    assert float(flat.abs().sum()) > 0
    b.zero()
    assert all(p.grad is None for p in model.parameters())
