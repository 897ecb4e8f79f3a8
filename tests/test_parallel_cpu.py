"""Data-parallel path on CPU: world_size-2 gloo processes (127.0.0.1).  The HIP kernels need a GPU, so what runs here is
everything AROUND them, with the real model's containers: replicas of `Stove` that start from different weights agree
after `ParamArena.sync`, the seed of the SPN structures is rank 0's, the flat gradient buffer is averaged by one
all-reduce, the clip shards of the ranks are disjoint and equally long.  (tests/test_gpu_dp.py runs the same path with
the kernels, through model.main.main and Trainer.train_step.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from stove_amd import parallel
from stove_amd.arena import ParamArena
from stove_amd.parallel import shard_batch, shard_order


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _cfg(seed):
    from stove_amd.video_prediction.config import StoveConfig
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height = 3, 32, 32
    cfg.device, cfg.dtype, cfg.random_seed = torch.device('cpu'), torch.float32, seed
    cfg.action_conditioned, cfg.action_space = False, None
    return cfg


def _fake_grad(numel, rank):
    return torch.sin(torch.arange(numel, dtype=torch.float32) * 0.01 + rank)


def _arena_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.load_data import DeviceClipLoader, StoveDataset
    from stove_amd.video_prediction.stove import Stove
    # rank 0 owns the seed; the other rank proposes a different one (or none) and must end up with rank 0's
    seed = parallel.agree_on_seed(7 if rank == 0 else None)
    drawn = parallel.agree_on_seed(None)
    torch.manual_seed(100 + rank)                       # every process initialises its own weights ...
    model = Stove(_cfg(seed))
    arena = ParamArena(model, world)
    own = arena.data.clone()
    arena.sync(0)                                       # ... and starts from rank 0's
    # gradients: what autograd would leave in the p.grad views
    for p in arena.params:
        o = arena.offset[id(p)]
        p.grad.copy_(_fake_grad(arena.numel, rank)[o:o + p.numel()].view(p.shape))
    arena.all_reduce()
    # clip shards of one epoch
    c = _cfg(seed)
    c.num_episodes, c.num_visible, c.num_rollout, c.frame_step = 4, 3, 2, 1
    data = {'X': np.zeros((4, 12, 3, 4, 4)), 'y': np.zeros((4, 12, 3, 4)), 'coord_lim': 10}
    loader = DeviceClipLoader(StoveDataset(c, data=data), 3, torch.device('cpu'), torch.float32, rank=rank, world=world, seed=5)
    epochs = []
    for _ in range(2):
        ids = []
        for batch in loader:
            assert batch['present_images'].shape == (3, 3, 3, 4, 4)
            ids.append(loader.last_clip_ids.clone())
        epochs.append(torch.stack(ids))
    torch.save({'seed': seed, 'drawn': drawn, 'own': own, 'data': arena.data.clone(), 'grad': arena.grad.clone(),
                'scope': model.sup.obj_spn._plan_cpu['leaf_order'].clone(), 'epochs': epochs, 'len': len(loader)},
               out + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_arena_replicas_agree(tmp_path):
    out = str(tmp_path / 'r')
    mp.spawn(_arena_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + '0'), torch.load(out + '1')
    assert r0['seed'] == r1['seed'] == 7 and r0['drawn'] == r1['drawn'] and 0 <= r0['drawn'] < 1000
    assert torch.equal(r0['scope'], r1['scope'])                           # same SPN region graph on both ranks
    assert not torch.equal(r0['own'], r1['own'])                           # the replicas did start apart
    assert torch.equal(r0['data'], r1['data']) and torch.equal(r0['data'], r0['own'])
    n = r0['grad'].numel()
    mean = (_fake_grad(n, 0) + _fake_grad(n, 1)) / 2
    # pads between tensors are not gradients: compare on parameter elements only
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.stove import Stove
    arena = ParamArena(Stove(_cfg(7)), 1)
    mask = torch.zeros(n, dtype=torch.bool)
    for p in arena.params:
        o = arena.offset[id(p)]
        mask[o:o + p.numel()] = True
    assert torch.equal(r0['grad'], r1['grad'])
    assert torch.allclose(r0['grad'][mask], mean[mask], rtol=0, atol=1e-7)
    assert float(r0['grad'][~mask].abs().sum()) == 0.0
    # shards: same length on both ranks, disjoint within an epoch, reshuffled between epochs
    assert r0['len'] == r1['len'] == (4 * 8 // 2) // 3
    for e in range(2):
        a, b = set(r0['epochs'][e].flatten().tolist()), set(r1['epochs'][e].flatten().tolist())
        assert len(a) == len(b) == r0['len'] * 3 and not (a & b)
    assert not torch.equal(r0['epochs'][0], r0['epochs'][1])


def test_eight_rank_replicas_agree(tmp_path):
    """The same contract at the world size the driver scales to (8 ranks, gloo on this host): one seed, one SPN structure, one set of
    parameters, the mean of eight gradient buffers on every rank, eight disjoint clip shards of equal length."""
    world = 8
    out = str(tmp_path / 'r')
    mp.spawn(_arena_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    rs = [torch.load(out + str(r)) for r in range(world)]
    assert all(r['seed'] == 7 for r in rs) and len({r['drawn'] for r in rs}) == 1
    n = rs[0]['grad'].numel()
    mean = sum(_fake_grad(n, r) for r in range(world)) / world
    for r in rs:
        assert torch.equal(r['scope'], rs[0]['scope']) and torch.equal(r['data'], rs[0]['own'])
        assert torch.equal(r['grad'], rs[0]['grad'])
    nz = rs[0]['grad'] != 0
    assert torch.allclose(rs[0]['grad'][nz], mean[nz], rtol=0, atol=2e-7)
    assert all(r['len'] == (4 * 8 // world) // 3 for r in rs)
    for e in range(2):
        sets = [set(r['epochs'][e].flatten().tolist()) for r in rs]
        assert all(len(s_) == rs[0]['len'] * 3 for s_ in sets)
        assert len(set().union(*sets)) == sum(len(s_) for s_ in sets)          # pairwise disjoint


def test_shard_order_is_a_partition():
    order = torch.randperm(103)
    parts = [shard_order(order, r, 4, 5) for r in range(4)]
    assert all(len(p) == 25 for p in parts)
    flat = torch.cat(parts).tolist()
    assert len(set(flat)) == 100 and set(flat) <= set(order.tolist())
    assert shard_batch(torch.arange(8), 1, 2).tolist() == [4, 5, 6, 7]


def test_single_process_helpers_are_noops():
    assert parallel.world() == 1 and parallel.rank() == 0
    assert parallel.agree_on_seed(13) == 13 and 0 <= parallel.agree_on_seed(None) < 1000
    assert parallel.broadcast_int(5) == 5
    t = torch.ones(3)
    parallel.broadcast_tensors([t])
    assert t.tolist() == [1, 1, 1]


# ---- the flat arena on a plain CPU fp64 module (what a Trainer on a non-GPU / non-fp32 model exchanges through)
def _make_model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(6, 8), nn.Tanh(), nn.Linear(8, 3)).double()


def _loss(model, x):
    # a mean over the batch, like the ELBO (stove.py:748), so rank-averaging the gradients is exact
    return (model(x) ** 2).sum(1).mean()


def _bucket_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = _make_model()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    bucket = ParamArena(model, world)
    bucket.sync(0)
    torch.manual_seed(1)
    x = torch.randn(8, 6, dtype=torch.float64)
    xs = shard_batch(x, rank, world)
    for _ in range(2):
        bucket.zero()
        _loss(model, xs).backward()
        bucket.all_reduce()
        for p in bucket.params:             # every grad is (and stays) a view into the one flat buffer
            assert p.grad.data_ptr() == bucket.view_of(p, bucket.grad).data_ptr()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)       # after the reduce, as train.py:471-472
    if rank == 0:
        torch.save(torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone(), out)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_bucket_matches_single_process(tmp_path):
    out = str(tmp_path / 'flat.pt')
    mp.spawn(_bucket_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    model = _make_model()
    torch.manual_seed(1)
    x = torch.randn(8, 6, dtype=torch.float64)
    _loss(model, x).backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(got, ref, rtol=1e-12, atol=1e-14)
