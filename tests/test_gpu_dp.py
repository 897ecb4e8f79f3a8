"""Data parallelism with the real HIP model: two processes (gloo, sharing cuda:0 on the 1-GPU box) each
run forward+backward on half of the batch and all-reduce the flat gradient bucket; the result must equal
the single-process gradient of the whole batch (the ELBO is a batch mean, stove.py:748)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpu_helpers import fill_analytic
from helpers import load_golden, t_

pytestmark = pytest.mark.gpu


def _cfg():
    from stove_amd.video_prediction.config import StoveConfig
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height = 3, 32, 32
    cfg.device, cfg.dtype, cfg.random_seed = torch.device('cuda:0'), torch.float32, 42
    cfg.action_conditioned, cfg.action_space = False, None
    return cfg


def _run(model, x, gold, lo, hi):
    lat, sd = t_(gold['eps_lat'])[lo:hi, ..., 0].float(), t_(gold['eps_std'])[lo:hi, ..., 0].float()
    steps = t_(gold['eps_steps']).float().permute(1, 0, 2, 3)[lo:hi].contiguous()
    table = {'latent': lat, 'std': sd, 'steps': steps}
    model.noise_fn = lambda kind, shape: table[kind].reshape(shape)
    elbo, _, _ = model(x[lo:hi].to('cuda:0'), 1)
    (-elbo).backward()
    return float(elbo)


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.stove import Stove
    gold = load_golden('g7_stove_n3_f32')
    model = fill_analytic(Stove(_cfg())).to('cuda:0')
    bucket = ParamArena(model, world)
    x = t_(gold['x']).float()
    per = x.shape[0] // world
    elbo = _run(model, x, gold, rank * per, (rank + 1) * per)
    bucket.all_reduce()
    e = torch.tensor([elbo], dtype=torch.float64)
    dist.all_reduce(e)
    if rank == 0:
        torch.save({'grads': {n: p.grad.cpu() for n, p in model.named_parameters()}, 'elbo': float(e) / world,
                    'bucket_numel': bucket.grad.numel()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process(tmp_path):
    from stove_amd.video_prediction.stove import Stove
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'dp.pt')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    gold = load_golden('g7_stove_n3_f32')
    model = fill_analytic(Stove(_cfg())).to('cuda:0')
    x = t_(gold['x']).float()
    elbo = _run(model, x, gold, 0, x.shape[0])
    used = [n for n, p in model.named_parameters() if p.grad is not None]
    ref = torch.cat([p.grad.reshape(-1) for n, p in model.named_parameters() if p.grad is not None]).cpu()
    flat = torch.cat([got['grads'][n].reshape(-1) for n in used])
    assert abs(got['elbo'] - elbo) < 1e-5 * abs(elbo)
    rel = float((flat - ref).abs().max() / ref.abs().max())
    assert rel < 2e-4, rel
    assert flat.numel() == 1410255                 # the parameters that receive a gradient
    for n, g in got['grads'].items():              # the unused dynamics cores 1-2 stay exactly zero in the bucket
        if n not in used:
            assert float(g.abs().max()) == 0.0, n
    assert got['bucket_numel'] >= 1410255          # exchanged as ONE flat buffer (all parameters, 16-byte aligned)
