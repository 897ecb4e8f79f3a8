"""Builds a Trainer from shell arguments and/or a saved run folder (reference model/main.py:13-170).

`main(sh_args=None, restore=None, extras=None) -> Trainer`, `restore_model`, `build_config` keep
the reference's behaviour; the device is the local rank's GPU (one process per GPU) and the
process group (RCCL) is initialised when launched through torchrun.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from .utils.utils import load_args, str_to_attr
from .video_prediction.load_data import StoveDataset


def _init_distributed():
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 and not dist.is_initialized():
        local = int(os.environ.get('LOCAL_RANK', '0'))
        # nccl (= RCCL over xGMI) with one process per GPU; STOVE_DIST_BACKEND=gloo lets several ranks share one GPU
        # (the 1-GPU test box) or run without one
        backend = os.environ.get('STOVE_DIST_BACKEND', 'nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    return world


def build_config(sh_args=None, restore=None, extras=None):
    update = {}
    sh_ckpt = sh_args.get('checkpoint_path') if sh_args is not None else None
    if restore is not None or sh_ckpt is not None:
        if sh_ckpt is not None:
            restore = '/'.join(sh_ckpt.split('/')[:-1])
        import pandas
        update.update(dict(pandas.read_csv(os.path.join(restore, 'config.txt')).to_numpy()))
        update['checkpoint_path'] = os.path.join(restore, 'checkpoints', 'ckpt')
        if sh_ckpt is not None:
            update['checkpoint_path'] = sh_ckpt
    for extra in (sh_args, extras):
        if extra is not None:
            update.update(extra)
    if str_to_attr(update.get('supairvised', False)):
        raise NotImplementedError('the supervised ablation (reference model/supairvised) is out of scope')
    from .video_prediction.config import StoveConfig
    config = load_args(StoveConfig(), update)
    torch.set_num_threads(config.max_threads)
    if isinstance(config.dtype, str):
        config.dtype = eval(config.dtype)
    world = _init_distributed()
    config.world_size = world
    local = int(os.environ.get('LOCAL_RANK', '0'))
    config.device = torch.device('cuda', local) if torch.cuda.is_available() else torch.device('cpu')
    if config.dtype not in (torch.double, torch.float):
        raise ValueError
    if config.device.type == 'cuda' and config.dtype != torch.float:
        print('[stove_amd] the HIP kernels compute in float32: dtype set to torch.float32')
        config.dtype = torch.float
    torch.set_default_dtype(config.dtype)
    config.skip = 0 if config.supair_only else 2
    # [amd] replicas must agree: rank 0's seed (drawn there if None) on every rank -- the SPN region graphs are built
    # from it (supair.py:37,42); a per-process draw (reference main.py:166-168) would give every GPU another structure
    from . import parallel
    config.rank = parallel.rank()
    if config.random_seed is None:
        print('Set new random seed.')
    config.random_seed = parallel.agree_on_seed(config.random_seed)
    # shared base of the per-epoch clip permutations and of the per-rank noise streams (noise seed = base + rank)
    config.dp_seed = parallel.broadcast_int(int(np.random.randint(low=0, high=2 ** 31 - 1))) if world > 1 else 0
    return config


def restore_model(restore, extras=None, config=None, load=True):
    if config is None:
        extras = dict(extras or {})
        extras.setdefault('nolog', True)
        config = build_config(restore=restore, extras=extras)
    from .video_prediction.stove import Stove
    stove = Stove(config).to(config.device).type(config.dtype)
    if load and stove.c.checkpoint_path is not None:
        ckpt = torch.load(stove.c.checkpoint_path, map_location=stove.c.device)
        stove.load_state_dict(ckpt['model_state_dict'])
    return stove


def main(sh_args=None, restore=None, extras=None):
    config = build_config(sh_args, restore, extras)
    train_dataset = StoveDataset(config)
    test_dataset = StoveDataset(config, test=True)
    config = load_args(config, train_dataset.data_info)
    stove = restore_model(restore, extras, config, load=False)
    from .video_prediction.train import Trainer
    return Trainer(config, stove, train_dataset, test_dataset)
