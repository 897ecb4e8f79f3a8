"""Small helpers on the boundary of the hot path (reference model/utils/utils.py).

`bw_transform` is on the path (stove.py:885-886); argument parsing and the CSV experiment
logger keep the reference's behaviour so run scripts and logs interchange.
"""
import ast
import itertools
import os
from datetime import datetime

import torch


def bw_transform(x):
    """(n, T, 3, w, h) one-ball-per-channel frames -> (n, T, 1, w, h): channel sum clamped to [0, 1].
    [amd] uint8 frames (the 8-bit device frame store, values round(255 v)) are scaled by 1/255 per channel on the way."""
    if x.is_cuda and x.dtype in (torch.float32, torch.uint8) and x.dim() == 5 and (x.shape[-1] * x.shape[-2]) % 4 == 0 \
            and not x.requires_grad:
        from .. import ops
        return ops.bw_transform(x)               # one fused pass on the GPU (the training input path)
    if x.dtype == torch.uint8:
        x = x.to(torch.float32) / 255.0
    return torch.clamp(x.sum(2), 0, 1).unsqueeze(2)


def str_to_float(argument):
    """'true'/'false' -> bool, numbers -> float/int, anything else unchanged."""
    if not isinstance(argument, str):
        return argument
    low = argument.lower()
    if low == 'true':
        return True
    if low == 'false':
        return False
    try:
        return float(argument) if ('.' in argument or 'e' in low) else int(argument)
    except ValueError:
        return argument


def str_to_attr(argument):
    """Python literal if the string parses as one, else the string itself."""
    try:
        return ast.literal_eval(argument)
    except Exception:
        return argument


def load_args(config, sh_args):
    """Override config attributes from a {name: value-or-string} dict; unknown names are reported and skipped."""
    for key, value in sh_args.items():
        if hasattr(config, key):
            setattr(config, key, str_to_attr(value))
        else:
            print("'{}' object has no attribute '{}'".format(type(config).__name__, key))
    return config


PERFORMANCE_COLUMNS = [
    'step', 'time', 'elbo', 'reward', 'min_ll', 'bg', 'patch', 'overlap', 'log_q', 'translik',
    'error', 'std_error', 'scale_x', 'scale_y', 'v_error', 'std_v_error',
    'z_std_0', 'z_std_1', 'z_std_2', 'z_std_3', 'z_std_4', 'z_std_5', 'swaps', 'type']


class ExperimentLogger:
    """runNNN/ directory with config.txt, performance.csv (24 fixed columns), checkpoints/, gifs/, states/."""

    def __init__(self, config, attributes=None, log_str=None):
        self.c = config
        self.attributes = list(attributes) if attributes is not None else list(PERFORMANCE_COLUMNS)
        self.log_str = log_str or ('{:d},' + (len(self.attributes) - 2) * '{:.5f},' + '{}\n')
        self.performance_str = ','.join(self.attributes) + '\n'
        if not self.c.nolog and not self.c.keep_folder:
            self.exp_dir = self.make_dir()
            self.img_dir = os.path.join(self.exp_dir, 'imgs')
            os.makedirs(self.img_dir)
            self.save_config()
            self.performance_file = os.path.join(self.exp_dir, 'performance.csv')
            with open(self.performance_file, 'w') as f:
                f.write(self.performance_str)
        elif not self.c.nolog:
            if self.c.checkpoint_path is None:
                raise ValueError('Keep folder only useful for restoring from folder!')
            self.exp_dir = '/'.join(self.c.checkpoint_path.split('/')[:-1])
            self.performance_file = os.path.join(self.exp_dir, 'performance.csv')
        else:
            self.exp_dir = os.path.join(self.c.experiment_dir, 'tmp')
        for sub in ('gifs', 'states', 'checkpoints'):
            os.makedirs(os.path.join(self.exp_dir, sub), exist_ok=True)
        self.rollout_gifs_dir = os.path.join(self.exp_dir, 'gifs')
        self.rollout_states_dir = os.path.join(self.exp_dir, 'states')
        self.checkpoint_dir = os.path.join(self.exp_dir, 'checkpoints')

    def make_dir(self):
        i = 0
        while os.path.exists(os.path.join(self.c.experiment_dir, 'run{:03d}'.format(i))):
            i += 1
        path = os.path.join(self.c.experiment_dir, 'run{:03d}'.format(i))
        os.makedirs(path)
        print('Logging to directory {}'.format(path))
        return path

    def save_config(self):
        with open(os.path.join(self.exp_dir, 'config.txt'), 'a') as f:
            f.write('setting name, setting value\n')
            for name in dir(self.c):
                if name.startswith('__'):
                    continue
                value = getattr(self.c, name)
                f.write('{},"{}"\n'.format(name, value) if isinstance(value, list) else '{},{}\n'.format(name, value))
            f.write('time,{}\n'.format(datetime.now().strftime('%Y-%m-%d %H:%M:%S')))

    def performance(self, perf_dict):
        values = [perf_dict.get(a, float('nan')) for a in self.attributes]
        values = [v.item() if torch.is_tensor(v) and v.numel() == 1 else v for v in values]
        line = self.log_str.format(*values)
        print(self.performance_str)
        print(line)
        if not self.c.nolog:
            with open(self.performance_file, 'a') as f:
                f.write(line)


def match_states(predicted, true, match_idxs=(0, 1), time_frame=5):
    """One global object permutation per sequence minimising the mean position error over the
    first `time_frame` steps.  numpy (n, T, o, d) in, permuted `predicted` out."""
    predicted, true = torch.from_numpy(predicted), torch.from_numpy(true)
    idx = list(match_idxs)
    perms = list(itertools.permutations(range(true.shape[2])))
    errs = []
    for perm in perms:
        d = ((predicted[:, :time_frame][:, :, list(perm)][..., idx] - true[:, :time_frame][..., idx]) ** 2).sum(-1)
        errs.append(torch.sqrt(d).mean((1, 2)))
    best = torch.stack(errs, 1).argmin(1)
    out = torch.stack([predicted[i][:, list(perms[j])] for i, j in enumerate(best.tolist())], 0)
    return out.numpy()


def settle_host_gc():
    """[amd] Keep the interpreter's full (generation-2) collections out of the step loop: a step is ~3 ms of device time that the
    host feeds ~1 ms ahead, and one full collection walks every object torch has created (80-120 ms measured on the MI355X box's host),
    which drains the device queue and shows up as one 80-120 ms step every few dozen.  Called once the loop is warm: collect what is
    garbage now, then move the survivors (modules, parameters, the arena, the loaded library) to the permanent generation so later
    collections only walk what the steps themselves allocate.  The collector stays enabled."""
    import gc
    gc.collect()
    gc.freeze()
