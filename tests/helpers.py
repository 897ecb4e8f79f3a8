"""Shared helpers for the parity tests (oracle side)."""
import os

import numpy as np
import torch

import stove_oracle as O
from analytic_weights import analytic_tensor, fan_ins

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


def oracle_setup(dtype=torch.float64, requires_grad=True, regime='analytic', **cfg):
    """config + SPN structures + analytic parameters (weight regime `regime`) keyed by reference state-dict names."""
    c = O.default_config(**cfg)
    structs = O.build_structs(c)
    shapes = O.param_shapes(c, structs)
    fi = fan_ins({k: tuple(v) for k, v in shapes.items()})
    params = {}
    for k, shp in shapes.items():
        t = analytic_tensor(k, shp, dtype, regime, fi.get(k))
        params[k] = t.requires_grad_() if requires_grad else t
    return c, structs, params


def t_(a, dtype=torch.float64):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
