"""Deterministic analytic parameter fill shared by the golden generator and the tests.

TEST FIXTURE DATA (the weight half of every golden input; no model arithmetic in here).  Golden fixtures store
only inputs/outputs; the ~1.45 M model parameters are regenerated from this
formula on both sides:  p.flatten()[i] = offset + amp * sin(0.37 * i + phase(name)).
"""
import math
import zlib

import numpy as np
import torch


def _amp_offset(name, shape):
    fan_in = shape[-1] if len(shape) > 1 else 1
    if name.endswith('.means'):
        return 0.4, 0.5
    if name.endswith('.sigma_params'):
        return 1.5, 0.0
    if name.endswith('.params'):
        return 1.0, 0.0
    if 'encoder.rnn.weight_ih' in name:
        return 0.02, 0.0
    if 'encoder.rnn.weight_hh' in name:
        return 0.05, 0.0
    if 'encoder.fc1.weight' in name:
        return 0.1, 0.0
    if 'encoder.fc2.weight' in name:
        return 0.3, 0.0
    if name.endswith('.bias') or 'bias_' in name:
        return 0.1, 0.02
    if name.endswith('.weight'):
        return 1.5 / math.sqrt(fan_in), 0.0
    return 0.1, 0.0


def analytic_tensor(name, shape, dtype=torch.float64):
    n = int(np.prod(shape)) if len(shape) else 1
    phase = float(zlib.crc32(name.encode()) % 997)
    amp, off = _amp_offset(name, tuple(shape))
    i = np.arange(n, dtype=np.float64)
    v = off + amp * np.sin(0.37 * i + phase)
    return torch.from_numpy(v.reshape(tuple(shape))).to(dtype)


def analytic_state_dict(shapes, dtype=torch.float64):
    """shapes: {name: shape} -> {name: tensor}.  `output_vector.params` aliases are
    filled from the `vector_list` entry they alias (checkpoint row)."""
    out = {}
    for name, shape in shapes.items():
        out[name] = analytic_tensor(name, shape, dtype)
    return out
