"""Helpers for the -m gpu parity tests: build product modules with the analytic weights."""
import numpy as np
import torch

from analytic_weights import analytic_tensor


def fill_analytic(module, prefix=''):
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(analytic_tensor(prefix + name, p.shape, torch.float64).to(p.dtype))
    return module


def err(a, b):
    """max |a-b| / max |b| with both moved to float64 numpy."""
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


_WORST = {}


def check(key, value, bar):
    """assert value < bar, and remember the worst value seen per key: tests/conftest.py writes them to
    gpurun_out/parity_errors.json at the end of the session (the tolerance bars are pinned at ~3x these)."""
    value = float(value)
    w = _WORST.get(key)
    if w is None or value > w[0]:
        _WORST[key] = (value, float(bar))
    assert value < bar, (key, value, bar)
