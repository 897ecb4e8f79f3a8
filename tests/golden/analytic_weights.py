"""Deterministic parameter fills shared by the golden generator and the tests.

TEST FIXTURE DATA (the weight half of every golden input; no model arithmetic in here).  Golden fixtures store
only inputs/outputs; the ~1.45 M model parameters are regenerated from a formula on both sides.  Three REGIMES
(round 5: parity in one weight regime certifies that regime only):

* ``'analytic'``  p.flatten()[i] = offset + amp * sin(0.37 * i + phase(name))  -- smooth, mid-range everywhere.
* ``'init'``      the statistics the reference starts training from (what `bench.py` times): SPN leaves and sum
                  weights ~ truncated normal(0, 0.1) cut at two sigma (reference `model/spn/rat_torch.py:11-18`,
                  `:111-118`, `:250-253` with `init_fn = truncated_normal_` of `:32`), `nn.Linear` weights and biases
                  ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)), `nn.LSTM` ~ U(-1/sqrt(hidden), 1/sqrt(hidden)), the recognition
                  head xavier-uniform with biases 0.1 (`model/video_prediction/encoder.py:23-26`) -- drawn from
                  numpy's frozen legacy `RandomState(crc32(name))` stream, so only the rule is committed.
* ``'stress'``    a trained / saturated model: leaf variance parameters rho = +-8 (variances AT `obj/bg_min/max_var`,
                  reference `config.py:102-106`), sum parameters x 20 (near one-hot mixtures), leaf means over the whole
                  pixel range, an encoder head that drives `constrain_zp` (reference `supair.py:125-147`) into both
                  ends of every sigmoid (`sx` at 0.1 and at `max_obj_scale`, positions at +-0.9: glimpses leave the
                  frame, stds near 0 and near 0.3), and a dynamics output layer x 8 so that `constrain_z_dyn`
                  (`dynamics.py:147-179`) saturates as well.
"""
import math
import zlib

import numpy as np
import torch

REGIMES = ('analytic', 'init', 'stress')


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


def _sine(name, shape, amp, off):
    n = int(np.prod(shape)) if len(shape) else 1
    phase = float(zlib.crc32(name.encode()) % 997)
    i = np.arange(n, dtype=np.float64)
    return off + amp * np.sin(0.37 * i + phase)


def _init_values(name, shape, fan_in):
    n = int(np.prod(shape)) if len(shape) else 1
    rs = np.random.RandomState(zlib.crc32((name + '|init').encode()) & 0x7fffffff)
    if name.endswith(('.means', '.sigma_params', '.params')):
        # truncated_normal_: of four standard normal candidates per entry the first inside (-2, 2); times 0.1
        cand = rs.standard_normal((n, 4))
        ok = np.abs(cand) < 2
        first = np.argmax(ok, axis=1)
        return 0.1 * cand[np.arange(n), first]
    if '.rnn.' in name:
        bound = 1.0 / math.sqrt(shape[0] // 4)
    elif 'encoder.fc' in name:
        # the recognition network's head (reference encoder.py:23-26): xavier_uniform_ weights, biases = 0.1
        if name.endswith('.bias'):
            return np.full(n, 0.1)
        bound = math.sqrt(6.0 / (shape[0] + shape[1]))
    else:
        if fan_in is None:
            fan_in = shape[-1] if len(shape) > 1 else 1
        bound = 1.0 / math.sqrt(fan_in)
    return rs.uniform(-bound, bound, size=n)


def _stress_values(name, shape):
    amp, off = _amp_offset(name, tuple(shape))
    if name.endswith('.sigma_params'):
        return 8.0 * np.sign(_sine(name, shape, 1.0, 0.0) + 1e-12)
    if name.endswith('.means'):
        return _sine(name, shape, 0.5, 0.5)
    if name.endswith('.params'):
        return _sine(name, shape, 20.0, 0.0)
    # (recognition network: amplitudes chosen so that the objects' codes stay DISTINCT -- with a stronger recurrence the LSTM runs into
    # a fixed point after three steps, objects 4-6 of the six-object cases get the same code to 1e-7, and the matcher's argmin
    # between them is decided below float32 resolution: a fixture no float32 implementation can be held to.  tools/fixture_robustness.py
    # checks that 3e-5 of noise on the codes leaves every matching decision of the stress fixtures unchanged.)
    if 'encoder.rnn.weight_ih' in name:
        return _sine(name, shape, 0.2, 0.0)
    if 'encoder.rnn.weight_hh' in name:
        return _sine(name, shape, 0.2, 0.0)
    if 'encoder.fc1.weight' in name:
        return _sine(name, shape, 0.4, 0.0)
    if 'encoder.fc2.weight' in name:
        return _sine(name, shape, 8.0, 0.0)
    if 'encoder.fc2.bias' in name:
        return _sine(name, shape, 2.0, 0.0)
    if name.startswith('dyn.out.') and name.endswith('.1.weight'):
        return _sine(name, shape, 8.0 * amp, 0.0)
    if name.startswith('dyn.att_net.') and name.endswith('.2.weight'):
        return _sine(name, shape, 2.0 * amp, 0.0)
    return _sine(name, shape, amp, off)


def analytic_tensor(name, shape, dtype=torch.float64, regime='analytic', fan_in=None):
    """`fan_in`: of the Linear layer a bias belongs to (the 'init' regime only; `fan_ins` collects them)."""
    shape = tuple(shape)
    if regime == 'analytic':
        amp, off = _amp_offset(name, shape)
        v = _sine(name, shape, amp, off)
    elif regime == 'init':
        v = _init_values(name, shape, fan_in)
    elif regime == 'stress':
        v = _stress_values(name, shape)
    else:
        raise ValueError(f'unknown weight regime {regime!r}')
    return torch.from_numpy(np.asarray(v, dtype=np.float64).reshape(shape)).to(dtype)


def fan_ins(shapes):
    """{bias name: fan_in of its layer} from a {name: shape} map (Linear: the weight's last dimension)."""
    out = {}
    for name in shapes:
        if name.endswith('.bias'):
            w = shapes.get(name[:-4] + 'weight')
            if w is not None and len(w) > 1:
                out[name] = int(w[-1])
    return out


def analytic_state_dict(shapes, dtype=torch.float64, regime='analytic'):
    """shapes: {name: shape} -> {name: tensor}.  `output_vector.params` aliases are
    filled from the `vector_list` entry they alias (checkpoint row)."""
    fi = fan_ins(shapes)
    return {name: analytic_tensor(name, shape, dtype, regime, fi.get(name)) for name, shape in shapes.items()}
