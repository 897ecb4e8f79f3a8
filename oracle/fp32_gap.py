"""The reference's own float32-vs-float64 gap on the round-5 weight-regime fixtures -> tests/golden/g16_reference_fp32_gap.json.

TEST INFRASTRUCTURE.  Run after `python oracle/make_goldens.py g16` (which writes float64 AND float32 fixtures of the 'init' and
'stress' regimes); reads both, records per quantity what a float32 run of the REFERENCE differs from its float64 run by -- the
yardstick the GPU parity bars of those regimes are set against (tests/gpu_helpers.regime_bar) -- and deletes the float32 fixtures
(only the float64 ones are committed)."""
import glob
import json
import os
import sys

import numpy as np

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
sys.path.insert(0, os.path.join(os.path.dirname(G)))
from gpu_helpers import err, err_l2, err_small  # noqa: E402


def grads(a, b, prefix='g_'):
    ks = [k for k in a.files if k.startswith(prefix) and k in b.files]
    return {'max': max(err(b[k], a[k]) for k in ks), 'l2': max(err_l2(b[k], a[k]) for k in ks), 'small': max(err_small(b[k], a[k]) for k in ks)}


def full_model_gap(a, b):
    e64, e32 = float(a['elbo']), float(b['elbo'])
    return {'elbo_f64': e64, 'elbo_rel': abs(e64 - e32) / abs(e64),
            'grad_norm_rel_max': max(abs(float(a[k]) - float(b[k])) / (float(a[k]) + 1e-9) for k in a.files if k.startswith('gn_')),
            'grad_tensor': grads(a, b),
            'prop': {k[2:]: err(b[k], a[k]) for k in a.files if k.startswith('p_') and not np.isnan(a[k]).any()},
            'rollout_z': err(b['roll_z'], a['roll_z'])}


def pair(stem):
    """(f64, f32) fixtures of one case, or None when the float32 run is not on disk (records of earlier runs are kept)."""
    if not os.path.exists(f'{G}/{stem}_f32.npz'):
        return None
    return np.load(f'{G}/{stem}_f64.npz'), np.load(f'{G}/{stem}_f32.npz')


JSON = f'{G}/g16_reference_fp32_gap.json'
out = json.load(open(JSON))['gaps'] if os.path.exists(JSON) else {}
for reg in ('init', 'stress'):
    for nm in ('n3', 'n6', 'ac3', 'grav3'):
        ab = pair(f'g7_stove_{nm}_{reg}')
        if ab:
            out[f'g7_{nm}_{reg}'] = full_model_gap(*ab)
    for nm in ('n3', 'n6'):
        ab = pair(f'g4_likelihood_{nm}_{reg}')
        if ab:
            a, b = ab
            out[f'g4_{nm}_{reg}'] = {'log_p': err(b['log_p'], a['log_p']), 'gz': err(b['gz'], a['gz']), 'grad_param': grads(a, b)}
    for nm in ('plain3', 'plain6', 'ac3', 'lim4'):
        ab = pair(f'g5_dynamics_{nm}_{reg}')
        if ab:
            a, b = ab
            out[f'g5_{nm}_{reg}'] = {'result': err(b['result'], a['result']), 'gs': err(b['gs'], a['gs']), 'grad_param': grads(a, b)}
# round 6: the full-length fixtures (B = 2, T = 100; make_goldens.py g17), all three regimes
for reg in ('analytic', 'init', 'stress'):
    for nm in ('n3', 'n6', 'ac3', 'grav3'):
        ab = pair(f'g17_stove_T100_{nm}' + ('' if reg == 'analytic' else '_' + reg))
        if ab:
            out[f'g17_{nm}_{reg}'] = full_model_gap(*ab)
json.dump({'what': "the reference's own float32-vs-float64 gap on the round-5 weight regimes and the round-6 full-length fixtures g17 (max |a-b| / max |b| unless named otherwise; l2 / small: "
                   "tests/gpu_helpers.err_l2 / err_small), from oracle/make_goldens.py g16 + oracle/fp32_gap.py; the float32 fixtures are not kept",
           'gaps': out}, open(JSON, 'w'), indent=1)
for f in sorted(set(glob.glob(f'{G}/*_init_f32.npz') + glob.glob(f'{G}/*_stress_f32.npz') + glob.glob(f'{G}/g17_*_f32.npz'))):
    os.remove(f)
print(json.dumps({k: v for k, v in out.items() if k.startswith('g17')}, indent=0)[:1500])
