"""The analytic parameter fill of the golden fixtures lives with the fixtures (tests/golden/analytic_weights.py: it is the
weight half of every golden input); this name is kept for the oracle-side scripts (oracle/make_goldens.py)."""
import os
import sys

_G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
if _G not in sys.path:
    sys.path.insert(0, _G)
import importlib.util as _u

_spec = _u.spec_from_file_location('_golden_analytic_weights', os.path.join(_G, 'analytic_weights.py'))
_m = _u.module_from_spec(_spec)
_spec.loader.exec_module(_m)
analytic_tensor, analytic_state_dict, fan_ins, REGIMES = _m.analytic_tensor, _m.analytic_state_dict, _m.fan_ins, _m.REGIMES
