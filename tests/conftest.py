import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """Achieved parity errors of the -m gpu run (tests/gpu_helpers.check) -> gpurun_out/parity_errors.json."""
    try:
        import json
        import gpu_helpers
    except Exception:
        return
    if not gpu_helpers._WORST:
        return
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_errors.json'), 'w') as f:
            json.dump({k: {'worst': v[0], 'bar': v[1]} for k, v in sorted(gpu_helpers._WORST.items())}, f, indent=1)
    except OSError:
        pass
