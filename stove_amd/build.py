"""Compile the HIP sources into stove_amd/libstove_hip.so (in-tree, gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libstove_hip.so')
SOURCES = ['capi.hip']
DEPS = ['capi.hip', 'common.h', 'spn_obj.hip', 'spn_bg.hip', 'spn_bg_mfma.hip', 'scene.hip', 'gnn.hip', 'match.hip', 'gnn_small.hip', 'gnn_small_bwd.hip', 'lstm.hip', 'arena.hip', 'state.hip',
        os.path.join('..', '..', 'include', 'stove_hip.h')]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for d in DEPS:
        p = os.path.join(CSRC, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_library(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-Wno-unused-value', '-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError('hipcc failed building libstove_hip.so')
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
