"""Compile the HIP sources into stove_amd/libstove_hip.so (in-tree, gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libstove_hip.so')
SOURCES = ['capi.hip']


def _deps():
    """Every file under csrc/ (capi.hip includes the other .hip files) plus the public header."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))) + [os.path.join('..', '..', 'include', 'stove_hip.h')]


STAMP = LIB + '.sha256'


def source_hash():
    """Content hash of every source the library is built from (mtimes say nothing after a fresh copy of the tree)."""
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        p = os.path.join(CSRC, d)
        h.update(d.encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build_library(force=False, verbose=False):
    """Build if the library is missing or was built from other sources.  Spawns hipcc: call it BEFORE the process touches
    the GPU (bench.py and smoke() do)."""
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
    with open(STAMP, 'w') as f:
        f.write(source_hash() + '\n')
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
