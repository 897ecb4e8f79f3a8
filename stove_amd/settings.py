"""Process-wide switches of the hot path, read from the environment ONCE at import; `set_overlap` changes the one switch afterwards.

There is one measurement switch: STOVE_NO_OVERLAP=1 runs every kernel on one stream, one after the other (the serial
timeline of tools/kernel_times.sh: each kernel's time alone).  Everything else that used to be switchable from the
environment was an A/B experiment whose losing side is recorded in docs/history/ and has been removed (round 5).
STOVE_DIST_BACKEND (main.py, bench.py) picks the torch.distributed backend for tests on a one-GPU box.
"""
import os

OVERLAP = os.environ.get('STOVE_NO_OVERLAP', '0') != '1'
# STOVE_LIB=/path/to/another/libstove_hip.so: load that build instead of the installed one (tools/ab_lib.sh alternates builds on one
# box without ever overwriting the installed file); its ABI version is checked like the installed library's.
LIB_OVERRIDE = os.environ.get('STOVE_LIB') or None


def set_overlap(on):
    """Switch the multi-stream overlap on / off AFTER import -- the Python-side flag the ops read at every call and the library's own
    (stove_set_overlap), together.  A test or a bench variant that sets STOVE_NO_OVERLAP in the environment of a process that has
    already imported this module changes nothing; this does.  Returns the previous setting."""
    global OVERLAP
    prev, OVERLAP = OVERLAP, bool(on)
    from . import _lib
    if _lib._lib is not None:                # (not loaded yet: _lib.load() applies OVERLAP when it does)
        _lib._lib.stove_set_overlap(1 if OVERLAP else 0)
    return prev
