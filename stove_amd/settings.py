"""Process-wide switches of the hot path, read ONCE at import.

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
