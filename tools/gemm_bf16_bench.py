"""Times the split-bf16 MFMA GEMM (csrc/gemm_bf16.hip) on the recognition network's shapes at the headline batch
(25 600 frames) next to the fp32 library GEMM it replaces.  Usage: python tools/gemm_bf16_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stove_amd import ops

TILE = int(os.environ.get('GEMM_TILE', '0'))

dev = torch.device('cuda:0')
n = 25600


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2] * 1e3


x = torch.rand(n, 1024, device=dev)
w_ih = torch.randn(1024, 1024, device=dev) * 0.03
w_hh = torch.randn(1024, 256, device=dev) * 0.06
h = torch.randn(n, 256, device=dev)
dg = torch.randn(n, 1024, device=dev)
dg2 = torch.randn(2 * n, 1024, device=dev)
h2 = torch.randn(2 * n, 256, device=dev)
cases = [
    ('gx = x W_ih^T        (25600x1024x1024)', lambda ns: ops.gemm_bf16(x, w_ih, None, False, False, ns, 1, tile=TILE), lambda: x @ w_ih.t(), 2 * n * 1024 * 1024),
    ('gh = h W_hh^T        (25600x1024x256)', lambda ns: ops.gemm_bf16(h, w_hh, None, False, False, ns, 1, tile=TILE), lambda: h @ w_hh.t(), 2 * n * 1024 * 256),
    ('dh = dg W_hh         (25600x256x1024)', lambda ns: ops.gemm_bf16(dg, w_hh, None, False, True, ns, 1, tile=TILE), lambda: dg @ w_hh, 2 * n * 1024 * 256),
    ('dW_ih = dgx^T x      (1024x1024x25600, split-K 8)', lambda ns: ops.gemm_bf16(dg, x, None, True, True, ns, 8, tile=TILE), lambda: dg.t() @ x, 2 * n * 1024 * 1024),
    ('dW_hh = dg^T h       (1024x256x51200, split-K 32)', lambda ns: ops.gemm_bf16(dg2, h2, None, True, True, ns, 32, tile=TILE), lambda: dg2.t() @ h2, 4 * n * 1024 * 256),
]
for name, mine, lib, flops in cases:
    t2, t1, tl = timeit(lambda: mine(2)), timeit(lambda: mine(1)), timeit(lib)
    print('%-52s split3 %7.1f us (%6.1f TF fp32-equiv, %6.1f TF bf16)   bf16 %7.1f us (%6.1f TF)   library fp32 %7.1f us (%5.1f TF)' % (
        name, t2, flops / t2 / 1e6, 3 * flops / t2 / 1e6, t1, flops / t1 / 1e6, tl, flops / tl / 1e6))
if os.environ.get('GEMM_DEBUG'):
    for tile, what in ((1, 'full'), (11, 'no loads in the loop (compute + staging of stale registers)'), (12, 'no MFMAs (loads + staging + LDS reads)'),
                       (3, '256 x 256: full'), (13, '256 x 256: no loads in the loop'), (14, '256 x 256: no MFMAs'), (15, '256 x 256: no conversion / LDS writes (stale images)')):
        t = timeit(lambda: ops.gemm_bf16(x, w_ih, None, False, False, 2, 1, tile=tile))
        print('gx variant %-70s %7.1f us' % (what, t))
