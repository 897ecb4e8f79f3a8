"""Isolated timings of the fused LSTM launches (csrc/gemm_bf16.hip EPI 1 / 2, PERM 4) next to the product + stand-alone cell pairs
they replace, at the headline shape (25 600 frames, H = 256).  Usage: python tools/lstm_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stove_amd import _lib, ops
from stove_amd._lib import check, ptr, stream

dev = torch.device('cuda:0')
lib = _lib.load()
n, H, D = int(os.environ.get('N', 25600)), 256, 1024


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


f = lambda *s: torch.randn(*s, device=dev)
x, w_ih, w_hh, bias = torch.rand(n, D, device=dev), f(4 * H, D) * 0.03, f(4 * H, H) * 0.06, f(4 * H) * 0.1
h, c0, gx = f(n, H), f(n, H), f(n, 4 * H)
gs, c1, h1 = torch.empty(n, 4 * H, device=dev), torch.empty(n, H, device=dev), torch.empty(n, H, device=dev)
dg, dhs, dc, dc2, dgo = f(n, 4 * H) * 0.01, f(n, H) * 0.01, f(n, H) * 0.01, torch.empty(n, H, device=dev), torch.empty(n, 4 * H, device=dev)
dg_all, dgx = f(2, n, 4 * H) * 0.01, torch.empty(n, 4 * H, device=dev)
gw = torch.zeros(4 * H, D, device=dev)
gwh = torch.zeros(4 * H, H, device=dev)
hs2 = f(2 * n, H)


def fwd_fused(tile, A, W, K, add, cp, b):
    check(lib.stove_lstm_gemm_cell_fwd(ptr(A), ptr(W), ptr(b) if b is not None else None, ptr(add) if add is not None else None,
                                       ptr(cp) if cp is not None else None, ptr(gs), ptr(c1), ptr(h1), n, H, K, A.stride(0), 2, tile, stream()), 'f')


def bwd_fused(tile, more):
    check(lib.stove_lstm_gemm_cell_bwd(ptr(dg), ptr(w_hh), ptr(dhs), ptr(gx), ptr(c0), ptr(c1.normal_() if False else c0), ptr(dc), None if more else ptr(dgo), ptr(dc2),
                                       ptr(dgx) if more else None, ptr(dg_all) if more else None, 2 if more else 0, n, H, 2, tile, stream()), 'b')


def gate_rows(A, B, C, sk):
    ws = torch.empty(lib.stove_gemm_bf16_ws_floats(C.shape[0], C.shape[1], sk), device=dev)
    check(lib.stove_gemm_bf16_gate_rows(ptr(A), ptr(B), ptr(C), ptr(C), C.shape[0], C.shape[1], A.shape[0], A.stride(0), B.stride(0), 2, sk, ptr(ws), stream()), 'g')


print('F1 x W_ih^T + cell0 fused, tile 256x128: %7.1f us   tile 128x128: %7.1f us   | product %7.1f + cell %6.1f' % (
    timeit(lambda: fwd_fused(1, x, w_ih, D, None, None, bias)), timeit(lambda: fwd_fused(2, x, w_ih, D, None, None, bias)),
    timeit(lambda: ops.gemm_bf16(x, w_ih, bias=bias, nsplit=2, splitk=1)),
    timeit(lambda: check(lib.stove_lstm_cell_fwd(ptr(gx), None, None, ptr(c1), ptr(h1), n, H, stream()), 'c'))))
print('F2 h W_hh^T + gx + cell fused, tile 256x128: %7.1f us   tile 128x128: %7.1f us   | product %7.1f + cell %6.1f' % (
    timeit(lambda: fwd_fused(1, h, w_hh, H, gx, c0, None)), timeit(lambda: fwd_fused(2, h, w_hh, H, gx, c0, None)),
    timeit(lambda: ops.gemm_bf16(h, w_hh, nsplit=2, splitk=1, add=gx, tile=2)),
    timeit(lambda: check(lib.stove_lstm_cell_fwd(ptr(gx), None, ptr(c0), ptr(c1), ptr(h1), n, H, stream()), 'c'))))
print('B1 dg W_hh + cell bwd fused, tile 256x128: %7.1f us   tile 128x128: %7.1f us   | product %7.1f + cell %6.1f' % (
    timeit(lambda: bwd_fused(1, False)), timeit(lambda: bwd_fused(2, False)),
    timeit(lambda: ops.gemm_bf16(dg, w_hh, None, False, True, 2, 1, add=dhs)),
    timeit(lambda: check(lib.stove_lstm_cell_bwd(ptr(gx), None, ptr(c0), ptr(c0), ptr(dhs), ptr(dc), ptr(dgo), ptr(dc2), None, None, 0, n, H, stream()), 'c'))))
print('B1 last (with dgx sum over 2 more) fused, tile 256x128: %7.1f us   tile 128x128: %7.1f us   | product %7.1f + cell %6.1f' % (
    timeit(lambda: bwd_fused(1, True)), timeit(lambda: bwd_fused(2, True)),
    timeit(lambda: ops.gemm_bf16(dg, w_hh, None, False, True, 2, 1, add=dhs)),
    timeit(lambda: check(lib.stove_lstm_cell_bwd(ptr(gx), None, None, ptr(c0), ptr(dhs), ptr(dc), None, ptr(dc2), ptr(dgx), ptr(dg_all), 2, n, H, stream()), 'c'))))
print('cell bwd stand-alone, interleaved: %7.1f us' % timeit(lambda: check(lib.stove_lstm_cell_bwd_il(ptr(gx), ptr(c0), ptr(c0), ptr(dhs), None, ptr(dgo), ptr(dc2), None, None, 0, n, H, stream()), 'c')))
print('dW_ih gate rows %7.1f us | plain %7.1f us' % (timeit(lambda: gate_rows(dg, x, gw, 8)), timeit(lambda: ops.gemm_bf16(dg, x, None, True, True, 2, out=gw))))
print('dW_hh gate rows %7.1f us | plain %7.1f us' % (timeit(lambda: gate_rows(dg_all.view(-1, 4 * H), hs2, gwh, 32)),
                                                     timeit(lambda: ops.gemm_bf16(dg_all.view(-1, 4 * H), hs2, None, True, True, 2, out=gwh))))
