"""The recognition network's forward chain in row chunks (ops._encoder_lstm_fwd_chunked) against the unchunked chain:
bitwise equality of h / c, and the time of the forward alone.  Usage: python tools/enc_chunks_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stove_amd import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)
n, D, H, K = 25600, 1024, 256, int(os.environ.get('OBJ', '3'))
x = torch.rand(n, D, device=dev)
w_ih = torch.randn(4 * H, D, device=dev) * 0.03
w_hh = torch.randn(4 * H, H, device=dev) * 0.06
b_ih = torch.randn(4 * H, device=dev) * 0.1
b_hh = torch.randn(4 * H, device=dev) * 0.1


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


ref = None
for chunks, stagger in ((1, True), (2, True), (2, False), (4, True), (4, False), (1, True)):
    ops.ENC_CHUNKS, ops.ENC_STAGGER = chunks, stagger
    with torch.no_grad():
        hs = ops.encoder_lstm(x, w_ih, w_hh, b_ih, b_hh, K, time_major=True)
        torch.cuda.synchronize()
        if ref is None:
            ref = hs.clone()
        same = torch.equal(hs, ref)
        t = timeit(lambda: ops.encoder_lstm(x, w_ih, w_hh, b_ih, b_hh, K, time_major=True))
    print('chunks %d stagger %d: forward %7.1f us   bit-identical to unchunked: %s' % (chunks, stagger, t, same))
