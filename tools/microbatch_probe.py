"""Measurement aid: would two 128-sequence micro-batches on two streams overlap one's T-serial recursion (one workgroup per
sequence: 128 of the 256 CUs) with the other's throughput kernels?  Forward only, no grad, eager launches (the forward costs the
host ~0.6 ms for ~1.3 ms of device work, so the host is not the limit).  Usage: python tools/microbatch_probe.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stove_amd.arena import ParamArena
from stove_amd.video_prediction.stove import Stove

dev = torch.device('cuda:0')
workload = sys.argv[1] if len(sys.argv) > 1 else 'billiards'
cfg = bench.build_config(workload, dev)
torch.manual_seed(0)
model = Stove(cfg).to(dev)
arena = ParamArena(model, 1)
data = bench.make_batch(workload, 256, 100, 0)
x = torch.from_numpy(data['X']).to(dev).contiguous()
xa, xb = x[:128].contiguous(), x[128:].contiguous()
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def fwd(t):
    with torch.no_grad():
        return model(t, 1, None)[0]


def full():
    fwd(x)


def halves_one_stream():
    fwd(xa)
    fwd(xb)


def halves_two_streams():
    main = torch.cuda.current_stream(dev)
    sa.wait_stream(main)
    sb.wait_stream(main)
    with torch.cuda.stream(sa):
        fwd(xa)
    with torch.cuda.stream(sb):
        fwd(xb)
    main.wait_stream(sa)
    main.wait_stream(sb)


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for name, fn in (('one batch of 256', full), ('two of 128, one stream', halves_one_stream), ('two of 128, two streams', halves_two_streams)):
    med, best = timed(fn)
    print('%-28s forward: median %.3f ms  min %.3f ms' % (name, med, best))
