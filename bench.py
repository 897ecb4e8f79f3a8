#!/usr/bin/env python
"""Headline benchmark: frames/s of one STOVE training step on synthetic billiards video.

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a rendezvous in the environment: starts N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: Stove.forward (encoder, matching, fused
inference recursion, both fused scene likelihoods) + elbo.backward() + [N>1: one RCCL all-reduce
of the flat gradient] + clip_grad_norm_(1) + Adam(amsgrad) step, i.e. the reference's training
step (train.py:443-473), run the way Trainer.train runs its non-logging steps: replayed as captured hipGraphs
(stove_amd/graphed.py; --step-mode eager enqueues every launch from the host and is reported under `variants`).
Workload (BASELINE.json configs[1]): 3-object billiards, 32x32, T=100, batch 256 per GPU (weak scaling), frames from the
build's numpy simulator, model with default initialisation; inputs are resident in HBM before the timed region, in the
format of the Trainer's device frame store (--frame-store: the bw plane by default, see load_data.DeviceClipLoader).
Rank 0 prints ONE JSON line (contract in the task statement); `roofline` is measured with HIP
events around the dominant kernel, `cpu_baseline` times the CPU oracle at the quoted batch with the best of 8/32/64 threads.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--batch', type=int, default=256, help='sequences per GPU')
    p.add_argument('--frames', type=int, default=100, help='T, frames per sequence')
    p.add_argument('--workload', default='billiards', choices=['billiards', 'multibilliards', 'gravity', 'avoidance'])
    p.add_argument('--res', type=int, default=32, help='frame side; 32 = BASELINE.json (fused scene pipeline), anything else runs the general-size '
                   'likelihood path (the reference\'s stock gravity / multibilliards data are 50 x 50): a side measurement, never the headline')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-batch', type=int, default=32)
    p.add_argument('--cpu-iters', type=int, default=8)
    p.add_argument('--profile-steps', type=int, default=3)
    p.add_argument('--encoder-gemm', default='bf16x3', choices=['bf16x3', 'fp32', 'bf16'],
                   help='recognition-network GEMMs: bf16x3 = fp32 products as 3 bf16 MFMAs on hi/lo-split operands (default), '
                        'fp32 = library GEMMs, bf16 = plain bf16 operands (reported variant, never the headline)')
    p.add_argument('--no-variants', action='store_true', help='skip the bf16-operand / fp32-library side measurements')
    p.add_argument('--frame-store', default='auto', choices=['auto', 'bw32', 'f32', 'u8'],
                   help="how the resident frames are kept, as config.frame_store of the Trainer's DeviceClipLoader: auto = bw32 (the bw plane, "
                        "bw_transform applied once at upload: bit-identical model input) when the model only consumes bw frames, else f32 colour; "
                        "u8 = 8-bit colour frames converted by the step's first kernel")
    p.add_argument('--step-mode', default='graph', choices=['graph', 'eager'],
                   help='graph (default): the step replayed as captured hipGraph(s), as Trainer.train runs its non-logging steps '
                        '(stove_amd/graphed.py); eager: every launch enqueued by the host (reported as a variant)')
    return p.parse_args()


def build_config(workload, device, res=32):
    from stove_amd.video_prediction.config import StoveConfig
    c = StoveConfig()
    c.width, c.height, c.channels = res, res, 1
    c.device, c.dtype = device, torch.float32
    c.random_seed = 42
    c.skip = 2
    c.print_every, c.plot_every = 10 ** 9, 1e19          # no logging side channel inside the timed region
    c.num_obj, c.action_conditioned, c.action_space = 3, False, None
    if workload == 'multibilliards':
        c.num_obj, c.debug_match_objects, c.overlap_beta, c.max_obj_scale = 6, 'greedy', 100.0, 0.22
    if workload == 'avoidance':
        c.action_conditioned, c.action_space, c.debug_core_appearance = True, 9, True
    return c


def make_batch(workload, n_seq, T, seed0, res=32):
    from stove_amd.envs import envs
    cache = os.path.join('/tmp', f'stove_bench_{workload}_{n_seq}_{T}_{seed0}' + ('' if res == 32 else f'_r{res}') + '.npz')
    if os.path.exists(cache):
        d = dict(np.load(cache))
    else:
        d = envs.synth_sequences(workload, n_seq, T, seed0=seed0, res=None if res == 32 else res)
        try:
            np.savez(cache, **d)
        except OSError:
            pass
    return d


def cpu_baseline(workload, T, n_seq, iters, full_batch=None, full_iters=2):
    """Time the CPU oracle (oracle/stove_oracle.py: the reference's ATen op sequence restated) on the host cores: same workload
    shape.  A bounded sample (B = n_seq) finds the thread count the path runs fastest with -- the reference pins torch to
    config.max_threads = 8 (config.py:59, main.py:134); 8 / 32 / 64 are tried -- then the batch the metric is
    quoted on (B = full_batch): one warm-up iteration, then the median of `full_iters` timed ones, is the reported `value`."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import stove_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    kw = {}
    if workload == 'multibilliards':
        kw = dict(num_obj=6, debug_match_objects='greedy', overlap_beta=100.0, max_obj_scale=0.22)
    c = O.default_config(**kw)
    structs = O.build_structs(c)
    torch.manual_seed(0)
    params = {}
    for k, shp in O.param_shapes(c, structs).items():
        scale = 0.1 if k.endswith(('means', 'sigma_params', 'params')) else 1.0 / max(1.0, float(shp[-1])) ** 0.5
        params[k] = (torch.randn(*shp) * scale).requires_grad_()
    g = torch.Generator().manual_seed(1)

    def one(x, nb):
        eps = O.draw_eps(nb, c.num_obj, T, generator=g)
        t0 = time.perf_counter()
        elbo, _ = O.stove_forward(c, params, structs, x, eps)
        (-elbo).backward()
        dt = time.perf_counter() - t0
        for p in params.values():
            p.grad = None
        return dt
    x = torch.from_numpy(make_batch(workload, n_seq, T, 10 ** 6)['X'])
    sweep = {}
    t_start = time.perf_counter()
    for threads in [t for t in (8, 32, 64) if t <= avail] or [min(8, avail)]:
        torch.set_num_threads(threads)
        times = [one(x, n_seq) for _ in range(3 if not sweep else 2)]        # the very first iteration also warms the allocator up
        sweep[threads] = min(times[1:]) if len(times) > 2 else min(times)
        if time.perf_counter() - t_start > 25.0:
            break
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    med = sweep[cores]
    out = {'value': n_seq * T / med, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
           'sample': f'{workload} B={n_seq} T={T} fp32 fwd+bwd, best of 2 after warm-up, {med:.2f} s/step',
           'cpu_model': _cpu_model(), 'cores_total': os.cpu_count(), 'cores_available': avail,
           'thread_sweep_frames_per_s': {str(k): round(n_seq * T / v, 1) for k, v in sweep.items()}}
    # the figure the metric is quoted on: B = full_batch with the best thread count (when the sample says it fits): one warm-up
    # iteration (allocator, first-touch of the 100x larger intermediates), then the median of `full_iters` timed ones
    if full_batch and full_batch > n_seq and med * full_batch / n_seq < 60.0:
        xf = torch.from_numpy(make_batch(workload, full_batch, T, 0)['X'])
        warm = one(xf, full_batch)
        times = sorted(one(xf, full_batch) for _ in range(max(1, full_iters)))
        dt = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
        out['sample_batch'] = {'value': out['value'], 'sample': out['sample']}
        out['value'] = full_batch * T / dt
        out['iterations_s'] = {'warmup': round(warm, 2), 'timed': [round(t, 2) for t in times]}
        out['sample'] = (f'{workload} B={full_batch} T={T} fp32 fwd+bwd, median of {len(times)} iterations ({dt:.1f} s) after one warm-up '
                         f'iteration, {cores} threads (the best of the 8/32/64 sweep on a B={n_seq} sample)')
    return out


def reference_parity(dev, encoder_gemm):
    """ELBO of the bench's own model path on the reference-generated fixture g7_stove_n3 (BASELINE.json configs[0]: B = 4, T = 8,
    three-object billiards; frames, injected noise and the reference's fp64 ELBO in tests/golden/g7_stove_n3_f64.npz, weights from
    the fixtures' analytic fill) -> |elbo - elbo_ref| / |elbo_ref|.  Reads the committed fixture only: neither the oracle nor the
    reference runs here."""
    import importlib.util
    gdir = os.path.join(ROOT, 'tests', 'golden')
    spec = importlib.util.spec_from_file_location('_golden_analytic_weights', os.path.join(gdir, 'analytic_weights.py'))
    aw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(aw)
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.stove import Stove
    gold = dict(np.load(os.path.join(gdir, 'g7_stove_n3_f64.npz')))
    cfg = build_config('billiards', dev)
    cfg.encoder_gemm = encoder_gemm
    model = Stove(cfg)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(aw.analytic_tensor(name, p.shape, torch.float64).float())
    model = model.to(dev)
    ParamArena(model, 1)                     # the flat arena path the timed step runs
    f = lambda a: torch.from_numpy(np.asarray(a)).float().to(dev)
    table = {'latent': f(gold['eps_lat'])[..., 0], 'std': f(gold['eps_std'])[..., 0], 'steps': f(gold['eps_steps']).permute(1, 0, 2, 3).contiguous()}
    model.noise_fn = lambda kind, shape: table[kind].reshape(shape)
    elbo, _, _ = model(f(gold['x']), 1, None)
    (-elbo).backward()
    e, ref = float(elbo.detach()), float(gold['elbo'])
    return {'fixture': 'tests/golden/g7_stove_n3_f64.npz (reference Stove.forward, fp64, B=4 T=8, injected noise)', 'elbo': e, 'elbo_reference': ref,
            'elbo_rel_vs_reference': abs(e - ref) / abs(ref), 'bar': 1e-4}


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def log(msg):
    sys.stderr.write('[bench %.1fs] %s\n' % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


_T0 = time.perf_counter()


def launch_ranks(a):
    """`python bench.py --gpus N` with no rendezvous in the environment: start N ranks (one per GPU) with torch.distributed.run and
    pass their output through.  Runs BEFORE this process touches the GPU (it never does: the workers are fresh processes)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    from stove_amd import build as _build
    _build.build_library()                      # once, here, instead of N ranks racing for it
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    log('launching %d ranks: %s' % (a.gpus, ' '.join(cmd[1:])))
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(a))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # hipcc (if the library is stale; keyed on a content hash of the sources) runs BEFORE this process touches the GPU:
    # torch.cuda.device_count() does not initialise HIP, torch.cuda.is_available() / set_device below do
    from stove_amd import build as _build
    if local == 0:
        _build.build_library()
    else:
        t_wait = time.time()
        while _build._stale():
            if time.time() - t_wait > 600:
                raise SystemExit('libstove_hip.so was not built by local rank 0 within 10 minutes')
            time.sleep(0.5)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the STOVE hot path has no CPU fallback')
    # one process per GPU; STOVE_DIST_BACKEND=gloo lets the multi-process path be exercised on a
    # single-GPU box (ranks then share cuda:0) -- the driver's runs use RCCL ('nccl')
    backend = os.environ.get('STOVE_DIST_BACKEND', 'nccl')
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from stove_amd import _lib
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.stove import Stove

    cfg = build_config(a.workload, dev, a.res)
    cfg.encoder_gemm = a.encoder_gemm
    if os.environ.get('STOVE_PIECES'):
        cfg.pipeline_pieces = int(os.environ['STOVE_PIECES'])
    torch.manual_seed(0)
    model = Stove(cfg).to(dev)
    bucket = ParamArena(model, world)          # parameters / gradients flat; grad buffer == all-reduce bucket
    bucket.sync(0)                             # replicas start from rank 0's parameters (one broadcast of the flat buffer)
    from stove_amd.optim import FlatAdam
    opt = FlatAdam(bucket, lr=cfg.learning_rate, amsgrad=cfg.debug_amsgrad)       # torch.optim.Adam's update as one launch

    log('model built; generating data')
    data = make_batch(a.workload, a.batch, a.frames, rank * a.batch, a.res)
    log('data ready')
    x = torch.from_numpy(data['X']).to(dev).contiguous()       # a batch as the DataLoader collates it (contiguous n,T,C,w,h)
    # ... and as the Trainer's device-resident frame store hands it over (load_data.DeviceClipLoader, config.frame_store)
    fs = a.frame_store
    bw_only = not (cfg.debug_core_appearance or cfg.debug_match_appearance)
    if fs == 'auto':
        fs = 'bw32' if bw_only else 'f32'
    if fs == 'bw32':
        if not bw_only:
            raise SystemExit('--frame-store bw32 needs a workload without appearance features')
        from stove_amd.utils.utils import bw_transform
        x = bw_transform(x)
        cfg.input_bw_plane = True
    elif fs == 'u8':
        x = torch.round(x * 255).to(torch.uint8)
    actions = torch.from_numpy(data['action']).float().to(dev) if 'action' in data else None
    torch.manual_seed(1234 + rank)

    minus_one = torch.tensor(-1.0, device=dev)

    def eager_step(i):
        bucket.zero()
        elbo, _, rewards = model(x, i + 1, actions)
        elbo.backward(minus_one)                                 # d(-ELBO): the loss of train.py:452 without the neg / fill launches
        bucket.all_reduce()
        opt.step(max_norm=1.0)                                    # clip_grad_norm_(1) folded into the Adam launch
        return elbo

    # the step as the Trainer runs it between logging steps: captured once, replayed (two graphs around the all-reduce for N > 1)
    from stove_amd.graphed import GraphedTrainStep
    graphed = GraphedTrainStep(model, bucket, opt, 1.0, world_size=world, alias_inputs=True)

    def graph_step(i):
        return graphed(x, actions)

    step = graph_step if a.step_mode == 'graph' else eager_step
    if a.step_mode == 'graph':
        step(0)                                   # capture (restores parameters / optimiser / generator: not a training step)
        torch.cuda.synchronize()
        log('step captured')
    if a.warmup > 0 and not os.environ.get('STOVE_BENCH_NO_GC_SETTLE'):
        # Host-side settling BEFORE the warm-up steps, not between them and the timed region: the collection takes tens of ms of host
        # time with the device idle, the chip drops its clocks, and the first timed steps then ran 3.34 / 3.17 / 3.04 ms against 2.90
        # steady (STOVE_BENCH_SERIES=1).  The warm-up steps now run straight into the timed region.
        if a.step_mode != 'graph':
            step(0)                               # eager mode: one step so that the long-lived objects exist (setup, as the capture is)
        torch.cuda.synchronize()
        from stove_amd.utils.utils import settle_host_gc
        settle_host_gc()            # as the Trainer does after its first steps (train.py): see the function's note
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    log('warm-up done')
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]       # device-side step boundaries (no host sync)
    t0 = time.perf_counter()
    marks[0].record()
    host_t = [t0]
    import gc
    if os.environ.get('STOVE_BENCH_GCLOG'):
        _g = {}
        def _cb(phase, info):
            if phase == 'start':
                _g['t'] = time.perf_counter()
            else:
                log('gc gen%d %.2f ms collected %d' % (info['generation'], (time.perf_counter() - _g['t']) * 1e3, info['collected']))
        gc.callbacks.append(_cb)
    if os.environ.get('STOVE_BENCH_WATCH'):           # where is the host when a step stalls: sample every thread's stack
        import threading, traceback
        main = threading.main_thread()
        pid = os.getpid()
        thresh = float(os.environ.get('STOVE_BENCH_WATCH_MS', '6')) * 1e-3

        def _proc(tid, name):
            try:
                with open('/proc/%d/task/%d/%s' % (pid, tid, name)) as f:
                    return f.read().strip()[:200]
            except OSError as e:
                return 'n/a (%s)' % e.__class__.__name__

        def _watch():
            seen = -1
            while len(host_t) <= a.steps:
                time.sleep(0.001)
                k = len(host_t) - 1
                late = time.perf_counter() - host_t[-1]
                if late > thresh and k != seen:
                    seen = k
                    out = ['step %d: host %.1f ms into it' % (k, late * 1e3)]
                    frames = sys._current_frames()
                    for th in threading.enumerate():
                        if th is threading.current_thread():
                            continue
                        fr = frames.get(th.ident)
                        out.append('  thread %s tid %s state/wchan %s / %s syscall %s' % (
                            th.name, th.native_id, _proc(th.native_id, 'stat').split(') ')[-1][:1], _proc(th.native_id, 'wchan'),
                            _proc(th.native_id, 'syscall')))
                        if fr is not None:
                            out.append(''.join('    ' + l for l in ''.join(traceback.format_stack(fr)[-7:]).splitlines(True)))
                    # native threads python does not know (HIP / HSA workers, the autograd engine's pool)
                    try:
                        known = {th.native_id for th in threading.enumerate()}
                        for tid in sorted(int(t) for t in os.listdir('/proc/%d/task' % pid)):
                            if tid not in known:
                                st = _proc(tid, 'stat')
                                out.append('  native tid %d %s state %s wchan %s syscall %s' % (
                                    tid, st[st.find('('):st.find(')') + 1], st.split(') ')[-1][:1], _proc(tid, 'wchan'), _proc(tid, 'syscall')[:60]))
                    except OSError:
                        pass
                    log('\n'.join(out))
        threading.Thread(target=_watch, daemon=True, name='watch').start()
    for i in range(a.steps):
        last = step(a.warmup + i)
        marks[i + 1].record()
        host_t.append(time.perf_counter())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    elbo_val = float(last.detach())
    series = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    per_step_ms = sorted(series)
    if per_step_ms[-1] > 2.0 * per_step_ms[len(per_step_ms) // 2]:      # a stalled step: say which, and what the host was doing
        log('stalled step(s): device ms ' + ' '.join('%.2f' % v for v in series))
        log('                 host ms   ' + ' '.join('%.2f' % ((host_t[i + 1] - host_t[i]) * 1e3) for i in range(a.steps)))
    log('timed region done: %.1f ms/step' % (dt / a.steps * 1e3))
    if os.environ.get('STOVE_BENCH_SERIES'):
        log('device ms per step: ' + ' '.join('%.3f' % v for v in series))
        log('host ms per step:   ' + ' '.join('%.3f' % ((host_t[i + 1] - host_t[i]) * 1e3) for i in range(a.steps)))

    # ---- per-kernel HIP-event timing of extra steps (profiling hooks off during the timed region)
    roofline = None
    lib = _lib.load()
    prof = None
    if a.profile_steps > 0:
        # every rank runs the extra steps (they contain the collective); only rank 0 records events
        if rank == 0:
            lib.stove_profile_enable(1)
        for i in range(a.profile_steps):
            eager_step(a.warmup + a.steps + i)       # event pairs around single launches: not through a captured graph
        torch.cuda.synchronize()
        if rank == 0:
            prof_w = _lib.profile_report(wall=True)
            prof = {k: v[:2] for k, v in prof_w.items()}
        # the same steps with the recognition network's forward chain on ONE stream (ops.ENC_CHUNKS = 1): every GEMM launch alone on
        # the chip -- the like-for-like launch time of the kernel the roofline is quoted on (with the chain in two row chunks on two
        # streams, as the timed region runs it, launches of that kernel overlap and each is timed with the other beside it)
        prof_serial = None
        from stove_amd import ops as _ops
        chunks_saved = _ops.ENC_CHUNKS
        if chunks_saved > 1:
            _ops.ENC_CHUNKS = 1
            if rank == 0:
                lib.stove_profile_enable(0)
            eager_step(a.warmup + a.steps + 2 * a.profile_steps)          # the allocator meets the other schedule's sizes untimed
            torch.cuda.synchronize()
            if rank == 0:
                lib.stove_profile_enable(1)
            for i in range(a.profile_steps):
                eager_step(a.warmup + a.steps + a.profile_steps + i)
            torch.cuda.synchronize()
            _ops.ENC_CHUNKS = chunks_saved
            if rank == 0:
                prof_serial = _lib.profile_report()
        if rank == 0:
            lib.stove_profile_enable(0)
    if rank == 0 and a.profile_steps > 0:
        if prof:
            name, (total_ms, count) = max(prof.items(), key=lambda kv: kv[1][0])
            n_obj = cfg.num_obj
            avg_ms = total_ms / count
            per_step = {k: round(v[0] / a.profile_steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:24]}
            if name.startswith('gemm_bf16'):
                # recognition-network GEMMs (csrc/gemm_bf16.hip), all launches of a step together (they differ in shape):
                # algorithmic flops = 2 M N K of the fp32 products -- x W_ih^T, (N-1) x h W_hh^T forward; (N-1) x dg W_hh,
                # dg^T h over the N-1 recurrent steps, dgx^T x backward (the head's 50-wide products run in enc_head_*_k on the fp32
                # matrix cores) -- against the dense bf16 MFMA peak (2.5 PFLOP/s,
                # MI355X_MICROARCH.md); the split-bf16 path issues 3 bf16 MFMA flops per algorithmic flop.
                nfr, Hh, Dd = a.batch * a.frames, 256, 32 * 32
                flops = 2.0 * nfr * (2 * Dd * 4 * Hh + 3 * (n_obj - 1) * Hh * 4 * Hh)
                ms_step = total_ms / a.profile_steps
                ach = flops / (ms_step * 1e-3) / 1e12
                passes = {'bf16x3': 3, 'bf16': 1}.get(a.encoder_gemm, 3)
                wall_step = prof_w[name][2] / a.profile_steps
                roofline = {'bound': 'mfma', 'kernel': name, 'avg_ms': avg_ms, 'launches': count, 'ms_per_step': ms_step,
                            'achieved': ach, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': ach / 2500.0, 'traffic': None,
                            'mfma_pipe_frac': passes * ach / 2500.0,
                            'ms_per_step_covered': wall_step, 'frac_covered': flops / (wall_step * 1e-3) / 1e12 / 2500.0,
                            'note': 'achieved = algorithmic fp32 flops (2MNK of the %d GEMM launches of a step) / their SUMMED launch time; the kernel '
                                    'issues %d bf16 MFMA flops per algorithmic flop (hi/lo split), so the matrix pipe runs at mfma_pipe_frac.  The '
                                    'forward chain runs as two row chunks on two streams: launches of the kernel overlap in time and each is timed '
                                    'with the other beside it, so the sum exceeds the time the launches cover (ms_per_step_covered, the union of '
                                    'their event spans; frac_covered = the same flops over that time)' % (count // a.profile_steps, passes),
                            'kernels_ms_per_step': per_step}
                if prof_serial and name in prof_serial:
                    ms1 = prof_serial[name][0] / a.profile_steps
                    roofline['one_stream'] = {'launches': prof_serial[name][1], 'ms_per_step': ms1, 'achieved': flops / (ms1 * 1e-3) / 1e12,
                                              'frac': flops / (ms1 * 1e-3) / 1e12 / 2500.0,
                                              'note': 'the same steps with the forward chain on one stream (STOVE_ENC_CHUNKS=1): no two launches '
                                                      'of the kernel overlap; measured in this run, after the passes above'}
            elif name.startswith(('dyn_loop', 'gnn_step', 'rollout', 'gnn_dw')):
                # GNN recursion: dense fp32 contraction -> 157.3 TFLOP/s (v_mfma_f32_16x16x4_f32; the packed-fp32 VALU
                # path of the small-graph kernels, v_pk_fma_f32, has the same peak on MI355X).
                # Algorithmic flops per (sequence, step), SURVEY.md section 8d: F = 17408 N + 26944 N (N-1) forward;
                # the backward is 2F: F of data gradients (dyn_loop_bwd_small_k) + F of weight gradients
                # (gnn_dw_small_k); the MFMA kernel of gnn.hip (N > 4) does both in one launch.
                F = 17408 * n_obj + 26944 * n_obj * (n_obj - 1)
                units = a.batch * (a.frames - 2)
                both = 'bwd' in name and 'small' not in name
                flops = (2 * F if both else F) * units
                ach = flops / (avg_ms * 1e-3) / 1e12
                roofline = {'bound': 'mfma', 'kernel': name, 'avg_ms': avg_ms, 'launches': count, 'achieved': ach,
                            'peak': 157.3, 'unit': 'TFLOP/s', 'frac': ach / 157.3, 'traffic': None,
                            'note': 'latency-bound: %d dependent time steps per launch, one sequence per CU' % (a.frames - 2),
                            'kernels_ms_per_step': per_step}
            else:
                # SPN / scene sweep: scan-shaped -> HBM 8 TB/s.  Algorithmic bytes per frame fwd+bwd =
                # 8200 + 32 N; one direction of it per launch.
                frames_per_launch = a.batch * (a.frames - 1)
                alg_bytes = (8200 + 32 * n_obj) / 2 * frames_per_launch
                ach = alg_bytes / (avg_ms * 1e-3) / 1e9
                roofline = {'bound': 'hbm', 'kernel': name, 'avg_ms': avg_ms, 'launches': count, 'achieved': ach,
                            'peak': 8000.0, 'unit': 'GB/s', 'frac': ach / 8000.0, 'traffic': None,
                            'note': 'VALU-bound sweep (~120 flop/B, ridge ~20 flop/B)', 'kernels_ms_per_step': per_step}
    if roofline is not None and prof and not roofline['kernel'].startswith('dyn_loop'):
        # the T-serial recursion (latency-bound): data-gradient kernel of the backward against the fp32 MFMA / VALU peak
        for k in ('dyn_loop_bwd_small_k', 'dyn_loop_bwd_k'):
            if k in prof:
                F = 17408 * cfg.num_obj + 26944 * cfg.num_obj * (cfg.num_obj - 1)
                fl = (2 * F if k == 'dyn_loop_bwd_k' else F) * a.batch * (a.frames - 2)
                ms = prof[k][0] / prof[k][1]
                roofline['recursion'] = {'kernel': k, 'avg_ms': ms, 'achieved': fl / (ms * 1e-3) / 1e12, 'peak': 157.3, 'unit': 'TFLOP/s',
                                         'frac': fl / (ms * 1e-3) / 1e12 / 157.3, 'note': 'latency-bound: %d dependent time steps per launch' % (a.frames - 2)}
    if roofline is not None and prof:
        # second kernel family of SURVEY.md section 8d: the SPN / scene sweep (all its launches of one step together) against
        # HBM with the algorithmic 8200 + 32 N bytes per frame forward + backward
        spn_ms = sum(v[0] for k, v in prof.items() if k.startswith(('objspn_', 'bgspn_', 'bg_', 'scene_', 'spn_bake', 'reduce_chunks'))) / a.profile_steps
        if spn_ms > 0:
            alg = (8200 + 32 * cfg.num_obj) * a.batch * (a.frames - 1)
            ach = alg / (spn_ms * 1e-3) / 1e9
            roofline['spn_sweep'] = {'bound': 'hbm', 'ms_per_step': round(spn_ms, 4), 'achieved': ach, 'peak': 8000.0, 'unit': 'GB/s',
                                     'frac': ach / 8000.0, 'note': 'launch durations summed (the chains overlap on three streams, so this exceeds their wall time); '
                                             'fp32-VALU bound: ~120 flop/B against a ridge of ~20 flop/B'}
    if roofline is not None:
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (if present)
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_traffic.json')))[-1:]:
            try:
                doc = json.load(open(path))
                tr = doc.get(roofline['kernel'])
                if doc.get('_source_hash') != _build.source_hash():
                    # counters of a library built from other sources are not this run's traffic: refuse them
                    roofline['traffic_source'] = os.path.basename(path) + ' (stale: profiled on other kernel sources, not quoted)'
                    tr = None
                if tr and a.workload == 'billiards' and a.batch == 256 and a.frames == 100:
                    roofline['traffic'] = tr['hbm_bytes_per_launch']
                    roofline['traffic_source'] = os.path.basename(path)
                    # the SPN / scene family of the same passes: bytes per launch x launches per step, summed over its kernels
                    steps_prof = doc.get('_steps_profiled') or (doc.get('flat_adam_k') or {}).get('launches_profiled')
                    if steps_prof and 'spn_sweep' in roofline:
                        fam = {k: v for k, v in doc.items() if isinstance(v, dict) and k.startswith(
                            ('objspn_', 'bgspn_', 'bg_', 'scene_', 'spn_bake', 'reduce_chunks'))}
                        roofline['spn_sweep']['traffic'] = sum(v['hbm_bytes_per_launch'] * v['launches_profiled'] for v in fam.values()) / steps_prof
                        roofline['spn_sweep']['algorithmic_bytes'] = (8200 + 32 * cfg.num_obj) * a.batch * (a.frames - 1)
                        roofline['spn_sweep']['traffic_note'] = 'HBM bytes per STEP of the family (counter passes, FETCH_SIZE x 2 + WRITE_SIZE)'

                    # the PMC passes serialise kernels and cannot run inside a timed bench: the figure is read from the
                    # committed summary of tools/profile_round.sh on this workload, not measured by this process
                    roofline['traffic_measured_in_run'] = False
            except (OSError, ValueError):
                pass
    log('kernel profile done')
    # ---- data parallel: what the collective costs by itself (HIP events around the all-reduce of the flat gradient, the stream
    # otherwise idle) and what RCCL says about the group, so that a multi-GPU line explains itself
    comm = None
    if world > 1:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 20
        for _ in range(3):
            bucket.all_reduce()
        torch.cuda.synchronize()
        dist.barrier()
        ev[0].record()
        for _ in range(reps):
            bucket.all_reduce()
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        t = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm = {'all_reduce_ms': float(t.item()), 'bytes': int(bucket.grad.numel() * 4), 'world_size_reported': dist.get_world_size(),
                'backend': dist.get_backend(), 'devices_visible': torch.cuda.device_count(),
                'note': 'one all-reduce (sum) of the flat fp32 gradient + the 1/world scale per step, between the backward graphs and the optimiser graph; '
                        'max over ranks of the mean of %d back-to-back calls' % reps}
    # ---- side measurements (N = 1 only): the same step with the recognition network's GEMMs on plain bf16 operands
    # (BASELINE.json configs[1] says "bf16"; SURVEY section 7: reported, not assumed) and on the fp32 library path, each with
    # its ELBO difference against the fp32 library path on THIS batch under identical noise.  Never the headline `value`.
    variants = None
    if rank == 0 and world == 1 and not a.no_variants and a.res == 32:
        g = torch.Generator(device='cpu').manual_seed(99)
        o = cfg.num_obj
        fixed = {'latent': torch.randn(a.batch, o, 12, generator=g).to(dev), 'std': torch.randn(a.batch, o, 12, generator=g).to(dev),
                 'steps': torch.randn(a.batch, a.frames - 2, o, 18, generator=g).to(dev)}
        snap = (bucket.data.clone(), {k: v.clone() for k, v in opt._flat.items()}, opt._seg_steps.clone())

        def elbo_of(mode):
            cfg.encoder_gemm = mode
            model.noise_fn = lambda kind, shape: fixed[kind].reshape(shape)
            with torch.no_grad():
                e, _, _ = model(x, 1, actions)
            model.noise_fn = None
            return float(e)

        def time_of(mode, eager=False):
            """median device time per step (event pairs): one stall of the host or the allocator after the mode switch must not
            decide a side measurement"""
            cfg.encoder_gemm = mode
            vstep = eager_step
            if a.step_mode == 'graph' and not eager:
                g_ = GraphedTrainStep(model, bucket, opt, 1.0, world_size=world, alias_inputs=True)
                vstep = lambda i: g_(x, actions)
            for i in range(3):
                vstep(i)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
            ev[0].record()
            for i in range(a.steps):
                vstep(i)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps))
            return ts[len(ts) // 2], ts[-1]

        e_ref = elbo_of('fp32')
        variants = {}
        for mode, label in (('bf16', 'bf16-operands (encoder GEMMs), fp32 accumulate'), ('bf16x3', 'split-bf16 x3 (default path)'),
                            ('fp32', 'fp32 library GEMMs')):
            e = elbo_of(mode)
            ms, ms_max = time_of(mode)
            variants[mode] = {'dtype': label, 'ms_per_step': ms, 'ms_per_step_max': ms_max, 'value': a.batch * a.frames / ms * 1e3, 'unit': 'frames/s',
                              'elbo': e, 'elbo_rel_delta_vs_fp32_library': abs(e - e_ref) / abs(e_ref)}
            with torch.no_grad():       # the timing steps trained: put the parameters and the optimiser state back
                bucket.data.copy_(snap[0])
                for k, v in snap[1].items():
                    opt._flat[k].copy_(v)
                opt._seg_steps.copy_(snap[2])
        cfg.encoder_gemm = a.encoder_gemm
        if fs != 'f32':
            # colour fp32 frames as the reference's loader hands them over: bw_transform runs INSIDE every step (reference
            # stove.py:885-886) -- what the headline's bw-plane store skips
            x_keep, plane_keep = x, getattr(cfg, 'input_bw_plane', False)
            x = torch.from_numpy(data['X']).to(dev).contiguous()
            cfg.input_bw_plane = False
            ms, ms_max = time_of(a.encoder_gemm)
            variants['store_f32'] = {'dtype': 'default path on colour fp32 frames, bw_transform inside the step (the reference\'s own step)', 'ms_per_step': ms,
                                     'ms_per_step_max': ms_max, 'value': a.batch * a.frames / ms * 1e3, 'unit': 'frames/s'}
            x, cfg.input_bw_plane = x_keep, plane_keep
            with torch.no_grad():
                bucket.data.copy_(snap[0])
                for k, v in snap[1].items():
                    opt._flat[k].copy_(v)
                opt._seg_steps.copy_(snap[2])
        ms, ms_max = time_of(a.encoder_gemm, eager=True)
        variants['eager'] = {'dtype': 'default path, every launch enqueued by the host (no graph replay)', 'ms_per_step': ms, 'ms_per_step_max': ms_max,
                             'value': a.batch * a.frames / ms * 1e3, 'unit': 'frames/s'}
        with torch.no_grad():
            bucket.data.copy_(snap[0])
            for k, v in snap[1].items():
                opt._flat[k].copy_(v)
            opt._seg_steps.copy_(snap[2])
        if a.workload == 'billiards':
            # BASELINE.json configs[3] (six objects: the O(N^2) stress) through the same replayed step, its own model
            try:
                cfg6 = build_config('multibilliards', dev)
                cfg6.encoder_gemm = a.encoder_gemm
                torch.manual_seed(0)
                m6 = Stove(cfg6).to(dev)
                b6 = ParamArena(m6, 1)
                o6 = FlatAdam(b6, lr=cfg6.learning_rate, amsgrad=cfg6.debug_amsgrad)
                from stove_amd.utils.utils import bw_transform as _bwt
                x6 = _bwt(torch.from_numpy(make_batch('multibilliards', a.batch, a.frames, 0)['X']).to(dev).contiguous())
                cfg6.input_bw_plane = True
                g6 = GraphedTrainStep(m6, b6, o6, 1.0, alias_inputs=True)
                for i in range(4):
                    g6(x6)
                torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
                ev[0].record()
                for i in range(a.steps):
                    last6 = g6(x6)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps))
                variants['multibilliards'] = {'dtype': 'BASELINE.json configs[3]: 6-object billiards, greedy matcher, overlap_beta 100, max_obj_scale 0.22, same step',
                                              'ms_per_step': ts[len(ts) // 2], 'ms_per_step_max': ts[-1], 'value': a.batch * a.frames / ts[len(ts) // 2] * 1e3,
                                              'unit': 'frames/s', 'elbo_last_step': float(last6)}
                del g6, m6, b6, o6, x6
            except Exception as exc:          # a side measurement must not take the headline line down
                variants['multibilliards'] = {'error': repr(exc)}
        log('variants done')
    parity = None
    if rank == 0 and not os.environ.get('STOVE_BENCH_NO_PARITY'):
        try:
            parity = reference_parity(dev, a.encoder_gemm)
        except Exception as exc:
            parity = {'error': repr(exc)}
        log('reference parity done')
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.res == 32:
        cpu = cpu_baseline(a.workload, a.frames, a.cpu_batch, a.cpu_iters, full_batch=a.batch)
        log('cpu baseline done')

    if rank == 0:
        frames = a.batch * a.frames * world * a.steps
        out = {
            'metric': 'frames/s (fwd+bwd) for 3-obj billiards 32x32 T=100, 1/2/4/8 GPU; ELBO delta vs ref',
            'value': frames / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3,
            'ms_per_step_p50': per_step_ms[len(per_step_ms) // 2], 'ms_per_step_p99': per_step_ms[min(len(per_step_ms) - 1, int(0.99 * len(per_step_ms)))],
            'ms_per_step_min': per_step_ms[0], 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,          # the reference publishes no number for this metric (its section 1)
            'dtype': 'f32' + ({'bf16x3': ' (encoder GEMMs: fp32 as 3 bf16 MFMAs on hi/lo-split operands, fp32 accumulate)', 'fp32': '', 'bf16': ' + bf16 encoder operands'}[a.encoder_gemm]),
            'data': 'synthetic',
            'config': {'workload': f'{a.workload} {cfg.num_obj}-object {a.res}x{a.res} T={a.frames} batch={a.batch}/GPU' + (
                           ' (BASELINE.json configs[1])' if a.workload == 'billiards' and a.batch == 256 and a.frames == 100 and a.res == 32 else '') + (
                           '' if a.res == 32 else ' (general-size likelihood path: not a BASELINE.json configuration)'),
                       'objects': cfg.num_obj, 'global_batch': a.batch * world,
                       'step': 'forward+backward+allreduce+clip+adam(amsgrad)', 'step_mode': a.step_mode + (
                           ' (captured hipGraph replay, as Trainer.train runs its non-logging steps)' if a.step_mode == 'graph' else ''),
                       'frame_store': fs + {'bw32': ' (bw plane fp32, made once at upload: the same model input as colour frames + bw_transform)',
                                            'f32': ' (colour fp32; the step starts with bw_transform)', 'u8': ' (8-bit colour, converted by the first kernel)'}[fs],
                       'parallelism': f'dp{world}',
                       'elbo_last_step': elbo_val,
                       'host_gc': 'collector enabled; long-lived objects frozen after warm-up (gc.collect + gc.freeze, as train.py does)'},
            'roofline': roofline, 'cpu_baseline': cpu, 'variants': variants,
            'elbo_rel_vs_reference': parity.get('elbo_rel_vs_reference') if parity else None, 'reference_parity': parity, 'comm': comm,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
