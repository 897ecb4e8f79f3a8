#!/usr/bin/env python
"""Headline benchmark: frames/s of one STOVE training step on synthetic billiards video.

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a rendezvous in the environment: starts N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: Stove.forward (bw_transform, encoder, matching, fused inference
recursion, both fused scene likelihoods) + elbo.backward() + [N>1: one RCCL all-reduce of the flat gradient] +
clip_grad_norm_(1) + Adam(amsgrad) step, i.e. the reference's training step (train.py:443-473), run the way Trainer.train
runs its non-logging steps: replayed as captured hipGraphs (stove_amd/graphed.py).
Workload (BASELINE.json configs[1]): 3-object billiards, 32x32, T=100, batch 256 per GPU (weak scaling), frames from the
build's numpy simulator, model with default initialisation; the colour fp32 frames -- what the reference's loader hands to
Stove.forward -- are resident in HBM before the timed region and bw_transform runs inside every step.
Rank 0 prints ONE JSON line (contract in the task statement): `roofline` is measured with HIP events around the dominant
kernel, `cpu_baseline` times the CPU oracle on a bounded sample, `variants` holds the other BASELINE.json configurations
(gravity, avoidance, multibilliards per-GPU shards; the cfg-1 evaluation protocol; Stove.rollout alone), each with its own
roofline block and its ELBO difference against the reference's number on the matching golden fixture.
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

# BASELINE.json configurations through the same step: model switches (reference run_files/experiments.sh:72-80, run_models.sh:17-25)
# and the golden fixture (reference Stove.forward, fp64, injected noise) the ELBO of the bench's model path is compared with
WORKLOADS = {
    'billiards': dict(cfg={}, golden='g7_stove_n3', label='BASELINE.json configs[1]'),
    'gravity': dict(cfg={}, golden='g7_stove_grav3', label='BASELINE.json configs[2], one GPU\'s shard (256 of the 2048 sequences)'),
    'multibilliards': dict(cfg=dict(num_obj=6, debug_match_objects='greedy', overlap_beta=100.0, max_obj_scale=0.22), golden='g7_stove_n6',
                           label='BASELINE.json configs[3]: 6-object billiards, greedy matcher, overlap_beta 100, max_obj_scale 0.22'),
    'avoidance': dict(cfg=dict(action_conditioned=True, action_space=9, debug_core_appearance=True), golden='g7_stove_ac3',
                      label='BASELINE.json configs[4], one GPU\'s shard (256 of the 1024 sequences): action-conditioned, appearance features'),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--batch', type=int, default=256, help='sequences per GPU')
    p.add_argument('--frames', type=int, default=100, help='T, frames per sequence')
    p.add_argument('--workload', default='billiards', choices=list(WORKLOADS))
    p.add_argument('--res', type=int, default=32, help='frame side; 32 = BASELINE.json (fused scene pipeline), anything else runs the general-size '
                   'likelihood path (the reference\'s stock gravity / multibilliards data are 50 x 50): a side measurement, never the headline')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-batch', type=int, default=32, help='sequences of the CPU baseline when the run holds fewer than --cpu-full-batch')
    p.add_argument('--cpu-full-batch', type=int, default=128, help='sequences of the CPU baseline\'s reported sample')
    p.add_argument('--profile-steps', type=int, default=3)
    p.add_argument('--encoder-gemm', default='bf16x3', choices=['bf16x3', 'fp32', 'bf16'],
                   help='recognition-network GEMMs: bf16x3 = fp32 products as 3 sixteen-bit MFMAs on hi/lo-split operands (default), '
                        'fp32 = library GEMMs, bf16 = plain bf16 operands (reported variant, never the headline)')
    p.add_argument('--no-variants', action='store_true', help='skip every side measurement')
    p.add_argument('--frame-store', default='f32', choices=['f32', 'bw32', 'u8', 'auto'],
                   help="how the resident frames are kept (config.frame_store of the Trainer's DeviceClipLoader): f32 (default) = colour fp32 "
                        "frames, bw_transform inside every step -- the reference's own step; bw32 = the bw plane made once at upload (the "
                        "Trainer's 'auto' choice for models that only consume bw frames; reported under variants); u8 = 8-bit colour frames")
    p.add_argument('--enc-chunks', type=int, default=0, help='row chunks of the recognition network\'s forward chain (0 = the default, 2; 1 = the chain on one stream: a profiling aid)')
    p.add_argument('--tile-gather', action='store_true', help='A/B: the glimpse-tile kernel of the scene forward with one lane per glimpse (stove_set_tile_lds(0)) whatever the object count')
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
    for k, v in WORKLOADS[workload]['cfg'].items():
        setattr(c, k, v)
    return c


def _cache_path(workload, n_seq, T, seed0, res):
    from stove_amd.envs import envs
    return os.path.join('/tmp', f'stove_bench_v{envs.SIMULATOR_VERSION}_{workload}_{n_seq}_{T}_{seed0}' + ('' if res == 32 else f'_r{res}') + '.npz')


def make_batch(workload, n_seq, T, seed0, res=32, workers=1):
    """Synthetic sequences of the build's numpy simulator (one environment per sequence, seed = seed0 + i; pinned to the reference's
    simulator by tests/golden/g0_envs.npz).  workers > 1 forks a pool: only before the process has touched the GPU."""
    from stove_amd.envs import envs
    cache = _cache_path(workload, n_seq, T, seed0, res)
    if os.path.exists(cache):
        return dict(np.load(cache))
    if workers > 1:
        d = envs.synth_sequences_parallel(workload, n_seq, T, seed0=seed0, res=None if res == 32 else res, workers=workers)
    else:
        d = envs.synth_sequences(workload, n_seq, T, seed0=seed0, res=None if res == 32 else res)
    try:
        np.savez(cache, **d)
    except OSError:
        pass
    return d


def _cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


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


# ---------------------------------------------------------------------------------------------- CPU baseline (the oracle, timed)
def _oracle_params(O, c, structs):
    torch.manual_seed(0)
    params = {}
    for k, shp in O.param_shapes(c, structs).items():
        scale = 0.1 if k.endswith(('means', 'sigma_params', 'params')) else 1.0 / max(1.0, float(shp[-1])) ** 0.5
        params[k] = (torch.randn(*shp) * scale).requires_grad_()
    return params


def cpu_baseline(workload, T, n_seq, data_small, data_full, full_batch):
    """Time the CPU oracle (oracle/stove_oracle.py: the reference's ATen op sequence restated) on the host cores, same workload
    shape, as a PROTOCOL with two named numbers on the SAME sample (B = full_batch sequences of the quoted batch, T frames, fp32,
    forward + backward):
      * reference_protocol: 8 threads -- the reference pins torch to config.max_threads = 8 (config.py:59, main.py:134);
      * best_of_sweep: the best of {8, 16, 32, 64, 128} threads, swept on that same batch size (8, 16, 32 always; beyond, the sweep stops at the first count
        30 % slower than the best so far: past the optimum the step only slows down).
    One warm-up iteration, one timed iteration per thread count (a step is 5-10 s), a second one for the two named counts
    (their value = the median of two).  `value` / `cores` = best_of_sweep.  Without the full batch (small --batch runs) the
    protocol runs on the B = n_seq sample instead and says so."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import stove_oracle as O
    avail = _cores()
    c = O.default_config(**WORKLOADS[workload]['cfg'])
    structs = O.build_structs(c)
    params = _oracle_params(O, c, structs)
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
    if data_full is not None and full_batch > n_seq:
        x, nb = torch.from_numpy(data_full['X'][:full_batch]), full_batch
    else:
        x, nb = torch.from_numpy(data_small['X']), n_seq
    counts = [t for t in (8, 16, 32, 64, 128) if t <= avail] or [min(8, avail)]
    t_start = time.perf_counter()
    torch.set_num_threads(counts[0])
    warm = one(x, nb)
    times = {}
    for threads in counts:
        torch.set_num_threads(threads)
        times[threads] = [one(x, nb)]
        best_so_far = min(v[0] for v in times.values())
        # bounds: past the optimum the step only gets slower with more threads (128 threads: 75 s for this step on a 256-core host) --
        # 8, 16 and 32 are always tried (host timings of one step scatter by 30 %), then the sweep stops at the first count 1.3 x off the
        # best; and the whole baseline stays under ~100 s of host time
        if (threads >= 32 and times[threads][0] > 1.3 * best_so_far) or time.perf_counter() - t_start > 60.0:
            break
    best = min(times, key=lambda k: times[k][0])
    for threads in sorted({counts[0], best}):
        torch.set_num_threads(threads)
        times[threads].append(one(x, nb))
    med = lambda v: sorted(v)[0] if len(v) == 1 else 0.5 * (sorted(v)[0] + sorted(v)[1])
    fps = lambda k: nb * T / med(times[k])
    best = max((counts[0], best), key=fps)          # of the two counts timed twice, by their medians (host timings of one step scatter)
    torch.set_num_threads(best)
    cores = best
    desc = f'{workload} B={nb} T={T} fp32 fwd+bwd' + (f' (a {nb}-sequence sample of the quoted batch of 256)' if nb < 256 else '')
    out = {'value': fps(best), 'unit': 'frames/s', 'cores': best, 'kind': 'port', 'value_batch': nb,
           'sample': f'{desc}; best of the thread sweep {sorted(times)} on this same batch, median of 2 iterations ({med(times[best]):.1f} s) after one warm-up',
           'reference_protocol': {'value': fps(counts[0]), 'cores': counts[0], 'seconds_per_step': round(med(times[counts[0]]), 2),
                                  'what': 'torch.set_num_threads(8) as the reference does (config.max_threads, main.py:134), median of 2'},
           'best_of_sweep': {'value': fps(best), 'cores': best, 'seconds_per_step': round(med(times[best]), 2)},
           'thread_sweep_frames_per_s': {str(k): round(nb * T / min(v), 1) for k, v in sorted(times.items())},
           'iterations_s': {'warmup': round(warm, 2), **{str(k): [round(t, 2) for t in v] for k, v in sorted(times.items())}},
           'cpu_model': _cpu_model(), 'cores_total': os.cpu_count(), 'cores_available': avail}
    try:        # what the port's wall time is next to the reference's own, measured where the reference can run (oracle/port_walltime.py)
        with open(os.path.join(ROOT, 'tests', 'golden', 'g18_port_walltime.json')) as f:
            pw = json.load(f)
        out['port_vs_reference_walltime'] = {'source': 'tests/golden/g18_port_walltime.json (oracle/port_walltime.py, build container, 8 threads)',
                                             'cpu': pw.get('cpu'), **{k: v['port_over_reference'] for k, v in pw['cases'].items()}}
    except (OSError, KeyError, ValueError):
        pass
    # Stove.rollout alone (reference stove.py:777-861; BASELINE.md quotes 3.7 ms per step for it on CPU): 92 generative steps, no grad
    try:
        zl = torch.rand(256, c.num_obj, 18) * 0.5
        with torch.no_grad():
            O.rollout(c, params, zl, 8)
            t0 = time.perf_counter()
            O.rollout(c, params, zl, 92)
            dt = time.perf_counter() - t0
        out['rollout'] = {'ms_per_step': dt / 92 * 1e3, 'value': 256 * 92 / dt, 'unit': 'frames/s', 'sample': f'O.rollout B=256, 92 steps, no grad, {cores} threads'}
    except Exception as exc:       # a side number
        out['rollout'] = {'error': repr(exc)}
    return out


# ---------------------------------------------------------------------------------------------- parity against the committed goldens
def _analytic_weights():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_golden_analytic_weights', os.path.join(ROOT, 'tests', 'golden', 'analytic_weights.py'))
    aw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(aw)
    return aw


def golden_model(dev, workload, encoder_gemm):
    """The bench's model path (flat arena) with the fixtures' analytic weights and the fixture's injected noise."""
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.stove import Stove
    aw = _analytic_weights()
    gold = dict(np.load(os.path.join(ROOT, 'tests', 'golden', WORKLOADS[workload]['golden'] + '_f64.npz')))
    cfg = build_config(workload, dev)
    cfg.encoder_gemm = encoder_gemm
    model = Stove(cfg)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(aw.analytic_tensor(name, p.shape, torch.float64).float())
    model = model.to(dev)
    arena = ParamArena(model, 1)                     # the flat arena path the timed step runs
    f = lambda a: torch.from_numpy(np.asarray(a)).float().to(dev)
    table = {'latent': f(gold['eps_lat'])[..., 0], 'std': f(gold['eps_std'])[..., 0], 'steps': f(gold['eps_steps']).permute(1, 0, 2, 3).contiguous()}
    model.noise_fn = lambda kind, shape: table[kind].reshape(shape)
    return model, arena, gold, f


def reference_parity(dev, workload, encoder_gemm):
    """ELBO of the bench's own model path on the reference-generated fixture of the workload (frames, injected noise and the
    reference's fp64 ELBO in tests/golden/g7_stove_*_f64.npz) -> |elbo - elbo_ref| / |elbo_ref|.  Reads the committed fixture only:
    neither the oracle nor the reference runs here."""
    model, arena, gold, f = golden_model(dev, workload, encoder_gemm)
    actions = f(gold['actions']) if 'actions' in gold else None
    elbo, _, _ = model(f(gold['x']), 1, actions)
    (-elbo).backward()
    e, ref = float(elbo.detach()), float(gold['elbo'])
    B, T = gold['x'].shape[:2]
    return {'fixture': f"tests/golden/{WORKLOADS[workload]['golden']}_f64.npz (reference Stove.forward, fp64, B={B} T={T}, injected noise)", 'elbo': e,
            'elbo_reference': ref, 'elbo_rel_vs_reference': abs(e - ref) / abs(ref), 'bar': 1e-4}


# ---------------------------------------------------------------------------------------------- one workload on the device
class Job:
    """Model + flat arena + FlatAdam + a resident batch of one workload, and its training step (eager or replayed)."""

    def __init__(self, workload, dev, data, encoder_gemm, frame_store, world=1, res=32, step_mode='graph'):
        from stove_amd.arena import ParamArena
        from stove_amd.optim import FlatAdam
        from stove_amd.video_prediction.stove import Stove
        self.workload, self.dev, self.world, self.step_mode = workload, dev, world, step_mode
        cfg = build_config(workload, dev, res)
        cfg.encoder_gemm = encoder_gemm
        torch.manual_seed(0)
        self.cfg = cfg
        self.model = Stove(cfg).to(dev)
        self.bucket = ParamArena(self.model, world)          # parameters / gradients flat; grad buffer == all-reduce bucket
        self.bucket.sync(0)                                  # replicas start from rank 0's parameters (one broadcast of the flat buffer)
        self.opt = FlatAdam(self.bucket, lr=cfg.learning_rate, amsgrad=cfg.debug_amsgrad)       # torch.optim.Adam's update as one launch
        self.x_color = torch.from_numpy(data['X']).to(dev).contiguous()       # a batch as the DataLoader collates it (contiguous n,T,C,w,h)
        self.batch, self.frames = self.x_color.shape[:2]
        self.actions = torch.from_numpy(data['action']).float().to(dev) if 'action' in data else None
        self.bw_only = not (cfg.debug_core_appearance or cfg.debug_match_appearance)
        self.minus_one = torch.tensor(-1.0, device=dev)
        self.graphed = None
        self.set_store(frame_store)

    def set_store(self, fs):
        """The resident frames as the Trainer's device frame store hands them over (load_data.DeviceClipLoader, config.frame_store)."""
        if fs == 'auto':
            fs = 'bw32' if self.bw_only else 'f32'
        if fs == 'bw32' and not self.bw_only:
            raise SystemExit('--frame-store bw32 needs a workload without appearance features')
        self.fs = fs
        self.cfg.input_bw_plane = False
        if fs == 'bw32':
            from stove_amd.utils.utils import bw_transform
            self.x = bw_transform(self.x_color)
            self.cfg.input_bw_plane = True
        elif fs == 'u8':
            self.x = torch.round(self.x_color * 255).to(torch.uint8)
        else:
            self.x = self.x_color
        self.graphed = None

    def eager_step(self, i):
        self.bucket.zero()
        elbo, _, rewards = self.model(self.x, i + 1, self.actions)
        elbo.backward(self.minus_one)                                 # d(-ELBO): the loss of train.py:452 without the neg / fill launches
        self.bucket.all_reduce()
        self.opt.step(max_norm=1.0)                                    # clip_grad_norm_(1) folded into the Adam launch
        return elbo

    def graph_step(self, i):
        # the step as the Trainer runs it between logging steps: captured once, replayed (two graphs around the all-reduce for N > 1)
        if self.graphed is None:
            from stove_amd.graphed import GraphedTrainStep
            self.graphed = GraphedTrainStep(self.model, self.bucket, self.opt, 1.0, world_size=self.world, alias_inputs=True)
        return self.graphed(self.x, self.actions)

    def step(self, i):
        return self.graph_step(i) if self.step_mode == 'graph' else self.eager_step(i)

    def snapshot(self):
        return (self.bucket.data.clone(), {k: v.clone() for k, v in self.opt._flat.items()}, self.opt._seg_steps.clone())

    def restore(self, snap):
        with torch.no_grad():       # timing steps train: put the parameters and the optimiser state back
            self.bucket.data.copy_(snap[0])
            for k, v in snap[1].items():
                self.opt._flat[k].copy_(v)
            self.opt._seg_steps.copy_(snap[2])

    def median_ms(self, steps, eager=False, warm=3):
        """median / max device time per step (event pairs): one stall of the host or the allocator must not decide a side measurement"""
        fn = self.eager_step if eager else self.step
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        for i in range(steps):
            last = fn(i)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
        return ts[len(ts) // 2], ts[-1], float(last.detach())

    def kernel_profile(self, n_steps, record=True, first_index=0, one_stream=False):
        """Per-kernel HIP-event timing of `n_steps` eager steps (event pairs around single launches on the stream each kernel is
        launched on: not through a captured graph).  one_stream: the same steps with the recognition network's forward chain on
        ONE stream (ops.ENC_CHUNKS = 1): every GEMM launch alone on the chip.  -> ({kernel: (total ms, launches, covered ms)}, ...)"""
        from stove_amd import _lib, ops as _ops
        lib = _lib.load()
        saved = _ops.ENC_CHUNKS
        if one_stream:
            _ops.ENC_CHUNKS = 1
            self.eager_step(first_index)          # the allocator meets the other schedule's sizes untimed
            torch.cuda.synchronize()
        if record:
            lib.stove_profile_enable(1)
        for i in range(n_steps):
            self.eager_step(first_index + 1 + i)
        torch.cuda.synchronize()
        _ops.ENC_CHUNKS = saved
        prof = None
        if record:
            prof = _lib.profile_report(wall=True)
            lib.stove_profile_enable(0)
        return prof


def gnn_flops(n_obj):
    """algorithmic flops of one GNN step per (sequence, step), SURVEY.md section 8d (off-diagonal pairs only)"""
    return 17408 * n_obj + 26944 * n_obj * (n_obj - 1)


def make_roofline(prof_w, prof_serial, n_obj, batch, frames, n_steps, encoder_gemm):
    """The `roofline` block from a kernel profile: the dominant kernel against the roofline that bounds it."""
    if not prof_w:
        return None
    prof = {k: v[:2] for k, v in prof_w.items()}
    name, (total_ms, count) = max(prof.items(), key=lambda kv: kv[1][0])
    avg_ms = total_ms / count
    per_step = {k: round(v[0] / n_steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:24]}
    F = gnn_flops(n_obj)
    if name.startswith('gemm_bf16'):
        # recognition-network GEMMs (csrc/gemm_bf16.hip), all launches of a step together (they differ in shape): algorithmic flops
        # = 2 M N K of the fp32 products -- x W_ih^T, (N-1) x h W_hh^T forward; (N-1) x dg W_hh, dg^T h over the N-1 recurrent steps,
        # dgx^T x backward (the head's 50-wide products run in enc_head_*_k) -- against the dense bf16 / f16 MFMA peak (2.5 PFLOP/s,
        # MI355X_MICROARCH.md); the split path issues 3 sixteen-bit MFMA flops per algorithmic flop.
        nfr, Hh, Dd = batch * frames, 256, 32 * 32
        flops = 2.0 * nfr * (2 * Dd * 4 * Hh + 3 * (n_obj - 1) * Hh * 4 * Hh)
        ms_step = total_ms / n_steps
        ach = flops / (ms_step * 1e-3) / 1e12
        passes = {'bf16x3': 3, 'bf16': 1}.get(encoder_gemm, 3)
        wall_step = prof_w[name][2] / n_steps
        roofline = {'bound': 'mfma', 'kernel': name, 'avg_ms': avg_ms, 'launches': count, 'ms_per_step': ms_step,
                    'achieved': ach, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': ach / 2500.0, 'traffic': None,
                    'mfma_pipe_frac': passes * ach / 2500.0,
                    'ms_per_step_covered': wall_step, 'frac_covered': flops / (wall_step * 1e-3) / 1e12 / 2500.0,
                    'note': 'achieved = algorithmic fp32 flops (2MNK of the %d GEMM launches of a step) / their SUMMED launch time; the kernel '
                            'issues %d sixteen-bit MFMA flops per algorithmic flop (hi/lo split), so the matrix pipe runs at mfma_pipe_frac.  The '
                            'forward chain runs as two row chunks on two streams: launches of the kernel overlap in time and each is timed '
                            'with the other beside it, so the sum exceeds the time the launches cover (ms_per_step_covered, the union of '
                            'their event spans; frac_covered = the same flops over that time)' % (count // n_steps, passes),
                    'kernels_ms_per_step': per_step}
        if prof_serial and name in prof_serial:
            ms1 = prof_serial[name][0] / n_steps
            roofline['one_stream'] = {'launches': prof_serial[name][1], 'ms_per_step': ms1, 'achieved': flops / (ms1 * 1e-3) / 1e12,
                                      'frac': flops / (ms1 * 1e-3) / 1e12 / 2500.0,
                                      'note': 'the same steps with the forward chain on one stream: no two launches of the kernel overlap; '
                                              'measured in this run, after the passes above'}
    elif name.startswith(('dyn_loop', 'gnn_step', 'rollout', 'gnn_dw')):
        # GNN recursion: dense fp32 contraction -> 157.3 TFLOP/s (v_mfma_f32_16x16x4_f32; the packed-fp32 VALU path of the
        # small-graph kernels, v_pk_fma_f32, has the same peak on MI355X).  The backward is 2F: F of data gradients
        # (dyn_loop_bwd_small_k) + F of weight gradients (gnn_dw_small_k); the MFMA kernel of gnn.hip (N > 6) does both in one launch.
        units = batch * (frames - 2)
        both = 'bwd' in name and 'small' not in name
        flops = (2 * F if both else F) * units
        ach = flops / (avg_ms * 1e-3) / 1e12
        roofline = {'bound': 'mfma', 'kernel': name, 'avg_ms': avg_ms, 'launches': count, 'achieved': ach,
                    'peak': 157.3, 'unit': 'TFLOP/s', 'frac': ach / 157.3, 'traffic': None,
                    'note': 'latency-bound: %d dependent time steps per launch, one sequence per CU' % (frames - 2),
                    'kernels_ms_per_step': per_step}
    else:
        # SPN / scene sweep: scan-shaped -> HBM 8 TB/s.  Algorithmic bytes per frame fwd+bwd = 8200 + 32 N;
        # one direction of it per launch.
        alg_bytes = (8200 + 32 * n_obj) / 2 * batch * (frames - 1)
        ach = alg_bytes / (avg_ms * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'kernel': name, 'avg_ms': avg_ms, 'launches': count, 'achieved': ach,
                    'peak': 8000.0, 'unit': 'GB/s', 'frac': ach / 8000.0, 'traffic': None,
                    'note': 'VALU-bound sweep (~120 flop/B, ridge ~20 flop/B)', 'kernels_ms_per_step': per_step}
    if not roofline['kernel'].startswith('dyn_loop'):
        # the T-serial recursion (latency-bound): forward and data-gradient kernels against the fp32 MFMA / VALU peak
        rec = {}
        for k in ('dyn_loop_fwd_small_k', 'dyn_loop_bwd_small_k', 'dyn_loop_fwd_k', 'dyn_loop_bwd_k'):
            if k in prof:
                fl = (2 * F if k == 'dyn_loop_bwd_k' else F) * batch * (frames - 2)
                ms = prof[k][0] / prof[k][1]
                rec[k] = {'avg_ms': ms, 'us_per_dependent_step': ms * 1e3 / (frames - 2), 'achieved': fl / (ms * 1e-3) / 1e12, 'peak': 157.3,
                          'unit': 'TFLOP/s', 'frac': fl / (ms * 1e-3) / 1e12 / 157.3}
        if rec:
            rec['note'] = 'latency-bound: %d dependent time steps per launch' % (frames - 2)
            roofline['recursion'] = rec
    # second kernel family of SURVEY.md section 8d: the SPN / scene sweep (all its launches of one step together) against
    # HBM with the algorithmic 8200 + 32 N bytes per frame forward + backward
    spn_ms = sum(v[0] for k, v in prof.items() if k.startswith(('objspn_', 'bgspn_', 'bg_', 'scene_', 'spn_bake', 'reduce_chunks'))) / n_steps
    if spn_ms > 0:
        alg = (8200 + 32 * n_obj) * batch * (frames - 1)
        ach = alg / (spn_ms * 1e-3) / 1e9
        roofline['spn_sweep'] = {'bound': 'hbm', 'ms_per_step': round(spn_ms, 4), 'achieved': ach, 'peak': 8000.0, 'unit': 'GB/s',
                                 'frac': ach / 8000.0, 'algorithmic_bytes': alg,
                                 'note': 'launch durations summed (the chains overlap on three streams, so this exceeds their wall time); '
                                         'fp32-VALU bound: ~120 flop/B against a ridge of ~20 flop/B'}
    return roofline


def attach_traffic(roofline, workload, batch, frames):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*pmc_traffic*.json of tools/profile_round.sh; the passes
    serialise kernels and cannot run inside a timed bench).  Only counters taken on THESE kernel sources and this workload are quoted."""
    import glob
    from stove_amd import build as _build
    suffix = '' if workload == 'billiards' else '_' + workload
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_pmc_traffic{suffix}.json')))
    if roofline is None or not paths or batch != 256 or frames != 100:
        return
    path = paths[-1]
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        return
    roofline['traffic_measured_in_run'] = False
    if doc.get('_source_hash') != _build.source_hash():
        roofline['traffic_source'] = os.path.basename(path) + ' (stale: profiled on other kernel sources, not quoted)'
        return
    tr = doc.get(roofline['kernel'])
    if tr:
        roofline['traffic'] = tr['hbm_bytes_per_launch']
        roofline['traffic_source'] = os.path.basename(path)
    steps_prof = doc.get('_steps_profiled') or (doc.get('flat_adam_k') or {}).get('launches_profiled')
    if steps_prof and 'spn_sweep' in roofline:
        fam = {k: v for k, v in doc.items() if isinstance(v, dict) and k.startswith(('objspn_', 'bgspn_', 'bg_', 'scene_', 'spn_bake', 'reduce_chunks'))}
        roofline['spn_sweep']['traffic'] = sum(v['hbm_bytes_per_launch'] * v['launches_profiled'] for v in fam.values()) / steps_prof
        roofline['spn_sweep']['traffic_note'] = 'HBM bytes per STEP of the family (counter passes, FETCH_SIZE x 2 + WRITE_SIZE)'


# ---------------------------------------------------------------------------------------------- the other BASELINE configurations
def workload_variant(dev, workload, data, a):
    """One of BASELINE.json's other configurations (a per-GPU shard of it) through the same replayed step, with its own model:
    ms / step, frames/s, roofline block from its own kernel profile, ELBO difference against the reference on its golden fixture."""
    job = Job(workload, dev, data, a.encoder_gemm, a.frame_store, 1)       # the headline's frame store (colour fp32 by default)
    job.step(0)                                   # capture
    ms, ms_max, last = job.median_ms(a.steps)
    out = {'workload': f'{workload} {job.cfg.num_obj}-object 32x32 T={job.frames} batch={job.batch}', 'what': WORKLOADS[workload]['label'], 'ms_per_step': ms,
           'ms_per_step_max': ms_max, 'value': job.batch * job.frames / ms * 1e3, 'unit': 'frames/s', 'frame_store': job.fs, 'elbo_last_step': last}
    prof = job.kernel_profile(2)
    out['roofline'] = make_roofline(prof, None, job.cfg.num_obj, job.batch, job.frames, 2, a.encoder_gemm)
    attach_traffic(out['roofline'], workload, job.batch, job.frames)
    del job
    par = reference_parity(dev, workload, a.encoder_gemm)
    out['elbo_rel_vs_reference'] = par['elbo_rel_vs_reference']
    out['reference_parity'] = par
    return out


def cfg1_eval_variant(dev, a):
    """BASELINE.json configs[0], SURVEY.md section 8d's cfg-1 protocol on the device: B = 4 sequences, the first 8 frames through
    Stove.forward + backward, then Stove.rollout for 92 steps from the last inferred state; 100 frames per sequence are counted.
    Inputs = the reference-generated fixture g7_stove_n3 (frames, injected noise), so the ELBO and the 92 rolled-out states are
    compared with the reference's numbers in the same breath."""
    model, arena, gold, f = golden_model(dev, 'billiards', a.encoder_gemm)
    x = f(gold['x'])
    minus_one = torch.tensor(-1.0, device=dev)
    n_roll = gold['roll_z'].shape[1]

    def once():
        arena.zero()
        elbo, prop, _ = model(x, 0, None)                # step 0: a logging step, prop_dict filled (the evaluation reads z from it)
        elbo.backward(minus_one)
        with torch.no_grad():
            zp, _ = model.rollout(prop['z'][:, -1], num=n_roll)
        return elbo, zp
    for _ in range(3):
        elbo, zp = once()
    torch.cuda.synchronize()
    reps = 10
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    host = []
    ev[0].record()
    for i in range(reps):
        t0 = time.perf_counter()
        elbo, zp = once()
        ev[i + 1].record()
        torch.cuda.synchronize()
        host.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    ms = ts[len(ts) // 2]
    B, T = x.shape[:2]
    e, ref = float(elbo.detach()), float(gold['elbo'])
    zr = torch.from_numpy(gold['roll_z']).to(dev)
    return {'what': 'BASELINE.json configs[0] on the device: B=4, T=8 forward + backward, then rollout(92); eager (launch-latency bound), synchronised per iteration',
            'ms_per_iteration': ms, 'ms_per_iteration_host': sorted(host)[len(host) // 2], 'value': B * (T + n_roll) / ms * 1e3, 'unit': 'frames/s',
            'frames_counted_per_sequence': T + n_roll, 'elbo_rel_vs_reference': abs(e - ref) / abs(ref),
            'rollout_z_err_vs_reference': float((zp.double() - zr).abs().max() / zr.abs().max()), 'fixture': 'tests/golden/g7_stove_n3_f64.npz'}


def rollout_variant(job, a):
    """Stove.rollout alone (reference stove.py:777-861, the path MCTS / PPO consume): B = 256 sequences x 92 generative steps, no grad,
    one persistent launch (rollout_fwd_small_k) from the state the model infers for the first 8 frames of the resident batch."""
    from stove_amd import _lib
    lib = _lib.load()
    model, dev = job.model, job.dev
    pe = job.cfg.print_every
    with torch.no_grad():
        job.cfg.print_every = 1
        try:
            _, prop, _ = model(job.x[:, :8].contiguous(), 0, job.actions[:, :8].contiguous() if job.actions is not None else None)
        finally:
            job.cfg.print_every = pe
        z_last = prop['z'][:, -1].contiguous()
        num = 92
        for _ in range(3):
            model.rollout(z_last, num=num)
        torch.cuda.synchronize()
        reps = 10
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            zp, _ = model.rollout(z_last, num=num)
            ev[i + 1].record()
        torch.cuda.synchronize()
        lib.stove_profile_enable(1)
        for i in range(3):
            model.rollout(z_last, num=num)
        torch.cuda.synchronize()
        prof = _lib.profile_report()
        lib.stove_profile_enable(0)
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    ms = ts[len(ts) // 2]
    B = z_last.shape[0]
    out = {'what': 'Stove.rollout B=%d x %d steps, no grad (reference stove.py:777-861)' % (B, num), 'ms': ms, 'us_per_step': ms * 1e3 / num,
           'value': B * num / ms * 1e3, 'unit': 'frames/s', 'finite': bool(torch.isfinite(zp).all())}
    k = max(prof.items(), key=lambda kv: kv[1][0]) if prof else None
    if k is not None:
        kms = k[1][0] / k[1][1]
        fl = gnn_flops(job.cfg.num_obj) * B * num
        out['roofline'] = {'bound': 'mfma', 'kernel': k[0], 'avg_ms': kms, 'us_per_dependent_step': kms * 1e3 / num, 'achieved': fl / (kms * 1e-3) / 1e12,
                           'peak': 157.3, 'unit': 'TFLOP/s', 'frac': fl / (kms * 1e-3) / 1e12 / 157.3, 'traffic': None,
                           'note': 'latency-bound: %d dependent steps per launch, one sequence per CU' % num}
    return out


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
    # stdout carries ONE line, the JSON record: everything else that writes to file descriptor 1 (RCCL prints a version banner from
    # C when a communicator is made) goes to stderr; the record is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    # ---- synthetic data, BEFORE the GPU is touched (the simulators run in a pool of forked workers)
    side = rank == 0 and world == 1 and not a.no_variants and a.res == 32
    workers = max(1, min(64, _cores() // max(1, world)))
    log('generating data (%d workers)' % workers)
    data = make_batch(a.workload, a.batch, a.frames, rank * a.batch, a.res, workers)
    side_data = {}
    if side and a.workload == 'billiards':
        for w in ('gravity', 'avoidance', 'multibilliards'):
            side_data[w] = make_batch(w, a.batch, a.frames, 0, 32, workers)
    cpu_small = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.res == 32:
        cpu_small = make_batch(a.workload, a.cpu_batch, a.frames, 10 ** 6, 32, workers)
    log('data ready')
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
    if a.tile_gather:
        _lib.load().stove_set_tile_lds(0)
    if a.enc_chunks > 0:
        from stove_amd import ops as _ops0
        _ops0.ENC_CHUNKS = a.enc_chunks
    job = Job(a.workload, dev, data, a.encoder_gemm, a.frame_store, world, a.res, a.step_mode)
    cfg, model, bucket = job.cfg, job.model, job.bucket
    fs = job.fs
    log('model built')
    torch.manual_seed(1234 + rank)
    step = job.step
    if a.step_mode == 'graph':
        step(0)                                   # capture (restores parameters / optimiser / generator: not a training step)
        torch.cuda.synchronize()
        log('step captured')
    if a.warmup > 0 and not os.environ.get('STOVE_BENCH_NO_GC_SETTLE'):
        # Host-side settling BEFORE the warm-up steps, not between them and the timed region: the collection takes tens of ms of host
        # time with the device idle, the chip drops its clocks, and the first timed steps then run slow.  The warm-up steps run
        # straight into the timed region.
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
    log('timed region done: %.3f ms/step' % (dt / a.steps * 1e3))
    if os.environ.get('STOVE_BENCH_SERIES'):
        log('device ms per step: ' + ' '.join('%.3f' % v for v in series))
        log('host ms per step:   ' + ' '.join('%.3f' % ((host_t[i + 1] - host_t[i]) * 1e3) for i in range(a.steps)))

    # ---- per-kernel HIP-event timing of extra steps (profiling hooks off during the timed region); every rank runs the extra
    # steps (they contain the collective), only rank 0 records events
    roofline = None
    if a.profile_steps > 0:
        from stove_amd import ops as _ops
        prof_w = job.kernel_profile(a.profile_steps, record=rank == 0, first_index=a.warmup + a.steps)
        prof_serial = None
        if _ops.ENC_CHUNKS > 1:
            ps = job.kernel_profile(a.profile_steps, record=rank == 0, first_index=a.warmup + a.steps + a.profile_steps + 1, one_stream=True)
            prof_serial = {k: v[:2] for k, v in ps.items()} if ps else None
        if rank == 0:
            roofline = make_roofline(prof_w, prof_serial, cfg.num_obj, a.batch, a.frames, a.profile_steps, a.encoder_gemm)
            attach_traffic(roofline, a.workload, a.batch, a.frames)
    log('kernel profile done')
    # ---- data parallel: what the collective costs by itself (HIP events around the all-reduce of the flat gradient, the stream
    # otherwise idle) and what RCCL says about the group, so that a multi-GPU line explains itself.  N = 1: a dry run of the same
    # call through a one-rank RCCL group (the path the 8-GPU run takes, minus the wire).
    comm = None
    try:
        comm = comm_probe(bucket, dev, world, rank)
    except Exception as exc:
        comm = {'error': repr(exc)}
    # ---- side measurements (N = 1 only).  Never the headline `value`.
    variants = None
    if side:
        variants = {}
        snap = job.snapshot()
        g = torch.Generator(device='cpu').manual_seed(99)
        o = cfg.num_obj
        fixed = {'latent': torch.randn(a.batch, o, 12, generator=g).to(dev), 'std': torch.randn(a.batch, o, 12, generator=g).to(dev),
                 'steps': torch.randn(a.batch, a.frames - 2, o, 18, generator=g).to(dev)}

        def elbo_of(mode):
            cfg.encoder_gemm = mode
            model.noise_fn = lambda kind, shape: fixed[kind].reshape(shape)
            with torch.no_grad():
                e, _, _ = model(job.x, 1, job.actions)
            model.noise_fn = None
            return float(e)

        def guarded(name, fn):
            try:
                variants[name] = fn()
            except Exception as exc:          # a side measurement must not take the headline line down
                variants[name] = {'error': repr(exc)}
            job.restore(snap)
            log('variant %s done' % name)

        # the recognition network's GEMMs on plain bf16 operands (BASELINE.json configs[1] says "bf16"; SURVEY section 7: reported, not
        # assumed) and on the fp32 library path, each with its ELBO difference against the fp32 library path on THIS batch under
        # identical noise
        e_ref = elbo_of('fp32')
        for mode, label in (('bf16', 'bf16-operands (encoder GEMMs), fp32 accumulate'), ('fp32', 'fp32 library GEMMs')):
            def gemm_variant(mode=mode, label=label):
                e = elbo_of(mode)
                job.graphed = None
                ms, ms_max, _ = job.median_ms(a.steps)
                return {'dtype': label, 'ms_per_step': ms, 'ms_per_step_max': ms_max, 'value': a.batch * a.frames / ms * 1e3, 'unit': 'frames/s',
                        'elbo': e, 'elbo_rel_delta_vs_fp32_library': abs(e - e_ref) / abs(e_ref)}
            guarded(mode, gemm_variant)
        cfg.encoder_gemm = a.encoder_gemm
        job.graphed = None
        e_def = elbo_of(a.encoder_gemm)
        variants['default_vs_fp32_library'] = {'elbo': e_def, 'elbo_rel_delta_vs_fp32_library': abs(e_def - e_ref) / abs(e_ref),
                                               'note': 'the headline path on this batch under the noise of the two variants above'}
        cfg.encoder_gemm = a.encoder_gemm
        if job.bw_only and fs != 'bw32':
            def bw_variant():
                job.set_store('bw32')
                ms, ms_max, _ = job.median_ms(a.steps)
                job.set_store(fs)
                return {'dtype': "default path on the bw plane kept by the loader (frame_store 'auto' of the Trainer: bw_transform once at upload, "
                                 'bit-identical model input)', 'ms_per_step': ms, 'ms_per_step_max': ms_max, 'value': a.batch * a.frames / ms * 1e3, 'unit': 'frames/s'}
            guarded('store_bw32', bw_variant)

        def eager_variant():
            ms, ms_max, _ = job.median_ms(a.steps, eager=True)
            return {'dtype': 'default path, every launch enqueued by the host (no graph replay)', 'ms_per_step': ms, 'ms_per_step_max': ms_max,
                    'value': a.batch * a.frames / ms * 1e3, 'unit': 'frames/s'}
        guarded('eager', eager_variant)
        def short_clip_variant():
            """the reference's DEFAULT training shape (config.py: batch 256, 8 visible frames per clip, train.py:431-473): the same step"""
            short = {k: (v[:, :8] if isinstance(v, np.ndarray) and v.ndim >= 2 and v.shape[1] == a.frames else v) for k, v in data.items()}
            j8 = Job(a.workload, dev, short, a.encoder_gemm, fs, 1)
            j8.step(0)
            ms, ms_max, _ = j8.median_ms(max(a.steps, 30))
            return {'what': 'the reference\'s default training shape: %d clips x 8 frames per step, same model and step (launch-latency bound: ~70 kernels in < 1 ms)' % j8.batch,
                    'ms_per_step': ms, 'ms_per_step_max': ms_max, 'value': j8.batch * 8 / ms * 1e3, 'unit': 'frames/s', 'steps_per_s': 1e3 / ms}
        guarded('default_training_shape', short_clip_variant)
        guarded('rollout', lambda: rollout_variant(job, a))
        guarded('cfg1_eval', lambda: cfg1_eval_variant(dev, a))
        for w in side_data:
            guarded(w, lambda w=w: workload_variant(dev, w, side_data[w], a))
        log('variants done')
    parity = None
    if rank == 0 and not os.environ.get('STOVE_BENCH_NO_PARITY'):
        try:
            parity = reference_parity(dev, a.workload, a.encoder_gemm)
        except Exception as exc:
            parity = {'error': repr(exc)}
        log('reference parity done')
    cpu = None
    if cpu_small is not None:
        cpu = cpu_baseline(a.workload, a.frames, a.cpu_batch, cpu_small, data if a.batch >= a.cpu_full_batch else None, a.cpu_full_batch)
        if variants and 'rollout' in variants and 'value' in variants['rollout'] and 'value' in cpu.get('rollout', {}):
            variants['rollout']['cpu_baseline'] = cpu['rollout']
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
            'dtype': 'f32' + ({'bf16x3': ' (encoder GEMMs: fp32 products as 3 sixteen-bit MFMAs on hi/lo-split operands -- half pieces forward, bf16 pieces backward -- fp32 accumulate)',
                               'fp32': '', 'bf16': ' + bf16 encoder operands'}[a.encoder_gemm]),
            'data': 'synthetic',
            'config': {'workload': f'{a.workload} {cfg.num_obj}-object {a.res}x{a.res} T={a.frames} batch={a.batch}/GPU' + (
                           ' (BASELINE.json configs[1])' if a.workload == 'billiards' and a.batch == 256 and a.frames == 100 and a.res == 32 else '') + (
                           '' if a.res == 32 else ' (general-size likelihood path: not a BASELINE.json configuration)'),
                       'objects': cfg.num_obj, 'global_batch': a.batch * world,
                       'step': 'bw_transform+forward+backward+allreduce+clip+adam(amsgrad)' if fs != 'bw32' else 'forward+backward+allreduce+clip+adam(amsgrad)',
                       'step_mode': a.step_mode + (
                           ' (captured hipGraph replay, as Trainer.train runs its non-logging steps)' if a.step_mode == 'graph' else ''),
                       'frame_store': fs + {'bw32': ' (bw plane fp32, made once at upload: the same model input as colour frames + bw_transform)',
                                            'f32': ' (colour fp32 as the reference\'s loader hands them over; the step starts with bw_transform)',
                                            'u8': ' (8-bit colour, converted by the first kernel)'}[fs],
                       'parallelism': f'dp{world}',
                       'timing_ms_per_step': {'mean': dt / a.steps * 1e3, 'p50': per_step_ms[len(per_step_ms) // 2], 'min': per_step_ms[0],
                                              'p99': per_step_ms[min(len(per_step_ms) - 1, int(0.99 * len(per_step_ms)))],
                                              'what': 'event-timed per step inside the timed region; mean = the headline'},
                       'elbo_last_step': elbo_val,
                       'host_gc': 'collector enabled; long-lived objects frozen after warm-up (gc.collect + gc.freeze, as train.py does)'},
            'roofline': roofline, 'cpu_baseline': cpu, 'variants': variants,
            'all_reduce_in_graph': bool(getattr(job.graphed, 'reduce_captured', False)) if world > 1 else None,
            'elbo_rel_vs_reference': parity.get('elbo_rel_vs_reference') if parity else None, 'reference_parity': parity, 'comm': comm,
        }
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist.is_initialized():
        dist.destroy_process_group()


def comm_probe(bucket, dev, world, rank):
    """Event-timed all-reduce of the flat gradient bucket through RCCL.  world > 1: the job's own group.  world == 1: a one-rank
    'nccl' group made for the purpose (a dry run of the call the data-parallel step makes; nothing crosses a wire)."""
    import socket
    made = False
    if world == 1:
        if os.environ.get('STOVE_BENCH_NO_COMM_DRYRUN'):
            return None
        if not dist.is_initialized():
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                port = sk.getsockname()[1]
            dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
            made = True
    g = bucket.grad
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 20

    def one():
        dist.all_reduce(g)
        if world > 1:
            g.mul_(1.0 / world)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev[0].record()
    for _ in range(reps):
        one()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    if world > 1:
        t = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    out = {'all_reduce_ms': ms, 'bytes': int(g.numel() * 4), 'world_size_reported': dist.get_world_size(), 'backend': dist.get_backend(),
           'devices_visible': torch.cuda.device_count(),
           'note': ('one all-reduce (sum) of the flat fp32 gradient + the 1/world scale per step, between the backward graphs and the optimiser graph; '
                    'max over ranks of the mean of %d back-to-back calls' % reps) if world > 1 else
                   'dry run: the same collective call through a one-rank RCCL group (launch + kernel cost of the call, nothing on the wire)'}
    if made:
        bucket.grad.zero_()
        dist.destroy_process_group()
    return out


if __name__ == '__main__':
    main()
