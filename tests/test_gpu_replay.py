"""The step the headline times (three captured hipGraphs replayed, stove_amd/graphed.py) against the step the goldens pin
(the same launches enqueued eagerly): BIT FOR BIT -- ELBO, the whole gradient arena and every parameter after EACH step, with
a different batch per step.  Same kernels, same launch geometry, fixed summation orders, no atomics: any difference is a bug
(an ordering hole between g_main and g_side, a scratch buffer read before it is rewritten, a kernel argument frozen at capture).

`poison`: between two replays every byte of the captured step's private memory pool -- all forward activations, the saved
streams of the recursion, the table-gradient scratch, the gate gradients, every workspace -- is overwritten with NaN bit
patterns (stove_fill_words over the pool's segments, torch.cuda.memory_snapshot), so that a replay which consumes anything the
PREVIOUS replay left behind (a stale read on the side chain) cannot reproduce the eager numbers.

Reference: the step is train.py:443-473 (forward, backward, clip, Adam); the replayed form has no reference counterpart."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

DEV = torch.device('cuda:0')

WORKLOADS = {
    'billiards': dict(num_obj=3),
    'avoidance': dict(num_obj=3, action_conditioned=True, action_space=9, debug_core_appearance=True),
    'multibilliards': dict(num_obj=6, debug_match_objects='greedy', overlap_beta=100.0, max_obj_scale=0.22),
}


def _cfg(workload):
    from stove_amd.video_prediction.config import StoveConfig
    cfg = StoveConfig()
    cfg.width, cfg.height, cfg.random_seed = 32, 32, 42
    cfg.device, cfg.dtype = DEV, torch.float32
    cfg.action_conditioned, cfg.action_space = False, None
    cfg.print_every, cfg.plot_every = 10 ** 9, 1e19
    for k, v in WORKLOADS[workload].items():
        setattr(cfg, k, v)
    return cfg


def _batches(workload, n_seq, T, n_steps):
    """a different batch per step: other sequences AND another window of them"""
    from stove_amd.envs import envs
    d = envs.synth_sequences(workload, n_seq * 2, T + n_steps, seed0=5)
    out = []
    for s in range(n_steps):
        rows = slice((s % 2) * n_seq, (s % 2) * n_seq + n_seq)
        x = torch.from_numpy(d['X'][rows, s:s + T]).to(DEV).contiguous()
        a = r = None
        if 'action' in d:
            a = torch.from_numpy(d['action'][rows, s:s + T]).float().to(DEV).contiguous()
            r = (torch.from_numpy(d['reward'][rows, s + 2:s + T]).float().to(DEV) + 1.0).contiguous()      # to [0, 1] for the BCE loss (load_data.py)
        out.append((x, a, r))
    return out


def poison_pool(graph, lib, stream):
    """NaN bit patterns over every segment of the graph's private pool.  Returns the number of bytes written."""
    pool = tuple(graph.pool())
    total = 0
    for seg in torch.cuda.memory_snapshot():
        if tuple(seg.get('segment_pool_id', (0, 0))) != pool:
            continue
        rc = lib.stove_fill_words(seg['address'], 0x7FC00000, seg['total_size'] // 4, stream)
        assert rc == 0
        total += seg['total_size']
    return total


def _run(workload, batches, graphed, poison=False, device_rng=False):
    from stove_amd import _lib
    from stove_amd.arena import ParamArena
    from stove_amd.graphed import GraphedTrainStep
    from stove_amd.optim import FlatAdam
    from stove_amd.video_prediction.stove import Stove
    cfg = _cfg(workload)
    torch.manual_seed(0)
    model = Stove(cfg).to(DEV)
    if not device_rng:
        table = {}

        def noise(kind, shape):
            key = (kind, tuple(shape))
            if key not in table:
                table[key] = torch.randn(shape, generator=torch.Generator().manual_seed(len(table) + 5)).to(DEV)
            return table[key]
        model.noise_fn = noise
    arena = ParamArena(model, 1)
    opt = FlatAdam(arena, lr=cfg.learning_rate, amsgrad=True)
    rl = torch.nn.BCELoss() if cfg.action_conditioned else None
    step = GraphedTrainStep(model, arena, opt, clip=1.0, reward_loss=rl)
    torch.manual_seed(77)                      # the device generator the draws come from when no noise is injected
    lib = _lib.load()
    out, poisoned = [], 0
    for i, (x, a, r) in enumerate(batches):
        w = 15000.0 * min(1.0, (i + 1) / 20000.0) if rl is not None else 0.0
        e = step(x, a, r, reward_weight=w) if graphed else step.eager(x, a, r, reward_weight=w)
        torch.cuda.synchronize()
        out.append((e.clone(), arena.grad.clone(), arena.data.clone(),
                    step.reward_value.clone() if step.reward_value is not None else None))
        if graphed and poison:
            poisoned += poison_pool(step.graphs[0], lib, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
    if graphed:
        assert step.graphs is not None and len(step.graphs) == 2 and step._side_exec      # main / side / optimiser graphs
    return out, poisoned


@pytest.mark.parametrize('workload', list(WORKLOADS))
@pytest.mark.parametrize('poison', [False, True])
def test_replayed_step_equals_eager_step_bitwise(workload, poison):
    steps = 5
    batches = _batches(workload, 6, 9, steps)
    assert not torch.equal(batches[0][0], batches[1][0]) and not torch.equal(batches[1][0], batches[2][0])
    ref, _ = _run(workload, batches, graphed=False)
    got, poisoned = _run(workload, batches, graphed=True, poison=poison)
    if poison:
        assert poisoned > (1 << 20)           # the pool was found and overwritten
    for i, ((e0, g0, p0, r0), (e1, g1, p1, r1)) in enumerate(zip(ref, got)):
        assert torch.isfinite(e0) and float(g0.abs().max()) > 0
        assert torch.equal(e0, e1), (workload, 'elbo', i, float(e0), float(e1))
        assert torch.equal(g0, g1), (workload, 'gradient arena', i, float((g0 - g1).abs().max()))
        assert torch.equal(p0, p1), (workload, 'parameters', i, float((p0 - p1).abs().max()))
        if r0 is not None:
            assert torch.equal(r0, r1), (workload, 'reward loss', i)
    assert not torch.equal(ref[0][2], ref[-1][2])         # it trained


def test_replayed_step_equals_eager_step_bitwise_with_the_forward_chain_in_row_chunks():
    """The same at a batch big enough for the recognition network's forward to run as two row chunks on two streams
    (ops._encoder_lstm_fwd_chunked: 8 192 frames per step), graph pools poisoned between the replays: a chunk stream that was
    not ordered against the main one -- in the capture or in the eager step -- would read stale or poisoned rows."""
    from stove_amd import ops
    n_seq, T = 128, 64
    assert ops._enc_chunks(n_seq * T, 1024, 256) is not None
    batches = _batches('billiards', n_seq, T, 3)
    ref, _ = _run('billiards', batches, graphed=False)
    got, poisoned = _run('billiards', batches, graphed=True, poison=True)
    assert poisoned > (1 << 20)
    for i, ((e0, g0, p0, _), (e1, g1, p1, _)) in enumerate(zip(ref, got)):
        assert torch.isfinite(e0) and float(g0.abs().max()) > 0
        assert torch.equal(e0, e1), ('elbo', i, float(e0), float(e1))
        assert torch.equal(g0, g1), ('gradient arena', i, float((g0 - g1).abs().max()))
        assert torch.equal(p0, p1), ('parameters', i, float((p0 - p1).abs().max()))


def test_replayed_step_equals_eager_step_bitwise_device_rng():
    """The same with the draws coming from the device generator inside the capture (torch's graph-safe Philox offsets): the
    replay consumes the generator exactly as the eager step does."""
    batches = _batches('billiards', 6, 9, 4)
    ref, _ = _run('billiards', batches, graphed=False, device_rng=True)
    got, _ = _run('billiards', batches, graphed=True, device_rng=True)
    for i, ((e0, g0, p0, _), (e1, g1, p1, _)) in enumerate(zip(ref, got)):
        assert torch.equal(e0, e1), ('elbo', i, float(e0), float(e1))
        assert torch.equal(g0, g1) and torch.equal(p0, p1), i
    assert not torch.equal(ref[0][0], ref[1][0])


def test_side_graph_starts_after_its_producer():
    """g_side is ordered behind g_main only by event nodes across two separately launched graphs: the poisoned replay above
    proves the data, this one the runtime behaviour the design relies on -- an event-record node is enqueued at hipGraphLaunch
    time, so the wait of the side graph cannot be satisfied by the PREVIOUS replay's record.  A main graph whose producer kernel
    is long (a fill of 256 MB) followed by a side graph that copies the buffer must see the new value every replay."""
    from stove_amd import _lib
    lib = _lib.load()
    import ctypes
    n = 64 << 20
    buf = torch.zeros(n, dtype=torch.float32, device=DEV)
    out = torch.zeros(n, dtype=torch.float32, device=DEV)
    val = torch.zeros(1, dtype=torch.float32, device=DEV)
    main, side = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    events = lib.stove_event_list_begin()
    g = torch.cuda.CUDAGraph()
    side_graph, nodes = ctypes.c_void_p(), ctypes.c_int()
    with torch.cuda.graph(g, stream=main):
        _lib.check(lib.stove_capture_begin(side.cuda_stream), 'begin')
        buf.copy_(val.expand(n))                                                  # producer, on the main capture
        _lib.check(lib.stove_stream_after(side.cuda_stream, main.cuda_stream), 'after')
        prev = _lib.force_stream(side.cuda_stream)
        try:                                                                      # consumer, on the side capture
            _lib.check(lib.stove_sum_chunks(buf.data_ptr(), out.data_ptr(), n, 1, _lib.stream()), 'sum_chunks')
        finally:
            _lib.force_stream(prev)
        _lib.check(lib.stove_capture_end(side.cuda_stream, ctypes.byref(side_graph), ctypes.byref(nodes)), 'end')
    assert lib.stove_event_list_end(events) == 1
    ex = ctypes.c_void_p()
    _lib.check(lib.stove_graph_instantiate(side_graph, ctypes.byref(ex)), 'instantiate')
    try:
        for k in range(1, 6):
            val.fill_(float(k))
            g.replay()
            _lib.check(lib.stove_graph_launch(ex.value, side.cuda_stream), 'launch')
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            assert float(out.min()) == float(k) and float(out.max()) == float(k), (k, float(out.min()), float(out.max()))
    finally:
        torch.cuda.synchronize()
        del g
        lib.stove_graph_destroy(ex.value)
        assert lib.stove_event_list_destroy(events) == 0


# ---------------------------------------------------------------------------------------------------------------------
# data parallel: two gloo ranks (sharing cuda:0 on the 1-GPU box), a different batch per rank and per step; the replayed
# step (g_main / g_side, all-reduce, g_opt) against the eager data-parallel step of the same ranks, bit for bit
# ---------------------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from stove_amd import _lib
    from stove_amd.arena import ParamArena
    from stove_amd.graphed import GraphedTrainStep
    from stove_amd.optim import FlatAdam
    from stove_amd.video_prediction.stove import Stove
    steps = 4
    batches = _batches('billiards', 4, 8, steps * world)[rank::world]            # disjoint batches per rank
    res = {}
    for graphed in (False, True):
        cfg = _cfg('billiards')
        torch.manual_seed(0)
        model = Stove(cfg).to(DEV)
        table = {}

        def noise(kind, shape, table=table):
            key = (kind, tuple(shape))
            if key not in table:
                table[key] = torch.randn(shape, generator=torch.Generator().manual_seed(len(table) + 5 + 100 * rank)).to(DEV)
            return table[key]
        model.noise_fn = noise
        arena = ParamArena(model, world)
        arena.sync(0)
        opt = FlatAdam(arena, lr=cfg.learning_rate, amsgrad=True)
        step = GraphedTrainStep(model, arena, opt, clip=1.0, world_size=world)
        rows = []
        for (x, a, r) in batches:
            e = step(x) if graphed else step.eager(x)
            torch.cuda.synchronize()
            rows.append((e.clone().cpu(), arena.grad.clone().cpu(), arena.data.clone().cpu()))
            if graphed:
                poison_pool(step.graphs[0], _lib.load(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
        res[graphed] = rows
        if graphed:
            assert len(step.graphs) == 2
    torch.save(res, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_replay_equals_eager_bitwise(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % k)) for k in range(2)]
    for k in range(2):
        for i, ((e0, g0, p0), (e1, g1, p1)) in enumerate(zip(r[k][False], r[k][True])):
            assert torch.equal(e0, e1), (k, i, float(e0), float(e1))
            assert torch.equal(g0, g1) and torch.equal(p0, p1), (k, i)
    for i in range(len(r[0][True])):                       # replicas: same reduced gradient, same parameters, different ELBOs
        assert torch.equal(r[0][True][i][1], r[1][True][i][1]) and torch.equal(r[0][True][i][2], r[1][True][i][2])
        assert not torch.equal(r[0][True][i][0], r[1][True][i][0])


def test_set_overlap_switch_after_import_changes_the_schedule_not_the_numbers():
    """settings.set_overlap: the one measurement switch can be flipped in a process that has already imported the package
    -- the Python-side flag the ops read and the library's own, together.  One stream or five: the same ELBO bit for bit (the forward
    runs the same kernels on the same operands) and the same gradients up to the summation order of the table-gradient partials
    (held-back vs immediate placement use two kernels)."""
    import torch
    from stove_amd import settings
    from stove_amd.arena import ParamArena
    from stove_amd.envs import envs
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.stove import Stove
    dev = torch.device('cuda:0')
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height, cfg.random_seed = 3, 32, 32, 42
    cfg.device, cfg.dtype, cfg.action_conditioned, cfg.action_space = dev, torch.float32, False, None
    torch.manual_seed(0)
    model = Stove(cfg).to(dev)
    arena = ParamArena(model, 1)
    B, T = 192, 100                                    # 19 200 encoder rows: the chunked forward chain and the held-back table gradients are on
    x = torch.from_numpy(envs.synth_sequences('billiards', 24, T, seed0=3)['X']).repeat(8, 1, 1, 1, 1).to(dev).contiguous()
    g = torch.Generator().manual_seed(9)
    noise = {'latent': torch.randn(B, 3, 12, generator=g).to(dev), 'std': torch.randn(B, 3, 12, generator=g).to(dev),
             'steps': torch.randn(B, T - 2, 3, 18, generator=g).to(dev)}
    model.noise_fn = lambda kind, shape: noise[kind].reshape(shape)
    res = {}
    prev = settings.OVERLAP
    try:
        for on in (True, False, True):
            assert settings.set_overlap(on) in (True, False) and settings.OVERLAP is on
            arena.zero_grad()
            elbo, _, _ = model(x, 1, None)
            (-elbo).backward()
            torch.cuda.synchronize()
            res.setdefault(on, []).append((float(elbo), arena.grad.clone()))
    finally:
        settings.set_overlap(prev)
    (e1, g1), (e3, g3) = res[True]
    (e2, g2), = res[False]
    assert e1 == e3 and torch.equal(g1, g3)                           # back on: the same schedule, the same bits
    assert e1 == e2
    assert float((g1 - g2).abs().max()) <= 2e-6 * float(g1.abs().max())
