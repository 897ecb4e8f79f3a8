"""The training step as captured hipGraphs (torch.cuda.CUDAGraph on ROCm).

A step is ~100 launches on three streams.  Enqueued one by one the host is part of the loop: at the reference's default
training shape (256 clips x 8 visible frames, config.py:17-21) it needs 2 ms for 1 ms of device work, and at the
100-frame shape any host hiccup (scheduler, allocator, collector) longer than the ~1.4 ms it runs ahead starves the device.
Captured once and replayed, a step is three launch calls and the host runs many steps ahead.

  g_main  [ zero_grad + Stove.forward + backward ]          captured on the training stream, short fork / join episodes included
  g_side  [ the parameter-gradient chain of the backward ]  its own capture of the side stream, launched on the side stream
  (join the side stream; data parallel: all-reduce of the flat gradient)
  g_opt   [ clip + Adam ]
g_main and g_side are ordered against each other by event nodes (csrc/common.h: stream_after across two captures).  As branches
of ONE graph the runtime queued the side chain behind the main chain: 3.6 instead of 3.1 ms per step.

What varies from step to step enters through device memory:
  * the batch: static input tensors (`alias_inputs=True` adopts the caller's tensors instead of copying into own ones: a
    caller that refills the same buffers, like bench.py or a loader gathering into `static_images()`, pays no copy);
  * the step-dependent scalars [lr, beta1, beta2, eps, max_norm, reward weight]: a ring of pinned host slots; slot (step mod
    SLOTS) is written by the host and copied to the device buffer the kernels read by an asynchronous copy enqueued in front
    of the replay, so the value travels in stream order and the host never waits for the device unless it is SLOTS steps ahead;
  * the random draws: torch's graph-safe Philox offsets;
  * Adam's step counts: on the device (`FlatAdam._seg_steps`).

Steps that need more than the loss (logging steps read prop_dict) go through the eager path of `Trainer.train_step`.
"""
import os

import torch
import torch.distributed as dist

SLOTS = 32          # how many steps the host may run ahead of the device
NHYPER = 8          # floats per slot: lr, beta1, beta2, eps, max_norm, reward weight, (2 spare)


def _cuda_backend():
    """The torch.distributed backend that serves CUDA (HIP) tensors in the default group: 'nccl' (= RCCL), 'gloo', ...  A group
    made with the default or a composite backend string reports 'undefined' / 'cpu:gloo,cuda:nccl' from get_backend(); the
    per-device map is what decides whether the gradient all-reduce can be captured."""
    import torch.distributed as dist
    try:
        cfg = str(dist.get_backend_config())
    except Exception:
        cfg = str(dist.get_backend())
    if ':' in cfg:
        table = dict(part.split(':', 1) for part in cfg.split(',') if ':' in part)
        return table.get('cuda')
    return cfg


class GraphedTrainStep:
    def __init__(self, stove, arena, optimizer, clip, supair_only=False, warmup=3, world_size=1, reward_loss=None,
                 alias_inputs=False, force_reduce=False, capture_reduce=True):
        """reward_loss: callable(pred, target) -> scalar for action-conditioned models (train.py:452-465); the loss is then
        -ELBO + w * reward_loss(rewards, targets) with the host-computed weight w (factor x ramp) passed per step."""
        self.stove, self.arena, self.opt, self.clip = stove, arena, optimizer, clip
        self.supair_only, self.warmup = supair_only, warmup
        self.world_size = world_size
        self.reward_loss = reward_loss
        self.alias_inputs = alias_inputs
        self.force_reduce = force_reduce          # send a single rank's gradient through the collective anyway (the 1-GPU RCCL test)
        self.capture_reduce = capture_reduce      # data parallel: the all-reduce as a node of the optimiser graph (False: eager, between the graphs)
        self.reduce_captured = False
        self.update_eager = False          # the optimiser's launches as eager calls (only after a failed capture of the all-reduce)
        self.graphs = None
        self.key = None
        self.reward_value = None
        self._draws = 0
        self._side_exec = None
        self._events = None          # the list of cross-capture events that order g_main and g_side (owned with the graphs)

    def _drop_side(self):
        """The side graph and the events that order it against the main graph go together (a re-capture, eager(), __del__)."""
        from . import _lib
        lib = _lib.load()
        if self._side_exec:
            lib.stove_graph_destroy(self._side_exec)
        self._side_exec = None
        if self._events:
            lib.stove_event_list_destroy(self._events)
        self._events = None

    def __del__(self):
        try:
            self.graphs = None       # the main graph's event nodes go first
            self._drop_side()
        except Exception:
            pass

    # ------------------------------------------------------------------ the step, in the two pieces a collective may separate
    def _fwd_bwd(self):
        self.arena.zero()
        # step_counter 1 with print_every / plot_every > 1: the no-logging branch of Stove.forward
        elbo, _, rewards = self.stove(self.x, 1, self.a, self.supair_only)
        if self.reward_loss is not None and self.r is not None:
            rl = self.reward_loss(rewards.flatten(), self.r.flatten())
            (-1.0 * elbo + self.hyper_dev[5] * rl).backward()
            self.reward_value = rl.detach()
        else:
            elbo.backward(self._minus_one)       # d(-ELBO)
        return elbo.detach()         # nothing may keep the autograd graph (and its AccumulateGrad nodes) alive

    def _update(self):
        self.opt.step(max_norm=self.clip, hyper_dev=self.hyper_dev)

    def _reduce(self):
        if self.world_size > 1 or self.force_reduce:
            self.arena.all_reduce(force=self.force_reduce)

    def _eager(self):
        elbo = self._fwd_bwd()
        self._reduce()
        self._update()
        return elbo

    def _prepare(self, dev):
        if getattr(self, 'hyper_dev', None) is None or self.hyper_dev.device != dev:
            self._minus_one = torch.tensor(-1.0, device=dev, dtype=torch.float32)
            self.hyper_dev = torch.zeros(NHYPER, dtype=torch.float32, device=dev)

    def eager(self, images, actions=None, targets=None, reward_weight=0.0):
        """The same step enqueued launch by launch on the current stream (what the replay is checked against)."""
        self._prepare(images.device)
        self.x, self.a, self.r = images, actions, targets
        self.hyper_dev.copy_(torch.tensor(self._hyper_now(reward_weight), dtype=torch.float32))
        elbo = self._eager()
        self.opt.count_step()
        if self.graphs is not None:
            torch.cuda.synchronize(images.device)     # replays still in flight hold the graphs' event nodes
            self.graphs = None         # the static inputs were rebound: capture again before the next replay
            self._drop_side()
        return elbo

    # ------------------------------------------------------------------ capture
    def _capture(self, images, actions, targets):
        dev = images.device
        own = (lambda t: t) if self.alias_inputs else (lambda t: t.clone())
        self.x = own(images)
        self.a = own(actions) if actions is not None else None
        self.r = own(targets) if targets is not None else None
        self._prepare(dev)
        self.ring = torch.zeros(SLOTS, NHYPER, dtype=torch.float32).pin_memory()
        self._slot_events = [None] * SLOTS
        self._n = 0
        self.hyper_dev.copy_(torch.tensor(self._hyper_now(0.0), dtype=torch.float32))
        # warm-up on a side stream (allocator pools, library workspaces, lazily created streams); parameters, optimiser
        # state and the generator are put back afterwards, so capturing does not count as training
        snap = (self.arena.data.clone(), {k: v.clone() for k, v in self.opt._flat.items()}, torch.cuda.get_rng_state(dev),
                self.opt._seg_steps.clone())

        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            self.arena.data.copy_(snap[0])
            for k, v in snap[1].items():
                self.opt._flat[k].copy_(v)
            self.opt._seg_steps.copy_(snap[3])
        torch.cuda.set_rng_state(snap[2], dev)      # ... which also puts the library's noise stream back (ops.NoiseSource shadows it)
        # Capture on the stream the warm-up ran on (the autograd nodes then see one stream throughout), as THREE graphs:
        #   g_main  zero_grad + forward + backward, with its short fork / join episodes (table bake, background-SPN chain)
        #   g_side  the parameter-gradient chain of the backward pass (SPN table gradients, the recursion's and the recognition
        #           network's weight gradients): a separate capture of the side stream, ordered against g_main by event nodes
        #           (stove_stream_after).  As branches of ONE graph the runtime put this chain on the main chain's queue, behind
        #           it (3.6 instead of 3.1 ms per step); as its own graph on its own stream it runs where the eager step has it.
        #   g_opt   clip + Adam, after the join of the side stream [and the all-reduce]
        self._stream = s
        from . import _lib, ops
        lib = _lib.load()
        self._side = ops._side_stream(dev)
        # debugging hooks (tools/graphdump, not part of the package): checked BEFORE the capture begins
        dump, dump_open = os.environ.get('STOVE_GRAPH_DUMP'), os.environ.get('STOVE_GRAPH_DUMP_OPEN')
        gd = None
        if dump or dump_open:
            import ctypes
            so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'graphdump', 'libgraphdump.so')
            if not os.path.exists(so):
                raise RuntimeError('STOVE_GRAPH_DUMP* needs %s (hipcc -shared -fPIC tools/graphdump/graphdump.hip): a debugging aid of '
                                   'the source tree, not of an installed package' % so)
            gd = ctypes.CDLL(so)
        g1 = torch.cuda.CUDAGraph(keep_graph=True) if dump else torch.cuda.CUDAGraph()
        split = True                              # (the one-graph capture of round 3 is gone: module docstring)
        torch.cuda.synchronize(dev)
        self.graphs = None                        # a re-capture (new batch shape): the previous graphs and their events go
        self._drop_side()
        import ctypes
        side_graph, side_nodes = ctypes.c_void_p(), ctypes.c_int()
        events = lib.stove_event_list_begin()     # the events that order the two captures: created below, destroyed with the graphs
        ns = getattr(self.stove, '_noise_source', None)          # exists after the warm-up steps if the step draws from it
        if ns is not None:
            ns.captured = 0
        try:
            self._capture_graphs(g1, s, split, lib, ops, side_graph, side_nodes, dump, dump_open, gd)
        except BaseException:
            lib.stove_event_list_end(events)
            lib.stove_event_list_destroy(events)
            raise
        lib.stove_event_list_end(events)
        self._events = events
        self._draws = ns.captured if ns is not None else 0       # draws from the library's generator inside the captured step
        self.key = self._key(images, actions, targets)

    def _capture_graphs(self, g1, s, split, lib, ops, side_graph, side_nodes, dump, dump_open, gd):
        import ctypes
        from . import _lib
        rc = 0
        with torch.cuda.graph(g1, stream=s):
            # the side capture lives strictly INSIDE the main one: torch synchronises the device before it begins a capture and
            # flushes deferred allocator events after it ends one -- either would invalidate a capture still open on the side stream
            if split:
                _lib.check(lib.stove_capture_begin(self._side.cuda_stream), 'stove_capture_begin')
                ops.SideMode.split, ops.SideMode.keep = True, []
            try:
                self.elbo = self._fwd_bwd()
            except BaseException:
                if split:            # the step failed inside the capture: end the side capture, free what it produced, keep the error
                    ops.SideMode.split = False
                    lib.stove_capture_end(self._side.cuda_stream, ctypes.byref(side_graph), ctypes.byref(side_nodes))
                    if side_graph.value:
                        ex = ctypes.c_void_p()
                        lib.stove_graph_instantiate(side_graph, ctypes.byref(ex))      # consumes (destroys) the graph ...
                        if ex.value:
                            lib.stove_graph_destroy(ex)                                # ... and the executable made from it goes too
                    ops.SideMode.keep = []
                raise
            if split:
                ops.SideMode.split = False
                rc = lib.stove_capture_end(self._side.cuda_stream, ctypes.byref(side_graph), ctypes.byref(side_nodes))
                _lib.check(rc, 'stove_capture_end')
            elif self.world_size <= 1 and not self.force_reduce:
                self._update()
            if dump_open:          # debugging: the main DAG while its capture is still open
                gd.graph_of_stream.restype = ctypes.c_void_p
                gd.graph_dump(ctypes.c_void_p(gd.graph_of_stream(ctypes.c_void_p(s.cuda_stream))), dump_open.encode())
        ops.SideMode.split, ops.SideMode.keep = False, []
        if split:
            ex = ctypes.c_void_p()
            _lib.check(lib.stove_graph_instantiate(side_graph, ctypes.byref(ex)), 'stove_graph_instantiate')
            self._side_exec, self._side_nodes = ex.value, side_nodes.value
        if dump:
            gd.graph_dump(ctypes.c_void_p(g1.raw_cuda_graph()), dump.encode())
            g1.instantiate()
        if split or self.world_size > 1 or self.force_reduce:
            # g_opt.  Data parallel: the gradient all-reduce is captured INTO it (RCCL collectives are capturable: the communicator
            # exists since the warm-up steps), so a replayed step is three graph launches and no host-issued collective in between.
            # If the capture of the collective fails on this stack, the all-reduce stays an eager call between the graphs.
            self.reduce_captured = False
            self.update_eager = False
            g2 = None
            import torch.distributed as dist
            # (only RCCL's collectives can be captured: gloo's all-reduce of a device tensor goes through the host, which invalidates
            # the capture and leaves the stream in a state the fallback cannot recover from)
            if (self.world_size > 1 or self.force_reduce) and self.capture_reduce and dist.is_initialized() and _cuda_backend() == 'nccl':
                try:
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, stream=s, pool=g1.pool()):
                        self._reduce()
                        self._update()
                    self.reduce_captured = True
                except Exception as exc:
                    # A capture that failed half-way may leave the stream unable to begin another one (seen with a backend whose
                    # collective synchronises with the host): no second attempt at a graph -- the all-reduce AND the optimiser's three
                    # launches stay eager calls behind the two backward graphs, after the stream and the communicator have been
                    # checked (a stream still capturing, or a collective that no longer completes, is an error, not a slow path).
                    import warnings
                    warnings.warn('GraphedTrainStep: the gradient all-reduce could not be captured (%r); all-reduce and optimiser '
                                  'run eagerly behind the backward graphs' % (exc,))
                    g2 = None
                    self.update_eager = True
                    torch.cuda.synchronize(self.x.device)
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError('GraphedTrainStep: the stream is still capturing after the failed capture of the all-reduce') from exc
                    probe = torch.ones(1, device=self.x.device)
                    dist.all_reduce(probe)
                    torch.cuda.synchronize(self.x.device)
                    if float(probe) != float(dist.get_world_size()):
                        raise RuntimeError('GraphedTrainStep: the communicator does not reduce correctly after the failed capture') from exc
            if g2 is None and not self.update_eager:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=s, pool=g1.pool()):
                    self._update()
            if self.world_size > 1 and dist.is_initialized():
                # every rank issues one all-reduce per step either way, but the ranks should agree on HOW (a mixed job is a sign of
                # a broken stack on some rank): say so once
                took = torch.tensor([1.0 if self.reduce_captured else 0.0], device=self.x.device)
                dist.all_reduce(took)
                if 0.0 < float(took) < float(dist.get_world_size()):
                    import warnings
                    warnings.warn('GraphedTrainStep: %d of %d ranks captured the all-reduce into the optimiser graph, the others call it eagerly'
                                  % (int(float(took)), dist.get_world_size()))
            self.graphs = (g1, g2)
        else:
            self.graphs = (g1,)

    @staticmethod
    def _key(images, actions, targets):
        return tuple(None if t is None else (tuple(t.shape), t.dtype) for t in (images, actions, targets))

    def _hyper_now(self, reward_weight):
        return self.opt.hyper(self.clip) + [float(reward_weight), 0.0, 0.0]

    def static_images(self):
        """The tensor the captured step reads its frames from (None before the first call): a loader that gathers its batch
        straight into it and passes it back skips the copy."""
        return getattr(self, 'x', None) if self.graphs is not None else None

    def __call__(self, images, actions=None, targets=None, reward_weight=0.0):
        """One optimisation step on the batch; returns the ELBO (device scalar, overwritten by the next call)."""
        if self.graphs is None or self._key(images, actions, targets) != self.key:
            self._capture(images, actions, targets)
        for dst, src in ((self.x, images), (self.a, actions), (self.r, targets)):
            if dst is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        slot = self._n % SLOTS
        if self._slot_events[slot] is not None:
            self._slot_events[slot].synchronize()     # the step that last used this slot has run (blocks only SLOTS steps ahead)
        else:
            self._slot_events[slot] = torch.cuda.Event()
        self.ring[slot] = torch.tensor(self._hyper_now(reward_weight), dtype=torch.float32)
        from . import _lib
        lib = _lib.load()
        main = torch.cuda.current_stream(self.x.device).cuda_stream
        self.hyper_dev.copy_(self.ring[slot], non_blocking=True)
        ns = getattr(self.stove, '_noise_source', None)
        if ns is not None and self._draws:
            ns.prepare(self._draws)        # the captured draw(s) of this replay: host generator in step, device state rewritten after a reseed
        self.graphs[0].replay()
        if self._side_exec:
            _lib.check(lib.stove_graph_launch(self._side_exec, self._side.cuda_stream), 'stove_graph_launch')
            _lib.check(lib.stove_stream_after(main, self._side.cuda_stream), 'stove_stream_after')      # the optimiser reads what it wrote
        if len(self.graphs) > 1:
            if not self.reduce_captured:
                self._reduce()
            if self.update_eager:
                self._update()
            else:
                self.graphs[1].replay()
        self._slot_events[slot].record()
        self._n += 1
        self.opt.count_step()
        return self.elbo
