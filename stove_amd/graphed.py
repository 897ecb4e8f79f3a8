"""The training step as ONE captured hipGraph (torch.cuda.CUDAGraph on ROCm).

At the reference's default training shape (256 clips x 8 visible frames, config.py:17-21) the GPU needs ~0.7 ms per step and
the host ~2.0 ms to enqueue its ~100 launches (tools/host_time.py): the step is launch-bound.  Capturing
zero_grad + Stove.forward + backward + clip + Adam once and replaying it removes the host from the loop.  What varies from
step to step enters through device memory: the batch (copied into static input tensors), the optimiser's step-dependent
constants (`FlatAdam.hyper`, 5 floats: the learning rate and the fixed betas / eps / clip norm; step counts live on the device), the random draws (torch's graph-safe Philox offsets).

Only steps that need nothing but the loss are replayed: logging steps (`step % print_every == 0`, which read prop_dict) and
multi-process runs (the all-reduce sits between backward and the optimiser) go through the eager path.
"""
import torch


class GraphedTrainStep:
    def __init__(self, stove, arena, optimizer, clip, supair_only=False, warmup=3):
        self.stove, self.arena, self.opt, self.clip = stove, arena, optimizer, clip
        self.supair_only, self.warmup = supair_only, warmup
        self.graph = None
        self.key = None

    def _eager(self, images, actions, hyper_dev=None):
        self.arena.zero()
        # step_counter 1 with print_every / plot_every > 1: the no-logging branch of Stove.forward
        elbo, _, _ = self.stove(images, 1, actions, self.supair_only)
        (-1.0 * elbo).backward()
        self.opt.step(max_norm=self.clip, hyper_dev=hyper_dev)
        return elbo.detach()         # nothing may keep the autograd graph (and its AccumulateGrad nodes) alive

    def _capture(self, images, actions):
        dev = images.device
        self.x = images.clone()
        self.a = actions.clone() if actions is not None else None
        self.hyper_dev = torch.empty(5, dtype=torch.float32, device=dev)
        # warm-up on a side stream (allocator pools, BLAS workspaces, lazily created streams); parameters, optimiser state
        # and the generator are put back afterwards, so capturing does not count as training
        snap = (self.arena.data.clone(), {k: v.clone() for k, v in self.opt._flat.items()}, torch.cuda.get_rng_state(dev),
                self.opt._seg_steps.clone())
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                self._set_hyper()
                self._eager(self.x, self.a, self.hyper_dev)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            self.arena.data.copy_(snap[0])
            for k, v in snap[1].items():
                self.opt._flat[k].copy_(v)
            self.opt._seg_steps.copy_(snap[3])
        torch.cuda.set_rng_state(snap[2], dev)
        # capture on the stream the warm-up ran on: the autograd nodes then see one stream throughout
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s):
            self.elbo = self._eager(self.x, self.a, self.hyper_dev)
        self.key = (tuple(images.shape), None if actions is None else tuple(actions.shape))

    def _set_hyper(self):
        # from pageable memory: the copy is staged before the call returns, so the next step's values cannot overtake it
        # (a pinned buffer rewritten by a host that runs ahead of the device did exactly that)
        self.hyper_dev.copy_(torch.tensor(self.opt.hyper(self.clip), dtype=torch.float32))

    def __call__(self, images, actions=None):
        """One optimisation step on the batch; returns the ELBO (device scalar, overwritten by the next call)."""
        key = (tuple(images.shape), None if actions is None else tuple(actions.shape))
        if self.graph is None or key != self.key:
            self._capture(images, actions)
        self.x.copy_(images, non_blocking=True)
        if actions is not None:
            self.a.copy_(actions, non_blocking=True)
        self._set_hyper()
        self.graph.replay()
        self.opt.count_step()
        return self.elbo
