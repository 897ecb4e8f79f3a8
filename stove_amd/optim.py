"""Adam / AMSGrad over a ParamArena as three launches (csrc/arena.hip: gradient scan + norm, update, step tick), gradient
clipping folded in.

torch.optim.Adam's multi-tensor path costs ~25 launches (0.2 ms) per step on the 162 parameter tensors; with
the parameters, gradients and moments flat it is one elementwise pass plus a scan that finds the gradient norm and which
parameter tensors received a gradient at all.  The class is a `torch.optim.Optimizer`
whose per-parameter state tensors (`exp_avg`, `exp_avg_sq`, `max_exp_avg_sq`) are views into flat buffers, so
`state_dict()` / `load_state_dict()` keep the reference's checkpoint layout (train.py:150-175) and the
learning-rate schedule still writes `param_groups[0]['lr']` (train.py:431-440).
"""
import torch

from . import _lib


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam's update, per-parameter semantics included: every parameter tensor (an arena SEGMENT) keeps its own
    step count, and a tensor steps only when it is trainable and received a gradient this step -- in the flat buffer "no
    gradient" is a slice that is exactly zero (frozen SuPAIR parameters under `supair_grad=False`, the dynamics cores 1-2
    that are never run, the dynamics network during `supair_only` pretraining).  So loaded moments of frozen parameters do
    not move them, and parameters that start training late get their own bias corrections (include/stove_hip.h).

    Deliberate deviation: a tensor whose gradient is exactly ZERO everywhere is treated like `grad is None` (no step, no moment
    decay, no step-count tick).  torch.optim.Adam still steps a parameter whose `.grad` is a zero tensor (its momentum keeps
    moving it) -- the reference reaches that state only through `optimizer.zero_grad()` leaving zero tensors on parameters that
    then receive no gradient, which for its models means "never used" (dynamics cores 1-2) or "frozen by a training phase":
    the cases listed above, where not moving is what torch does for `grad is None`.  `strict_zero_grad=True`
    (`config.strict_adam_zero_grad`) gives the other reading: once a tensor has received a gradient it keeps stepping on zero
    gradients, as torch.optim.Adam does on zero-filled `.grad` tensors.  tests/test_gpu_optim.py pins both readings."""

    def __init__(self, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, amsgrad=False, strict_zero_grad=False):
        if arena.data.dtype != torch.float32 or not arena.data.is_cuda:
            raise RuntimeError('FlatAdam needs a float32 ParamArena on a GPU')
        super().__init__(arena.params, dict(lr=lr, betas=betas, eps=eps, amsgrad=amsgrad, weight_decay=0))
        self.arena = arena
        # strict_zero_grad: a tensor that has once received a gradient keeps stepping on all-zero gradients (see the class docstring)
        self.strict_zero_grad = bool(strict_zero_grad)
        self._steps = 0                     # calls of step(); the per-segment counts live on the device
        names = ['exp_avg', 'exp_avg_sq'] + (['max_exp_avg_sq'] if amsgrad else [])
        self._flat = {k: torch.zeros_like(arena.data) for k in names}
        dev = arena.data.device
        nseg = len(arena.params)
        seg = torch.zeros(arena.numel // 4, dtype=torch.int32)
        bounds = [arena.offset[id(p)] // 4 for p in arena.params] + [arena.numel // 4]
        for s_, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
            seg[lo:hi] = s_
        self._seg_of4 = seg.to(dev)
        self._seg_steps = torch.zeros(nseg, dtype=torch.float32, device=dev)
        self._trainable_key, self._trainable = None, None
        self._ws = torch.zeros(_lib.load().stove_flat_adam_ws_bytes(nseg), dtype=torch.uint8, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._bind()

    def _bind(self):
        steps = self._seg_steps.cpu().tolist()
        for p, t in zip(self.arena.params, steps):
            st = self.state[p]
            st['step'] = torch.tensor(float(t))
            for k, flat in self._flat.items():
                st[k] = self.arena.view_of(p, flat)

    def _trainable_dev(self):
        key = tuple(p.requires_grad for p in self.arena.params)
        if key != self._trainable_key:
            self._trainable = torch.tensor(key, dtype=torch.uint8).to(self.arena.data.device)
            self._trainable_key = key
        return self._trainable

    def hyper(self, max_norm=None):
        """[lr, beta1, beta2, eps, max_norm]: the constants a captured step reads from device memory (stove_amd/graphed.py);
        the bias corrections are formed on the device from the per-segment step counts."""
        g = self.param_groups[0]
        return [float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(max_norm or 0.0)]

    @torch.no_grad()
    def step(self, closure=None, max_norm=None, hyper_dev=None):
        """One step on arena.grad.  `max_norm`: clip_grad_norm_(params, max_norm) applied on the fly (the gradient
        buffer itself keeps the unclipped values); returns the total gradient norm (device scalar, overwritten by the next
        step) if clipping.  `hyper_dev`: device tensor holding `self.hyper(...)` -- the step then takes its constants from
        there (captured graphs; the owner of the graph calls count_step per replay)."""
        if closure is not None:
            raise NotImplementedError('closures are not supported')
        ar = self.arena
        ar.check()
        group = self.param_groups[0]
        vmax = self._flat.get('max_exp_avg_sq')
        if hyper_dev is None:
            self._steps += 1
        ar.drop_prefetched()                # tables baked from the old parameters are stale from here on
        with torch.cuda.device(ar.data.device):
            _lib.check(_lib.load().stove_flat_adam(
                ar.data.data_ptr(), ar.grad.data_ptr(), self._flat['exp_avg'].data_ptr(), self._flat['exp_avg_sq'].data_ptr(),
                None if vmax is None else vmax.data_ptr(), ar.numel, self._seg_of4.data_ptr(), self._trainable_dev().data_ptr(),
                self._seg_steps.data_ptr(), len(ar.params), self._ws.data_ptr(), self._norm.data_ptr(),
                None if hyper_dev is None else hyper_dev.data_ptr(), float(group['lr']), float(group['betas'][0]),
                float(group['betas'][1]), float(group['eps']), float(max_norm) if max_norm is not None else 0.0,
                (1 if max_norm is not None else 0) | (2 if self.strict_zero_grad else 0), _lib.stream()), 'stove_flat_adam')
        return self._norm if max_norm is not None else None

    def count_step(self):
        """A step was applied outside step() (replay of a captured graph)."""
        self._steps += 1

    def state_dict(self):
        self._bind()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Accepts a torch.optim.Adam state dict of the same parameter list (parameters that never received a
        gradient have no entry there: their moments and step counts stay zero)."""
        super().load_state_dict(state_dict)
        steps = []
        with torch.no_grad():
            for p in self.arena.params:
                st = self.state.get(p, {})
                for k, flat in self._flat.items():
                    view = self.arena.view_of(p, flat)
                    if k in st:
                        view.copy_(st[k])
                    else:
                        view.zero_()
                steps.append(float(st['step']) if 'step' in st else 0.0)
        self._seg_steps.copy_(torch.tensor(steps, dtype=torch.float32))
        self.arena.drop_prefetched()
        self._steps = int(max(steps)) if steps else 0
        self._bind()
