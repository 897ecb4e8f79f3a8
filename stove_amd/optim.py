"""Adam / AMSGrad over a ParamArena as ONE kernel (csrc/arena.hip), gradient clipping folded in.

torch.optim.Adam's multi-tensor path costs ~25 launches (0.2 ms) per step on the 162 parameter tensors; with
the parameters, gradients and moments flat it is one elementwise pass.  The class is a `torch.optim.Optimizer`
whose per-parameter state tensors (`exp_avg`, `exp_avg_sq`, `max_exp_avg_sq`) are views into flat buffers, so
`state_dict()` / `load_state_dict()` keep the reference's checkpoint layout (train.py:150-175) and the
learning-rate schedule still writes `param_groups[0]['lr']` (train.py:431-440).
"""
import torch

from . import _lib


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, amsgrad=False):
        if arena.data.dtype != torch.float32 or not arena.data.is_cuda:
            raise RuntimeError('FlatAdam needs a float32 ParamArena on a GPU')
        super().__init__(arena.params, dict(lr=lr, betas=betas, eps=eps, amsgrad=amsgrad, weight_decay=0))
        self.arena = arena
        self._steps = 0
        names = ['exp_avg', 'exp_avg_sq'] + (['max_exp_avg_sq'] if amsgrad else [])
        self._flat = {k: torch.zeros_like(arena.data) for k in names}
        self._bind()

    def _bind(self):
        for p in self.arena.params:
            st = self.state[p]
            st['step'] = torch.tensor(float(self._steps))
            for k, flat in self._flat.items():
                st[k] = self.arena.view_of(p, flat)

    def hyper(self, max_norm=None, step=None):
        """[lr, beta1, beta2, eps, 1 - beta1^step, sqrt(1 - beta2^step), max_norm] of step `step` (default: the next one),
        the constants `stove_flat_adam_dev` reads from device memory (captured-graph steps, stove_amd/graphed.py)."""
        g = self.param_groups[0]
        t = self._steps + 1 if step is None else step
        b1, b2 = float(g['betas'][0]), float(g['betas'][1])
        return [float(g['lr']), b1, b2, float(g['eps']), 1.0 - b1 ** t, (1.0 - b2 ** t) ** 0.5, float(max_norm or 0.0)]

    @torch.no_grad()
    def step(self, closure=None, max_norm=None, hyper_dev=None):
        """One step on arena.grad.  `max_norm`: clip_grad_norm_(params, max_norm) applied on the fly (the gradient
        buffer itself keeps the unclipped values); returns the total gradient norm (device scalar) if clipping.
        `hyper_dev`: device tensor holding `self.hyper(...)` -- the step then takes its constants from there (and does
        not advance the host-side step count: the owner of the captured graph does, see count_step)."""
        if closure is not None:
            raise NotImplementedError('closures are not supported')
        ar = self.arena
        ar.check()
        group = self.param_groups[0]
        norm = None
        if hyper_dev is not None:
            with torch.cuda.device(ar.data.device):
                if max_norm is not None:
                    norm = torch.linalg.vector_norm(ar.grad).reshape(1)
                vmax = self._flat.get('max_exp_avg_sq')
                _lib.check(_lib.load().stove_flat_adam_dev(
                    ar.data.data_ptr(), ar.grad.data_ptr(), self._flat['exp_avg'].data_ptr(), self._flat['exp_avg_sq'].data_ptr(),
                    None if vmax is None else vmax.data_ptr(), None if norm is None else norm.data_ptr(), ar.numel,
                    hyper_dev.data_ptr(), _lib.stream()), 'stove_flat_adam_dev')
            return norm
        self._steps += 1
        with torch.cuda.device(ar.data.device):
            if max_norm is not None:
                norm = torch.linalg.vector_norm(ar.grad).reshape(1)
            vmax = self._flat.get('max_exp_avg_sq')
            _lib.check(_lib.load().stove_flat_adam(
                ar.data.data_ptr(), ar.grad.data_ptr(), self._flat['exp_avg'].data_ptr(), self._flat['exp_avg_sq'].data_ptr(),
                None if vmax is None else vmax.data_ptr(), None if norm is None else norm.data_ptr(), ar.numel,
                float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps']), self._steps,
                float(max_norm) if max_norm is not None else 0.0, _lib.stream()), 'stove_flat_adam')
        return norm

    def count_step(self):
        """A step was applied outside step() (replay of a captured graph)."""
        self._steps += 1

    def state_dict(self):
        for p in self.arena.params:
            self.state[p]['step'] = torch.tensor(float(self._steps))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Accepts a torch.optim.Adam state dict of the same parameter list (parameters that never received a
        gradient have no entry there: their moments stay zero)."""
        super().load_state_dict(state_dict)
        steps = 0
        with torch.no_grad():
            for p in self.arena.params:
                st = self.state.get(p, {})
                for k, flat in self._flat.items():
                    if k in st:
                        self.arena.view_of(p, flat).copy_(st[k])
                if 'step' in st:
                    steps = max(steps, int(float(st['step'])))
        self._steps = steps
        self._bind()
