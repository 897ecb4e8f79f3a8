"""Flat parameter arena: every parameter of the model is a view into ONE fp32 buffer, every
gradient a view into ONE gradient buffer.

Why (MI355X-first, no counterpart in the reference, which keeps 162 separate tensors and lets
ATen bake / differentiate them one by one):
  * the SPN tables and the GNN parameter image are baked from the arena by two kernels, and
    their gradients are pushed back by two kernels (csrc/arena.hip) -- this replaces ~250 small
    ATen launches per training step (stack / index / sigmoid / softmax, their backward, and the
    61 + 26 gradient clones of AccumulateGrad);
  * the gradient buffer IS the data-parallel all-reduce bucket: no pack / unpack;
  * gradient clipping is one norm + one scale over the flat buffer.
The reference's parameter names, shapes and state-dict keys are untouched (parameters stay
`nn.Parameter`s of their modules, only their storage moves), so checkpoints and optimiser
state interchange.

Contract: build the arena AFTER the model is on its device / dtype, zero gradients with
`arena.zero_grad()` (not `optimizer.zero_grad()`, which would detach the gradient views); the
kernels check that the views are intact and raise otherwise.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib, ops
from . import settings as _settings


def _unique_params(module):
    seen, out = set(), []
    for name, p in module.named_parameters():
        if id(p) not in seen:
            seen.add(id(p))
            out.append((name, p))
    return out


class ParamArena:
    ALIGN = 4           # floats: every tensor starts 16-byte aligned

    def __init__(self, model, world_size=None):
        named = _unique_params(model)
        if not named:
            raise ValueError('model has no parameters')
        dev, dt = named[0][1].device, named[0][1].dtype
        for n, p in named:
            if p.device != dev or p.dtype != dt:
                raise ValueError('parameter %s is %s/%s, arena is %s/%s' % (n, p.device, p.dtype, dev, dt))
        self.model = model
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.offset, off = {}, 0
        for p in self.params:
            self.offset[id(p)] = off
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.data = torch.zeros(off, dtype=dt, device=dev)
        self.grad = torch.zeros(off, dtype=dt, device=dev)
        with torch.no_grad():
            for p in self.params:
                o, n = self.offset[id(p)], p.numel()
                self.data[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.data[o:o + n].view(p.shape)
        self._on_gpu_params = dev.type == 'cuda' and dt == torch.float32
        self._bind_grads()
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.world_size = world_size
        self._spn = self._gnn = None
        sup = model if hasattr(model, 'obj_spn') else getattr(model, 'sup', None)         # a bare Supair / Dynamics works too
        dyn = model if hasattr(model, 'param_image') else getattr(model, 'dyn', None)
        self._on_gpu = dev.type == 'cuda' and dt == torch.float32        # the index plans are device-agnostic
        if sup is not None and hasattr(sup, 'obj_spn'):
            self._plan_spn(sup)
            sup._arena = self
        if dyn is not None and hasattr(dyn, 'param_image'):
            self._plan_gnn(dyn)
            dyn._arena = self

    # ------------------------------------------------------------------ views
    def _bind_grads(self):
        for p in self.params:
            if not p.requires_grad:                     # frozen (Trainer.disable_supair_grad): no gradient view
                p.grad = None
                if self._on_gpu_params:
                    ops.register_grad_view(p, None)
                continue
            o, n = self.offset[id(p)], p.numel()
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + self.grad.element_size() * o:
                p.grad = self.grad[o:o + n].view(p.shape)
            if self._on_gpu_params:                     # kernels that produce this parameter's gradient add it here themselves
                ops.register_grad_view(p, p.grad)

    def view_of(self, p, flat=None):
        o = self.offset[id(p)]
        return (self.data if flat is None else flat)[o:o + p.numel()].view(p.shape)

    def check(self):
        """The kernels read / write the flat buffers: refuse to run on detached views."""
        es = self.data.element_size()
        for p in (self.params[0], self.params[-1]):
            o = self.offset[id(p)]
            if p.data_ptr() != self.data.data_ptr() + es * o:
                raise RuntimeError('ParamArena: parameter storage was rebound (model.to()/type() after the arena '
                                   'was built?) -- build a new ParamArena')
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + es * o):
                raise RuntimeError('ParamArena: gradient views were detached -- use arena.zero_grad() instead of '
                                   'optimizer.zero_grad() / p.grad = None')

    # ------------------------------------------------------------------ GradBucket interface
    def zero_grad(self):
        self.grad.zero_()
        self._bind_grads()

    def drop_prefetched(self):
        """Forget tables / images baked ahead of time (prefetch_images) that nobody consumed: they were made from the
        parameters as they were THEN.  Called at the end of Stove.forward and by everything that changes the parameters."""
        self._pre = None

    zero = zero_grad

    def pack(self):
        return self.grad

    def all_reduce(self, force=False):
        """Average the gradients over the ranks: ONE collective on the flat buffer (no-op for one rank; `force` sends a
        single rank's buffer through the backend anyway -- the 1-GPU RCCL test)."""
        if self.world_size <= 1 and not force:
            return
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
        if self.world_size > 1:
            self.grad.div_(self.world_size)

    def sync(self, src=0):
        """Every rank takes rank `src`'s parameters: ONE broadcast of the flat buffer (all parameters are views into it)."""
        if self.world_size > 1:
            dist.broadcast(self.data, src)
        self.drop_prefetched()

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_(params, max_norm) on the flat buffer (pads are zero)."""
        total = torch.linalg.vector_norm(self.grad)
        self.grad.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total

    # ------------------------------------------------------------------ SPN tables
    def _plan_spn(self, sup):
        obj, bg = sup.obj_spn, sup.bg_spn
        if obj._kind != 'obj' or bg._kind != 'bg' or bg.num_dims != 1024:
            return                      # other frame sizes: the SPN tables are baked / differentiated per tensor (RatSpn.tables + autograd)
        dev = self.data.device
        off = self.offset
        oleaves, osums = list(obj.vector_list[0]), list(obj.vector_list[2])
        lo, so = obj._plan_cpu['leaf_order'].tolist(), obj._plan_cpu['sum_order'].tolist()
        bleaves = list(bg.vector_list[0])

        def i32(v):
            return torch.tensor(v, dtype=torch.int32, device=dev)
        t = {'obj_mu': i32([off[id(oleaves[i].means)] for i in lo]),
             'obj_rho': i32([off[id(oleaves[i].sigma_params)] for i in lo]),
             'obj_sum': i32([off[id(osums[i].params)] for i in so]),
             'bg_mu': i32([off[id(v.means)] for v in bleaves]),
             'bg_rho': i32([off[id(v.sigma_params)] for v in bleaves]),
             'bg_gidx': bg._plan_cpu['gidx'].to(torch.int32).reshape(-1).to(dev)}
        plan = _lib.SpnArenaPlan()
        for k, v in t.items():
            setattr(plan, k, v.data_ptr())
        plan.obj_root = off[id(obj.output_vector.params)]
        plan.bg_root = off[id(bg.output_vector.params)]
        plan.obj_vmin, plan.obj_vmax = obj.args.gauss_min_sigma, obj.args.gauss_max_sigma
        plan.bg_vmin, plan.bg_vmax = bg.args.gauss_min_sigma, bg.args.gauss_max_sigma
        opl, bpl = obj._plan(dev), bg._plan(dev)
        self._spn = {'plan': plan, 'keep': t, 'ints': (opl['scope'], opl['leaf_slot'], bpl['side']),
                     'params': list(obj.parameters()) + list(bg.parameters())}

    @property
    def has_spn(self):
        return self._on_gpu and self._spn is not None

    def prefetch_images(self, cores=(0,)):
        """Bake the SPN tables and gather the GNN parameter image(s) on the second stream NOW (called at the top of
        Stove.forward): they depend on the parameters only, so their two small launches leave the critical path between the
        recognition network and the recursion / the scene likelihood.  Consumed once by spn_tables() / gnn_image()."""
        if not self._on_gpu or not _settings.OVERLAP:
            return
        dev = self.data.device
        main, side = torch.cuda.current_stream(dev), ops._side_stream(dev, 'pre')      # its own stream: joined again inside the forward
        side.wait_stream(main)                       # behind the optimiser's update of self.data
        with torch.cuda.stream(side):
            pre = {}
            if self.has_spn:
                pre['spn'] = self._bake_spn()
            if self.has_gnn:
                for k in cores:
                    pre[('gnn', k)] = self._gather_gnn(k)
            ev = torch.cuda.Event()
            ev.record(side)
        self._pre = (pre, ev, side)

    def _take(self, key):
        pre = getattr(self, '_pre', None)
        if pre is None or key not in pre[0]:
            return None
        out = pre[0].pop(key)
        torch.cuda.current_stream(self.data.device).wait_event(pre[1])
        for t in (out if isinstance(out, tuple) else (out,)):
            t.record_stream(torch.cuda.current_stream(self.data.device))
        return out

    def spn_tables(self):
        """-> obj_tabs (coef, wsum, wroot, scope, leaf_slot), bg_tabs (coef, wroot, side), baked by one launch."""
        self.check()
        bufs = self._take('spn')
        if bufs is None:
            bufs = self._bake_spn()
        oc, ow, orr, bc, bw, dense = bufs
        scope, slot, side = self._spn['ints']
        return ((oc.view(24, 25, 10, 3), ow.view(12, 100, 10), orr.view(6, 100), scope, slot),
                (bc.view(3, 1024, 6, 3), bw.view(3, 36), side, dense))

    def _bake_spn(self):
        lib = _lib.load()
        dev = self.data.device
        with torch.cuda.device(dev):
            buf = torch.empty(18000 + 12000 + 600 + 55296 + 108, dtype=torch.float32, device=dev)
            oc, ow, orr, bc, bw = torch.split(buf, [18000, 12000, 600, 55296, 108])
            _lib.check(lib.stove_spn_bake(self.data.data_ptr(), ctypes.byref(self._spn['plan']), oc.data_ptr(), ow.data_ptr(),
                                          orr.data_ptr(), bc.data_ptr(), bw.data_ptr(), _lib.stream()), 'stove_spn_bake')
            # the background coefficients once more as the operand image of the scene forward's leaf GEMM: parameters only,
            # so it is made here (ahead of time on the second stream when prefetched) and not at the head of the scene chain
            dense = torch.empty(lib.stove_bg_dense_floats(), dtype=torch.float32, device=dev)
            _lib.check(lib.stove_bg_dense(self._spn['ints'][2].data_ptr(), bc.data_ptr(), dense.data_ptr(), _lib.stream()), 'stove_bg_dense')
        return oc, ow, orr, bc, bw, dense

    def spn_sink(self, grads):
        """grads = (obj_coef, obj_wsum, obj_wroot, bg_coef, bg_wroot) table gradients -> accumulated into self.grad."""
        req = [p.requires_grad for p in self._spn['params']]
        if not any(req):
            return                                      # frozen SPNs
        if not all(req):
            raise RuntimeError('ParamArena: the SPN parameters must be frozen or trainable together')
        self.check()
        lib = _lib.load()
        g = _lib.SpnTableGrads()
        g.obj_coef, g.obj_wsum, g.obj_wroot, g.bg_coef, g.bg_wroot = [x.data_ptr() for x in grads]
        with torch.cuda.device(self.data.device):
            _lib.check(lib.stove_spn_bake_bwd(self.data.data_ptr(), ctypes.byref(self._spn['plan']), ctypes.byref(g),
                                              self.grad.data_ptr(), _lib.stream()), 'stove_spn_bake_bwd')

    # ------------------------------------------------------------------ GNN parameter image
    def _plan_gnn(self, dyn):
        dev = self.data.device

        def index_of(p):
            return (self.offset[id(p)] + torch.arange(p.numel(), dtype=torch.float64)).view(p.shape)
        self._gnn = {}
        for k in range(3):
            w, v, wt = dyn.param_image(k, leaf=index_of, pad_value=-1.0)
            pf, pt = ops.gnn_pack_perms(torch.device('cpu'))            # + the packed sections the small-graph kernels copy into LDS
            self._gnn[k] = (torch.cat([w, wt, v, w[pf], wt[pt]]).to(torch.int32).to(dev), torch.cat([w, v]).to(torch.int32).to(dev))
        self._gnn_params = [p for n, p in dyn.named_parameters()
                            if n.split('.')[0] in ('state_enc', 'self_cores', 'rel_cores', 'att_net', 'affector', 'out')]

    @property
    def has_gnn(self):
        return self._on_gpu and self._gnn is not None

    def gnn_image(self, core_idx=0):
        """[W | W^T | vectors] image of one core, gathered from the arena by one launch."""
        self.check()
        img = self._take(('gnn', core_idx))
        return img if img is not None else self._gather_gnn(core_idx)

    def _gather_gnn(self, core_idx):
        lib = _lib.load()
        src = self._gnn[core_idx][0]
        with torch.cuda.device(self.data.device):
            img = torch.empty(src.numel(), dtype=torch.float32, device=self.data.device)
            _lib.check(lib.stove_arena_gather(self.data.data_ptr(), src.data_ptr(), img.data_ptr(), src.numel(), _lib.stream()),
                       'stove_arena_gather')
        return img

    def gnn_sink(self, g, core_idx=0):
        """g = [dW | dvectors] gradient image -> accumulated into self.grad."""
        req = [p.requires_grad for p in self._gnn_params]
        if not any(req):
            return
        if not all(req):
            raise RuntimeError('ParamArena: the GNN core parameters must be frozen or trainable together')
        self.check()
        lib = _lib.load()
        src = self._gnn[core_idx][1]
        with torch.cuda.device(self.data.device):
            _lib.check(lib.stove_arena_scatter_add(g.data_ptr(), src.data_ptr(), self.grad.data_ptr(), src.numel(), _lib.stream()),
                       'stove_arena_scatter_add')
