"""torch.autograd bindings of the HIP kernels (libstove_hip.so, C ABI in include/stove_hip.h).

Every op runs on the current HIP stream of the tensors' device, allocates its outputs and
workspaces through PyTorch's caching allocator, and raises if the library is missing or the
tensors are not on a GPU -- there is no CPU or eager fallback.
"""
import ctypes
import math

import torch

from . import _lib
from . import settings as _settings
from ._lib import SpnTableGrads, SpnTables, check, ptr, stream


def _f32(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('stove_amd HIP ops need tensors on a GPU (cuda:N); there is no CPU path')
    if t.dtype != torch.float32:
        raise RuntimeError('stove_amd HIP kernels compute in float32; got %s' % t.dtype)
    return t.contiguous()


def _tables(obj=None, bg=None, bg_dense=None):
    """obj = (scope, leaf_slot, coef, wsum, wroot), bg = (side, coef, wroot) device tensors; bg_dense: the prebuilt operand
    image of the scene forward's leaf GEMM (stove_bg_dense) or None."""
    t = SpnTables()
    if obj is not None:
        t.obj_scope, t.obj_leaf_slot, t.obj_coef, t.obj_wsum, t.obj_wroot = [ptr(x) for x in obj]
    if bg is not None:
        t.bg_side, t.bg_coef, t.bg_wroot = [ptr(x) for x in bg]
    t.bg_dense = ptr(bg_dense) if bg_dense is not None else None
    return t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 4) // 4 + 1, dtype=torch.float32, device=device)


class _ObjSpnFn(torch.autograd.Function):
    """RatSpn.forward of the object SPN (reference rat_torch.py:354-357)."""

    @staticmethod
    def forward(ctx, inputs, marg, coef, wsum, wroot, scope, leaf_slot):
        lib = _lib.load()
        inputs, marg = _f32(inputs), _f32(marg)
        coef, wsum, wroot = _f32(coef), _f32(wsum), _f32(wroot)
        n = inputs.shape[0]
        dev = inputs.device
        with torch.cuda.device(dev):
            xw = torch.empty(lib.stove_objspn_tile_floats(n) + 1, dtype=torch.float32, device=dev)
            out = torch.empty(n, dtype=torch.float32, device=dev)
            t = _tables(obj=(scope, leaf_slot, coef, wsum, wroot))
            check(lib.stove_objspn_fwd(ctypes.byref(t), ptr(inputs), ptr(marg), ptr(xw), ptr(out), n, stream()),
                  'stove_objspn_fwd')
        ctx.save_for_backward(inputs, marg, coef, wsum, wroot, scope, leaf_slot, xw, out)
        ctx.has_marg = marg is not None
        return out.unsqueeze(1)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        inputs, marg, coef, wsum, wroot, scope, leaf_slot, xw, out = ctx.saved_tensors
        n = inputs.shape[0]
        dev = inputs.device
        dout = _f32(dout.reshape(-1))
        with torch.cuda.device(dev):
            need_x, need_m = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and marg is not None
            d_in = torch.empty_like(inputs) if need_x else None
            d_m = torch.empty_like(inputs) if need_m else None
            g_coef, g_wsum, g_wroot = torch.empty_like(coef), torch.empty_like(wsum), torch.empty_like(wroot)
            g = SpnTableGrads()
            g.obj_coef, g.obj_wsum, g.obj_wroot = ptr(g_coef), ptr(g_wsum), ptr(g_wroot)
            ws = _ws(lib.stove_objspn_bwd_ws_bytes(n), dev)
            t = _tables(obj=(scope, leaf_slot, coef, wsum, wroot))
            check(lib.stove_objspn_bwd(ctypes.byref(t), ptr(marg), ptr(xw), ptr(out), ptr(dout), ptr(d_in), ptr(d_m),
                                       ctypes.byref(g), ptr(ws), n, stream()), 'stove_objspn_bwd')
        return d_in, d_m, g_coef, g_wsum, g_wroot, None, None


class _BgSpnFn(torch.autograd.Function):
    """RatSpn.forward of the background SPN: the tuned 1024-dimension kernels (32 x 32 frames) or, for any other number of
    dimensions, the general-size ones (csrc/spn_bg_generic.hip)."""

    @staticmethod
    def forward(ctx, inputs, marg, coef, wroot, side):
        lib = _lib.load()
        inputs, marg, coef, wroot = _f32(inputs), _f32(marg), _f32(coef), _f32(wroot)
        n, D = inputs.shape[0], inputs.shape[1]
        if side.shape[-1] != D or coef.shape[1] != D:
            raise ValueError('background SPN tables are for %d dimensions, inputs have %d' % (side.shape[-1], D))
        dev = inputs.device
        with torch.cuda.device(dev):
            out = torch.empty(n, dtype=torch.float32, device=dev)
            t = _tables(bg=(side, coef, wroot))
            if D == 1024:
                ell = torch.empty(lib.stove_bgspn_saved_floats(n) + 1, dtype=torch.float32, device=dev)
                check(lib.stove_bgspn_fwd(ctypes.byref(t), ptr(inputs), ptr(marg), ptr(ell), ptr(out), n, stream()), 'stove_bgspn_fwd')
            else:
                ell = torch.empty(lib.stove_bgspn_saved_floats_d(n, D) + 1, dtype=torch.float32, device=dev)
                check(lib.stove_bgspn_fwd_d(ctypes.byref(t), ptr(inputs), ptr(marg), ptr(ell), ptr(out), n, D, stream()), 'stove_bgspn_fwd_d')
        ctx.save_for_backward(inputs, marg, coef, wroot, side, ell, out)
        return out.unsqueeze(1)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        inputs, marg, coef, wroot, side, ell, out = ctx.saved_tensors
        n, D = inputs.shape[0], inputs.shape[1]
        dev = inputs.device
        dout = _f32(dout.reshape(-1))
        with torch.cuda.device(dev):
            need_x, need_m = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and marg is not None
            d_in = torch.empty_like(inputs) if need_x else None
            d_m = torch.empty_like(inputs) if need_m else None
            g_coef, g_wroot = torch.empty_like(coef), torch.empty_like(wroot)
            g = SpnTableGrads()
            g.bg_coef, g.bg_wroot = ptr(g_coef), ptr(g_wroot)
            t = _tables(bg=(side, coef, wroot))
            if D == 1024:
                ws = _ws(lib.stove_bgspn_bwd_ws_bytes(n), dev)
                check(lib.stove_bgspn_bwd(ctypes.byref(t), ptr(inputs), ptr(marg), ptr(ell), ptr(out), ptr(dout), ptr(d_in),
                                          ptr(d_m), ctypes.byref(g), ptr(ws), n, stream()), 'stove_bgspn_bwd')
            else:
                ws = _ws(lib.stove_bgspn_bwd_ws_bytes_d(n, D), dev)
                check(lib.stove_bgspn_bwd_d(ctypes.byref(t), ptr(inputs), ptr(marg), ptr(ell), ptr(out), ptr(dout), ptr(d_in),
                                            ptr(d_m), ctypes.byref(g), ptr(ws), n, D, stream()), 'stove_bgspn_bwd_d')
        return d_in, d_m, g_coef, g_wroot, None


_SIDE_STREAMS = {}


def _side_stream(device, kind='side'):
    """Extra streams per device: 'side' for the work that only feeds the optimiser (the parameter-gradient chain of the backward
    pass), 'pre' for the parameter-only launches at the top of a step (table bake, GNN image gather)."""
    key = (str(device), kind)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class SideMode:
    """How the 'side' stream is driven.  Eager (default): torch's stream context, joined into the main stream when the backward
    pass is over.  `split` (set by graphed.GraphedTrainStep while it captures): the side stream is under its OWN capture -- its
    launches become a second graph that the replay launches on the side stream and joins before the optimiser's graph -- so
    torch's current stream stays the main one (allocations come from the main capture's pool) and only the library's launches
    are redirected; tensors allocated for side work are kept alive until the capture ends (`keep`), because a block freed
    inside a capture may be handed out again to main-stream work that runs concurrently with the side graph."""
    split = False
    keep = []


def _side_keep(*tensors):
    if SideMode.split and _lib._FORCED_STREAM is not None:
        SideMode.keep.extend(t for t in tensors if t is not None)


def run_on_side(dev, fn, bufs=(), after_main=True):
    """Enqueue the library launches of fn() on the side stream [after everything the main stream holds so far]; `bufs`: tensors
    they touch that main-stream code owns (the allocator must not recycle them before the side stream is done)."""
    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
    if after_main:
        check(_lib.load().stove_stream_after(side.cuda_stream, main.cuda_stream), 'stove_stream_after')
    if SideMode.split:
        prev = _lib.force_stream(side.cuda_stream)
        try:
            fn()
        finally:
            _lib.force_stream(prev)
    else:
        with torch.cuda.stream(side):
            fn()
    for b in bufs:
        if b is not None:
            b.record_stream(side)


def join_side_after_backward(dev):
    """The main stream waits for the side stream once, when the backward pass is over (eager mode; a split capture joins at replay)."""
    if SideMode.split:
        return
    torch.autograd.Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream(dev).wait_stream(_side_stream(dev)))


class _SceneFn(torch.autograd.Function):
    """Supair.likelihood fused (reference supair.py:44-110)."""

    @staticmethod
    def forward(ctx, frames, z, obj_coef, obj_wsum, obj_wroot, bg_coef, bg_wroot,
                obj_scope, obj_leaf_slot, bg_side, n_obj, beta, sink=None, bg_dense=None, with_grad=True, geom=None):
        # geom = (W, H, align_corners) for frames other than 32 x 32 / align_corners=False (stove_scene_fwd_any), else None
        lib = _lib.load()
        z = _f32(z)
        tabs = [_f32(x) for x in (obj_coef, obj_wsum, obj_wroot, bg_coef, bg_wroot)]
        # frames: (nf, 1024) dense, or a (n, T', 1024) time-slice of longer clips (x[:, 1:] in Stove.forward) whose rows are
        # contiguous: handed to the kernels as it lies in memory (frame map) instead of a 100 MB copy per step
        seq_frames = seq_stride = 0
        if frames.dim() == 3:
            if frames.stride(2) == 1 and frames.stride(1) == frames.shape[2] and frames.stride(0) >= frames.shape[1] * frames.shape[2] \
                    and frames.stride(0) % frames.shape[2] == 0 and frames.dtype == torch.float32 and frames.is_cuda:
                seq_frames, seq_stride = frames.shape[1], frames.stride(0) // frames.shape[2]
                nf = frames.shape[0] * frames.shape[1]
            else:
                frames = _f32(frames.reshape(-1, frames.shape[2]))
                nf = frames.shape[0]
        else:
            frames = _f32(frames)
            nf = frames.shape[0]
        ctx.frame_map = (nf, seq_frames, seq_stride)
        dev = frames.device
        with torch.cuda.device(dev):
            ll = torch.empty(nf, dtype=torch.float32, device=dev)
            parts = torch.empty(nf, 3, dtype=torch.float32, device=dev)
            # with a backward to come, the object SPN runs forward + backward (at unit upstream gradient) in one pass
            # (grad mode is off inside a Function's forward: the caller says whether it was on)
            grad = int(bool(with_grad) and any(ctx.needs_input_grad))
            t = _tables(obj=(obj_scope, obj_leaf_slot, tabs[0], tabs[1], tabs[2]), bg=(bg_side, tabs[3], tabs[4]), bg_dense=bg_dense)
            if geom is not None:
                W, H, ac = geom
                if frames.shape[-1] != W * H or tabs[3].numel() != 3 * W * H * 6 * 3:
                    raise ValueError('scene_likelihood: frames %s / background tables %s do not match a %d x %d frame'
                                     % (tuple(frames.shape), tuple(tabs[3].shape), W, H))
                saved = torch.empty(lib.stove_scene_saved_floats_any(nf, n_obj, W * H, grad) + 1, dtype=torch.float32, device=dev)
                check(lib.stove_scene_fwd_any(ctypes.byref(t), frames.data_ptr(), ptr(z), nf, n_obj, seq_frames, seq_stride, W, H, int(ac),
                                              float(beta), ptr(ll), ptr(parts), ptr(saved), stream(), grad), 'stove_scene_fwd_any')
            else:
                saved = torch.empty(lib.stove_scene_fwd_floats(nf, n_obj, grad) + 1, dtype=torch.float32, device=dev)
                check(lib.stove_scene_fwd_from(ctypes.byref(t), frames.data_ptr(), ptr(z), nf, n_obj, seq_frames, seq_stride, float(beta),
                                               ptr(ll), ptr(parts), ptr(saved), stream(), None, grad), 'stove_scene_fwd_from')
        ctx.save_for_backward(frames, z, *tabs, obj_scope, obj_leaf_slot, bg_side, saved)
        ctx.n_obj, ctx.beta, ctx.sink, ctx.geom = n_obj, float(beta), sink, geom
        ctx.set_materialize_grads(False)          # no zero tensors (one fill launch each) for the outputs nothing differentiates
        ctx.mark_non_differentiable(parts)
        return ll, parts

    @staticmethod
    def backward(ctx, dll, _dparts):
        lib = _lib.load()
        frames, z, oc, ow, orr, bc, bw, obj_scope, obj_leaf_slot, bg_side, saved = ctx.saved_tensors
        (nf, seq_frames, seq_stride), n_obj = ctx.frame_map, ctx.n_obj
        dev = frames.device
        if dll is None:
            return (None,) * 16
        dll = _f32(dll)
        with torch.cuda.device(dev):
            dz = torch.empty_like(z)
            grads = [torch.empty_like(x) for x in (oc, ow, orr, bc, bw)]
            g = SpnTableGrads()
            g.obj_coef, g.obj_wsum, g.obj_wroot, g.bg_coef, g.bg_wroot = [ptr(x) for x in grads]
            t = _tables(obj=(obj_scope, obj_leaf_slot, oc, ow, orr), bg=(bg_side, bc, bw))
            if ctx.geom is not None:
                W, H, ac = ctx.geom
                ws = _ws(lib.stove_scene_bwd_ws_bytes_any(nf, n_obj, W * H), dev)
                overlap = ctx.sink is not None and _settings.OVERLAP
                main, side = torch.cuda.current_stream(dev), (_side_stream(dev) if overlap else None)
                check(lib.stove_scene_bwd_any(ctypes.byref(t), frames.data_ptr(), ptr(z), nf, n_obj, seq_frames, seq_stride, W, H, int(ac),
                                              ctx.beta, ptr(saved), ptr(dll), ptr(dz), ctypes.byref(g), ptr(ws), main.cuda_stream,
                                              side.cuda_stream if overlap else None), 'stove_scene_bwd_any')
                if overlap:
                    run_on_side(dev, lambda: ctx.sink(grads), (ws, saved, *grads), after_main=False)     # ordered by the C call above
                    join_side_after_backward(dev)
                    return (None, dz) + (None,) * 14
                if ctx.sink is not None:
                    ctx.sink(grads)
                    grads = [None] * 5
                return (None, dz, *grads) + (None,) * 9
            ws = _ws(lib.stove_scene_bwd_ws_bytes(nf, n_obj), dev)
            if ctx.sink is not None and _settings.OVERLAP:
                # Flat-arena path: the table gradients only feed the optimiser.  Their passes (and the arena sink) go to a
                # second stream and overlap with what autograd enqueues next on this one: the recursion's backward,
                # latency-bound with one sequence per CU.  The main stream waits for them at the end of the backward pass.
                main, side = torch.cuda.current_stream(dev), _side_stream(dev)
                check(lib.stove_scene_bwd_overlap(ctypes.byref(t), frames.data_ptr(), ptr(z), nf, n_obj, seq_frames, seq_stride, ctx.beta,
                                                  ptr(saved), ptr(dll), ptr(dz), ctypes.byref(g), ptr(ws), main.cuda_stream,
                                                  side.cuda_stream),
                      'stove_scene_bwd_overlap')
                run_on_side(dev, lambda: ctx.sink(grads), (ws, saved, *grads), after_main=False)     # ordered by the C call above
                join_side_after_backward(dev)
                return (None, dz) + (None,) * 14
            check(lib.stove_scene_bwd(ctypes.byref(t), frames.data_ptr(), ptr(z), nf, n_obj, seq_frames, seq_stride, ctx.beta,
                                      ptr(saved), ptr(dll), ptr(dz), ctypes.byref(g), ptr(ws), stream()), 'stove_scene_bwd')
        if ctx.sink is not None:               # flat parameter arena: table gradients go straight into the bucket
            ctx.sink(grads)
            grads = [None] * 5
        return (None, dz, *grads) + (None,) * 9


class _ObjSpnAnyFn(torch.autograd.Function):
    """RatSpn.forward of the object SPN for any glimpse size / vector widths (csrc/spn_obj_generic.hip; reference
    rat_torch.py:354-357 with probabilistic_models.py:8-22 at non-default config.patch_* / obj_spn_num_*)."""

    @staticmethod
    def forward(ctx, inputs, marg, coef, wsum, wroot, lscope, slot, shape):
        lib = _lib.load()
        inputs, marg = _f32(inputs), _f32(marg)
        coef, wsum, wroot = _f32(coef), _f32(wsum), _f32(wroot)
        R, G, S, D, lmax = shape
        n, dev = inputs.shape[0], inputs.device
        if inputs.shape[1] != D or coef.numel() != R * 4 * lmax * G * 3 or wsum.numel() != R * 2 * G * G * S or wroot.numel() != R * S * S:
            raise ValueError('objspn_any: tables do not match the shape %s' % (shape,))
        with torch.cuda.device(dev):
            saved = torch.empty(lib.stove_objspn_saved_floats_any(n, R, G, S, D, lmax) + 1, dtype=torch.float32, device=dev)
            out = torch.empty(n, dtype=torch.float32, device=dev)
            check(lib.stove_objspn_fwd_any(ptr(inputs), ptr(marg), ptr(lscope), ptr(coef), ptr(wsum), ptr(wroot), ptr(saved), ptr(out),
                                           n, R, G, S, D, lmax, stream()), 'stove_objspn_fwd_any')
        ctx.save_for_backward(inputs, marg, coef, wsum, wroot, lscope, slot, saved)
        ctx.shape = shape
        return out.unsqueeze(1)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        inputs, marg, coef, wsum, wroot, lscope, slot, saved = ctx.saved_tensors
        R, G, S, D, lmax = ctx.shape
        n, dev = inputs.shape[0], inputs.device
        dout = _f32(dout.reshape(-1))
        with torch.cuda.device(dev):
            d_in = torch.empty_like(inputs) if ctx.needs_input_grad[0] else None
            d_marg = torch.empty_like(marg) if (marg is not None and ctx.needs_input_grad[1]) else None
            g_coef, g_wsum, g_wroot = torch.empty_like(coef), torch.empty_like(wsum), torch.empty_like(wroot)
            ws = _ws(lib.stove_objspn_bwd_ws_bytes_any(n, R, G, S, D, lmax), dev)
            check(lib.stove_objspn_bwd_any(ptr(inputs), ptr(marg), ptr(lscope), ptr(slot), ptr(coef), ptr(wsum), ptr(wroot), ptr(saved),
                                           ptr(dout), ptr(d_in), ptr(d_marg), ptr(g_coef), ptr(g_wsum), ptr(g_wroot), ptr(ws),
                                           n, R, G, S, D, lmax, stream()), 'stove_objspn_bwd_any')
        return d_in, d_marg, g_coef, g_wsum, g_wroot, None, None, None


class _GaussLlFn(torch.autograd.Function):
    """SimpleBG / SimpleObj.forward (reference probabilistic_models.py:42-90): rows of pixels under one fixed Normal."""

    @staticmethod
    def forward(ctx, x, marg, mean, scale):
        lib = _lib.load()
        x, marg = _f32(x), _f32(marg)
        n, d = x.shape
        with torch.cuda.device(x.device):
            out = torch.empty(n, dtype=torch.float32, device=x.device)
            check(lib.stove_gauss_ll_fwd(ptr(x), ptr(marg), ptr(out), n, d, float(mean), float(scale), stream()), 'stove_gauss_ll_fwd')
        ctx.save_for_backward(x, marg)
        ctx.ms = (float(mean), float(scale))
        return out.unsqueeze(-1)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, marg = ctx.saved_tensors
        n, d = x.shape
        dout = _f32(dout.reshape(-1))
        with torch.cuda.device(x.device):
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            dm = torch.empty_like(marg) if ctx.needs_input_grad[1] else None
            check(lib.stove_gauss_ll_bwd(ptr(x), ptr(marg), ptr(dout), ptr(dx), ptr(dm), n, d, ctx.ms[0], ctx.ms[1], stream()), 'stove_gauss_ll_bwd')
        return dx, dm, None, None


def gauss_ll(x, marg, mean, scale):
    return _GaussLlFn.apply(x, marg, mean, scale)


def objspn_any_apply(inputs, marg, coef, wsum, wroot, lscope, slot, shape):
    return _ObjSpnAnyFn.apply(inputs, marg, coef, wsum, wroot, lscope, slot, tuple(int(v) for v in shape))


def objspn_apply(inputs, marg, coef, wsum, wroot, scope, leaf_slot):
    return _ObjSpnFn.apply(inputs, marg, coef, wsum, wroot, scope, leaf_slot)


def bgspn_apply(inputs, marg, coef, wroot, side):
    return _BgSpnFn.apply(inputs, marg, coef, wroot, side)


@torch.no_grad()
def objspn_mpe(inputs, leaf_means, coef, wsum, wroot, scope, leaf_slot, return_pick=False):
    """inputs (n,100) -> MPE reconstructions (n,100) [, pick (n,5) int32 = replica + its 4 leaf components]."""
    lib = _lib.load()
    inputs, leaf_means, coef, wsum, wroot = _f32(inputs), _f32(leaf_means), _f32(coef), _f32(wsum), _f32(wroot)
    n, dev = inputs.shape[0], inputs.device
    with torch.cuda.device(dev):
        xw = torch.empty(lib.stove_objspn_tile_floats(n) + 1, dtype=torch.float32, device=dev)
        out = torch.empty(n, 100, dtype=torch.float32, device=dev)
        pick = torch.empty(n, 5, dtype=torch.int32, device=dev) if return_pick else None
        t = _tables(obj=(scope, leaf_slot, coef, wsum, wroot))
        check(lib.stove_objspn_mpe(ctypes.byref(t), ptr(leaf_means), ptr(inputs), ptr(xw), ptr(out), ptr(pick), n, stream()),
              'stove_objspn_mpe')
    return (out, pick) if return_pick else out


@torch.no_grad()
def render_frames(bg, patches, frames_per_patch, z, n_obj):
    """bg (1024,), patches (., 100), z (nf*n_obj, 4)=[sx,sy,x,y] -> (nf, 1024) = clamp(bg + pasted patches, 0, 1);
    patch row of (frame f, object k) = (f // frames_per_patch) * n_obj + k, or row 0 for all when frames_per_patch == 0."""
    lib = _lib.load()
    bg, patches, z = _f32(bg), _f32(patches), _f32(z)
    nf, dev = z.shape[0] // n_obj, z.device
    rows = patches.numel() // 100
    need = 1 if frames_per_patch == 0 else ((nf + frames_per_patch - 1) // frames_per_patch) * n_obj
    if bg.numel() != 1024 or rows < need or z.shape[-1] != 4:
        raise ValueError('render_frames: bad shapes bg %s patches %s z %s' % (tuple(bg.shape), tuple(patches.shape), tuple(z.shape)))
    with torch.cuda.device(dev):
        out = torch.empty(nf, 1024, dtype=torch.float32, device=dev)
        check(lib.stove_render_frames(ptr(bg), ptr(patches), int(frames_per_patch), ptr(z), ptr(out), nf, int(n_obj), stream()),
              'stove_render_frames')
    return out


def scene_likelihood(frames, z, obj_tabs, bg_tabs, n_obj, beta, sink=None, geom=None):
    """frames (nf,1024), z (nf*n_obj,4)=[sx,sy,x,y]; obj_tabs=(coef,wsum,wroot,scope,leaf_slot),
    bg_tabs=(coef,wroot,side[,dense]) -> ll (nf,), parts (nf,3)=(bg, patches, overlap).
    `sink(table_grads)`: receives the five table gradients in backward instead of autograd (ParamArena).
    `geom` = (W, H, align_corners): frames of W*H pixels / the other sampling convention (stove_scene_fwd_any); None = 32 x 32, False."""
    oc, ow, orr, osc, ols = obj_tabs
    bc, bw, bs = bg_tabs[:3]
    dense = bg_tabs[3] if len(bg_tabs) > 3 else None      # ParamArena: made with the bake, ahead of the scene chain
    if sink is not None and not z.requires_grad and torch.is_grad_enabled():
        z = z.detach().requires_grad_()        # the sink needs the backward to run
    return _SceneFn.apply(frames, z, oc, ow, orr, bc, bw, osc, ols, bs, int(n_obj), float(beta), sink, dense, torch.is_grad_enabled(),
                          geom)


def scene_glimpses(frames, z, n_obj):
    """patches_from_z + masks_from_z in one pass (no grad): -> patches (np,100), keep = 1-clamp(marg) (np,100)."""
    lib = _lib.load()
    frames, z = _f32(frames), _f32(z)
    nf = frames.shape[0]
    n_p = nf * n_obj
    dev = frames.device
    with torch.cuda.device(dev):
        tile = torch.empty(lib.stove_objspn_tile_floats(n_p) + 1, dtype=torch.float32, device=dev)
        patches = torch.empty(n_p, 100, dtype=torch.float32, device=dev)
        keep = torch.empty(n_p, 100, dtype=torch.float32, device=dev)
        check(lib.stove_scene_glimpses(ptr(frames), ptr(z), nf, n_obj, ptr(tile), ptr(patches), ptr(keep), stream()),
              'stove_scene_glimpses')
    return patches, keep


@torch.no_grad()
def glimpse_mean(x_color, z, n_obj):
    """x_color (nf, C, 32, 32), z (nf*n_obj, 4)=[sx,sy,x,y] -> (nf*n_obj, C): mean colour of every object's glimpse."""
    lib = _lib.load()
    x_color, z = _f32(x_color), _f32(z)
    nf, C = x_color.shape[0], x_color.shape[1]
    with torch.cuda.device(x_color.device):
        emb = torch.empty(nf * n_obj, C, dtype=torch.float32, device=x_color.device)
        check(lib.stove_glimpse_mean(ptr(x_color), ptr(z), ptr(emb), nf, n_obj, C, stream()), 'stove_glimpse_mean')
    return emb


class NoiseSource:
    """Standard-normal draws from the library's counter-based generator (csrc/state.hip noise_normal_k): the state -- [seed, call
    number] -- lives in device memory and is advanced on the device, so a captured draw replays with fresh noise and costs nothing
    on the device's serial chain.  It shadows torch's generator of the device: seed = its seed, call number = its Philox offset / 4,
    and every draw advances that offset by 4 on the host -- so torch.manual_seed (also with the same value again),
    torch.cuda.set_rng_state and per-rank seeding restart / restore this stream exactly as they do torch's own."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._seed = self._expected = None
        self.state = None
        self.captured = 0            # draws enqueued under stream capture since the capturing caller last cleared this

    def _gen(self):
        torch.cuda.init()
        return torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()]

    def prepare(self, n_draws=1):
        """Host side of n_draws draws that are about to be enqueued (eagerly, or by replaying a graph that holds them): if torch's
        generator was reseeded / restored since the last draw, the device state is rewritten from it (one 16-byte copy in stream
        order); then the generator's offset moves on as the device counter will."""
        gen = self._gen()
        seed, off = int(gen.initial_seed()), int(gen.get_offset())
        if self.state is None or seed != self._seed or off != self._expected:
            val = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed, off // 4], dtype=torch.int64)
            if self.state is None:
                self.state = val.to(self.device)
            else:
                self.state.copy_(val, non_blocking=False)
            self._seed = seed
        gen.set_offset(off + 4 * n_draws)
        self._expected = off + 4 * n_draws

    def normal(self, numel):
        """(numel,) fp32 draws, enqueued on the library's current stream (Stove runs it on the parameter stream ahead of its consumer)."""
        if not torch.cuda.is_current_stream_capturing():
            self.prepare(1)
        elif self.state is None:
            raise RuntimeError('NoiseSource: first use inside a stream capture (run one eager step first)')
        else:
            self.captured += 1
        out = torch.empty(int(numel), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(_lib.load().stove_noise_normal(ptr(out), int(numel), self.state.data_ptr(), stream()), 'stove_noise_normal')
        return out


def wave_sum_selftest(x):
    lib = _lib.load()
    x = _f32(x)
    out = torch.empty_like(x)
    check(lib.stove_selftest_wave_sum(ptr(x), ptr(out), x.numel() // 64, stream()), 'selftest')
    return out


# ------------------------------------------------------------------------------------------------
# GNN dynamics core and the fused inference recursion (csrc/gnn.hip)
# ------------------------------------------------------------------------------------------------
GNN_W_FLOATS = 22528
GNN_V_FLOATS = 672


# layers of the W image: (offset, OUT, K) as the forward uses them; the W^T image holds the same layers transposed (K, OUT swapped)
_GNN_LAYERS = ((0, 32, 32), (1024, 32, 32), (2048, 32, 32), (3072, 256, 32), (11264, 32, 64), (13312, 32, 64), (15360, 32, 32),
               (16384, 32, 32), (17408, 32, 32), (18432, 32, 32), (19456, 32, 64), (21504, 32, 32))
_GNN_PERMS = {}


def gnn_pack_perms(device):
    """Index tensors p_f, p_t with packed_W = W[p_f], packed_WT = WT[p_t]: every layer in the [K/4][OUT][4] order the small-graph
    kernels keep in LDS (csrc/gnn_small.hip: element (o, k) of an (OUT, K) layer at ((k >> 2) OUT + o) 4 + (k & 3))."""
    key = str(device)
    if key not in _GNN_PERMS:
        import numpy as np
        perms = []
        for transposed in (False, True):
            src = np.zeros(GNN_W_FLOATS, dtype=np.int64)
            for off, out_f, k_f in _GNN_LAYERS:
                OUT, K = (k_f, out_f) if transposed else (out_f, k_f)
                o, k = np.meshgrid(np.arange(OUT), np.arange(K), indexing='ij')
                src[off + ((k >> 2) * OUT + o) * 4 + (k & 3)] = off + o * K + k
            perms.append(torch.from_numpy(src).to(device))
        _GNN_PERMS[key] = tuple(perms)
    return _GNN_PERMS[key]


def _gnn_image(w_img, v_img, wt_img):
    if v_img is None:                          # prebuilt [W | W^T | vectors | W packed | W^T packed] image (ParamArena.gnn_image)
        need = _lib.load().stove_gnn_param_floats()
        if w_img.numel() != need:              # an image in an older layout would be read past its end by the kernels
            raise ValueError('GNN parameter image has %d floats, libstove_hip.so expects %d (stove_gnn_param_floats)' % (w_img.numel(), need))
        return w_img
    pf, pt = gnn_pack_perms(w_img.device)
    with torch.no_grad():
        packed = torch.cat([w_img.detach()[pf], wt_img.detach()[pt]])
    return torch.cat([w_img, wt_img, v_img, packed]).contiguous()


class _GnnStepFn(torch.autograd.Function):
    """Dynamics.forward core (reference dynamics.py:181-265): s_in (B,N,sin_dim) -> result, dynamic_pred (B,N,32)."""

    @staticmethod
    def forward(ctx, s_in, w_img, v_img, wt_img, lim_enc, elu, sink=None):
        lib = _lib.load()
        s_in = _f32(s_in)
        B, N, sd = s_in.shape
        dev = s_in.device
        params = _gnn_image(_f32(w_img), _f32(v_img), _f32(wt_img))
        ctx.sink = sink
        with torch.cuda.device(dev):
            res = torch.empty(B, N, 32, dtype=torch.float32, device=dev)
            pred = torch.empty(B, N, 32, dtype=torch.float32, device=dev)
            check(lib.stove_gnn_fwd(ptr(s_in), ptr(params), ptr(res), ptr(pred), B, N, sd, int(lim_enc), int(elu), stream()),
                  'stove_gnn_fwd')
        ctx.save_for_backward(s_in, params)
        ctx.cfg = (int(lim_enc), int(elu))
        return res, pred

    @staticmethod
    def backward(ctx, dres, dpred):
        lib = _lib.load()
        s_in, params = ctx.saved_tensors
        B, N, sd = s_in.shape
        dev = s_in.device
        lim_enc, elu = ctx.cfg
        with torch.cuda.device(dev):
            dres = _f32(dres) if dres is not None else torch.zeros(B, N, 32, device=dev)
            dpred = _f32(dpred) if dpred is not None else None
            d_s = torch.empty_like(s_in)
            g = torch.empty(lib.stove_gnn_grad_floats(), dtype=torch.float32, device=dev)
            ws = _ws(lib.stove_gnn_bwd_ws_bytes(B, N), dev)
            check(lib.stove_gnn_bwd(ptr(s_in), ptr(params), ptr(dres), ptr(dpred), ptr(d_s), ptr(g), ptr(ws), B, N, sd,
                                    lim_enc, elu, stream()), 'stove_gnn_bwd')
        if ctx.sink is not None:
            ctx.sink(g)
            return d_s, None, None, None, None, None, None
        return d_s, g[:GNN_W_FLOATS], g[GNN_W_FLOATS:], None, None, None, None


class _DynLoopFn(torch.autograd.Function):
    """The T-serial inference recursion of Stove.stove_forward in one persistent kernel."""

    @staticmethod
    def forward(ctx, z1, zsup, zsstd, eps, extra, w_img, v_img, wt_img, lim_enc, elu, consts, want_pred, sink=None):
        lib = _lib.load()
        z1, zsup, zsstd, eps = _f32(z1), _f32(zsup), _f32(zsstd), _f32(eps)
        extra = _f32(extra)
        B, Ts, N = zsup.shape[:3]
        sd = 16 + (extra.shape[-1] if extra is not None else 0)
        dev = z1.device
        params = _gnn_image(_f32(w_img), _f32(v_img), _f32(wt_img))
        with torch.cuda.device(dev):
            def out(d):
                return torch.empty(B, Ts, N, d, dtype=torch.float32, device=dev)
            z, zdyn, zdstd, mean, std = out(18), out(16), out(16), out(18), out(18)
            pred = out(32) if want_pred else None
            # saved activations (291 MB at B=256, T=100, N=3) so the backward does not recompute the forward
            act = None
            if any(ctx.needs_input_grad):
                act = torch.empty(lib.stove_dynloop_act_floats(B, Ts, N) + 1, dtype=torch.float32, device=dev)
            check(lib.stove_dynloop_fwd(ptr(z1), ptr(zsup), ptr(zsstd), ptr(eps), ptr(extra), ptr(params), ptr(z), ptr(zdyn),
                                        ptr(zdstd), ptr(mean), ptr(std), ptr(pred), ptr(act), B, Ts, N, sd, int(lim_enc),
                                        int(elu), *[float(c) for c in consts], stream()), 'stove_dynloop_fwd')
        ctx.save_for_backward(z1, zsup, zsstd, eps, extra, params, z, act)
        ctx.cfg = (int(lim_enc), int(elu), tuple(float(c) for c in consts), sd)
        ctx.sink = sink
        ctx.set_materialize_grads(False)          # no zero tensors (one fill launch each) for the outputs nothing differentiates
        ctx.mark_non_differentiable(zdstd)
        if pred is None:
            pred = z.new_zeros(0)
            ctx.mark_non_differentiable(pred)
        return z, zdyn, zdstd, mean, std, pred

    @staticmethod
    def backward(ctx, dz, dzdyn, _dzdstd, dmean, dstd, dpred):
        lib = _lib.load()
        z1, zsup, zsstd, eps, extra, params, z, act = ctx.saved_tensors
        lim_enc, elu, consts, sd = ctx.cfg
        B, Ts, N = zsup.shape[:3]
        dev = z1.device

        def up(g, shape_like):
            return None if g is None or g.numel() == 0 else _f32(g)
        with torch.cuda.device(dev):
            dz1 = torch.empty_like(z1)
            dzsup, dzsstd = torch.empty_like(zsup), torch.empty_like(zsstd)
            dextra = torch.empty_like(extra) if extra is not None else None
            g = torch.empty(lib.stove_gnn_grad_floats(), dtype=torch.float32, device=dev)
            ws = _ws(lib.stove_dynloop_bwd_ws_bytes_ts(B, Ts, N), dev)
            if ctx.sink is not None and _settings.OVERLAP:
                # weight gradients (only the optimiser reads them) on the second stream, under the encoder's backward GEMMs
                main, side = torch.cuda.current_stream(dev), _side_stream(dev)
                check(lib.stove_dynloop_bwd_overlap(ptr(z1), ptr(zsup), ptr(zsstd), ptr(eps), ptr(extra), ptr(params), ptr(z), ptr(act),
                                                    ptr(up(dz, z)), ptr(up(dzdyn, z)), ptr(up(dmean, z)), ptr(up(dstd, z)),
                                                    ptr(up(dpred, z)), ptr(dz1), ptr(dzsup), ptr(dzsstd), ptr(dextra), ptr(g), ptr(ws),
                                                    B, Ts, N, sd, lim_enc, elu, *consts, main.cuda_stream, side.cuda_stream),
                      'stove_dynloop_bwd_overlap')
                run_on_side(dev, lambda: ctx.sink(g), (ws, g, act), after_main=False)
                join_side_after_backward(dev)
                return (dz1, dzsup, dzsstd, None, dextra) + (None,) * 8
            check(lib.stove_dynloop_bwd(ptr(z1), ptr(zsup), ptr(zsstd), ptr(eps), ptr(extra), ptr(params), ptr(z), ptr(act),
                                        ptr(up(dz, z)), ptr(up(dzdyn, z)), ptr(up(dmean, z)), ptr(up(dstd, z)),
                                        ptr(up(dpred, z)), ptr(dz1), ptr(dzsup), ptr(dzsstd), ptr(dextra), ptr(g), ptr(ws),
                                        B, Ts, N, sd, lim_enc, elu, *consts, stream()), 'stove_dynloop_bwd')
        if ctx.sink is not None:
            ctx.sink(g)
            return (dz1, dzsup, dzsstd, None, dextra) + (None,) * 8
        return (dz1, dzsup, dzsstd, None, dextra, g[:GNN_W_FLOATS], g[GNN_W_FLOATS:], None, None, None, None, None, None)


def _sunk(t, sink):
    """With a gradient sink the parameters are not autograd inputs: make sure the backward still runs."""
    if sink is not None and not t.requires_grad and torch.is_grad_enabled():
        return t.detach().requires_grad_()
    return t


def gnn_step(s_in, image, lim_enc=2, elu=False, sink=None):
    """image = (w_img, v_img, wt_img) from Dynamics.param_image(), or (flat image, None, None) + a gradient sink."""
    return _GnnStepFn.apply(_sunk(s_in, sink), image[0], image[1], image[2], lim_enc, elu, sink)


def dyn_loop(z1, zsup, zsstd, eps, extra, image, lim_enc, elu, consts, want_pred=False, sink=None):
    return _DynLoopFn.apply(_sunk(z1, sink), zsup, zsstd, eps, extra, image[0], image[1], image[2], lim_enc, elu, consts,
                            want_pred, sink)


@torch.no_grad()
def rollout(z_last, extra, image, num, lim_enc, elu, consts, want_std=False, want_pred=False):
    """Generative rollout (forward only): z_last (B,N,18), extra (B,A,N,E) or None -> z_pred (B,num,N,18)."""
    lib = _lib.load()
    z_last, extra = _f32(z_last), _f32(extra)
    B, N = z_last.shape[:2]
    A = extra.shape[1] if extra is not None else 1
    sd = 16 + (extra.shape[-1] if extra is not None else 0)
    dev = z_last.device
    params = _gnn_image(_f32(image[0]), _f32(image[1]), _f32(image[2]))
    with torch.cuda.device(dev):
        z_pred = torch.empty(B, num, N, 18, dtype=torch.float32, device=dev)
        zstd = torch.empty(B, num, N, 16, dtype=torch.float32, device=dev) if want_std else None
        pred = torch.empty(B, num, N, 32, dtype=torch.float32, device=dev) if want_pred else None
        check(lib.stove_rollout_fwd(ptr(z_last), ptr(extra), ptr(params), ptr(z_pred), ptr(zstd), ptr(pred), B, num, A, N, sd,
                                    int(lim_enc), int(elu), *[float(c) for c in consts], stream()), 'stove_rollout_fwd')
    return z_pred, zstd, pred


MATCH_MODES = {'3_only': 0, 'greedy': 1, 'volatile': 2, '3_only_serial': 3}


@torch.no_grad()
def match_objects(feat, mode):
    """feat (B,T,N,F) matching features -> idx (B,T,N) int64 [, perm (B,T,N,N) for 'volatile']."""
    lib = _lib.load()
    feat = _f32(feat.detach())
    B, T, N, Fd = feat.shape
    dev = feat.device
    with torch.cuda.device(dev):
        idx = torch.empty(B, T, N, dtype=torch.int64, device=dev)
        perm = None
        check(lib.stove_match_objects(ptr(feat), ptr(idx), ptr(perm), B, T, N, Fd, MATCH_MODES[mode], stream()),
              'stove_match_objects')
    return idx, perm


# ------------------------------------------------------------------------------------------------
# SuPAIR state pipeline and ELBO assembly (csrc/state.hip)
# ------------------------------------------------------------------------------------------------
def _host_floats(vals, n):
    vals = [float(v) for v in vals]
    if len(vals) != n:
        raise ValueError('expected %d floats, got %d' % (n, len(vals)))
    return (ctypes.c_float * n)(*vals)


class _SupairStateFn(torch.autograd.Function):
    """constrain_zp + object matching + gather + fix_supair + velocities (reference supair.py:112-149,
    stove.py:172-198, 200-563): raw codes -> zfix (n,T,o,8), zl, sl (n,T-skip,o,6), init6 (n,o,6), idx."""

    @staticmethod
    def forward(ctx, codes, span_low, n, T, o, skip, fix, mode, lat_noise=None):
        lib = _lib.load()
        codes = _f32(codes)
        dev = codes.device
        Ts = T - skip
        kc = _host_floats(span_low, 16)
        lat_dim = 0
        if lat_noise is not None:               # the recursion's full initial state [six SuPAIR values | 0.01 x noise] from this launch
            lat_noise = _f32(lat_noise).reshape(n, o, -1)
            lat_dim = lat_noise.shape[-1]
        ld = 6 + lat_dim
        with torch.cuda.device(dev):
            def f(*shape):
                return torch.empty(*shape, dtype=torch.float32, device=dev)
            zc, pos, zfix = f(n, T, o, 8), f(n, T, o, 2), f(n, T, o, 8)
            idx = torch.empty(n, T, o, dtype=torch.int64, device=dev)
            hits = torch.empty(n, T, o, dtype=torch.uint8, device=dev)
            zl, sl, init6 = f(n, Ts, o, 6), f(n, Ts, o, 6), f(n, o, ld)
            check(lib.stove_supair_state_fwd2(ptr(codes), kc, ptr(zc), ptr(pos), ptr(idx), ptr(zfix), ptr(hits), ptr(zl), ptr(sl),
                                              ptr(init6), ld, ptr(lat_noise) if lat_noise is not None else None, lat_dim, n, T, o, skip,
                                              int(bool(fix)), MATCH_MODES[mode], stream()), 'stove_supair_state_fwd2')
        ctx.save_for_backward(zc, idx, hits, zfix)
        ctx.init_ld = ld
        ctx.cfg = (tuple(span_low), n, T, o, skip, codes.shape)
        ctx.set_materialize_grads(False)          # no zero tensors (one fill launch each) for the outputs nothing differentiates
        ctx.mark_non_differentiable(idx)
        return zfix, zl, sl, init6, idx

    @staticmethod
    def backward(ctx, g_zfix, g_zl, g_sl, g_init6, _g_idx):
        lib = _lib.load()
        zc, idx, hits, zfix = ctx.saved_tensors
        span_low, n, T, o, skip, shape = ctx.cfg
        dev = zc.device
        kc = _host_floats(span_low, 16)
        gs = [None if g is None else _f32(g) for g in (g_zfix, g_zl, g_sl, g_init6)]
        with torch.cuda.device(dev):
            ws = torch.empty(n * T * o * 8, dtype=torch.float32, device=dev)
            g_codes = torch.empty(shape, dtype=torch.float32, device=dev)
            check(lib.stove_supair_state_bwd2(ptr(zc), ptr(idx), ptr(hits), ptr(zfix), ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]),
                                              ctx.init_ld, kc, ptr(ws), ptr(g_codes), n, T, o, skip, stream()), 'stove_supair_state_bwd2')
        return g_codes, None, None, None, None, None, None, None, None


class _ZallFn(torch.autograd.Function):
    """z of the scene likelihood for frames 1..T-1 (reference stove.py:731-736 + sy_from_quotient) -> (n*(T-1)*o, 4), and zs handed
    through: the caller gives the SECOND output to the other consumers of zs (the ELBO terms), so that their gradient arrives here
    and is added by the backward kernel instead of by an accumulation launch of its own between this backward and the recursion's."""

    @staticmethod
    def forward(ctx, zfix, zs, n, T, o, skip):
        lib = _lib.load()
        zfix, zs = _f32(zfix), _f32(zs)
        dev = zfix.device
        with torch.cuda.device(dev):
            zall = torch.empty(n * (T - 1) * o, 4, dtype=torch.float32, device=dev)
            check(lib.stove_zall_fwd(ptr(zfix), ptr(zs), ptr(zall), n, T, o, skip, stream()), 'stove_zall_fwd')
        ctx.save_for_backward(zfix, zs)
        ctx.cfg = (n, T, o, skip)
        ctx.set_materialize_grads(False)
        return zall, zs.view_as(zs)

    @staticmethod
    def backward(ctx, g_zall, g_pass):
        lib = _lib.load()
        zfix, zs = ctx.saved_tensors
        n, T, o, skip = ctx.cfg
        if g_zall is None:
            return None, g_pass, None, None, None, None
        with torch.cuda.device(zfix.device):
            g_zfix, g_zs = torch.empty_like(zfix), torch.empty_like(zs)
            check(lib.stove_zall_bwd(ptr(zfix), ptr(zs), ptr(_f32(g_zall)), ptr(_f32(g_pass)) if g_pass is not None else None, ptr(g_zfix), ptr(g_zs),
                                     n, T, o, skip, stream()), 'stove_zall_bwd')
        return g_zfix, g_zs, None, None, None, None


class _ElboFn(torch.autograd.Function):
    """mean(trans_lik + img_lik - log_q) + mean(img_lik_sup) (reference stove.py:738-748) -> elbo (), stats (2,) =
    (mean trans_lik, mean log_q)."""

    @staticmethod
    def forward(ctx, zs, mean, std, zdyn, lik, trans_std, n, T, o, skip):
        lib = _lib.load()
        zs, mean, std, zdyn, lik = _f32(zs), _f32(mean), _f32(std), _f32(zdyn), _f32(lik)
        dev = zs.device
        ts = _host_floats(trans_std, 16)
        with torch.cuda.device(dev):
            part = torch.empty(n * 4, dtype=torch.float32, device=dev)
            out3 = torch.empty(3, dtype=torch.float32, device=dev)
            check(lib.stove_elbo_fwd(ptr(zs), ptr(mean), ptr(std), ptr(zdyn), ptr(lik), ts, ptr(part), ptr(out3), n, T, o, skip, stream()),
                  'stove_elbo_fwd')
        ctx.save_for_backward(zs, mean, std, zdyn)
        ctx.cfg = (tuple(trans_std), n, T, o, skip, lik.shape)
        stats = out3[1:]
        ctx.set_materialize_grads(False)          # no zero tensors (one fill launch each) for the outputs nothing differentiates
        ctx.mark_non_differentiable(stats)
        return out3[0], stats

    @staticmethod
    def backward(ctx, g_elbo, _g_stats):
        lib = _lib.load()
        zs, mean, std, zdyn = ctx.saved_tensors
        trans_std, n, T, o, skip, lik_shape = ctx.cfg
        ts = _host_floats(trans_std, 16)
        dev = zs.device
        if g_elbo is None:
            return (None,) * 10
        with torch.cuda.device(dev):
            g = _f32(g_elbo).reshape(1)
            g_zs, g_mean, g_std, g_zdyn = (torch.empty_like(t) for t in (zs, mean, std, zdyn))
            g_lik = torch.empty(lik_shape, dtype=torch.float32, device=dev)
            check(lib.stove_elbo_bwd(ptr(zs), ptr(mean), ptr(std), ptr(zdyn), ts, ptr(g), ptr(g_zs), ptr(g_mean), ptr(g_std), ptr(g_zdyn),
                                     ptr(g_lik), n, T, o, skip, stream()), 'stove_elbo_bwd')
        return g_zs, g_mean, g_std, g_zdyn, g_lik, None, None, None, None, None


def supair_state(codes, span_low, n, T, o, skip, fix, mode, lat_noise=None):
    """lat_noise (n, o, L) standard-normal draws: the fourth output is then the recursion's whole initial state (n, o, 6 + L) =
    [z_sup_full[:, skip-1] | 0.01 lat_noise] instead of the six SuPAIR values alone."""
    return _SupairStateFn.apply(codes, tuple(float(v) for v in span_low), int(n), int(T), int(o), int(skip), bool(fix), mode, lat_noise)


def zall(zfix, zs, n, T, o, skip):
    """-> (z_all, zs handed through): use the second value wherever zs is differentiated next (see _ZallFn)."""
    return _ZallFn.apply(zfix, zs, int(n), int(T), int(o), int(skip))


def elbo(zs, mean, std, zdyn, lik, trans_std, n, T, o, skip):
    return _ElboFn.apply(zs, mean, std, zdyn, lik, tuple(float(v) for v in trans_std), int(n), int(T), int(o), int(skip))


@torch.no_grad()
def bw_transform(x):
    """(n, T, C, w, h) frames -> (n, T, 1, w, h): channel sum clamped to [0, 1], one pass (reads the channels once).
    fp32 frames, or uint8 frames holding round(255 v) (the 8-bit device frame store): then each channel is divided by 255 first."""
    lib = _lib.load()
    if not x.is_cuda:
        raise RuntimeError('stove_amd HIP ops need tensors on a GPU (cuda:N); there is no CPU path')
    u8 = x.dtype == torch.uint8
    x = x.contiguous() if u8 else _f32(x)
    n, T, C, w, h = x.shape
    with torch.cuda.device(x.device):
        out = torch.empty(n, T, 1, w, h, dtype=torch.float32, device=x.device)
        if u8:
            check(lib.stove_bw_transform_u8(ptr(x), ptr(out), n * T, C, w * h, stream()), 'stove_bw_transform_u8')
        else:
            check(lib.stove_bw_transform(ptr(x), ptr(out), n * T, C, w * h, stream()), 'stove_bw_transform')
    return out


def colsum(a):
    """a.sum(0) of a contiguous (rows, cols) fp32 matrix (cols % 4 == 0) at HBM speed, fixed order."""
    lib = _lib.load()
    rows, cols = a.shape
    if cols % 4 != 0 and cols > 64:
        return a.sum(0)
    with torch.cuda.device(a.device):
        out = torch.empty(cols, dtype=torch.float32, device=a.device)
        ws = torch.empty(lib.stove_colsum_ws_floats(rows, cols) + 1, dtype=torch.float32, device=a.device)
        check(lib.stove_colsum(ptr(a), ptr(out), ptr(ws), rows, cols, stream()), 'stove_colsum')
    return out


def small_tn(a, b):
    """a^T @ b for a (rows, M), b (rows, N) with M * N <= 256 (csrc/arena.hip small_tn_part_k), fixed order."""
    lib = _lib.load()
    rows, M, N = a.shape[0], a.shape[1], b.shape[1]
    with torch.cuda.device(a.device):
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
        ws = torch.empty(lib.stove_small_tn_ws_floats(rows, M, N) + 1, dtype=torch.float32, device=a.device)
        check(lib.stove_small_tn(ptr(a), ptr(b), ptr(out), ptr(ws), rows, M, N, stream()), 'stove_small_tn')
    return out


class _LinearFn(torch.autograd.Function):
    """torch.nn.functional.linear on a 2-D fp32 input with the bias gradient as a chunked column sum: ATen reduces the
    (76 800, 50) head gradients of the recognition network at 0.07 TB/s (226 us per step), csrc/arena.hip takes ~10 us."""

    @staticmethod
    def _narrow(x, weight):
        return x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.shape[0] <= 64 and weight.shape[1] <= 64

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        if _LinearFn._narrow(x, weight):          # narrow layers (the action embedding): one thread per output, no library GEMM
            lib = _lib.load()
            y = torch.empty(x.shape[0], weight.shape[0], dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                check(lib.stove_small_linear(ptr(_f32(x)), ptr(_f32(weight)), ptr(_f32(bias)), ptr(y), x.shape[0], weight.shape[1], weight.shape[0], 0,
                                             stream()), 'stove_small_linear')
            return y
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            if _LinearFn._narrow(x, weight):
                lib = _lib.load()
                gx = torch.empty_like(x)
                with torch.cuda.device(x.device):      # dx = g W: the same kernel with W read as (in = out_features, out = in_features)
                    check(lib.stove_small_linear(ptr(_f32(g)), ptr(_f32(weight)), None, ptr(gx), g.shape[0], weight.shape[0], weight.shape[1], 1,
                                                 stream()), 'stove_small_linear')
            else:
                gx = torch.mm(g, weight)
        gw = None
        if ctx.needs_input_grad[1]:
            if g.shape[1] * x.shape[1] <= 256 and g.shape[0] >= 2048 and g.dtype == torch.float32 and x.dtype == torch.float32 and g.is_cuda:
                gw = small_tn(g, x.contiguous())                              # narrow layer over many rows: one thread per output
            else:
                gw = _splitk_tn(g, x)                                         # (out, in) = g^T x over all rows: split-K
        gb = colsum(g) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


class _RewardHeadFn(torch.autograd.Function):
    """reward = sigmoid(head1(sum_objects head0(pred))) of the action-conditioned model (reference dynamics.py:254-263) as one kernel
    each way (csrc/reward_head.hip).  pred (items, o, 32); the ten parameter tensors of the two heads in module order."""

    @staticmethod
    def forward(ctx, pred, *params):
        lib = _lib.load()
        pred = _f32(pred)
        items, o = pred.shape[0], pred.shape[1]
        dev = pred.device
        flat = torch.cat([_f32(p).reshape(-1) for p in params])
        if flat.numel() != lib.stove_reward_head_param_floats():
            raise RuntimeError('reward head: unexpected parameter shapes')
        with torch.cuda.device(dev):
            reward = torch.empty(items, dtype=torch.float32, device=dev)
            saved = torch.empty(lib.stove_reward_head_saved_floats(items, o) + 1, dtype=torch.float32, device=dev)
            check(lib.stove_reward_head_fwd(ptr(pred), ptr(flat), ptr(reward), ptr(saved), items, o, stream()), 'stove_reward_head_fwd')
        ctx.save_for_backward(pred, flat, reward, saved)
        ctx.shapes = [tuple(p.shape) for p in params]
        return reward

    @staticmethod
    def backward(ctx, d_reward):
        lib = _lib.load()
        pred, flat, reward, saved = ctx.saved_tensors
        items, o = pred.shape[0], pred.shape[1]
        dev = pred.device
        with torch.cuda.device(dev):
            d_pred = torch.empty_like(pred)
            g = torch.empty_like(flat)
            ws = torch.empty(lib.stove_reward_head_bwd_ws_floats(items) + 1, dtype=torch.float32, device=dev)
            check(lib.stove_reward_head_bwd(ptr(pred), ptr(flat), ptr(reward), ptr(saved), ptr(_f32(d_reward)), ptr(d_pred), ptr(g), ptr(ws),
                                            items, o, stream()), 'stove_reward_head_bwd')
        grads, off = [], 0
        for shp in ctx.shapes:
            n = 1
            for d in shp:
                n *= d
            grads.append(g[off:off + n].view(shp))
            off += n
        return (d_pred, *grads)


def reward_head(pred, params):
    """pred (..., o, 32) -> reward (..., 1) in (0, 1); params: head0.0.weight, head0.0.bias, head0.2.*, head1.0.*, head1.2.*, head1.4.*."""
    shape = pred.shape
    r = _RewardHeadFn.apply(pred.reshape(-1, shape[-2], shape[-1]), *params)
    return r.view(*shape[:-2], 1)


def linear(x, weight, bias):
    """F.linear for (..., in) fp32 GPU inputs with a fast bias gradient."""
    shape = x.shape
    out = _LinearFn.apply(_f32(x.reshape(-1, shape[-1])), weight, bias)
    return out.view(*shape[:-1], weight.shape[0])


# ------------------------------------------------------------------------------------------------
# Gradients straight into a flat arena.  A ParamArena registers the gradient view of every parameter it owns (keyed by the
# parameter's storage address); the recognition network's backward passes then ADD their weight gradients into those views
# inside the kernels that produce them (split-K sums, column sums, the head's reduction) and hand autograd None for them.
# That removes one AccumulateGrad `add_` launch per parameter from the critical path (8 per step) and lets the weight-gradient
# GEMMs run on the second stream, since nothing on the main stream consumes their result before the optimiser.
# The semantics are those of .backward() (accumulate into .grad); torch.autograd.grad() on arena-bound parameters must
# switch it off (ops.DIRECT_GRADS = False).
# ------------------------------------------------------------------------------------------------
DIRECT_GRADS = True       # tests flip it to compare with autograd's AccumulateGrad path (bit-identical gradients)
_GRAD_VIEWS = {}            # id(parameter) -> (weak reference to the parameter, its gradient view): dies with the parameter


def register_grad_view(p, g):
    """ParamArena: g is the flat-gradient view of parameter p (None: forget it).  Keyed on the parameter OBJECT (not its
    address, which a later tensor may reuse): the entry goes away with the parameter or when a new arena rebinds it."""
    import weakref
    key = id(p)
    if g is None:
        _GRAD_VIEWS.pop(key, None)
        return

    def _gone(ref, key=key):
        ent = _GRAD_VIEWS.get(key)
        if ent is not None and ent[0] is ref:
            del _GRAD_VIEWS[key]
    _GRAD_VIEWS[key] = (weakref.ref(p, _gone), g)


def _grad_views(*params, needs=None):
    """The registered gradient views of all of `params`, or None if direct accumulation is off or any of them is not bound
    to a live view: the view must belong to this very parameter object, still BE its `.grad` (an arena that was replaced,
    an optimizer.zero_grad(set_to_none) or a plain Adam detach it) and the parameter must want a gradient (`needs`: the
    ctx.needs_input_grad flags of the parameters; torch.autograd.grad() with other inputs leaves them False).  None sends the
    gradients through autograd as usual."""
    if not DIRECT_GRADS:
        return None
    if needs is not None and not all(needs):
        return None
    # .backward() accumulates into .grad of every leaf; torch.autograd.grad(outputs, inputs) wants the gradients of `inputs` RETURNED
    # and accumulates nothing.  The engine tells them apart: asked about a leaf's AccumulateGrad node it answers during .backward()
    # (True: the node will run) and refuses during autograd.grad() (RuntimeError) -> step aside and return the gradients.
    try:
        for p in params:
            if not torch._C._will_engine_execute_node(torch.autograd.graph.get_gradient_edge(p).node):
                return None
    except RuntimeError:
        return None
    except AttributeError:          # an older torch without the query: .backward() semantics assumed (ops.DIRECT_GRADS = False otherwise)
        pass
    out = []
    for p in params:
        ent = _GRAD_VIEWS.get(id(p))
        if ent is None or ent[0]() is not p or not p.requires_grad:
            return None
        g = ent[1]
        if p.grad is None or p.grad.data_ptr() != g.data_ptr() or g.shape != p.shape or g.dtype != torch.float32 \
                or not g.is_contiguous() or g.device != p.device:
            return None
        out.append(g)
    return out


class _EncoderHeadFusedFn(torch.autograd.Function):
    """fc2(sigmoid(fc1(h))) of the recognition network (reference encoder.py:53-56) as ONE kernel each way (csrc/head_fused.hip):
    fc1, sigmoid and fc2 chained on the matrix cores, h read once; the backward returns dL/dh and all four parameter gradients.
    H = 256, HID <= 64, OUT = 8.  split: fc1's forward as three half-piece MFMAs per product (default) or on the fp32 MFMA."""

    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2, frames, split=True):
        lib = _lib.load()
        rows, H, HID, OUT = h.shape[0], h.shape[1], w1.shape[0], w2.shape[0]
        ctx.frames = int(frames)
        h1 = torch.empty(rows, HID, dtype=torch.float32, device=h.device)
        codes = torch.empty(rows, OUT, dtype=torch.float32, device=h.device)
        with torch.cuda.device(h.device):
            check(lib.stove_enc_head_fwd(ptr(h), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(h1), ptr(codes), rows, H, HID, OUT, ctx.frames,
                                         int(bool(split)), stream()), 'stove_enc_head_fwd')
        ctx.save_for_backward(h, w1, b1, w2, b2, h1)
        return codes

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        h, w1, b1, w2, b2, h1 = ctx.saved_tensors
        rows, H, HID, OUT = h.shape[0], h.shape[1], w1.shape[0], w2.shape[0]
        g = _f32(g)
        dev = h.device
        gh = torch.empty_like(h)
        views = _grad_views(w1, b1, w2, b2, needs=ctx.needs_input_grad[1:5])
        outs = views if views is not None else [torch.empty_like(t) for t in (w1, b1, w2, b2)]
        with torch.cuda.device(dev):
            ws = torch.empty(lib.stove_enc_head_bwd_ws_floats(rows, HID), dtype=torch.float32, device=dev)
            check(lib.stove_enc_head_bwd(ptr(g), ptr(h1), ptr(h), ptr(w1), ptr(w2), ptr(gh), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]),
                                         ptr(outs[3]), int(views is not None), ptr(ws), rows, H, HID, OUT, ctx.frames, stream()), 'stove_enc_head_bwd')
        if views is not None:
            return gh, None, None, None, None, None, None
        return (gh, *outs, None, None)


def encoder_head(h, w1, b1, w2, b2, gemm='bf16x3', step_major=False):
    """(..., 256) -> (..., 8): fc2(sigmoid(fc1(h))).  gemm: 'fp32' runs fc1's forward on the fp32 MFMA instead of three half-piece
    MFMAs per product (see encoder_lstm).  step_major: h is (steps, n, 256) as the LSTM kernels write it and the result is
    (n, steps, 8) -- the kernels write the small output transposed instead of a permute + copy each way."""
    shape = h.shape
    fused = shape[-1] == 256 and w2.shape[0] == 8 and w1.shape[0] <= 64
    split = gemm != 'fp32'
    if step_major and h.dim() == 3 and fused:
        out = _EncoderHeadFusedFn.apply(_f32(h.reshape(-1, 256)), _f32(w1), _f32(b1), _f32(w2), _f32(b2), shape[1], split)
        return out.view(shape[1], shape[0], 8)
    if step_major:
        return encoder_head(h, w1, b1, w2, b2, gemm).transpose(0, 1)
    if not fused:                   # other head shapes than RnnStates' 256 -> <= 64 -> 8: two plain linear layers
        return linear(torch.sigmoid(linear(h, w1, b1)), w2, b2)
    out = _EncoderHeadFusedFn.apply(_f32(h.reshape(-1, shape[-1])), _f32(w1), _f32(b1), _f32(w2), _f32(b2), 0, split)
    return out.view(*shape[:-1], w2.shape[0])


def _splitk_tn(a, b):
    """a^T @ b for tall a (K, M), b (K, N) with K >> M, N (weight gradients over all frames).

    rocBLAS picks a low-occupancy kernel for this shape (40 TFLOP/s fp32 on MI355X for 1024 x 256 x 25 600);
    the same product as a batched GEMM over S chunks of K plus a sum runs at 120-140 TFLOP/s (tools/gemm_probe3.py)."""
    K = a.shape[0]
    S = 1
    for cand in (16, 8, 4, 2) if a.shape[1] * b.shape[1] <= (1 << 19) else ():      # a 1024 x 1024 output already fills the chip
        if K % cand == 0 and K // cand >= 1024:
            S = cand
            break
    if S == 1:
        return torch.mm(a.t(), b)
    parts = torch.bmm(a.view(S, K // S, -1).transpose(1, 2), b.view(S, K // S, -1))
    out = torch.empty(parts.shape[1:], dtype=parts.dtype, device=parts.device)
    if out.numel() % 4 != 0:
        return parts.sum(0)
    # ATen's reduce_kernel takes 225 us for this 16-way sum of 1 MB slabs; a plain chunk sum takes ~5 us
    check(_lib.load().stove_sum_chunks(ptr(parts), ptr(out), out.numel(), S, stream()), 'stove_sum_chunks')
    return out


class _blas:
    """Pick the BLAS backend per GEMM shape: on this stack hipBLASLt (torch's default) is the faster one for the weight
    gradients, rocBLAS for the NT products of the forward (x W_ih^T: 428 vs 480 us, h W_hh^T: 123 vs 150 us;
    tools/blas_probe.py).  Restores the previous preference on exit."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        try:
            self.prev = torch.backends.cuda.preferred_blas_library()
            torch.backends.cuda.preferred_blas_library(self.name)
        except Exception:           # backend not available in this build: keep the default
            self.prev = None

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.backends.cuda.preferred_blas_library(self.prev)
        return False


# rows below which the recurrent products of the recognition network leave the 256 x 256 tile, and the tile they take then
SMALL_M_ROWS = 4096
SMALL_M_TILE = 2

# gate activations of the LSTM cell kernels on v_exp_f32 / v_rcp_f32 (absolute error ~1e-7; 0: IEEE division + ocml tanhf)
FAST_CELL = 1


# Row chunks of the recognition network's forward chain (two: measured best, DESIGN.md section 3 "Streams"; 1 = the unchunked chain,
# which the bench's one-stream kernel profile and the equality test select)
ENC_CHUNKS = 2


def _enc_chunks(n, D, H):
    """Equal row chunks [r0, r1) of the recognition network's forward chain (every product and cell of it is row-wise independent), or
    None for the unchunked chain: the 256 x 256 tile on whole tiles of every chunk, chunks big enough to fill most of the chip."""
    c = ENC_CHUNKS
    if not (c > 1 and _settings.OVERLAP and n % 256 == 0 and n // c >= 4096 and (4 * H) % 256 == 0 and D % 32 == 0 and H % 32 == 0):
        return None
    tiles = n // 256
    bounds = [int(round(tiles * i / c)) * 256 for i in range(c)] + [n]
    if any(b1 - b0 < 2048 for b0, b1 in zip(bounds, bounds[1:])):
        return None
    return list(zip(bounds, bounds[1:]))


def _encoder_lstm_fwd_chunked(lib, x, w_ih, w_hh, bias, hs, cs, num_steps, ns, chunks):
    """The forward chain [x W_ih^T, then per step h W_hh^T + gx and the cell] over row chunks on two streams: while one chunk's
    cell kernel streams its gates through HBM the other chunk's product keeps the matrix cores busy, and the workgroups of two
    products fill the chip where one launch leaves a partial round of tiles.  Row-wise the same kernels on the same operands:
    bit-identical to the unchunked chain.  -> [gs_0 (= gx), gs_1, ...]"""
    n, D = x.shape
    H = w_hh.shape[1]
    dev = x.device
    main = torch.cuda.current_stream(dev)
    gss = [torch.empty(n, 4 * H, dtype=torch.float32, device=dev) for _ in range(num_steps)]

    def chain(r0, r1):
        rows = r1 - r0
        st = stream()
        check(lib.stove_gemm_bf16(ptr(x[r0:]), ptr(w_ih), ptr(bias), None, ptr(gss[0][r0:]), rows, 4 * H, D, x.stride(0), w_ih.stride(0), 4 * H,
                                  0, 0, ns, 1, 3, None, st), 'stove_gemm_bf16')
        for k in range(num_steps):
            if k > 0:
                check(lib.stove_gemm_bf16(ptr(hs[k - 1][r0:]), ptr(w_hh), None, ptr(gss[0][r0:]), ptr(gss[k][r0:]), rows, 4 * H, H, H, w_hh.stride(0),
                                          4 * H, 0, 0, ns, 1, 3, None, st), 'stove_gemm_bf16')
            check(lib.stove_lstm_cell_fwd(ptr(gss[k][r0:]), None, ptr(cs[k - 1][r0:]) if k > 0 else None, ptr(cs[k][r0:]), ptr(hs[k][r0:]),
                                          rows, H, FAST_CELL, st), 'stove_lstm_cell_fwd')

    # even chunks on the main stream, odd chunks on the 'enc' stream forked from it (more than two concurrently active capture
    # streams cost the graph replay more than they hide: DESIGN.md)
    enc = _side_stream(dev, 'enc')
    enc.wait_stream(main)
    for c, (r0, r1) in enumerate(chunks):
        if c % 2 == 0:
            chain(r0, r1)
        else:
            with torch.cuda.stream(enc):
                chain(r0, r1)
    main.wait_stream(enc)
    return gss


class _EncoderLstmFn(torch.autograd.Function):
    """num_steps LSTM steps on the SAME input x (reference encoder.py:43-51): hs (n, num_steps, H).

    MFMA GEMMs (csrc/gemm_bf16.hip; library fp32 GEMMs with gemm='fp32') + the fused gate kernels of csrc/lstm.hip.  The
    input projection x W_ih^T is computed once; its gradient is the sum of the per-step gate gradients (accumulated
    in-kernel), so dW_ih is ONE (4H x n) @ (n x D) GEMM."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, num_steps, time_major=False, gemm='bf16x3'):
        lib = _lib.load()
        x, w_ih, w_hh = _f32(x), _f32(w_ih), _f32(w_hh)
        n, H = x.shape[0], w_hh.shape[1]
        dev = x.device
        ns = {'bf16x3': 2, 'bf16': 1, 'fp32': 0}[gemm]
        if not gemm_ok(x.shape[1], H):
            ns = 0                               # odd sizes: library GEMMs
        ns_bwd = ns
        # the forward products' hi / lo pieces are IEEE halves (nsplit 3: 2^-22 of the value in two pieces instead of bf16's 2^-18;
        # frames, hidden states and weights sit inside half's range), the backward's stay bf16 (gradients need fp32's exponent range)
        ns = 3 if ns == 2 else ns
        with torch.cuda.device(dev):
            if ns and _enc_chunks(n, x.shape[1], H) is not None:
                gx = None               # the chunked chains below make their own rows of it
            elif ns:
                gx = _gemm_rows_balanced(x, w_ih, _f32(b_ih + b_hh), ns)
            else:
                with _blas('hipblas'):
                    gx = torch.addmm(b_ih + b_hh, x, w_ih.t())
            hs = torch.empty(num_steps, n, H, dtype=torch.float32, device=dev)
            cs = torch.empty(num_steps, n, H, dtype=torch.float32, device=dev)
            # gs[k]: gate pre-activations of step k = gx + h_{k-1} W_hh^T (the recurrent GEMM adds gx in its epilogue, so the
            # cell kernels read ONE (n, 4H) tensor per step, forward and backward); gs[0] is gx itself
            gss = []
            if gx is None:
                gss = _encoder_lstm_fwd_chunked(lib, x, w_ih, w_hh, _f32(b_ih + b_hh), hs, cs, num_steps, ns, _enc_chunks(n, x.shape[1], H))
                gx = gss[0]
            for k in range(num_steps if not gss else 0):
                gs = gx
                if k > 0 and ns:
                    # K = 256 (8 k-steps): the 128 x 128 tile (1600 workgroups) beats 256 x 128; the 256 x 256 tile is level with it
                    # warm and 5 % ahead from cold caches
                    # (fewer than 4 096 rows -- the reference's default training shape has 2 048 -- are 32 workgroups of the wide tile:
                    # the 128 x 128 tile gives every second CU one)
                    gs = gemm_bf16(hs[k - 1], w_hh, nsplit=ns, splitk=1, add=gx, tile=3 if ((4 * H) % 256 == 0 and n >= SMALL_M_ROWS) else SMALL_M_TILE)
                elif k > 0:
                    with _blas('hipblas'):
                        gs = torch.addmm(gx, hs[k - 1], w_hh.t())
                gss.append(gs)
                check(lib.stove_lstm_cell_fwd(ptr(gs), None, ptr(cs[k - 1]) if k > 0 else None, ptr(cs[k]), ptr(hs[k]),
                                              n, H, FAST_CELL, stream()), 'stove_lstm_cell_fwd')
        ctx.save_for_backward(x, w_ih, w_hh, gx, hs, cs, b_ih, b_hh, *gss[1:])
        ctx.num_steps = num_steps
        ctx.time_major = bool(time_major)
        ctx.ns = ns_bwd
        return hs if time_major else hs.transpose(0, 1)

    @staticmethod
    def backward(ctx, dhs):
        lib = _lib.load()
        x, w_ih, w_hh, gx, hs, cs, b_ih, b_hh = ctx.saved_tensors[:8]
        gss = [gx] + list(ctx.saved_tensors[8:])
        K = ctx.num_steps
        ns = ctx.ns
        n, H = x.shape[0], w_hh.shape[1]
        dev = x.device
        dhs = _f32(dhs if ctx.time_major else dhs.transpose(0, 1))                       # (K, n, H)
        # Parameter gradients straight into the arena's views (see _grad_views): no AccumulateGrad launches, and the two
        # weight-gradient GEMMs run on the second stream next to the rest of this backward -- only the optimiser reads them.
        views = _grad_views(w_ih, w_hh, b_ih, b_hh, needs=ctx.needs_input_grad[1:5]) if (ns and gemm_ok(n) and not ctx.needs_input_grad[0]) else None
        fork = views is not None and _settings.OVERLAP
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if fork else main

        def wgrad(dy, inp, out=None):
            """dy^T @ inp over all rows: both operands are stored K-major for this product."""
            if ns and gemm_ok(dy.shape[0]):
                # (the 256 x 256 tile for dW_ih is 45 us shorter alone and 11 us LONGER in the step: it holds every register of its CUs
                # and the bias sums on the second stream wait for it -- DESIGN.md)
                return gemm_bf16(dy, inp, None, True, True, ns, out=out)
            return _splitk_tn(dy, inp)

        def on_side(fn, *bufs):
            if fork:
                run_on_side(dev, fn, bufs)
            else:
                fn()
        with torch.cuda.device(dev):
            dgx = torch.empty_like(gx)
            # gate gradients of steps 1..K-1 are kept ((K-1) x 105 MB at 25 600 frames): dW_hh is ONE GEMM over all of
            # them, and step 0 -- the last to run, h_{-1} = 0 -- adds them to its own to form the gradient of the shared
            # input projection, dgx (its own gate gradient is stored nowhere else)
            dg_all = torch.empty(max(K - 1, 1), n, 4 * H, dtype=torch.float32, device=dev)
            dc = [torch.empty(n, H, dtype=torch.float32, device=dev) for _ in range(2)]
            dh = dhs[K - 1]
            d_whh = None
            for k in range(K - 1, -1, -1):
                dg = dg_all[k - 1] if k > 0 else None
                check(lib.stove_lstm_cell_bwd(ptr(gss[k]), None, ptr(cs[k - 1]) if k > 0 else None, ptr(cs[k]), ptr(dh),
                                              ptr(dc[(k + 1) % 2]) if k < K - 1 else None, ptr(dg) if dg is not None else None,
                                              ptr(dc[k % 2]), ptr(dgx) if k == 0 else None, ptr(dg_all) if (k == 0 and K > 1) else None,
                                              K - 1 if k == 0 else 0, n, H, FAST_CELL, stream()), 'stove_lstm_cell_bwd')
                if k == 1 and views is not None:
                    # all gate gradients W_hh sees are written: its GEMM starts here, beside the last recurrent step
                    on_side(lambda: wgrad(dg_all.view(-1, 4 * H), hs[:K - 1].view(-1, H), out=views[1]), dg_all, hs)
                if k > 0:
                    if ns:
                        dh = gemm_bf16(dg, w_hh, None, False, True, ns, 1, add=dhs[k - 1])     # dhs[k-1] + dg W_hh
                    else:
                        dh = torch.addmm(dhs[k - 1], dg, w_hh)
            if views is not None:
                # the bias sums join dW_hh on the second stream; dW_ih, the last and longest product, stays on this one, so the
                # step's final join finds the second stream long finished (a join the main stream has to WAIT at costs ~40 us
                # of inter-queue signalling on top of the work)
                def bias_sums():
                    ws = torch.empty(lib.stove_colsum_ws_floats(n, 4 * H) + 1, dtype=torch.float32, device=dev)
                    _side_keep(ws)
                    check(lib.stove_colsum2(ptr(dgx), ptr(views[2]), ptr(views[3]), 1, ptr(ws), n, 4 * H, stream()), 'stove_colsum2')
                on_side(bias_sums, dgx)
                wgrad(dgx, x, out=views[0])
                if fork:
                    join_side_after_backward(dev)
                return None, None, None, None, None, None, None, None
            d_whh = wgrad(dg_all.view(-1, 4 * H), hs[:K - 1].view(-1, H)) if K > 1 else None
            if d_whh is None:
                d_whh = torch.zeros_like(w_hh)
            d_wih = wgrad(dgx, x)
            d_b = colsum(dgx)
        dx = torch.mm(dgx, w_ih) if ctx.needs_input_grad[0] else None
        return dx, d_wih, d_whh, d_b, d_b, None, None, None


def encoder_lstm(x, w_ih, w_hh, b_ih, b_hh, num_steps, time_major=False, gemm='bf16x3'):
    """-> hs (n, num_steps, H), or (num_steps, n, H) with time_major=True: the layout the kernels produce; the row-wise
    head can run on it directly, which saves two 78 MB transposes per step (only its 8-wide output is permuted).
    gemm: 'bf16x3' = fp32 products as three sixteen-bit MFMAs on hi/lo-split operands (csrc/gemm_bf16.hip, split16.h): IEEE-half
    pieces in the forward products (2^-22 of the value per product), bf16 pieces in the gradient products (2^-18; fp32's exponent
    range); 'fp32' = library fp32 GEMMs, 'bf16' = plain bf16 operands with fp32 accumulation (the reported, never default, variant)."""
    return _EncoderLstmFn.apply(x, w_ih, w_hh, b_ih, b_hh, num_steps, time_major, gemm)


def gemm_splitk(M, N, K):
    """K slices that fill the chip when the output has few tiles (weight gradients: 1024 x 1024 over K = 25 600)."""
    tiles = ((M + 255) // 256) * ((N + 127) // 128)
    s = 1
    while tiles * s * 2 <= 256 and K // (s * 2) >= 256:
        s *= 2
    return s


def gemm_bf16(a, b, bias=None, a_kmajor=False, b_kmajor=False, nsplit=2, splitk=None, add=None, tile=0, out=None):
    """C (M, N) = sum_k a(m,k) b(n,k) (+ bias) (+ add) on the bf16 matrix cores with fp32 in / out (csrc/gemm_bf16.hip).
    a: (M, K), or (K, M) when a_kmajor; b: (N, K), or (K, N) when b_kmajor.  nsplit 2 = hi+lo bf16 pieces (3 MFMAs, 4.5e-6 of the largest
    entry), 1 = plain bf16, 3 = hi+lo IEEE-half pieces (3 MFMAs, 1-4e-7; row-major a and b only; operands must lie inside half's
    range: |x| < 65504, and entries below 2^-14 keep an absolute error of 2^-25 instead of a relative one).
    splitk None = chosen from the shape (only without bias / add).  out: a contiguous (M, N) tensor the product is ADDED to
    (a gradient view; returned)."""
    def rows_ok(t):          # a row-strided view (padded leading dimension) is taken as it lies
        return t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]
    a = a if rows_ok(a) else _f32(a)
    b = b if rows_ok(b) else _f32(b)
    M, K = (a.shape[1], a.shape[0]) if a_kmajor else a.shape
    N = b.shape[1] if b_kmajor else b.shape[0]
    if splitk is None:
        splitk = gemm_splitk(M, N, K) if bias is None and add is None else 1
    if (splitk == 1 and out is None and not a_kmajor and (bias is not None or add is not None) and K >= 512 and K % 256 == 0
            and M * N % 4 == 0):
        # Short activations x weights products (the reference's default training shape has 2 048 frames per step: 16-64 workgroups of
        # the tile, each walking all of K at ~1.5 us per k-step): split K so that every CU gets a workgroup.  The add term becomes
        # the initial value of C (split-K accumulates into C when add == C), a bias rides the slice sum (fewer than 16 slices).
        bm, bn = (128, 128) if tile == 2 else (256, 128)
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        sk = 1
        while tiles * sk * 2 <= 256 and K // (sk * 2) >= 128 and sk * 2 <= 8:
            sk *= 2
        if sk > 1 and not (bias is not None and add is not None):
            splitk = sk
            if add is not None and not (add.is_contiguous() and tuple(add.shape) == (M, N) and add.dtype == torch.float32 and add.data_ptr() % 16 == 0):
                out, add = _f32(add).clone(), None          # (a dense add term rides the slice sum; anything else becomes C's initial value)
    lib = _lib.load()
    if out is not None:
        if add is not None or bias is not None or out.shape != (M, N) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError('gemm_bf16: out= takes a contiguous fp32 (M, N) tensor and no bias / add')
        c = add = out
    else:
        c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        ws = torch.empty(lib.stove_gemm_bf16_ws_floats(M, N, splitk), dtype=torch.float32, device=a.device) if splitk > 1 else None
        _side_keep(ws, c)
        check(lib.stove_gemm_bf16(a.data_ptr(), b.data_ptr(), ptr(bias) if bias is not None else None, ptr(_f32(add)) if add is not None else None,
                                  ptr(c), M, N, K, a.stride(0), b.stride(0), N, int(a_kmajor), int(b_kmajor), nsplit, splitk, tile,
                                  ptr(ws) if ws is not None else None, stream()), 'stove_gemm_bf16')
    return c


def _gemm_rows_balanced(x, w, bias, ns):
    """x W^T + bias for a tall x on the 256 x 128 tile.  The tile count is rarely a multiple of the CU count (25 600 frames:
    800 tiles on 256 CUs = three full rounds and 32 tiles that keep an eighth of the chip busy for a fourth); the rows of the
    incomplete round go through a second launch that splits K eight ways instead (256 short workgroups)."""
    M, N = x.shape[0], w.shape[0]
    if N % 256 == 0 and M >= 4096 and x.shape[1] >= 512:
        # 256 x 256 workgroup tile (eight waves of 64 x 128): 2/3 of the operand bytes per flop through L2 and LDS, half the
        # barriers; 25 600 x 1024 x 1024 isolated 183 us against 229 (256 x 128) / 211 (128 x 128), profiles/r04_gemm_tiles.txt
        return gemm_bf16(x, w, bias=bias, nsplit=ns, splitk=1, tile=3)
    cus = torch.cuda.get_device_properties(x.device).multi_processor_count
    tiles_n = (N + 127) // 128
    tiles_m = (M + 255) // 256
    rounds = (tiles_m * tiles_n) // cus
    main_m = (rounds * cus) // tiles_n                     # M-tiles of the complete rounds
    tail_tiles = (tiles_m - main_m) * tiles_n
    K = x.shape[1]
    if rounds == 0 or tail_tiles == 0 or tail_tiles * 4 > cus or K % 256 != 0 or N % 4 != 0:
        return gemm_bf16(x, w, bias=bias, nsplit=ns, splitk=1)             # tile chosen from the shape (gemm_tile)
    lib = _lib.load()
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    rows = main_m * 256
    sk = 8
    with torch.cuda.device(x.device):
        check(lib.stove_gemm_bf16(ptr(x), ptr(w), ptr(bias), None, ptr(out), rows, N, K, x.stride(0), w.stride(0), N, 0, 0, ns, 1, 1, None,
                                  stream()), 'stove_gemm_bf16')
        xt, ot = x[rows:], out[rows:]
        ws = torch.empty(lib.stove_gemm_bf16_ws_floats(M - rows, N, sk), dtype=torch.float32, device=x.device)
        check(lib.stove_gemm_bf16(ptr(xt), ptr(w), ptr(bias), None, ptr(ot), M - rows, N, K, x.stride(0), w.stride(0), N, 0, 0, ns, sk, 1,
                                  ptr(ws), stream()), 'stove_gemm_bf16')
    return out


def gemm_ok(*dims):
    """Shapes the MFMA GEMM takes (float4 granularity); anything else goes to the library."""
    return all(int(d) % 4 == 0 and int(d) > 0 for d in dims)
