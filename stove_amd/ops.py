"""torch.autograd bindings of the HIP kernels (libstove_hip.so, C ABI in include/stove_hip.h).

Every op runs on the current HIP stream of the tensors' device, allocates its outputs and
workspaces through PyTorch's caching allocator, and raises if the library is missing or the
tensors are not on a GPU -- there is no CPU or eager fallback.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import SpnTableGrads, SpnTables, check, ptr, stream


def _f32(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError('stove_amd HIP kernels compute in float32; got %s' % t.dtype)
    return t.contiguous()


def _tables(obj=None, bg=None):
    """obj = (scope, leaf_slot, coef, wsum, wroot), bg = (side, coef, wroot) device tensors."""
    t = SpnTables()
    if obj is not None:
        t.obj_scope, t.obj_leaf_slot, t.obj_coef, t.obj_wsum, t.obj_wroot = [ptr(x) for x in obj]
    if bg is not None:
        t.bg_side, t.bg_coef, t.bg_wroot = [ptr(x) for x in bg]
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
    """RatSpn.forward of the background SPN."""

    @staticmethod
    def forward(ctx, inputs, marg, coef, wroot, side):
        lib = _lib.load()
        inputs, marg, coef, wroot = _f32(inputs), _f32(marg), _f32(coef), _f32(wroot)
        n = inputs.shape[0]
        dev = inputs.device
        with torch.cuda.device(dev):
            ell = torch.empty(lib.stove_bgspn_saved_floats(n) + 1, dtype=torch.float32, device=dev)
            out = torch.empty(n, dtype=torch.float32, device=dev)
            t = _tables(bg=(side, coef, wroot))
            check(lib.stove_bgspn_fwd(ctypes.byref(t), ptr(inputs), ptr(marg), ptr(ell), ptr(out), n, stream()),
                  'stove_bgspn_fwd')
        ctx.save_for_backward(inputs, marg, coef, wroot, side, ell, out)
        return out.unsqueeze(1)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        inputs, marg, coef, wroot, side, ell, out = ctx.saved_tensors
        n = inputs.shape[0]
        dev = inputs.device
        dout = _f32(dout.reshape(-1))
        with torch.cuda.device(dev):
            need_x, need_m = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and marg is not None
            d_in = torch.empty_like(inputs) if need_x else None
            d_m = torch.empty_like(inputs) if need_m else None
            g_coef, g_wroot = torch.empty_like(coef), torch.empty_like(wroot)
            g = SpnTableGrads()
            g.bg_coef, g.bg_wroot = ptr(g_coef), ptr(g_wroot)
            ws = _ws(lib.stove_bgspn_bwd_ws_bytes(n), dev)
            t = _tables(bg=(side, coef, wroot))
            check(lib.stove_bgspn_bwd(ctypes.byref(t), ptr(inputs), ptr(marg), ptr(ell), ptr(out), ptr(dout), ptr(d_in),
                                      ptr(d_m), ctypes.byref(g), ptr(ws), n, stream()), 'stove_bgspn_bwd')
        return d_in, d_m, g_coef, g_wroot, None


class _SceneFn(torch.autograd.Function):
    """Supair.likelihood fused (reference supair.py:44-110)."""

    @staticmethod
    def forward(ctx, frames, z, obj_coef, obj_wsum, obj_wroot, bg_coef, bg_wroot,
                obj_scope, obj_leaf_slot, bg_side, n_obj, beta):
        lib = _lib.load()
        frames, z = _f32(frames), _f32(z)
        tabs = [_f32(x) for x in (obj_coef, obj_wsum, obj_wroot, bg_coef, bg_wroot)]
        nf = frames.shape[0]
        dev = frames.device
        with torch.cuda.device(dev):
            ll = torch.empty(nf, dtype=torch.float32, device=dev)
            parts = torch.empty(nf, 3, dtype=torch.float32, device=dev)
            saved = torch.empty(lib.stove_scene_saved_floats(nf, n_obj) + 1, dtype=torch.float32, device=dev)
            t = _tables(obj=(obj_scope, obj_leaf_slot, tabs[0], tabs[1], tabs[2]), bg=(bg_side, tabs[3], tabs[4]))
            check(lib.stove_scene_fwd(ctypes.byref(t), ptr(frames), ptr(z), nf, n_obj, float(beta), ptr(ll), ptr(parts),
                                      ptr(saved), stream()), 'stove_scene_fwd')
        ctx.save_for_backward(frames, z, *tabs, obj_scope, obj_leaf_slot, bg_side, saved)
        ctx.n_obj, ctx.beta = n_obj, float(beta)
        ctx.mark_non_differentiable(parts)
        return ll, parts

    @staticmethod
    def backward(ctx, dll, _dparts):
        lib = _lib.load()
        frames, z, oc, ow, orr, bc, bw, obj_scope, obj_leaf_slot, bg_side, saved = ctx.saved_tensors
        nf, n_obj = frames.shape[0], ctx.n_obj
        dev = frames.device
        dll = _f32(dll)
        with torch.cuda.device(dev):
            dz = torch.empty_like(z)
            grads = [torch.empty_like(x) for x in (oc, ow, orr, bc, bw)]
            g = SpnTableGrads()
            g.obj_coef, g.obj_wsum, g.obj_wroot, g.bg_coef, g.bg_wroot = [ptr(x) for x in grads]
            ws = _ws(lib.stove_scene_bwd_ws_bytes(nf, n_obj), dev)
            t = _tables(obj=(obj_scope, obj_leaf_slot, oc, ow, orr), bg=(bg_side, bc, bw))
            check(lib.stove_scene_bwd(ctypes.byref(t), ptr(frames), ptr(z), nf, n_obj, ctx.beta, ptr(saved), ptr(dll),
                                      ptr(dz), ctypes.byref(g), ptr(ws), stream()), 'stove_scene_bwd')
        return (None, dz, *grads, None, None, None, None, None)


def objspn_apply(inputs, marg, coef, wsum, wroot, scope, leaf_slot):
    return _ObjSpnFn.apply(inputs, marg, coef, wsum, wroot, scope, leaf_slot)


def bgspn_apply(inputs, marg, coef, wroot, side):
    return _BgSpnFn.apply(inputs, marg, coef, wroot, side)


def scene_likelihood(frames, z, obj_tabs, bg_tabs, n_obj, beta):
    """frames (nf,1024), z (nf*n_obj,4)=[sx,sy,x,y]; obj_tabs=(coef,wsum,wroot,scope,leaf_slot),
    bg_tabs=(coef,wroot,side) -> ll (nf,), parts (nf,3)=(bg, patches, overlap)."""
    oc, ow, orr, osc, ols = obj_tabs
    bc, bw, bs = bg_tabs
    return _SceneFn.apply(frames, z, oc, ow, orr, bc, bw, osc, ols, bs, int(n_obj), float(beta))


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


def wave_sum_selftest(x):
    lib = _lib.load()
    x = _f32(x)
    out = torch.empty_like(x)
    check(lib.stove_selftest_wave_sum(ptr(x), ptr(out), x.numel() // 64, stream()), 'selftest')
    return out
