"""Negative tests of the C ABI (include/stove_hip.h): bad arguments come back as a non-zero code that stove_error_string() names,
before anything is enqueued -- no launch, no fault, the stream stays usable.  The checks are csrc/validate.h (the same header the
sanitizer-built host driver tests/abi/validate_driver.cpp exercises in the CPU suite); here they are reached through the library's
real entry points with real device buffers."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
INVALID = 1          # hipErrorInvalidValue


@pytest.fixture(scope='module')
def env():
    from stove_amd import _lib, ops
    from stove_amd.spn.rat_torch import RatSpn  # noqa: F401
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.supair import Supair
    lib = _lib.load()
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height = 3, 32, 32
    cfg.device, cfg.dtype, cfg.random_seed = torch.device(DEV), torch.float32, 42
    sup = Supair(cfg).to(DEV)
    return lib, _lib, ops, sup


def _expect_invalid(lib, code, what):
    assert code == INVALID, (what, code)
    msg = lib.stove_error_string(code)
    assert msg and b'invalid' in msg.lower(), (what, msg)
    torch.cuda.synchronize()                       # nothing was enqueued, nothing faulted


def test_gnn_and_recursion_reject_bad_shapes(env):
    lib, _lib, ops, _ = env
    S = _lib.stream()
    B, N = 4, 3
    s = torch.zeros(B, N, 16, device=DEV)
    params = torch.zeros(lib.stove_gnn_param_floats(), device=DEV)
    res = torch.empty(B, N, 32, device=DEV)
    p = _lib.ptr
    _expect_invalid(lib, lib.stove_gnn_fwd(p(s), p(params), p(res), None, B, 0, 16, 2, 0, S), 'gnn_fwd N = 0')
    _expect_invalid(lib, lib.stove_gnn_fwd(p(s), p(params), p(res), None, B, 9, 16, 2, 0, S), 'gnn_fwd N = 9')
    _expect_invalid(lib, lib.stove_gnn_fwd(p(s), p(params), p(res), None, B, N, 33, 2, 0, S), 'gnn_fwd sin_dim = 33')
    _expect_invalid(lib, lib.stove_gnn_fwd(p(s), None, p(res), None, B, N, 16, 2, 0, S), 'gnn_fwd NULL params')
    _expect_invalid(lib, lib.stove_gnn_fwd(p(s), p(params), None, None, B, N, 16, 2, 0, S), 'gnn_fwd NULL result')
    assert lib.stove_gnn_fwd(None, None, None, None, 0, N, 16, 2, 0, S) == 0            # B = 0: a valid empty call
    g = torch.empty(lib.stove_gnn_grad_floats(), device=DEV)
    _expect_invalid(lib, lib.stove_gnn_bwd(p(s), p(params), p(res), None, p(s), p(g), None, B, N, 16, 2, 0, S), 'gnn_bwd NULL workspace')
    # recursion: action-conditioned input width without the `extra` stream; a negative step count; rollout with actions but A = 0
    Ts = 5
    z1 = torch.zeros(B, N, 18, device=DEV)
    zs = torch.zeros(B, Ts, N, 6, device=DEV)
    eps = torch.zeros(B, Ts, N, 18, device=DEV)
    outs = [torch.empty(B, Ts, N, d, device=DEV) for d in (18, 16, 16, 18, 18)]
    args = [p(z1), p(zs), p(zs), p(eps), None, p(params)] + [p(o) for o in outs] + [None, None]
    _expect_invalid(lib, lib.stove_dynloop_fwd(*args, B, Ts, N, 23, 2, 0, 0.3, 0.04, 0.04, S), 'dynloop_fwd sin_dim 23 without extra')
    _expect_invalid(lib, lib.stove_dynloop_fwd(*args, B, -1, N, 16, 2, 0, 0.3, 0.04, 0.04, S), 'dynloop_fwd Ts = -1')
    args_bad = list(args)
    args_bad[3] = None
    _expect_invalid(lib, lib.stove_dynloop_fwd(*args_bad, B, Ts, N, 16, 2, 0, 0.3, 0.04, 0.04, S), 'dynloop_fwd NULL eps')
    zp = torch.empty(B, 4, N, 18, device=DEV)
    _expect_invalid(lib, lib.stove_rollout_fwd(p(z1), p(z1), p(params), p(zp), None, None, B, 4, 0, N, 20, 2, 0, 0.3, 0.04, 0.04, S),
                    'rollout actions with A = 0')


def test_scene_and_spn_operator_reject_bad_arguments(env):
    lib, _lib, ops, sup = env
    S = _lib.stream()
    p = _lib.ptr
    nf, n_obj = 8, 3
    frames = torch.rand(nf, 1024, device=DEV)
    z = torch.tensor([0.3, 0.3, 0.0, 0.0], device=DEV).repeat(nf * n_obj, 1).contiguous()
    ll = torch.empty(nf, device=DEV)
    saved = torch.empty(lib.stove_scene_saved_floats(nf, n_obj), device=DEV)
    with torch.no_grad():
        oc, ow, orr, oscope, oslot = sup.obj_spn.tables()
        bc, bw, bside = sup.bg_spn.tables()
    obj = (oscope, oslot, oc, ow, orr)
    t = ops._tables(obj=obj, bg=(bside, bc, bw))
    ok = lib.stove_scene_fwd(ctypes.byref(t), p(frames), p(z), nf, n_obj, 0, 0, 10.0, p(ll), None, p(saved), S)
    assert ok == 0
    torch.cuda.synchronize()
    ref = ll.clone()
    bad = [
        ('NULL table struct', lambda: lib.stove_scene_fwd(None, p(frames), p(z), nf, n_obj, 0, 0, 10.0, p(ll), None, p(saved), S)),
        ('n_obj = 0', lambda: lib.stove_scene_fwd(ctypes.byref(t), p(frames), p(z), nf, 0, 0, 0, 10.0, p(ll), None, p(saved), S)),
        ('n_obj = 9', lambda: lib.stove_scene_fwd(ctypes.byref(t), p(frames), p(z), nf, 9, 0, 0, 10.0, p(ll), None, p(saved), S)),
        ('clip stride shorter than the slice', lambda: lib.stove_scene_fwd(ctypes.byref(t), p(frames), p(z), nf, n_obj, 4, 3, 10.0, p(ll), None, p(saved), S)),
        ('frames not a whole number of slices', lambda: lib.stove_scene_fwd(ctypes.byref(t), p(frames), p(z), nf, n_obj, 3, 4, 10.0, p(ll), None, p(saved), S)),
        ('NULL frames', lambda: lib.stove_scene_fwd(ctypes.byref(t), None, p(z), nf, n_obj, 0, 0, 10.0, p(ll), None, p(saved), S)),
        ('NULL saved', lambda: lib.stove_scene_fwd(ctypes.byref(t), p(frames), p(z), nf, n_obj, 0, 0, 10.0, p(ll), None, None, S)),
        ('negative frame count', lambda: lib.stove_scene_fwd(ctypes.byref(t), p(frames), p(z), -nf, n_obj, 0, 0, 10.0, p(ll), None, p(saved), S)),
    ]
    for what, call in bad:
        _expect_invalid(lib, call(), 'scene_fwd ' + what)
    t_hole = ops._tables(obj=obj, bg=None)
    _expect_invalid(lib, lib.stove_scene_fwd(ctypes.byref(t_hole), p(frames), p(z), nf, n_obj, 0, 0, 10.0, p(ll), None, p(saved), S),
                    'scene_fwd tables without the background SPN')
    # backward: NULL workspace / NULL gradient struct
    dll = torch.ones(nf, device=DEV)
    dz = torch.empty_like(z)
    _expect_invalid(lib, lib.stove_scene_bwd(ctypes.byref(t), p(frames), p(z), nf, n_obj, 0, 0, 10.0, p(saved), p(dll), p(dz), None, None, S),
                    'scene_bwd NULL grads and workspace')
    # the operator alone
    x = torch.rand(5, 100, device=DEV)
    xw = torch.empty(lib.stove_objspn_tile_floats(5), device=DEV)
    out = torch.empty(5, device=DEV)
    _expect_invalid(lib, lib.stove_objspn_fwd(None, p(x), None, p(xw), p(out), 5, S), 'objspn_fwd NULL tables')
    _expect_invalid(lib, lib.stove_objspn_fwd(ctypes.byref(t), p(x), None, p(xw), p(out), -5, S), 'objspn_fwd n < 0')
    _expect_invalid(lib, lib.stove_bgspn_fwd(ctypes.byref(t_hole), p(frames), None, p(saved), p(ll), nf, S), 'bgspn_fwd tables without the background SPN')
    assert lib.stove_objspn_fwd(None, None, None, None, None, 0, S) == 0                 # n = 0: a valid empty call
    # after all of that the stream still works and gives the same numbers
    assert lib.stove_scene_fwd(ctypes.byref(t), p(frames), p(z), nf, n_obj, 0, 0, 10.0, p(ll), None, p(saved), S) == 0
    torch.cuda.synchronize()
    assert torch.equal(ll, ref)


def test_gemm_rejects_bad_leading_dimensions(env):
    lib, _lib, ops, _ = env
    S = _lib.stream()
    p = _lib.ptr
    A = torch.rand(64, 64, device=DEV)
    Bm = torch.rand(64, 64, device=DEV)
    C = torch.empty(64, 64, device=DEV)

    def gemm(a=A, b=Bm, c=C, M=64, N=64, K=64, lda=64, ldb=64, ldc=64, nsplit=2, splitk=1, ws=None, b_off=0):
        return lib.stove_gemm_bf16(p(a), p(b) + b_off, None, None, p(c), M, N, K, lda, ldb, ldc, 0, 0, nsplit, splitk, 0, ws, S)
    assert gemm() == 0
    _expect_invalid(lib, gemm(lda=32), 'gemm lda shorter than a row')
    _expect_invalid(lib, gemm(ldb=66), 'gemm ldb not a multiple of 4')
    _expect_invalid(lib, gemm(b_off=4, N=63), 'gemm B not 16-byte aligned')
    _expect_invalid(lib, gemm(K=0), 'gemm K = 0')
    _expect_invalid(lib, gemm(nsplit=4), 'gemm nsplit = 4')
    _expect_invalid(lib, gemm(splitk=4), 'gemm split-K without a workspace')
    _expect_invalid(lib, gemm(M=-1), 'gemm M < 0')
    assert gemm(M=0) == 0
    torch.cuda.synchronize()
