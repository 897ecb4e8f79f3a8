"""ctypes binding of libstove_hip.so (C ABI in include/stove_hip.h).  No fallback: a missing
library or a non-GPU tensor is an error."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
from . import settings as _settings

LIB_PATH = _settings.LIB_OVERRIDE or os.path.join(_HERE, 'libstove_hip.so')
_lib = None
ABI_VERSION = 5

EXPORTS = [
    'stove_abi_version', 'stove_error_string', 'stove_selftest_wave_sum',
    'stove_objspn_tile_floats', 'stove_objspn_fwd', 'stove_objspn_bwd_ws_bytes', 'stove_objspn_bwd',
    'stove_bgspn_saved_floats', 'stove_bgspn_fwd', 'stove_bgspn_bwd_ws_bytes', 'stove_bgspn_bwd',
    'stove_scene_saved_floats', 'stove_scene_fwd', 'stove_scene_bwd_ws_bytes', 'stove_scene_bwd',
    'stove_scene_glimpses',
    'stove_gnn_param_floats', 'stove_gnn_grad_floats', 'stove_gnn_blocks', 'stove_gnn_fwd', 'stove_gnn_bwd_ws_bytes',
    'stove_gnn_bwd', 'stove_dynloop_act_floats', 'stove_dynloop_fwd', 'stove_dynloop_bwd_ws_bytes', 'stove_dynloop_bwd', 'stove_rollout_fwd', 'stove_match_objects', 'stove_profile_enable', 'stove_profile_report', 'stove_gnn_debug_stamps', 'stove_lstm_cell_fwd', 'stove_lstm_cell_bwd',
    'stove_spn_bake', 'stove_spn_bake_bwd', 'stove_arena_gather', 'stove_arena_scatter_add', 'stove_debug_set_stamps',
    'stove_supair_state_fwd', 'stove_supair_state_bwd', 'stove_zall_fwd', 'stove_zall_bwd', 'stove_elbo_fwd', 'stove_elbo_bwd', 'stove_flat_adam', 'stove_flat_adam_ws_bytes', 'stove_gemm_bf16', 'stove_gemm_bf16_ws_floats', 'stove_sum_chunks', 'stove_colsum_ws_floats', 'stove_colsum', 'stove_bw_transform', 'stove_dynloop_bwd_ws_bytes_ts', 'stove_scene_bwd_overlap', 'stove_dynloop_bwd_overlap', 'stove_glimpse_mean', 'stove_objspn_mpe', 'stove_render_frames',
    'stove_enc_head_fwd', 'stove_enc_head_bwd_ws_floats', 'stove_enc_head_bwd', 'stove_colsum2', 'stove_small_tn', 'stove_small_tn_ws_floats', 'stove_supair_state_fwd2', 'stove_supair_state_bwd2', 'stove_bg_dense', 'stove_bg_dense_floats',
    'stove_bw_transform_u8', 'stove_stream_after', 'stove_capture_begin', 'stove_capture_end', 'stove_graph_instantiate', 'stove_graph_launch', 'stove_graph_destroy',
    'stove_reward_head_param_floats', 'stove_reward_head_saved_floats', 'stove_reward_head_bwd_ws_floats', 'stove_reward_head_fwd',
    'stove_bgspn_saved_floats_d', 'stove_bgspn_fwd_d', 'stove_bgspn_bwd_ws_bytes_d', 'stove_bgspn_bwd_d', 'stove_noise_normal', 'stove_set_overlap', 'stove_set_tile_lds', 'stove_event_list_begin', 'stove_event_list_end', 'stove_event_list_destroy', 'stove_fill_words',
    'stove_reward_head_bwd', 'stove_small_linear', 'stove_set_fork_stream', 'stove_scene_fwd_from', 'stove_scene_fwd_floats', 'stove_gauss_ll_fwd', 'stove_gauss_ll_bwd', 'stove_objspn_saved_floats_any', 'stove_objspn_bwd_ws_bytes_any', 'stove_objspn_fwd_any', 'stove_objspn_bwd_any', 'stove_scene_saved_floats_any', 'stove_scene_bwd_ws_bytes_any', 'stove_scene_fwd_any', 'stove_scene_bwd_any', 'stove_scene_bwd_from',
]


class SpnTables(Structure):
    _fields_ = [('obj_scope', c_void_p), ('obj_leaf_slot', c_void_p), ('obj_coef', c_void_p),
                ('obj_wsum', c_void_p), ('obj_wroot', c_void_p),
                ('bg_side', c_void_p), ('bg_coef', c_void_p), ('bg_wroot', c_void_p), ('bg_dense', c_void_p)]


class SpnTableGrads(Structure):
    _fields_ = [('obj_coef', c_void_p), ('obj_wsum', c_void_p), ('obj_wroot', c_void_p),
                ('bg_coef', c_void_p), ('bg_wroot', c_void_p)]


class SpnArenaPlan(Structure):
    _fields_ = [('obj_mu', c_void_p), ('obj_rho', c_void_p), ('obj_sum', c_void_p),
                ('bg_mu', c_void_p), ('bg_rho', c_void_p), ('bg_gidx', c_void_p),
                ('obj_root', c_int), ('bg_root', c_int),
                ('obj_vmin', c_float), ('obj_vmax', c_float), ('bg_vmin', c_float), ('bg_vmax', c_float)]


def _declare(lib):
    P, I, F, S = c_void_p, c_int, c_float, c_size_t
    T, G = POINTER(SpnTables), POINTER(SpnTableGrads)
    A = POINTER(SpnArenaPlan)
    sig = {
        'stove_abi_version': (I, []),
        'stove_error_string': (c_char_p, [I]),
        'stove_selftest_wave_sum': (I, [P, P, I, P]),
        'stove_objspn_tile_floats': (S, [I]),
        'stove_objspn_fwd': (I, [T, P, P, P, P, I, P]),
        'stove_objspn_bwd_ws_bytes': (S, [I]),
        'stove_objspn_bwd': (I, [T, P, P, P, P, P, P, G, P, I, P]),
        'stove_bgspn_saved_floats': (S, [I]),
        'stove_bgspn_fwd': (I, [T, P, P, P, P, I, P]),
        'stove_bgspn_bwd_ws_bytes': (S, [I]),
        'stove_bgspn_bwd': (I, [T, P, P, P, P, P, P, P, G, P, I, P]),
        'stove_scene_saved_floats': (S, [I, I]),
        'stove_scene_fwd': (I, [T, P, P, I, I, I, I, F, P, P, P, P]),
        'stove_scene_fwd_from': (I, [T, P, P, I, I, I, I, F, P, P, P, P, P, I]),
        'stove_scene_fwd_floats': (S, [I, I, I]),
        'stove_scene_bwd_from': (I, [T, P, P, I, I, I, I, F, P, P, P, G, P, P, P, P]),
        'stove_scene_bwd_ws_bytes': (S, [I, I]),
        'stove_scene_bwd': (I, [T, P, P, I, I, I, I, F, P, P, P, G, P, P]),
        'stove_scene_bwd_overlap': (I, [T, P, P, I, I, I, I, F, P, P, P, G, P, P, P]),
        'stove_gauss_ll_fwd': (I, [P, P, P, I, I, F, F, P]),
        'stove_gauss_ll_bwd': (I, [P, P, P, P, P, I, I, F, F, P]),
        'stove_objspn_saved_floats_any': (S, [I, I, I, I, I, I]),
        'stove_objspn_bwd_ws_bytes_any': (S, [I, I, I, I, I, I]),
        'stove_objspn_fwd_any': (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
        'stove_objspn_bwd_any': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
        'stove_scene_saved_floats_any': (S, [I, I, I, I]),
        'stove_scene_bwd_ws_bytes_any': (S, [I, I, I]),
        'stove_scene_fwd_any': (I, [T, P, P, I, I, I, I, I, I, I, F, P, P, P, P, I]),
        'stove_scene_bwd_any': (I, [T, P, P, I, I, I, I, I, I, I, F, P, P, P, G, P, P, P]),
        'stove_scene_glimpses': (I, [P, P, I, I, P, P, P, P]),
        'stove_gnn_param_floats': (S, []),
        'stove_gnn_grad_floats': (S, []),
        'stove_gnn_blocks': (I, [I, I]),
        'stove_gnn_fwd': (I, [P, P, P, P, I, I, I, I, I, P]),
        'stove_gnn_bwd_ws_bytes': (S, [I, I]),
        'stove_gnn_bwd': (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
        'stove_dynloop_act_floats': (S, [I, I, I]),
        'stove_dynloop_fwd': (I, [P] * 13 + [I] * 6 + [F] * 3 + [P]),
        'stove_dynloop_bwd_ws_bytes': (S, [I, I]),
        'stove_dynloop_bwd': (I, [P] * 19 + [I] * 6 + [F] * 3 + [P]),
        'stove_dynloop_bwd_overlap': (I, [P] * 19 + [I] * 6 + [F] * 3 + [P, P]),
        'stove_rollout_fwd': (I, [P] * 6 + [I] * 7 + [F] * 3 + [P]),
        'stove_match_objects': (I, [P, P, P, I, I, I, I, I, P]),
        'stove_profile_enable': (None, [I]),
        'stove_lstm_cell_fwd': (I, [P, P, P, P, P, I, I, I, P]),
        'stove_lstm_cell_bwd': (I, [P] * 10 + [I, I, I, I, P]),
        'stove_gnn_debug_stamps': (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
        'stove_profile_report': (S, [c_char_p, S]),
        'stove_spn_bake': (I, [P, A, P, P, P, P, P, P]),
        'stove_spn_bake_bwd': (I, [P, A, G, P, P]),
        'stove_debug_set_stamps': (None, [P]),
        'stove_glimpse_mean': (I, [P, P, P, I, I, I, P]),
        'stove_objspn_mpe': (I, [T, P, P, P, P, P, I, P]),
        'stove_render_frames': (I, [P, P, I, P, P, I, I, P]),
        'stove_enc_head_fwd': (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
        'stove_enc_head_bwd_ws_floats': (S, [I, I]),
        'stove_enc_head_bwd': (I, [P, P, P, P, P, P, P, P, P, P, I, P, I, I, I, I, I, P]),
        'stove_colsum2': (I, [P, P, P, I, P, I, I, P]),
        'stove_small_tn_ws_floats': (S, [I, I, I]),
        'stove_small_tn': (I, [P, P, P, P, I, I, I, P]),
        'stove_dynloop_bwd_ws_bytes_ts': (S, [I, I, I]),
        'stove_bw_transform': (I, [P, P, I, I, I, P]),
        'stove_bw_transform_u8': (I, [P, P, I, I, I, P]),
        'stove_stream_after': (I, [P, P]),
        'stove_small_linear': (I, [P, P, P, P, I, I, I, I, P]),
        'stove_reward_head_param_floats': (S, []),
        'stove_reward_head_saved_floats': (S, [I, I]),
        'stove_reward_head_bwd_ws_floats': (S, [I]),
        'stove_reward_head_fwd': (I, [P, P, P, P, I, I, P]),
        'stove_reward_head_bwd': (I, [P] * 8 + [I, I, P]),
        'stove_set_fork_stream': (I, [I, P, I]),
        'stove_noise_normal': (I, [P, S, P, P]),
        'stove_bgspn_saved_floats_d': (S, [I, I]),
        'stove_bgspn_fwd_d': (I, [T, P, P, P, P, I, I, P]),
        'stove_bgspn_bwd_ws_bytes_d': (S, [I, I]),
        'stove_bgspn_bwd_d': (I, [T, P, P, P, P, P, P, P, G, P, I, I, P]),
        'stove_set_overlap': (I, [I]),
        'stove_set_tile_lds': (I, [I]),
        'stove_event_list_begin': (P, []),
        'stove_event_list_end': (I, [P]),
        'stove_event_list_destroy': (I, [P]),
        'stove_fill_words': (I, [P, ctypes.c_uint32, S, P]),
        'stove_capture_begin': (I, [P]),
        'stove_capture_end': (I, [P, POINTER(c_void_p), POINTER(c_int)]),
        'stove_graph_instantiate': (I, [P, POINTER(c_void_p)]),
        'stove_graph_launch': (I, [P, P]),
        'stove_graph_destroy': (I, [P]),
        'stove_colsum_ws_floats': (S, [I, I]),
        'stove_colsum': (I, [P, P, P, I, I, P]),
        'stove_sum_chunks': (I, [P, P, S, I, P]),
        'stove_flat_adam_ws_bytes': (S, [I]),
        'stove_gemm_bf16_ws_floats': (S, [I, I, I]),
        'stove_gemm_bf16': (I, [P, P, P, P, P] + [I] * 11 + [P, P]),
        'stove_flat_adam': (I, [P] * 5 + [S, P, P, P, I, P, P, P, F, F, F, F, F, I, P]),
        'stove_supair_state_fwd': (I, [P] * 10 + [I] * 6 + [P]),
        'stove_supair_state_bwd': (I, [P] * 11 + [I] * 4 + [P]),
        'stove_bg_dense_floats': (S, []),
        'stove_bg_dense': (I, [P, P, P, P]),
        'stove_supair_state_fwd2': (I, [P] * 10 + [I, P, I] + [I] * 6 + [P]),
        'stove_supair_state_bwd2': (I, [P] * 8 + [I] + [P] * 3 + [I] * 4 + [P]),
        'stove_zall_fwd': (I, [P, P, P, I, I, I, I, P]),
        'stove_zall_bwd': (I, [P, P, P, P, P, P, I, I, I, I, P]),
        'stove_elbo_fwd': (I, [P] * 8 + [I] * 4 + [P]),
        'stove_elbo_bwd': (I, [P] * 11 + [I] * 4 + [P]),
        'stove_arena_gather': (I, [P, P, P, I, P]),
        'stove_arena_scatter_add': (I, [P, P, P, I, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name in OPTIONAL_SIGS:
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = OPTIONAL_SIGS[name]


OPTIONAL_SIGS = {}


def load():
    """Return the loaded library; raise if it has not been built (run `python -m stove_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -m stove_amd.build` '
                '(hipcc --offload-arch=gfx950). stove_amd has no CPU fallback.')
        lib = ctypes.CDLL(LIB_PATH)
        _declare(lib)
        if lib.stove_abi_version() != ABI_VERSION:
            raise RuntimeError(f'{LIB_PATH} has ABI version {lib.stove_abi_version()}, this binding needs {ABI_VERSION}: '
                               'rebuild it with `python -m stove_amd.build --force`')
        # the A/B measurement switches live in the binding, not in the library (which never reads the environment)
        from . import settings
        lib.stove_set_overlap(1 if settings.OVERLAP else 0)
        if settings.LIB_OVERRIDE:             # an A/B run: say which build this process measured
            import sys
            sys.stderr.write('[stove_amd] STOVE_LIB: loaded %s\n' % LIB_PATH)
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        msg = load().stove_error_string(code)
        raise RuntimeError(f'{what} failed: hipError {code} ({msg.decode() if msg else "?"})')


def ptr(t):
    """Device pointer of a contiguous float32/int32 CUDA(HIP) tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('stove_amd HIP ops need tensors on a GPU (cuda:N); there is no CPU path')
    if not t.is_contiguous():
        raise RuntimeError('stove_amd HIP ops need contiguous tensors')
    return t.data_ptr()


_FORCED_STREAM = None


def stream():
    """The stream the C entry points enqueue on: torch's current stream, unless a section redirected the library's launches
    (force_stream: work routed to the second stream while torch's allocator keeps seeing the current one, ops.run_on_side)."""
    return _FORCED_STREAM if _FORCED_STREAM is not None else torch.cuda.current_stream().cuda_stream


def force_stream(handle):
    """Redirect the library's launches to the raw stream `handle` (None: back to torch's current stream); returns the previous setting."""
    global _FORCED_STREAM
    prev, _FORCED_STREAM = _FORCED_STREAM, handle
    return prev


def profile_report(wall=False):
    """{kernel name: (total ms, launches)} recorded since stove_profile_enable(1); clears the records.
    wall=True: (total ms, launches, ms covered by the launches' time spans -- less than the total where launches of the
    kernel overlap on two streams)."""
    lib = load()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.stove_profile_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, ms, cnt, cover = line.split('\t')
        name = name.strip('()').split('<')[0].split('::')[-1]
        tot, c, w = out.get(name, (0.0, 0, 0.0))
        out[name] = (tot + float(ms), c + int(cnt), w + float(cover))
    return out if wall else {k: v[:2] for k, v in out.items()}
