"""CPU-side checks (-m "not gpu"): the C ABI library loads and exports every declared symbol,
the host-side structure code reproduces the reference's SPN structure, and the product path
refuses to run without a GPU (no silent fallback)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import stove_oracle as O
from helpers import GOLDEN, load_golden, t_

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from stove_amd import _lib, build
    build.build_library()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, 'include', 'stove_hip.h')).read()
    declared = set(re.findall(r'^(?:int|size_t|void\*?|const char\*)\s+(stove_[a-z0-9_]+)\s*\(', header, flags=re.M))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().stove_abi_version() == _lib.ABI_VERSION
    # size queries are pure host functions
    assert _lib.load().stove_objspn_tile_floats(65) == 2 * 100 * 2 * 64
    assert _lib.load().stove_scene_saved_floats(10, 3) > 0


@pytest.mark.parametrize('seed', [7, 42])
def test_region_graph_matches_reference_structure(seed):
    from stove_amd.spn import probabilistic_models as prob
    with open(os.path.join(GOLDEN, 'g1_spn_structure.json')) as f:
        gold = json.load(f)
    c = O.default_config(random_seed=seed)
    for kind, spn in (('obj', prob._get_obj_spn(c, seed)), ('bg', prob._get_bg_spn(c, seed))):
        g = gold[f'{kind}_{seed}']
        pos = {}
        for li, layer in enumerate(spn.vector_list):
            assert len(layer) == len(g['layers'][li])
            for i, vec in enumerate(layer):
                pos[id(vec)] = (li, i)
                ref = g['layers'][li][i]
                if li == 0:
                    assert list(vec.scope) == ref
                elif li % 2 == 1:
                    a, b = vec.inputs
                    assert [*pos[id(a)], *pos[id(b)]] == ref
                else:
                    assert [pos[id(p)][1] for p in vec.inputs] == ref
        assert list(pos[id(spn.output_vector)]) == g['root']
        assert spn._kind == kind


def test_state_dict_names_match_reference_inventory():
    from stove_amd.video_prediction.stove import Stove
    from stove_amd.video_prediction.config import StoveConfig
    for extra in ({}, dict(action_conditioned=True, action_space=9, debug_core_appearance=True)):
        c = O.default_config(**extra)
        cfg = StoveConfig()
        cfg.num_obj, cfg.width, cfg.height, cfg.random_seed = 3, 32, 32, 42
        cfg.device, cfg.dtype = torch.device('cpu'), torch.float32
        cfg.action_conditioned, cfg.action_space = c.action_conditioned, c.action_space
        cfg.debug_core_appearance = c.debug_core_appearance
        shapes = O.param_shapes(c, O.build_structs(c))
        sd = Stove(cfg).state_dict()
        for k, shp in shapes.items():
            assert k in sd and tuple(sd[k].shape) == tuple(shp), k
        extra_keys = set(sd) - set(shapes)
        assert extra_keys == {'sup.obj_spn.output_vector.params', 'sup.bg_spn.output_vector.params'}


def test_ops_refuse_cpu_tensors():
    from stove_amd.spn import probabilistic_models as prob
    c = O.default_config()
    spn = prob._get_obj_spn(c, 42)
    with pytest.raises(RuntimeError, match='GPU'):
        spn(torch.rand(4, 100), None)


def test_host_side_units_against_reference():
    from stove_amd.video_prediction.stove import Stove
    from stove_amd.video_prediction.config import StoveConfig
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height, cfg.random_seed = 3, 32, 32, 42
    cfg.device, cfg.dtype, cfg.action_conditioned = torch.device('cpu'), torch.float64, False
    torch.set_default_dtype(torch.float64)
    try:
        st = Stove(cfg)
        g = load_golden('g10_units')
        m, s = st.sup.constrain_zp(t_(g['zp']))
        assert np.abs(m.numpy() - g['zp_mean']).max() < 1e-14 and np.abs(s.numpy() - g['zp_std']).max() < 1e-14
        mc, sc = st.dyn.constrain_z_dyn(t_(g['zd']), t_(g['zds']))
        assert np.abs(mc.numpy() - g['zd_c']).max() < 1e-14 and np.abs(sc.numpy() - g['zds_c']).max() < 1e-14
        assert np.abs(st.v_from_state(t_(g['zsup'])).numpy() - g['v_full']).max() < 1e-14
        assert np.abs(st.v_std_from_pos(t_(g['zsups'])).numpy() - g['vstd_full']).max() < 1e-14
        gf = load_golden('g6_fix_supair')
        a, b = st.fix_supair(t_(gf['z']), t_(gf['zstd']))
        assert np.abs(a.numpy() - gf['z_fixed']).max() < 1e-15 and np.abs(b.numpy() - gf['zstd_fixed']).max() < 1e-15
    finally:
        torch.set_default_dtype(torch.float32)


def test_prediction_error_permutation_gather():
    """Trainer.prediction_error (reference train.py:475-563): the best-permutation reordering as one gather equals the
    per-sequence Python loop of the reference."""
    import itertools
    from types import SimpleNamespace
    import torch
    from stove_amd.video_prediction.train import Trainer
    g = torch.Generator().manual_seed(0)
    n, T, o = 9, 6, 3
    true = torch.rand(n, T, o, 4, generator=g)
    pred = true[:, :, torch.tensor([2, 0, 1])] + 0.01 * torch.randn(n, T, o, 4, generator=g)
    pred[::2] = true[::2][:, :, torch.tensor([1, 0, 2])] + 0.01 * torch.randn(5, T, o, 4, generator=g)
    me = SimpleNamespace(c=SimpleNamespace(supair_only=False, num_obj=o))
    res = Trainer.prediction_error(me, pred, true, return_matched=True)
    perms = list(itertools.permutations(range(o)))
    errs = torch.stack([torch.sqrt(((pred[:, :4, list(p), :2] - true[:, :4, :, :2]) ** 2).sum(-1)).mean((1, 2)) for p in perms], 1)
    best = errs.argmin(1).tolist()
    pos_m = torch.stack([pred[i, :, list(perms[j]), :2] for i, j in enumerate(best)], 0)
    vel_m = torch.stack([pred[i, :, list(perms[j]), 2:4] for i, j in enumerate(best)], 0)
    assert torch.equal(res['pos_matched'], pos_m) and torch.equal(res['vel_matched'], vel_m)
    assert float(res['error']) < 0.05 and float(res['swaps']) == 1.0
    # SuPAIR-only: one permutation per image
    me.c.supair_only = True
    res = Trainer.prediction_error(me, pred, true, return_matched=True)
    pf, tf = pred[..., :2].flatten(end_dim=1), true[..., :2].flatten(end_dim=1)
    errs = torch.stack([torch.sqrt(((pf[:, list(p)] - tf) ** 2).sum(-1)).mean(1) for p in perms], 1)
    best = errs.argmin(1).tolist()
    want = torch.stack([pf[i][list(perms[j])] for i, j in enumerate(best)], 0).reshape(n, T, o, 2)
    assert torch.equal(res['pos_matched'], want)


def test_settle_host_gc_keeps_the_collector_enabled():
    """utils.settle_host_gc (called by Trainer.train and bench.py once the step loop is warm): a full collection, then the
    survivors frozen -- the collector itself stays on, and a later full collection no longer walks the frozen objects."""
    import gc
    from stove_amd.utils.utils import settle_host_gc
    keep = [[i] for i in range(1000)]
    before = gc.get_freeze_count()
    settle_host_gc()
    try:
        assert gc.isenabled()
        assert gc.get_freeze_count() > before
        assert len(keep) == 1000
    finally:
        gc.unfreeze()


def test_recognition_network_row_chunks_cover_the_batch_in_whole_tiles():
    """ops._enc_chunks (host logic of the chunked forward chain, reference encoder.py:43-51): the chunks partition the rows, every
    boundary on a 256-row tile, and shapes the 256 x 256 tile cannot take whole (or that are too small to fill the chip) stay unchunked."""
    from stove_amd import ops
    saved = ops.ENC_CHUNKS
    try:
        ops.ENC_CHUNKS = 2
        assert ops._enc_chunks(25600, 1024, 256) == [(0, 12800), (12800, 25600)]
        assert ops._enc_chunks(2048, 1024, 256) is None            # the reference's default training shape: 256 clips x 8 frames
        assert ops._enc_chunks(25600, 2500, 256) is None           # 50 x 50 frames: K is not a whole number of k-steps
        assert ops._enc_chunks(25601, 1024, 256) is None
        ops.ENC_CHUNKS = 3
        ch = ops._enc_chunks(25600, 1024, 256)
        assert ch[0][0] == 0 and ch[-1][1] == 25600 and all(a[1] == b[0] for a, b in zip(ch, ch[1:])) and all(r0 % 256 == 0 for r0, _ in ch)
        ops.ENC_CHUNKS = 1
        assert ops._enc_chunks(25600, 1024, 256) is None
    finally:
        ops.ENC_CHUNKS = saved


def test_abi_validation_layer_under_sanitizers(tmp_path):
    """SURVEY section 5 (sanitizers): the C ABI's argument checks (stove_amd/csrc/validate.h, called first by the library's entry points)
    compiled host-only with -fsanitize=address,undefined and driven with NULL tables, empty and negative sizes, out-of-range object
    counts, bad frame maps, short / misaligned leading dimensions: every case returns the documented code and nothing is dereferenced."""
    import shutil
    import subprocess
    cxx = shutil.which('g++') or shutil.which('c++')
    if cxx is None:
        pytest.skip('no host C++ compiler')
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'abi', 'validate_driver.cpp')
    exe = str(tmp_path / 'validate_driver')
    r = subprocess.run([cxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-Wall', '-Werror',
                        '-o', exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1'))
    assert r.returncode == 0, r.stdout + r.stderr
    assert '0 failure(s)' in r.stdout


def test_library_entry_points_call_the_validation_layer():
    """Every check of validate.h is wired into an entry point of csrc/capi.hip (the driver above would otherwise test dead code)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'stove_amd', 'csrc', 'validate.h')) as f:
        checks = set(re.findall(r'^inline int (\w+)\(', f.read(), re.M))
    with open(os.path.join(root, 'stove_amd', 'csrc', 'capi.hip')) as f:
        used = set(re.findall(r'STOVE_VALIDATE\((\w+)\(', f.read()))
    helpers = {'obj_tables', 'bg_tables', 'table_grads', 'frame_map', 'gnn_shape'}
    assert checks - helpers == used, (sorted(checks - helpers - used), sorted(used - checks))


def test_product_path_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under stove_amd/ or model/ (the product) may import, load or name it, and nothing
    there may reach for /root/reference; bench.py may only import it inside cpu_baseline() (after the timed region) and
    __graft_entry__.py only inside smoke()."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for top in ('stove_amd', 'model'):
        for dirpath, _, files in os.walk(os.path.join(root, top)):
            for fn in files:
                if not fn.endswith(('.py', '.hip', '.h')):
                    continue
                with open(os.path.join(dirpath, fn), errors='replace') as f:
                    text = f.read()
                if re.search(r'stove_oracle|/root/reference|import oracle|from oracle', text):
                    offenders.append(os.path.join(dirpath, fn))
    assert not offenders, offenders

    def importers(path):
        """names of the functions whose bodies import stove_oracle (module level = '<module>')"""
        with open(path) as f:
            tree = ast.parse(f.read())
        found = []

        def visit(node, fn):
            for child in ast.iter_child_nodes(node):
                name = child.name if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)) else fn
                if isinstance(child, (ast.Import, ast.ImportFrom)):
                    mods = [a.name for a in child.names] + ([child.module] if isinstance(child, ast.ImportFrom) and child.module else [])
                    if any(m and 'stove_oracle' in m for m in mods):
                        found.append(fn)
                visit(child, name)
        visit(tree, '<module>')
        return found
    assert importers(os.path.join(root, 'bench.py')) == ['cpu_baseline']
    assert importers(os.path.join(root, '__graft_entry__.py')) == ['smoke']
