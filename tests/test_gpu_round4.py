"""Round-4 additions on the GPU: the library's counter-based noise generator, the step head off the serial chain."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def test_noise_source_statistics_and_reproducibility():
    """ops.NoiseSource (csrc/state.hip noise_normal_k: Philox-4x32-10 + Box-Muller): standard normal moments, a pure function of
    (seed, call number, element), seeded by torch.manual_seed, advanced on the device."""
    from stove_amd import ops
    torch.manual_seed(123)
    src = ops.NoiseSource(DEV)
    a = src.normal(4_000_001)
    b = src.normal(4_000_001)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and not torch.equal(a, b)
    x = a.double()
    assert abs(float(x.mean())) < 2e-3 and abs(float(x.var()) - 1.0) < 3e-3
    assert abs(float((x ** 3).mean())) < 1e-2 and abs(float((x ** 4).mean()) - 3.0) < 3e-2
    assert float(x.abs().max()) > 4.5 and float(x.abs().max()) < 6.5              # tails present, nothing absurd
    assert abs(float((a[:-1] * a[1:]).double().mean())) < 2e-3                    # neighbours (the two halves of a Box-Muller pair) uncorrelated
    assert abs(float((a * b).double().mean())) < 2e-3                             # successive calls uncorrelated
    assert int(src.state[1]) == 2
    # same seed, same call -> same numbers, whatever the length of the draw
    torch.manual_seed(123)
    src2 = ops.NoiseSource(DEV)
    c = src2.normal(1003)
    assert torch.equal(c, a[:1003])
    # a reseed of torch's generator reseeds the source
    torch.manual_seed(124)
    d = src2.normal(1003)
    assert not torch.equal(d, c) and int(src2.state[1]) == 1


def test_model_draws_from_the_library_generator_by_default():
    """Stove.forward with config.device_noise = 'philox' (default): different noise every call, the same after a reseed; 'torch'
    keeps torch.randn."""
    from stove_amd.envs import envs
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.stove import Stove
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height, cfg.random_seed = 3, 32, 32, 42
    cfg.device, cfg.dtype = DEV, torch.float32
    cfg.action_conditioned, cfg.action_space = False, None
    torch.manual_seed(0)
    model = Stove(cfg).to(DEV)
    x = torch.from_numpy(envs.synth_sequences('billiards', 4, 8, seed0=1)['X']).to(DEV)
    for mode in ('philox', 'torch'):
        cfg.device_noise = mode
        torch.manual_seed(5)
        e1 = float(model(x, 1)[0])
        e2 = float(model(x, 1)[0])
        torch.manual_seed(5)
        model.reset_noise()
        e3 = float(model(x, 1)[0])
        assert np.isfinite(e1) and e1 != e2 and e1 == e3, (mode, e1, e2, e3)
    assert model._noise_source.state is not None


def _greedy_serial(feat):
    """Stove._greedy_match_objects (reference stove.py:432-514) on one track, slot order, first minimum wins: numpy, exact inputs"""
    T, N, _ = feat.shape
    idx = np.zeros((T, N), dtype=np.int64)
    idx[0] = np.arange(N)
    for t in range(1, T):
        prev = (feat[t - 1][idx[t - 1]] + 1.0) * 0.5
        cur = (feat[t] + 1.0) * 0.5
        e = ((prev[:, None, :] - cur[None, :, :]) ** 2).sum(-1)
        for _ in range(N):
            a, j = np.unravel_index(np.argmin(e), e.shape)       # first occurrence in row-major (slot-major) order
            idx[t, a] = j
            e[a, :] = 3.0e38
            e[:, j] = 3.0e38
    return idx


@pytest.mark.parametrize('N,T,F', [(6, 100, 2), (6, 7, 2), (2, 2, 2), (8, 33, 2), (5, 100, 5), (4, 50, 2)])
def test_greedy_matcher_frames_in_parallel_equals_the_serial_walk(N, T, F):
    """csrc/match.hip match_greedy_frames_k + match_greedy_compose_k against the serial rule on tracks that sit on a coarse grid
    (every distance exact in fp32, exact ties in most frames: the tie flag and the slot-order redo are exercised) and on generic tracks."""
    from stove_amd import ops
    g = torch.Generator().manual_seed(N * 100 + T)
    B = 48
    coarse = torch.randint(-8, 9, (B, T, N, F), generator=g).float() / 8.0
    smooth = (torch.rand(B, 1, N, F, generator=g) * 2 - 1) + 0.02 * torch.randn(B, T, N, F, generator=g).cumsum(1)
    smooth = (smooth * 4096).round() / 4096              # exact products: the numpy rule decides on the same numbers
    for name, feat in (('ties', coarse), ('generic', smooth)):
        idx, _ = ops.match_objects(feat.to(DEV), 'greedy')
        want = np.stack([_greedy_serial(feat[b].double().numpy()) for b in range(B)])
        got = idx.cpu().numpy()
        assert got.shape == (B, T, N) and (np.sort(got, -1) == np.arange(N)).all(), name        # permutations
        assert (got == want).all(), (name, int((got != want).sum()))
    # the fused state pipeline runs the same kernels (stove_supair_state_fwd2): exercised by the N = 6 goldens (g6 / g7_stove_n6)
