"""GPU parity tests of the MPE rendering path (Supair.spn_max_activation / spn_mpe / reconstruct_from_z, reference
supair.py:357-498): object-SPN MPE kernel and frame-rendering kernel against the reference-generated golden (g12) and the
CPU oracle.  The MPE walk is index work (argmax per sum node): the chosen components must match exactly; the rendered
pixels are fp32 bilinear sums, tolerance 2e-5 absolute on values in [0, 1]."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import stove_oracle as O
from helpers import load_golden, oracle_setup, t_
from test_gpu_spn import _supair_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
PIX_TOL = 2e-5


def _maxabs(a, b):
    return float(np.abs(a.detach().double().cpu().numpy() - np.asarray(b, dtype=np.float64)).max())


def test_reconstruct_vs_reference_golden():
    g = load_golden('g12_reconstruct_f32')
    _, _, _, sup = _supair_pair(3)
    z, x = t_(g['z']).float().to(DEV), t_(g['x']).float().to(DEV)
    assert _maxabs(sup.spn_max_activation(sup.bg_spn), g['bg_max']) < 1e-6
    assert _maxabs(sup.spn_max_activation(), g['obj_max']) < 1e-6
    mpe = sup.spn_mpe(z.flatten(end_dim=1), x.flatten(end_dim=1))
    assert mpe.shape == g['mpe_patches'].shape and mpe.is_cuda
    assert _maxabs(mpe, g['mpe_patches']) < 1e-6              # same leaf components everywhere
    for key, kw in (('recon_max', {}), ('recon_mpe', dict(x=x, max_activation=False, single_image=False)),
                    ('recon_mpe_single', dict(x=x[:, 0], max_activation=False, single_image=True))):
        r = sup.reconstruct_from_z(z, **kw)
        assert r.shape == g[key].shape and r.dtype == torch.float32
        assert _maxabs(r, g[key]) < PIX_TOL, key
    # states with extra columns (the Trainer passes full 18-dim states) use the first four
    z18 = torch.cat([z, torch.randn(*z.shape[:3], 14, device=DEV)], -1)
    assert torch.equal(sup.reconstruct_from_z(z18), sup.reconstruct_from_z(z))
    with pytest.raises(ValueError):
        sup.reconstruct_from_z(z, max_activation=False)
    with pytest.raises(ValueError):
        sup.spn_mpe(z.flatten(end_dim=1)[:2], x.flatten(end_dim=1))


@pytest.mark.parametrize('n_frames', [70, 1, 22])
def test_mpe_walk_vs_oracle_ragged(n_frames):
    """Glimpse counts that are not multiples of the 64-sample tile; component choices identical to the oracle's."""
    c, structs, params, sup = _supair_pair(3)
    params = {k: v.detach() for k, v in params.items()}
    g = torch.Generator().manual_seed(100 + n_frames)
    x64 = torch.rand(n_frames, 1, 32, 32, generator=g, dtype=torch.float64) ** 2
    z64 = torch.zeros(n_frames, 3, 4, dtype=torch.float64)
    z64[..., 0] = 0.1 + 0.6 * torch.rand(n_frames, 3, generator=g, dtype=torch.float64)
    z64[..., 1] = z64[..., 0] * (0.75 + 0.5 * torch.rand(n_frames, 3, generator=g, dtype=torch.float64))
    z64[..., 2:] = 1.9 * torch.rand(n_frames, 3, 2, generator=g, dtype=torch.float64) - 0.95
    ref = O.spn_mpe(c, params, structs, z64, x64)
    ours = sup.spn_mpe(z64.float().to(DEV), x64.float().to(DEV))
    diff = np.abs(ours.double().cpu().numpy() - ref.numpy()).reshape(n_frames * 3, -1).max(1)
    assert (diff < 1e-6).all(), 'glimpses with a different walk: %s' % np.nonzero(diff >= 1e-6)[0]
    # the kernel's picks are consistent with its output: rebuild the patch from (replica, 4 components) on the host
    patches = sup.patches_from_z(x64.float().to(DEV), z64.float().to(DEV).flatten(end_dim=1)).flatten(start_dim=1)
    out, pick = sup.obj_spn.mpe(patches, return_pick=True)
    assert torch.equal(out.view_as(ours), ours)
    spn = sup.obj_spn
    order = spn._plan_cpu['leaf_order'].view(6, 4)
    pick = pick.cpu().numpy()
    assert len(np.unique(pick, axis=0)) > 1
    for j in range(0, n_frames * 3, 7):
        want = np.zeros(100)
        for L in range(4):
            leaf = spn.vector_list[0][int(order[pick[j, 0], L])]
            want[leaf.scope] = leaf.means[:, pick[j, 1 + L]].detach().cpu().numpy()
        assert np.abs(np.clip(want, 0, 1) - out[j].cpu().numpy()).max() < 1e-7


def test_mpe_reconstruction_is_fixed_point_of_its_own_walk():
    """Size-independent property: feeding an MPE reconstruction back selects a walk that explains it at least as well --
    the reconstruction of a leaf-mean image whose components all sit on one walk reproduces that image."""
    _, _, _, sup = _supair_pair(3)
    spn = sup.obj_spn
    img = torch.as_tensor(spn.reconstruct(spn.max_activation_idxs(), 0, False), dtype=torch.float32, device=DEV)
    out = spn.mpe(img.clamp(0, 1).view(1, -1).repeat(130, 1))
    assert torch.equal(out, out[:1].expand_as(out))            # same input -> same walk in every lane / tile


def _render_torch(bg, patches, z, n_obj):
    """reconstruct_from_z's paste loop with torch ops (supair.py:484-498 semantics) on the device."""
    nf = z.shape[0] // n_obj
    rec = bg.view(1, 1, 32, 32).repeat(nf, 1, 1, 1)
    zf = z.view(nf, n_obj, 4)
    for k in range(n_obj):
        zk = zf[:, k]
        zero = torch.zeros_like(zk[:, 0])
        th = torch.stack([1 / zk[:, 0], zero, -zk[:, 2] / zk[:, 0], zero, 1 / zk[:, 1], -zk[:, 3] / zk[:, 1]], 1).view(-1, 2, 3)
        grid = F.affine_grid(th, (nf, 1, 32, 32), align_corners=False)
        rec = rec + F.grid_sample(patches[:, k].reshape(nf, 1, 10, 10), grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    return rec.clamp(0, 1).view(nf, 1024)


@pytest.mark.parametrize('nf,n_obj', [(256 * 100, 3), (37, 6), (5, 1)])
def test_render_frames_full_size(nf, n_obj):
    """The headline batch (256 sequences x 100 frames) and odd shapes against torch's affine_grid + grid_sample."""
    from stove_amd import ops
    g = torch.Generator().manual_seed(5)
    bg = torch.rand(1024, generator=g).to(DEV) * 0.5
    patches = torch.rand(nf, n_obj, 100, generator=g).to(DEV)
    z = torch.zeros(nf * n_obj, 4)
    z[:, 0] = 0.1 + 0.7 * torch.rand(nf * n_obj, generator=g)
    z[:, 1] = z[:, 0] * (0.75 + 0.5 * torch.rand(nf * n_obj, generator=g))
    z[:, 2:] = 1.9 * torch.rand(nf * n_obj, 2, generator=g) - 0.95
    z = z.to(DEV)
    out = ops.render_frames(bg, patches, 1, z, n_obj)
    ref = _render_torch(bg, patches, z, n_obj)
    assert float((out - ref).abs().max()) < PIX_TOL
    # one shared patch for every object (max-activation mode) == the same patch repeated
    shared = ops.render_frames(bg, patches[0, 0].contiguous(), 0, z, n_obj)
    rep = ops.render_frames(bg, patches[0, 0].view(1, 1, 100).repeat(nf, n_obj, 1), 1, z, n_obj)
    assert torch.equal(shared, rep)
    # patches held for T frames (single-image mode)
    if nf % 5 == 0:
        held = ops.render_frames(bg, patches[:nf // 5].contiguous(), 5, z, n_obj)
        full = ops.render_frames(bg, patches[:nf // 5].unsqueeze(1).repeat(1, 5, 1, 1).flatten(end_dim=1), 1, z, n_obj)
        assert torch.equal(held, full)
    # zero patches leave the clamped background
    blank = ops.render_frames(bg, torch.zeros_like(patches), 1, z, n_obj)
    assert torch.equal(blank, bg.clamp(0, 1).view(1, -1).expand(nf, -1))


def test_render_and_mpe_empty():
    from stove_amd import ops
    _, _, _, sup = _supair_pair(3)
    assert ops.render_frames(torch.zeros(1024, device=DEV), torch.zeros(1, 100, device=DEV), 0, torch.zeros(0, 4, device=DEV), 3).shape == (0, 1024)
    assert sup.obj_spn.mpe(torch.zeros(0, 100, device=DEV)).shape == (0, 100)
    with pytest.raises(ValueError):
        ops.render_frames(torch.zeros(1024, device=DEV), torch.zeros(2, 100, device=DEV), 1, torch.zeros(9, 4, device=DEV), 3)
