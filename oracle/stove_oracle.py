"""CPU oracle for the STOVE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a plain-PyTorch (CPU, autograd) restatement of the reference
algorithm for the path named in BASELINE.json `north_star`:

    SuPAIR scene likelihood (SPN over glimpses + background)   model/spn/*, model/video_prediction/supair.py
    relational GNN dynamics core + inference recursion         model/video_prediction/dynamics.py, stove.py

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / the timed CPU baseline.
Nothing under `stove_amd/` imports it; the product path raises when the HIP
library is missing instead of falling back to this code.

Parity pin: the reference has no tests of its own, so the
oracle is pinned against outputs of the reference itself, produced by
`oracle/make_goldens.py` importing `/root/reference` in the build container and
committed as `tests/golden/*.npz` (see `tests/test_oracle_goldens.py`).

It is written functionally (plain dicts of tensors keyed by the reference's
state-dict names, a plain config namespace), not as a copy of the reference's
`nn.Module` classes; every function cites the reference lines it restates.
All file:line citations are relative to /root/reference.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------
# configuration (model/video_prediction/config.py:6-134 -- the attributes the
# hot path reads, with the reference defaults)
# --------------------------------------------------------------------------
def default_config(**overrides):
    c = SimpleNamespace(
        num_obj=3, width=32, height=32, channels=1, cl=32, skip=2,
        patch_width=10, patch_height=10,
        obj_min_var=0.12, obj_max_var=0.35, bg_min_var=0.002, bg_max_var=0.16,
        obj_spn_num_gauss=10, obj_spn_num_sums=10,
        scale_var=0.3, pos_var=0.3,
        min_obj_scale=0.1, max_obj_scale=0.8, min_y_scale=0.75, max_y_scale=1.25,
        obj_pos_bound=0.9, overlap_beta=10.0,
        transition_lik_std=[0.01, 0.01, 0.01, 0.01],
        debug_latent_q_std=0.04, debug_nonlinear='relu',
        debug_fix_supair=True, debug_match_objects='3_only',
        debug_bg_model=False, debug_obj_spn=False, debug_simple_bg_var=0.1, debug_simple_obj_var=0.2,
        debug_match_appearance=False, debug_core_appearance=False,
        debug_appearance_dim=3, debug_bw=True,
        debug_no_latents=False, debug_no_reuse=False, debug_no_velocity=False,      # full_state ablations, stove.py:140-163
        action_conditioned=False, action_space=None,
        random_seed=42,
        align_corners=False,     # the runnable reference (torch >= 1.3); True = the torch 1.0.1 convention the reference was written for
    )
    for k, v in overrides.items():
        setattr(c, k, v)
    return c


# --------------------------------------------------------------------------
# region graph + RAT-SPN structure
#   region_graph.py:54-95 (random_split), :118-155 (make_layers)
#   rat_torch.py:279-331 (_make_spn_from_region_graph)
# --------------------------------------------------------------------------
def spn_structure(num_dims, seed, num_splits, depth):
    """Layered structure of the random SPN the reference builds with
    `num_splits` calls of `random_split(2, depth)` on `range(num_dims)`.

    Returns a dict:
      layers[0]            : list of leaf scopes (sorted tuples, lexicographic order)
      layers[odd]          : list of (layer_a, idx_a, layer_b, idx_b) product inputs,
                             in1 = lexicographically smaller child region
      layers[even > 0]     : list of lists of product indices (children of each sum,
                             in creation order), regions in lexicographic order
    The order of products inside a layer follows CPython's iteration order of
    the `set` of partition tuples, exactly as the reference gets it
    (region_graph.py:141); the same sequence of `set.add` calls reproduces it.
    """
    rng = np.random.RandomState(seed)
    root = tuple(range(num_dims))
    regions = {root}
    partitions = set()
    child_parts = {}

    def split(region, rec):
        # region_graph.py:54-95
        if rec < 1 or len(region) == 1:
            return
        perm = list(rng.permutation(list(region)))
        n_parts = min(len(perm), 2)
        q, r = divmod(len(perm), n_parts)
        parts, pos = [], 0
        for k in range(n_parts):
            inc = q + 1 if k < r else q
            sub = tuple(sorted(perm[pos:pos + inc]))
            parts.append(sub)
            regions.add(sub)
            pos += inc
        part = tuple(sorted(parts))
        if part not in partitions:
            partitions.add(part)
            child_parts[region] = child_parts.get(region, []) + [part]
        if rec > 1:
            for sub in part:
                split(sub, rec - 1)

    for _ in range(num_splits):
        split(root, depth)

    # make_layers, region_graph.py:118-155
    leaves = sorted(r for r in regions if r not in child_parts)
    layers_regions = [leaves]
    seen_r, seen_p = set(leaves), set()
    raw_layers = [leaves]
    while len(seen_r) != len(regions) or len(seen_p) != len(partitions):
        p_layer = [p for p in partitions
                   if p not in seen_p and all(r in seen_r for r in p)]
        raw_layers.append(p_layer)
        seen_p.update(p_layer)
        r_layer = sorted(r for r in regions if r not in seen_r
                         and all(p in seen_p for p in child_parts[r]))
        raw_layers.append(r_layer)
        seen_r.update(r_layer)

    # vectors, rat_torch.py:285-331
    where = {}                      # region -> (layer, idx) of its distribution vector
    for i, r in enumerate(leaves):
        where[r] = (0, i)
    layers = [[tuple(r) for r in leaves]]
    prods_of = {}
    for li in range(1, len(raw_layers)):
        if li % 2 == 1:
            cur = []
            for i, part in enumerate(raw_layers[li]):
                a, b = part[0], part[1]
                cur.append((where[a][0], where[a][1], where[b][0], where[b][1]))
                res = tuple(sorted(a + b))
                prods_of.setdefault(res, []).append(i)
            layers.append(cur)
        else:
            cur = []
            for i, region in enumerate(raw_layers[li]):
                cur.append(list(prods_of[region]))
                where[region] = (li, i)
            layers.append(cur)
            prods_of = {}
    return {'num_dims': num_dims, 'layers': layers,
            'root': where[root]}


def spn_forward(struct, params, prefix, x, marg, num_gauss, num_sums, vmin, vmax, child_values=False):
    """RatSpn.forward (rat_torch.py:354-357) -> (B, 1); with child_values also the per-sum-vector
    `child + log w` tensors (B, children, sums) of compute_activations(get_sum_child_acts=True), :333-352.

    params[prefix + 'vector_list.L.i.{means,sigma_params,params}'] as in the
    reference state dict.  Leaf: rat_torch.py:83-109 (the 'sigma' is a variance);
    product: :147-163; sum: :202-222.
    """
    layers = struct['layers']
    acts, kept = {}, {}
    if marg is not None:
        keep = 1.0 - torch.clamp(marg, 0.0, 1.0)
    for i, scope in enumerate(layers[0]):
        mu = params[f'{prefix}vector_list.0.{i}.means']
        rho = params[f'{prefix}vector_list.0.{i}.sigma_params']
        var = vmin + (vmax - vmin) * torch.sigmoid(rho)
        xs = x[:, list(scope)].unsqueeze(-1)                       # (B, S, 1)
        lp = -(xs - mu) ** 2 / (2.0 * var) - 0.5 * torch.log(var) - 0.5 * LOG_2PI
        if marg is not None:
            lp = lp * keep[:, list(scope)].unsqueeze(-1)
        acts[(0, i)] = lp.sum(1)                                   # (B, G)
    for li in range(1, len(layers)):
        if li % 2 == 1:
            for i, (la, ia, lb, ib) in enumerate(layers[li]):
                d1, d2 = acts[(la, ia)], acts[(lb, ib)]
                out = d1.unsqueeze(1) + d2.unsqueeze(2)            # [b, j2, j1]
                acts[(li, i)] = out.reshape(d1.shape[0], -1)
        else:
            for i, kids in enumerate(layers[li]):
                w = params[f'{prefix}vector_list.{li}.{i}.params']
                logw = torch.log_softmax(w, 0)
                child = torch.cat([acts[(li - 1, k)] for k in kids], 1)
                kept[(li, i)] = child.unsqueeze(-1) + logw
                acts[(li, i)] = torch.logsumexp(kept[(li, i)], 1)
    if child_values:
        return acts[struct['root']], kept
    return acts[struct['root']]


def spn_decode(struct, params, prefix, pick, node=0):
    """RatSpn.reconstruct(idxs, node, sample=False), rat_torch.py:359-372 with the per-node walks of
    :126-135 (leaf: component means on its scope), :177-183 (product: node = row * size1 + col, input 0 takes
    col, input 1 takes row), :224-229 (sum: child index -> walk the concatenated product vectors).
    `pick[(layer, i)]` = chosen child per sum node of that vector.  Returns the (num_dims,) float64 array."""
    layers = struct['layers']
    out = np.zeros(struct['num_dims'])

    def size(li, i):
        if li == 0:
            return params[f'{prefix}vector_list.0.{i}.means'].shape[1]
        if li % 2 == 1:
            la, ia, lb, ib = layers[li][i]
            return size(la, ia) * size(lb, ib)
        return params[f'{prefix}vector_list.{li}.{i}.params'].shape[1]

    def walk(li, i, n):
        if li == 0:
            out[list(layers[0][i])] = params[f'{prefix}vector_list.0.{i}.means'][:, n].detach().double().numpy()
        elif li % 2 == 1:
            la, ia, lb, ib = layers[li][i]
            s1 = size(la, ia)
            walk(la, ia, n % s1)
            walk(lb, ib, n // s1)
        else:
            k = int(pick[(li, i)][n])
            for pi in layers[li][i]:
                sz = size(li - 1, pi)
                if k < sz:
                    return walk(li - 1, pi, k)
                k -= sz
    walk(*struct['root'], node)
    return out


def spn_max_activation(struct, params, prefix, dtype):
    """Supair.spn_max_activation, supair.py:357-380: walk along the largest sum weights; clip to [0, 1]."""
    pick = {}
    for li in range(2, len(struct['layers']), 2):
        for i in range(len(struct['layers'][li])):
            pick[(li, i)] = np.argmax(params[f'{prefix}vector_list.{li}.{i}.params'].detach().numpy(), 0)
    return torch.as_tensor(np.clip(spn_decode(struct, params, prefix, pick), 0.0, 1.0)).to(dtype)


def spn_mpe(c, params, structs, z, x):
    """Supair.spn_mpe for the object SPN, supair.py:382-424.  z (nT, o, 4), x (nT, ch, H, W) -> (nT, o, ph*pw):
    per glimpse, the walk along argmax_k (child_k + log w_k) of every sum node (no marginalisation), clipped."""
    patches = glimpses(c, x, z.flatten(0, 1))
    _, kept = spn_forward(structs['obj'], params, 'sup.obj_spn.', patches.flatten(1), None,
                          c.obj_spn_num_gauss, c.obj_spn_num_sums, c.obj_min_var, c.obj_max_var, child_values=True)
    rec = []
    for j in range(patches.shape[0]):
        pick = {k: np.argmax(v[j].detach().numpy(), 0) for k, v in kept.items()}
        rec.append(spn_decode(structs['obj'], params, 'sup.obj_spn.', pick))
    rec = torch.as_tensor(np.clip(np.stack(rec), 0.0, 1.0)).to(z.dtype)
    return rec.view(x.shape[0], c.num_obj, -1)


def reconstruct_from_z(c, params, structs, z, x=None, max_activation=True, single_image=True):
    """Supair.reconstruct_from_z, supair.py:426-498.  z (n, T, o, >=4) -> frames (n, T, ch, W, H): the background
    SPN's max-activation image plus every object's patch (max-activation, or the MPE reconstruction of its glimpse
    of x) pasted through the inverse transform, clamped to [0, 1]."""
    z = z[..., :4]
    n, T, o = z.shape[:3]
    bg = spn_max_activation(structs['bg'], params, 'sup.bg_spn.', z.dtype).view(c.width, c.height)
    rec = bg.unsqueeze(0).repeat(n * T, 1, 1).unsqueeze(1)
    if max_activation:
        patch = spn_max_activation(structs['obj'], params, 'sup.obj_spn.', z.dtype).view(c.patch_width, c.patch_height)
        patches = patch.expand(n * T, o, c.channels, c.patch_width, c.patch_height)
    else:
        z_in, x_in = (z[:, 0], x) if single_image else (z.flatten(0, 1), x.flatten(0, 1))
        patches = spn_mpe(c, params, structs, z_in, x_in).view(z_in.shape[0], o, c.channels, c.patch_width, c.patch_height)
        if single_image:
            patches = patches.unsqueeze(1).repeat(1, T, 1, 1, 1, 1).flatten(0, 1)
    z_img = z.flatten(0, 1)
    for k in range(o):
        rec = rec + _sample(patches[:, k].contiguous(), _theta(_z_inverse(z_img[:, k])), c.width, c.height, _ac(c))
    return torch.clamp(rec.view(n, T, c.channels, c.width, c.height), 0, 1)


def obj_spn_structure(c):
    # probabilistic_models.py:8-22
    d = c.channels * c.patch_width * c.patch_height
    return spn_structure(d, c.random_seed, 6, 2)


def bg_spn_structure(c):
    # probabilistic_models.py:25-39
    d = c.width * c.height * c.channels
    return spn_structure(d, c.random_seed, 3, 1)


# --------------------------------------------------------------------------
# spatial transformer pieces (supair.py:193-239, 241-276, 278-356)
# torch >= 1.3 default align_corners=False is what the runnable reference
# computes (SURVEY.md section 7 'align_corners fork').
# --------------------------------------------------------------------------
def _theta(z):
    # expand_z, supair.py:193-216: [sx, sy, x, y] -> [[sx,0,x],[0,sy,y]]
    zero = torch.zeros_like(z[:, 0])
    return torch.stack([z[:, 0], zero, z[:, 2], zero, z[:, 1], z[:, 3]], 1).view(-1, 2, 3)


def _z_inverse(z):
    # invert_z, supair.py:218-239
    return torch.stack([1.0 / z[:, 0], 1.0 / z[:, 1], -z[:, 2] / z[:, 0], -z[:, 3] / z[:, 1]], 1)


def _sample(img, theta, out_h, out_w, ac=False):
    grid = F.affine_grid(theta, (img.shape[0], img.shape[1], out_h, out_w), align_corners=bool(ac))
    return F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=bool(ac))


def _ac(c):
    return bool(getattr(c, 'align_corners', False))


def glimpses(c, x_img, z_obj):
    """patches_from_z, supair.py:241-276: (nT,ch,H,W),(nT*N,4) -> (nT*N,ch,ph,pw)."""
    n_obj = z_obj.shape[0] // x_img.shape[0]
    x_rep = x_img.unsqueeze(1).expand(-1, n_obj, -1, -1, -1).reshape(-1, *x_img.shape[1:])
    return _sample(x_rep, _theta(z_obj), c.patch_width, c.patch_height, _ac(c))


def masks_from_z(c, z_img):
    """supair.py:278-356: z (nT,N,4) -> marg_patch (nT*N,ch,10,10), bg_mask (nT,ch,H,W), overlap (nT,N)."""
    n = z_img.shape[0]
    ones = z_img.new_ones(n, c.channels, c.width, c.height)
    bg = z_img.new_zeros(n, c.channels, c.width, c.height)
    per_obj = []
    for k in range(z_img.shape[1]):
        zk = z_img[:, k]
        seen = _sample(1.0 - bg, _theta(zk), c.patch_width, c.patch_height, _ac(c))
        per_obj.append(1.0 - seen)
        box = _sample(ones, _theta(_z_inverse(zk)), c.width, c.height, _ac(c))
        bg = torch.clamp(bg + box, 0, 1)
    marg = torch.stack(per_obj, 1)
    overlap = marg.flatten(2).mean(2)
    return marg.flatten(0, 1), bg, overlap


def simple_gauss(x, marg, mean, scale):
    """SimpleBG / SimpleObj.forward, probabilistic_models.py:42-90: Normal(mean, scale).log_prob per pixel, weighted by
    (1 - marg) as it comes (no clamp), summed per row -> (B, 1).  (`scale` is what the reference calls var.)"""
    lp = -(x - mean) ** 2 / (2.0 * scale * scale) - math.log(scale) - 0.5 * LOG_2PI
    return (lp * (1.0 - marg)).sum(1, keepdim=True)


def scene_likelihood(c, params, structs, x, z_obj, parts=False):
    """Supair.likelihood, supair.py:44-110.  x (n,T',ch,H,W), z_obj (n*T'*N,4) [sx,sy,x,y] -> (n*T',).
    config.debug_bg_model / debug_obj_spn (supair.py:33-42): the fixed-Gaussian models instead of the SPNs."""
    x_img = x.flatten(0, 1)
    z_img = z_obj.view(-1, c.num_obj, 4)
    marg_patch, marg_bg, overlap = masks_from_z(c, z_img)
    if getattr(c, 'debug_bg_model', False):
        bg_ll = simple_gauss(x_img.flatten(1), marg_bg.flatten(1), 0.0, c.debug_simple_bg_var)[:, 0]
    else:
        bg_ll = spn_forward(structs['bg'], params, 'sup.bg_spn.', x_img.flatten(1), marg_bg.flatten(1),
                            6, 3, c.bg_min_var, c.bg_max_var)[:, 0]
    patches = glimpses(c, x_img, z_obj)
    if getattr(c, 'debug_obj_spn', False):
        p_ll = simple_gauss(patches.flatten(1), marg_patch.flatten(1), 0.8, c.debug_simple_obj_var)[:, 0]
    else:
        p_ll = spn_forward(structs['obj'], params, 'sup.obj_spn.', patches.flatten(1), marg_patch.flatten(1),
                           c.obj_spn_num_gauss, c.obj_spn_num_sums, c.obj_min_var, c.obj_max_var)[:, 0]
    p_ll = (p_ll * z_obj[:, 0] * z_obj[:, 1]).view(-1, c.num_obj).sum(1)
    # Exponential(beta).log_prob(r) = log(beta) - beta r   (supair.py:84-85)
    ov_ll = (math.log(c.overlap_beta) - c.overlap_beta * overlap).sum(1)
    total = torch.stack([bg_ll, p_ll, ov_ll], -1).sum(-1)
    if parts:
        return total, bg_ll, p_ll, ov_ll
    return total


# --------------------------------------------------------------------------
# SuPAIR recognition side (encoder.py:28-57, supair.py:112-156)
# --------------------------------------------------------------------------
def encoder_forward(c, params, frames):
    """RnnStates.forward: the same flattened frame fed for num_obj LSTM steps."""
    xf = frames.flatten(1)
    w_ih, w_hh = params['sup.encoder.rnn.weight_ih_l0'], params['sup.encoder.rnn.weight_hh_l0']
    b = params['sup.encoder.rnn.bias_ih_l0'] + params['sup.encoder.rnn.bias_hh_l0']
    hid = w_hh.shape[1]
    h = xf.new_zeros(xf.shape[0], hid)
    cell = xf.new_zeros(xf.shape[0], hid)
    outs = []
    x_proj = xf @ w_ih.t() + b
    for _ in range(c.num_obj):
        gates = x_proj + h @ w_hh.t()
        i, f, g, o = gates.chunk(4, 1)
        cell = torch.sigmoid(f) * cell + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(cell)
        outs.append(h)
    hs = torch.stack(outs, 1)                                       # (nT, N, 256)
    y = torch.sigmoid(hs @ params['sup.encoder.fc1.weight'].t() + params['sup.encoder.fc1.bias'])
    return y @ params['sup.encoder.fc2.weight'].t() + params['sup.encoder.fc2.bias']


def constrain_zp(c, zp):
    # supair.py:112-149
    sig = torch.sigmoid(zp)
    mean = torch.stack([
        sig[:, 0] * (c.max_obj_scale - c.min_obj_scale) + c.min_obj_scale,
        sig[:, 1] * (c.max_y_scale - c.min_y_scale) + c.min_y_scale,
        (2 * sig[:, 2] - 1) * c.obj_pos_bound,
        (2 * sig[:, 3] - 1) * c.obj_pos_bound], 1)
    std = torch.cat([c.scale_var * sig[:, 4:6], c.pos_var * sig[:, 6:8]], 1)
    return mean, std


def sy_from_quotient(z):
    # supair.py:151-156
    return torch.cat([z[..., 0:1], z[..., 0:1] * z[..., 1:2], z[..., 2:]], -1)


# --------------------------------------------------------------------------
# object matching + glitch smoothing (stove.py:200-329, 432-514, 516-571)
# --------------------------------------------------------------------------
def _match_features(c, z_sup, z_sup_std, app):
    z = (z_sup + 1) / 2
    cols = [2, 3]
    if app is not None:
        z = torch.cat([z, app], -1)
        if c.debug_match_appearance:
            cols += [4, 5, 6]
    if z_sup_std is not None:
        z = torch.cat([z, z_sup_std], -1)
    return z, cols


def _split_matched(zm, z_sup_std, app):
    z_sup = 2 * zm[..., :4] - 1
    off = 4
    app_m = None
    if app is not None:
        app_m = zm[..., 4:7]
        off = 7
    std_m = zm[..., off:off + 4] if z_sup_std is not None else None
    return z_sup, std_m, app_m


def match_3only(c, z_sup, z_sup_std=None, app=None):
    """_3_only_match_objects, stove.py:200-329 (N must be 3)."""
    z, cols = _match_features(c, z_sup, z_sup_std, app)
    n_obj = c.num_obj
    out = [z[:, 0]]
    for t in range(1, z.shape[1]):
        cur = z[:, t][..., cols].detach()                           # (n, o, d)
        prev = out[-1][..., cols].detach()
        # err[n, i, j] = |prev_i - cur_j|^2 ; row i picks its nearest current object
        err = ((prev.unsqueeze(2) - cur.unsqueeze(1)) ** 2).sum(-1)
        idx = err.argmin(-1)
        ok = (idx[:, 0] != idx[:, 1]) & (idx[:, 1] != idx[:, 2]) & (idx[:, 0] != idx[:, 2])
        bad = ~ok
        if bad.any():
            fe = err[bad].clone()
            fixed = torch.zeros(fe.shape[0], n_obj, dtype=torch.long)
            for o in range(n_obj):
                f_idx = fe.argmin(-1)                               # per row minima
                s = f_idx[:, o]
                fixed[:, o] = s
                fe[torch.arange(fe.shape[0]), :, s] = 1e12          # knock the column out
            idx = idx.clone()
            idx[bad] = fixed
        out.append(torch.gather(z[:, t], 1, idx.unsqueeze(-1).expand(-1, -1, z.shape[-1])))
    return _split_matched(torch.stack(out, 1), z_sup_std, app)


def match_greedy(c, z_sup, z_sup_std=None, app=None):
    """_greedy_match_objects, stove.py:432-514."""
    z, cols = _match_features(c, z_sup, z_sup_std, app)
    zm = torch.zeros_like(z)
    zm[:, 0] = z[:, 0]
    rows = torch.arange(z.shape[0])
    for t in range(1, z.shape[1]):
        cur = z[:, t][..., cols].detach()
        prev = zm[:, t - 1][..., cols].detach()
        err = ((prev.unsqueeze(2) - cur.unsqueeze(1)) ** 2).sum(-1)
        perm = torch.zeros_like(err)
        for _ in range(z.shape[2]):
            flat = err.view(err.shape[0], -1).argmin(1)
            ix, iy = flat // z.shape[2], flat % z.shape[2]
            perm[rows, ix, iy] = 1.0
            err[rows, ix, :] = err.max() + 1
            err[rows, :, iy] = err.max() + 1
        zm[:, t] = perm @ z[:, t]
    return _split_matched(zm, z_sup_std, app)


def fix_supair(z, z_std):
    """stove.py:516-571: glitch = |dz| to previous AND next frame > 0.095 on dims 0,1."""
    zz = torch.cat([z, z_std], -1).clone()
    d = (zz[:, 1:, :, :2] - zz[:, :-1, :, :2]).abs().detach()
    pad = torch.zeros_like(d[:, :1])
    hit = (torch.cat([pad, d], 1) > 0.095) & (torch.cat([d, pad], 1) > 0.095)
    hit = torch.cat(zz.shape[-1] // 2 * [hit], -1)
    smooth = torch.zeros_like(zz)
    smooth[:, 1:-1] = (zz[:, :-2] + zz[:, 2:]) / 2
    zz = torch.where(hit, smooth, zz)
    return zz[..., :4], zz[..., 4:]


def v_from_state(z_sup):
    # stove.py:54-80
    v = z_sup[:, 1:, :, 2:] - z_sup[:, :-1, :, 2:]
    full = torch.cat([z_sup[:, 1:], v], -1)
    return torch.cat([torch.zeros_like(full[:, :1]), full], 1)


def v_std_from_pos(z_sup_std):
    # stove.py:82-101
    vs = torch.sqrt(z_sup_std[:, 1:, :, 2:] ** 2 + z_sup_std[:, :-1, :, 2:] ** 2)
    full = torch.cat([z_sup_std[:, 1:], vs], -1)
    return torch.cat([torch.zeros_like(full[:, :1]), full], 1)


# --------------------------------------------------------------------------
# dynamics core (dynamics.py:147-265)
# --------------------------------------------------------------------------
def _lin(params, name, x):
    return x @ params[name + '.weight'].t() + params[name + '.bias']


def _phi(c):
    # dynamics.py:109 -- the selection is inverted: the default 'relu' gives leaky_relu(0.01)
    return F.elu if c.debug_nonlinear == 'leaky_relu' else F.leaky_relu


def dynamics_forward(c, params, s, actions=None, app=None, lim_enc=2, core=0, with_pred=False):
    """Dynamics.forward + core, dynamics.py:181-265.  s (B,N,16) -> (B,N,32), reward."""
    phi = _phi(c)
    n_obj = s.shape[1]
    if actions is not None:
        emb = _lin(params, 'dyn.action_embedding_layer', actions).view(s.shape[0], n_obj, 4)
        s = torch.cat([s, emb], -1)
    if app is not None:
        s = torch.cat([s, app], -1)
    s = torch.cat([s[..., :lim_enc], _lin(params, 'dyn.state_enc', s)[..., lim_enc:]], -1)

    h = phi(_lin(params, f'dyn.self_cores.{core}.0', s))
    self_dyn = _lin(params, f'dyn.self_cores.{core}.1', h) + h

    a1 = s.unsqueeze(2).expand(-1, -1, n_obj, -1)                   # [b,i,j] = s_i
    a2 = s.unsqueeze(1).expand(-1, n_obj, -1, -1)                   # [b,i,j] = s_j
    dist = ((a1[..., 0] - a2[..., 0]) ** 2 + (a1[..., 1] - a2[..., 1]) ** 2).unsqueeze(-1)
    comb = torch.cat([a1, a2, dist], -1)
    r = phi(_lin(params, f'dyn.rel_cores.{core}.0', comb))
    r = phi(_lin(params, f'dyn.rel_cores.{core}.1', r))
    rel = _lin(params, f'dyn.rel_cores.{core}.2', r) + r
    a = phi(_lin(params, f'dyn.att_net.{core}.0', comb))
    a = phi(_lin(params, f'dyn.att_net.{core}.1', a))
    att = torch.exp(_lin(params, f'dyn.att_net.{core}.2', a))
    off_diag = (1 - torch.eye(n_obj, dtype=s.dtype)).view(1, n_obj, n_obj, 1)
    rel_dyn = (rel * off_diag * att).sum(2)
    pred = self_dyn + rel_dyn

    f1 = torch.tanh(_lin(params, f'dyn.affector.{core}.0', pred))
    f2 = torch.tanh(_lin(params, f'dyn.affector.{core}.1', f1)) + f1
    f3 = _lin(params, f'dyn.affector.{core}.2', f2)
    o1 = torch.tanh(_lin(params, f'dyn.out.{core}.0', torch.cat([f3, s], 2)))
    result = _lin(params, f'dyn.out.{core}.1', o1) + o1

    reward = 0
    if c.action_conditioned:
        q = _lin(params, 'dyn.reward_head0.2', torch.relu(_lin(params, 'dyn.reward_head0.0', pred))).sum(1)
        q = torch.relu(_lin(params, 'dyn.reward_head1.0', q))
        q = torch.relu(_lin(params, 'dyn.reward_head1.2', q))
        reward = torch.sigmoid(_lin(params, 'dyn.reward_head1.4', q).view(-1, 1))
    if with_pred:
        return result, reward, pred
    return result, reward


def constrain_z_dyn(c, z, z_std=None):
    # dynamics.py:147-179
    zc = 2 * torch.sigmoid(z) - 1
    if z_std is None:
        return zc, None
    sg = torch.sigmoid(z_std)
    std = torch.cat([c.pos_var * sg[..., :2], 0.04 * sg[..., 2:4], c.debug_latent_q_std * sg[..., 4:]], -1)
    return zc, std


def transition_std(c):
    # dynamics.py:112-120
    std = list(c.transition_lik_std)
    if len(std) == 4:
        std = std + 12 * [0.01]
    if len(std) != c.cl // 2:
        raise ValueError('Specify valid transition_lik_std.')
    return std


def normal_log_prob(x, mean, std):
    return -((x - mean) ** 2) / (2 * std ** 2) - torch.log(std) - 0.5 * LOG_2PI


# --------------------------------------------------------------------------
# Stove.forward (stove.py:599-775, 863-897) with injected noise
# --------------------------------------------------------------------------
def bw_transform(x):
    # utils.py:10-15
    return torch.clamp(x.sum(2), 0, 1).unsqueeze(2)


def draw_eps(B, N, T, cl=32, skip=2, generator=None, dtype=torch.float32):
    """Standard-normal draws in the reference's order (stove.py:667, 679, 146/167):
    latent prior (B,N,cl/2-4,1), std prior (same), then one (B,N,cl/2+2) per t = skip..T-1."""
    g = generator
    eps = {'latent': torch.randn(B, N, cl // 2 - 4, 1, generator=g, dtype=dtype),
           'std': torch.randn(B, N, cl // 2 - 4, 1, generator=g, dtype=dtype),
           'steps': [torch.randn(B, N, cl // 2 + 2, generator=g, dtype=dtype) for _ in range(skip, T)]}
    return eps


def object_embedding(c, z, x_color):
    # stove.py:573-597 -- mean colour of each object's glimpse of the colour frame
    zp = sy_from_quotient(z[..., :4].detach())
    pat = glimpses(c, x_color.flatten(0, 1), zp.flatten(0, 2))
    return pat.mean((-1, -2)).view(*z.shape[:-1], 3)


def stove_forward(c, params, structs, x_color, eps, actions=None, detail=False, code_values=None):
    """Stove.forward -> stove_forward.  x_color (B,T,3,H,W) in [0,1]; eps from draw_eps.
    code_values (BT, N, 8): evaluate everything behind the recognition network AT these codes (the gradient still flows into the
    recognition network as usual) -- the reference's numbers at another implementation's float32 codes, for fixtures whose
    gradients amplify a 1e-5 difference of the codes a thousandfold (tests: the 'stress' weight regime)."""
    x = bw_transform(x_color) if c.debug_bw else x_color
    B, T = x.shape[:2]
    N, cl, skip = c.num_obj, c.cl, c.skip

    code = encoder_forward(c, params, x.flatten(0, 1))              # (BT, N, 8)
    if code_values is not None:
        code = code + (code_values.to(code.dtype).view_as(code) - code.detach())
    zs, zs_std = constrain_zp(c, code.flatten(0, 1))
    zs, zs_std = zs.view(B, T, N, 4), zs_std.view(B, T, N, 4)

    app = None
    if c.debug_core_appearance or c.debug_match_appearance:
        app = object_embedding(c, zs, x_color)
    matcher = {'3_only': match_3only, 'greedy': match_greedy}[c.debug_match_objects]
    if c.debug_match_objects == '3_only' and N != 3:
        raise ValueError('Matching Function not compatible w/ specified number of objects.')
    zs, zs_std, app = matcher(c, zs, zs_std, app)
    if c.debug_fix_supair:
        zs, zs_std = fix_supair(zs, zs_std)
    zs_full = v_from_state(zs)                                      # (B,T,N,6)
    zs_std_full = v_std_from_pos(zs_std)

    # initial state, stove.py:663-685
    lat0 = (0.0 + 0.01 * eps['latent']).squeeze()
    std0 = (0.1 + 0.01 * eps['std']).squeeze()
    z = {skip - 1: torch.cat([zs_full[:, skip - 1], lat0], -1)}
    dyn_std0 = torch.cat([zs_std_full[:, skip - 1, :, 2:], std0], -1)
    tstd = torch.tensor(transition_std(c), dtype=x.dtype).view(1, 1, -1)

    z_l, zdyn_l, zdyn_std_l, logq_l, zstd_l, rewards = [], [], [], [], [], []
    for t in range(skip, T):
        act = actions[:, t - 1] if actions is not None else None
        ap = app[:, t - 1] if (app is not None and c.debug_core_appearance) else None
        out, rew = dynamics_forward(c, params, z[t - 1][..., 2:], act, ap)
        rewards.append(rew)
        m, sd = constrain_z_dyn(c, out[..., :cl // 2], out[..., cl // 2:])
        zdyn = torch.cat([z[t - 1][..., 2:4] + m[..., :2], m[..., 2:]], -1)
        # full_state, stove.py:103-170 (default flags)
        ms, ss = zs_full[:, t], zs_std_full[:, t]
        s_d, s_s = sd[..., :4], ss[..., 2:6]
        mean_xv = (s_s ** 2 * zdyn[..., :4] + s_d ** 2 * ms[..., 2:6]) / (s_d ** 2 + s_s ** 2)
        std_xv = s_d * s_s / torch.sqrt(s_d ** 2 + s_s ** 2)
        if c.debug_no_latents:                 # stove.py:140-148: q(z) over the six SuPAIR dimensions only, latents are zeros
            mean = torch.cat([ms[..., :2], mean_xv], -1)
            std = torch.cat([ss[..., :2], std_xv], -1)
        elif c.debug_no_velocity:              # stove.py:154-160: velocities ~ N(0, 1) instead of the fused estimate
            mean = torch.cat([ms[..., :2], mean_xv[..., :2], torch.zeros_like(mean_xv[..., 2:]), zdyn[..., 4:]], -1)
            std = torch.cat([ss[..., :2], std_xv[..., :2], torch.ones_like(std_xv[..., 2:]), sd[..., 4:]], -1)
        else:                                  # (debug_no_reuse, stove.py:151-153, is overwritten by the if / else that follows it
            mean = torch.cat([ms[..., :2], mean_xv, zdyn[..., 4:]], -1)          # in the reference: the flag changes nothing)
            std = torch.cat([ss[..., :2], std_xv, sd[..., 4:]], -1)
        zt = mean + std * eps['steps'][t - skip][..., :mean.shape[-1]]
        lq_t = normal_log_prob(zt, mean, std)
        if c.debug_no_latents:
            zt = torch.cat([zt, torch.zeros_like(zdyn[..., 4:])], -1)
        z[t] = zt
        z_l.append(zt); zdyn_l.append(zdyn); zdyn_std_l.append(sd); zstd_l.append(std)
        logq_l.append(lq_t)

    z_s = torch.stack(z_l, 1)                                       # (B,T-2,N,18)
    zdyn_s = torch.stack(zdyn_l, 1)
    logq = torch.stack(logq_l, 1).sum((-2, -1)).flatten()
    z_f = sy_from_quotient(z_s.flatten(0, 2))
    img_lik = scene_likelihood(c, params, structs, x[:, skip:], z_f[..., :4])
    z_sup1 = sy_from_quotient(zs[:, 1:skip])
    img_lik_sup = scene_likelihood(c, params, structs, x[:, 1:skip], z_sup1.flatten(0, 2))
    trans = normal_log_prob(z_s[..., 2:], zdyn_s, tstd).sum((-2, -1)).flatten(0, 1)
    elbo = torch.mean(trans + img_lik - logq) + torch.mean(img_lik_sup)

    if c.action_conditioned:
        rewards = torch.stack(rewards, 1)
    else:
        rewards = torch.tensor([float(r) for r in rewards])
    if not detail:
        return elbo, rewards
    info = {
        'z': sy_from_quotient(z_s), 'z_dyn': zdyn_s,
        'z_sup': sy_from_quotient(zs_full[:, skip:]),
        'z_std': torch.stack(zstd_l, 1).mean((0, 1, 2)),
        'z_dyn_std': torch.stack(zdyn_std_l, 1)[..., :4].mean((0, 1, 2)),
        'z_sup_std': zs_std_full[:, skip:].mean((0, 1, 2)),
        'log_q': logq.mean(), 'translik': trans.mean(),
        'img_lik': img_lik, 'img_lik_sup': img_lik_sup,
        'obj_appearances': app[:, skip:] if app is not None else None,
    }
    return elbo, rewards, info


def rollout(c, params, z_last, num, actions=None, appearance=None, eps=None):
    """Stove.rollout, stove.py:777-861.  z_last (B,N,18) with [sx,sy,...].  eps None: mean prediction -> (z, rewards);
    eps = list of `num` (B,N,16) standard-normal draws: the sampling branch (:833-838, `Normal(mean, std).rsample()`
    fed back) -> (z, log_q, rewards)."""
    cl = c.cl
    scale = z_last[..., :2]
    z = [z_last]
    rewards, log_qs = [], []
    for t in range(1, num + 1):
        act = actions[:, (t - 1) % actions.shape[1]] if actions is not None else None
        out, rew = dynamics_forward(c, params, z[-1][..., 2:], act, appearance)
        rewards.append(rew)
        m, sd = constrain_z_dyn(c, out[..., :cl // 2], out[..., cl // 2:])
        nxt = torch.cat([z[-1][..., 2:4] + m[..., :2], m[..., 2:]], -1)
        if eps is not None:
            mean = nxt
            nxt = mean + sd * eps[t - 1]
            log_qs.append(normal_log_prob(nxt, mean, sd))
        z.append(torch.cat([scale, nxt], -1))
    if c.action_conditioned:
        rewards = torch.stack(rewards, 1)
    else:
        rewards = torch.tensor([float(r) for r in rewards])
    if eps is not None:
        return torch.stack(z[1:], 1), torch.stack(log_qs, 1), rewards
    return torch.stack(z[1:], 1), rewards


# --------------------------------------------------------------------------
# parameter inventory (state-dict names and shapes of the reference model;
# SURVEY.md section 5 'Checkpoint / resume')
# --------------------------------------------------------------------------
def spn_param_shapes(struct, prefix, num_gauss, num_sums):
    shapes = {}
    layers = struct['layers']
    size = {}
    for i, scope in enumerate(layers[0]):
        shapes[f'{prefix}vector_list.0.{i}.means'] = (len(scope), num_gauss)
        shapes[f'{prefix}vector_list.0.{i}.sigma_params'] = (len(scope), num_gauss)
        size[(0, i)] = num_gauss
    for li in range(1, len(layers)):
        last = li == len(layers) - 1
        for i, item in enumerate(layers[li]):
            if li % 2 == 1:
                size[(li, i)] = size[(item[0], item[1])] * size[(item[2], item[3])]
            else:
                n_in = sum(size[(li - 1, k)] for k in item)
                n_out = 1 if last else num_sums
                shapes[f'{prefix}vector_list.{li}.{i}.params'] = (n_in, n_out)
                size[(li, i)] = n_out
    return shapes


def param_shapes(c, structs):
    cl = c.cl
    img = c.channels * c.width * c.height
    shapes = {
        'sup.encoder.rnn.weight_ih_l0': (1024, img), 'sup.encoder.rnn.weight_hh_l0': (1024, 256),
        'sup.encoder.rnn.bias_ih_l0': (1024,), 'sup.encoder.rnn.bias_hh_l0': (1024,),
        'sup.encoder.fc1.weight': (50, 256), 'sup.encoder.fc1.bias': (50,),
        'sup.encoder.fc2.weight': (8, 50), 'sup.encoder.fc2.bias': (8,),
    }
    shapes.update(spn_param_shapes(structs['obj'], 'sup.obj_spn.', c.obj_spn_num_gauss, c.obj_spn_num_sums))
    shapes.update(spn_param_shapes(structs['bg'], 'sup.bg_spn.', 6, 3))
    enc_in = cl // 2
    if c.action_conditioned:
        enc_in += 4
        shapes['dyn.action_embedding_layer.weight'] = (c.num_obj * 4, c.action_space)
        shapes['dyn.action_embedding_layer.bias'] = (c.num_obj * 4,)
        for nm, (o, i) in {'reward_head0.0': (cl, cl), 'reward_head0.2': (cl, cl),
                           'reward_head1.0': (cl // 2, cl), 'reward_head1.2': (cl // 4, cl // 2),
                           'reward_head1.4': (1, cl // 4)}.items():
            shapes[f'dyn.{nm}.weight'] = (o, i)
            shapes[f'dyn.{nm}.bias'] = (o,)
    if c.debug_core_appearance:
        enc_in += c.debug_appearance_dim
    shapes['dyn.state_enc.weight'] = (cl, enc_in)
    shapes['dyn.state_enc.bias'] = (cl,)
    dims = {'self_cores': [(cl, cl), (cl, cl)],
            'rel_cores': [(2 * cl, 2 * cl + 1), (cl, 2 * cl), (cl, cl)],
            'att_net': [(2 * cl, 2 * cl + 1), (cl, 2 * cl), (1, cl)],
            'affector': [(cl, cl), (cl, cl), (cl, cl)],
            'out': [(cl, 2 * cl), (cl, cl)]}
    for grp, lst in dims.items():
        for core in range(3):
            for j, (o, i) in enumerate(lst):
                shapes[f'dyn.{grp}.{core}.{j}.weight'] = (o, i)
                shapes[f'dyn.{grp}.{core}.{j}.bias'] = (o,)
    return shapes


def build_structs(c):
    return {'obj': obj_spn_structure(c), 'bg': bg_spn_structure(c)}


# --------------------------------------------------------------------------
# remaining entry points of the path: the non-permutation matcher and the SuPAIR-only ELBO
# --------------------------------------------------------------------------
def match_volatile(c, z_sup, z_sup_std=None, app=None):
    """_volatile_match_objects, stove.py:331-430: every current object goes to its nearest previous slot."""
    z, cols = _match_features(c, z_sup, z_sup_std, app)
    out = [z[:, 0]]
    n_obj = z.shape[2]
    for t in range(1, z.shape[1]):
        cur = z[:, t][..., cols].detach()
        prev = out[-1][..., cols].detach()
        # err[n, j, a] = |prev_a - cur_j|^2 ; column j (current object) picks its row-wise minimum over a
        err = ((prev.unsqueeze(1) - cur.unsqueeze(2)) ** 2).sum(-1)
        col_idx = err.argmin(-2).flatten()
        flat = torch.arange(col_idx.shape[0]) * n_obj + col_idx
        perm = torch.zeros(z.shape[0] * n_obj * n_obj, dtype=z.dtype)
        perm[flat] = 1
        out.append(perm.view(z.shape[0], n_obj, n_obj) @ z[:, t])
    return _split_matched(torch.stack(out, 1), z_sup_std, app)


def supair_forward(c, params, structs, x, eps):
    """Supair.forward, supair.py:504-551: ELBO of SuPAIR alone; eps (n*T*N, 4) standard-normal draws."""
    code = encoder_forward(c, params, x.flatten(0, 1))
    mean, std = constrain_zp(c, code.flatten(0, 1))
    z = mean + std * eps
    log_q = normal_log_prob(z, mean, std).sum(-1).view(-1, c.num_obj).sum(-1)
    log_p = scene_likelihood(c, params, structs, x, sy_from_quotient(z))
    return torch.mean(log_p - log_q), sy_from_quotient(z).view(*x.shape[:2], c.num_obj, 4), log_q.mean()
