"""SuPAIR scene model (sum-product attend-infer-repeat with a fixed number of objects).

Module/API surface of the reference's model/video_prediction/supair.py (`Supair`, :14-551).
`likelihood` -- 91-97 % of the reference's forward time -- is ONE fused HIP pipeline
(stove_amd/csrc/scene.hip + spn_obj.hip + spn_bg.hip): glimpse extraction, occlusion masks,
both SPN sweeps, patch scaling and the overlap prior, with an analytic backward to z and to
every SPN parameter.  The stand-alone `patches_from_z` / `masks_from_z` API methods (debug
plots, appearance embedding; not on the hot path) stay PyTorch-ROCm host code.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions import Normal

from .. import ops
from ..spn import probabilistic_models as prob
from . import encoder


class Supair(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.c = config
        self.step_counter = 0          # set by the calling trainer
        self.prop_dict = {}            # exported metrics
        self.encoder = encoder.RnnStates(self.c)
        # the fixed-Gaussian debug models of the reference (supair.py:33-42) are plain objects, not sub-modules, as there
        self.obj_spn = prob._get_simple_obj(self.c) if self.c.debug_obj_spn else prob._get_obj_spn(self.c, seed=self.c.random_seed)
        self.bg_spn = prob._get_simple_bg(self.c) if self.c.debug_bg_model else prob._get_bg_spn(self.c, seed=self.c.random_seed)

    # ------------------------------------------------------------------ likelihood
    def likelihood(self, x, z_obj, log_from=0):
        """log p(x, z) per frame.

        x (n, T, c, w, h) frames; z_obj (n*T*O, 4) = [sx, sy, x, y] -> (n*T,), prop_dict.
        log p = bgSPN(x | mask) + sum_k objSPN(glimpse_k | occlusion_k) sx_k sy_k
                + sum_k log Exponential(overlap_beta)(overlap_k)      (reference supair.py:44-110)
        `log_from` (build addition) restricts the logged part means to frames x[:, log_from:].
        """
        if self.c.channels != 1:
            raise NotImplementedError('SPN kernels are built for single-channel frames')
        if self.obj_spn._kind != 'obj' or self.bg_spn._kind != 'bg' or self.c.patch_width != 10 or self.c.patch_height != 10:
            # [amd] other glimpse sizes / SPN vector widths (config.patch_width, patch_height, obj_spn_num_gauss, obj_spn_num_sums): the
            # reference's op sequence with the general-size SPN operators (csrc/spn_obj_generic.hip, spn_bg_generic.hip); the fused
            # scene pipeline is instantiated for 10 x 10 glimpses and the default 10 / 10 widths
            return self._likelihood_general(x, z_obj, log_from)
        geom = None
        if x.shape[-1] != 32 or x.shape[-2] != 32 or bool(getattr(self.c, 'align_corners', False)):
            # [amd] any other frame size (the reference's stock gravity / multibilliards data are 50 x 50, envs.py:771-773, 841-844)
            # and the torch-1.0.1 sampling convention (align_corners=True): the same fused pipeline with the geometry at run time and
            # the general-size background operator (stove_scene_fwd_any).  config.scene_composed = True: the reference's own op sequence
            # on ATen's sampler instead (_likelihood_general, the cross-check of the tests; ~10x slower).
            if getattr(self.c, 'scene_composed', False):
                return self._likelihood_general(x, z_obj, log_from)
            geom = (int(x.shape[-1]), int(x.shape[-2]), bool(getattr(self.c, 'align_corners', False)))
        frames = x.flatten(start_dim=2)                 # (n, T', 1024) view: a time-slice of longer clips is NOT copied (ops._SceneFn)
        arena = getattr(self, '_arena', None)
        if arena is not None and arena.has_spn:         # flat parameter arena: one bake launch, gradients sunk
            obj_tabs, bg_tabs = arena.spn_tables()
            sink = arena.spn_sink
        else:
            obj_tabs, bg_tabs, sink = self.obj_spn.tables(), self.bg_spn.tables(), None
        log_p_xz, parts = ops.scene_likelihood(
            frames, z_obj.reshape(-1, 4), obj_tabs, bg_tabs, self.c.num_obj, self.c.overlap_beta, sink, geom)
        if ((self.step_counter % self.c.print_every == 0)
                or (self.step_counter % self.c.plot_every == 0)):
            if self.c.debug:
                # `log_from`: first time index whose frames enter the logged bg/patch/overlap means
                m = parts.view(x.shape[0], x.shape[1], 3)[:, log_from:].mean((0, 1))
                self.prop_dict['bg'] = m[0]
                self.prop_dict['patch'] = m[1]
                self.prop_dict['overlap'] = m[2]
        return log_p_xz, self.prop_dict

    def _likelihood_general(self, x, z_obj, log_from=0):
        """Supair.likelihood as the reference composes it (supair.py:44-110) for any frame size / sampling convention: masks and
        glimpses through the spatial-transformer API below (PyTorch-ROCm affine_grid / grid_sample on the GPU, differentiable in z),
        both SPN sweeps on the HIP operators (ops.objspn_apply; ops.bgspn_apply with the general-size kernels of
        csrc/spn_bg_generic.hip), scaling, overlap prior and the sum as written there."""
        c = self.c
        if not x.is_cuda:
            raise RuntimeError('stove_amd HIP ops need tensors on a GPU (cuda:N); there is no CPU path')
        x_img = x.flatten(end_dim=1)
        z_obj = z_obj.reshape(-1, 4)
        z_img = z_obj.view(-1, c.num_obj, 4)
        marg_patch, marg_bg, overlap = self.masks_from_z(z_img)
        bg_ll = self.bg_spn.forward(x_img.flatten(start_dim=1), marg_bg.flatten(start_dim=1))[:, 0]
        patches = self.patches_from_z(x_img, z_obj)
        patch_ll = self.obj_spn.forward(patches.flatten(start_dim=1), marg_patch.flatten(start_dim=1))[:, 0]
        patch_ll = (patch_ll * z_obj[:, 0] * z_obj[:, 1]).view(-1, c.num_obj).sum(1)
        overlap_ll = (math.log(c.overlap_beta) - c.overlap_beta * overlap).sum(1)          # Exponential(beta).log_prob
        log_p_xz = torch.stack([bg_ll, patch_ll, overlap_ll], -1).sum(-1)
        if ((self.step_counter % c.print_every == 0) or (self.step_counter % c.plot_every == 0)) and c.debug:
            sel = slice(None) if log_from == 0 else None
            if sel is None:
                keep = torch.zeros(x.shape[0], x.shape[1], dtype=torch.bool, device=x.device)
                keep[:, log_from:] = True
                sel = keep.flatten()
            self.prop_dict['bg'] = bg_ll[sel].mean().detach()
            self.prop_dict['patch'] = patch_ll[sel].mean().detach()
            self.prop_dict['overlap'] = overlap_ll[sel].mean().detach()
        return log_p_xz, self.prop_dict

    # ------------------------------------------------------------------ state codes
    def constrain_zp(self, zp):
        """(nTo, 8) raw codes -> mean, std (nTo, 4) of [sx, sy/sx, x, y] (reference supair.py:112-149)."""
        c = self.c
        key = (str(zp.device), zp.dtype, c.max_obj_scale, c.min_obj_scale, c.max_y_scale, c.min_y_scale,
               c.obj_pos_bound, c.scale_var, c.pos_var)
        if getattr(self, '_zp_consts_key', None) != key:          # constants live on the device, built once
            span = [c.max_obj_scale - c.min_obj_scale, c.max_y_scale - c.min_y_scale,
                    2 * c.obj_pos_bound, 2 * c.obj_pos_bound, c.scale_var, c.scale_var, c.pos_var, c.pos_var]
            low = [c.min_obj_scale, c.min_y_scale, -c.obj_pos_bound, -c.obj_pos_bound, 0.0, 0.0, 0.0, 0.0]
            self._zp_consts = (zp.new_tensor(span), zp.new_tensor(low))
            self._zp_consts_key = key
        span, low = self._zp_consts
        out = torch.addcmul(low, torch.sigmoid(zp), span)
        return out[:, :4], out[:, 4:]

    def zp_span_low(self):
        """The 8 spans and 8 lows of constrain_zp as host floats (the fused state pipeline's constants)."""
        c = self.c
        span = [c.max_obj_scale - c.min_obj_scale, c.max_y_scale - c.min_y_scale,
                2 * c.obj_pos_bound, 2 * c.obj_pos_bound, c.scale_var, c.scale_var, c.pos_var, c.pos_var]
        low = [c.min_obj_scale, c.min_y_scale, -c.obj_pos_bound, -c.obj_pos_bound, 0.0, 0.0, 0.0, 0.0]
        return span + low

    @staticmethod
    def sy_from_quotient(z):
        """[sx, sy/sx, ...] -> [sx, sy, ...]."""
        return torch.cat([z[..., 0:1], z[..., 0:1] * z[..., 1:2], z[..., 2:]], -1)

    @staticmethod
    def quotient_from_sy(z):
        """[sx, sy, ...] -> [sx, sy/sx, ...]."""
        return torch.cat([z[..., 0:1], z[..., 1:2] / z[..., 0:1], z[..., 2:]], -1)

    def get_z_sup_sample(self, zp_mean, zp_std):
        """Reparameterised sample of q(z|x) and its log-density summed over the 4 dims."""
        dist = Normal(zp_mean, zp_std)
        z = dist.rsample()
        return self.sy_from_quotient(z), dist.log_prob(z).sum(-1)

    # ------------------------------------------------------------------ spatial transformer API
    @staticmethod
    def expand_z(z):
        """[sx, sy, x, y] -> affine matrices [[sx, 0, x], [0, sy, y]]  (nTo, 2, 3)."""
        zero = torch.zeros_like(z[:, 0])
        return torch.stack([z[:, 0], zero, z[:, 2], zero, z[:, 1], z[:, 3]], 1).view(-1, 2, 3)

    @staticmethod
    def invert_z(z):
        """Parameters of the inverse transform: [1/sx, 1/sy, -x/sx, -y/sy]."""
        return torch.stack([1.0 / z[:, 0], 1.0 / z[:, 1], -z[:, 2] / z[:, 0], -z[:, 3] / z[:, 1]], 1)

    def _sample(self, img, theta, h_out, w_out):
        ac = bool(getattr(self.c, 'align_corners', False))
        grid = F.affine_grid(theta, (img.shape[0], img.shape[1], h_out, w_out), align_corners=ac)
        return F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=ac)

    def patches_from_z(self, x_img, z_obj):
        """(nT, c, w, h), (nT*o, 4) -> glimpses (nT*o, c, patch_w, patch_h)."""
        o = z_obj.shape[0] // x_img.shape[0]
        x_obj = x_img.unsqueeze(1).expand(-1, o, -1, -1, -1).reshape(-1, *x_img.shape[1:])
        return self._sample(x_obj, self.expand_z(z_obj), self.c.patch_width, self.c.patch_height)

    def masks_from_z(self, z_img):
        """(nT, o, 4) -> marginalisation masks: per-glimpse (nT*o, c, pw, ph), background
        (nT, c, w, h) and overlap ratios (nT, o).  Objects are processed in order; each sees the
        boxes pasted by the earlier ones (and everything outside the frame) as marginalised."""
        c = self.c
        n = z_img.shape[0]
        ones = z_img.new_ones(n, c.channels, c.width, c.height)
        bg = z_img.new_zeros(n, c.channels, c.width, c.height)
        per_obj = []
        for k in range(z_img.shape[1]):
            zk = z_img[:, k]
            per_obj.append(1.0 - self._sample(1.0 - bg, self.expand_z(zk), c.patch_width, c.patch_height))
            bg = torch.clamp(bg + self._sample(ones, self.expand_z(self.invert_z(zk)), c.width, c.height), 0, 1)
        marg = torch.stack(per_obj, 1)
        return marg.flatten(end_dim=1), bg, marg.flatten(start_dim=2).mean(dim=2)

    # ------------------------------------------------------------------ MPE rendering (reference supair.py:357-498)
    @torch.no_grad()
    def spn_max_activation(self, spn=None):
        """Input-independent most probable input of one SPN: follow the largest weight of every sum node from the root
        and take the means of the Gaussians reached; clipped to [0, 1].  -> (num_dims,).  (supair.py:357-380)"""
        spn = self.obj_spn if spn is None else spn
        img = np.clip(spn.reconstruct(spn.max_activation_idxs(), 0, sample=False), 0.0, 1.0)
        return torch.as_tensor(img, device=self.c.device).type(self.c.dtype)

    @torch.no_grad()
    def spn_mpe(self, z, x, spn=None):
        """MPE reconstruction of every object's glimpse.  z (nT, o, >=4) [sx, sy, x, y], x (nT, c, w, h) ->
        (nT, o, patch pixels): each glimpse is pushed through the SPN, every sum node keeps its strongest child, and the
        patch is rebuilt from the Gaussian means on that path (supair.py:382-424).  One kernel instead of the
        reference's per-glimpse Python walk."""
        spn = self.bg_spn if isinstance(spn, str) and spn == 'bg' else (self.obj_spn if spn is None else spn)
        if x.shape[0] != z.shape[0]:
            raise ValueError('x and z need to have same batch_dim.')
        patches = self.patches_from_z(x, z[..., :4].flatten(end_dim=1))
        recons = spn.mpe(patches.flatten(start_dim=1))
        return recons.view(x.shape[0], self.c.num_obj, -1).type(self.c.dtype)

    @torch.no_grad()
    def reconstruct_from_z(self, z, x=None, max_activation=True, single_image=True):
        """Render frames from object states: the background SPN's max-activation image plus one patch per object pasted
        at its (scale, position), clamped to [0, 1].  z (n, T, o, >=4) -> (n, T, c, w, h).

        max_activation=True: every object shows the object SPN's max-activation patch (independent of any image);
        False: the MPE reconstruction of its glimpse of x -- x (n, T, c, w, h), or with single_image x (n, c, w, h) whose
        patches (cut at z[:, 0]) are reused for all T frames.  (supair.py:426-498)"""
        c = self.c
        z = z[..., :4]
        n, T, o = z.shape[:3]
        if c.channels != 1:
            raise NotImplementedError('reconstruct_from_z: single-channel SPN inputs only (as the reference, supair.py:452-464)')
        bg = self.spn_max_activation(self.bg_spn)
        if max_activation:
            patches, per = self.spn_max_activation(self.obj_spn).view(1, -1), 0
        else:
            if x is None:
                raise ValueError('Need x for reconstructions.')
            z_in, x_in = (z[:, 0], x) if single_image else (z.flatten(end_dim=1), x.flatten(end_dim=1))
            patches, per = self.spn_mpe(z_in, x_in, spn=self.obj_spn), (T if single_image else 1)
        if (c.width, c.height) == (32, 32) and (c.patch_width, c.patch_height) == (10, 10) and not bool(getattr(c, 'align_corners', False)):
            frames = ops.render_frames(bg.float(), patches.float(), per, z.reshape(-1, 4).float(), o)
            return frames.view(n, T, c.channels, c.width, c.height).type(c.dtype)
        # any other frame size / sampling convention: the reference's paste through the inverse transform (supair.py:480-498)
        zf = z.reshape(n * T, o, 4)
        if per == 0:
            pat = patches.view(1, 1, -1).expand(n * T, o, -1)
        else:
            pat = patches.view(-1, o, patches.shape[-1]).repeat_interleave(per, dim=0)[:n * T] if per > 1 else patches.view(n * T, o, -1)
        rec = bg.view(1, c.channels, c.width, c.height).expand(n * T, -1, -1, -1).type(c.dtype)
        for k in range(o):
            pk = pat[:, k].reshape(n * T, c.channels, c.patch_width, c.patch_height).type(c.dtype)
            rec = rec + self._sample(pk, self.expand_z(self.invert_z(zf[:, k].type(c.dtype))), c.width, c.height)
        return torch.clamp(rec, 0, 1).view(n, T, c.channels, c.width, c.height)

    # ------------------------------------------------------------------ SuPAIR-only ELBO
    def forward(self, x):
        """ELBO of SuPAIR alone on (n, T, c, w, h) frames -> mean ELBO, prop_dict."""
        codes = self.encoder(x.flatten(end_dim=1))
        zp_mean, zp_std = self.constrain_zp(codes.flatten(end_dim=1))
        z_obj, log_q = self.get_z_sup_sample(zp_mean, zp_std)
        log_q = log_q.view(-1, self.c.num_obj).sum(-1)
        log_p, _ = self.likelihood(x, z_obj)
        elbo = log_p - log_q
        average_elbo = torch.mean(elbo)
        if ((self.step_counter % self.c.print_every == 0)
                or (self.step_counter % self.c.plot_every == 0)):
            self.prop_dict['z'] = z_obj.view(*x.shape[0:2], self.c.num_obj, 4).detach()
            if self.c.debug:
                self.prop_dict['log_q'] = log_q.mean().detach()
                self.prop_dict['z_std'] = zp_std.mean(0).detach()
        return average_elbo, self.prop_dict
