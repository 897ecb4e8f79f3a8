"""STOVE: structured object-aware video prediction (SuPAIR + relational dynamics).

Module/API surface of the reference's model/video_prediction/stove.py (`Stove`, :11-897):
`forward(x, step_counter, actions=None, pretrain=False) -> (elbo, prop_dict, rewards)`,
`rollout(z_last, num, sample, return_std, actions, appearance)`, the helper methods and the
`prop_dict` side channel.  What differs is where the work happens on an MI355X:

  * recognition (encoder, constrain, smoothing, velocities, ELBO assembly) is batched
    PyTorch-ROCm host code -- a few dozen launches per step, none of them T-serial;
  * the two T-serial parts are single kernels: object matching (csrc/match.hip) and the whole
    inference recursion dyn -> constrain -> fuse-with-SuPAIR -> sample for t = skip..T-1
    (csrc/gnn.hip, `ops.dyn_loop`), instead of ~60 launches per frame;
  * both image likelihoods are the fused scene/SPN pipeline (`Supair.likelihood`).

Noise: all reparameterisation draws go through `self.noise_fn(kind, shape)` when set (parity
tests inject the reference's draws: 'latent' (n,o,12), 'std' (n,o,12), 'steps' (n,T-skip,o,18));
otherwise they come from the device generator.
"""
import math

import torch
import torch.nn as nn
from torch.distributions import Normal

from .. import ops
from .. import settings as _settings
from ..utils.utils import bw_transform
from .dynamics import Dynamics
from .supair import Supair

_LOG_2PI = math.log(2.0 * math.pi)


def _normal_log_prob(x, mean, std):
    return -0.5 * ((x - mean) / std) ** 2 - torch.log(std) - 0.5 * _LOG_2PI


class Stove(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.c = config
        self.step_counter = 0
        self.prop_dict = {}
        self.noise_fn = None
        self.sup = Supair(config)
        self.dyn = Dynamics(config)
        self.reconstruct_from_z = self.sup.reconstruct_from_z

        # priors of the unstructured latent and its std at t = skip-1 (plain attributes)
        self.latent_prior = Normal(torch.tensor([0.0], device=self.c.device), torch.tensor([0.01], device=self.c.device))
        self.z_std_prior = Normal(torch.tensor([0.1], device=self.c.device), torch.tensor([0.01], device=self.c.device))

        mode = self.c.debug_match_objects
        if mode == '3_only':
            if self.c.num_obj != 3:
                raise ValueError('Matching Function not compatible w/ specified number of objects.')
            self.match_objects = self._3_only_match_objects
        elif mode == 'volatile':
            self.match_objects = self._volatile_match_objects
        elif mode == 'greedy':
            self.match_objects = self._greedy_match_objects
        else:
            raise ValueError('Specify valid self.c.debug_match_ojects.')

    # ------------------------------------------------------------------ noise
    def _noise(self, kind, shape, like):
        if self.noise_fn is not None:
            return self.noise_fn(kind, shape).to(device=like.device, dtype=like.dtype)
        return torch.randn(shape, device=like.device, dtype=like.dtype)

    def _draw_ahead(self, numel, dev):
        """All of a step's standard-normal draws from the library's generator, on the parameter stream (ops 'pre'), behind nothing:
        -> (tensor, event to wait for before reading it)."""
        src = getattr(self, '_noise_source', None)
        if src is None or src.device != dev:
            src = self._noise_source = ops.NoiseSource(dev)
        main, side = torch.cuda.current_stream(dev), ops._side_stream(dev, 'pre')
        if not _settings.OVERLAP:
            return src.normal(numel), None
        side.wait_stream(main)                   # allocator order (the block may have been in use on `main`); nothing of this step is on `main` yet
        with torch.cuda.stream(side):
            out = src.normal(numel)
            ev = torch.cuda.Event()
            ev.record(side)
        out.record_stream(main)
        return out, ev

    # ------------------------------------------------------------------ SuPAIR state helpers
    def v_from_state(self, z_sup):
        """(n,T,o,4) [sx,sy/sx,x,y] -> (n,T,o,6) with velocities x_t - x_{t-1}; the t=0 row is zero."""
        full = torch.cat([z_sup[:, 1:], z_sup[:, 1:, :, 2:] - z_sup[:, :-1, :, 2:]], -1)
        return torch.cat([torch.zeros_like(full[:, :1]), full], 1)

    def v_std_from_pos(self, z_sup_std):
        """Std of the finite-difference velocity: sqrt(s_t^2 + s_{t-1}^2); the t=0 row is zero."""
        v_std = torch.sqrt(z_sup_std[:, 1:, :, 2:] ** 2 + z_sup_std[:, :-1, :, 2:] ** 2)
        full = torch.cat([z_sup_std[:, 1:], v_std], -1)
        return torch.cat([torch.zeros_like(full[:, :1]), full], 1)

    def full_state(self, z_dyn, std_dyn, z_sup, std_sup):
        """q(z_t): scales from SuPAIR, (x, v) as the product of the dynamics and SuPAIR Gaussians,
        latents from the dynamics.  Returns a reparameterised sample, its log-density, mean, std."""
        s_d, s_s = std_dyn[..., :4], std_sup[..., 2:6]
        var = s_d ** 2 + s_s ** 2
        mean_xv = (s_s ** 2 * z_dyn[..., :4] + s_d ** 2 * z_sup[..., 2:6]) / var
        std_xv = s_d * s_s / torch.sqrt(var)
        c = self.c
        if c.debug_no_latents:                  # reference stove.py:140-148: sample the six SuPAIR dimensions only, latents are zeros
            mean = torch.cat([z_sup[..., :2], mean_xv], -1)
            std = torch.cat([std_sup[..., :2], std_xv], -1)
            z_s = mean + std * self._noise('step', mean.shape, mean)
            log_q = _normal_log_prob(z_s, mean, std)
            return torch.cat([z_s, torch.zeros_like(z_dyn[..., 4:])], -1), log_q, mean, std
        if c.debug_no_velocity:                 # :154-160: velocities ~ N(0, 1) instead of the fused estimate
            mean = torch.cat([z_sup[..., :2], mean_xv[..., :2], torch.zeros_like(mean_xv[..., 2:]), z_dyn[..., 4:]], -1)
            std = torch.cat([std_sup[..., :2], std_xv[..., :2], torch.ones_like(std_xv[..., 2:]), std_dyn[..., 4:]], -1)
        else:                                   # (debug_no_reuse, :151-153, is overwritten by the if / else behind it in the reference:
            mean = torch.cat([z_sup[..., :2], mean_xv, z_dyn[..., 4:]], -1)       # the flag changes nothing there, and nothing here)
            std = torch.cat([std_sup[..., :2], std_xv, std_dyn[..., 4:]], -1)
        z_s = mean + std * self._noise('step', mean.shape, mean)
        return z_s, _normal_log_prob(z_s, mean, std), mean, std

    def transition_lik(self, means, results):
        """log p(z_t | z_{t-1}) of the inferred states under the generative dynamics (fixed std)."""
        return _normal_log_prob(results, means, self.dyn.transition_lik_std.to(results.device, results.dtype))

    # ------------------------------------------------------------------ object matching
    def _match(self, mode, z_sup, z_sup_std, obj_appearances):
        feats = [z_sup[..., 2:4]]
        if obj_appearances is not None and self.c.debug_match_appearance:
            feats.append(2 * obj_appearances - 1)          # the kernel rescales (v+1)/2; colours are already in [0,1]
        idx, perm = ops.match_objects(torch.cat(feats, -1), mode)

        def permute(t):
            if t is None:
                return None
            if perm is not None:
                return torch.matmul(perm.to(t.dtype), t)
            return torch.gather(t, 2, idx.unsqueeze(-1).expand(-1, -1, -1, t.shape[-1]))
        z_m, std_m, app_m = permute(z_sup), permute(z_sup_std), permute(obj_appearances)
        if z_sup_std is None and obj_appearances is not None:
            return z_m, app_m
        if z_sup_std is not None:
            return z_m, std_m, app_m
        return z_m

    def _3_only_match_objects(self, z_sup, z_sup_std=None, obj_appearances=None):
        """Nearest-neighbour slot assignment over time with greedy repair (3 objects only)."""
        return self._match('3_only', z_sup, z_sup_std, obj_appearances)

    def _greedy_match_objects(self, z_sup, z_sup_std=None, obj_appearances=None):
        """Greedy bipartite matching over time (any number of objects)."""
        return self._match('greedy', z_sup, z_sup_std, obj_appearances)

    def _volatile_match_objects(self, z_sup, z_sup_std=None, obj_appearances=None):
        """Per-object nearest previous slot; does not guarantee a permutation."""
        return self._match('volatile', z_sup, z_sup_std, obj_appearances)

    def fix_supair(self, z, z_std=None):
        """Replace glitches -- states whose first two dims jump by > 0.095 w.r.t. BOTH neighbours
        in time -- by the mean of the neighbours (the 2-dim mask is tiled over all dims)."""
        has_std = z_std is not None
        zz = torch.cat([z, z_std], -1) if has_std else z
        jump = (zz[:, 1:, :, :2] - zz[:, :-1, :, :2]).abs().detach()
        pad = torch.zeros_like(jump[:, :1])
        hit = (torch.cat([pad, jump], 1) > 0.095) & (torch.cat([jump, pad], 1) > 0.095)
        hit = hit.repeat(1, 1, 1, zz.shape[-1] // 2)
        smooth = torch.zeros_like(zz)
        smooth[:, 1:-1] = (zz[:, :-2] + zz[:, 2:]) / 2
        out = torch.where(hit, smooth, zz)
        if has_std:
            return torch.chunk(out, 2, dim=-1)
        return out

    def object_embedding(self, z, x_color):
        """Mean colour of each object's glimpse of the colour frame: (n,T,o,3)."""
        z_patch = self.sup.sy_from_quotient(z[..., :4].detach())
        c = self.c
        if x_color.is_cuda and x_color.dtype == torch.float32 and x_color.shape[-1] == 32 and x_color.shape[-2] == 32 \
                and c.patch_width == 10 and c.patch_height == 10 and not bool(getattr(c, 'align_corners', False)):
            emb = ops.glimpse_mean(x_color.flatten(end_dim=1), z_patch.flatten(end_dim=2), z.shape[-2])     # one launch
            return emb.view(*z.shape[:-1], x_color.shape[2])
        patches = self.sup.patches_from_z(x_color.flatten(end_dim=1), z_patch.flatten(end_dim=2))
        return patches.mean((-1, -2)).view(*z.shape[:-1], 3)

    # ------------------------------------------------------------------ forward
    def stove_forward(self, x, actions=None, x_color=None):
        c = self.c
        n, T = x.shape[:2]
        o, skip, cl = c.num_obj, c.skip, c.cl

        # [amd] the full_state ablations (reference stove.py:140-160, config.py:79,132,134) change what q(z) is made of: they run on
        # the op-by-op chain (host time loop over the GNN step kernel, full_state and the ELBO terms in PyTorch), not in the fused
        # recursion / ELBO kernels, which implement the default q(z)
        ablated = bool(c.debug_no_latents or c.debug_no_velocity)
        fused_dyn = bool(getattr(c, 'fused_dynamics', True)) and not ablated
        fused_elbo = bool(getattr(c, 'fused_elbo', True)) and not ablated
        arena = getattr(self.dyn, '_arena', None)
        if arena is not None and torch.is_grad_enabled() and fused_dyn:
            arena.prefetch_images()          # parameter-only launches, off the critical path (second stream)

        # 1. SuPAIR states for every frame, consistent object order, smoothing, velocities.
        # Without appearance features the whole chain (constrain_zp, matching, gather, fix_supair, velocities) is the
        # fused state pipeline (csrc/state.hip); the PyTorch chain below it is the same computation op by op.
        # device RNG: the three draws of the reference (latent prior, the unused std prior, the step noise) as ONE launch, made
        # before the state pipeline so that it can write the recursion's initial state [SuPAIR | 0.01 latent noise] itself.
        # [amd] With the library's counter-based generator (config.device_noise = 'philox') the draw depends on nothing of this
        # step: it is enqueued on the parameter stream BEFORE the recognition network and joined where the state pipeline reads it.
        Ts = T - skip
        pooled = init_full = pooled_ev = None
        nl = n * o * (cl // 2 - 4)
        n_pool = 2 * nl + n * Ts * o * (cl // 2 + 2)
        draw = self.noise_fn is None and fused_dyn
        if draw and x.is_cuda and getattr(c, 'device_noise', 'philox') == 'philox':
            pooled, pooled_ev = self._draw_ahead(n_pool, x.device)
        codes = self.sup.encoder(x.flatten(end_dim=1))
        fused_state = bool(getattr(c, 'fused_state', True)) and not c.debug_match_appearance
        if pooled_ev is not None:
            torch.cuda.current_stream(x.device).wait_event(pooled_ev)
        elif draw and pooled is None:                              # (no overlap: _draw_ahead already drew on this stream, no event to wait for)
            pooled = self._noise('pooled', (n_pool,), codes)      # [latent | std | steps]
        if fused_state:
            zfix, zsup_loop, zsstd_loop, init6, idx = ops.supair_state(
                codes.flatten(end_dim=1), self.sup.zp_span_low(), n, T, o, skip, c.debug_fix_supair, c.debug_match_objects,
                lat_noise=pooled[:nl].view(n, o, cl // 2 - 4) if pooled is not None else None)
            if pooled is not None:
                init_full, init6 = init6, init6[..., :6]
            z_sup = zfix[..., :4]
            obj_appearances = None
            if c.debug_core_appearance:
                # appearance embedding of the UNMATCHED slots (as the reference computes it), permuted like the states
                z_pre, _ = self.sup.constrain_zp(codes.flatten(end_dim=1).detach())
                app = self.object_embedding(z_pre.view(n, T, o, 4), x_color)
                obj_appearances = torch.gather(app, 2, idx.unsqueeze(-1).expand(-1, -1, -1, app.shape[-1]))
        else:
            z_sup, z_sup_std = self.sup.constrain_zp(codes.flatten(end_dim=1))
            z_sup, z_sup_std = z_sup.view(n, T, o, 4), z_sup_std.view(n, T, o, 4)
            app = None
            if c.debug_core_appearance or c.debug_match_appearance:
                app = self.object_embedding(z_sup, x_color)
            z_sup, z_sup_std, obj_appearances = self.match_objects(z_sup, z_sup_std, app)
            if c.debug_fix_supair:
                z_sup, z_sup_std = self.fix_supair(z_sup, z_sup_std)
            z_sup_full = self.v_from_state(z_sup)
            z_sup_std_full = self.v_std_from_pos(z_sup_std)
            zsup_loop, zsstd_loop, init6 = z_sup_full[:, skip:], z_sup_std_full[:, skip:], z_sup_full[:, skip - 1]

        # 2. initial state at t = skip-1 and the inference recursion
        if init_full is not None:
            init_z = init_full
        else:
            if pooled is not None:
                lat0 = 0.01 * pooled[:nl].view(n, o, cl // 2 - 4)
            else:
                lat0 = 0.01 * self._noise('latent', (n, o, cl // 2 - 4), z_sup)
                _ = 0.1 + 0.01 * self._noise('std', (n, o, cl // 2 - 4), z_sup)     # drawn as in the reference, unused
            init_z = torch.cat([init6, lat0], -1)
        use_app = bool(c.debug_core_appearance)
        if fused_dyn:
            extra = []
            if actions is not None:
                emb = self.dyn.embed_actions(actions[:, skip - 1:T - 1])
                extra.append(emb.view(n, Ts, o, self.dyn.n_action_enc))
            if use_app:
                extra.append(obj_appearances[:, skip - 1:T - 1])
            extra = torch.cat(extra, -1) if extra else None
            if pooled is not None:
                eps = pooled[2 * n * o * (cl // 2 - 4):].view(n, Ts, o, cl // 2 + 2)
            else:
                eps = self._noise('steps', (n, Ts, o, cl // 2 + 2), z_sup)
            image, sink = self.dyn.kernel_params(0)
            z_s, z_dyn_s, z_dyn_std_s, mean_s, z_std_s, pred = ops.dyn_loop(
                init_z, zsup_loop, zsstd_loop, eps, extra, image,
                2, self.dyn.use_elu, self.dyn.loop_consts(), want_pred=bool(c.action_conditioned), sink=sink)
            rewards = self.dyn.reward_from_pred(pred) if c.action_conditioned else torch.zeros(Ts)
        else:
            z_prev, zs, zd, zds, zm, zst, rew = init_z, [], [], [], [], [], []
            eps_all = self._noise('steps', (n, Ts, o, 6 if c.debug_no_latents else cl // 2 + 2), z_sup)
            saved_fn = self.noise_fn
            for t in range(skip, T):
                tmp, reward = self.dyn(z_prev[..., 2:], 0, actions[:, t - 1] if actions is not None else None,
                                       obj_appearances[:, t - 1] if use_app else None)
                m, sd = self.dyn.constrain_z_dyn(tmp[..., :cl // 2], tmp[..., cl // 2:])
                z_dyn_t = torch.cat([z_prev[..., 2:4] + m[..., :2], m[..., 2:]], -1)
                self.noise_fn = lambda kind, shape, _e=eps_all[:, t - skip]: _e
                z_t, _, mean_t, std_t = self.full_state(z_dyn_t, sd, zsup_loop[:, t - skip], zsstd_loop[:, t - skip])
                self.noise_fn = saved_fn
                zs.append(z_t); zd.append(z_dyn_t); zds.append(sd); zm.append(mean_t); zst.append(std_t); rew.append(reward)
                z_prev = z_t
            z_s, z_dyn_s, z_dyn_std_s = torch.stack(zs, 1), torch.stack(zd, 1), torch.stack(zds, 1)
            mean_s, z_std_s = torch.stack(zm, 1), torch.stack(zst, 1)
            rewards = torch.stack(rew, 1) if c.action_conditioned else torch.zeros(Ts)

        # 3. ELBO: image likelihood (SPNs), q(z|x) and the generative transition likelihood.
        # The reference scores frames skip..T-1 (sampled z) and frame 1..skip-1 (SuPAIR mean) in two
        # likelihood calls (stove.py:731-736); here both go through ONE fused scene launch.
        if fused_state:
            z_all, z_s = ops.zall(zfix, z_s, n, T, o, skip)        # (z_s handed through: its ELBO gradient is added by zall's backward kernel)
        else:
            z_all = torch.cat([z_sup[:, 1:skip], z_s[..., :4]], 1)             # (n, T-1, o, 4) [sx, sy/sx, x, y]
            z_all = self.sup.sy_from_quotient(z_all.flatten(end_dim=2))
        lik_all, sup_prop = self.sup.likelihood(x[:, 1:], z_all, log_from=skip - 1)
        self.prop_dict.update(sup_prop)
        lik_all = lik_all.view(n, T - 1)
        if fused_elbo:
            # log q(z), the transition likelihood and all the means in two launches (csrc/state.hip)
            average_elbo, stats = ops.elbo(z_s, mean_s, z_std_s, z_dyn_s, lik_all,
                                           self.dyn.transition_lik_std_host, n, T, o, skip)
            trans_mean, logq_mean = stats[0], stats[1]
        else:
            log_z_s = _normal_log_prob(z_s[..., :mean_s.shape[-1]], mean_s, z_std_s)          # (six dims under debug_no_latents)
            img_lik = lik_all[:, skip - 1:].reshape(-1)
            img_lik_sup = lik_all[:, :skip - 1].reshape(-1)
            log_z_f = log_z_s.sum((-2, -1)).flatten()
            trans_lik = self.transition_lik(means=z_dyn_s, results=z_s[..., 2:]).sum((-2, -1)).flatten(end_dim=1)
            elbo = trans_lik + img_lik - log_z_f
            average_elbo = torch.mean(elbo) + torch.mean(img_lik_sup)
            trans_mean, logq_mean = trans_lik.mean(), log_z_f.mean()

        if (self.step_counter % c.print_every == 0) or (self.step_counter % c.plot_every == 0):
            pd = self.prop_dict
            pd['z'] = self.sup.sy_from_quotient(z_s).detach()
            pd['z_dyn'] = z_dyn_s.detach()
            pd['z_sup'] = self.sup.sy_from_quotient(zsup_loop).detach()
            pd['z_std'] = z_std_s.mean((0, 1, 2)).detach()
            nan2 = torch.full((2,), float('nan'), device=z_s.device, dtype=z_s.dtype)
            pd['z_dyn_std'] = torch.cat([nan2, z_dyn_std_s[..., :4].mean((0, 1, 2)).detach()])
            pd['z_sup_std'] = zsstd_loop.mean((0, 1, 2)).detach()
            pd['log_q'] = logq_mean.detach()
            pd['translik'] = trans_mean.detach()
            pd['obj_appearances'] = obj_appearances[:, skip:].detach() if obj_appearances is not None else None
        return average_elbo, self.prop_dict, rewards

    # ------------------------------------------------------------------ generative rollout
    def rollout(self, z_last, num=None, sample=False, return_std=False, actions=None, appearance=None):
        """Roll the dynamics forward from z_last (n, o, cl//2+2) with [sx, sy, ...]; scales stay fixed.
        -> z_pred (n, num, o, cl//2+2), rewards   (plus stds / log-probs for return_std / sample)."""
        c = self.c
        cl = c.cl
        if num is None:
            num = c.num_rollout
        n, o = z_last.shape[:2]
        if not sample:
            extra = []
            if actions is not None:
                emb = self.dyn.embed_actions(actions)
                extra.append(emb.view(n, actions.shape[1], o, self.dyn.n_action_enc))
            if appearance is not None:
                a_len = actions.shape[1] if actions is not None else 1
                extra.append(appearance.unsqueeze(1).expand(-1, a_len, -1, -1))
            extra = torch.cat(extra, -1).contiguous() if extra else None
            z_full, z_stds, pred = ops.rollout(z_last, extra, self.dyn.kernel_params(0)[0], num, 2, self.dyn.use_elu,
                                              self.dyn.loop_consts(), want_std=return_std,
                                              want_pred=bool(c.action_conditioned))
            rewards = self.dyn.reward_from_pred(pred) if c.action_conditioned else torch.zeros(num)
            if return_std:
                return z_full, z_stds.detach(), rewards
            return z_full, rewards

        # sampling rollout: the sampled state feeds back, so this stays a host loop of single steps
        scale = z_last[..., :2]
        z, log_qs, rew = [z_last], [], []
        a_len = actions.shape[1] if actions is not None else 1
        for t in range(1, num + 1):
            act = actions[:, (t - 1) % a_len] if actions is not None else None
            tmp, reward = self.dyn(z[-1][..., 2:], 0, act, appearance)
            m, sd = self.dyn.constrain_z_dyn(tmp[..., :cl // 2], tmp[..., cl // 2:])
            nxt = torch.cat([z[-1][..., 2:4] + m[..., :2], m[..., 2:]], -1)
            smp = nxt + sd * self._noise('rollout', nxt.shape, nxt)
            log_qs.append(_normal_log_prob(smp, nxt, sd))
            z.append(torch.cat([scale, smp], -1))
            rew.append(reward)
        rewards = torch.stack(rew, 1) if c.action_conditioned else torch.zeros(num)
        return torch.stack(z[1:], 1), torch.stack(log_qs, 1), rewards

    def forward(self, x, step_counter, actions=None, pretrain=False):
        """x (n, T, 3, w, h) colour frames in [0,1] -> (elbo, prop_dict, rewards)."""
        self.step_counter = step_counter
        self.sup.step_counter = step_counter
        self.dyn.step_counter = step_counter
        x_color = x
        if self.c.debug_bw:
            # [amd] a loader that keeps the training set as the bw plane itself (load_data.DeviceClipLoader, frame_store='bw32':
            # bw_transform applied once, at upload) hands over (n, T, 1, w, h) frames and says so in config.input_bw_plane
            if not (x.shape[2] == 1 and x.dtype != torch.uint8 and getattr(self.c, 'input_bw_plane', False)):
                x = bw_transform(x)
        elif x.dtype == torch.uint8:
            x = x.to(self.c.dtype) / 255.0
        if x_color.dtype == torch.uint8 and (self.c.debug_core_appearance or self.c.debug_match_appearance):
            x_color = x_color.to(x.dtype) / 255.0
        if pretrain:
            elbo, prop_dict = self.sup(x)
            return elbo, prop_dict, 0
        try:
            if self.c.debug_core_appearance or self.c.debug_match_appearance:
                return self.stove_forward(x, actions=actions, x_color=x_color)
            return self.stove_forward(x, actions=actions)
        finally:
            arena = getattr(self.dyn, '_arena', None)
            if arena is not None:
                arena.drop_prefetched()      # whatever was baked ahead and not consumed must not outlive this call
