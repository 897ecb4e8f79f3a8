"""Relational (interaction-network) dynamics model whose core runs as a HIP kernel.

Module/API surface and parameter names of the reference's model/video_prediction/dynamics.py
(`Dynamics`, :8-265): state encoder, self / relation / attention / affector / output MLPs
(three "cores" are allocated as in the reference, only core 0 is ever used), optional action
embedding and reward head.  `forward` packs the parameters of the chosen core into the image
the kernel reads (W | W^T | vectors, see csrc/gnn.hip) and launches one fused GNN step; the
T-serial inference loop uses the same image through `Stove` (ops.dyn_loop).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class Dynamics(nn.Module):
    def __init__(self, config, enc_input_size=None):
        super().__init__()
        self.c = config
        self.step_counter = 0
        self.prop_dict = {}
        cl = self.c.cl
        if cl != 32:
            raise NotImplementedError('the GNN kernel is built for cl = 32')
        if enc_input_size is None:
            enc_input_size = cl // 2

        if self.c.action_conditioned:
            self.n_action_enc = 4
            self.action_embedding_layer = nn.Linear(self.c.action_space, self.c.num_obj * self.n_action_enc)
            enc_input_size += self.n_action_enc
            self.reward_head0 = nn.Sequential(nn.Linear(cl, cl), nn.ReLU(), nn.Linear(cl, cl))
            self.reward_head1 = nn.Sequential(
                nn.Linear(cl, cl // 2), nn.ReLU(), nn.Linear(cl // 2, cl // 4), nn.ReLU(), nn.Linear(cl // 4, 1))
        if self.c.debug_core_appearance:
            enc_input_size += self.c.debug_appearance_dim
        self.enc_input_size = enc_input_size
        self.state_enc = nn.Linear(enc_input_size, cl)

        def stack(dims):
            cores = nn.ModuleList()
            for _ in range(3):
                cores.append(nn.ModuleList([nn.Linear(i, o) for i, o in dims]))
            return cores
        self.self_cores = stack([(cl, cl), (cl, cl)])
        self.rel_cores = stack([(1 + 2 * cl, 2 * cl), (2 * cl, cl), (cl, cl)])
        self.att_net = stack([(1 + 2 * cl, 2 * cl), (2 * cl, cl), (cl, 1)])
        self.affector = stack([(cl, cl), (cl, cl), (cl, cl)])
        self.out = stack([(cl + cl, cl), (cl, cl)])

        # plain attributes (not buffers), as in the reference: absent from the state dict
        self.diag_mask = 1 - torch.eye(self.c.num_obj, dtype=self.c.dtype).unsqueeze(2).unsqueeze(0).to(self.c.device)
        if self.c.debug_xavier:
            raise NotImplementedError('debug_xavier touches layers that do not exist (reference dynamics.py:141-145)')
        # NB the reference's selection is inverted (dynamics.py:109): 'leaky_relu' selects ELU,
        # anything else (the default 'relu') selects leaky_relu(0.01).  Kept on purpose.
        self.use_elu = self.c.debug_nonlinear == 'leaky_relu'
        self.nonlinear = F.elu if self.use_elu else F.leaky_relu

        std = list(self.c.transition_lik_std)
        if len(std) == 4:
            std = std + 12 * [0.01]
        elif len(std) != cl // 2:
            raise ValueError('Specify valid transition_lik_std.')
        self.transition_lik_std = torch.tensor([[std]], dtype=torch.float32, device=self.c.device)
        self.transition_lik_std_host = [float(v) for v in std]     # the fused ELBO kernel takes them as launch constants

    # ------------------------------------------------------------------ kernel parameter image
    def param_image(self, core_idx=0, leaf=None, pad_value=0.0):
        """(W image, vector image, W^T image) of one core as flat float32 tensors.

        `leaf(p)` substitutes every parameter (ParamArena passes arena indices through the same
        packing code to get the gather table of its one-launch image; pads are then `pad_value`)."""
        k = core_idx
        cl = self.c.cl
        if leaf is not None:
            class _L:                                   # a Linear whose weight / bias went through leaf()
                def __init__(self, lin):
                    self.weight, self.bias = leaf(lin.weight), leaf(lin.bias)
            wrap = lambda lin: _L(lin)                  # noqa: E731
        else:
            wrap = lambda lin: lin                      # noqa: E731
        this = self
        self = type('_View', (), {})()                  # same packing code on wrapped layers
        self.state_enc = wrap(this.state_enc)
        for name in ('self_cores', 'rel_cores', 'att_net', 'affector', 'out'):
            setattr(self, name, {k: [wrap(lin) for lin in getattr(this, name)[k]]})
        self.enc_input_size = this.enc_input_size
        r0, a0 = self.rel_cores[k][0], self.att_net[k][0]
        enc_w = F.pad(self.state_enc.weight, (0, cl - self.enc_input_size), value=pad_value)
        ef = torch.cat([r0.weight[:, :cl], r0.weight[:, cl:2 * cl], a0.weight[:, :cl], a0.weight[:, cl:2 * cl]], 0)
        mats = [enc_w, self.self_cores[k][0].weight, self.self_cores[k][1].weight, ef,
                self.rel_cores[k][1].weight, self.att_net[k][1].weight, self.rel_cores[k][2].weight,
                self.affector[k][0].weight, self.affector[k][1].weight, self.affector[k][2].weight,
                self.out[k][0].weight, self.out[k][1].weight]
        w_img = torch.cat([m.reshape(-1) for m in mats])
        with torch.no_grad():
            wt_img = torch.cat([m.t().reshape(-1) for m in mats])
        a2 = self.att_net[k][2]
        vecs = [self.state_enc.bias, self.self_cores[k][0].bias, self.self_cores[k][1].bias,
                r0.bias, r0.weight[:, 2 * cl], a0.bias, a0.weight[:, 2 * cl],
                self.rel_cores[k][1].bias, self.att_net[k][1].bias, self.rel_cores[k][2].bias,
                a2.weight.reshape(-1), F.pad(a2.bias, (0, cl - 1), value=pad_value),
                self.affector[k][0].bias, self.affector[k][1].bias, self.affector[k][2].bias,
                self.out[k][0].bias, self.out[k][1].bias]
        v_img = torch.cat(vecs)
        return w_img, v_img, wt_img

    def kernel_params(self, core_idx=0):
        """-> (image tuple for ops.gnn_step / dyn_loop / rollout, gradient sink or None).  With a ParamArena the
        image is one gather launch and the gradient image is scattered straight into the flat gradient buffer."""
        arena = getattr(self, '_arena', None)
        if arena is not None and arena.has_gnn:
            return (arena.gnn_image(core_idx), None, None), (lambda g: arena.gnn_sink(g, core_idx))
        return self.param_image(core_idx), None

    def loop_consts(self):
        """(pos std bound, velocity std bound, latent std bound) of constrain_z_dyn."""
        return (self.c.pos_var, 0.04, self.c.debug_latent_q_std)

    # ------------------------------------------------------------------ API
    def constrain_z_dyn(self, z, z_std=None):
        """Means to (-1, 1); stds to (0, pos_var) / (0, 0.04) / (0, debug_latent_q_std)."""
        z_c = 2 * torch.sigmoid(z) - 1
        if z_std is None:
            return z_c, None
        sg = torch.sigmoid(z_std)
        z_std = torch.cat([self.c.pos_var * sg[..., :2], 0.04 * sg[..., 2:4],
                           self.c.debug_latent_q_std * sg[..., 4:]], -1)
        return z_c, z_std

    def core_inputs(self, s, actions=None, obj_appearances=None):
        """[state | action embedding | appearance] as the encoder expects it."""
        if actions is not None:
            emb = self.embed_actions(actions)
            s = torch.cat([s, emb.view(*emb.shape[:-1], self.c.num_obj, self.n_action_enc)], -1)
        if obj_appearances is not None:
            s = torch.cat([s, obj_appearances], -1)
        return s

    def embed_actions(self, actions):
        """action_embedding_layer(actions) (dynamics.py:238-244); on the GPU the narrow-layer kernel instead of a library GEMM."""
        lay = self.action_embedding_layer
        if actions.is_cuda and actions.dtype == torch.float32 and lay.weight.dtype == torch.float32:
            return ops.linear(actions, lay.weight, lay.bias)
        return lay(actions)

    def reward_from_pred(self, dynamic_pred):
        """(…, o, cl) -> (…, 1) predicted reward in (0, 1)."""
        if dynamic_pred.is_cuda and dynamic_pred.dtype == torch.float32 and self.c.cl == 32 and getattr(self.c, 'fused_reward_head', True):
            h0, h1 = self.reward_head0, self.reward_head1
            return ops.reward_head(dynamic_pred, [h0[0].weight, h0[0].bias, h0[2].weight, h0[2].bias,
                                                  h1[0].weight, h1[0].bias, h1[2].weight, h1[2].bias, h1[4].weight, h1[4].bias])
        q = self.reward_head0(dynamic_pred).sum(-2)
        return torch.sigmoid(self.reward_head1(q))

    def forward(self, s, core_idx, actions=None, obj_appearances=None, lim_enc=2):
        """One prediction step: s (n, o, cl//2) -> (n, o, cl) means|stds of the next state, reward."""
        s = self.core_inputs(s, actions, obj_appearances)
        if s.shape[-1] != self.enc_input_size:
            raise ValueError('core input has %d dims, the encoder expects %d' % (s.shape[-1], self.enc_input_size))
        image, sink = self.kernel_params(core_idx)
        result, dynamic_pred = ops.gnn_step(s, image, lim_enc, self.use_elu, sink)
        if self.c.action_conditioned:
            return result, self.reward_from_pred(dynamic_pred).view(-1, 1)
        return result, 0
