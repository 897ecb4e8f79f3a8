"""Random sum-product networks (RAT-SPN) whose sweep runs as HIP kernels on gfx950.

Module/API surface of the reference's model/spn/rat_torch.py (SpnArgs :21-35, GaussVector :64,
ProductVector :138, SumVector :185, RatSpn :256): the same constructor arguments, the same
`vector_list.L.i.{means,sigma_params,params}` / `output_vector.params` state-dict names and the
same `RatSpn.forward(inputs, marginalized=None) -> (B, num_classes)` contract.  The node
vectors only hold parameters and structure; evaluation is one fused kernel sweep
(stove_amd/csrc/spn_obj.hip, spn_bg.hip) over tables baked from the parameters:

  leaf      log N(x; mu, v) = a x^2 + b x + c,   v = vmin + (vmax - vmin) sigmoid(sigma_params)
            (the reference's "sigma" is a variance, rat_torch.py:85-99)
  sum       softmax(params, 0) in the linear domain; the kernel evaluates
            log sum_k exp(child_k) w_k as a bilinear form of max-shifted exponentials.

Two shapes have kernels, the two STOVE builds (probabilistic_models.py): the 100-dim object SPN
(6 x random_split(2,2), 10 gaussians, 10 sums) and the 1024-dim background SPN
(3 x random_split(2,1), 6 gaussians).  Other shapes raise NotImplementedError.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from . import region_graph


def truncated_normal_(tensor, mean=0, std=0.1):
    """Fill with N(mean, std) truncated to two standard deviations: of four candidate draws per
    element keep the first that lies within (-2, 2)."""
    with torch.no_grad():
        cand = tensor.new_empty(tuple(tensor.shape) + (4,)).normal_()
        ok = (cand > -2) & (cand < 2)
        first = ok.max(-1, keepdim=True)[1]
        tensor.copy_(cand.gather(-1, first).squeeze(-1))
        tensor.mul_(std).add_(mean)
    return tensor


class BasicParamProvider:
    def grab_sum_parameters(self, num_inputs, num_sums):
        return nn.Parameter(torch.empty(num_inputs, num_sums))

    def grab_leaf_parameters(self, scope, number, name=None):
        return nn.Parameter(torch.empty(len(scope), number))


class SpnArgs:
    def __init__(self):
        self.linear_sum_weights = False
        self.normalized_sums = True
        self.num_sums = 20
        self.param_provider = BasicParamProvider()
        self.gauss_min_sigma = 0.1
        self.gauss_max_sigma = 1.0
        self.gauss_mean_of_means = 0.0
        self.dist = 'Gauss'
        self.init_fn = truncated_normal_
        self.gauss_min_mean = None
        self.gauss_max_mean = None


class NodeVector(nn.Module):
    """A vector of SPN nodes over one region (parameter + structure holder)."""

    def __init__(self, name):
        super().__init__()
        self.name = name

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, other):
        return isinstance(other, NodeVector) and self.name == other.name

    def init_params(self, init_fn=None):
        pass

    def num_params(self):
        return 0

    def reconstruct(self, idxs, node_num, sample):
        """The input (num_dims,) this node explains best when every sum node below it follows `idxs[vector][node]`
        (reference rat_torch.py:126-135 leaf, :177-183 product, :224-229 sum): leaves write the mean (or a draw) of
        the reached component on their scope, everything outside this node's scope stays 0."""
        out = np.zeros(_num_dims_of(self))
        todo = [(self, int(node_num))]
        while todo:
            vec, n = todo.pop()
            if isinstance(vec, GaussVector):
                mu = vec.means[:, n].detach().cpu().double().numpy()
                if sample:
                    mu = np.random.normal(mu, np.sqrt(vec.variance()[:, n].detach().cpu().double().numpy()))
                out[vec.scope] = mu
            elif isinstance(vec, ProductVector):
                first = vec.inputs[0].size                  # node = row * first + col: input 0 <- col, input 1 <- row
                todo += [(vec.inputs[1], n // first), (vec.inputs[0], n % first)]
            else:
                k = int(idxs[vec][n])
                for child in vec.inputs:                    # k indexes the concatenation of the product vectors
                    if k < child.size:
                        todo.append((child, k))
                        break
                    k -= child.size
        return out


def _num_dims_of(vec):
    while not isinstance(vec, GaussVector):
        vec = vec.inputs[0]
    return vec.num_dims


class GaussVector(NodeVector):
    def __init__(self, region, args, name, num_dims=0):
        super().__init__(name)
        self.args = args
        self.scope = sorted(int(i) for i in region)
        self.local_size = len(self.scope)
        self.size = args.num_gauss
        self.num_dims = num_dims
        self.means = args.param_provider.grab_leaf_parameters(self.scope, args.num_gauss)
        self.sigma_params = args.param_provider.grab_leaf_parameters(self.scope, args.num_gauss)

    def init_params(self, init_fn=None):
        init_fn = init_fn or truncated_normal_
        init_fn(self.means, mean=self.args.gauss_mean_of_means, std=0.1)
        init_fn(self.sigma_params, mean=0.0, std=0.1)

    def num_params(self):
        return self.means.numel() + self.sigma_params.numel()

    def variance(self):
        a = self.args
        return a.gauss_min_sigma + (a.gauss_max_sigma - a.gauss_min_sigma) * torch.sigmoid(self.sigma_params)


class ProductVector(NodeVector):
    def __init__(self, vector1, vector2, name):
        super().__init__(name)
        self.inputs = [vector1, vector2]          # plain list: children are not sub-modules
        assert not set(vector1.scope) & set(vector2.scope)
        self.scope = sorted(set(vector1.scope) | set(vector2.scope))
        self.size = vector1.size * vector2.size


class SumVector(NodeVector):
    def __init__(self, prod_vectors, num_sums, args, name=''):
        super().__init__(name)
        self.inputs = prod_vectors
        self.size = num_sums
        self.args = args
        self.scope = self.inputs[0].scope
        for v in self.inputs:
            assert set(v.scope) == set(self.scope)
        self.params = args.param_provider.grab_sum_parameters(sum(v.size for v in prod_vectors), num_sums)

    def init_params(self, init_fn=None):
        (init_fn or truncated_normal_)(self.params)

    def num_params(self):
        return self.params.numel()


class RatSpn(nn.Module):
    def __init__(self, num_classes, region_graph, args=None, name=None):
        super().__init__()
        self.args = args if args is not None else SpnArgs()
        self.name = name if name is not None else str(id(self))
        self._region_graph = region_graph
        self.num_classes = num_classes
        self.num_dims = region_graph.get_num_items()
        self.vector_list = nn.ModuleList()
        self.output_vector = None
        self._build()
        self.init_params(self.args.init_fn)
        self._make_plan()

    # ------------------------------------------------------------------ structure
    def _build(self):
        layers = self._region_graph.make_layers()
        self.rg_layers = layers
        dist_of = {}
        leaf_layer = nn.ModuleList()
        for i, region in enumerate(layers[0]):
            if self.args.dist != 'Gauss':
                raise NotImplementedError('only Gaussian leaves')
            vec = GaussVector(region, self.args, '{}_gauss_{}'.format(self.name, i), num_dims=self.num_dims)
            leaf_layer.append(vec)
            dist_of[region] = vec
        self.vector_list.append(leaf_layer)
        products_of = {}
        for li in range(1, len(layers)):
            cur = nn.ModuleList()
            if li % 2 == 1:
                for i, partition in enumerate(layers[li]):
                    a, b = partition[0], partition[1]
                    vec = ProductVector(dist_of[a], dist_of[b], '{}_prod_{}_{}'.format(self.name, li, i))
                    cur.append(vec)
                    products_of.setdefault(tuple(sorted(a + b)), []).append(vec)
            else:
                n_out = self.num_classes if li == len(layers) - 1 else self.args.num_sums
                for i, region in enumerate(layers[li]):
                    vec = SumVector(products_of[region], n_out, self.args, name='{}_sum_{}_{}'.format(self.name, li, i))
                    cur.append(vec)
                    dist_of[region] = vec
            self.vector_list.append(cur)
        self.output_vector = dist_of[self._region_graph.get_root_region()]

    def init_params(self, init_fn):
        for layer in self.vector_list:
            for vec in layer:
                vec.init_params(init_fn)

    def get_sum_params(self):
        return {v: v.params for layer in self.vector_list for v in layer if isinstance(v, SumVector)}

    def num_params(self):
        return sum(v.num_params() for layer in self.vector_list for v in layer)

    def reconstruct(self, idxs, node_num, sample):
        """Input reached from root node `node_num` by following `idxs` (sum vector -> chosen child per node), leaves giving
        their component mean (or a draw if `sample`); numpy (num_dims,).  Reference rat_torch.py:359-372."""
        return self.output_vector.reconstruct(idxs, node_num, sample)

    def max_activation_idxs(self):
        """{sum vector: argmax child per node by weight} (the `max_idxs` of Supair.spn_max_activation, supair.py:371-372);
        one device->host copy for all sum vectors."""
        sums = list(self.get_sum_params().items())
        flat = torch.cat([p.detach().argmax(0).reshape(-1) for _, p in sums]).cpu().numpy()
        out, pos = {}, 0
        for v, p in sums:
            out[v] = flat[pos:pos + p.shape[1]]
            pos += p.shape[1]
        return out

    def mpe(self, inputs, return_pick=False):
        """Per row of inputs (B, D): the reconstruction along argmax_k(child_k + log w_k) of every sum node, clamped to
        [0, 1] (what Supair.spn_mpe builds from compute_activations(get_sum_child_acts=True) + reconstruct,
        supair.py:407-421).  Object-SPN shape only (one kernel, ops.objspn_mpe)."""
        if self._kind == 'obj_any':
            return self._mpe_any(inputs, return_pick)
        if self._kind != 'obj':
            raise NotImplementedError('RatSpn.mpe: gfx950 kernel exists for the object SPN shape only')
        coef, wsum, wroot, scope, leaf_slot = self.tables()
        mu = torch.stack([v.means for v in self.vector_list[0]])[self._plan(coef.device)['leaf_order']]
        return ops.objspn_mpe(inputs, mu.detach(), coef.detach(), wsum.detach(), wroot.detach(), scope, leaf_slot, return_pick)

    @torch.no_grad()
    def _mpe_any(self, inputs, return_pick=False):
        """mpe() for the general-size object SPN: the upward pass on the HIP operator (its saved leaf / sum-node values), the
        downward argmax walk as batched gathers on the device (plot-time only, not part of the training step)."""
        coef, wsum, wroot, lscope, slot, (R, G, S, D, lmax) = self.tables()
        lib = ops._lib.load()
        x = inputs.float().contiguous()
        n, dev = x.shape[0], x.device
        with torch.cuda.device(dev):
            saved = torch.empty(lib.stove_objspn_saved_floats_any(n, R, G, S, D, lmax) + 1, dtype=torch.float32, device=dev)
            out = torch.empty(n, dtype=torch.float32, device=dev)
            ops.check(lib.stove_objspn_fwd_any(ops.ptr(x), None, ops.ptr(lscope), ops.ptr(coef), ops.ptr(wsum), ops.ptr(wroot), ops.ptr(saved),
                                               ops.ptr(out), n, R, G, S, D, lmax, ops.stream()), 'stove_objspn_fwd_any')
        sv = saved[:n * (R * 4 * G + R * 2 * S + 1)].view(n, -1)
        ell = sv[:, :R * 4 * G].view(n, R, 2, 2, G)                       # [b][r][j][side][g]
        s_ = sv[:, R * 4 * G:R * 4 * G + R * 2 * S].view(n, R, 2, S)
        top = (s_[:, :, 1, :, None] + s_[:, :, 0, None, :]).reshape(n, R * S * S) + torch.log(wroot).reshape(1, -1)      # node k1 * S + k0
        pick = top.argmax(1)
        r, k1, k0 = pick // (S * S), (pick % (S * S)) // S, pick % S
        ar = torch.arange(n, device=dev)
        rec = torch.zeros(n, D, dtype=torch.float32, device=dev)
        mu = torch.cat([v.means for v in self.vector_list[0]])[self._plan(dev)['gidx']]                       # (R*4, Lmax, G)
        picks = [r]
        for j, k in ((0, k0), (1, k1)):
            e = ell[ar, r, j]                                                  # (n, 2, G)
            prod = (e[:, 1, :, None] + e[:, 0, None, :]).reshape(n, G * G)    # node g1 * G + g0
            child = (prod + torch.log(wsum.view(R, 2, G * G, S)[r, j, :, k])).argmax(1)
            g1, g0 = child // G, child % G
            picks += [g0, g1]
            for side, gsel in ((0, g0), (1, g1)):
                q = r * 4 + j * 2 + side                                       # leaf row
                px = lscope[q].long()                                          # (n, Lmax), -1 padded
                val = mu[q, :, :].gather(2, gsel.view(n, 1, 1).expand(n, lmax, 1))[..., 0]
                ok = px >= 0
                rec[ar[:, None].expand_as(px)[ok], px[ok]] = val[ok]
        rec = rec.clamp(0.0, 1.0)
        return (rec, torch.stack(picks, 1).int()) if return_pick else rec

    # ------------------------------------------------------------------ kernel plan
    def _make_plan(self):
        """Map the layered structure onto one of the two kernel shapes and precompute the
        integer tables.  They are plain attributes (not buffers: `Module.type(dtype)` would cast
        them to float), cached per device on first use."""
        self._kind = None
        self._plan_cpu, self._plan_dev = {}, {}
        vl = self.vector_list
        a = self.args
        if a.linear_sum_weights or not a.normalized_sums or a.gauss_min_mean is not None \
                or a.gauss_max_mean is not None or not (a.gauss_min_sigma < a.gauss_max_sigma):
            return
        root = self.output_vector
        if self.num_classes != 1 or not isinstance(root, SumVector):
            return
        leaves = list(vl[0])
        leaf_idx = {id(v): i for i, v in enumerate(leaves)}
        tuned = (len(vl) == 5 and self.num_dims == 100 and a.num_gauss == 10 and a.num_sums == 10 and len(root.inputs) == 6
                 and len(leaves) == 24 and len(vl[2]) == 12 and all(len(v.scope) == 25 for v in leaves)
                 and not getattr(self, '_no_tuned', False))
        if tuned:
            sums = list(vl[2])
            sum_idx = {id(v): i for i, v in enumerate(sums)}
            R = len(root.inputs)
            leaf_order, sum_order = [], []
            for prod in root.inputs:                       # replica r = r-th child of the root sum
                for s in prod.inputs:                      # (in1, in2) = sides 0, 1
                    if len(s.inputs) != 1:
                        return
                    sum_order.append(sum_idx[id(s)])
                    for leaf in s.inputs[0].inputs:
                        if len(leaf.scope) != 25:
                            return
                        leaf_order.append(leaf_idx[id(leaf)])
            scope = torch.tensor([leaves[i].scope for i in leaf_order], dtype=torch.int32)      # (24,25)
            slot = torch.zeros(R, 100, dtype=torch.int32)
            for r in range(R):
                for L in range(4):
                    for i, p in enumerate(leaves[leaf_order[r * 4 + L]].scope):
                        slot[r, p] = L * 25 + i
            self._plan_cpu = {'scope': scope, 'leaf_slot': slot, 'leaf_order': torch.tensor(leaf_order),
                              'sum_order': torch.tensor(sum_order)}
            self._kind = 'obj'
        elif (len(vl) == 5 and a.num_gauss <= 16 and a.num_sums <= 16 and self.num_dims <= 1024 and len(root.inputs) <= 8
              # the table-gradient kernel stages groups of 16 samples in LDS (csrc/spn_obj_generic.hip oa_shape_ok): 160 KiB
              and 2 * len(root.inputs) * (4 * a.num_gauss + 2 * a.num_sums) + 2 + 2 * self.num_dims <= 2560):
            # [amd] the object SPN's structure with any glimpse size / vector widths (config.patch_width / patch_height /
            # obj_spn_num_gauss / obj_spn_num_sums, reference config.py:99-100, 119-120): the general-size operator of
            # csrc/spn_obj_generic.hip.  Replica r = r-th child of the root; its leaves in (sum 0: leaf 0, leaf 1; sum 1: leaf 0, leaf 1) order
            sums = list(vl[2])
            sum_idx = {id(v): i for i, v in enumerate(sums)}
            R, D = len(root.inputs), self.num_dims
            start, off = {}, 0
            for i, leaf in enumerate(leaves):
                start[i] = off
                off += len(leaf.scope)
            leaf_lists, sum_order = [], []
            for prod in root.inputs:
                if len(prod.inputs) != 2:
                    return
                for s_ in prod.inputs:
                    if not isinstance(s_, SumVector) or len(s_.inputs) != 1 or len(s_.inputs[0].inputs) != 2:
                        return
                    sum_order.append(sum_idx[id(s_)])
                    for leaf in s_.inputs[0].inputs:
                        if not isinstance(leaf, GaussVector):
                            return
                        leaf_lists.append(leaf_idx[id(leaf)])
            lmax = max(len(leaves[i].scope) for i in leaf_lists)
            lscope = torch.full((R * 4, lmax), -1, dtype=torch.int32)
            gidx = torch.zeros(R * 4, lmax, dtype=torch.long)
            slot = torch.full((R, D), -1, dtype=torch.int32)
            for q, li in enumerate(leaf_lists):
                r, l = q // 4, q % 4
                for i, p in enumerate(leaves[li].scope):
                    lscope[q, i] = p
                    gidx[q, i] = start[li] + i
                    slot[r, p] = l * lmax + i
            if int(slot.min()) < 0:
                return
            self._plan_cpu = {'lscope': lscope, 'gidx': gidx, 'slot': slot, 'sum_order': torch.tensor(sum_order),
                              'pad': (lscope < 0)}
            self._any_shape = (R, a.num_gauss, a.num_sums, D, lmax)
            self._kind = 'obj_any'
        elif len(vl) == 3 and a.num_gauss == 6:
            # background SPN over any number of dimensions D (c x w x h: probabilistic_models.py:25-39): three replicas, each a
            # product of two Gaussian leaves that split the D pixels between them (512 / 512 for 32 x 32 frames, 1250 / 1250 for
            # the reference's stock 50 x 50 data).  gidx: row of pixel p of replica r in the concatenated leaf coefficients.
            R, D = len(root.inputs), self.num_dims
            if R != 3 or len(leaves) != 2 * R:
                return
            start, off = {}, 0
            for i, leaf in enumerate(leaves):
                start[i] = off
                off += len(leaf.scope)
            side = torch.zeros(R, D, dtype=torch.int32)
            gidx = torch.full((R, D), -1, dtype=torch.long)
            for r, prod in enumerate(root.inputs):
                if len(prod.inputs) != 2 or any(not isinstance(v, GaussVector) for v in prod.inputs):
                    return
                for s_, leaf in enumerate(prod.inputs):
                    li = leaf_idx[id(leaf)]
                    for i, p in enumerate(leaf.scope):
                        side[r, p] = s_
                        gidx[r, p] = start[li] + i
            if int(gidx.min()) < 0:
                return
            self._plan_cpu = {'side': side, 'gidx': gidx}
            self._kind = 'bg'

    def _force_general_plan(self):
        """Tests: plan the default object-SPN shape onto the general-size operator instead of the tuned kernels."""
        self._no_tuned = True
        self._make_plan()

    def _leaf_coef(self, flat=False):
        """(n_leaves, S, G, 3) = (a, b, c) with leaf log-density sum_p w_p (a x^2 + b x + c); flat: leaves of different scope
        sizes concatenated along their pixel rows -> (sum of S, G, 3)."""
        a = self.args
        comb = torch.cat if flat else torch.stack
        mu = comb([v.means for v in self.vector_list[0]])
        rho = comb([v.sigma_params for v in self.vector_list[0]])
        var = a.gauss_min_sigma + (a.gauss_max_sigma - a.gauss_min_sigma) * torch.sigmoid(rho)
        inv = 1.0 / var
        return torch.stack([-0.5 * inv, mu * inv, -0.5 * mu * mu * inv - 0.5 * torch.log(2.0 * math.pi * var)], -1)

    def _plan(self, device):
        key = str(device)
        if key not in self._plan_dev:
            self._plan_dev[key] = {k: v.to(device) for k, v in self._plan_cpu.items()}
        return self._plan_dev[key]

    def tables(self):
        """Baked float tables (differentiable functions of the parameters) + integer plan."""
        if self._kind == 'obj':
            pl = self._plan(self.output_vector.params.device)
            coef = self._leaf_coef()[pl['leaf_order']]                                    # (24,25,10,3)
            w = torch.stack([v.params for v in self.vector_list[2]])                      # (12,100,10)
            wsum = torch.softmax(w, 1)[pl['sum_order']]
            wroot = torch.softmax(self.output_vector.params, 0).view(6, 100)
            return (coef.contiguous(), wsum.contiguous(), wroot.contiguous(), pl['scope'], pl['leaf_slot'])
        if self._kind == 'obj_any':
            pl = self._plan(self.output_vector.params.device)
            R, G, S, D, lmax = self._any_shape
            coef = self._leaf_coef(flat=True)[pl['gidx']]                                # (R*4, Lmax, G, 3); padded rows: unused
            w = torch.stack([v.params for v in self.vector_list[2]])                      # (2R, G*G, S)
            wsum = torch.softmax(w, 1)[pl['sum_order']]
            wroot = torch.softmax(self.output_vector.params, 0).view(R, S * S)
            return (coef.contiguous(), wsum.contiguous(), wroot.contiguous(), pl['lscope'], pl['slot'], self._any_shape)
        if self._kind == 'bg':
            pl = self._plan(self.output_vector.params.device)
            coef = self._leaf_coef(flat=True)[pl['gidx']]                                # (3, D, 6, 3)
            wroot = torch.softmax(self.output_vector.params, 0).view(3, 36)
            return (coef.contiguous(), wroot.contiguous(), pl['side'])
        raise NotImplementedError(
            'RatSpn: no gfx950 kernel for this SPN shape (dims=%d); kernels exist for the STOVE '
            'object (Nx random_split(2,2), up to 16 gaussians / sums, up to 1024 dims within the 160 KiB LDS budget of the table gradients: 2 R (4 G + 2 S) + 2 D <= 2558) and background (any dims, 3x random_split(2,1), 6 gaussians) SPNs'
            % self.num_dims)

    # ------------------------------------------------------------------ evaluation
    def forward(self, inputs, marginalized=None):
        """Root log-density (B, 1); `marginalized` in [0,1] weighs each input's leaf term by
        (1 - clamp(marginalized, 0, 1))."""
        tabs = self.tables()
        if self._kind == 'obj':
            return ops.objspn_apply(inputs, marginalized, *tabs)
        if self._kind == 'obj_any':
            return ops.objspn_any_apply(inputs, marginalized, *tabs)
        return ops.bgspn_apply(inputs, marginalized, *tabs)


def _demo():
    rg = region_graph.RegionGraph(range(100), seed=42)
    for _ in range(6):
        rg.random_split(2, 2)
    args = SpnArgs()
    args.num_sums = 10
    args.num_gauss = 10
    spn = RatSpn(1, region_graph=rg, name='spn', args=args)
    print('parameters:', spn.num_params(), 'kernel:', spn._kind)


if __name__ == '__main__':
    _demo()
