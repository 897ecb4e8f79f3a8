"""Object and background SPN factories (reference model/spn/probabilistic_models.py:8-39)."""
from .rat_torch import RatSpn, SpnArgs
from .region_graph import RegionGraph


def _get_obj_spn(c, seed):
    """SPN over one c.channels x patch_width x patch_height glimpse: six two-level random binary splits."""
    rg = RegionGraph(range(c.channels * c.patch_width * c.patch_height), seed=seed)
    for _ in range(6):
        rg.random_split(2, 2)
    args = SpnArgs()
    args.num_gauss = c.obj_spn_num_gauss
    args.num_sums = c.obj_spn_num_sums
    args.gauss_min_sigma = c.obj_min_var
    args.gauss_max_sigma = c.obj_max_var
    return RatSpn(1, region_graph=rg, args=args, name='obj-spn')


def _get_bg_spn(c, seed):
    """SPN over the whole frame: three one-level random binary splits, 6 gaussians per leaf."""
    rg = RegionGraph(range(c.width * c.height * c.channels), seed=seed)
    for _ in range(3):
        rg.random_split(2, 1)
    args = SpnArgs()
    args.num_gauss = 6
    args.num_sums = 3
    args.gauss_min_sigma = c.bg_min_var
    args.gauss_max_sigma = c.bg_max_var
    return RatSpn(1, region_graph=rg, args=args, name='bg-spn')


class _FixedGauss:
    """The reference's fixed-Gaussian debug models (probabilistic_models.py:42-90): no parameters, every pixel scored under one
    Normal and weighted by (1 - marg) as it comes; `_kind = 'simple'` sends Supair.likelihood through its composed path."""
    _kind = 'simple'

    def __init__(self, mean, var):
        self.mean, self.var = float(mean), float(var)        # `var` is used as the scale of the Normal, as in the reference

    def forward(self, img_flat, marg_flat):
        """(n, d), (n, d) -> (n, 1) (csrc/spn_obj_generic.hip gauss_ll_fwd_k / _bwd_k)."""
        from .. import ops
        return ops.gauss_ll(img_flat, marg_flat, self.mean, self.var)

    __call__ = forward


def _get_simple_bg(c):
    """SimpleBG (config.debug_bg_model): a black background, Normal(0, debug_simple_bg_var)."""
    return _FixedGauss(0.0, c.debug_simple_bg_var)


def _get_simple_obj(c):
    """SimpleObj (config.debug_obj_spn): white objects, Normal(0.8, debug_simple_obj_var)."""
    return _FixedGauss(0.8, c.debug_simple_obj_var)
