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
