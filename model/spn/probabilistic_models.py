from stove_amd.spn.probabilistic_models import *  # noqa: F401,F403
from stove_amd.spn.probabilistic_models import _get_obj_spn, _get_bg_spn  # noqa: F401
