from stove_amd.spn.rat_torch import *  # noqa: F401,F403
