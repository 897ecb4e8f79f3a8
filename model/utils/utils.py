from stove_amd.utils.utils import *  # noqa: F401,F403
