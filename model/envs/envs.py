from stove_amd.envs.envs import *  # noqa: F401,F403
