from stove_amd.spn.region_graph import *  # noqa: F401,F403
