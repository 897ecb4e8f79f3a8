from stove_amd.main import *  # noqa: F401,F403
from stove_amd.main import main, restore_model, build_config  # noqa: F401
