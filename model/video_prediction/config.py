from stove_amd.video_prediction.config import *  # noqa: F401,F403
from stove_amd.video_prediction.config import StoveConfig  # noqa: F401
