from stove_amd.video_prediction.stove import *  # noqa: F401,F403
from stove_amd.video_prediction.stove import Stove  # noqa: F401
