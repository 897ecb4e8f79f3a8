from stove_amd.video_prediction.dynamics import *  # noqa: F401,F403
from stove_amd.video_prediction.dynamics import Dynamics  # noqa: F401
