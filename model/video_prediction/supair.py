from stove_amd.video_prediction.supair import *  # noqa: F401,F403
from stove_amd.video_prediction.supair import Supair  # noqa: F401
