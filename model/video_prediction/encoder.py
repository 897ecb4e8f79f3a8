from stove_amd.video_prediction.encoder import *  # noqa: F401,F403
from stove_amd.video_prediction.encoder import RnnStates  # noqa: F401
