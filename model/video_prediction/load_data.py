from stove_amd.video_prediction.load_data import *  # noqa: F401,F403
from stove_amd.video_prediction.load_data import StoveDataset, load  # noqa: F401
