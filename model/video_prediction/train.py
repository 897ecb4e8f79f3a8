from stove_amd.video_prediction.train import *  # noqa: F401,F403
from stove_amd.video_prediction.train import Trainer, AbstractTrainer  # noqa: F401
