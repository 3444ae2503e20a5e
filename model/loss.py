from shapeclipper_amd.model.loss import *  # noqa: F401,F403  (drop-in alias of the reference's model/loss.py)
