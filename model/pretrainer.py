from shapeclipper_amd.model.pretrainer import *  # noqa: F401,F403  (drop-in alias of the reference's model/pretrainer.py)
