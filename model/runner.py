from shapeclipper_amd.model.runner import *  # noqa: F401,F403  (drop-in alias of the reference's model/runner.py)
