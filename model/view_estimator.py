from shapeclipper_amd.model.view_estimator import *  # noqa: F401,F403  (drop-in alias of the reference's model/view_estimator.py)
