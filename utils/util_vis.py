from shapeclipper_amd.utils.util_vis import *  # noqa: F401,F403  (drop-in alias of the reference's utils/util_vis.py)
