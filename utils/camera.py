from shapeclipper_amd.utils.camera import *  # noqa: F401,F403  (drop-in alias of the reference's utils/camera.py)
from shapeclipper_amd.utils.camera import pose  # noqa: F401
