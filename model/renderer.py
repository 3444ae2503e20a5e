from shapeclipper_amd.model.renderer import *  # noqa: F401,F403  (drop-in alias of the reference's model/renderer.py)
