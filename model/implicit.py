from shapeclipper_amd.model.implicit import *  # noqa: F401,F403  (drop-in alias of the reference's model/implicit.py)
