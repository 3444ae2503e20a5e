from shapeclipper_amd.data.synthetic import *  # noqa: F401,F403
