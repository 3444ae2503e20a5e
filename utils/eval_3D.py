from shapeclipper_amd.utils.eval_3D import *  # noqa: F401,F403  (drop-in alias of the reference's utils/eval_3D.py)
