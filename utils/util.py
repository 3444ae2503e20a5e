from shapeclipper_amd.utils.util import *  # noqa: F401,F403  (drop-in alias of the reference's utils/util.py)
from shapeclipper_amd.utils.util import log, EasyDict  # noqa: F401
