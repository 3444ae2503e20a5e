from shapeclipper_amd.utils.options import *  # noqa: F401,F403  (drop-in alias of the reference's utils/options.py)
from shapeclipper_amd.utils.options import set, parse_arguments, save_options_file  # noqa: F401
