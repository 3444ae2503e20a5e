from shapeclipper_amd.model.graph import *  # noqa: F401,F403  (drop-in alias of the reference's model/graph.py)
