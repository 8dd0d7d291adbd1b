"""Alias of ppt_amd.models.pointbert.misc under the reference's module path (models/pointbert/misc.py)."""
from ppt_amd.models.pointbert.misc import *          # noqa: F401,F403
from ppt_amd.models.pointbert.misc import farthest_point_sample, fps, index_points  # noqa: F401
