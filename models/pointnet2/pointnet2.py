"""Alias of ppt_amd.models.pointnet2.pointnet2 under the reference's module path (models/pointnet2/pointnet2.py)."""
from ppt_amd.models.pointnet2.pointnet2 import *          # noqa: F401,F403
from ppt_amd.models.pointnet2.pointnet2 import (Pointnet2_Msg, Pointnet2_Ssg, PointNetSetAbstraction,  # noqa: F401
                                                PointNetSetAbstractionMsg)
