"""Alias of ppt_amd.models.pointbert.pointnet2_utils under the reference's module path."""
from ppt_amd.models.pointbert.pointnet2_utils import *          # noqa: F401,F403
from ppt_amd.models.pointbert.pointnet2_utils import (DGCNN_Propagation, PointNetFeaturePropagation, farthest_point_sample,  # noqa: F401
                                                      index_points, knn_point, square_distance)
