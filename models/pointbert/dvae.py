"""Alias of ppt_amd.models.pointbert.dvae under the reference's module path (models/pointbert/dvae.py)."""
from ppt_amd.models.pointbert.dvae import *          # noqa: F401,F403
from ppt_amd.models.pointbert.dvae import Encoder, Group, knn_point, square_distance  # noqa: F401
