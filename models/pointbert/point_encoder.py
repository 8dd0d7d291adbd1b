"""Alias of ppt_amd.models.pointbert.point_encoder under the reference's module path (models/pointbert/point_encoder.py)."""
from ppt_amd.models.pointbert.point_encoder import *          # noqa: F401,F403
from ppt_amd.models.pointbert.point_encoder import (Attention, Block, Mlp, PointTransformer, PointTransformer_partseg,  # noqa: F401
                                                    TransformerEncoder)
