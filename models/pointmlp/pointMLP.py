"""Alias of ppt_amd.models.pointmlp.pointMLP under the reference's module path (models/pointmlp/pointMLP.py)."""
from ppt_amd.models.pointmlp.pointMLP import *          # noqa: F401,F403
from ppt_amd.models.pointmlp.pointMLP import (ConvBNReLU1D, ConvBNReLURes1D, LocalGrouper, Model, PosExtraction,  # noqa: F401
                                              PreExtraction, get_activation, pointMLP, pointMLPElite)
