"""Alias of ppt_amd.models.ULIP_models under the reference's module path (models/ULIP_models.py)."""
from ppt_amd.models.ULIP_models import *          # noqa: F401,F403
from ppt_amd.models.ULIP_models import (ULIP_PointBERT, ULIP_PN_MSG, ULIP_PN_SSG, ULIP_PointBERT_partseg, ULIP_WITH_IMAGE, PromptLearner, Transformer,  # noqa: F401
                                        ResidualAttentionBlock, LayerNorm, QuickGELU, get_metric_names,
                                        cfg_from_yaml_file, dataset_classnames, tokenize_prompts, unfreeze_list)
