"""mmdet.models.utils.gaussian_target (2.25.0) -> the restatements in oracle/centernet.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..", "..", ".."))
from oracle.centernet import (gaussian_radius, gen_gaussian_target as _gen, get_local_maximum, get_topk_from_heatmap,  # noqa: E402,F401
                              transpose_and_gather_feat)


def gen_gaussian_target(heatmap, center, radius, k=1):
    assert k == 1
    _gen(heatmap, center, radius)
    return heatmap
