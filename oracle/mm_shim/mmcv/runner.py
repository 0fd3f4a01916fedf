"""mmcv.runner: force_fp32 casts fp16 arguments to fp32 when ``self.fp16_enabled``; the reference sets
fp16_enabled = config.fp16_enabled = False (model.py:90) -> identity decorator."""
import functools
from torch import nn


def force_fp32(apply_to=None, out_fp16=False):
    def wrap(fn):
        @functools.wraps(fn)
        def inner(self, *a, **k):
            assert not getattr(self, "fp16_enabled", False), "mm_shim.force_fp32: fp16 path not restated"
            return fn(self, *a, **k)
        return inner
    return wrap


class BaseModule(nn.Module):
    """mmcv.runner.BaseModule: nn.Module + init_cfg bookkeeping; init_weights() is never called by the reference."""

    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = init_cfg
