"""torchvision stand-in: model.py:15 does `from torchvision import models` and never uses it on the hot path."""
from . import models  # noqa: F401
