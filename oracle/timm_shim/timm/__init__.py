"""Minimal ``timm`` stand-in so the reference's transfuser.py (transfuser.py:5,380,442) imports
unmodified in the authoring container.  Test infrastructure; resolves through oracle.regnet / oracle.resnet."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
from oracle import regnet as _regnet  # noqa: E402
from oracle import resnet as _resnet  # noqa: E402
from oracle import convnext as _convnext  # noqa: E402

_REGISTRY = {}


def register(name, fn):
    _REGISTRY[name] = fn


def create_model(architecture, pretrained=False, **kw):
    in_chans = kw.get("in_chans", 3)        # late_fusion.py:155 builds the LiDAR trunk with in_chans=...
    if architecture in _REGISTRY:
        try:
            return _REGISTRY[architecture](in_chans=in_chans)
        except TypeError:
            assert in_chans == 3, "registered test net does not take in_chans"
            return _REGISTRY[architecture]()
    if architecture == "regnety_032":
        return _regnet.regnety_032(in_chans)  # pretrained weights need network: seeded random init instead
    if architecture in _resnet.ARCH:         # the reference's DEFAULT trunks (transfuser.py:15); used under timm's own attribute names
        return _resnet.ARCH[architecture](in_chans)
    if architecture in _convnext.ARCH:       # the re-labelling branch of transfuser.py:395-416 / 457-471
        return _convnext.ARCH[architecture](in_chans)
    raise ValueError("timm shim only provides regnety_032, resnet18/34/50, convnext_tiny/small/base (+registered test nets), got %r" % architecture)
