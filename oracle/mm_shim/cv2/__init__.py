"""cv2 stand-in: model.py:3 imports it for the visualisation helpers only (model.py:877-1031), which the training /
inference hot path never calls.  Any use raises."""


def __getattr__(name):
    raise AttributeError("cv2 shim: visualisation only (model.py:845-1031) is out of scope; tried cv2.%s" % name)
