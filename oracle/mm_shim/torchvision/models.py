def __getattr__(name):
    raise AttributeError("torchvision.models shim: only the regnety_032 path (timm) is in scope; tried %s" % name)
