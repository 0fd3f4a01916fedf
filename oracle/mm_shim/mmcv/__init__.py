"""mmcv-full==1.5.3 stand-in (README.md:64 of the reference): the four symbols of model.py:20-22."""
