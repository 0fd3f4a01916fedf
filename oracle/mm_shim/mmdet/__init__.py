"""mmdet==2.25.0 stand-in (environment.yml:28 of the reference): the symbols of model.py:24-30.  Leaf arithmetic lives in
oracle/centernet.py (restated from the published 2.25.0 sources; parity unpinned)."""
