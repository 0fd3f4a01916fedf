"""CPU oracle for the TransFuser training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``transfuser_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and there only as the checker / baseline.

Parity status: the reference (autonomousvision/transfuser @ branch 2022) ships
no tests and no golden vectors, and its arithmetic partly lives in un-vendored
third-party wheels (timm 0.5.4, mmdet 2.25.0, torch_scatter).  The pieces that
ARE importable in the authoring container (``team_code_transfuser/transfuser.py``
on top of ``oracle/timm_shim``; ``data.py:lidar_to_histogram_features`` is
restated verbatim against ``np.histogramdd``) are pinned by
``tests/golden/make_golden.py`` + ``tests/test_oracle_pinning.py``.  The timm /
mmdet / torch_scatter restatements (regnet.py, centernet.py) are "parity
unpinned": restated from the published algorithms of the pinned versions.
"""
