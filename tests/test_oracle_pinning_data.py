"""oracle/correspondences.py pinned to the reference's lidar_bev_cam_correspondences / correspondences_at_one_scale
(team_code_transfuser/data.py:632-842): the committed golden (tests/golden/correspondences.npz, written by make_golden.py FROM the
reference's source) and, in the authoring container, the live source.  Cells with <= 5 points are compared for equality (the reference is
deterministic there); cells with more draw random.sample from Python's global generator: each of the five rows must be an entry of the
cell's list, no more often than it occurs there."""
import os
import sys
from collections import Counter

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402
from oracle import correspondences as oc  # noqa: E402


def check_against(ref_bev, ref_cam, cloud, seed=0):
    lb, lc = oc.cell_lists(cloud)
    bev, cam = oc.lidar_bev_cam_correspondences(cloud, seed=seed, sample=0)
    exact = crowded = 0
    for lst, ref, mine, ncell in ((lb, ref_bev.reshape(64, 5, 2), bev.reshape(64, 5, 2), 64), (lc, ref_cam.reshape(110, 5, 2), cam.reshape(110, 5, 2), 110)):
        for cell in range(ncell):
            entries = lst.get(cell, [])
            if len(entries) <= 5:
                want = np.zeros((5, 2), np.int64)
                want[:len(entries)] = np.array(entries, np.int64).reshape(-1, 2)
                assert np.array_equal(ref[cell], want), ("the restatement's list differs from the reference", cell, entries, ref[cell].tolist())
                assert np.array_equal(mine[cell], want), (cell, entries, mine[cell].tolist())
                exact += 1
            else:
                have = Counter(entries)
                for rows in (ref[cell], mine[cell]):
                    drawn = Counter(map(tuple, rows.tolist()))
                    assert all(drawn[k] <= have.get(k, 0) for k in drawn), (cell, rows.tolist())
                crowded += 1
    return exact, crowded


def test_correspondences_oracle_matches_reference_golden():
    gold = np.load(os.path.join(HERE, "golden", "correspondences.npz"))
    clouds = mg.correspondence_clouds()
    e, c = check_against(gold["corr_sparse_bev"], gold["corr_sparse_cam"], clouds["sparse"])
    assert e >= 160 and c <= 10, (e, c)            # the sparse cloud exercises the deterministic branch almost everywhere
    e, c = check_against(gold["corr_dense_bev"], gold["corr_dense_cam"], clouds["dense"], seed=5)
    assert c >= 100, (e, c)                        # the dense one the sampled branch


def test_crowded_cells_are_sampled_uniformly():
    """The counter-based draw that replaces random.sample: over many seeds every entry of a crowded cell is picked about equally often and
    a draw never repeats an entry."""
    cloud = mg.correspondence_clouds()["dense"]
    bx, by, cx, cy, key = oc.project_pairs(cloud)
    cells = bx * 8 + by
    c = int(np.bincount(cells, minlength=64).argmax())
    m = np.nonzero(cells == c)[0]
    hits = Counter()
    for seed in range(400):
        pr = oc.priority(seed, 0, 0, c, key[m])
        pick = np.lexsort((key[m], pr))[:5]
        assert len(set(pick.tolist())) == 5
        hits.update(pick.tolist())
    expect = 400 * 5 / len(m)
    assert max(hits.values()) <= expect * 2.5 + 8 and len(hits) >= 0.9 * len(m), (len(m), expect, max(hits.values()), len(hits))


@pytest.mark.skipif(not os.path.isdir("/root/reference/team_code_transfuser"), reason="reference checkout only exists in the authoring container")
def test_correspondences_golden_is_current_with_live_reference_source():
    live = mg.correspondences_golden()
    gold = np.load(os.path.join(HERE, "golden", "correspondences.npz"))
    for k in gold.files:
        assert np.array_equal(gold[k], live[k]), k
