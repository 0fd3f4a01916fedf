"""GPU-side batch preparation (csrc/dataprep.cpp, transfuser_amd.data.GpuBatchPrep) vs the reference's own data.py functions
(tests/golden/dataprep.npz), on the host emulator; the same check runs on the MI355X in tests/test_kernels_gpu.py."""
import os

import numpy as np
import pytest

import dataprep_cases as dc


@pytest.fixture(autouse=True)
def _backend(emu_backend):
    yield


def test_dataprep_matches_reference_functions():
    dc.check_dataprep("cpu")


@pytest.mark.skipif(not os.path.isdir("/root/reference/team_code_transfuser"), reason="reference checkout only exists in the authoring container")
def test_golden_is_current_with_live_reference_source():
    """The committed fixture equals what the reference's data.py source produces NOW (ast-extracted functions)."""
    live = dc.mg.dataprep_golden()
    gold = np.load(os.path.join(dc.HERE, "golden", "dataprep.npz"))
    for k in gold.files:
        assert np.array_equal(gold[k], live[k]), k


def test_synthetic_dataset_and_cli_parser():
    from transfuser_amd.data import SyntheticDataset, make_datasets
    from transfuser_amd.train import build_parser
    from transfuser_amd.config import GlobalConfig
    tr, va = make_datasets("synthetic:16", GlobalConfig(), height=160)
    assert len(tr) == 16 and len(va) == 2
    a, b = tr[3], tr[3]
    assert all((a[k] == b[k]).all() for k in a) and a["rgb"].shape == (3, 160, 704) and a["lidar"].shape == (2, 256, 256)
    args = build_parser().parse_args(["--batch_size", "10", "--backbone", "latentTF", "--zero_redundancy_optimizer", "1"])
    ref_flags = ["id", "epochs", "lr", "batch_size", "logdir", "load_file", "start_epoch", "setting", "root_dir", "schedule", "schedule_reduce_epoch_01",
                 "schedule_reduce_epoch_02", "backbone", "image_architecture", "lidar_architecture", "use_velocity", "n_layer", "wp_only", "use_target_point_image",
                 "use_point_pillars", "parallel_training", "val_every", "no_bev_loss", "sync_batch_norm", "zero_redundancy_optimizer", "use_disk_cache"]
    assert all(hasattr(args, f) for f in ref_flags)        # every flag of team_code_transfuser/train.py:30-70
    assert (args.epochs, args.lr, args.schedule_reduce_epoch_01, args.schedule_reduce_epoch_02, args.val_every) == (41, 1e-4, 30, 40, 5)
