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


def _write_route(route_dir, n_frames, seed, Hs=48, Ws=160, S=500, N=600):
    """One route in the reference's on-disk format (team_code_autopilot/data_agent.py:243-272, autopilot.py:304-345), tiny images."""
    import json
    from PIL import Image
    rng = np.random.default_rng(seed)
    for sub in ("rgb", "depth", "semantics", "topdown", "lidar", "label_raw", "measurements"):
        os.makedirs(os.path.join(route_dir, sub))

    def pose(th, x, y):
        m = np.eye(4); m[0, 0] = m[1, 1] = np.cos(th); m[0, 1] = -np.sin(th); m[1, 0] = np.sin(th); m[0, 3] = x; m[1, 3] = y
        return m

    for t in range(n_frames):
        Image.fromarray(rng.integers(0, 256, (Hs, Ws, 3), dtype=np.uint8)).save(os.path.join(route_dir, "rgb", "%04d.png" % t))
        Image.fromarray(rng.integers(0, 256, (Hs, Ws, 3), dtype=np.uint8)).save(os.path.join(route_dir, "depth", "%04d.png" % t))
        Image.fromarray(rng.integers(0, 28, (Hs, Ws), dtype=np.uint8)).save(os.path.join(route_dir, "semantics", "%04d.png" % t))
        Image.fromarray(rng.integers(0, 256, (S, S, 3), dtype=np.uint8)).save(os.path.join(route_dir, "topdown", "encoded_%04d.png" % t))
        pts = np.stack([rng.uniform(-25, 25, N), rng.uniform(-8, 40, N), rng.uniform(-4, 1, N), rng.uniform(0, 1, N)], -1).astype(np.float32)
        np.save(os.path.join(route_dir, "lidar", "%04d.npy" % t), np.array([t, pts], dtype=object), allow_pickle=True)
        ego = pose(0.3 + 0.02 * t, 10.0 + 1.5 * t, -4.0 + 0.2 * t)
        objs = [dict(id=7, ego_matrix=ego.tolist(), extent=[0.7, 2.4, 1.0], position=[0.0, 0.0, 0.0], yaw=0.0, speed=3.0, brake=0.0, num_points=50, distance=0.0)]
        for j in range(4):
            objs.append(dict(id=100 + j, ego_matrix=pose(0.1 * j, 3.0 * j, 2.0).tolist(), extent=[0.8, 2.0 + 0.1 * j, 0.9],
                             position=[float(rng.uniform(-12, 12)), float(rng.uniform(2, 28)), 0.0], yaw=float(rng.uniform(-3, 3)), speed=float(rng.uniform(0, 8)),
                             brake=float(j % 2), num_points=10 + j, distance=5.0))
        json.dump(objs, open(os.path.join(route_dir, "label_raw", "%04d.json" % t), "w"))
        json.dump(dict(theta=0.3 + 0.02 * t, x=10.0 + 1.5 * t, y=-4.0 + 0.2 * t, x_command=30.0, y_command=5.0, speed=3.5, ego_matrix=ego.tolist()),
                  open(os.path.join(route_dir, "measurements", "%04d.json" % t), "w"))


def test_train_main_on_a_reference_layout_dataset(tmp_path, monkeypatch):
    """End to end: train.main on a tiny ON-DISK dataset in the reference's layout root/<scenario>/<town>/<route>/{rgb,depth,semantics,topdown,
    lidar,label_raw,measurements} (config.py:209-243) - GlobalConfig enumerates train / val towns (Town02 / Town05 withheld), CARLA_Data
    decodes the raw frames (through the --use_disk_cache directory the second time), the Trainer runs GpuBatchPrep on the collated raw batch,
    one epoch trains and validates, checkpoints + args.txt + the loss log are written (train.py:27-211)."""
    import json
    import torch
    import model_cases as mc  # noqa: F401  (registers regnety_tiny)
    from transfuser_amd import train as T
    from transfuser_amd.config import GlobalConfig
    root = tmp_path / "data"
    for town, seed in (("Town01_Scenario1", 1), ("Town05_Scenario1", 2)):
        _write_route(str(root / "Scenario1" / town / "route0"), 11, seed)
    cfg = GlobalConfig(root_dir=str(root), setting='02_05_withheld')
    assert [os.path.basename(p) for p in cfg.train_data] == ["Town01_Scenario1"] and [os.path.basename(p) for p in cfg.val_data] == ["Town05_Scenario1"]
    assert len(GlobalConfig(root_dir=str(root), setting='all').train_data) == 2
    monkeypatch.setenv("SCRATCH", str(tmp_path / "scratch"))
    monkeypatch.setattr(GlobalConfig, "img_resolution", (32, 64))
    monkeypatch.setattr(GlobalConfig, "img_width", 64)
    argv = ["--root_dir", str(root), "--setting", "02_05_withheld", "--epochs", "1", "--batch_size", "2", "--parallel_training", "0", "--logdir", str(tmp_path / "log"),
            "--image_architecture", "regnety_tiny", "--lidar_architecture", "regnety_tiny", "--n_layer", "1", "--num_workers", "0", "--val_every", "1", "--use_disk_cache", "1"]
    tr = T.main(argv)
    assert tr.cur_epoch == 1 and len(tr.dataloader_train.dataset) == 2 and len(tr.dataloader_val.dataset) == 2
    logdir = tmp_path / "log" / "transfuser"
    recs = [json.loads(l) for l in open(logdir / "losses.jsonl")]
    assert any("loss_total" in r for r in recs) and any("val_loss_total" in r for r in recs) and all(np.isfinite(v) for r in recs for v in r.values())
    assert (logdir / "model_1.pth").exists() and (logdir / "optimizer_1.pth").exists() and json.load(open(logdir / "args.txt"))["setting"] == "02_05_withheld"
    cached = os.listdir(tmp_path / "scratch" / "dataset_cache")
    assert len(cached) == 4                                       # 2 training + 2 validation frames, decoded once
    osd = torch.load(logdir / "optimizer_1.pth")
    assert "layout" in osd and osd["layout"][0][0]                # moments are saved with their (name, offset, numel) layout
    ds = tr.dataloader_train.dataset
    a, b = ds[0], ds[0]                                           # second access comes from the cache: same decoded arrays
    assert torch.equal(a["rgb_u8"], b["rgb_u8"]) and torch.equal(a["lidar_raw"], b["lidar_raw"]) and a["label"].shape == (20, 7)
    with pytest.raises(ValueError):
        T.main(["--use_disk_cache", "1", "--root_dir", "synthetic:4", "--parallel_training", "0"])
