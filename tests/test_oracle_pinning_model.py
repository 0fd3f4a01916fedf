"""Pins oracle/model_cpu.py + oracle/centernet.py (rows a9-a13, f1 of SURVEY.md section 8) to the reference's OWN
team_code_transfuser/model.py, imported unmodified on top of oracle/mm_shim (cv2 / torchvision / mmcv / mmdet stand-ins),
oracle/timm_shim and oracle/scatter_shim:

 * golden: tests/golden/lidar_centernet_tiny.npz, written by tests/golden/make_golden.py FROM THE REFERENCE MODULE
   (LidarCenterNet.forward losses + gradients, get_targets, loss, decode_heatmap, forward_gru, forward_ego +
   get_bbox_local_metric, control_pid) - runs everywhere (CPU suite, GPU box);
 * live (authoring container only, /root/reference present): the same comparisons against the imported module, every
   parameter gradient included, plus one full-size regnety_032 forward at H=256.

What stays unpinned are the mmdet 2.25 / mmcv 1.5.3 LEAF functions (gaussian_radius, gen_gaussian_target, the four loss
formulas, top-k helpers): both sides resolve them to oracle/centernet.py (the packages are not installable here)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402
from oracle import centernet as oc, model_cpu, regnet as oreg  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "lidar_centernet_tiny.npz"))
TOL = 2e-5  # same PyTorch-CPU arithmetic on both sides; allows for a different BLAS thread count / reduction split
HEAD_KEYS = ("loss_center_heatmap", "loss_wh", "loss_offset", "loss_yaw_class", "loss_yaw_res", "loss_velocity", "loss_brake")
HAVE_REF = os.path.isdir("/root/reference/team_code_transfuser")
live = pytest.mark.skipif(not HAVE_REF, reason="reference checkout only exists in the authoring container")


def _close(a, b, what, tol=TOL):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol * max(1.0, b.abs().max().item() if b.numel() else 1.0), (what, err)


def _make_net():
    return oreg.RegNet(mg.TINY["widths"], mg.TINY["depths"], mg.TINY["group_w"], mg.TINY["se_ratio"])


def _oracle_model(cfg):
    torch.manual_seed(0)
    o = model_cpu.LidarCenterNet(cfg, 'cpu', 'transFuser', use_velocity=False, make_net=_make_net)
    mg.seeded_fill(o, 4321)
    return o


def _oracle_outputs():
    """Everything model_golden() records, computed by the oracle restatement."""
    out = {}
    cfg = mg.model_config()
    o = _oracle_model(cfg)
    o.train()
    b = mg.model_batch()
    losses = mg.call_model(o, b)
    sum(w * losses[k] for w, k in zip(mg.LOSS_WEIGHTS, cfg.detailed_losses)).backward()
    out["model_losses"] = np.array([float(losses[k].detach()) for k in cfg.detailed_losses])
    named = dict(o.named_parameters())
    for k in mg.MODEL_GRAD_KEYS:
        out["model_grad_" + k] = named[k].grad.numpy()
    out["_named"] = named
    lab = mg.target_labels()
    t, af = o.head.get_targets(lab, torch.zeros_like(lab[:, :, 0]), lab.sum(-1) == 0., (3, 1, 16, 16))
    for k, v in t.items():
        out["tgt_" + k] = v.numpy()
    out["tgt_avg_factor"] = np.array([int(af)])
    preds = mg.head_preds()
    l = o.head.loss(preds, lab, torch.zeros_like(lab[:, :, 0]), lab.sum(-1) == 0.)
    out["head_losses"] = np.array([float(l[k]) for k in HEAD_KEYS])
    boxes, labels = oc.decode_heatmap(preds, o.head.num_dir_bins, k=20, kernel=cfg.center_net_max_pooling_kernel)
    out["decode_boxes"], out["decode_labels"] = boxes.numpy(), labels.numpy()
    g = torch.Generator().manual_seed(8)
    z, tp = torch.randn(3, 512, generator=g), torch.randn(3, 2, generator=g) * 10
    with torch.no_grad():
        out["gru_wp"] = o.forward_gru(z, tp).numpy()
    o.eval()
    cfg.bb_confidence_threshold = 0.0
    cfg.top_k_center_keypoints = 20
    with torch.no_grad():
        wp, boxes, _ = model_cpu.forward_ego(o, b['rgb'][:1], b['lidar'][:1], b['target_point'][:1], b['target_point_image'][:1],
                                             b['ego_vel'][:1].reshape(-1, 1))
    out["ego_wp"] = wp.numpy()
    out["ego_boxes"] = np.stack([bb[0] for bb in boxes])
    out["ego_brake_conf"] = np.array([[bb[1], bb[2]] for bb in boxes])
    return out


@pytest.fixture(scope="module")
def oracle_out():
    return _oracle_outputs()


def test_forward_losses_and_gradients_match_reference_golden(oracle_out):
    """LidarCenterNet.forward (model.py:733-805): the 11 losses and a spread of parameter gradients (every head, pred_bev, join/GRU/output,
    decoders, backbone ends) + the L2 norm of EVERY parameter gradient."""
    _close(oracle_out["model_losses"], GOLD["model_losses"], "losses")
    for k in mg.MODEL_GRAD_KEYS:
        _close(oracle_out["model_grad_" + k], GOLD["model_grad_" + k], k)
    named = oracle_out["_named"]
    names = [str(n) for n in GOLD["model_grad_names"]]
    assert sorted(n for n, p in named.items() if p.grad is not None) == names
    norms = np.array([named[n].grad.double().norm().item() for n in names])
    assert np.all(np.abs(norms - GOLD["model_grad_norms"]) <= 1e-4 * np.maximum(GOLD["model_grad_norms"], 1e-6)), \
        [(n, a, b) for n, a, b in zip(names, norms, GOLD["model_grad_norms"]) if abs(a - b) > 1e-4 * max(b, 1e-6)][:5]


def test_get_targets_matches_reference_golden(oracle_out):
    """get_targets + angle2class (model.py:250-267,285-374): integer maps exactly, float maps exactly (same scalar arithmetic)."""
    for k in ("center_heatmap_target", "wh_target", "yaw_class_target", "yaw_res_target", "offset_target", "velocity_target",
              "brake_target", "wh_offset_target_weight"):
        a, b = oracle_out["tgt_" + k], GOLD["tgt_" + k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert np.array_equal(a, b), k
    assert int(oracle_out["tgt_avg_factor"][0]) == int(GOLD["tgt_avg_factor"][0]) > 1
    assert GOLD["tgt_wh_offset_target_weight"][2].sum() == 0          # the empty sample stays empty


def test_head_loss_decode_gru_match_reference_golden(oracle_out):
    _close(oracle_out["head_losses"], GOLD["head_losses"], "head losses (model.py:150-248)")
    assert np.array_equal(oracle_out["decode_labels"], GOLD["decode_labels"])
    _close(oracle_out["decode_boxes"], GOLD["decode_boxes"], "decode_heatmap (model.py:436-497)")
    _close(oracle_out["gru_wp"], GOLD["gru_wp"], "forward_gru (model.py:611-646)")


def test_forward_ego_matches_reference_golden(oracle_out):
    _close(oracle_out["ego_wp"], GOLD["ego_wp"], "forward_ego waypoints")
    assert oracle_out["ego_boxes"].shape == GOLD["ego_boxes"].shape and GOLD["ego_boxes"].shape[1:] == (6, 3)
    _close(oracle_out["ego_boxes"], GOLD["ego_boxes"], "get_bbox_local_metric (model.py:810-843)", 1e-4)
    _close(oracle_out["ego_brake_conf"], GOLD["ego_brake_conf"], "brake / confidence")


def test_product_control_pid_matches_reference_golden():
    """control_pid (model.py:648-683) of the PRODUCT class against the reference's three successive calls (stateful PID)."""
    from transfuser_amd.model import LidarCenterNet
    cfg = mg.model_config()
    prod = LidarCenterNet.__new__(LidarCenterNet)         # controllers only (built lazily by control_pid): no parameters, no device
    torch.nn.Module.__init__(prod)
    prod.config = cfg
    wp = torch.from_numpy(GOLD["ego_wp"])
    for i in range(3):
        s, t, b = prod.control_pid(wp + 0.5 * i, torch.tensor([1.0 + i]), bool(i == 2))
        _close([float(s), float(t), float(b)], GOLD["ego_pid"][i], "control_pid call %d" % i, 1e-5)


# ------------------------------------------------------------------------------------------------ live reference import
def _import_reference():
    for s in ("mm_shim", "scatter_shim", "timm_shim"):
        sys.path.insert(0, os.path.join(ROOT, "oracle", s))
    sys.path.insert(0, "/root/reference/team_code_transfuser")
    import timm
    timm.register("regnety_tiny", _make_net)
    import model as ref
    return ref


@live
def test_live_reference_model_all_gradients():
    """Same weights, same batch: the reference's LidarCenterNet vs oracle.model_cpu - 11 losses and ALL parameter gradients."""
    ref = _import_reference()
    cfg = mg.model_config()
    torch.manual_seed(0)
    r = ref.LidarCenterNet(cfg, 'cpu', 'transFuser', 'regnety_tiny', 'regnety_tiny', use_velocity=False)
    o = _oracle_model(cfg)
    assert set(r.state_dict()) == set(o.state_dict())
    mg.seeded_fill(r, 4321)
    for (ka, va), (kb, vb) in zip(sorted(r.state_dict().items()), sorted(o.state_dict().items())):
        assert ka == kb and torch.equal(va, vb)
    r.train(); o.train()
    for seed in (0, 1):
        b = mg.model_batch(seed=seed)
        for m in (r, o):
            for p in m.parameters():
                p.grad = None
        lr, lo = mg.call_model(r, b), mg.call_model(o, b)
        assert set(lr) == set(lo) == set(cfg.detailed_losses)
        for k in lr:
            _close(lo[k].detach(), lr[k].detach(), k)
        sum(w * lr[k] for w, k in zip(mg.LOSS_WEIGHTS, cfg.detailed_losses)).backward()
        sum(w * lo[k] for w, k in zip(mg.LOSS_WEIGHTS, cfg.detailed_losses)).backward()
        po = dict(o.named_parameters())
        n = 0
        for name, p in r.named_parameters():
            assert (p.grad is None) == (po[name].grad is None), name
            if p.grad is not None:
                _close(po[name].grad, p.grad, name)
                n += 1
        assert n > 300


@live
def test_live_reference_head_targets_loss_decode_gru():
    ref = _import_reference()
    cfg = mg.model_config()
    torch.manual_seed(0)
    r = ref.LidarCenterNet(cfg, 'cpu', 'transFuser', 'regnety_tiny', 'regnety_tiny', use_velocity=False)
    o = _oracle_model(cfg)
    mg.seeded_fill(r, 4321)
    for res, shape in ((64, (3, 1, 16, 16)), (256, (3, 1, 64, 64))):
        cfg.lidar_resolution_width = cfg.lidar_resolution_height = res
        lab = mg.target_labels(res)
        gl, ign = torch.zeros_like(lab[:, :, 0]), lab.sum(-1) == 0.
        tr, ar = r.head.get_targets([lab], [gl], [ign], shape)
        to, ao = o.head.get_targets(lab, gl, ign, shape)
        assert int(ar) == int(ao)
        for k in tr:
            assert tr[k].dtype == to[k].dtype and torch.equal(tr[k], to[k]), (res, k)
        preds = mg.head_preds(res=shape[-1])
        lr = r.head.loss(*[[p] for p in preds], [lab], gt_labels=[gl], gt_bboxes_ignore=[ign], img_metas=None)
        lo = o.head.loss(preds, lab, gl, ign)
        for k in HEAD_KEYS:
            _close(lo[k], lr[k], k)
    cfg.top_k_center_keypoints = 100
    preds = mg.head_preds(res=64)
    br = r.head.get_bboxes(*[[p] for p in preds])
    bo, lo_ = oc.decode_heatmap(preds, o.head.num_dir_bins, k=100, kernel=cfg.center_net_max_pooling_kernel)
    assert torch.equal(torch.stack([x[0] for x in br]), bo) and torch.equal(torch.stack([x[1] for x in br]), lo_)
    g = torch.Generator().manual_seed(8)
    z, tp = torch.randn(5, 512, generator=g), torch.randn(5, 2, generator=g) * 10
    with torch.no_grad():
        assert torch.equal(r.forward_gru(z, tp)[0], o.forward_gru(z, tp))


@live
def test_live_reference_model_regnety032_forward_h256():
    """Full-size LidarCenterNet (RegNetY-3.2GF x2, 4 GPT layers), B=1, H=256: the 11 losses of the reference's model.py == oracle."""
    ref = _import_reference()
    from transfuser_amd.config import GlobalConfig
    from transfuser_amd.data import synthetic_batch
    from oracle import hist
    cfg = GlobalConfig(); cfg.n_layer = 4; cfg.use_target_point_image = True
    cfg.embd_pdrop = cfg.attn_pdrop = cfg.resid_pdrop = 0.0
    torch.manual_seed(0)
    r = ref.LidarCenterNet(cfg, 'cpu', 'transFuser', 'regnety_032', 'regnety_032', use_velocity=False)
    o = model_cpu.LidarCenterNet(cfg, 'cpu', 'transFuser', use_velocity=False)
    sd = r.state_dict()
    g = torch.Generator().manual_seed(1)
    for k, v in sd.items():
        if v.dtype.is_floating_point and k.endswith('bn.weight'):
            v.copy_(torch.rand(v.shape, generator=g) * 0.5 + 0.5)
    o.load_state_dict(sd, strict=True)
    b = synthetic_batch(1, 256, 704, seed=3, hist_fn=hist.lidar_to_histogram_features, n_points=4096)
    r.train(); o.train()
    with torch.no_grad():
        lr, lo = mg.call_model(r, b), mg.call_model(o, b)
    for k in lr:
        _close(lo[k], lr[k], k)
