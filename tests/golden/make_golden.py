#!/usr/bin/env python
"""Generates tests/golden/*.npz from the REFERENCE ITSELF: team_code_transfuser/transfuser.py imported unmodified
from /root/reference on top of oracle/timm_shim (authoring container only; /root/reference does not exist on the
GPU box, which is why the vectors are committed).  Re-run: python tests/golden/make_golden.py

Weights are not stored: both sides fill every state_dict entry from one seeded generator in sorted-key order
(`seeded_fill`), so the fixture only holds inputs' seeds and the reference's outputs."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
TINY = dict(widths=[24, 48, 72, 96], depths=[1, 2, 1, 1], group_w=24, se_ratio=0.25)


def seeded_fill(module, seed=1234):
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for k in sorted(sd):
            v = sd[k]
            if not v.dtype.is_floating_point:
                continue
            if k.endswith("running_var"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif "bn" in k and k.endswith("weight"):
                v.copy_(torch.rand(v.shape, generator=g) * 0.5 + 0.75)
            else:
                v.copy_(torch.randn(v.shape, generator=g) * (0.3 / max(1.0, float(np.sqrt(v[0].numel()))) if v.dim() > 1 else 0.1))


def golden_config():
    from transfuser_amd.config import GlobalConfig
    cfg = GlobalConfig()
    cfg.n_layer = 2
    cfg.use_target_point_image = True
    cfg.embd_pdrop = cfg.attn_pdrop = cfg.resid_pdrop = 0.0
    return cfg


def golden_inputs(seed=7):
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(0, 256, (2, 3, 64, 128), generator=g).float(), torch.rand(2, 3, 64, 64, generator=g), torch.rand(2, 1, generator=g) * 8)


def geo_config():
    """Geometric fusion (BASELINE config 4) at toy size: anchors 2x3 / 3x3 so the FIXED x8/x4/x2/x1 factors of
    geometric_fusion.py:139,177,216 fit a 64x96 image and a 96x96 BEV."""
    cfg = golden_config()
    cfg.img_vert_anchors, cfg.img_horz_anchors, cfg.lidar_vert_anchors, cfg.lidar_horz_anchors = 2, 3, 3, 3
    cfg.n_embd = 32
    return cfg


def geo_inputs(seed=11):
    g = torch.Generator().manual_seed(seed)
    bev = torch.stack((torch.randint(0, 3, (2, 3, 3, 5), generator=g), torch.randint(0, 2, (2, 3, 3, 5), generator=g)), -1)
    cam = torch.stack((torch.randint(0, 3, (2, 3, 2, 5), generator=g), torch.randint(0, 3, (2, 3, 2, 5), generator=g)), -1)
    bev[0, 0, :2] = 0
    cam[1, 1] = 0
    return (torch.randint(0, 256, (2, 3, 64, 96), generator=g).float(), torch.rand(2, 3, 96, 96, generator=g), torch.rand(2, 1, generator=g) * 8, bev, cam)


def geo_golden(ref_mod):
    """Outputs AND a few parameter gradients of the reference's GeometricFusionBackbone (train mode)."""
    cfg = geo_config()
    out = {}
    for use_vel in (0, 1):
        torch.manual_seed(0)
        m = ref_mod.GeometricFusionBackbone(cfg, "regnety_tiny", "regnety_tiny", use_velocity=use_vel)
        seeded_fill(m)
        m.train()
        img, lid, vel, bev, cam = geo_inputs()
        feats, grid, fused = m(img, lid, vel, bev, cam)
        (feats[0].square().mean() + grid.square().mean() + fused.square().mean()).backward()
        tag = "geo_vel%d" % use_vel
        for i, f in enumerate(feats):
            out["%s_p%d" % (tag, i + 2)] = f.detach().numpy()
        out[tag + "_grid"] = grid.detach().numpy()
        out[tag + "_fused"] = fused.detach().numpy()
        for n in GEO_GRAD_KEYS + (("vel_emb2.weight",) if use_vel else ()):
            out["%s_grad_%s" % (tag, n)] = dict(m.named_parameters())[n].grad.numpy()
        assert m.lidar_conv4.weight.grad is None   # quirk Q4
    return out


GEO_GRAD_KEYS = ("image_conv1.weight", "lidar_conv3.weight", "image_projection2.2.weight", "lidar_projection4.4.bias", "lidar_deconv1.weight",
                 "image_deconv4.bias", "image_encoder.features.s1.b1.conv1.conv.weight", "lidar_encoder._model.conv1.weight")


PILLAR_KW = dict(min_x=-16, max_x=16, min_y=-32, max_y=0, pixels_per_meter=8)


def pillar_inputs(seed=21):
    g = torch.Generator().manual_seed(seed)
    pts = torch.stack([torch.rand(2, 2000, generator=g) * 40 - 20, torch.rand(2, 2000, generator=g) * 40 - 36, torch.rand(2, 2000, generator=g) * 5 - 4,
                       torch.rand(2, 2000, generator=g)], -1)
    pts[0, :20, 0] = torch.nextafter(torch.tensor(16.0), torch.tensor(0.0)); pts[0, :20, 1] = -1.0     # x_idx = 256 edge (clamped)
    pts[1, :40, :2] = torch.round(pts[1, :40, :2] * 8) / 8                                              # exactly on cell edges
    return pts, torch.tensor([2000, 1500], dtype=torch.int32)


def pillar_golden(ref_mod):
    """Reference PointPillarNet (point_pillar.py imported unmodified on top of oracle/scatter_shim): sparse canvas, pillar rows, grads."""
    torch.manual_seed(0)
    m = ref_mod.PointPillarNet(9, [32, 32], **PILLAR_KW)
    seeded_fill(m, 77)
    m.train()
    pts, num = pillar_inputs()
    canvas = m(pts, num)
    (canvas * torch.linspace(0.5, 1.5, 256).view(1, 1, 1, 256)).sum().backward()
    nz = torch.nonzero(canvas[:, 0:1].abs() + canvas.abs().sum(1, keepdim=True) > 0)[:, [0, 2, 3]]
    nz = torch.unique(nz, dim=0)
    out = dict(pillar_cells=nz.numpy().astype(np.int16), pillar_feat=canvas.detach()[nz[:, 0], :, nz[:, 1], nz[:, 2]].numpy(),
               pillar_nnz=np.array([(canvas != 0).sum().item()]))
    for n, p in m.named_parameters():
        out["pillar_grad_" + n] = p.grad.numpy()
    return out


# ---------------------------------------------------------------- model.py (LidarCenterNet / LidarCenterNetHead), rows a9-a13 + f1
def model_config(lidar_res=64):
    """Toy-size LidarCenterNet config shared by the golden generator and the pinning tests (tests/model_cases.tiny_config values)."""
    cfg = golden_config()
    cfg.lidar_resolution_width = cfg.lidar_resolution_height = lidar_res
    cfg.bev_resolution_width = cfg.bev_resolution_height = 40
    return cfg


def model_batch(B=2, H=32, W=64, lidar_res=64, bev_res=40, seed=0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import model_cases as mc
    return mc.small_batch(B, H, W, lidar_res, bev_res, seed)


def target_labels(res=64):
    """(3, 20, 7) label tensor exercising get_targets (model.py:285-374): ignored all-zero rows, Gaussians clipped at every border,
    two boxes landing on the same cell (later row overwrites), a tiny box (radius clamps to 2), yaw at the bin edges / negative, brake 0/1."""
    lab = torch.zeros(3, 20, 7)
    rows = [  # x, y, w, h, yaw, speed, brake   (pixels of the res x res BEV frame)
        (0.4, 0.7, 9.0, 5.0, 0.0, 1.5, 0), (res - 0.6, res - 0.3, 12.0, 20.0, 3.1, 0.0, 1), (0.9, res - 1.2, 4.0, 4.0, -3.1, 7.9, 0),
        (res - 1.1, 1.9, 30.0, 8.0, 1.5707963, 2.0, 1), (31.2, 17.9, 1.0, 1.0, -0.2617994, 3.0, 0), (31.9, 17.1, 16.0, 6.0, 0.2617994, 4.0, 1),
        (12.49, 40.51, 7.5, 7.5, 2.8797933, 5.5, 0), (50.0, 50.0, 63.0, 63.0, -1.0, 6.5, 1)]
    for i, r in enumerate(rows):
        lab[0, i] = torch.tensor(r)
    lab[0, 10] = torch.tensor((20.3, 9.6, 10.0, 3.0, 0.7, 1.0, 1))          # after a gap of ignored rows
    g = torch.Generator().manual_seed(5)
    k = 20                                                                   # sample 1: all 20 rows real
    lab[1, :, 0:2] = torch.rand(k, 2, generator=g) * (res - 1)
    lab[1, :, 2:4] = torch.rand(k, 2, generator=g) * (res / 4) + 1
    lab[1, :, 4] = torch.rand(k, generator=g) * 6.2831853 - 3.1415926
    lab[1, :, 5] = torch.rand(k, generator=g) * 8
    lab[1, :, 6] = (torch.rand(k, generator=g) < 0.5).float()
    return lab                                                               # sample 2: no boxes at all (avg_factor -> max(1, 0))


def head_preds(B=3, res=16, nbins=12, seed=3):
    g = torch.Generator().manual_seed(seed)
    r = lambda c: torch.randn(B, c, res, res, generator=g)
    return [r(1).sigmoid(), r(2) * 3, r(2), r(nbins), r(1), r(1) * 2, r(2)]


MODEL_GRAD_KEYS = ("head.heatmap_head.0.weight", "head.heatmap_head.2.bias", "head.wh_head.2.weight", "head.offset_head.0.bias",
                   "head.yaw_class_head.2.weight", "head.yaw_res_head.2.weight", "head.velocity_head.2.bias", "head.brake_head.2.weight",
                   "pred_bev.0.weight", "pred_bev.2.bias", "join.0.weight", "join.4.bias", "decoder.weight_ih", "decoder.weight_hh",
                   "decoder.bias_ih", "decoder.bias_hh", "output.weight", "output.bias", "seg_decoder.deconv1.2.weight", "depth_decoder.deconv3.2.weight",
                   "_model.transformer1.pos_emb", "_model.transformer4.blocks.1.attn.proj.weight", "_model.change_channel_conv_image.weight",
                   "_model.image_encoder.features.s1.b1.conv1.conv.weight", "_model.lidar_encoder._model.s2.b1.conv2.conv.weight", "_model.up_conv3.weight")
LOSS_WEIGHTS = [1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.3, 0.4]      # non-zero everywhere so every head gets a gradient


def call_model(m, b):
    return m(b['rgb'], b['lidar'], ego_waypoint=b['ego_waypoint'], target_point=b['target_point'], target_point_image=b['target_point_image'],
             ego_vel=b['ego_vel'].reshape(-1, 1), bev=b['bev'], label=b['label'], depth=b['depth'], semantic=b['semantic'])


def model_golden(ref_model):
    """The reference's OWN model.py (imported unmodified over oracle/{mm,timm,scatter}_shim): LidarCenterNet.forward losses + gradients,
    LidarCenterNetHead.get_targets / loss / decode_heatmap, forward_gru, forward_ego + get_bbox_local_metric, control_pid."""
    out = {}
    cfg = model_config()
    torch.manual_seed(0)
    m = ref_model.LidarCenterNet(cfg, 'cpu', 'transFuser', 'regnety_tiny', 'regnety_tiny', use_velocity=False)
    seeded_fill(m, 4321)
    m.train()
    b = model_batch()
    losses = call_model(m, b)
    assert set(losses) == set(cfg.detailed_losses) and len(losses) == 11
    sum(w * losses[k] for w, k in zip(LOSS_WEIGHTS, cfg.detailed_losses)).backward()
    out["model_losses"] = np.array([float(losses[k]) for k in cfg.detailed_losses], np.float64)
    named = dict(m.named_parameters())
    for k in MODEL_GRAD_KEYS:
        out["model_grad_" + k] = named[k].grad.numpy()
    names = sorted(n for n, p in named.items() if p.grad is not None)
    out["model_grad_norms"] = np.array([named[n].grad.double().norm().item() for n in names])
    out["model_grad_names"] = np.array(names)
    # --- get_targets (model.py:285-374) on the crafted labels, 16x16 map of a 64x64 frame
    lab = target_labels()
    t, af = m.head.get_targets([lab], [torch.zeros_like(lab[:, :, 0])], [lab.sum(-1) == 0.], (3, 1, 16, 16))
    for k, v in t.items():
        out["tgt_" + k] = v.numpy()
    out["tgt_avg_factor"] = np.array([int(af)])
    # --- loss (model.py:150-248) on random predictions
    preds = head_preds()
    l = m.head.loss(*[[p] for p in preds], [lab], gt_labels=[torch.zeros_like(lab[:, :, 0])], gt_bboxes_ignore=[lab.sum(-1) == 0.], img_metas=None)
    out["head_losses"] = np.array([float(l[k]) for k in ("loss_center_heatmap", "loss_wh", "loss_offset", "loss_yaw_class", "loss_yaw_res",
                                                         "loss_velocity", "loss_brake")], np.float64)
    # --- decode_heatmap / get_bboxes (model.py:376-497)
    cfg.top_k_center_keypoints = 20
    res = m.head.get_bboxes(*[[p] for p in preds])
    out["decode_boxes"] = torch.stack([r[0] for r in res]).numpy()
    out["decode_labels"] = torch.stack([r[1] for r in res]).numpy()
    # --- forward_gru (model.py:611-646)
    g = torch.Generator().manual_seed(8)
    z, tp = torch.randn(3, 512, generator=g), torch.randn(3, 2, generator=g) * 10
    with torch.no_grad():
        out["gru_wp"] = m.forward_gru(z, tp)[0].numpy()
    # --- forward_ego (model.py:685-731) incl. get_bbox_local_metric (:810-843); eval mode, batch of one
    m.eval()
    cfg.bb_confidence_threshold = 0.0
    with torch.no_grad():
        wp, boxes = m.forward_ego(b['rgb'][:1], b['lidar'][:1], b['target_point'][:1], b['target_point_image'][:1], b['ego_vel'][:1].reshape(-1, 1))
    out["ego_wp"] = wp.numpy()
    out["ego_boxes"] = np.stack([bb[0] for bb in boxes])
    out["ego_brake_conf"] = np.array([[bb[1], bb[2]] for bb in boxes])
    # --- control_pid (model.py:648-683): three successive calls (the PID controllers keep state)
    pid = []
    for i in range(3):
        s, t_, br = m.control_pid(wp + 0.5 * i, torch.tensor([1.0 + i]), bool(i == 2))
        pid.append([float(s), float(t_), float(br)])
    out["ego_pid"] = np.array(pid)
    return out


def main():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "scatter_shim"))
    sys.path.insert(0, os.path.join(ROOT, "oracle", "timm_shim"))
    sys.path.insert(0, os.path.join(ROOT, "oracle", "mm_shim"))
    sys.path.insert(0, "/root/reference/team_code_transfuser")
    import timm
    from oracle import regnet as oreg, hist
    timm.register("regnety_tiny", lambda: oreg.RegNet(TINY["widths"], TINY["depths"], TINY["group_w"], TINY["se_ratio"]))
    import transfuser as ref   # the reference module, unmodified
    cfg = golden_config()
    out = {}
    for use_vel in (False, True):
        torch.manual_seed(0)
        m = ref.TransfuserBackbone(cfg, "regnety_tiny", "regnety_tiny", use_velocity=use_vel)
        seeded_fill(m)
        for mode in ("train", "eval"):
            getattr(m, mode)()
            img, lid, vel = golden_inputs()
            with torch.no_grad():
                feats, grid, fused = m(img, lid, vel)
            tag = "vel%d_%s" % (use_vel, mode)
            for i, f in enumerate(feats):
                out["%s_p%d" % (tag, i + 2)] = f.numpy()
            out[tag + "_grid"] = grid.numpy()
            out[tag + "_fused"] = fused.numpy()
            seeded_fill(m)   # train-mode forward updated the running stats: reset for the next case
    seg, dep = ref.SegDecoder(cfg, 512), ref.DepthDecoder(cfg, 512)
    seeded_fill(seg, 5); seeded_fill(dep, 6)
    x = torch.randn(2, 512, 2, 4, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        out["seg"] = seg(x).numpy()
        out["depth"] = dep(x).numpy()
        out["normalize_imagenet"] = ref.normalize_imagenet(golden_inputs()[0]).numpy()[:, :, :4, :8]
    np.savez_compressed(os.path.join(HERE, "transfuser_backbone_tiny.npz"), **out)
    import geometric_fusion as ref_geo   # the reference module, unmodified
    np.savez_compressed(os.path.join(HERE, "geometric_fusion_tiny.npz"), **geo_golden(ref_geo))
    import point_pillar as ref_pp        # the reference module, unmodified (torch_scatter = oracle/scatter_shim)
    np.savez_compressed(os.path.join(HERE, "point_pillar.npz"), **pillar_golden(ref_pp))
    import model as ref_model            # the reference module, unmodified (cv2 / torchvision / mmcv / mmdet = oracle/mm_shim)
    np.savez_compressed(os.path.join(HERE, "lidar_centernet_tiny.npz"), **model_golden(ref_model))
    np.savez_compressed(os.path.join(HERE, "dataprep.npz"), **dataprep_golden())
    np.savez_compressed(os.path.join(HERE, "correspondences.npz"), **correspondences_golden())
    # H1: numpy.histogramdd (the reference's algorithm, data.py:446-470) on a seeded cloud with edge cases -> sparse golden
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(-20, 20, 20000), rng.uniform(-36, 4, 20000), rng.uniform(-4, 1, 20000)], 1).astype(np.float32)
    pts[:40, 0] = 16.0; pts[40:80, 1] = 0.0; pts[80:120, 0] = -16.0; pts[120:160, 1] = -32.0; pts[160:200, 2] = -2.3
    pts[200:700, :2] = np.round(pts[200:700, :2] * 8) / 8
    h = hist.lidar_to_histogram_features(pts)
    idx = np.nonzero(h)
    np.savez_compressed(os.path.join(HERE, "lidar_hist.npz"), points=pts, idx=np.stack(idx).astype(np.int16), val=(h[idx] * 5).round().astype(np.uint8))
    print("wrote", sorted(os.listdir(HERE)))


# ---------------------------------------------------------------- data.py per-sample preparation
def reference_data_functions():
    """Functions of the reference's data.py / utils.py EXECUTED FROM THEIR OWN SOURCE (ast-extracted, so the module-level imports of cv2 /
    ujson / skimage - absent here - are never run).  ``rotate`` (skimage) is only provided as the identity: angle-0 cases."""
    import ast
    from copy import deepcopy
    ns = {"np": np, "deepcopy": deepcopy, "rotate": lambda img, deg: img}
    want = {"utils.py": None, "data.py": {"get_depth", "align", "lidar_to_histogram_features", "decode_pil_to_npy", "crop_image_cv2", "crop_seg", "load_crop_bev_npy",
                                          "get_bbox_label", "parse_labels", "get_waypoints", "transform_waypoints"}}
    for fn, names in want.items():
        src = open(os.path.join("/root/reference/team_code_transfuser", fn)).read()
        for node in ast.parse(src).body:
            if isinstance(node, ast.FunctionDef) and (names is None or node.name in names):
                exec(compile(ast.Module([node], []), fn, "exec"), ns)
    return ns


def dataprep_raw(B=2, seed=5, Hs=48, Ws=160, S=500, N=3000):
    """Raw decoded arrays of B frames (what CARLA_Data.__getitem__ returns, small images) + label / measurement dicts."""
    rng = np.random.default_rng(seed)
    raw = dict(rgb_u8=rng.integers(0, 256, (B, Hs, Ws, 3), dtype=np.uint8), depth_u8=rng.integers(0, 256, (B, Hs, Ws, 3), dtype=np.uint8),
               sem_u8=rng.integers(0, 28, (B, Hs, Ws, 1), dtype=np.uint8), bev_u8=rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8))
    raw["depth_u8"][:, :, : Ws // 2, 0] = 0          # small depths (< 50 m) in half of the image, so the clip is exercised on both sides
    raw["depth_u8"][:, :, : Ws // 4, 1] = rng.integers(0, 13, (B, Hs, Ws // 4), dtype=np.uint8)
    pts = np.stack([rng.uniform(-25, 25, (B, N)), rng.uniform(-40, 8, (B, N)), rng.uniform(-4, 1, (B, N)), rng.uniform(0, 1, (B, N))], -1).astype(np.float32)
    pts[0, :50, :2] = np.round(pts[0, :50, :2] * 8) / 8
    raw["lidar_raw"] = pts
    raw["num_points"] = np.array([N, N - 700], np.int32)[:B]
    def pose(th, x, y):
        m = np.eye(4); m[0, 0] = m[1, 1] = np.cos(th); m[0, 1] = -np.sin(th); m[1, 0] = np.sin(th); m[0, 3] = x; m[1, 3] = y
        return m
    raw["ego_matrix"] = np.stack([pose(0.3 + 0.1 * b, 10.0 + b, -4.0) for b in range(B)])
    labels, meas = [], []
    for b in range(B):
        fr = []
        for t in range(5):
            objs = [dict(id=7, ego_matrix=pose(0.3 + 0.1 * b + 0.02 * t, 10.0 + b + 1.5 * t, -4.0 + 0.2 * t).tolist(), extent=[0.7, 2.4, 1.0], position=[0.0, 0.0, 0.0],
                         yaw=0.0, speed=3.0, brake=0.0, num_points=50, distance=0.0)]
            for j in range(6):
                objs.append(dict(id=100 + j, ego_matrix=pose(0.1 * j, 3.0 * j, 2.0).tolist(), extent=[0.8, 2.0 + 0.1 * j, 0.9], position=[float(rng.uniform(-20, 20)), float(rng.uniform(-2, 36)), 0.0],
                                 yaw=float(rng.uniform(-3, 3)), speed=float(rng.uniform(0, 8)), brake=float(j % 2), num_points=int(j), distance=5.0))
            fr.append(objs)
        labels.append(fr)
        meas.append(dict(theta=0.3 + 0.1 * b, x=10.0 + b, y=-4.0, x_command=30.0, y_command=5.0 - b, speed=3.5 + b, ego_matrix=raw["ego_matrix"][b].tolist()))
    return raw, labels, meas


def dataprep_golden():
    """Outputs of the reference's own functions (no augmentation: degree 0) on dataprep_raw()."""
    f = reference_data_functions()
    raw, labels, meas = dataprep_raw()
    B = raw["rgb_u8"].shape[0]
    crop = (16, 64)
    out = dict(rgb=[], depth=[], semantic=[], bev=[], lidar=[], label=[], ego_waypoint=[], target_point=[])
    conv = np.uint8([0, 0, 0, 0, 4, 0, 5, 2, 6, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 3, 0, 0, 5])
    for b in range(B):
        out["rgb"].append(f["crop_image_cv2"](raw["rgb_u8"][b], crop=crop, crop_shift=0))
        out["depth"].append(f["get_depth"](f["crop_image_cv2"](raw["depth_u8"][b], crop=crop, crop_shift=0)))
        out["semantic"].append(conv[f["crop_seg"](raw["sem_u8"][b][..., 0], crop=crop, crop_shift=0)])
        enc = np.moveaxis(raw["bev_u8"][b], -1, 0)
        out["bev"].append(f["load_crop_bev_npy"](f["decode_pil_to_npy"](enc).astype(np.uint8), 0))
        n = int(raw["num_points"][b])
        lid = f["align"](raw["lidar_raw"][b, :n].copy(), meas[b], meas[b], degree=0)
        out["lidar"].append(f["lidar_to_histogram_features"](lid))
        boxes = f["parse_labels"](labels[b][0], rad=-0.0)
        wps = f["transform_waypoints"](f["get_waypoints"](labels[b], 5))
        ego = np.array([m[:2, 3] for m, flag in wps[7][1:]])
        lab = np.zeros((20, 7), np.float32)
        arr = np.array(list(boxes.values()))
        if arr.shape[0]:
            lab[:arr.shape[0]] = arr
        out["label"].append(lab); out["ego_waypoint"].append(ego)
        th = meas[b]["theta"]
        R = np.array([[np.cos(np.pi / 2 + th), -np.sin(np.pi / 2 + th)], [np.sin(np.pi / 2 + th), np.cos(np.pi / 2 + th)]])
        out["target_point"].append(R.T.dot(np.array([meas[b]["x_command"] - meas[b]["x"], meas[b]["y_command"] - meas[b]["y"]])))
    res = {"dp_" + k: np.stack(v) for k, v in out.items()}
    idx = np.nonzero(res["dp_lidar"])
    res["dp_lidar_idx"] = np.stack(idx).astype(np.int16); res["dp_lidar_val"] = (res.pop("dp_lidar")[idx] * 5).round().astype(np.uint8)
    res["dp_depth"] = res["dp_depth"].astype(np.float64)
    return res


# ---------------------------------------------------------------- data.py:632-842 LiDAR <-> camera correspondences (geometric fusion)
def correspondence_clouds(seed=31):
    """Raw CARLA-frame clouds (x left, y forward, z up), float32: a sparse one (most cells hold <= 5 points: the deterministic branch of
    correspondences_at_one_scale), a dense one (random.sample branch) and edge points: on the |x| = 16 / y = 32 / y = 0 borders, on cell
    borders, straight ahead, at the +-30 degree seams of the three cameras and behind the rotated cameras' image planes."""
    rng = np.random.default_rng(seed)
    def cloud(n):
        return np.stack([rng.uniform(-20, 20, n), rng.uniform(-5, 40, n), rng.uniform(-3, 3, n)], 1).astype(np.float32)
    edge = np.float32([[16, 5, 0], [-16, 5, 0], [15.999999, 5, 0], [0, 32, 0], [0, 31.999998, 0], [0, 0, 0], [0, 1e-3, 0], [4, 4, 0.5], [-4, 4, 0.5], [8, 8, -1], [0, 10, 0],
                       [5.7735027, 10, 0], [-5.7735027, 10, 0], [10, 17.320508, 0], [-10, 17.320508, 0], [15, 1, 0], [-15, 1, 0], [15, 0.5, 2], [-15, 0.5, -2], [12, 20.784609, 1],
                       [2.0, 16.0, 0.1], [-2.0, 16.0, 0.1], [1.9999999, 8.0, 2.0], [12.0, 4.0, 0.25], [-12.0, 4.0, -0.25], [3, 31, 2.9], [3, 31, -2.9]])
    return {"sparse": np.concatenate([cloud(260), edge]), "dense": np.concatenate([edge, cloud(6000)])}


def correspondences_golden():
    """lidar_bev_cam_correspondences (data.py:675-842, with correspondences_at_one_scale :632-673) executed from the reference's source."""
    import ast
    import random
    from copy import deepcopy
    ns = {"np": np, "deepcopy": deepcopy, "random": random}
    src = open("/root/reference/team_code_transfuser/data.py").read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ("correspondences_at_one_scale", "lidar_bev_cam_correspondences"):
            exec(compile(ast.Module([node], []), "data.py", "exec"), ns)
    out = {}
    for name, c in correspondence_clouds().items():
        random.seed(123)
        bev, cam = ns["lidar_bev_cam_correspondences"](c.copy())
        out["corr_%s_bev" % name], out["corr_%s_cam" % name] = bev.astype(np.int16), cam.astype(np.int16)
        assert np.array_equal(out["corr_%s_bev" % name], bev) and bev.shape == (8, 8, 5, 2) and cam.shape == (22, 5, 5, 2)
    return out


if __name__ == "__main__":
    main()
