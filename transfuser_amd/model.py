"""``LidarCenterNet`` - the full trainable TransFuser model - as a drop-in for
team_code_transfuser/model.py:538-805 (constructor :545, ``forward`` :733-805 returning the dict of
11 losses, ``forward_gru`` :611) and ``LidarCenterNetHead`` (:34-514: heads :93-99,127-147,
targets :285-374, losses :150-248 with mmdet 2.25 semantics).

Same parameter names/shapes (reference checkpoints load with strict=True), same call signature,
same return value; everything between the input tensors and the loss scalars runs in the HIP
kernels of libtransfuser_hip.so.  ``forward_ego`` / ``control_pid`` are built on the same kernels plus
the fused decode_heatmap kernel; not built: the visualisation branch.
"""
from collections import deque

import numpy as np
import os
import torch
from torch import nn

from . import functions as F_
from . import ops
from .geometric_fusion import GeometricFusionBackbone
from .point_pillar import PointPillarNet
from .transfuser import DepthDecoder, LateFusionBackbone, SegDecoder, TransfuserBackbone, latentTFBackbone, nchw

HEAD_ORDER = ("heatmap_head", "wh_head", "offset_head", "yaw_class_head", "yaw_res_head", "velocity_head", "brake_head")
_FORK_DECODERS = int(os.environ.get("TF_FORK_DECODERS", "2"))      # round 6: 0 = decoders on the main stream, 1 = forked after the backbone, 2 (default) = forked as soon as the image grid exists

LOSS_KEYS = ("loss_center_heatmap", "loss_wh", "loss_offset", "loss_yaw_class", "loss_yaw_res", "loss_velocity", "loss_brake")


class PIDController(object):
    """model.py:517-535: P + I (mean of the last n errors) + D (last difference)."""

    def __init__(self, K_P=1.0, K_I=0.0, K_D=0.0, n=20):
        self._K_P, self._K_I, self._K_D = K_P, K_I, K_D
        self._window = deque([0 for _ in range(n)], maxlen=n)

    def step(self, error):
        self._window.append(error)
        if len(self._window) >= 2:
            integral, derivative = np.mean(self._window), self._window[-1] - self._window[-2]
        else:
            integral = derivative = 0.0
        return self._K_P * error + self._K_I * integral + self._K_D * derivative


def get_lidar_to_bevimage_transform():
    """utils.py:29-37: LiDAR metres -> BEV pixels (rotate, shift by (16, 32) m, 8 px/m)."""
    T = np.array([[0, -1, 16], [-1, 0, 32], [0, 0, 1]], dtype=np.float32)
    T[:2, :] *= 8
    return T


class LidarCenterNetHead(nn.Module):
    """Parameter holder + loss front-end of the CenterNet head (model.py:34-248)."""

    def __init__(self, in_channel, feat_channel, num_classes, train_cfg=None, **unused):
        super().__init__()
        assert num_classes == 1, "the TransFuser head predicts a single class (model.py:589)"
        self.num_classes = num_classes
        self.num_dir_bins = train_cfg.num_dir_bins
        self.train_cfg = train_cfg
        mk = lambda oc: nn.Sequential(nn.Conv2d(in_channel, feat_channel, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                                      nn.Conv2d(feat_channel, oc, kernel_size=1))
        self.heatmap_head = mk(num_classes)
        self.wh_head = mk(2)
        self.offset_head = mk(2)
        self.yaw_class_head = mk(self.num_dir_bins)
        self.yaw_res_head = mk(1)
        self.velocity_head = mk(1)
        self.brake_head = mk(2)

    @property
    def pred_channels(self):
        return 9 + self.num_dir_bins

    def heads(self):
        return [getattr(self, n) for n in HEAD_ORDER]


# order of the eight 3x3 convolutions on p2 inside the parameter arena (train.ParamArena groups them): the heads that can carry a zero loss
# weight (velocity, brake - quirk Q6) last, so that the live ones are a PREFIX of the merged 64 -> 512 convolution
MERGED_ORDER = ("heatmap_head", "wh_head", "offset_head", "yaw_class_head", "yaw_res_head", "pred_bev", "velocity_head", "brake_head")


def merged_head_convs(model):
    """(W (8 Ch, C, 3, 3) channels_last, b (8 Ch,), dW, db) - views over the eight first convolutions of the CenterNet heads + pred_bev - when
    the parameter arena made them adjacent in MERGED_ORDER (one 64 -> 512 convolution instead of eight 64 -> 64 ones); None
    otherwise (plain torch parameters: the per-head path)."""
    seqs = [getattr(model.head, n) if n != "pred_bev" else model.pred_bev for n in MERGED_ORDER]
    ws, bs = [sq[0].weight for sq in seqs], [sq[0].bias for sq in seqs]
    w0 = ws[0]
    key = (w0.data_ptr(), w0.grad.data_ptr() if w0.grad is not None else 0, bs[0].data_ptr())
    cached = getattr(model, "_merged_heads", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    out = None
    if all(p.grad is not None for p in ws + bs) and all(w.shape == w0.shape and w.permute(0, 2, 3, 1).is_contiguous() for w in ws):
        def adjacent(ts):
            return all(ts[i + 1].data_ptr() == t.data_ptr() + t.numel() * 4 for i, t in enumerate(ts[:-1])) and \
                ts[0].untyped_storage().data_ptr() == ts[-1].untyped_storage().data_ptr()
        if adjacent(ws) and adjacent(bs) and adjacent([p.grad for p in ws]) and adjacent([p.grad for p in bs]):
            Ch, C = w0.shape[0], w0.shape[1]
            n = len(ws)
            wview = lambda t: torch.as_strided(t, (n * Ch, C, 3, 3), (9 * C, 1, 3 * C, C), t.storage_offset())
            bview = lambda t: torch.as_strided(t, (n * Ch,), (1,), t.storage_offset())
            out = (wview(w0.data), bview(bs[0].data), wview(w0.grad), bview(bs[0].grad))
    model._merged_heads = (key, out)
    return out


@F_.routes_param_grads
class HeadsFn(torch.autograd.Function):
    """The 7 CenterNet heads + pred_bev on p2 in one autograd node (model.py:127-147,581-585,759):
    8 x [conv3x3 64->64 + ReLU + conv1x1]; head outputs are packed as (B,h,w,9+bins) logits
    [hm, wh(2), off(2), yaw_cls(bins), yaw_res, vel, brake(2)] for the fused loss kernel.
    Inside train.Engine the eight 3x3 convolutions are ONE 64 -> 512 convolution over the arena-adjacent weights (merged_head_convs), its
    backward one weight-gradient and one input-gradient launch over the live heads' channels (a prefix of the 512)."""

    @staticmethod
    def forward(ctx, p2, model, *params):
        B, H, W, C = p2.shape
        seqs = model.head.heads() + [model.pred_bev]
        names = list(HEAD_ORDER) + ["pred_bev"]
        P = model.head.pred_channels
        pred = torch.empty(B, H, W, P, dtype=torch.float32, device=p2.device)
        bev = torch.empty(B, H, W, seqs[-1][2].weight.shape[0], dtype=torch.float32, device=p2.device)
        merged = merged_head_convs(model) if (F_._INPLACE and ops._AB_MERGE_HEADS) else None
        hid_all = None
        if merged is not None:
            hid_all = ops.conv_fwd(p2, merged[0], merged[1], 1, 1, 1, relu=True)          # (B, H, W, 8 Ch)
            Ch = seqs[0][0].weight.shape[0]
            hid2 = hid_all.view(-1, hid_all.shape[-1])
        hids, off = [], 0
        for i, sq in enumerate(seqs):
            if merged is not None:
                j = MERGED_ORDER.index(names[i])
                h2 = hid2[:, j * Ch:(j + 1) * Ch]                                          # row stride 8 Ch
            else:
                hid = ops.conv_fwd(p2, sq[0].weight, sq[0].bias, 1, 1, 1, relu=True)
                h2 = hid.view(-1, hid.shape[-1])
            k = sq[2].weight.shape[0]
            dst = bev.view(-1, k) if i == len(seqs) - 1 else pred.view(-1, P)[:, off:off + k]
            ops.linear_fwd(h2, F_.w2d(sq[2].weight), sq[2].bias, out=dst)
            if i < len(seqs) - 1:
                off += k
            hids.append(h2)
        ctx.saved = (p2, model, hids, merged)
        return pred, bev

    @staticmethod
    def backward(ctx, dpred, dbev):
        p2, model, hids, merged = ctx.saved
        seqs = model.head.heads() + [model.pred_bev]
        names = list(HEAD_ORDER) + ["pred_bev"]
        B, H, W, C = p2.shape
        P = model.head.pred_channels
        dpred, dbev = dpred.contiguous(), dbev.contiguous()
        db_pred = ops.colsum(dpred.view(-1, P), 1, B * H * W, P)
        dp2 = torch.empty_like(p2)
        off = 0
        dead = getattr(model, "_dead_heads", ())   # heads whose loss weight is 0 (config.detailed_losses_weights: velocity, brake - quirk Q6):
        first = True                               # their incoming gradient is exactly 0, so is everything their backward would add
        Ch = hids[0].shape[-1]
        dh_all = None
        if merged is not None:
            # live heads = a prefix of MERGED_ORDER unless a head in front of the tail is dead too (then every channel is differentiated)
            live = [n for n in MERGED_ORDER if n not in dead]
            if list(MERGED_ORDER[:len(live)]) != live:
                # the live heads are not a channel prefix (e.g. wp_only: seven dead heads in front of pred_bev): differentiating all 512 merged channels would
                # run the weight / input gradient over mostly-zero channels and write zeros into the dead heads' gradients - the per-head backward below
                # skips them instead (the merged FORWARD is kept: hids[i] are its strided views)
                merged = None
            else:
                dh_all = torch.empty(B * H * W, len(live) * Ch, dtype=torch.float32, device=p2.device)
        for i, sq in enumerate(seqs):
            k = sq[2].weight.shape[0]
            last = i == len(seqs) - 1
            if not last and HEAD_ORDER[i] in dead:
                off += k
                continue
            g2 = dbev.view(-1, k) if last else dpred.view(-1, P)[:, off:off + k]
            h2 = hids[i]
            ops.linear_wgrad(g2, h2, F_.w2d(F_.gbuf(sq[2].weight)))
            if last:
                F_.bias_grad(g2, sq[2].bias)
            else:
                ops.axpby(F_.gbuf(sq[2].bias), db_pred[0, off:off + k], 1.0, 1.0, out=F_.gbuf(sq[2].bias))
                off += k
            if merged is not None:
                j = MERGED_ORDER.index(names[i])
                ops.linear_dgrad(g2, F_.w2d(sq[2].weight), out=dh_all[:, j * Ch:(j + 1) * Ch], mask=h2)      # ReLU backward in the epilogue
                continue
            dh = ops.linear_dgrad(g2, F_.w2d(sq[2].weight), mask=h2).view(B, H, W, Ch)     # ReLU backward in the epilogue
            F_.bias_grad(dh.view(-1, Ch), sq[0].bias)
            ops.conv_wgrad(dh, p2, F_.gbuf(sq[0].weight), 1, 1, 1)
            ops.conv_dgrad(dh, sq[0].weight, p2.shape, 1, 1, 1, out=dp2, accumulate=not first)
            first = False
        if merged is not None:
            n = dh_all.shape[1]
            W_all, _, dW_all, db_all = merged
            ops.colsum(dh_all, 1, B * H * W, n, 1.0, out=db_all[:n].view(1, -1), accumulate=True)
            dh4 = dh_all.view(B, H, W, n)
            ops.conv_wgrad(dh4, p2, dW_all[:n], 1, 1, 1)
            ops.conv_dgrad(dh4, W_all[:n], p2.shape, 1, 1, 1, out=dp2)
        ctx.saved = None
        return (dp2, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class LidarCenterNet(nn.Module):
    def __init__(self, config, device, backbone, image_architecture='resnet34', lidar_architecture='resnet18', use_velocity=True):
        super().__init__()
        self.device = device
        self.config = config
        self.pred_len = config.pred_len
        self.use_target_point_image = config.use_target_point_image
        self.gru_concat_target_point = config.gru_concat_target_point
        self.use_point_pillars = config.use_point_pillars
        if self.use_point_pillars:   # model.py:554-559
            self.point_pillar_net = PointPillarNet(config.num_input, config.num_features, min_x=config.min_x, max_x=config.max_x,
                                                   min_y=config.min_y, max_y=config.max_y, pixels_per_meter=int(config.pixels_per_meter))
            assert backbone != 'latentTF', "latentTF overwrites the LiDAR input with a positional grid (latentTF.py:132-137)"
        self.backbone = backbone
        if backbone == 'transFuser':
            self._model = TransfuserBackbone(config, image_architecture, lidar_architecture, use_velocity=use_velocity)
        elif backbone == 'latentTF':
            self._model = latentTFBackbone(config, image_architecture, lidar_architecture, use_velocity=use_velocity)
        elif backbone == 'geometric_fusion':
            self._model = GeometricFusionBackbone(config, image_architecture, lidar_architecture, use_velocity=use_velocity)
        elif backbone == 'late_fusion':
            self._model = LateFusionBackbone(config, image_architecture, lidar_architecture, use_velocity=use_velocity)
        else:
            raise TypeError("The chosen vision backbone does not exist. The options are: transFuser, late_fusion, geometric_fusion, latentTF")   # model.py:573
        if config.multitask:
            self.seg_decoder = SegDecoder(config, config.perception_output_features)
            self.depth_decoder = DepthDecoder(config, config.perception_output_features)
        channel = config.channel
        self.pred_bev = nn.Sequential(nn.Conv2d(channel, channel, kernel_size=(3, 3), stride=1, padding=(1, 1), bias=True),
                                      nn.ReLU(inplace=True),
                                      nn.Conv2d(channel, 3, kernel_size=(1, 1), stride=1, padding=0, bias=True))
        self.head = LidarCenterNetHead(channel, channel, 1, train_cfg=config)
        self.i = 0
        self.join = nn.Sequential(nn.Linear(512, 256), nn.ReLU(inplace=True), nn.Linear(256, 128), nn.ReLU(inplace=True),
                                  nn.Linear(128, 64), nn.ReLU(inplace=True))
        self.decoder = nn.GRUCell(input_size=4 if self.gru_concat_target_point else 2, hidden_size=config.gru_hidden_size)
        self.output = nn.Linear(config.gru_hidden_size, 3)
        self.register_buffer("bev_class_weight", torch.tensor([1., 1., 3.]), persistent=False)  # model.py:762
        for m in self.modules():  # 3x3 conv weights live channels_last (= the kernels' (Cout,kh,kw,Cin) layout)
            if isinstance(m, nn.Conv2d) and m.kernel_size != (1, 1):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        self.to(device)

    # ------------------------------------------------------------------ waypoints
    def forward_gru(self, z, target_point):
        pred_wp = F_.WaypointFn.apply(z, target_point, self, *self.join.parameters(), *self.decoder.parameters(), *self.output.parameters())
        return pred_wp, None, None, None, None

    # ------------------------------------------------------------------ training forward
    def forward(self, rgb, lidar_bev, ego_waypoint, target_point, target_point_image, ego_vel, bev, label, depth, semantic,
                num_points=None, save_path=None, bev_points=None, cam_points=None):
        cfg = self.config
        if ops.uses_zero_arena():
            ops.zero_scratch_reset(rgb.device)   # one memset per step for every atomically accumulated statistic (opt-in experiment)
        extra = target_point_image if self.use_target_point_image else None
        kw = dict(lidar_extra=extra)
        if self.use_point_pillars:   # model.py:736-742: lidar_bev is the raw cloud (B, N, 4); pillars -> rot90 -> cat target point
            kw = dict(lidar_nhwc=self.point_pillar_net.forward_nhwc(lidar_bev, num_points, extra))
        if self.backbone == 'geometric_fusion':   # model.py:749-750
            features, grid, fused = self._model.forward_nhwc(rgb, lidar_bev, ego_vel, bev_points, cam_points, **kw)
        else:
            features, grid, fused = self._model.forward_nhwc(rgb, lidar_bev, ego_vel, **kw)
        # Round 6: the segmentation and the depth decoder (+ their losses) as two more parallel branches of the step (hipGraph: two more graph branches) - they only
        # read the image feature grid and join at the weighted loss sum; their first layers (8 x 22 ... 64 x 176 maps) are latency-sized and fill the gaps of the FPN's
        # and the heads' kernels, and autograd replays their backward on the same streams.  Forked BEFORE the heads are enqueued, like the LiDAR trunk in _run.
        # Same-lease A/B (three rounds): 44.77 -> 43.74 ms/step in fp32, bf16 32.70 -> 31.83, latentTF fp16 B = 16 46.39 -> 45.67, geometric fusion 27.39 -> 26.48.
        forked = None
        if cfg.multitask and _FORK_DECODERS and grid.is_cuda:
            main = torch.cuda.current_stream(grid.device)
            if getattr(self, "_dec_streams", None) is None or self._dec_streams[0].device != grid.device:
                self._dec_streams = (torch.cuda.Stream(grid.device), torch.cuda.Stream(grid.device))
            forked = []
            for st, dec, tgt, kind in ((self._dec_streams[0], self.seg_decoder, semantic, "sem"), (self._dec_streams[1], self.depth_decoder, depth, "dep")):
                ev = getattr(self._model, "_grid_ready", None) if _FORK_DECODERS > 1 else None
                if ev is not None:
                    st.wait_event(ev)          # (2: start as soon as the grid exists - beside the LiDAR reducer and the FPN - not after everything enqueued on main)
                else:
                    st.wait_stream(main)
                with torch.cuda.stream(st):
                    logits = dec.forward_nhwc(grid)
                    l = F_.CrossEntropyFn.apply(logits, tgt.contiguous(), None) if kind == "sem" else F_.L1Fn.apply(logits.squeeze(-1), tgt.contiguous(), True)
                grid.record_stream(st)
                forked.append((st, l))
        pred_wp, _, _, _, _ = self.forward_gru(fused, target_point)
        p2 = features[0]
        pred, bev_logits = HeadsFn.apply(p2, self, *self.head.parameters(), *self.pred_bev.parameters())
        bev_up = F_.UpsampleFn.apply(bev_logits, cfg.bev_resolution_height, cfg.bev_resolution_width, True)
        loss = {
            "loss_wp": F_.L1Fn.apply(pred_wp, ego_waypoint.contiguous(), False),
            "loss_bev": F_.CrossEntropyFn.apply(bev_up, bev.contiguous(), self.bev_class_weight),
        }
        det = F_.CenterNetLossFn.apply(pred, label.contiguous(), self.head.num_dir_bins, p2.shape[2] / float(cfg.lidar_resolution_width),
                                       p2.shape[1] / float(cfg.lidar_resolution_height))
        for i, k in enumerate(LOSS_KEYS):
            loss[k] = det[i]
        if cfg.multitask:
            if forked is not None:
                main = torch.cuda.current_stream(grid.device)
                for st, l in forked:
                    main.wait_stream(st)
                    l.record_stream(main)
                l_sem, l_dep = forked[0][1], forked[1][1]
            else:
                seg_logits = self.seg_decoder.forward_nhwc(grid)
                depth_logits = self.depth_decoder.forward_nhwc(grid)
                l_sem = F_.CrossEntropyFn.apply(seg_logits, semantic.contiguous(), None)
                l_dep = F_.L1Fn.apply(depth_logits.squeeze(-1), depth.contiguous(), True)
            loss["loss_depth"] = l_dep * cfg.ls_depth if cfg.ls_depth != 1.0 else l_dep
            loss["loss_semantic"] = l_sem * cfg.ls_seg if cfg.ls_seg != 1.0 else l_sem
        else:
            loss["loss_depth"] = torch.zeros_like(loss["loss_wp"])
            loss["loss_semantic"] = torch.zeros_like(loss["loss_wp"])
        self.i += 1
        self._last = dict(pred_wp=pred_wp, pred=pred, bev_up=bev_up, features=features, grid=grid, fused=fused)
        return loss

    # ------------------------------------------------------------------ inference
    def get_bbox_local_metric(self, bbox):
        """model.py:810-843: box in BEV pixels -> 4 corners + centre + velocity arrow in metres (x front, y right, ego at the origin)."""
        x, y, w, h, yaw, speed, brake, confidence = bbox
        cfg = self.config
        w = w / cfg.bounding_box_divisor / cfg.pixels_per_meter
        h = h / cfg.bounding_box_divisor / cfg.pixels_per_meter
        center = np.linalg.inv(get_lidar_to_bevimage_transform()) @ np.array([x, y, 1.0]) + np.array(cfg.lidar_pos)
        center[1] = -center[1]
        box = np.array([[-h, -w, 1], [-h, w, 1], [h, w, 1], [h, -w, 1], [0, 0, 1], [0, h * speed * 0.5, 1]])
        R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
        shift = np.array([center[0], center[1], 0])
        for i in range(box.shape[0]):
            box[i] = R @ box[i] + shift
        return box, brake, confidence

    def control_pid(self, waypoints, velocity, is_stuck):
        """model.py:648-683: waypoints (1, 4, 2) + speed -> (steer, throttle, brake)."""
        assert waypoints.size(0) == 1
        cfg = self.config
        if not hasattr(self, "turn_controller"):    # model.py:607-608
            self.turn_controller = PIDController(K_P=cfg.turn_KP, K_I=cfg.turn_KI, K_D=cfg.turn_KD, n=cfg.turn_n)
            self.speed_controller = PIDController(K_P=cfg.speed_KP, K_I=cfg.speed_KI, K_D=cfg.speed_KD, n=cfg.speed_n)
        wp = waypoints[0].data.cpu().numpy()
        wp[:, 0] += cfg.lidar_pos[0]
        speed = velocity[0].data.cpu().numpy()
        desired_speed = np.linalg.norm(wp[0] - wp[1]) * 2.0
        if is_stuck:
            desired_speed = np.array(cfg.default_speed)
        brake = (desired_speed < cfg.brake_speed) or ((speed / desired_speed) > cfg.brake_ratio)
        delta = np.clip(desired_speed - speed, 0.0, cfg.clip_delta)
        throttle = np.clip(self.speed_controller.step(delta), 0.0, cfg.clip_throttle)
        throttle = throttle if not brake else 0.0
        aim = (wp[1] + wp[0]) / 2.0
        angle = np.degrees(np.arctan2(aim[1], aim[0])) / 90.0
        if speed < 0.01 or brake:
            angle = 0.0
        steer = np.clip(self.turn_controller.step(angle), -1.0, 1.0)
        return steer, throttle, brake

    @torch.no_grad()
    def forward_ego(self, rgb, lidar_bev, target_point, target_point_image, ego_vel, bev_points=None, cam_points=None, save_path=None,
                    expert_waypoints=None, stuck_detector=0, forced_move=False, num_points=None, rgb_back=None, debug=False):
        """model.py:685-731 (the visualisation branch is not built): (pred_wp (B, 4, 2), [(bbox (6, 3) ndarray, brake, confidence), ...]
        of sample 0).  Backbone, heads and the whole decode_heatmap (NMS + top-k + gather) run in HIP kernels; only the final
        per-box coordinate change is host numpy, as in the reference."""
        cfg = self.config
        extra = target_point_image if self.use_target_point_image else None
        kw = dict(lidar_extra=extra)
        if self.use_point_pillars:
            kw = dict(lidar_nhwc=self.point_pillar_net.forward_nhwc(lidar_bev, num_points, extra))
        if self.backbone == 'geometric_fusion':
            features, grid, fused = self._model.forward_nhwc(rgb, lidar_bev, ego_vel, bev_points, cam_points, **kw)
        else:
            features, grid, fused = self._model.forward_nhwc(rgb, lidar_bev, ego_vel, **kw)
        pred_wp, _, _, _, _ = self.forward_gru(fused, target_point)
        pred, _ = HeadsFn.apply(features[0], self, *self.head.parameters(), *self.pred_bev.parameters())
        boxes = ops.centernet_decode(pred, self.head.num_dir_bins, cfg.top_k_center_keypoints, cfg.center_net_max_pooling_kernel, 4.0)[0]
        boxes = boxes[boxes[:, -1] > cfg.bb_confidence_threshold]
        self.i += 1
        self._last_boxes = boxes
        return pred_wp, [self.get_bbox_local_metric(b) for b in boxes.cpu().numpy()]

    def load_reference_checkpoint(self, path_or_state_dict, strict=True):
        """Load a checkpoint written by the reference's train.py (possibly from a DistributedDataParallel model: keys prefixed with
        'module.', stripped exactly like submission_agent.py:94-96).  Parameter names / shapes are the reference's, conv weights are
        converted to this package's channels-last storage in place."""
        sd = torch.load(path_or_state_dict, map_location="cpu") if isinstance(path_or_state_dict, str) else path_or_state_dict
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        own = self.state_dict()
        with torch.no_grad():
            missing = [k for k in own if k not in sd]
            unexpected = [k for k in sd if k not in own]
            if strict and (missing or unexpected):
                raise RuntimeError("load_reference_checkpoint: missing %s, unexpected %s" % (missing[:5], unexpected[:5]))
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(v)      # copy_ keeps our memory format (channels_last conv weights) and the arena views
        return missing, unexpected
