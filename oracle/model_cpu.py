"""CPU restatement of LidarCenterNet (team_code_transfuser/model.py:538-805) and of the training
step of Engine.train (train.py:295-318).  TEST INFRASTRUCTURE / CPU baseline.

``backbone_module`` lets the authoring-container tests plug in the reference's own
TransfuserBackbone (imported unmodified) instead of oracle.transfuser_cpu.
"""
import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from . import transfuser_cpu
from .centernet import LidarCenterNetHead


class LidarCenterNet(nn.Module):
    def __init__(self, config, device='cpu', backbone='transFuser', image_architecture='regnety_032',
                 lidar_architecture='regnety_032', use_velocity=True, backbone_module=None, make_net=None):
        super().__init__()
        self.config = config
        self.pred_len = config.pred_len
        self.use_target_point_image = config.use_target_point_image
        self.gru_concat_target_point = config.gru_concat_target_point
        self.use_point_pillars = config.use_point_pillars
        if self.use_point_pillars:   # model.py:554-559
            from .pillars import PointPillarNet
            self.point_pillar_net = PointPillarNet(config.num_input, config.num_features, min_x=config.min_x, max_x=config.max_x,
                                                   min_y=config.min_y, max_y=config.max_y, pixels_per_meter=int(config.pixels_per_meter))
        tf = backbone_module or transfuser_cpu
        assert backbone in ('transFuser', 'latentTF', 'geometric_fusion', 'late_fusion')
        self.backbone = backbone
        kw = {} if backbone_module is not None else dict(make_net=make_net)
        cls = dict(transFuser='TransfuserBackbone', latentTF='latentTFBackbone', geometric_fusion='GeometricFusionBackbone',
                   late_fusion='LateFusionBackbone')[backbone]
        cls = getattr(tf, cls)
        self._model = cls(config, image_architecture, lidar_architecture, use_velocity=use_velocity, **kw)
        if config.multitask:
            self.seg_decoder = tf.SegDecoder(config, config.perception_output_features)
            self.depth_decoder = tf.DepthDecoder(config, config.perception_output_features)
        ch = config.channel
        self.pred_bev = nn.Sequential(nn.Conv2d(ch, ch, 3, 1, 1), nn.ReLU(inplace=True), nn.Conv2d(ch, 3, 1))
        self.head = LidarCenterNetHead(ch, ch, 1, train_cfg=config)
        self.join = nn.Sequential(nn.Linear(512, 256), nn.ReLU(inplace=True), nn.Linear(256, 128), nn.ReLU(inplace=True),
                                  nn.Linear(128, 64), nn.ReLU(inplace=True))
        self.decoder = nn.GRUCell(input_size=4 if self.gru_concat_target_point else 2, hidden_size=config.gru_hidden_size)
        self.output = nn.Linear(config.gru_hidden_size, 3)

    def forward_gru(self, z, target_point):  # model.py:611-646
        z = self.join(z)
        x = torch.zeros(z.shape[0], 2, dtype=z.dtype, device=z.device)
        tp = target_point.clone()
        tp[:, 1] *= -1
        wps = []
        for _ in range(self.pred_len):
            x_in = torch.cat([x, tp], dim=1) if self.gru_concat_target_point else x
            z = self.decoder(x_in, z)
            x = self.output(z)[:, :2] + x
            wps.append(x)
        pred_wp = torch.stack(wps, dim=1)
        shift = torch.zeros_like(pred_wp)
        shift[:, :, 0] = self.config.lidar_pos[0]
        return pred_wp - shift

    def forward(self, rgb, lidar_bev, ego_waypoint, target_point, target_point_image, ego_vel, bev, label, depth, semantic,
                num_points=None, save_path=None, bev_points=None, cam_points=None):  # model.py:733-805
        if self.use_point_pillars:   # model.py:736-738
            lidar_bev = torch.rot90(self.point_pillar_net(lidar_bev, num_points), -1, dims=(2, 3))
        if self.use_target_point_image:
            lidar_bev = torch.cat((lidar_bev, target_point_image), dim=1)
        if self.backbone == 'geometric_fusion':   # model.py:749-750
            features, grid, fused = self._model(rgb, lidar_bev, ego_vel, bev_points, cam_points)
        else:
            features, grid, fused = self._model(rgb, lidar_bev, ego_vel)
        pred_wp = self.forward_gru(fused, target_point)
        cfg = self.config
        pred_bev = F.interpolate(self.pred_bev(features[0]), (cfg.bev_resolution_height, cfg.bev_resolution_width),
                                 mode='bilinear', align_corners=True)
        loss = dict(loss_wp=torch.mean(torch.abs(pred_wp - ego_waypoint)),
                    loss_bev=F.cross_entropy(pred_bev, bev, weight=pred_bev.new_tensor([1., 1., 3.])).mean())
        preds = self.head.forward_single(features[0])
        loss.update(self.head.loss(preds, label, torch.zeros_like(label[:, :, 0]), label.sum(dim=-1) == 0.))
        if cfg.multitask:
            loss['loss_depth'] = cfg.ls_depth * F.l1_loss(self.depth_decoder(grid), depth).mean()
            loss['loss_semantic'] = cfg.ls_seg * F.cross_entropy(self.seg_decoder(grid), semantic).mean()
        else:
            loss['loss_depth'] = torch.zeros_like(loss['loss_wp'])
            loss['loss_semantic'] = torch.zeros_like(loss['loss_wp'])
        self._last = dict(pred_wp=pred_wp, pred_bev=pred_bev, preds=preds, features=features, grid=grid, fused=fused)
        return loss


def get_lidar_to_bevimage_transform():
    """utils.py:29-37."""
    T = np.array([[0, -1, 16], [-1, 0, 32], [0, 0, 1]], dtype=np.float32)
    T[:2, :] *= 8
    return T


def get_bbox_local_metric(bbox, config):
    """model.py:810-843: pixels -> metres, x front / y right, 4 corners + centre + velocity arrow."""
    x, y, w, h, yaw, speed, brake, confidence = bbox
    w = w / config.bounding_box_divisor / config.pixels_per_meter
    h = h / config.bounding_box_divisor / config.pixels_per_meter
    T_inv = np.linalg.inv(get_lidar_to_bevimage_transform())
    center = T_inv @ np.array([x, y, 1.0]) + np.array(config.lidar_pos)
    center[1] = -center[1]
    box = np.array([[-h, -w, 1], [-h, w, 1], [h, w, 1], [h, -w, 1], [0, 0, 1], [0, h * speed * 0.5, 1]])
    R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    for i in range(box.shape[0]):
        box[i] = R @ box[i]
        box[i] = box[i] + np.array([center[0], center[1], 0])
    return box, brake, confidence


def forward_ego(model, rgb, lidar_bev, target_point, target_point_image, ego_vel):
    """LidarCenterNet.forward_ego (model.py:685-731) without the visualisation branch: (pred_wp, [(bbox (6,3), brake, confidence)])."""
    from .centernet import decode_heatmap
    cfg = model.config
    if model.use_target_point_image:
        lidar_bev = torch.cat((lidar_bev, target_point_image), dim=1)
    features, grid, fused = model._model(rgb, lidar_bev, ego_vel)
    pred_wp = model.forward_gru(fused, target_point)
    preds = model.head.forward_single(features[0])
    boxes, _ = decode_heatmap(preds, model.head.num_dir_bins, k=cfg.top_k_center_keypoints, kernel=cfg.center_net_max_pooling_kernel)
    boxes = boxes[0]
    boxes = boxes[boxes[:, -1] > cfg.bb_confidence_threshold]
    return pred_wp, [get_bbox_local_metric(b, cfg) for b in boxes.detach().cpu().numpy()], boxes


def total_loss(losses, config):
    """train.py:307-311."""
    w = dict(zip(config.detailed_losses, config.detailed_losses_weights))
    total = torch.tensor(0.0)
    for k, v in losses.items():
        total = total + w[k] * v
    return total


def train_step(model, optimizer, batch, config):
    """One iteration of Engine.train (train.py:304-316)."""
    optimizer.zero_grad(set_to_none=True)
    losses = model(batch['rgb'], batch['lidar'], ego_waypoint=batch['ego_waypoint'], target_point=batch['target_point'],
                   target_point_image=batch['target_point_image'], ego_vel=batch['ego_vel'].reshape(-1, 1), bev=batch['bev'],
                   label=batch['label'], depth=batch['depth'], semantic=batch['semantic'],
                   **{k: batch[k] for k in ('bev_points', 'cam_points', 'num_points') if k in batch})
    loss = total_loss(losses, config)
    loss.backward()
    optimizer.step()
    return loss.detach(), {k: v.detach() for k, v in losses.items()}


def make_optimizer(model, lr=1e-4):
    """train.py:142: AdamW, torch defaults (betas .9/.999, eps 1e-8, weight_decay 0.01)."""
    return torch.optim.AdamW(model.parameters(), lr=lr)
