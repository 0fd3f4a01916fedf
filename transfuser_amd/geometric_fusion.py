"""Geometric-fusion backbone (BASELINE config 4), MI355X-native drop-in for
team_code_transfuser/geometric_fusion.py (``GeometricFusionBackbone`` :6-288).

Same constructor, parameter names / shapes (checkpoints interchange with strict=True) and return values as the
reference.  Per stage the reference projects both feature maps to ``n_embd`` channels, average-pools them to the anchor
grids, lets every BEV cell sum the image features of its 5 projected LiDAR points (and vice versa), runs a 3-layer MLP,
up-samples by a FIXED factor 8/4/2/1 and adds the result back through a 1x1 conv.  Here that whole stage is one autograd
node (``functions.GeoStageFn``) built from the gather kernel G1 and the MFMA GEMM engine.

Reference behaviours kept on purpose:
* fixed scale factors: like the reference this backbone only works when stage i's maps are exactly (anchors x 8/4/2/1),
  i.e. 160x704 images and 256x256 BEVs (at 256x704 the reference fails with a shape error; so do we, with a message);
* quirk Q4 (:264): stage 4's image branch reads the stage-3 LiDAR embedding, so ``lidar_conv4`` never gets a gradient
  (``unused_parameters`` lists it: torch.optim.AdamW skips grad-None parameters, and so does our flat AdamW);
* ``bev_points`` / ``img_points`` are re-viewed flat as (B*h*w*5, 2) (:133,146) whatever their nominal shape is.
"""
import types

import torch
from torch import nn

from . import functions as F_
from .transfuser import _FusionBackbone, nchw


class GeometricFusionBackbone(_FusionBackbone):
    SCALES = (8, 4, 2, 1)   # geometric_fusion.py:139,177,216 and none at :253

    def __init__(self, config, image_architecture='resnet34', lidar_architecture='resnet18', use_velocity=0):
        super().__init__()
        self.use_velocity = use_velocity
        chs = self._build_common(config, image_architecture, lidar_architecture)
        del self.lidar_encoder._model.stem   # geometric_fusion.py:417 (whole stem dropped; ``bn1`` keeps the BatchNorm)
        self.avgpool_img = nn.AdaptiveAvgPool2d((config.img_vert_anchors, config.img_horz_anchors))
        self.avgpool_lidar = nn.AdaptiveAvgPool2d((config.lidar_vert_anchors, config.lidar_horz_anchors))
        E = config.n_embd
        for i in range(1, 5):
            setattr(self, "image_conv%d" % i, nn.Conv2d(chs[i], E, 1))
        for i in range(1, 5):
            setattr(self, "image_deconv%d" % i, nn.Conv2d(E, chs[i], 1))
        if use_velocity:
            for i in range(1, 5):
                setattr(self, "vel_emb%d" % i, nn.Linear(1, chs[i]))
        for i in range(1, 5):
            setattr(self, "lidar_conv%d" % i, nn.Conv2d(chs[i], E, 1))
        for i in range(1, 5):
            setattr(self, "lidar_deconv%d" % i, nn.Conv2d(E, chs[i], 1))
        mlp = lambda: nn.Sequential(nn.Linear(E, E), nn.ReLU(True), nn.Linear(E, E), nn.ReLU(True), nn.Linear(E, E), nn.ReLU(True))
        for i in range(1, 5):
            setattr(self, "image_projection%d" % i, mlp())
        for i in range(1, 5):
            setattr(self, "lidar_projection%d" % i, mlp())
        self._build_neck(chs)
        geom = types.SimpleNamespace(ih=config.img_vert_anchors, iw=config.img_horz_anchors, lh=config.lidar_vert_anchors, lw=config.lidar_horz_anchors)
        self._stages = [types.SimpleNamespace(
            geom=geom, scale=self.SCALES[i - 1], use_prev=(i == 4),
            image_conv=getattr(self, "image_conv%d" % i), lidar_conv=getattr(self, "lidar_conv%d" % i),
            image_deconv=getattr(self, "image_deconv%d" % i), lidar_deconv=getattr(self, "lidar_deconv%d" % i),
            image_projection=getattr(self, "image_projection%d" % i), lidar_projection=getattr(self, "lidar_projection%d" % i),
            vel_emb=getattr(self, "vel_emb%d" % i) if use_velocity else None) for i in range(1, 5)]

    def unused_parameters(self):
        """Parameters the reference's graph never reaches (grad stays None there): quirk Q4."""
        out = [] if self.config.n_scale < 1 else [self.lidar_conv4.weight, self.lidar_conv4.bias]
        for i, st in enumerate(self._stages, 1):          # stages that do not fuse (n_scale < 5 - i) keep all their modules untouched
            if self.config.n_scale < 5 - i:
                mods = [st.image_conv, st.image_deconv, st.lidar_deconv, st.image_projection, st.lidar_projection] + ([st.vel_emb] if st.vel_emb is not None else [])
                mods += [st.lidar_conv] if i != 4 else []
                out += [p for m in mods for p in m.parameters()]
        return out

    @staticmethod
    def _flat_idx(pts, B, n):
        """(B, ..., 2) int64 correspondences -> (B, n, 5, 2), the reference's flat re-view (:133,146)."""
        assert pts is not None and pts.dtype == torch.int64 and pts.numel() == B * n * 5 * 2, \
            "geometric fusion needs int64 bev_points / cam_points with B*%d*5*2 elements" % n
        return pts.contiguous().view(B, n, 5, 2)

    def forward_nhwc(self, image, lidar, velocity, bev_points, img_points, lidar_extra=None, lidar_nhwc=None):
        cfg = self.config
        B = image.shape[0]
        g = self._stages[0].geom
        bev_idx = self._flat_idx(bev_points, B, g.lh * g.lw)
        img_idx = self._flat_idx(img_points, B, g.ih * g.iw)
        vel = velocity.reshape(B, 1).contiguous() if self.use_velocity else None
        carry = {"lid_e": None}

        def fuse(i, x, y):
            if cfg.n_scale < 5 - i:     # geometric_fusion.py:123,161,201,241: stage i fuses iff n_scale >= 5 - i
                return x, y
            st = self._stages[i - 1]
            assert not st.use_prev or carry["lid_e"] is not None, "stage 4 reads stage 3's LiDAR embedding (quirk Q4): n_scale must be >= 2"
            params = [p for m in (st.image_conv, st.lidar_conv, st.image_deconv, st.lidar_deconv, st.image_projection, st.lidar_projection)
                      for p in m.parameters()] + (list(st.vel_emb.parameters()) if st.vel_emb is not None else [])
            x, y, lid_e = F_.GeoStageFn.apply(x, y, carry["lid_e"] if st.use_prev else None, st, vel, bev_idx, img_idx, *params)
            if lid_e is not None:
                carry["lid_e"] = lid_e
            return x, y
        return self._run(image, lidar, lidar_extra, fuse, lidar_nhwc)

    def forward(self, image, lidar, velocity, bev_points, img_points):
        feats, grid, fused = self.forward_nhwc(image, lidar, velocity, bev_points, img_points)
        return tuple(nchw(p) for p in feats), nchw(grid), fused
