"""CPU fp32 restatement of team_code_transfuser/transfuser.py (TEST INFRASTRUCTURE).

Pinned in the authoring container against the reference's own module imported unmodified
(tests/test_oracle_pinning.py, fixtures from tests/golden/make_golden.py).  Parameter names
are the reference's, so a reference state_dict loads with strict=True.
"""
import math
import torch
from torch import nn
import torch.nn.functional as F

from . import regnet


def normalize_imagenet(x):
    """transfuser.py:419-428."""
    mean = x.new_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = x.new_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return (x / 255.0 - mean) / std


class SelfAttention(nn.Module):
    """transfuser.py:491-527: separate k/q/v Linear, 4 heads, softmax(qk^T/sqrt(hs)), no mask."""

    def __init__(self, c, n_head, attn_pdrop, resid_pdrop):
        super().__init__()
        self.key = nn.Linear(c, c)
        self.query = nn.Linear(c, c)
        self.value = nn.Linear(c, c)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.resid_drop = nn.Dropout(resid_pdrop)
        self.proj = nn.Linear(c, c)
        self.n_head = n_head

    def forward(self, x):
        B, T, C = x.shape
        hs = C // self.n_head
        split = lambda t: t.view(B, T, self.n_head, hs).transpose(1, 2)
        k, q, v = split(self.key(x)), split(self.query(x)), split(self.value(x))
        att = self.attn_drop(F.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hs)), dim=-1))
        y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
        return self.resid_drop(self.proj(y))


class Block(nn.Module):
    """transfuser.py:530-549 (MLP activation is ReLU)."""

    def __init__(self, c, n_head, block_exp, attn_pdrop, resid_pdrop):
        super().__init__()
        self.ln1 = nn.LayerNorm(c)
        self.ln2 = nn.LayerNorm(c)
        self.attn = SelfAttention(c, n_head, attn_pdrop, resid_pdrop)
        self.mlp = nn.Sequential(nn.Linear(c, block_exp * c), nn.ReLU(True), nn.Linear(block_exp * c, c),
                                 nn.Dropout(resid_pdrop))

    def forward(self, x):
        x = x + self.attn(self.ln1(x))
        return x + self.mlp(self.ln2(x))


class GPT(nn.Module):
    """transfuser.py:284-366, incl. quirk Q1 (raw view back to NCHW at :363-364)."""

    def __init__(self, n_embd, config, use_velocity):
        super().__init__()
        self.n_embd = n_embd
        self.n_img = config.img_vert_anchors * config.img_horz_anchors
        self.n_lid = config.lidar_vert_anchors * config.lidar_horz_anchors
        self.pos_emb = nn.Parameter(torch.zeros(1, self.n_img + self.n_lid, n_embd))
        self.use_velocity = use_velocity
        if use_velocity:
            self.vel_emb = nn.Linear(1, n_embd)
        self.drop = nn.Dropout(config.embd_pdrop)
        self.blocks = nn.Sequential(*[Block(n_embd, config.n_head, config.block_exp, config.attn_pdrop, config.resid_pdrop)
                                      for _ in range(config.n_layer)])
        self.ln_f = nn.LayerNorm(n_embd)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(mean=config.gpt_linear_layer_init_mean, std=config.gpt_linear_layer_init_std)
                m.bias.data.zero_()
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(config.gpt_layer_norm_init_weight)

    def forward(self, img, lid, velocity):
        B = lid.shape[0]
        ih, iw = img.shape[2:]
        lh, lw = lid.shape[2:]
        tok = torch.cat((img.flatten(2).transpose(1, 2), lid.flatten(2).transpose(1, 2)), dim=1)
        x = self.pos_emb + tok
        if self.use_velocity:
            x = x + self.vel_emb(velocity).unsqueeze(1)
        x = self.ln_f(self.blocks(self.drop(x)))
        # Q1: (hw, C) token memory is reinterpreted as (C, h, w) WITHOUT permuting back
        return (x[:, :self.n_img].contiguous().view(B, -1, ih, iw),
                x[:, self.n_img:].contiguous().view(B, -1, lh, lw))


class _Trunk(nn.Module):
    """The reference's re-labelled timm trunk (transfuser.py:383-393, 445-455).

    The aliases are real attributes (so state_dict carries the duplicate keys the reference
    checkpoints carry).  ``lidar=True`` reproduces LidarEncoder's conv1 swap + ``del stem.conv``.
    """

    def __init__(self, net, in_channels=None):
        super().__init__()
        net.fc = None
        if hasattr(net, "stages"):          # ConvNeXt: transfuser.py:395-416 (image) / 457-471 (LiDAR), restated line by line
            net.conv1 = net.stem._modules['0']
            net.bn1 = net.stem._modules['1']
            net.act1 = nn.Sequential()
            net.maxpool = nn.Sequential()
            for i in range(4):
                setattr(net, "layer%d" % (i + 1), net.stages._modules[str(i)])
            net.global_pool = net.head
            net.global_pool.flatten = nn.Sequential()
            net.global_pool.fc = nn.Sequential()
            net.head = nn.Sequential()
            if in_channels is None:         # ImageCNN only: ConvNeXt has no stem entry in feature_info (transfuser.py:407-411)
                net.feature_info.append(net.feature_info[3])
                net.feature_info[3] = net.feature_info[2]
                net.feature_info[2] = net.feature_info[1]
                net.feature_info[1] = net.feature_info[0]
            tmp = net.global_pool.norm
            net.global_pool.norm = nn.LayerNorm((512, 1, 1), tmp.eps, tmp.elementwise_affine)     # out_features = perception_output_features
            if in_channels is not None:     # transfuser.py:473-490: new first conv (keeps the old bias parameter), old stem entry deleted
                old = net.conv1
                net.conv1 = nn.Conv2d(in_channels, old.out_channels, old.kernel_size, old.stride, old.padding, bias=True)
                del net.stem._modules['0']
                net.conv1.bias = old.bias
            self.net = net
            return
        if not hasattr(net, "stem"):        # ResNet (the reference's default architectures): timm's own names, nothing to re-label
            if in_channels is not None:     # transfuser.py:475-477
                old = net.conv1
                net.conv1 = nn.Conv2d(in_channels, old.out_channels, old.kernel_size, old.stride, old.padding, bias=False)
            self.net = net
            return
        net.conv1 = net.stem.conv
        net.bn1 = net.stem.bn
        net.act1 = nn.Sequential()
        net.maxpool = nn.Sequential()
        for i in range(1, 5):
            setattr(net, "layer%d" % i, getattr(net, "s%d" % i))
        net.global_pool = nn.AdaptiveAvgPool2d(1)
        net.head = nn.Sequential()
        if in_channels is not None:
            old = net.conv1
            net.conv1 = nn.Conv2d(in_channels, old.out_channels, old.kernel_size, old.stride, old.padding, bias=False)
            del net.stem.conv
        self.net = net


class ImageCNN(nn.Module):
    def __init__(self, make_net):
        super().__init__()
        self.normalize = True
        self.features = _Trunk(make_net()).net


class LidarEncoder(nn.Module):
    def __init__(self, make_net, in_channels):
        super().__init__()
        self._model = _Trunk(make_net(), in_channels).net


class TransfuserBackbone(nn.Module):
    """transfuser.py:7-211."""

    def __init__(self, config, image_architecture='regnety_032', lidar_architecture='regnety_032', use_velocity=True,
                 make_net=None):
        super().__init__()
        self.config = config
        make_net = make_net or regnet.regnety_032
        self.avgpool_img = nn.AdaptiveAvgPool2d((config.img_vert_anchors, config.img_horz_anchors))
        self.avgpool_lidar = nn.AdaptiveAvgPool2d((config.lidar_vert_anchors, config.lidar_horz_anchors))
        self.image_encoder = ImageCNN(make_net)
        in_ch = config.num_features[-1] if config.use_point_pillars else 2 * config.lidar_seq_len
        if config.use_target_point_image:
            in_ch += 1
        self.lidar_encoder = LidarEncoder(make_net, in_ch)
        chs = [f['num_chs'] for f in self.image_encoder.features.feature_info]
        for i in range(1, 5):
            setattr(self, "transformer%d" % i, GPT(chs[i], config, use_velocity))
        pf = config.perception_output_features
        if chs[4] != pf:
            self.change_channel_conv_image = nn.Conv2d(chs[4], pf, (1, 1))
            self.change_channel_conv_lidar = nn.Conv2d(chs[4], pf, (1, 1))
        else:
            self.change_channel_conv_image = nn.Sequential()
            self.change_channel_conv_lidar = nn.Sequential()
        ch = config.bev_features_chanels
        self.relu = nn.ReLU(inplace=True)
        self.upsample = nn.Upsample(scale_factor=config.bev_upsample_factor, mode='bilinear', align_corners=False)
        self.up_conv5 = nn.Conv2d(ch, ch, (1, 1))
        self.up_conv4 = nn.Conv2d(ch, ch, (1, 1))
        self.up_conv3 = nn.Conv2d(ch, ch, (1, 1))
        self.c5_conv = nn.Conv2d(pf, ch, (1, 1))

    def top_down(self, x):
        p5 = self.relu(self.c5_conv(x))
        p4 = self.relu(self.up_conv5(self.upsample(p5)))
        p3 = self.relu(self.up_conv4(self.upsample(p4)))
        p2 = self.relu(self.up_conv3(self.upsample(p3)))
        return p2, p3, p4, p5

    def forward(self, image, lidar, velocity):
        im, li = self.image_encoder.features, self.lidar_encoder._model
        x = im.maxpool(im.act1(im.bn1(im.conv1(normalize_imagenet(image)))))      # transfuser.py:136-143 (act1 / maxpool are empty for RegNet)
        y = li.maxpool(li.act1(li.bn1(li.conv1(lidar))))
        for i in range(1, 5):
            x = getattr(im, "layer%d" % i)(x)
            y = getattr(li, "layer%d" % i)(y)
            fx, fy = getattr(self, "transformer%d" % i)(self.avgpool_img(x), self.avgpool_lidar(y), velocity)
            x = x + F.interpolate(fx, size=x.shape[2:], mode='bilinear', align_corners=False)
            y = y + F.interpolate(fy, size=y.shape[2:], mode='bilinear', align_corners=False)
        x = self.change_channel_conv_image(x)
        y = self.change_channel_conv_lidar(y)
        fused = torch.flatten(im.global_pool(x), 1) + torch.flatten(li.global_pool(y), 1)
        return self.top_down(y), x, fused


class latentTFBackbone(TransfuserBackbone):
    """team_code_transfuser/latentTF.py:8-217: LiDAR channels 0/1 overwritten IN PLACE by a (-1..1) meshgrid (:132-137,
    quirk Q5); LidarEncoder deletes the whole stem (:416)."""

    def __init__(self, config, image_architecture='regnety_032', lidar_architecture='regnety_032', use_velocity=True, make_net=None):
        super().__init__(config, image_architecture, lidar_architecture, use_velocity, make_net)
        del self.lidar_encoder._model.stem

    def forward(self, image, lidar, velocity):
        x = torch.linspace(-1, 1, self.config.lidar_resolution_width)
        y = torch.linspace(-1, 1, self.config.lidar_resolution_height)
        y_grid, x_grid = torch.meshgrid(x, y, indexing='ij')
        lidar[:, 0] = y_grid.unsqueeze(0).to(lidar.dtype)
        lidar[:, 1] = x_grid.unsqueeze(0).to(lidar.dtype)
        return super().forward(image, lidar, velocity)


class LateFusionBackbone(nn.Module):
    """team_code_transfuser/late_fusion.py:5-111: the two trunks run WITHOUT any exchange (timm models used as they
    are - no re-labelling, the LiDAR trunk is created with in_chans, late_fusion.py:126-130,155-159), 1x1 reducers to 512, FPN on the
    LiDAR map, fused = gap(image) + gap(lidar) (+ vel_emb(velocity)).  ``make_net`` builds either trunk (any of oracle.regnet / resnet / convnext:
    the reference takes every timm architecture, :126,158); a ConvNeXt trunk's pooled vector goes through LayerNorm(512) (:23-33,92,103)."""

    def __init__(self, config, image_architecture='regnety_032', lidar_architecture='regnety_032', use_velocity=0, make_net=None):
        super().__init__()
        self.config = config
        make_net = make_net or regnet.regnety_032
        in_ch = config.num_features[-1] if config.use_point_pillars else 2 * config.lidar_seq_len
        if config.use_target_point_image:
            in_ch += 1

        def bare(net):
            for name in ("fc", "classifier", "global_pool", "head"):
                setattr(net, name, nn.Sequential())
            return net
        self.image_encoder = nn.Module()
        self.image_encoder.normalize = True
        self.image_encoder.features = bare(make_net())
        self.lidar_encoder = nn.Module()
        self.lidar_encoder._model = bare(make_net(in_chans=in_ch))
        pf = config.perception_output_features
        self.norm_after_pool_img = nn.LayerNorm((pf,), eps=1e-06) if image_architecture.startswith('convnext') else nn.Sequential()
        self.norm_after_pool_lidar = nn.LayerNorm((pf,), eps=1e-06) if lidar_architecture.startswith('convnext') else nn.Sequential()
        self.use_velocity = use_velocity
        if use_velocity:
            self.vel_emb = nn.Linear(1, pf)
        ch = config.bev_features_chanels
        self.relu = nn.ReLU(inplace=True)
        nf = self.image_encoder.features.num_features
        self.reduce_channels_conv_image = nn.Conv2d(nf, pf, (1, 1)) if nf != pf else nn.Sequential()
        self.reduce_channels_conv_lidar = nn.Conv2d(self.lidar_encoder._model.num_features, pf, (1, 1)) if nf != pf else nn.Sequential()
        self.upsample = nn.Upsample(scale_factor=config.bev_upsample_factor, mode='bilinear', align_corners=False)
        self.up_conv5 = nn.Conv2d(ch, ch, (1, 1))
        self.up_conv4 = nn.Conv2d(ch, ch, (1, 1))
        self.up_conv3 = nn.Conv2d(ch, ch, (1, 1))
        self.c5_conv = nn.Conv2d(pf, ch, (1, 1))

    top_down = TransfuserBackbone.top_down

    def forward(self, image, lidar, velocity):
        x = self.reduce_channels_conv_image(self.image_encoder.features(normalize_imagenet(image)))
        y = self.reduce_channels_conv_lidar(self.lidar_encoder._model(lidar))
        fused = self.norm_after_pool_img(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)) + self.norm_after_pool_lidar(torch.flatten(F.adaptive_avg_pool2d(y, 1), 1))
        if self.use_velocity:
            fused = fused + self.vel_emb(velocity)
        return self.top_down(y), x, fused


class GeometricFusionBackbone(TransfuserBackbone):
    """team_code_transfuser/geometric_fusion.py:6-288 (BASELINE config 4), in the reference's operation order: 1x1 conv C->n_embd
    on the full map, adaptive pool, gather (the reference's B x B advanced index + ``torch.diagonal`` :134-135 is restated as the
    equivalent per-sample gather), sum over the 5 correspondences, 3 x [Linear+ReLU], ``F.interpolate(scale_factor=8/4/2)``
    (none at stage 4), 1x1 conv n_embd->C, residual add, optional velocity embedding.  Quirk Q4 (:264): stage 4's image side
    reads ``lidar_embd_layer3``.  LidarEncoder deletes the whole stem (:417)."""

    SCALES = (8, 4, 2, None)

    def __init__(self, config, image_architecture='regnety_032', lidar_architecture='regnety_032', use_velocity=0, make_net=None):
        super().__init__(config, image_architecture, lidar_architecture, use_velocity, make_net)
        for i in range(1, 5):
            delattr(self, "transformer%d" % i)
        del self.lidar_encoder._model.stem
        self.use_velocity = use_velocity
        chs = [f['num_chs'] for f in self.image_encoder.features.feature_info]
        E = config.n_embd
        mlp = lambda: nn.Sequential(nn.Linear(E, E), nn.ReLU(True), nn.Linear(E, E), nn.ReLU(True), nn.Linear(E, E), nn.ReLU(True))
        for i in range(1, 5):
            setattr(self, "image_conv%d" % i, nn.Conv2d(chs[i], E, 1))
            setattr(self, "image_deconv%d" % i, nn.Conv2d(E, chs[i], 1))
            setattr(self, "lidar_conv%d" % i, nn.Conv2d(chs[i], E, 1))
            setattr(self, "lidar_deconv%d" % i, nn.Conv2d(E, chs[i], 1))
            setattr(self, "image_projection%d" % i, mlp())
            setattr(self, "lidar_projection%d" % i, mlp())
            if use_velocity:
                setattr(self, "vel_emb%d" % i, nn.Linear(1, chs[i]))

    @staticmethod
    def _gather_sum(emb, pts, h, w):
        """emb (B,E,hs,ws); pts flat (B*h*w*5, 2) = (x, y): out[b,:,i,j] = sum_k emb[b, :, y, x]   (:133-136)."""
        B, E = emb.shape[:2]
        pts = pts.reshape(B, h * w * 5, 2)
        b = torch.arange(B).view(B, 1).expand(B, h * w * 5)
        g = emb.permute(0, 2, 3, 1)[b, pts[..., 1], pts[..., 0]]            # (B, h*w*5, E)
        # same memory layout as the reference before its sum (:135-136: (B, E, h, w, 5) contiguous, reduced over the last axis), so the
        # 5-term additions associate identically -> bit-exact against the reference import
        return g.view(B, h, w, 5, E).permute(0, 4, 1, 2, 3).contiguous().sum(-1).permute(0, 2, 3, 1)

    def forward(self, image, lidar, velocity, bev_points, img_points):
        im, li = self.image_encoder.features, self.lidar_encoder._model
        x = im.maxpool(im.act1(im.bn1(im.conv1(normalize_imagenet(image)))))      # transfuser.py:136-143 (act1 / maxpool are empty for RegNet)
        y = li.maxpool(li.act1(li.bn1(li.conv1(lidar))))
        lid_embd = {}
        for i in range(1, 5):
            x = getattr(im, "layer%d" % i)(x)
            y = getattr(li, "layer%d" % i)(y)
            if self.config.n_scale < 5 - i:
                continue
            img_e = self.avgpool_img(getattr(self, "image_conv%d" % i)(x))
            lid_e = self.avgpool_lidar(getattr(self, "lidar_conv%d" % i)(y))
            lid_embd[i] = lid_e
            hi, wi = img_e.shape[-2:]
            hl, wl = lid_e.shape[-2:]
            sf = self.SCALES[i - 1]
            up = (lambda t: F.interpolate(t, scale_factor=sf, mode='bilinear', align_corners=False)) if sf else (lambda t: t)
            bev = getattr(self, "image_projection%d" % i)(self._gather_sum(img_e, bev_points, hl, wl)).permute(0, 3, 1, 2).contiguous()
            y = y + getattr(self, "lidar_deconv%d" % i)(up(bev))
            if self.use_velocity:
                vel = getattr(self, "vel_emb%d" % i)(velocity).unsqueeze(-1).unsqueeze(-1)
                y = y + vel
            src = lid_embd[3] if i == 4 else lid_e                           # quirk Q4 (:264)
            img = getattr(self, "lidar_projection%d" % i)(self._gather_sum(src, img_points, hi, wi)).permute(0, 3, 1, 2).contiguous()
            x = x + getattr(self, "image_deconv%d" % i)(up(img))
            if self.use_velocity:
                x = x + vel
        x = self.change_channel_conv_image(x)
        y = self.change_channel_conv_lidar(y)
        fused = torch.flatten(im.global_pool(x), 1) + torch.flatten(li.global_pool(y), 1)
        return self.top_down(y), x, fused


def _decoder(config, latent, out_ch):
    c1, c2, c3 = config.deconv_channel_num_1, config.deconv_channel_num_2, config.deconv_channel_num_3
    conv = lambda a, b: nn.Conv2d(a, b, 3, 1, 1)
    return (nn.Sequential(conv(latent, c1), nn.ReLU(True), conv(c1, c2), nn.ReLU(True)),
            nn.Sequential(conv(c2, c3), nn.ReLU(True), conv(c3, c3), nn.ReLU(True)),
            nn.Sequential(conv(c3, c3), nn.ReLU(True), conv(c3, out_ch)))


class SegDecoder(nn.Module):
    """transfuser.py:214-246."""

    def __init__(self, config, latent_dim=512):
        super().__init__()
        self.config = config
        self.deconv1, self.deconv2, self.deconv3 = _decoder(config, latent_dim, config.num_class)

    def forward(self, x):
        x = F.interpolate(self.deconv1(x), scale_factor=self.config.deconv_scale_factor_1, mode='bilinear', align_corners=False)
        x = F.interpolate(self.deconv2(x), scale_factor=self.config.deconv_scale_factor_2, mode='bilinear', align_corners=False)
        return self.deconv3(x)


class DepthDecoder(SegDecoder):
    """transfuser.py:249-281: same trunk, 1 output channel, sigmoid + squeeze."""

    def __init__(self, config, latent_dim=512):
        nn.Module.__init__(self)
        self.config = config
        self.deconv1, self.deconv2, self.deconv3 = _decoder(config, latent_dim, 1)

    def forward(self, x):
        return torch.sigmoid(super().forward(x)).squeeze(1)
