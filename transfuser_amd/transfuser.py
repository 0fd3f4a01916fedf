"""TransFuser backbone + auxiliary decoders, MI355X-native drop-in for
team_code_transfuser/transfuser.py (``TransfuserBackbone`` :7-211, ``SegDecoder`` :214, ``DepthDecoder``
:249, ``GPT`` :284, ``ImageCNN`` :369, ``LidarEncoder`` :431, ``SelfAttention`` :491, ``Block`` :530).

Same constructor signatures, parameter names / shapes and return values; the arithmetic runs in the
hand-written HIP kernels of ``libtransfuser_hip.so`` (there is no PyTorch fallback).  Internally
feature maps are NHWC; tensors returned through the public API are NCHW-shaped views of them.
"""
import types

import torch
from torch import nn

from . import functions as F_
from . import regnet
from . import resnet
from . import convnext


def nchw(x):
    """NHWC tensor -> NCHW-shaped view (channels_last memory format), zero copy."""
    return x.permute(0, 3, 1, 2)


def nhwc(x):
    """NCHW-shaped tensor -> contiguous NHWC (free when the tensor is already channels_last)."""
    return x.permute(0, 2, 3, 1).contiguous()


class SelfAttention(nn.Module):
    def __init__(self, n_embd, n_head, attn_pdrop, resid_pdrop):
        super().__init__()
        assert n_embd % n_head == 0
        self.key = nn.Linear(n_embd, n_embd)
        self.query = nn.Linear(n_embd, n_embd)
        self.value = nn.Linear(n_embd, n_embd)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.resid_drop = nn.Dropout(resid_pdrop)
        self.proj = nn.Linear(n_embd, n_embd)
        self.n_head = n_head
        self._fused = None

    def fused(self):
        """(Wkqv (3C,C), bkqv (3C), grad views) when key/query/value are adjacent in the parameter arena
        (transfuser_amd.train.ParamArena lays them out that way): one N=3C GEMM instead of three."""
        f = self._fused
        kw = self.key.weight
        if f is not None and f[4] == (kw.data_ptr(), kw.grad.data_ptr() if kw.grad is not None else 0):
            return f[:4]
        ws = [self.key.weight, self.query.weight, self.value.weight]
        bs = [self.key.bias, self.query.bias, self.value.bias]
        C = kw.shape[0]

        def adjacent(ts):
            return all(t.is_contiguous() and ts[i + 1].data_ptr() == t.data_ptr() + t.numel() * 4 for i, t in enumerate(ts[:-1])) and \
                ts[0].untyped_storage().data_ptr() == ts[-1].untyped_storage().data_ptr()

        if any(p.grad is None for p in ws + bs):
            return None
        if not (adjacent(ws) and adjacent(bs) and adjacent([p.grad for p in ws]) and adjacent([p.grad for p in bs])):
            return None
        view = lambda t, shape: torch.as_strided(t, shape, (shape[1], 1) if len(shape) == 2 else (1,), t.storage_offset())
        f = (view(ws[0].data, (3 * C, C)), view(bs[0].data, (3 * C,)), view(ws[0].grad, (3 * C, C)), view(bs[0].grad, (3 * C,)),
             (kw.data_ptr(), kw.grad.data_ptr()))
        self._fused = f
        return f[:4]


class Block(nn.Module):
    def __init__(self, n_embd, n_head, block_exp, attn_pdrop, resid_pdrop):
        super().__init__()
        self.ln1 = nn.LayerNorm(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        self.attn = SelfAttention(n_embd, n_head, attn_pdrop, resid_pdrop)
        self.mlp = nn.Sequential(nn.Linear(n_embd, block_exp * n_embd), nn.ReLU(True), nn.Linear(block_exp * n_embd, n_embd),
                                 nn.Dropout(resid_pdrop))


class GPT(nn.Module):
    """transfuser.py:284-366.  ``forward`` takes/returns NHWC maps of the two branches and performs the
    whole fusion stage (pool -> tokens -> blocks -> ln_f -> raw view -> up-sample -> residual add)."""

    _site_base = 0

    def __init__(self, n_embd, n_head, block_exp, n_layer, img_vert_anchors, img_horz_anchors, lidar_vert_anchors, lidar_horz_anchors,
                 seq_len, embd_pdrop, attn_pdrop, resid_pdrop, config, use_velocity=True):
        super().__init__()
        self.n_embd = n_embd
        self.n_head = n_head
        self.seq_len = 1
        self.geom = types.SimpleNamespace(ih=img_vert_anchors, iw=img_horz_anchors, lh=lidar_vert_anchors, lw=lidar_horz_anchors)
        self.config = config
        self.pos_emb = nn.Parameter(torch.zeros(1, img_vert_anchors * img_horz_anchors + lidar_vert_anchors * lidar_horz_anchors, n_embd))
        self.use_velocity = use_velocity
        if use_velocity:
            self.vel_emb = nn.Linear(self.seq_len, n_embd)
        self.drop = nn.Dropout(embd_pdrop)
        self.embd_pdrop, self.attn_pdrop, self.resid_pdrop = float(embd_pdrop), float(attn_pdrop), float(resid_pdrop)
        self.pdrop_any = max(self.embd_pdrop, self.attn_pdrop, self.resid_pdrop) > 0
        self.blocks = nn.Sequential(*[Block(n_embd, n_head, block_exp, attn_pdrop, resid_pdrop) for _ in range(n_layer)])
        self.ln_f = nn.LayerNorm(n_embd)
        self.block_size = self.seq_len
        self.site_base = GPT._site_base
        GPT._site_base += 4 * n_layer + 1
        self.seed = None  # int32 device tensor shared by the backbone (dropout RNG key)
        self.apply(self._init_weights)

    def _init_weights(self, module):  # transfuser.py:324-331
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=self.config.gpt_linear_layer_init_mean, std=self.config.gpt_linear_layer_init_std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(self.config.gpt_layer_norm_init_weight)

    def site(self, i):
        return self.site_base + i

    def forward(self, image_nhwc, lidar_nhwc, velocity, cut=None):
        """``cut``: the backbone's backward-cut hook for this stage (functions.gpt_stage); None = one autograd node."""
        assert image_nhwc.shape[-1] == self.n_embd and lidar_nhwc.shape[-1] == self.n_embd
        return F_.gpt_stage(self, image_nhwc, lidar_nhwc, velocity, cut)


class _Stem:
    """First conv + BN of a trunk as StemFn sees them (the modules stay registered under the reference's names)."""

    def __init__(self, conv, bn, normalize, owner=None, names=None, maxpool=False):
        """``owner`` / ``names`` = (module, (conv attribute, bn attribute)): when given the two layers are looked up at every call, so a
        BatchNorm replaced after construction (torch.nn.SyncBatchNorm.convert_sync_batchnorm, train.py:133) is picked up."""
        self._conv, self._bn, self.normalize, self._owner, self._names = conv, bn, normalize, owner, names
        self.maxpool = bool(maxpool)      # ResNet: the 3x3 / s2 max pool behind act1 (transfuser.py:139,143) runs inside StemFn

    @property
    def conv(self):
        return getattr(self._owner, self._names[0]) if self._owner is not None else self._conv

    @property
    def bn(self):
        return getattr(self._owner, self._names[1]) if self._owner is not None else self._bn

    def __call__(self, s0, s1=None):
        if isinstance(self.bn, nn.LayerNorm):      # ConvNeXt patchify stem: conv (bias) + LayerNorm2d
            return F_.CnxStemFn.apply(s0, s1, self, self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias)
        return F_.StemFn.apply(s0, s1, self, self.conv.weight, self.bn.weight, self.bn.bias)


def _relabel(net):
    """The reference's re-labelling of a timm RegNet (transfuser.py:383-393, 445-455)."""
    net.fc = None
    net.conv1 = net.stem.conv
    net.bn1 = net.stem.bn
    net.act1 = nn.Sequential()
    net.maxpool = nn.Sequential()
    net.layer1, net.layer2, net.layer3, net.layer4 = net.s1, net.s2, net.s3, net.s4
    net.global_pool = nn.AdaptiveAvgPool2d(output_size=1)
    net.head = nn.Sequential()


def _create_trunk(architecture, pretrained, out_features=512, lidar_in_channels=None):
    """timm.create_model + the reference's re-labelling for the architectures built here: RegNetY (transfuser.py:383-393), ConvNeXt (:395-416)
    and the ResNets its constructors default to (timm's own names: nothing to re-label).  lidar_in_channels: the LidarEncoder variant
    (transfuser.py:473-490: new first convolution, old one deleted)."""
    if convnext.is_convnext(architecture):
        return convnext.relabel(convnext.create_model(architecture, pretrained=pretrained), out_features, lidar_in_channels)
    if resnet.is_resnet(architecture):
        net = resnet.create_model(architecture, pretrained=pretrained)
        net.fc = None
    else:
        net = regnet.create_model(architecture, pretrained=pretrained)
        _relabel(net)
    if lidar_in_channels is not None:
        old = net.conv1
        net.conv1 = nn.Conv2d(lidar_in_channels, old.out_channels, kernel_size=old.kernel_size, stride=old.stride, padding=old.padding, bias=False)
        if hasattr(net, "stem"):
            del net.stem.conv  # transfuser.py:482-483
    return net


class ImageCNN(nn.Module):
    def __init__(self, architecture, normalize=True, out_features=512):
        super().__init__()
        self.normalize = normalize
        self.features = _create_trunk(architecture, True, out_features)


class LidarEncoder(nn.Module):
    def __init__(self, architecture, in_channels=2, out_features=512):
        super().__init__()
        self._model = _create_trunk(architecture, False, out_features, in_channels)


class _FusionBackbone(nn.Module):
    """What the reference's backbones share: the two RegNet trunks, the 1512->512 channel reducers, the FPN top-down path
    (transfuser.py:91-118) and the two-stream execution of the trunks."""

    _reducers = ("change_channel_conv_image", "change_channel_conv_lidar")
    # Backward cuts (set by train.Engine; empty = one autograd graph).  A cut is a key (stage, where, j):
    #   (i, 2, 0)  after fusion stage i (both maps);  (i, 0, 0)  between the trunks' stage i and GPT i (both maps);
    #   (i, 1, j)  inside GPT i in front of Block j (the token matrix; j = 0: between the embedding and Block 0).
    # Forward order of the keys = their tuple order; an int c is shorthand for (c, 2, 0).
    _cuts = frozenset()

    @staticmethod
    def cut_key(c):
        return (int(c), 2, 0) if isinstance(c, int) else tuple(int(v) for v in c)

    def _cut(self, key, tensors, leaves=None):
        """The backward is severed at ``tensors`` when ``key`` is one of the engine's cuts: values are untouched (same memory, no copy);
        the engine restarts the backward of everything before the cut from the detached leaves, with the gradients that arrived at them,
        after it has handed the later segments' gradient range to the all-reduce.  tensors=None only asks whether the key is a cut;
        ``leaves`` may name leaves that already stand in for some of the tensors downstream (functions.gpt_stage's residual maps)."""
        hit = key in self._cuts
        if tensors is None:
            return hit
        if not (hit and torch.is_grad_enabled() and tensors[0].requires_grad):
            return tensors
        leaves = tuple(l if l is not None else t.detach().requires_grad_(True) for t, l in zip(tensors, leaves or (None,) * len(tensors)))
        self._boundaries.append((tuple(tensors), leaves))
        return leaves

    def _build_common(self, config, image_architecture, lidar_architecture):
        self.config = config
        self.image_encoder = ImageCNN(architecture=image_architecture, normalize=True, out_features=config.perception_output_features)
        in_channels = config.num_features[-1] if config.use_point_pillars else 2 * config.lidar_seq_len
        if config.use_target_point_image:
            in_channels += 1
        self.lidar_encoder = LidarEncoder(architecture=lidar_architecture, in_channels=in_channels, out_features=config.perception_output_features)
        return [f['num_chs'] for f in self.image_encoder.features.feature_info]

    def _build_neck(self, chs):
        config = self.config
        pf = config.perception_output_features
        if chs[4] != pf:
            self.change_channel_conv_image = nn.Conv2d(chs[4], pf, (1, 1))
            self.change_channel_conv_lidar = nn.Conv2d(chs[4], pf, (1, 1))
        else:
            self.change_channel_conv_image = nn.Sequential()
            self.change_channel_conv_lidar = nn.Sequential()
        channel = config.bev_features_chanels
        self.relu = nn.ReLU(inplace=True)
        self.upsample = nn.Upsample(scale_factor=config.bev_upsample_factor, mode='bilinear', align_corners=False)
        self.up_conv5 = nn.Conv2d(channel, channel, (1, 1))
        self.up_conv4 = nn.Conv2d(channel, channel, (1, 1))
        self.up_conv3 = nn.Conv2d(channel, channel, (1, 1))
        self.c5_conv = nn.Conv2d(pf, channel, (1, 1))
        has_pool = lambda net: isinstance(getattr(net, "maxpool", None), nn.MaxPool2d)
        self._img_stem = _Stem(None, None, True, self.image_encoder.features, ("conv1", "bn1"), has_pool(self.image_encoder.features))
        self._lid_stem = _Stem(None, None, False, self.lidar_encoder._model, ("conv1", "bn1"), has_pool(self.lidar_encoder._model))

    def _side_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device)
            self._side = st
        return st

    def _conv(self, conv, x, relu=False):
        if isinstance(conv, nn.Sequential):
            return x
        return F_.ConvFn.apply(x, conv.weight, conv.bias, relu)

    def _up(self, x):
        B, H, W, C = x.shape
        f = int(self.config.bev_upsample_factor)
        return F_.UpsampleFn.apply(x, H * f, W * f, False)

    def top_down_nhwc(self, x):
        p5 = self._conv(self.c5_conv, x, True)
        p4 = self._conv(self.up_conv5, self._up(p5), True)
        p3 = self._conv(self.up_conv4, self._up(p4), True)
        p2 = self._conv(self.up_conv3, self._up(p3), True)
        return p2, p3, p4, p5

    def top_down(self, x):
        return tuple(nchw(p) for p in self.top_down_nhwc(nhwc(x)))

    def _run(self, image, lidar, lidar_extra, fuse, lidar_nhwc=None):
        """Stem + 4 stages of both trunks with ``fuse(i, x, y) -> (x, y)`` after stage i, then the neck.
        The two trunks are independent between fusion stages: the LiDAR branch runs on a side HIP stream so its blocks fill
        the tail rounds of the image branch's kernels (and vice versa); under hipGraph capture the fork/join below become
        graph edges.  Autograd replays each node's backward on its forward stream, so the backward overlaps the same way."""
        im, li = self.image_encoder.features, self.lidar_encoder._model
        side = self._side_stream(image.device) if image.is_cuda else None
        main = torch.cuda.current_stream(image.device) if image.is_cuda else None

        def lidar_branch(fn):
            if side is None:
                return fn()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                out = fn()
            return out

        # fork BEFORE the image branch's kernels are enqueued (side.wait_stream(main) orders the LiDAR branch after everything already on
        # main): the two trunks of a stage are then independent graph branches instead of image-then-LiDAR
        if lidar_nhwc is not None:    # PointPillars canvas (already NHWC, rotated, target-point channel appended): differentiable stem
            from .point_pillar import PillarStemFn
            st = self._lid_stem
            y = lidar_branch(lambda: PillarStemFn.apply(lidar_nhwc, st, st.conv.weight, st.bn.weight, st.bn.bias))
        else:
            y = lidar_branch(lambda: self._lid_stem(lidar.contiguous(), lidar_extra.contiguous() if lidar_extra is not None else None))
        x = self._img_stem(image.contiguous())
        self._boundaries = []
        def stage(net, i):      # re-labelled / ResNet names (layer1..4), plain timm RegNet (s1..s4) or ConvNeXt (stages.0..3) names (late_fusion.py)
            return getattr(net, "layer%d" % i, None) or getattr(net, "s%d" % i, None) or net.stages[i - 1]

        for i in range(1, 5):
            y = lidar_branch(lambda y=y, i=i: stage(li, i)(y))
            x = stage(im, i)(x)
            if side is not None:
                main.wait_stream(side)          # join: the fusion stage consumes both branches on the main stream
                y.record_stream(main)
            x, y = self._cut((i, 0, 0), (x, y))
            x, y = fuse(i, x, y)
            x, y = self._cut((i, 2, 0), (x, y))
            if side is not None:
                y.record_stream(side)
        x = self._conv(getattr(self, self._reducers[0]), x)
        self._grid_ready = None
        if main is not None:       # the image feature grid is final here: consumers that only need it (the segmentation / depth decoders, forked by
            self._grid_ready = torch.cuda.Event()          # LidarCenterNet.forward) may start beside the LiDAR reducer, the pooling and the FPN below
            self._grid_ready.record(main)
        y = self._conv(getattr(self, self._reducers[1]), y)
        fused = self._pooled(x, y, im, li)
        return self.top_down_nhwc(y), x, fused


    def _pooled(self, x, y, im, li):
        """fused_features = pooled image vector + pooled LiDAR vector (transfuser.py:203-208)."""
        gp_i, gp_l = getattr(im, "global_pool", None), getattr(li, "global_pool", None)
        if isinstance(getattr(gp_i, "norm", None), nn.LayerNorm):     # ConvNeXt: global_pool = the re-labelled head (pool -> LayerNorm((512, 1, 1)))
            return F_.PoolNormFn.apply(x, gp_i.norm, gp_i.norm.weight, gp_i.norm.bias) + F_.PoolNormFn.apply(y, gp_l.norm, gp_l.norm.weight, gp_l.norm.bias)
        return F_.GlobalPoolAddFn.apply(x, y)


class TransfuserBackbone(_FusionBackbone):
    """Multi-scale fusion transformer for image + LiDAR features (transfuser.py:7-211)."""

    def __init__(self, config, image_architecture='resnet34', lidar_architecture='resnet18', use_velocity=True):
        super().__init__()
        chs = self._build_common(config, image_architecture, lidar_architecture)
        for i in range(1, 5):
            setattr(self, "transformer%d" % i, GPT(n_embd=chs[i], n_head=config.n_head, block_exp=config.block_exp, n_layer=config.n_layer,
                                                   img_vert_anchors=config.img_vert_anchors, img_horz_anchors=config.img_horz_anchors,
                                                   lidar_vert_anchors=config.lidar_vert_anchors, lidar_horz_anchors=config.lidar_horz_anchors,
                                                   seq_len=config.seq_len, embd_pdrop=config.embd_pdrop, attn_pdrop=config.attn_pdrop,
                                                   resid_pdrop=config.resid_pdrop, config=config, use_velocity=use_velocity))
        self._build_neck(chs)
        self.register_buffer("dropout_seed", torch.zeros(1, dtype=torch.int32), persistent=False)

    def forward_nhwc(self, image, lidar, velocity, lidar_extra=None, lidar_nhwc=None):
        """image (B,3,H,W) 0..255, lidar (B,2|3,256,256) [+ lidar_extra (B,1,256,256) instead of torch.cat], or
        ``lidar_nhwc`` = the PointPillars canvas; returns NHWC tensors: (p2,p3,p4,p5), image_features_grid, fused_features."""
        def fuse(i, x, y):
            gpt = getattr(self, "transformer%d" % i)
            gpt.seed = self.dropout_seed
            return gpt(x, y, velocity, cut=lambda j, t, l=None, i=i: self._cut((i, 1, j), t, l))
        return self._run(image, lidar, lidar_extra, fuse, lidar_nhwc)

    def forward(self, image, lidar, velocity):
        feats, grid, fused = self.forward_nhwc(image, lidar, velocity)
        return tuple(nchw(p) for p in feats), nchw(grid), fused


class LateFusionBackbone(_FusionBackbone):
    """team_code_transfuser/late_fusion.py:5-111: both RegNetY trunks run without any exchange between the stages (on two
    HIP streams, like the fused backbones), 1x1 reducers 1512 -> 512, FPN on the LiDAR map, fused = gap(image) + gap(lidar) (+ vel_emb).
    Module / parameter names are the reference's (timm models used as they are: ``features.stem.*`` / ``_model.stem.*`` with in_chans
    input channels, no conv1/layerN aliases; ``reduce_channels_conv_*``), so late-fusion checkpoints load.  All three trunk families of
    late_fusion.py:5-33,126-132,155-159: RegNetY, ResNet (timm's own conv1 / bn1 / maxpool / layer1..4; the reducers vanish for a 512-wide image
    trunk - for BOTH branches, the reference tests the image width twice, :45-52) and ConvNeXt (``stem.0/1``, ``stages.i``; the pooled vector
    goes through ``norm_after_pool_*`` = LayerNorm(512), :23-33,92,103); the two branches may use different families."""

    _reducers = ("reduce_channels_conv_image", "reduce_channels_conv_lidar")

    def __init__(self, config, image_architecture='resnet34', lidar_architecture='resnet18', use_velocity=0):
        super().__init__()
        self.config = config
        in_channels = config.num_features[-1] if config.use_point_pillars else 2 * config.lidar_seq_len
        if config.use_target_point_image:
            in_channels += 1
        def plain(architecture, pretrained, in_chans=3):      # timm.create_model + late_fusion.py:129-132,157-160: classifier / pool / head -> empty
            mod = convnext if convnext.is_convnext(architecture) else resnet if resnet.is_resnet(architecture) else regnet
            net = mod.create_model(architecture, pretrained=pretrained, in_chans=in_chans)
            for name in ("fc", "classifier", "global_pool", "head"):
                setattr(net, name, nn.Sequential())
            return net
        self.image_encoder = nn.Module()
        self.image_encoder.normalize = True
        self.image_encoder.features = plain(image_architecture, True)
        self.lidar_encoder = nn.Module()
        self.lidar_encoder._model = plain(lidar_architecture, False, in_channels)
        pf = config.perception_output_features
        self.norm_after_pool_img = nn.LayerNorm((pf,), eps=1e-06) if image_architecture.startswith('convnext') else nn.Sequential()
        self.norm_after_pool_lidar = nn.LayerNorm((pf,), eps=1e-06) if lidar_architecture.startswith('convnext') else nn.Sequential()
        self.use_velocity = use_velocity
        if use_velocity:
            self.vel_emb = nn.Linear(1, pf)
        channel = config.bev_features_chanels
        self.relu = nn.ReLU(inplace=True)
        nf = self.image_encoder.features.num_features
        self.reduce_channels_conv_image = nn.Conv2d(nf, pf, (1, 1)) if nf != pf else nn.Sequential()
        self.reduce_channels_conv_lidar = nn.Conv2d(self.lidar_encoder._model.num_features, pf, (1, 1)) if nf != pf else nn.Sequential()
        self.upsample = nn.Upsample(scale_factor=config.bev_upsample_factor, mode='bilinear', align_corners=False)
        self.up_conv5 = nn.Conv2d(channel, channel, (1, 1))
        self.up_conv4 = nn.Conv2d(channel, channel, (1, 1))
        self.up_conv3 = nn.Conv2d(channel, channel, (1, 1))
        self.c5_conv = nn.Conv2d(pf, channel, (1, 1))
        def stem_of(net, normalize):
            if isinstance(net, convnext.ConvNeXt):
                return _Stem(None, None, normalize, net.stem, ("0", "1"))
            if isinstance(net, resnet.ResNet):
                return _Stem(None, None, normalize, net, ("conv1", "bn1"), isinstance(getattr(net, "maxpool", None), nn.MaxPool2d))
            return _Stem(None, None, normalize, net.stem, ("conv", "bn"))
        self._img_stem = stem_of(self.image_encoder.features, True)
        self._lid_stem = stem_of(self.lidar_encoder._model, False)

    def _pooled(self, x, y, im, li):
        """late_fusion.py:89-106: pool -> flatten -> norm_after_pool (LayerNorm for a ConvNeXt trunk, else nothing) per branch, then the sum."""
        ni, nl = self.norm_after_pool_img, self.norm_after_pool_lidar
        if not isinstance(ni, nn.LayerNorm) and not isinstance(nl, nn.LayerNorm):
            return F_.GlobalPoolAddFn.apply(x, y)
        zero = lambda t: torch.zeros(t.shape[0], 1, 1, t.shape[3], dtype=t.dtype, device=t.device)      # GlobalPoolAddFn pools two maps: pool one
        pi = F_.PoolNormFn.apply(x, ni, ni.weight, ni.bias) if isinstance(ni, nn.LayerNorm) else F_.GlobalPoolAddFn.apply(x, zero(x))
        pl = F_.PoolNormFn.apply(y, nl, nl.weight, nl.bias) if isinstance(nl, nn.LayerNorm) else F_.GlobalPoolAddFn.apply(y, zero(y))
        return pi + pl

    def forward_nhwc(self, image, lidar, velocity, lidar_extra=None, lidar_nhwc=None):
        feats, grid, fused = self._run(image, lidar, lidar_extra, lambda i, x, y: (x, y), lidar_nhwc)
        if self.use_velocity:   # optional flag (train.py default 0): a (B,1) x (1,512) outer product - bookkeeping-size ATen op
            fused = fused + torch.nn.functional.linear(velocity, self.vel_emb.weight, self.vel_emb.bias)
        return feats, grid, fused

    def forward(self, image, lidar, velocity):
        feats, grid, fused = self.forward_nhwc(image, lidar, velocity)
        return tuple(nchw(p) for p in feats), nchw(grid), fused


class latentTFBackbone(TransfuserBackbone):
    """team_code_transfuser/latentTF.py:8-217 (BASELINE config 5): identical modules, but the two LiDAR histogram channels
    are replaced by a fixed (-1..1) positional grid (:132-137); a third input channel (target point) is kept.  The
    reference's LidarEncoder deletes the whole ``stem`` here (latentTF.py:416), so ``_model.stem.bn.*`` keys are absent
    (``_model.bn1.*`` remain) - reproduced for checkpoint compatibility (quirk Q5)."""

    def __init__(self, config, image_architecture='resnet34', lidar_architecture='resnet18', use_velocity=True):
        super().__init__(config, image_architecture, lidar_architecture, use_velocity)
        if hasattr(self.lidar_encoder._model, "stem"):
            del self.lidar_encoder._model.stem
        self._grid = None

    def _pos_grid(self, B, device):
        g = self._grid
        H, W = self.config.lidar_resolution_height, self.config.lidar_resolution_width
        if g is None or g.shape[0] != B or g.device != device:
            x = torch.linspace(-1, 1, self.config.lidar_resolution_width)
            y = torch.linspace(-1, 1, self.config.lidar_resolution_height)
            y_grid, x_grid = torch.meshgrid(x, y, indexing='ij')   # latentTF.py:132-134 (sic: x along rows)
            g = torch.stack((y_grid, x_grid), 0).unsqueeze(0).expand(B, 2, H, W).contiguous().to(device)
            self._grid = g
        return g

    def forward_nhwc(self, image, lidar, velocity, lidar_extra=None):
        grid = self._pos_grid(lidar.shape[0], lidar.device)
        if lidar_extra is None and lidar.shape[1] > 2:
            lidar_extra = lidar[:, 2:].contiguous()
        return super().forward_nhwc(image, grid, velocity, lidar_extra=lidar_extra)


class _Decoder(nn.Module):
    def __init__(self, config, latent_dim, out_ch):
        super().__init__()
        self.config = config
        self.latent_dim = latent_dim
        c1, c2, c3 = config.deconv_channel_num_1, config.deconv_channel_num_2, config.deconv_channel_num_3
        self.deconv1 = nn.Sequential(nn.Conv2d(latent_dim, c1, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c1, c2, 3, 1, 1), nn.ReLU(True))
        self.deconv2 = nn.Sequential(nn.Conv2d(c2, c3, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c3, c3, 3, 1, 1), nn.ReLU(True))
        self.deconv3 = nn.Sequential(nn.Conv2d(c3, c3, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c3, out_ch, 3, 1, 1))

    def forward_nhwc(self, x):
        cfg = self.config
        cv = lambda c, t, r, link=0: F_.ConvFn.apply(t, c.weight, c.bias, r, link)
        x = cv(self.deconv1[2], cv(self.deconv1[0], x, True), True)
        x = F_.UpsampleFn.apply(x, x.shape[1] * int(cfg.deconv_scale_factor_1), x.shape[2] * int(cfg.deconv_scale_factor_1), False)
        x = cv(self.deconv2[2], cv(self.deconv2[0], x, True), True)
        x = F_.UpsampleFn.apply(x, x.shape[1] * int(cfg.deconv_scale_factor_2), x.shape[2] * int(cfg.deconv_scale_factor_2), False)
        # the last ReLU's backward rides in the epilogue of the final layer's input-gradient kernel (one pass over the full-resolution map less)
        return cv(self.deconv3[2], cv(self.deconv3[0], x, True, 1), False, 2)


class SegDecoder(_Decoder):
    """transfuser.py:214-246; returns (B, num_class, H, W) logits."""

    def __init__(self, config, latent_dim=512):
        super().__init__(config, latent_dim, config.num_class)
        self.num_class = config.num_class

    def forward(self, x):
        return nchw(self.forward_nhwc(nhwc(x)))


class DepthDecoder(_Decoder):
    """transfuser.py:249-281; ``forward`` returns sigmoid depth (B, H, W); the training loss consumes the
    pre-sigmoid logits (``forward_nhwc``) so sigmoid + L1 run in one kernel."""

    def __init__(self, config, latent_dim=512):
        super().__init__(config, latent_dim, 1)

    def forward(self, x):
        from . import ops
        logits = self.forward_nhwc(nhwc(x))
        return ops.sigmoid(logits).squeeze(-1)
