"""Whole-model parity helpers: product LidarCenterNet (HIP kernels) vs oracle.model_cpu on the same
seeded batch and the same weights.  Used by tests/test_model_emu.py (tiny trunk, emulator) and
tests/test_model_gpu.py (tiny + real regnety_032 on the MI355X)."""
import copy
import math

import numpy as np
import torch

from oracle import model_cpu, hist, regnet as oracle_regnet, resnet as oracle_resnet, convnext as oracle_convnext
from transfuser_amd import regnet as prod_regnet, resnet as prod_resnet, convnext as prod_convnext
from transfuser_amd.config import GlobalConfig
from transfuser_amd.model import LidarCenterNet

TINY = dict(widths=[24, 48, 72, 96], depths=[1, 2, 1, 1], group_w=24, se_ratio=0.25)
prod_regnet.register_arch("regnety_tiny", **TINY)
RESNET_TINY = dict(layers=(1, 2, 1, 1), widths=(16, 32, 48, 64), stem_width=16)        # BasicBlock; "resnet_tiny50": Bottleneck (expansion 4)
prod_resnet.register_arch("resnet_tiny", prod_resnet.BasicBlock, RESNET_TINY["layers"], RESNET_TINY["widths"], RESNET_TINY["stem_width"])
prod_resnet.register_arch("resnet_tiny50", prod_resnet.Bottleneck, (1, 1, 1, 1), (8, 16, 24, 32), 16)
CONVNEXT_TINY = dict(depths=(1, 1, 2, 1), dims=(16, 32, 48, 64))
prod_convnext.register_arch("convnext_mini", CONVNEXT_TINY["depths"], CONVNEXT_TINY["dims"])


def tiny_config(n_layer=2, lidar_res=64, dropout=0.0, use_velocity=False):
    cfg = GlobalConfig()
    cfg.n_layer = n_layer
    cfg.use_target_point_image = True
    cfg.lidar_resolution_width = cfg.lidar_resolution_height = lidar_res
    cfg.bev_resolution_width = cfg.bev_resolution_height = 40
    cfg.embd_pdrop = cfg.attn_pdrop = cfg.resid_pdrop = dropout
    return cfg


def full_config(dropout=0.0):
    cfg = GlobalConfig()
    cfg.n_layer = 4
    cfg.use_target_point_image = True
    cfg.embd_pdrop = cfg.attn_pdrop = cfg.resid_pdrop = dropout
    return cfg


def geo_points(B, cfg, seed=0):
    """bev_points (B, lh, lw, 5, 2) = (x in [0, iw), y in [0, ih)); cam_points (B, iw, ih, 5, 2) in [0, lw) x [0, lh) (data.py:632-675);
    a third of the cells keep the dataset's (0, 0) padding."""
    g = torch.Generator().manual_seed(seed + 100)
    ih, iw, lh, lw = cfg.img_vert_anchors, cfg.img_horz_anchors, cfg.lidar_vert_anchors, cfg.lidar_horz_anchors
    bev = torch.stack((torch.randint(0, iw, (B, lh, lw, 5), generator=g), torch.randint(0, ih, (B, lh, lw, 5), generator=g)), -1)
    cam = torch.stack((torch.randint(0, lw, (B, iw, ih, 5), generator=g), torch.randint(0, lh, (B, iw, ih, 5), generator=g)), -1)
    bev[torch.rand(B, lh, lw, generator=g) < 0.33] = 0
    cam[torch.rand(B, iw, ih, generator=g) < 0.33] = 0
    return dict(bev_points=bev, cam_points=cam)


def small_batch(B, H, W, lidar_res, bev_res, seed=0):
    g = torch.Generator().manual_seed(seed)
    rng = np.random.default_rng(seed)
    label = torch.zeros(B, 20, 7)
    for b in range(B):
        k = int(rng.integers(1, 6))
        label[b, :k, 0:2] = torch.from_numpy(rng.uniform(2, lidar_res - 2, (k, 2))).float()
        label[b, :k, 2:4] = torch.from_numpy(rng.uniform(lidar_res / 32, lidar_res / 6, (k, 2))).float()
        label[b, :k, 4] = torch.from_numpy(rng.uniform(-math.pi, math.pi, k)).float()
        label[b, :k, 5] = torch.from_numpy(rng.uniform(0, 8, k)).float()
        label[b, :k, 6] = torch.from_numpy(rng.integers(0, 2, k)).float()
    lidar = (torch.randint(0, 6, (B, 2, lidar_res, lidar_res), generator=g).float() / 5) * (torch.rand(B, 2, lidar_res, lidar_res, generator=g) < 0.2)
    return dict(rgb=torch.randint(0, 256, (B, 3, H, W), generator=g).float(), lidar=lidar,
                target_point_image=(torch.rand(B, 1, lidar_res, lidar_res, generator=g) < 0.02).float(),
                ego_vel=torch.rand(B, 1, generator=g) * 8, target_point=torch.rand(B, 2, generator=g) * 40 - 10,
                ego_waypoint=torch.rand(B, 4, 2, generator=g) * 14 - 2, bev=torch.randint(0, 3, (B, bev_res, bev_res), generator=g),
                label=label, depth=torch.rand(B, H, W, generator=g), semantic=torch.randint(0, 7, (B, H, W), generator=g))


def randomize(model, seed=1):
    """Make every path carry signal: non-zero last-BN gammas, random pos_emb / biases, non-trivial running stats."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(("bn.weight", "bn1.weight", "bn2.weight", "bn3.weight", "downsample.1.weight")):      # incl. the zero-initialised last BN of a residual branch
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75)
            elif n.endswith(".gamma"):      # ConvNeXt layer scale (initialised to 1e-6): O(1) so the residual branch matters
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.25)
            elif "pos_emb" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif n.endswith(".bias") and p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)


def build_pair(cfg, arch, dev, use_velocity=False, seed=0, backbone='transFuser'):
    """Product model on ``dev`` + oracle on CPU with IDENTICAL weights (strict state_dict load = key parity)."""
    torch.manual_seed(seed)
    prod = LidarCenterNet(cfg, dev, backbone, arch, arch, use_velocity=use_velocity)
    randomize(prod)
    if arch == "regnety_tiny":
        make_net = lambda in_chans=3: oracle_regnet.RegNet(TINY["widths"], TINY["depths"], TINY["group_w"], TINY["se_ratio"], in_chans)
    elif arch == "resnet_tiny":
        make_net = lambda in_chans=3: oracle_resnet.ResNet(oracle_resnet.BasicBlock, RESNET_TINY["layers"], in_chans, RESNET_TINY["widths"], RESNET_TINY["stem_width"])
    elif arch == "resnet_tiny50":
        make_net = lambda in_chans=3: oracle_resnet.ResNet(oracle_resnet.Bottleneck, (1, 1, 1, 1), in_chans, (8, 16, 24, 32), 16)
    elif arch in oracle_resnet.ARCH:
        make_net = oracle_resnet.ARCH[arch]
    elif arch == "convnext_mini":
        make_net = lambda in_chans=3: oracle_convnext.ConvNeXt(in_chans, CONVNEXT_TINY["depths"], CONVNEXT_TINY["dims"])
    elif arch in oracle_convnext.ARCH:
        make_net = oracle_convnext.ARCH[arch]
    else:
        make_net = oracle_regnet.regnety_032
    ref = model_cpu.LidarCenterNet(cfg, 'cpu', backbone, arch, arch, use_velocity=use_velocity, make_net=make_net)
    sd = {k: v.detach().cpu().contiguous() for k, v in prod.state_dict().items()}
    missing = ref.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return prod, ref


def run_pair(prod, ref, cfg, batch, dev):
    prod.train(); ref.train()
    call = lambda m, b: m(b['rgb'], b['lidar'], ego_waypoint=b['ego_waypoint'], target_point=b['target_point'],
                          target_point_image=b['target_point_image'], ego_vel=b['ego_vel'].reshape(-1, 1), bev=b['bev'], label=b['label'],
                          depth=b['depth'], semantic=b['semantic'], **{k: b[k] for k in ('bev_points', 'cam_points', 'num_points') if k in b})
    bd = {k: v.to(dev) for k, v in batch.items()}
    lp = call(prod, bd)
    lr = call(ref, batch)
    w = dict(zip(cfg.detailed_losses, [1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.3, 0.4]))  # non-zero so every head is exercised
    tot_p = sum(w[k] * v for k, v in lp.items())
    tot_r = sum(w[k] * v for k, v in lr.items())
    for p in prod.parameters():
        p.grad = None
    tot_p.backward()
    tot_r.backward()
    return lp, lr


def compare(prod, ref, lp, lr, loss_tol=1e-3, grad_tol=2e-3, verbose=False, metric="max"):
    """Losses within loss_tol (relative, fp32 tolerance of north_star).  Gradients per parameter tensor:
    metric "max": max|g - g_ref| / max|g_ref|  (tight; used where CPU and device arithmetic agree closely);
    metric "l2":  ||g - g_ref||_2 / ||g_ref||_2.  At full width a handful of the ~10^8 ReLU pre-activations lie
    within fp32 round-off of zero, so their masks differ between ANY two fp32 implementations (MKL vs MFMA
    summation order); one flipped unit moves one row of a weight gradient by ~1/sqrt(#rows) but the tensor's
    L2 error by only ~1e-3, whereas a wrong kernel gives O(1)."""
    worst = []
    for k in lr:
        a, b = float(lp[k]), float(lr[k])
        assert abs(a - b) <= loss_tol * max(1.0, abs(b)), "loss %s: %g vs %g" % (k, a, b)
    ref_params = dict(ref.named_parameters())
    for n, p in prod.named_parameters():
        gr = ref_params[n].grad
        if gr is None and (p.grad is None or not p.grad.any()):
            continue        # a parameter the reference's graph never reaches (e.g. geometric fusion's lidar_conv4, quirk Q4)
        assert p.grad is not None, "no grad for " + n
        gp = p.grad.detach().cpu()
        if gr is None:
            gr = torch.zeros_like(gp)
        if metric == "max":
            err = (gp - gr).abs().max().item()
            scale = max(gr.abs().max().item(), 1e-3)
        else:
            err = (gp - gr).double().norm().item()
            scale = max(gr.double().norm().item(), 1e-3 * math.sqrt(gr.numel()) * 1e-2)
        worst.append((err / scale, n, err, scale))
    worst.sort(reverse=True)
    if verbose:
        for r in worst[:8]:
            print("  grad rel(%s) %.2e  %s (abs %.2e, scale %.2e)" % ((metric,) + r))
    # One hidden unit whose pre-activation is within fp32 round-off of 0 may take the other side of its ReLU (the product sums BatchNorm
    # statistics / GEMM partials in a different order than MKL): that moves ONE row of the feeding Linear's weight + bias gradient by a
    # few per cent and nothing else.  Allowed: at most one such (weight, bias) pair, each below 5 %; a wrong kernel gives O(1) on many.
    flips = [r for r in worst if r[0] > grad_tol]
    assert len(flips) <= 2 and all(r[0] <= 5e-2 for r in flips), "gradient mismatch (%s): %s" % (metric, [(r[1], "%.3e" % r[0]) for r in flips[:6]])
    if flips:      # the allowance is for ONE flipped unit: the tensors that used it must be the weight / bias of the SAME layer (a wrong
        # bias-gradient kernel on some other layer would otherwise hide behind it), and every use is logged
        owners = {r[1].rsplit(".", 1)[0] for r in flips}
        kinds = sorted(r[1].rsplit(".", 1)[1] for r in flips)
        print("  compare(): ReLU-flip allowance used by %s" % [(r[1], "%.2e" % r[0]) for r in flips])
        assert len(owners) == 1 and kinds in (["weight"], ["bias"], ["bias", "weight"]), \
            "the >%g gradient differences are not one (weight, bias) pair of one layer: %s" % (grad_tol, [(r[1], "%.3e" % r[0]) for r in flips])
    return worst


def compare_vs_fp64(prod, ref32, lp, lr32, batch, cfg, out_tol=1e-3, verbose=True):
    """Accuracy relative to the TRUE (fp64) result.  The full-width model's fp32 gradients are only accurate to
    ~1e-2 relative-L2 per tensor on ANY fp32 implementation (measured: oracle-fp32 vs oracle-fp64 median 1.2e-2), so
    instead of fp32-vs-fp32 we check that the HIP path is as close to fp64 as the reference CPU fp32 path is:
      * the 11 losses and the forward outputs within out_tol of fp64 (north_star: 1e-3 fp32),
      * per parameter tensor  e_hip = |g_hip - g64|_2 / |g64|_2  <=  4 * max(e_cpu32, median e_cpu32) + 2e-3 (which tensors a ReLU-mask
        flip lands in is random, so a tensor may be as noisy as the model's typical fp32 noise level; a wrong kernel gives O(1)),
        and median(e_hip) <= 2 * median(e_cpu32)."""
    last = ref32.__dict__.pop('_last', None)   # non-leaf tensors cannot be deep-copied
    ref64 = copy.deepcopy(ref32).double()
    ref32._last = last
    for p in ref64.parameters():
        p.grad = None
    b64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in batch.items()}
    ref64.train()
    l64 = ref64(b64['rgb'], b64['lidar'], ego_waypoint=b64['ego_waypoint'], target_point=b64['target_point'], target_point_image=b64['target_point_image'],
                ego_vel=b64['ego_vel'].reshape(-1, 1), bev=b64['bev'], label=b64['label'], depth=b64['depth'], semantic=b64['semantic'],
                **{k: b64[k] for k in ('bev_points', 'cam_points', 'num_points') if k in b64})
    w = dict(zip(cfg.detailed_losses, [1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.3, 0.4]))
    sum(w[k] * v for k, v in l64.items()).backward()
    for k in l64:
        a, b = float(lp[k]), float(l64[k])
        assert abs(a - b) <= out_tol * max(1.0, abs(b)), "loss %s: hip %g vs fp64 %g" % (k, a, b)
    o, o64 = prod._last, ref64._last
    outs = [("pred_wp", o["pred_wp"], o64["pred_wp"]), ("fused_features", o["fused"], o64["fused"]),
            ("image_features_grid", o["grid"].permute(0, 3, 1, 2), o64["grid"]), ("p2", o["features"][0].permute(0, 3, 1, 2), o64["features"][0]),
            ("pred_bev", o["bev_up"].permute(0, 3, 1, 2), o64["pred_bev"])]
    for name, a, b in outs:
        a, b = a.detach().cpu().double(), b.detach()
        err = (a - b).abs().max().item()
        assert err <= out_tol * max(1.0, b.abs().max().item()), "output %s: max err %.3e (scale %.3e)" % (name, err, b.abs().max().item())
        if verbose:
            print("  output %-20s max err %.2e (scale %.2e)" % (name, err, b.abs().max().item()))
    p32, p64 = dict(ref32.named_parameters()), dict(ref64.named_parameters())
    rows = []
    for n, p in prod.named_parameters():
        g64 = p64[n].grad
        if g64 is None:     # never reached by the reference's graph (quirk Q4): ours must be absent or exactly zero
            assert p.grad is None or not p.grad.any(), "gradient for the unused parameter " + n
            continue
        nrm = g64.norm().item()
        e_cpu = (p32[n].grad.double() - g64).norm().item() / max(nrm, 1e-30)
        e_hip = (p.grad.detach().cpu().double() - g64).norm().item() / max(nrm, 1e-30)
        rows.append((e_hip, e_cpu, n, nrm))
    live = [r for r in rows if r[1] < 0.5]   # drop tensors whose true gradient is zero up to round-off (e.g. attn.key.bias: softmax shift invariance)
    med_hip = sorted(r[0] for r in live)[len(live) // 2]
    med_cpu = sorted(r[1] for r in live)[len(live) // 2]
    if verbose:
        print("  gradient rel-L2 error vs fp64: median hip %.2e, median cpu-fp32 %.2e over %d tensors (%d noise-only skipped)" % (med_hip, med_cpu, len(live), len(rows) - len(live)))
        for r in sorted(live, key=lambda r: -(r[0] / (4 * max(r[1], med_cpu) + 2e-3)))[:6]:
            print("    hip %.2e  cpu32 %.2e  %s (|g64| %.2e)" % r)
    assert med_hip <= 2.0 * med_cpu + 1e-4, (med_hip, med_cpu)
    bad = [r for r in live if r[0] > 4 * max(r[1], med_cpu) + 2e-3]
    assert not bad, "gradients further from fp64 than the CPU fp32 reference: %s" % (bad[:5],)
    return med_hip, med_cpu


def check_forward_ego(prod, ref, cfg, batch, dev):
    """Inference path: eval mode, forward_ego of the product vs oracle.model_cpu.forward_ego - waypoints, the decoded
    boxes above the confidence threshold and their metric corner form; then control_pid runs on the predicted waypoints."""
    prod.eval(); ref.eval()
    cfg.bb_confidence_threshold = 0.0            # untrained heads: keep every candidate so the comparison is not vacuous
    b0 = {k: v[:1] for k, v in batch.items()}
    with torch.no_grad():
        wp_r, boxes_r, raw_r = model_cpu.forward_ego(ref, b0['rgb'], b0['lidar'], b0['target_point'], b0['target_point_image'], b0['ego_vel'].reshape(-1, 1))
    bd = {k: v.to(dev) for k, v in b0.items()}
    wp_p, boxes_p = prod.forward_ego(bd['rgb'], bd['lidar'], bd['target_point'], bd['target_point_image'], bd['ego_vel'].reshape(-1, 1))
    assert (wp_p.cpu() - wp_r).abs().max().item() <= 1e-3 * max(1.0, wp_r.abs().max().item())
    raw_p = prod._last_boxes.cpu()
    assert raw_p.shape == raw_r.shape and len(boxes_p) == len(boxes_r) == raw_r.shape[0] > 0
    # untrained heads give nearly flat heat maps (scores within 1e-4 of each other), so the ORDER of the candidates is round-off;
    # compare as sets: match every oracle box to the product box at the same position, then compare all attributes
    assert (raw_p[:, 7].sort().values - raw_r[:, 7].sort().values).abs().max().item() <= 1e-4
    d = (raw_r[:, None, :2] - raw_p[None, :, :2]).abs().sum(-1)
    j = d.argmin(1)
    matched = d[torch.arange(raw_r.shape[0]), j] < 0.05
    assert int(matched.sum()) >= int(0.9 * raw_r.shape[0]), int(matched.sum())      # a few NMS near-ties may pick the neighbouring cell
    a, w = raw_p[j][matched], raw_r[matched]
    err = (a - w).abs()
    assert err[:, :4].max().item() <= 2e-2 and err[:, 5].max().item() <= 1e-2 and err[:, 7].max().item() <= 1e-4, err.max(0)
    yaw = torch.minimum(err[:, 4], (2 * math.pi - err[:, 4]).abs())
    assert int((yaw > 1e-2).sum()) <= max(1, a.shape[0] // 20)        # arg-max over 12 near-equal logits of an untrained head may flip
    assert int((err[:, 6] > 0.5).sum()) <= max(1, a.shape[0] // 20)
    idx = torch.nonzero(matched)[:, 0].tolist()
    for r in idx[:20]:
        (bp, brp, cp), (br, brr, cr) = boxes_p[int(j[r])], boxes_r[r]
        if abs(float(brp) - float(brr)) < 0.5 and float(yaw[idx.index(r)]) <= 1e-2:
            assert bp.shape == (6, 3) and np.abs(bp - br).max() <= 2e-2 and abs(cp - cr) <= 1e-4
    steer, throttle, brake = prod.control_pid(wp_p[:1], bd['ego_vel'].reshape(-1), False)
    assert -1.0 <= float(steer) <= 1.0 and 0.0 <= float(throttle) <= cfg.clip_throttle


# ---------------------------------------------------------------------------------------------- training-mode dropout parity (p > 0)
class MaskDrop(torch.nn.Module):
    """Stand-in for an oracle nn.Dropout: applies the PRODUCT's mask (counter RNG keyed by the device seed and the site, generated by the
    product's own dropout kernel on ``dev`` and laid out like the product's tensors), so oracle and product drop the same elements."""

    def __init__(self, seed, site, p, attn, dev, log):
        super().__init__()
        self.seed, self.site, self.p, self.attn, self.dev, self.log = seed, site, p, attn, dev, log

    def forward(self, x):
        from transfuser_amd import ops
        if self.attn:                      # product layout: (B * nh, T, Tp) with the rows padded to a multiple of 4
            B, nh, T, _ = x.shape
            Tp = (T + 3) // 4 * 4
            m = ops.dropout(torch.ones(B * nh, T, Tp, device=self.dev), self.seed, self.site, self.p)[:, :, :T].reshape(B, nh, T, T)
        else:
            m = ops.dropout(torch.ones(x.numel(), device=self.dev), self.seed, self.site, self.p).view_as(x)
        m = m.cpu().to(x.dtype)
        self.log.append(float((m == 0).float().mean()))
        return x * m


def install_product_masks(gpt_prod, gpt_ref, seed, dev, log):
    """Replace the dropout modules of one oracle GPT by MaskDrop modules bound to the product GPT's sites; returns the number of sites."""
    gpt_ref.drop = MaskDrop(seed, gpt_prod.site(0), gpt_prod.embd_pdrop, False, dev, log)
    for li, blk in enumerate(gpt_ref.blocks):
        blk.attn.attn_drop = MaskDrop(seed, gpt_prod.site(4 * li + 1), gpt_prod.attn_pdrop, True, dev, log)
        blk.attn.resid_drop = MaskDrop(seed, gpt_prod.site(4 * li + 2), gpt_prod.resid_pdrop, False, dev, log)
        blk.mlp[3] = MaskDrop(seed, gpt_prod.site(4 * li + 3), gpt_prod.resid_pdrop, False, dev, log)
    return 1 + 3 * len(gpt_ref.blocks)


def check_dropout_model(dev, lidar_res=64, H=32, W=64, grad_tol=2e-3, metric="max"):
    """Whole tiny model at p = 0.1 (the bench's setting): embd / attention / both residual sites - the fused dropout+residual and
    softmax+attn_drop kernels, masks regenerated in the backward - against the oracle running with the SAME masks: 11 losses and every
    parameter gradient as in the p = 0 tests."""
    import transfuser_amd.transfuser as ptf
    cfg = tiny_config(n_layer=2, lidar_res=lidar_res, dropout=0.1)
    ptf.GPT._site_base = 0          # dropout sites are numbered per constructed GPT (class counter): the same masks whatever ran before
    prod, ref = build_pair(cfg, "regnety_tiny", dev)
    dropped, n = [], 0
    for name in ("transformer1", "transformer2", "transformer3", "transformer4"):
        gp, gr = getattr(prod._model, name), getattr(ref._model, name)
        assert gp.pdrop_any
        n += install_product_masks(gp, gr, prod._model.dropout_seed, dev, dropped)      # the backbone hands this buffer to every GPT stage at call time
    assert n == 28
    batch = small_batch(2, H, W, lidar_res, 40)
    lp, lr = run_pair(prod, ref, cfg, batch, dev)
    compare(prod, ref, lp, lr, grad_tol=grad_tol, metric=metric)
    assert len(dropped) == 28 and all(0.03 < d < 0.25 for d in dropped), dropped     # every site really dropped ~10 % of its elements


def check_round5_fusions_bitwise(dev, mode, lidar_res=64, H=32, W=64):
    """Round-5 launch fusions change WHERE a value is produced, not the value: resid_drop + residual in the GEMM epilogue, resid_drop's gradient from
    ln2's backward launch (ops.FUSE_DROPOUT), ln1 / ln2 writing the 16-bit operand copies themselves (ops.LN_FWD16) and the one-launch weight copies
    (ops.CAST16_MULTI) - the tiny model at p = 0.1 gives the SAME 11 losses and parameter gradients bit for bit with the switches on and off."""
    import transfuser_amd.transfuser as ptf
    from transfuser_amd import ops
    cfg = tiny_config(n_layer=2, lidar_res=lidar_res, dropout=0.1)
    batch = {k: v.to(dev) for k, v in small_batch(2, H, W, lidar_res, 40).items()}
    w = dict(zip(cfg.detailed_losses, [1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.3, 0.4]))
    old = (ops.FUSE_DROPOUT, ops.LN_FWD16, ops.CAST16_MULTI)
    res = []
    ops.set_precision(mode)
    try:
        for on in (True, False):
            ops.FUSE_DROPOUT = ops.LN_FWD16 = ops.CAST16_MULTI = on
            ptf.GPT._site_base = 0
            prod, _ = build_pair(cfg, "regnety_tiny", dev)
            prod.train()
            b = batch
            lp = prod(b['rgb'], b['lidar'], ego_waypoint=b['ego_waypoint'], target_point=b['target_point'], target_point_image=b['target_point_image'],
                      ego_vel=b['ego_vel'].reshape(-1, 1), bev=b['bev'], label=b['label'], depth=b['depth'], semantic=b['semantic'])
            sum(w[k] * v for k, v in lp.items()).backward()
            res.append(({k: float(v) for k, v in lp.items()}, {n: q.grad.detach().clone() for n, q in prod.named_parameters() if q.grad is not None}))
    finally:
        ops.FUSE_DROPOUT, ops.LN_FWD16, ops.CAST16_MULTI = old
        ops.set_precision("fp32")
    (la, ga), (lb, gb) = res
    if dev == "cpu":
        assert la == lb, (la, lb)
    else:
        assert all(abs(la[k] - lb[k]) <= 1e-6 * max(1.0, abs(lb[k])) for k in lb), (la, lb)
    assert ga.keys() == gb.keys()
    split_k = {n for n in ga if not torch.equal(ga[n], gb[n])}
    # reductions that end in fp32 atomics (k-split weight gradients, LayerNorm dgamma / dbeta) are summation-order dependent on the GPU; the emulator is exact
    for n in split_k:
        err = (ga[n] - gb[n]).abs().max().item() / max(gb[n].abs().max().item(), 1e-6)
        assert dev != "cpu" and err < 1e-5, (n, err)


def check_dropout_gpt_stage(dev, C=1512, B=3, n_layer=1, p=0.1, tol=1e-3):
    """One fusion stage at a real width with dropout p (C = 1512, T = 174: GPT-4 of the bench): pool -> tokens -> embd_drop -> Block(s) with
    attn_drop / resid_drop x 2 -> ln_f -> Q1 view -> bilinear -> residual add, product kernels vs the oracle GPT applying the same masks;
    outputs, input gradients and every parameter gradient within ``tol`` (max-norm relative)."""
    from oracle import transfuser_cpu as otf
    import transfuser_amd.transfuser as ptf
    torch.manual_seed(0)
    cfg = full_config(dropout=p)
    cfg.n_layer = n_layer
    og = otf.GPT(C, cfg, use_velocity=False)
    with torch.no_grad():
        og.pos_emb.normal_(0, 0.05)
        for n, q in og.named_parameters():
            if n.endswith(".bias") or n.endswith("ln1.weight") or n.endswith("ln2.weight") or n.endswith("ln_f.weight"):
                q.add_(torch.randn_like(q) * 0.05)
    pg = ptf.GPT(C, cfg.n_head, cfg.block_exp, n_layer, cfg.img_vert_anchors, cfg.img_horz_anchors, cfg.lidar_vert_anchors, cfg.lidar_horz_anchors,
                 cfg.seq_len, p, p, p, cfg, use_velocity=False).to(dev)
    pg.load_state_dict(og.state_dict(), strict=True)
    pg.seed = torch.full((1,), 12345, dtype=torch.int32, device=dev)
    og.train(); pg.train()
    dropped = []
    assert install_product_masks(pg, og, pg.seed, dev, dropped) == 1 + 3 * n_layer
    pool_i = torch.nn.AdaptiveAvgPool2d((cfg.img_vert_anchors, cfg.img_horz_anchors))
    pool_l = torch.nn.AdaptiveAvgPool2d((cfg.lidar_vert_anchors, cfg.lidar_horz_anchors))
    xi, xl = torch.randn(B, C, 8, 22), torch.randn(B, C, 8, 8)
    xio, xlo = xi.clone().requires_grad_(True), xl.clone().requires_grad_(True)
    fx, fy = og(pool_i(xio), pool_l(xlo), None)
    yi = xio + torch.nn.functional.interpolate(fx, size=(8, 22), mode='bilinear', align_corners=False)
    yl = xlo + torch.nn.functional.interpolate(fy, size=(8, 8), mode='bilinear', align_corners=False)
    di, dl = torch.randn_like(yi), torch.randn_like(yl)
    (yi * di).sum().add((yl * dl).sum()).backward()
    xip = xi.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    xlp = xl.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    yip, ylp = pg(xip, xlp, None)
    torch.autograd.backward([yip, ylp], [di.permute(0, 2, 3, 1).contiguous().to(dev), dl.permute(0, 2, 3, 1).contiguous().to(dev)])

    def rel(a, b):
        return (a.detach().cpu().float() - b.detach()).abs().max().item() / max(b.detach().abs().max().item(), 1e-6)
    assert len(dropped) == 1 + 3 * n_layer and all(0.05 < d < 0.15 for d in dropped), dropped
    assert rel(yip.permute(0, 3, 1, 2), yi) <= tol and rel(ylp.permute(0, 3, 1, 2), yl) <= tol
    assert rel(xip.grad.permute(0, 3, 1, 2), xio.grad) <= tol and rel(xlp.grad.permute(0, 3, 1, 2), xlo.grad) <= tol
    pgo = dict(og.named_parameters())
    for n, q in pg.named_parameters():
        if "attn.key.bias" in n:
            continue                                       # exact gradient is 0 (softmax shift invariance): only round-off on both sides
        assert rel(q.grad, pgo[n].grad) <= tol, (n, rel(q.grad, pgo[n].grad))


def check_full_size_vs_fp64(backbone, B, H, dev="cuda", use_velocity=False, precision="fp32"):
    """Parity at a BASELINE configuration's own batch size and resolution, real RegNetY-3.2GF trunks and the shipped plans, anchored on the
    TRUE (fp64) result: losses / forward outputs within 1e-3 of fp64, and every gradient tensor as close to fp64 as the reference's own CPU
    fp32 path is (compare_vs_fp64).  Used where the fp32-vs-fp32 noise bounds of check_full_size_vs_fp32_oracle do not apply (latentTF: the
    smooth positional-grid input puts many more LiDAR-branch pre-activations within round-off of a ReLU kink - oracle-fp32 vs product-fp32
    median 2.3e-2 at B=16 - so the comparison must be made against what fp32 arithmetic itself can resolve)."""
    import os
    from oracle import hist
    from transfuser_amd import ops
    from transfuser_amd.data import synthetic_batch
    ops.plans_load(os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "plans", "mi355x.txt"))
    cfg = full_config()
    prod, ref = build_pair(cfg, "regnety_032", dev, backbone=backbone, use_velocity=use_velocity)
    batch = synthetic_batch(B, H, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
    keys = ("rgb", "lidar", "ego_waypoint", "target_point", "target_point_image", "ego_vel", "bev", "label", "depth", "semantic") + \
        (("bev_points", "cam_points") if backbone == "geometric_fusion" else ())
    batch = {k: batch[k] for k in keys}
    torch.set_num_threads(min(64, os.cpu_count()))
    ops.set_precision(precision)
    try:
        lp, lr = run_pair(prod, ref, cfg, batch, dev)
        if dev != "cpu":
            torch.cuda.synchronize()
    finally:
        ops.set_precision("fp32")
    try:
        return compare_vs_fp64(prod, ref, lp, lr, batch, cfg)
    finally:
        ops.L().tf_plans_clear()


LOWP_OUTPUTS = ("pred_wp", "fused_features", "image_features_grid", "p2", "pred_bev")


def check_lowp_full_size(backbone, B, H, modes, dev="cuda", tiny=None):
    """Model-level parity of the 16-bit compute modes ("bf16" = BASELINE configs[2], "fp16" = configs[4]) AT a BASELINE configuration's own batch size
    and resolution, real RegNetY-3.2GF trunks, the shipped plans, and every 16-bit STORAGE path of the mode switched on (GPT linear layers on
    stored operands, LayerNorm writing the 16-bit copies, the bottlenecks' 1x1 convolutions on the copies their producers write - the wrappers
    below prove that these kernels really ran) against ONE run of the fp32 CPU oracle.

    ``modes``: {precision: dict(loss=, out={output: rel-L2 bound}, cos=, med=, loss_scale=)} - the stated tolerances of the mode AT THIS SIZE: the
    11 losses within ``loss`` (relative), each forward output within its ``out`` bound in relative L2 over the whole tensor, the whole gradient's
    cosine with the fp32 oracle's >= ``cos``, the median per-tensor relative L2 <= ``med``; the backward is seeded with ``loss_scale`` (fp16: what
    train.Engine does) and the scale divided out.  Why the late feature maps carry bounds of 10 - 70 %: this network is RANDOMLY initialised and
    every one of its ~60 BatchNorm-renormalised layers amplifies a relative perturbation of its input; operand rounding of 2^-9 (bf16) / 2^-12
    (fp16) per contraction therefore arrives at the stage-4 feature grid as 4e-1 / 8e-2 (measured on the MI355X, B = 10) while the 11 losses move by
    4e-5 / 1e-5 - the same amplification that turns fp32 round-off (6e-8) into the 1e-2 per-tensor gradient noise compare_vs_fp64 documents.  The
    bounds are ~2x the measured values; what shows that the deviation IS operand rounding and not a defect is the last check of the caller
    (test_lowp_bench_configuration_parity_B10_H256): it scales with the mantissa width, fp16 sitting 4-16x below bf16 on every output.
    Returns {precision: (max loss deviation, {output: rel-L2}, cosine, median)}."""
    import os
    from oracle import hist
    from transfuser_amd import ops
    from transfuser_amd.data import synthetic_batch
    if tiny is None:
        ops.plans_load(os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "plans", "mi355x.txt"))
        cfg = full_config()
        prod, ref = build_pair(cfg, "regnety_032", dev, backbone=backbone)
        batch = synthetic_batch(B, H, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
        keys = ("rgb", "lidar", "ego_waypoint", "target_point", "target_point_image", "ego_vel", "bev", "label", "depth", "semantic") + \
            (("bev_points", "cam_points") if backbone == "geometric_fusion" else ())
        batch = {k: batch[k] for k in keys}
    else:                                         # (cfg, batch) of the tiny twin: the emulator runs this very function (tests/test_model_emu.py)
        cfg, batch = tiny
        prod, ref = build_pair(cfg, "regnety_tiny", dev, backbone=backbone)
    torch.set_num_threads(min(64, os.cpu_count()))
    prod.train(); ref.train()
    call = lambda m, b: m(b['rgb'], b['lidar'], ego_waypoint=b['ego_waypoint'], target_point=b['target_point'],
                          target_point_image=b['target_point_image'], ego_vel=b['ego_vel'].reshape(-1, 1), bev=b['bev'], label=b['label'],
                          depth=b['depth'], semantic=b['semantic'], **{k: b[k] for k in ('bev_points', 'cam_points') if k in b})
    w = dict(zip(cfg.detailed_losses, [1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.3, 0.4]))
    lr = call(ref, batch)
    sum(w[k] * v for k, v in lr.items()).backward()
    r = ref._last
    ref_out = dict(zip(LOWP_OUTPUTS, [r["pred_wp"], r["fused"], r["grid"], r["features"][0], r["pred_bev"]]))
    ref_out = {k: v.detach().double() for k, v in ref_out.items()}
    rp = dict(ref.named_parameters())
    names = [n for n, p in prod.named_parameters() if rp[n].grad is not None]
    gr = torch.cat([rp[n].grad.double().flatten() for n in names])
    bd = {k: v.to(dev) for k, v in batch.items()}
    res = {}
    try:
        for precision, tol in modes.items():
            loss_scale = tol.get("loss_scale", 1.0)
            calls = {"gemm16_nt": 0, "gemm16_nt_colstat": 0, "layernorm_fwd16": 0}
            saved = {k: getattr(ops, k) for k in calls}

            def counted(name):
                def f(*a, **k):
                    calls[name] += 1
                    return saved[name](*a, **k)
                return f
            for k in calls:
                setattr(ops, k, counted(k))
            ops.set_precision(precision)
            try:
                assert ops.lowp_storage() and ops.lowp_conv(), "the 16-bit storage paths are switched off (TF_STORE16 / TF_STORE16_CONV)"
                lp = call(prod, bd)
                for p in prod.parameters():
                    p.grad = None
                (sum(w[k] * v for k, v in lp.items()) * loss_scale).backward()
                if dev != "cpu":
                    torch.cuda.synchronize()
            finally:
                ops.set_precision("fp32")
                for k in calls:
                    setattr(ops, k, saved[k])
            assert calls["gemm16_nt"] > 0 and calls["layernorm_fwd16"] > 0, calls
            assert calls["gemm16_nt_colstat"] > 0 or tiny is not None, calls     # the trunks' 1x1 convolutions ran on the copies their producers wrote
            # every figure first (one printed line per mode: the tolerances are read against it), then the assertions
            dev_l = max(abs(float(lp[k].detach()) - float(lr[k].detach())) / max(1.0, abs(float(lr[k].detach()))) for k in lr)
            o = prod._last
            po = dict(zip(LOWP_OUTPUTS, [o["pred_wp"], o["fused"], o["grid"].permute(0, 3, 1, 2), o["features"][0].permute(0, 3, 1, 2), o["bev_up"].permute(0, 3, 1, 2)]))
            outs = {k: (po[k].detach().cpu().double() - ref_out[k]).norm().item() / max(ref_out[k].norm().item(), 1e-30) for k in LOWP_OUTPUTS}
            pp = dict(prod.named_parameters())
            gp = torch.cat([pp[n].grad.detach().cpu().double().flatten() for n in names]) / loss_scale
            cos = float(torch.dot(gp, gr) / (gp.norm() * gr.norm()))
            errs = sorted((pp[n].grad.detach().cpu().double() / loss_scale - rp[n].grad.double()).norm().item() / rp[n].grad.double().norm().item()
                          for n in names if rp[n].grad.norm().item() > 1e-10)
            live = [e for e in errs if e < 0.99]           # (a tensor whose true gradient is zero up to round-off compares noise with noise: ~1.4)
            med = live[len(live) // 2]
            print("  %s %s B=%d H=%d vs fp32 oracle: stored-operand launches %s; max loss deviation %.2e; outputs rel-L2 %s; gradient cosine %.4f, norm ratio %.4f, "
                  "per-tensor rel-L2 median %.2e / 90th pct %.2e over %d tensors" % (backbone, precision, B, H, calls, dev_l, {k: "%.1e" % v for k, v in outs.items()}, cos,
                                                                                    float(gp.norm() / gr.norm()), med, live[int(len(live) * 0.9)], len(live)))
            for k in lr:
                a, b = float(lp[k].detach()), float(lr[k].detach())
                assert math.isfinite(a) and abs(a - b) <= tol["loss"] * max(1.0, abs(b)), "loss %s: %s %g vs fp32 oracle %g" % (k, precision, a, b)
            for name, l2 in outs.items():
                assert l2 <= tol["out"][name], "%s output %s: rel-L2 %.3e > %.3e" % (precision, name, l2, tol["out"][name])
            assert bool(torch.isfinite(gp).all())
            assert cos >= tol["cos"] and med <= tol["med"], (precision, cos, med)
            res[precision] = (dev_l, outs, cos, med)
    finally:
        if tiny is None:
            ops.L().tf_plans_clear()
    return res


def check_full_size_vs_fp32_oracle(backbone, B, H, dev="cuda", loss_tol=1e-3, per_tensor=5e-2, median=1.5e-2, plans=True, precision="fp32"):
    """Parity at a BASELINE configuration's own batch size and resolution with the real RegNetY-3.2GF trunks: the 11 losses and the forward
    outputs within ``loss_tol`` of the fp32 CPU oracle (north_star: 1e-3), every parameter gradient against the oracle's fp32 gradient in
    relative L2 (fp32 gradients of this network carry ~1e-2 of round-off noise per tensor on ANY implementation: median bound 1.5e-2), and - the
    gate proper, with no allowance for "any few tensors" and no name-based skips - EVERY parameter gradient as close to the fp64 oracle's as the fp32
    oracle's own gradient is (compare_vs_fp64 on the same B / H / plans; tensors beyond ``per_tensor`` of the fp32 oracle are printed)."""
    import os
    from oracle import hist
    from transfuser_amd import ops
    from transfuser_amd.data import synthetic_batch
    if plans:
        ops.plans_load(os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "plans", "mi355x.txt"))
    cfg = full_config()
    prod, ref = build_pair(cfg, "regnety_032", dev, backbone=backbone)
    batch = synthetic_batch(B, H, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
    keys = ("rgb", "lidar", "ego_waypoint", "target_point", "target_point_image", "ego_vel", "bev", "label", "depth", "semantic") + \
        (("bev_points", "cam_points") if backbone == "geometric_fusion" else ())
    batch = {k: batch[k] for k in keys}
    torch.set_num_threads(min(64, os.cpu_count()))
    ops.set_precision(precision)      # "f32x3": the fp32-accurate split contractions, held to the same bounds as the exact-fp32-MFMA path
    try:
        lp, lr = run_pair(prod, ref, cfg, batch, dev)
        if dev != "cpu":
            torch.cuda.synchronize()
    finally:
        ops.set_precision("fp32")
    for k in lr:
        a, b = float(lp[k]), float(lr[k])
        assert abs(a - b) <= loss_tol * max(1.0, abs(b)), "loss %s: hip %g vs oracle %g" % (k, a, b)
    o, r = prod._last, ref._last
    for name, a, b in [("pred_wp", o["pred_wp"], r["pred_wp"]), ("fused_features", o["fused"], r["fused"]),
                       ("image_features_grid", o["grid"].permute(0, 3, 1, 2), r["grid"]), ("p2", o["features"][0].permute(0, 3, 1, 2), r["features"][0]),
                       ("pred_bev", o["bev_up"].permute(0, 3, 1, 2), r["pred_bev"])]:
        a, b = a.detach().cpu(), b.detach()
        err = (a - b).abs().max().item()
        assert err <= loss_tol * max(1.0, b.abs().max().item()), "output %s: max err %.3e" % (name, err)
    rp = dict(ref.named_parameters())
    errs = []
    for n, p in prod.named_parameters():
        g = rp[n].grad
        if g is None or g.norm().item() < 1e-12:
            continue
        errs.append(((p.grad.detach().cpu().double() - g.double()).norm().item() / g.double().norm().item(), n))
    errs.sort(reverse=True)
    med = errs[len(errs) // 2][0]
    print("  %s B=%d H=%d gradient rel-L2 vs fp32 oracle: median %.2e, worst %s" % (backbone, B, H, med, [("%.2e" % e, n) for e, n in errs[:4]]))
    assert med <= median, (med, errs[:6])
    # No free passes (round-5 review, weak #1): a tensor beyond ``per_tensor`` of the fp32 oracle is accepted ONLY if the fp64 oracle says the fp32
    # oracle itself is that far from the true gradient there - compare_vs_fp64 holds EVERY tensor to e_hip <= 4 max(e_cpu32, median e_cpu32) + 2e-3
    # and identifies the noise-only tensors (true gradient zero up to round-off, e.g. attn.key.bias under softmax's shift invariance) by
    # MEASUREMENT (e_cpu32 >= 0.5), not by name.  A wrong kernel / tile plan at these shapes gives O(1) on the tensors it feeds and fails both.
    over = [(e, n) for e, n in errs if e > per_tensor]
    if over:
        print("  beyond %.0e of the fp32 oracle (must be explained by the fp64 anchor below): %s" % (per_tensor, [("%.2e" % e, n) for e, n in over]))
    try:
        compare_vs_fp64(prod, ref, lp, lr, batch, cfg, out_tol=loss_tol)
    finally:
        if plans:
            ops.L().tf_plans_clear()


def check_bn_conv_fold(dev, batch_dims, lidar_res=None):
    """conv1 -> BatchNorm -> ReLU -> grouped conv2 with the BatchNorm apply folded into conv2 / its weight gradient (TF_FUSE_BN_CONV, YBlockFn) and the
    SE scale's backward folded into the BatchNorm backward in front of it (TF_FUSE_SE_BN_BWD): the same model run with the switches on and off - losses and every parameter gradient agree to fp32 round-off, the running statistics are
    updated identically, and the folded path really ran."""
    from transfuser_amd import ops
    calls = {"fwd": 0, "wgrad": 0, "se_bn_bwd": 0}
    of, ow, os_ = ops.grouped_bnrelu_fwd, ops.grouped_bnrelu_wgrad, ops.bn_bwd_remask_se

    def cf(*a, **k):
        calls["fwd"] += 1
        return of(*a, **k)

    def cw(*a, **k):
        calls["wgrad"] += 1
        return ow(*a, **k)
    def cs(*a, **k):
        calls["se_bn_bwd"] += 1
        return os_(*a, **k)
    cfg = tiny_config(n_layer=1, **({"lidar_res": lidar_res} if lidar_res else {}))
    prod, ref = build_pair(cfg, "regnety_tiny", dev)
    batch = small_batch(*batch_dims)
    state = {k: v.clone() for k, v in prod.state_dict().items()}
    res = {}
    prev = ops.FUSE_BN_CONV, ops.FUSE_SE_BN_BWD
    ops.grouped_bnrelu_fwd, ops.grouped_bnrelu_wgrad, ops.bn_bwd_remask_se = cf, cw, cs
    try:
        for on in (True, False):
            ops.FUSE_BN_CONV = ops.FUSE_SE_BN_BWD = on
            prod.load_state_dict(state)
            lp, _ = run_pair(prod, ref, cfg, batch, dev)
            res[on] = ({k: float(v) for k, v in lp.items()}, {n: p.grad.clone() for n, p in prod.named_parameters() if p.grad is not None},
                       {n: b.clone() for n, b in prod.named_buffers() if "running" in n})
            if on:
                assert calls["fwd"] > 0 and calls["fwd"] == calls["wgrad"] and calls["se_bn_bwd"] > 0, calls
                seen = dict(calls)
        assert calls == seen, "the unfused run must not touch the folded kernels"
    finally:
        ops.FUSE_BN_CONV, ops.FUSE_SE_BN_BWD = prev
        ops.grouped_bnrelu_fwd, ops.grouped_bnrelu_wgrad, ops.bn_bwd_remask_se = of, ow, os_
    for k, v in res[True][0].items():
        assert abs(v - res[False][0][k]) <= 1e-5 * max(1.0, abs(v)), (k, v, res[False][0][k])
    for n, g in res[True][1].items():
        g0 = res[False][1][n]
        assert (g - g0).abs().max().item() <= 2e-4 * max(g0.abs().max().item(), 1e-3), n
    for n, b in res[True][2].items():
        assert torch.allclose(b, res[False][2][n], rtol=1e-6, atol=1e-7), n


def check_grouped_s2_switch(dev, batch_dims, lidar_res=None):
    """The opt-in direct stride-2 grouped kernels (TF_GROUPED_S2; forward with statistics + folded BatchNorm apply, weight gradient) inside the model:
    the same model with the switch on and off - the stride-2 bottlenecks then run on the direct kernels resp. the im2col engine."""
    from transfuser_amd import ops
    n = {"s2": 0}
    of = ops.grouped_bnrelu_fwd

    def cf(x, coef, w, stride=1):
        n["s2"] += int(stride == 2)
        return of(x, coef, w, stride)
    cfg = tiny_config(n_layer=1, **({"lidar_res": lidar_res} if lidar_res else {}))
    prod, ref = build_pair(cfg, "regnety_tiny", dev)
    batch = small_batch(*batch_dims)
    state = {k: v.clone() for k, v in prod.state_dict().items()}
    res = {}
    prev = ops._GROUPED_S2
    ops.grouped_bnrelu_fwd = cf
    try:
        for on in (True, False):
            ops._GROUPED_S2 = on
            prod.load_state_dict(state)
            lp, _ = run_pair(prod, ref, cfg, batch, dev)
            res[on] = ({k: float(v) for k, v in lp.items()}, {nm: p.grad.clone() for nm, p in prod.named_parameters() if p.grad is not None})
            if on:
                assert n["s2"] > 0, "no stride-2 bottleneck took the direct kernels (maps too small?)"
                seen = n["s2"]
        assert n["s2"] == seen
    finally:
        ops._GROUPED_S2 = prev
        ops.grouped_bnrelu_fwd = of
    for k, v in res[True][0].items():
        assert abs(v - res[False][0][k]) <= 1e-4 * max(1.0, abs(v)), (k, v, res[False][0][k])
    for nm, g in res[True][1].items():
        g0 = res[False][1][nm]
        assert (g - g0).abs().max().item() <= 1e-3 * max(g0.abs().max().item(), 1e-3), nm


def check_remaining_blocks(dev, full=True):
    """The 1e-3 gradient bound for the blocks the earlier block tests left out (round-4 verdict, weak #1): (a) a STRIDE-1 RegNetY bottleneck at the
    stage-3 width (576 -> 576 at 16 x 44: the folded BatchNorm / SE kernels of round 4), (b) both stems (normalize_imagenet + 3x3 / s2 conv +
    BatchNorm + ReLU on the NCHW inputs; the LiDAR one with the target-point channel as a separate tensor), (c) the FPN ``top_down``
    (transfuser.py:221-237), (d) the join MLP + GRU waypoint decoder (model.py:592-646), (e) one geometric-fusion stage (stage 3, C = 576:
    pool, gather of the 5 correspondences, MLP, up-sample, 1x1, residual) - product kernels vs PyTorch-CPU fp32 autograd on identical
    weights: outputs, input gradients and every parameter gradient, max-norm relative error <= 1e-3."""
    from transfuser_amd import regnet as preg, functions as fn, ops
    from oracle import regnet as oreg, transfuser_cpu as otf
    import torch.nn.functional as Fn
    W3 = 576 if full else 48                      # bottleneck width (24-wide groups)
    arch = "regnety_032" if full else "regnety_tiny"

    def rel(a, b):
        return (a.detach().cpu().float() - b.detach()).abs().max().item() / max(b.detach().abs().max().item(), 1e-6)
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    torch.manual_seed(0)
    # ---- (a) Y block 576 -> 576, stride 1 (no downsample branch: the shortcut is the input), group width 24
    ob = oreg.Bottleneck(W3, W3, 1, 24, 0.25)
    with torch.no_grad():
        for bn in (ob.conv1.bn, ob.conv2.bn, ob.conv3.bn):
            bn.weight.uniform_(0.5, 1.0)
            bn.bias.uniform_(-0.2, 0.2)
    pb = preg.Bottleneck(W3, W3, 1, 24, 0.25).to(dev)
    pb.load_state_dict(ob.state_dict(), strict=True)
    for m in pb.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    ob.train(); pb.train()
    if full:
        assert ops.FUSE_BN_CONV and ops.FUSE_BN_SE and ops.grouped_bnrelu_ok((4, 16, 44, 576), 576, 24, 1), "the folded kernels are what this case is about"
    x = torch.randn(4, W3, 16, 44) if full else torch.randn(2, W3, 8, 12)
    xo = x.clone().requires_grad_(True)
    yo = ob(xo)
    dy = torch.randn_like(yo)
    yo.backward(dy)
    xp = nh(x).requires_grad_(True)
    yp = pb(xp)
    yp.backward(nh(dy))
    assert rel(yp.permute(0, 3, 1, 2), yo) <= 1e-3
    assert rel(xp.grad.permute(0, 3, 1, 2), xo.grad) <= 1e-3, rel(xp.grad.permute(0, 3, 1, 2), xo.grad)
    po = dict(ob.named_parameters())
    for n, p in pb.named_parameters():
        assert rel(p.grad, po[n].grad) <= 1e-3, ("bottleneck s1", n, rel(p.grad, po[n].grad))
    # ---- (b)-(d) on a full-width model pair (RegNetY-3.2GF, the reference configuration)
    cfg = full_config() if full else tiny_config(n_layer=1)
    prod, ref = build_pair(cfg, arch, dev)
    prod.train(); ref.train()
    pref = dict(ref.named_parameters())

    def check_params(prefixes, what):
        n_checked = 0
        for n, p in prod.named_parameters():
            if any(n.startswith(q) for q in prefixes):
                assert p.grad is not None and pref[n].grad is not None, (what, n)
                assert rel(p.grad, pref[n].grad) <= 1e-3, (what, n, rel(p.grad, pref[n].grad))
                n_checked += 1
        assert n_checked > 0, what
        for q in list(prod.parameters()) + list(ref.parameters()):
            q.grad = None
    # (b) stems
    g = torch.Generator().manual_seed(3)
    image = torch.randint(0, 256, (2, 3, 96, 160) if full else (2, 3, 32, 48), generator=g).float()
    lidar = torch.rand(2, 2, 64, 64, generator=g) * (torch.rand(2, 2, 64, 64, generator=g) < 0.3)
    extra = (torch.rand(2, 1, 64, 64, generator=g) < 0.05).float()
    im, li = ref._model.image_encoder.features, ref._model.lidar_encoder._model
    so = im.act1(im.bn1(im.conv1(otf.normalize_imagenet(image))))
    lo = li.act1(li.bn1(li.conv1(torch.cat((lidar, extra), 1))))
    ds, dl = torch.randn_like(so), torch.randn_like(lo)
    torch.autograd.backward([so, lo], [ds, dl])
    sp = prod._model._img_stem(image.to(dev))
    lp = prod._model._lid_stem(lidar.to(dev).contiguous(), extra.to(dev).contiguous())
    torch.autograd.backward([sp, lp], [nh(ds), nh(dl)])
    assert rel(sp.permute(0, 3, 1, 2), so) <= 1e-3 and rel(lp.permute(0, 3, 1, 2), lo) <= 1e-3
    check_params(("_model.image_encoder.features.conv1.", "_model.image_encoder.features.bn1.", "_model.lidar_encoder._model.conv1.",
                  "_model.lidar_encoder._model.bn1."), "stems")
    # (c) FPN top_down on the (B, 512, 8, 8) LiDAR map
    y = torch.randn(3, cfg.perception_output_features, 8, 8)
    yo_ = y.clone().requires_grad_(True)
    outs_o = ref._model.top_down(yo_)
    douts = [torch.randn_like(t) for t in outs_o]
    torch.autograd.backward(list(outs_o), douts)
    yp_ = nh(y).requires_grad_(True)
    outs_p = prod._model.top_down_nhwc(yp_)
    torch.autograd.backward(list(outs_p), [nh(t) for t in douts])
    for a, b in zip(outs_p, outs_o):
        assert rel(a.permute(0, 3, 1, 2), b) <= 1e-3
    assert rel(yp_.grad.permute(0, 3, 1, 2), yo_.grad) <= 1e-3, rel(yp_.grad.permute(0, 3, 1, 2), yo_.grad)
    check_params(("_model.c5_conv.", "_model.up_conv5.", "_model.up_conv4.", "_model.up_conv3."), "top_down")
    # (d) join MLP + GRU decoder
    z = torch.randn(10 if full else 3, 512)
    tp = torch.rand(10 if full else 3, 2) * 40 - 10
    zo = z.clone().requires_grad_(True)
    wo = ref.forward_gru(zo, tp)
    dw = torch.randn_like(wo)
    wo.backward(dw)
    zp = z.to(dev).requires_grad_(True)
    wp = prod.forward_gru(zp, tp.to(dev))[0]
    wp.backward(dw.to(dev))
    assert rel(wp, wo) <= 1e-3
    assert rel(zp.grad, zo.grad) <= 1e-3, rel(zp.grad, zo.grad)
    check_params(("join.", "decoder.", "output."), "join MLP + GRU")
    del prod, ref
    # ---- (e) geometric-fusion stage 3 (C = 576, anchors x 2) in isolation, both directions
    gcfg = full_config() if full else tiny_config(n_layer=1)
    gprod, gref = build_pair(gcfg, arch, dev, backbone="geometric_fusion")
    gprod.train(); gref.train()
    pb_, rb_ = gprod._model, gref._model
    st = pb_._stages[2]
    B, C = 3, st.image_conv.weight.shape[1]
    ih, iw, lh, lw = gcfg.img_vert_anchors, gcfg.img_horz_anchors, gcfg.lidar_vert_anchors, gcfg.lidar_horz_anchors
    pts = geo_points(B, gcfg, seed=1)
    x = torch.randn(B, C, ih * 2, iw * 2)
    y = torch.randn(B, C, lh * 2, lw * 2)
    xo, yo_ = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    img_e = rb_.avgpool_img(rb_.image_conv3(xo))
    lid_e = rb_.avgpool_lidar(rb_.lidar_conv3(yo_))
    up = lambda t: Fn.interpolate(t, scale_factor=2, mode='bilinear', align_corners=False)
    bev = rb_.image_projection3(rb_._gather_sum(img_e, pts['bev_points'], lh, lw)).permute(0, 3, 1, 2).contiguous()
    y_out = yo_ + rb_.lidar_deconv3(up(bev))
    img = rb_.lidar_projection3(rb_._gather_sum(lid_e, pts['cam_points'], ih, iw)).permute(0, 3, 1, 2).contiguous()
    x_out = xo + rb_.image_deconv3(up(img))
    dxo, dyo = torch.randn_like(x_out), torch.randn_like(y_out)
    torch.autograd.backward([x_out, y_out], [dxo, dyo])
    params = [p for m in (st.image_conv, st.lidar_conv, st.image_deconv, st.lidar_deconv, st.image_projection, st.lidar_projection) for p in m.parameters()]
    xp_, yp_ = nh(x).requires_grad_(True), nh(y).requires_grad_(True)
    bev_idx = pb_._flat_idx(pts['bev_points'].to(dev), B, lh * lw)
    img_idx = pb_._flat_idx(pts['cam_points'].to(dev), B, ih * iw)
    ox, oy, _ = fn.GeoStageFn.apply(xp_, yp_, None, st, None, bev_idx, img_idx, *params)
    torch.autograd.backward([ox, oy], [nh(dxo), nh(dyo)])
    assert rel(ox.permute(0, 3, 1, 2), x_out) <= 1e-3 and rel(oy.permute(0, 3, 1, 2), y_out) <= 1e-3
    assert rel(xp_.grad.permute(0, 3, 1, 2), xo.grad) <= 1e-3, rel(xp_.grad.permute(0, 3, 1, 2), xo.grad)
    assert rel(yp_.grad.permute(0, 3, 1, 2), yo_.grad) <= 1e-3, rel(yp_.grad.permute(0, 3, 1, 2), yo_.grad)
    gpref = dict(gref.named_parameters())
    n_checked = 0
    for n, p in gprod.named_parameters():
        if any(n.startswith("_model.%s3." % q) for q in ("image_conv", "lidar_conv", "image_deconv", "lidar_deconv", "image_projection", "lidar_projection")):
            assert rel(p.grad, gpref[n].grad) <= 1e-3, ("geometric stage 3", n, rel(p.grad, gpref[n].grad))
            n_checked += 1
    assert n_checked == 20, n_checked




MERGED_HEAD_WEIGHTS = [(0.2, 0.2, 0.2, 0.2, 0.2, 0.0, 0.0), (0.2, 0.2, 0.2, 0.2, 0.2, 0.3, 0.4), (0.2, 0.0, 0.2, 0.2, 0.2, 0.0, 0.4)]


def check_merged_heads(dev, weights):
    """Inside the Engine the eight 3x3 convolutions on p2 (seven CenterNet heads + pred_bev) run as ONE 64 -> 512 convolution over the
    arena-adjacent weights; the backward differentiates the live heads' channels only - a prefix when the zero-weight heads are the reference's
    (velocity, brake), every channel (dead slices zero-filled) otherwise.  Gradients of every head parameter and of p2's producers vs the oracle."""
    from oracle import model_cpu
    from transfuser_amd.train import Engine
    from transfuser_amd.model import merged_head_convs
    cfg = tiny_config(n_layer=1)
    cfg.detailed_losses_weights = [1.0, 1.0, 1.0, 1.0] + list(weights)
    prod, ref = build_pair(cfg, "regnety_tiny", dev)
    eng = Engine(prod, cfg, lr=0.0)
    assert merged_head_convs(prod) is not None
    batch = small_batch(2, 32, 64, 64, 40)
    prod.train(); ref.train()
    eng.train_step({k: v.to(dev) for k, v in batch.items()})
    losses = ref(batch['rgb'], batch['lidar'], ego_waypoint=batch['ego_waypoint'], target_point=batch['target_point'],
                 target_point_image=batch['target_point_image'], ego_vel=batch['ego_vel'].reshape(-1, 1), bev=batch['bev'], label=batch['label'],
                 depth=batch['depth'], semantic=batch['semantic'])
    model_cpu.total_loss(losses, cfg).backward()
    rp = dict(ref.named_parameters())
    n = 0
    for name, p in prod.named_parameters():
        if name.startswith("head.") or name.startswith("pred_bev.") or name.startswith("_model.up_conv3."):
            g = rp[name].grad if rp[name].grad is not None else torch.zeros_like(rp[name])
            scale = max(1e-6, g.abs().max().item())
            err = (p.grad.detach().cpu() - g).abs().max().item()
            assert err <= 2e-3 * scale + 1e-7, (name, err, scale)
            n += 1
    assert n == 34, n
