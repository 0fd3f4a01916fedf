"""Whole-model parity on a real MI355X: product LidarCenterNet (libtransfuser_hip.so) vs the CPU oracle."""
import numpy as np
import pytest
import torch

import model_cases as mc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _gpu():
    assert torch.cuda.is_available()
    from transfuser_amd import _lib
    assert not _lib.is_test_backend()
    _lib.load()
    yield
    torch.cuda.synchronize()


def test_tiny_model_losses_and_grads():
    # larger maps than the emulated CPU test: stage-4 is 5x11 / 4x4, so tokens are not replicas of one pixel
    cfg = mc.tiny_config(n_layer=2, lidar_res=128)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cuda")
    batch = mc.small_batch(2, 160, 352, 128, 40)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
    mc.compare(prod, ref, lp, lr, grad_tol=5e-3, verbose=True, metric="l2")


def test_tiny_latentTF_losses_and_grads():
    """BASELINE config 5 backbone (latentTF.py:118-217): positional grid replaces the LiDAR histogram."""
    cfg = mc.tiny_config(n_layer=2, lidar_res=128)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cuda", backbone="latentTF")
    batch = mc.small_batch(2, 160, 352, 128, 40)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


@pytest.mark.parametrize("H,B", [(160, 2), (256, 1)], ids=["H160_reference_resolution", "H256_bench_resolution"])
def test_regnety032_model_losses_and_grads(H, B):
    """The real architecture (RegNetY-3.2GF x2, 4 GPT stages x 4 layers, 168.0 M parameters) at the
    reference resolution (160x704) and at the BASELINE bench resolution (256x704, non-uniform pooling)."""
    from oracle import hist
    from transfuser_amd.data import synthetic_batch
    cfg = mc.full_config()
    prod, ref = mc.build_pair(cfg, "regnety_032", "cuda")
    assert sum(p.numel() for p in prod.parameters()) == 168018327
    batch = synthetic_batch(B, H, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


def test_default_resnet_trunks_reference_resolution():
    """the reference's DEFAULT trunks (transfuser.py:15: 'resnet34' image / here also LiDAR; timm names, no re-labelling) at the
    reference resolution 160x704, B = 2: 11 losses within 1e-3 of the CPU oracle, gradients anchored like the RegNet full-size test."""
    from oracle import hist
    from transfuser_amd.data import synthetic_batch
    cfg = mc.full_config()
    prod, ref = mc.build_pair(cfg, "resnet34", "cuda")
    assert any(k.endswith("image_encoder.features.layer2.0.downsample.0.weight") for k in prod.state_dict())
    batch = synthetic_batch(2, 160, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


def test_convnext_trunks_reference_resolution():
    """``convnext_tiny`` trunks (the re-labelling branch transfuser.py:395-416 / 457-471) at the reference resolution 160x704, B = 2:
    losses within 1e-3 of the CPU oracle, gradients anchored on the fp64 oracle like the RegNet / ResNet full-size tests."""
    from oracle import hist
    from transfuser_amd.data import synthetic_batch
    cfg = mc.full_config()
    prod, ref = mc.build_pair(cfg, "convnext_tiny", "cuda")
    assert tuple(dict(prod.named_parameters())["_model.image_encoder.features.global_pool.norm.weight"].shape) == (512, 1, 1)
    batch = synthetic_batch(2, 160, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


def test_tiny_geometric_fusion_losses_and_grads():
    """BASELINE config 4 backbone (geometric_fusion.py): gather kernel G1, velocity embeddings, quirk Q4."""
    cfg = mc.tiny_config(n_layer=1, lidar_res=96)
    cfg.img_vert_anchors, cfg.img_horz_anchors, cfg.lidar_vert_anchors, cfg.lidar_horz_anchors = 2, 3, 3, 3
    cfg.n_embd = 32
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cuda", backbone="geometric_fusion", use_velocity=True)
    batch = mc.small_batch(2, 64, 96, 96, 40)
    batch.update(mc.geo_points(2, cfg))
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
    assert prod._model.lidar_conv4.weight.grad is None
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


def test_regnety032_geometric_fusion_reference_resolution():
    """Config 4 at its only valid resolution (160x704 + 256x256, default anchors, n_embd 512), real RegNetY-3.2GF trunks:
    48.6 M-parameter backbone; the Engine (flat arena, AdamW skipping the unreachable lidar_conv4) takes steps."""
    from oracle import hist
    from transfuser_amd.data import synthetic_batch
    from transfuser_amd.train import Engine
    cfg = mc.full_config()
    prod, ref = mc.build_pair(cfg, "regnety_032", "cuda", backbone="geometric_fusion")
    assert sum(p.numel() for p in prod._model.parameters()) == 48619940
    batch = synthetic_batch(2, 160, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
    batch = {k: batch[k] for k in Engine.BATCH_KEYS + Engine.GEO_KEYS}
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)
    w4 = prod._model.lidar_conv4.weight.detach().clone()
    eng = Engine(prod, cfg, lr=1e-3)
    bd = {k: v.cuda() for k, v in batch.items()}
    l0 = float(eng.train_step(bd)[0])
    for _ in range(3):
        l1 = float(eng.train_step(bd)[0])
    assert l1 < l0 and torch.equal(prod._model.lidar_conv4.weight, w4)   # untouched: no weight decay on grad-None parameters


def test_tiny_point_pillars_model():
    """Row H2 through the whole model on the MI355X (see tests/test_model_emu.py::test_point_pillars_model_matches_oracle)."""
    cfg = mc.tiny_config(n_layer=1, lidar_res=64)
    cfg.use_point_pillars = True
    cfg.min_x, cfg.max_x, cfg.min_y, cfg.max_y = -4, 4, -8, 0
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cuda")
    batch = mc.small_batch(2, 32, 64, 64, 40)
    g = torch.Generator().manual_seed(5)
    batch["lidar"] = torch.stack([torch.rand(2, 3000, generator=g) * 10 - 5, torch.rand(2, 3000, generator=g) * 10 - 9,
                                  torch.rand(2, 3000, generator=g) * 5 - 4, torch.rand(2, 3000, generator=g)], -1)
    batch["num_points"] = torch.tensor([3000, 2500], dtype=torch.int32)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


def test_full_size_point_pillars_step():
    """B=10 x 40000 raw points (32768 valid) at the bench resolution: pillar ids equal torch.unique's on the same cloud (integer
    exact, checked on the host), losses finite and decreasing over eager AdamW steps."""
    from transfuser_amd import ops
    from transfuser_amd.data import synthetic_batch
    from transfuser_amd.model import LidarCenterNet
    from transfuser_amd.train import Engine
    cfg = mc.full_config()
    cfg.use_point_pillars = True
    torch.manual_seed(0)
    model = LidarCenterNet(cfg, "cuda", "transFuser", "regnety_032", "regnety_032", use_velocity=False)
    mc.randomize(model)
    model.train()
    batch = synthetic_batch(10, 256, 704, seed=0, hist_fn=lambda pts: np.zeros((2, 256, 256), np.float32))
    pts, num = batch["lidar_raw"], batch["num_points"]
    ix = ops.pillar_index(pts.cuda(), num.cuda(), -16, 16, -32, 0, 8)
    rows = []
    for b in range(10):
        p = pts[b, :num[b]]
        keep = (p[:, 0] >= -16) & (p[:, 0] < 16) & (p[:, 1] >= -32) & (p[:, 1] < 0)
        c = ((p[keep][:, [0, 1]] - torch.tensor([-16, -32])) * 8).long()
        rows.append(torch.nn.functional.pad(c, (1, 0), value=b))
    uniq, inverse = torch.cat(rows).unique(return_inverse=True, dim=0)
    assert torch.equal(ix["inv"].cpu().long(), inverse)
    assert torch.equal(ix["cellkey"].cpu().long(), (uniq[:, 0] * ix["GX"] + uniq[:, 1]) * ix["GY"] + uniq[:, 2])
    bd = {k: v.cuda() for k, v in batch.items()}
    bd["lidar"] = bd["lidar_raw"]
    eng = Engine(model, cfg, lr=1e-4)
    first = None
    for it in range(3):
        tot, det = eng.train_step(bd)
        assert all(torch.isfinite(v) for v in det.values())
        first = float(tot) if first is None else first
    assert float(tot) < first


def test_forward_ego_inference_path():
    """SURVEY.md 8f-1 on the MI355X: eval-mode forward_ego (backbone, heads, fused decode_heatmap kernel) vs the oracle."""
    cfg = mc.tiny_config(n_layer=2, lidar_res=128)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cuda")
    batch = mc.small_batch(2, 160, 352, 128, 40)
    mc.check_forward_ego(prod, ref, cfg, batch, "cuda")


@pytest.mark.parametrize("dropout", [0.0, 0.1], ids=["p0", "p0.1"])
def test_engine_graph_replay_matches_eager(dropout):
    """hipGraph-captured training == eager training STEP FOR STEP: same kernels in the same order on the same batch, the capture's warm-up
    iterations leave no trace (parameters / AdamW state / BatchNorm statistics / dropout seed restored), so the loss SEQUENCE, the final
    parameters and the AdamW step counter must coincide (up to the fp32-atomics noise explained at the assertion).  One tiling without
    split-K is pinned; dropout uses the counter-based RNG keyed by the (restored) device seed, so p > 0 is covered as well."""
    from transfuser_amd import ops, transfuser as ptf
    from transfuser_amd.train import Engine
    cfg = mc.tiny_config(n_layer=1, dropout=dropout)
    batch = {k: v.cuda() for k, v in mc.small_batch(2, 32, 64, 64, 40).items()}
    outs, params, steps, rmean = [], [], [], []
    ops.force_plan(64, 64, 16, 1)
    try:
        for use_graph in (False, True):
            ptf.GPT._site_base = 0          # dropout sites are numbered per constructed GPT (class counter): same numbering for both models
            prod, _ = mc.build_pair(cfg, "regnety_tiny", "cuda")
            prod.train()
            eng = Engine(prod, cfg, lr=1e-3, use_graph=use_graph, autotune=False)
            losses = []
            for _ in range(5):
                tot, det = eng.train_step(batch)
                losses.append([float(tot)] + [float(det[k]) for k in cfg.detailed_losses])
            torch.cuda.synchronize()
            outs.append(torch.tensor(losses, dtype=torch.float64))
            params.append(eng.arena.params.detach().clone())
            steps.append(float(eng.optimizer.state[0]))
            rmean.append(torch.cat([b.detach().float().flatten() for n, b in prod.named_buffers() if n.endswith("running_mean")]))
    finally:
        ops.force_plan(0)
    assert steps == [5.0, 5.0], steps                                   # one AdamW update per train_step call, graph or not
    # Not bitwise: the SE squeeze / pooled column sums accumulate with fp32 atomics (ops.colsum(pooled=True)), so two runs of the SAME
    # program differ in the last bits and AdamW amplifies round-off-level gradients to +-lr on isolated weights.  A left-over warm-up
    # update (the r01 bug: 3 updates before the first replay) shifts the whole loss sequence by two steps - several per cent.
    rel_steps = ((outs[0] - outs[1]).abs() / outs[0].abs().clamp_min(1e-3)).max(dim=1).values
    print("  graph vs eager: max relative loss difference per step %s; params max abs diff %.2e, mean %.2e" %
          (["%.1e" % v for v in rel_steps.tolist()], (params[0] - params[1]).abs().max().item(), (params[0] - params[1]).abs().mean().item()))
    # the first replay IS training step 1: a stray warm-up update would show here at the per-cent level (the loss falls ~2 % per step)
    assert rel_steps[0].item() <= 1e-5 and rel_steps[1].item() <= 2e-4, rel_steps
    if dropout == 0:
        assert rel_steps.max().item() <= 2e-4 and (params[0] - params[1]).abs().mean().item() <= 2e-6 and (rmean[0] - rmean[1]).abs().max().item() <= 1e-4
    else:   # with dropout, round-off-level weight-gradient entries (sign flips under AdamW) make the two trajectories drift apart faster
        assert rel_steps.max().item() <= 2e-2 and (params[0] - params[1]).abs().mean().item() <= 2e-4


def test_engine_segmented_graphs_match_single_graph():
    """The multi-GPU step structure on one GPU: backward cut after fusion stages 3, 2, 1 -> four hipGraphs sharing one memory pool (the
    autograd graph built while capturing piece 0 is consumed by the later captures) + an AdamW graph, arena re-laid-out in backward-ready
    order.  Must reproduce the single-graph engine exactly (losses of 4 steps, final parameters by name)."""
    from transfuser_amd import ops, transfuser as ptf
    from transfuser_amd.train import Engine
    cfg = mc.tiny_config(n_layer=2, lidar_res=128, dropout=0.1)
    batch = {k: v.cuda() for k, v in mc.small_batch(2, 160, 352, 128, 40).items()}
    res = []
    ops.force_plan(64, 64, 16, 1)
    try:
        for cuts in ((), (3, 2, 1)):
            ptf.GPT._site_base = 0
            prod, _ = mc.build_pair(cfg, "regnety_tiny", "cuda")
            prod.train()
            eng = Engine(prod, cfg, lr=1e-3, use_graph=True, autotune=False, cuts=cuts)
            losses = [float(eng.train_step(batch)[0]) for _ in range(4)]
            torch.cuda.synchronize()
            assert len(eng._graphs) == len(cuts) + 1
            res.append((losses, {n: p.detach().clone() for n, p in prod.named_parameters()}))
    finally:
        ops.force_plan(0)
    a, b = torch.tensor(res[0][0], dtype=torch.float64), torch.tensor(res[1][0], dtype=torch.float64)
    rel = (a - b).abs() / a.abs()
    assert rel[0].item() <= 1e-5 and rel[1].item() <= 2e-4 and rel.max().item() <= 2e-2, (res[0][0], res[1][0])   # fp32 atomics + dropout: see test_engine_graph_replay_matches_eager
    num = sum((p - res[1][1][n]).abs().sum().item() for n, p in res[0][1].items())
    den = sum(p.numel() for p in res[0][1].values())
    assert num / den <= 2e-4, num / den


def test_bench_configuration_parity_B10_H256():
    """Parity ON the benchmarked configuration (BASELINE configs[1]: B=10, 3x256x704 + 3x256x256, RegNetY-3.2GF x2, 4x4 GPT layers, fp32,
    dropout 0, the shipped tuned plans): the 11 losses and the forward outputs within 1e-3 of the fp32 CPU oracle, and every parameter
    gradient against the oracle's fp32 gradient in relative L2 (see mc.check_full_size_vs_fp32_oracle for the bounds)."""
    mc.check_full_size_vs_fp32_oracle("transFuser", 10, 256)


def test_dropout_paths_match_oracle_on_gpu():
    """p = 0.1 - the bench's setting - on the MI355X against the ORACLE (not against the unfused product kernels): the whole tiny model
    (28 dropout sites: embd_drop, attn_drop inside the softmax kernels, the two fused dropout + residual adds per Block, masks regenerated in
    the backward) with the oracle's nn.Dropout modules applying the product's masks, then one fusion stage at the GPT-4 width
    (C = 1512, T = 174) where outputs, input gradients and every parameter gradient must agree within 1e-3."""
    mc.check_dropout_model("cuda", lidar_res=128, H=160, W=352, grad_tol=5e-3, metric="l2")
    mc.check_dropout_gpt_stage("cuda", C=1512, B=3, n_layer=1, p=0.1)
    mc.check_dropout_gpt_stage("cuda", C=216, B=2, n_layer=2, p=0.1)


def test_latentTF_full_size_B16_H256_parity():
    """BASELINE configs[4] at its own batch size (latentTF.py:118-217, bs=16/GPU, 3x256x704): real RegNetY-3.2GF trunks, shipped plans."""
    mc.check_full_size_vs_fp64("latentTF", 16, 256)


def test_geometric_fusion_full_size_B12_H160_parity():
    """BASELINE configs[3] at its own batch size (geometric_fusion.py:93-288, bs=12/GPU, its only valid resolution 160x704)."""
    mc.check_full_size_vs_fp64("geometric_fusion", 12, 160)


def test_single_block_gradients_within_1e3():
    """north_star's 1e-3 held for GRADIENTS at block level, where no cascade of ~10^8 ReLU kinks amplifies round-off: one RegNetY
    bottleneck (stride 2, SE, downsample) and one GPT Block at the stage-4 width (C=1512, T=174), product kernels vs PyTorch-CPU fp32
    autograd on identical weights - input gradient and every parameter gradient, max-norm relative error <= 1e-3."""
    from transfuser_amd import regnet as preg, functions as fn, transfuser as ptf
    from oracle import regnet as oreg, transfuser_cpu as otf
    torch.manual_seed(0)
    # ---- Y block 216 -> 576, stride 2, group width 24
    ob = oreg.Bottleneck(216, 576, 2, 24, 0.25)
    with torch.no_grad():
        ob.conv3.bn.weight.uniform_(0.5, 1.0)
    pb = preg.Bottleneck(216, 576, 2, 24, 0.25).cuda()
    pb.load_state_dict(ob.state_dict(), strict=True)
    for m in pb.modules():                # 3x3 weights live channels_last on the product path (LidarCenterNet.__init__ does this for the whole model)
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    ob.train(); pb.train()
    x = torch.randn(4, 216, 32, 44)
    xo = x.clone().requires_grad_(True)
    yo = ob(xo)
    dy = torch.randn_like(yo)
    yo.backward(dy)
    xp = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    yp = pb(xp)
    yp.backward(dy.permute(0, 2, 3, 1).contiguous().cuda())

    def rel(a, b):
        return (a.detach().cpu().float() - b.detach()).abs().max().item() / max(b.detach().abs().max().item(), 1e-6)
    assert rel(yp.permute(0, 3, 1, 2), yo) <= 1e-3
    assert rel(xp.grad.permute(0, 3, 1, 2), xo.grad) <= 1e-3, rel(xp.grad.permute(0, 3, 1, 2), xo.grad)
    po = dict(ob.named_parameters())
    for n, p in pb.named_parameters():
        assert rel(p.grad, po[n].grad) <= 1e-3, (n, rel(p.grad, po[n].grad))
    # ---- one fusion stage with ONE GPT Block at the stage-4 width: pool -> tokens -> Block (C=1512, T=174, 4 heads) -> ln_f -> Q1 view ->
    # bilinear up-sample -> residual add, both branches (transfuser.py:150-157,333-366,530-549)
    cfg = mc.full_config()
    cfg.n_layer = 1
    C, B = 1512, 3
    og = otf.GPT(C, cfg, use_velocity=False)
    with torch.no_grad():
        og.pos_emb.normal_(0, 0.05)
        for n, q in og.named_parameters():
            if n.endswith(".bias") or n.endswith("ln1.weight") or n.endswith("ln2.weight") or n.endswith("ln_f.weight"):
                q.add_(torch.randn_like(q) * 0.05)
    pg = ptf.GPT(C, cfg.n_head, cfg.block_exp, 1, cfg.img_vert_anchors, cfg.img_horz_anchors, cfg.lidar_vert_anchors, cfg.lidar_horz_anchors,
                 cfg.seq_len, 0.0, 0.0, 0.0, cfg, use_velocity=False).cuda()
    pg.load_state_dict(og.state_dict(), strict=True)
    og.train(); pg.train()
    pool_i = torch.nn.AdaptiveAvgPool2d((cfg.img_vert_anchors, cfg.img_horz_anchors))
    pool_l = torch.nn.AdaptiveAvgPool2d((cfg.lidar_vert_anchors, cfg.lidar_horz_anchors))
    xi, xl = torch.randn(B, C, 8, 22), torch.randn(B, C, 8, 8)
    xio, xlo = xi.clone().requires_grad_(True), xl.clone().requires_grad_(True)
    fx, fy = og(pool_i(xio), pool_l(xlo), None)
    yi = xio + torch.nn.functional.interpolate(fx, size=(8, 22), mode='bilinear', align_corners=False)
    yl = xlo + torch.nn.functional.interpolate(fy, size=(8, 8), mode='bilinear', align_corners=False)
    di, dl = torch.randn_like(yi), torch.randn_like(yl)
    (yi * di).sum().add((yl * dl).sum()).backward()
    xip = xi.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    xlp = xl.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    yip, ylp = pg(xip, xlp, None)
    torch.autograd.backward([yip, ylp], [di.permute(0, 2, 3, 1).contiguous().cuda(), dl.permute(0, 2, 3, 1).contiguous().cuda()])
    assert rel(yip.permute(0, 3, 1, 2), yi) <= 1e-3 and rel(ylp.permute(0, 3, 1, 2), yl) <= 1e-3
    assert rel(xip.grad.permute(0, 3, 1, 2), xio.grad) <= 1e-3 and rel(xlp.grad.permute(0, 3, 1, 2), xlo.grad) <= 1e-3
    pgo = dict(og.named_parameters())
    for n, p in pg.named_parameters():
        if "attn.key.bias" in n:
            continue                                       # exact gradient is 0 (softmax shift invariance): only round-off on both sides
        assert rel(p.grad, pgo[n].grad) <= 1e-3, (n, rel(p.grad, pgo[n].grad))


def test_f32x3_split_mode_model_parity():
    """tf_set_precision(2) ("f32x3": plain GEMMs as exact bf16x3 splits on the bf16 MFMA) on the REAL architecture (RegNetY-3.2GF x2, 168 M
    parameters, 256x704) held to exactly the bar of the exact-fp32-MFMA path: losses / forward outputs within 1e-3 of the fp64 oracle and every
    gradient tensor as close to fp64 as the reference's own CPU fp32 path is (compare_vs_fp64) - it is an fp32-accurate mode, not a
    reduced-precision one."""
    from oracle import hist
    from transfuser_amd import ops
    from transfuser_amd.data import synthetic_batch
    cfg = mc.full_config()
    prod, ref = mc.build_pair(cfg, "regnety_032", "cuda")
    batch = synthetic_batch(1, 256, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
    ops.set_precision("f32x3")
    try:
        lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
        torch.cuda.synchronize()
    finally:
        ops.set_precision("fp32")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


def test_bf16_mfma_mode_model_parity():
    """BASELINE configs[2] compute mode (tf_set_precision(1): every engine contraction rounds its operands to bf16 and runs on the bf16
    MFMA, fp32 accumulation / storage / master weights) against the fp32 CPU oracle.  Stated tolerance: losses and forward outputs within
    3e-2 relative (bf16 has an 8-bit mantissa: ~4e-3 per operand, accumulated over ~60 layers); gradients: global cosine >= 0.9 with the
    fp32 gradient.  The operation-level test (test_bf16_mfma_mode) pins the exact rounding semantics; this one bounds the end-to-end drift."""
    from transfuser_amd import ops
    cfg = mc.tiny_config(n_layer=2, lidar_res=128)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cuda")
    batch = mc.small_batch(2, 160, 352, 128, 40)
    ops.set_precision("bf16")
    try:
        lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
        torch.cuda.synchronize()
    finally:
        ops.set_precision("fp32")
    for k in lr:
        a, b = float(lp[k].detach()), float(lr[k].detach())
        assert abs(a - b) <= 3e-2 * max(1.0, abs(b)), "loss %s: bf16 %g vs fp32 oracle %g" % (k, a, b)
    rp = dict(ref.named_parameters())
    errs = sorted((p.grad.detach().cpu().double() - rp[n].grad.double()).norm().item() / max(rp[n].grad.double().norm().item(), 1e-12)
                  for n, p in prod.named_parameters() if rp[n].grad is not None and rp[n].grad.norm().item() > 1e-10 and "attn.key.bias" not in n)
    gp = torch.cat([p.grad.detach().cpu().double().flatten() for n, p in prod.named_parameters() if rp[n].grad is not None])
    gr = torch.cat([rp[n].grad.double().flatten() for n, p in prod.named_parameters() if rp[n].grad is not None])
    cos = float(torch.dot(gp, gr) / (gp.norm() * gr.norm()))
    print("  bf16 mode: gradient vs fp32 oracle: global cosine %.4f, per-tensor rel-L2 median %.2e, 90th pct %.2e, max %.2e" %
          (cos, errs[len(errs) // 2], errs[int(len(errs) * 0.9)], errs[-1]))
    # this network amplifies round-off ~1e5x into its gradients (fp32 vs fp64: 1e-2 per tensor, see compare_vs_fp64), so bf16 operand
    # rounding (4e-3) cannot give per-tensor agreement; what low-precision training needs is the descent DIRECTION: global cosine >= 0.9
    assert cos >= 0.9 and errs[len(errs) // 2] <= 0.6, (cos, errs[len(errs) // 2])


def _precision_engine_run(precision, steps, B=10, H=256, dropout=0.1, lr=1e-4):
    from transfuser_amd import ops, transfuser as ptf
    from transfuser_amd.data import synthetic_batch
    from transfuser_amd.model import LidarCenterNet
    from transfuser_amd.train import Engine
    cfg = mc.full_config(dropout=dropout)
    ptf.GPT._site_base = 0
    torch.manual_seed(0)
    model = LidarCenterNet(cfg, "cuda", "transFuser", "regnety_032", "regnety_032", use_velocity=False)
    mc.randomize(model)
    model.train()
    hist_fn = lambda pts: ops.lidar_hist(torch.from_numpy(pts).cuda()[None])[0].cpu().numpy()
    batch = {k: v.cuda() for k, v in synthetic_batch(B, H, 704, seed=0, hist_fn=hist_fn).items()}
    eng = Engine(model, cfg, lr=lr, precision=precision)
    try:
        out = []
        for _ in range(steps):
            tot, det = eng.train_step(batch)
            out.append([float(tot)] + [float(det[k]) for k in cfg.detailed_losses])
        torch.cuda.synchronize()
    finally:
        ops.set_precision("fp32")
    del eng, model
    torch.cuda.empty_cache()
    return torch.tensor(out, dtype=torch.float64)


def test_bf16_full_size_forward_parity():
    """BASELINE configs[2] (bf16 contractions, fp32 master weights / AdamW) on the REAL model at the bench resolution: the 11 losses of the
    168 M-parameter model at 256x704 within 3e-2 (relative, stated tolerance of the bf16 mode: 8-bit mantissa operands, fp32 accumulation,
    ~60 layers) of the fp32 CPU oracle; the gradient keeps a global cosine >= 0.75 with the fp32 oracle's gradient at B = 1 (measured 0.83 on the
    MI355X: at batch 1 the train-mode BatchNorm layers of the stage-4 maps normalise over 64-176 values per channel and amplify the operand
    rounding; the training-trajectory test below is the bar that matters for the mode)."""
    from oracle import hist
    from transfuser_amd import ops
    from transfuser_amd.data import synthetic_batch
    cfg = mc.full_config()
    prod, ref = mc.build_pair(cfg, "regnety_032", "cuda")
    batch = synthetic_batch(1, 256, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
    ops.set_precision("bf16")
    try:
        lp, lr = mc.run_pair(prod, ref, cfg, batch, "cuda")
        torch.cuda.synchronize()
    finally:
        ops.set_precision("fp32")
    for k in lr:
        a, b = float(lp[k].detach()), float(lr[k].detach())
        assert abs(a - b) <= 3e-2 * max(1.0, abs(b)), "loss %s: bf16 %g vs fp32 oracle %g" % (k, a, b)
    rp = dict(ref.named_parameters())
    gp = torch.cat([p.grad.detach().cpu().double().flatten() for n, p in prod.named_parameters() if rp[n].grad is not None])
    gr = torch.cat([rp[n].grad.double().flatten() for n, p in prod.named_parameters() if rp[n].grad is not None])
    cos = float(torch.dot(gp, gr) / (gp.norm() * gr.norm()))
    print("  bf16 full size (B=1, 256x704): max loss deviation %.2e, gradient global cosine vs fp32 oracle %.4f" %
          (max(abs(float(lp[k]) - float(lr[k])) / max(1.0, abs(float(lr[k]))) for k in lr), cos))
    assert cos >= 0.75, cos


def test_bf16_training_trajectory_matches_fp32():
    """BASELINE configs[2]: 20 AdamW steps (B=10, 256x704, dropout 0.1 with identical masks, lr 1e-4) in bf16 against the same 20 steps in
    exact fp32: the total loss stays within 5 % of the fp32 curve at EVERY step and falls by at least 80 % of what the fp32 run gains."""
    c32 = _precision_engine_run("fp32", 20)
    c16 = _precision_engine_run("bf16", 20)
    rel = ((c16[:, 0] - c32[:, 0]).abs() / c32[:, 0].abs())
    print("  20-step trajectory: fp32 loss %.4f -> %.4f, bf16 %.4f -> %.4f; max relative deviation of the total loss %.2e (step %d)" %
          (c32[0, 0], c32[-1, 0], c16[0, 0], c16[-1, 0], rel.max().item(), int(rel.argmax())))
    assert rel.max().item() <= 5e-2, rel.tolist()
    assert (c16[0, 0] - c16[-1, 0]) >= 0.8 * (c32[0, 0] - c32[-1, 0]) > 0, (c32[:, 0].tolist(), c16[:, 0].tolist())


def test_full_size_step_properties():
    """BASELINE config[1] shape (B=10, 3x256x704 + BEV): finite losses, loss decreases over AdamW steps on a
    fixed batch, gradients of the zero-weighted heads are exactly zero (quirk Q6)."""
    from transfuser_amd.data import synthetic_batch
    from transfuser_amd.model import LidarCenterNet
    from transfuser_amd import ops
    from transfuser_amd.train import Engine
    cfg = mc.full_config()
    torch.manual_seed(0)
    model = LidarCenterNet(cfg, "cuda", "transFuser", "regnety_032", "regnety_032", use_velocity=False)
    mc.randomize(model)
    model.train()
    hist_fn = lambda pts: ops.lidar_hist(torch.from_numpy(pts).cuda()[None])[0].cpu().numpy()
    batch = {k: v.cuda() for k, v in synthetic_batch(10, 256, 704, seed=0, hist_fn=hist_fn).items()}
    eng = Engine(model, cfg, lr=1e-4)
    first = None
    for it in range(4):
        tot, det = eng.train_step(batch)
        assert all(torch.isfinite(v) for v in det.values())
        first = float(tot) if first is None else first
    assert float(tot) < first, (first, float(tot))
    g = model.head.velocity_head[2].weight.grad
    assert float(g.abs().max()) == 0.0


def test_train_cli_epochs_schedule_validate_save(tmp_path):
    """transfuser_amd.train.main with the reference's flags (train.py:30-70) on the synthetic dataset: 2 epochs of hipGraph steps, the LR drop
    (train.py:194-199), validation (setting != 'all', train.py:201-202, 321-342), per-epoch checkpoints + args.txt + the loss log; the
    checkpoint loads back into a fresh model and resumes (--load_file / --start_epoch)."""
    import json
    import os
    from transfuser_amd import train
    argv = ["--id", "t", "--logdir", str(tmp_path), "--root_dir", "synthetic:8", "--batch_size", "2", "--epochs", "2", "--schedule_reduce_epoch_01", "1",
            "--setting", "02_05_withheld", "--val_every", "1", "--parallel_training", "0", "--num_workers", "0"]
    tr = train.main(argv)
    d = os.path.join(str(tmp_path), "t")
    assert tr.cur_epoch == 2 and abs(float(tr.eng.optimizer.state[1]) - 1e-5) < 1e-9 and float(tr.eng.optimizer.state[0]) == 8.0   # 2 epochs x 4 steps; lr x0.1 once
    assert json.load(open(os.path.join(d, "args.txt")))["backbone"] == "transFuser"
    rows = [json.loads(l) for l in open(os.path.join(d, "losses.jsonl"))]
    assert any("val_loss_total" in r for r in rows) and sum("loss_total" in r for r in rows) == 2
    assert all(v == v for r in rows for v in r.values())
    for e in (1, 2):
        assert os.path.exists(os.path.join(d, "model_%d.pth" % e)) and os.path.exists(os.path.join(d, "optimizer_%d.pth" % e))
    tr2 = train.main(argv[:-2] + ["--num_workers", "0", "--epochs", "3", "--start_epoch", "2", "--load_file", os.path.join(d, "model_2.pth")])
    assert tr2.cur_epoch == 3 and float(tr2.eng.optimizer.state[0]) == 12.0        # AdamW step counter continued from the checkpoint


def test_engine_overlap_path_on_real_rccl_single_rank():
    """Runs ``_rccl_single_rank_body`` in a CHILD process: the checks below need a real "nccl" (RCCL) process group, and c10d's teardown of it
    (``destroy_process_group`` / the watchdog thread at interpreter exit) has aborted the interpreter at the end of a long pytest session on the
    MI355X box (SIGABRT inside destroy_process_group after every assertion had passed; the same test passes alone).  The child prints a marker
    once all its assertions have passed and leaves with ``os._exit(0)``, so the collective library's shutdown path cannot take the suite down."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_model_gpu as t; t._rccl_single_rank_body(); "
            "print('RCCL-SINGLE-RANK-OK', flush=True); import os; os._exit(0)") % (here, os.path.dirname(here))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert "RCCL-SINGLE-RANK-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def _rccl_single_rank_body():
    """The multi-GPU step on REAL RCCL with the one GPU this box has: a world-size-1 "nccl" process group, backward cut into 4 hipGraph pieces,
    and the reducer forced to behave as on 2 ranks (all-reduce of every segment's arena range on the side stream between the graph replays,
    x 1/2 scale, AdamW graph waiting for the side stream).  With one rank the all-reduce is the identity, so every gradient arrives halved -
    and AdamW is invariant to a constant gradient scale (up to eps = 1e-8): the trajectory must match the plain single-GPU engine.  Also runs
    SyncBatchNorm's collectives (all_gather / all_reduce of the statistics) on RCCL: identical to local BatchNorm at world size 1."""
    import os
    import torch.distributed as dist
    from transfuser_amd import ops, functions as F_, transfuser as ptf
    from transfuser_amd.train import Engine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    from transfuser_amd import _lib
    _lib.load()
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        cfg = mc.tiny_config(n_layer=2, lidar_res=128)
        batch = {k: v.cuda() for k, v in mc.small_batch(2, 160, 352, 128, 40).items()}
        res = []
        ops.force_plan(64, 64, 16, 1)
        try:
            for fake_world in (1, 2):
                ptf.GPT._site_base = 0
                prod, _ = mc.build_pair(cfg, "regnety_tiny", "cuda")
                prod.train()
                eng = Engine(prod, cfg, lr=1e-3, use_graph=True, autotune=False, cuts=(3, 2, 1))
                eng.reducer.world = fake_world
                losses = [float(eng.train_step(batch)[0]) for _ in range(4)]
                torch.cuda.synchronize()
                res.append((losses, {n: p.detach().clone() for n, p in prod.named_parameters()}, eng.arena.grads.abs().sum().item()))
        finally:
            ops.force_plan(0)
        a, b = torch.tensor(res[0][0], dtype=torch.float64), torch.tensor(res[1][0], dtype=torch.float64)
        rel = (a - b).abs() / a.abs()
        assert rel[0].item() <= 1e-5 and rel.max().item() <= 5e-3, (res[0][0], res[1][0])
        assert abs(res[1][2] / res[0][2] - 0.5) < 0.02, (res[0][2], res[1][2])          # the arena really went through all-reduce + scale
        num = sum((p - res[1][1][n]).abs().sum().item() for n, p in res[0][1].items())
        den = sum(p.numel() for p in res[0][1].values())
        assert num / den <= 1e-4, num / den
        # SyncBatchNorm over RCCL (world size 1): same losses as local BatchNorm
        ptf.GPT._site_base = 0
        p1, _ = mc.build_pair(cfg, "regnety_tiny", "cuda")
        p2, _ = mc.build_pair(cfg, "regnety_tiny", "cuda")
        F_.convert_sync_batchnorm(p2)
        p1.train(); p2.train()
        call = lambda m: m(batch['rgb'], batch['lidar'], ego_waypoint=batch['ego_waypoint'], target_point=batch['target_point'],
                           target_point_image=batch['target_point_image'], ego_vel=batch['ego_vel'].reshape(-1, 1), bev=batch['bev'], label=batch['label'],
                           depth=batch['depth'], semantic=batch['semantic'])
        l1, l2 = call(p1), call(p2)
        sum(l2.values()).backward()
        for k in l1:
            assert abs(float(l1[k]) - float(l2[k])) <= 1e-4 * max(1.0, abs(float(l1[k]))), k
        torch.cuda.synchronize()
    finally:
        pass        # no destroy_process_group(): the caller leaves the process right after the success marker (see the test's docstring)
