"""End-to-end parity of the product model (HIP kernels, emulated on the host) against the CPU oracle on a
tiny RegNetY trunk: 11 losses, every parameter gradient, state_dict key identity, one optimizer step."""
import pytest
import torch

import model_cases as mc


@pytest.fixture(autouse=True)
def _backend(emu_backend):
    yield


def test_tiny_model_losses_and_grads():
    cfg = mc.tiny_config(n_layer=2)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
    batch = mc.small_batch(2, 32, 64, 64, 40)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare(prod, ref, lp, lr, verbose=True)


def test_tiny_model_with_velocity_and_odd_image():
    cfg = mc.tiny_config(n_layer=1)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu", use_velocity=True)
    batch = mc.small_batch(1, 64, 96, 64, 40, seed=3)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare(prod, ref, lp, lr)


def test_engine_two_steps_match_oracle_adamw():
    """Engine (flat arena, fused QKV GEMMs, one-launch AdamW) vs the oracle's Engine.train mirror
    (torch.optim.AdamW) over two iterations: losses of step 2 and all parameters after step 2."""
    from oracle import model_cpu
    from transfuser_amd.train import Engine
    cfg = mc.tiny_config(n_layer=1)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
    eng = Engine(prod, cfg, lr=1e-3)
    assert prod._model.transformer1.blocks[0].attn.fused() is not None, "arena must make k/q/v adjacent"
    from transfuser_amd.model import merged_head_convs
    assert merged_head_convs(prod) is not None, "arena must make the eight 3x3 head convolutions adjacent (one 64 -> 512 convolution)"
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    batch = mc.small_batch(2, 32, 64, 64, 40)
    prod.train(); ref.train()
    for it in range(2):
        tot_p, lp = eng.train_step(batch)
        tot_r, lr_ = model_cpu.train_step(ref, opt, batch, cfg)
        assert abs(float(tot_p) - float(tot_r)) <= 1e-3 * max(1.0, abs(float(tot_r))), (it, float(tot_p), float(tot_r))
    rp = dict(ref.named_parameters())
    # AdamW moves a weight by ~lr * sign(g) when |g| is at round-off level, so two fp32 implementations may differ by
    # up to 2*lr per step on isolated elements; everything else must agree closely.
    diffs = [((p.detach() - rp[n].detach()).abs(), n) for n, p in prod.named_parameters()]
    worst = max((d.max().item(), n) for d, n in diffs)
    assert worst[0] < 4.4e-3, worst
    mean = sum(d.sum().item() for d, _ in diffs) / sum(d.numel() for d, _ in diffs)
    assert mean < 2e-5, mean
    # running BN statistics follow too
    rb = dict(ref.named_buffers())
    for n, b in prod.named_buffers():
        if "running" in n:
            assert (b - rb[n]).abs().max().item() < 1e-3 * max(1.0, rb[n].abs().max().item()), n


def test_latentTF_backbone_matches_oracle():
    """BASELINE config 5 backbone (latentTF.py): positional grid instead of the LiDAR histogram; state_dict keys as the
    reference (no _model.stem.*).  LiDAR 128x128 so that stage-4 BatchNorm sees 32 (not 8) values per channel: with the
    SAME grid in both samples a 2x2 map makes BN-backward a round-off amplifier (degenerate, checked by hand)."""
    cfg = mc.tiny_config(n_layer=1, lidar_res=128)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu", backbone="latentTF")
    assert not any(k.startswith("_model.lidar_encoder._model.stem") for k in prod.state_dict())
    batch = mc.small_batch(2, 32, 64, 128, 40)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


def test_geometric_fusion_backbone_matches_oracle():
    """BASELINE config 4 backbone (geometric_fusion.py): gather kernel G1 + per-stage MLPs, velocity embeddings on,
    quirk Q4 (lidar_conv4 unreachable).  Anchors 2x3 / 3x3 so the fixed x8/x4/x2/x1 factors give a 64x96 image and a 96x96 BEV."""
    cfg = mc.tiny_config(n_layer=1, lidar_res=96)
    cfg.img_vert_anchors, cfg.img_horz_anchors, cfg.lidar_vert_anchors, cfg.lidar_horz_anchors = 2, 3, 3, 3
    cfg.n_embd = 32
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu", backbone="geometric_fusion", use_velocity=True)
    batch = mc.small_batch(2, 64, 96, 96, 40)
    batch.update(mc.geo_points(2, cfg))
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    assert prod._model.lidar_conv4.weight.grad is None and ref._model.lidar_conv4.weight.grad is None
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


def test_geometric_fusion_engine_skips_unreachable_parameters():
    """Engine + geometric fusion: two AdamW steps track the oracle's torch.optim.AdamW, and lidar_conv4 (grad None in the
    reference, quirk Q4) is bit-identical afterwards - no weight decay applied, exactly like torch.optim.AdamW."""
    from oracle import model_cpu
    from transfuser_amd.train import Engine
    cfg = mc.tiny_config(n_layer=1, lidar_res=96)
    cfg.img_vert_anchors, cfg.img_horz_anchors, cfg.lidar_vert_anchors, cfg.lidar_horz_anchors = 2, 3, 3, 3
    cfg.n_embd = 32
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu", backbone="geometric_fusion", use_velocity=True)
    w4 = prod._model.lidar_conv4.weight.detach().clone()
    eng = Engine(prod, cfg, lr=1e-3)
    assert eng.arena.active_numel < eng.arena.numel
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    batch = mc.small_batch(2, 64, 96, 96, 40)
    batch.update(mc.geo_points(2, cfg))
    prod.train(); ref.train()
    for it in range(2):
        tot_p, _ = eng.train_step(batch)
        tot_r, _ = model_cpu.train_step(ref, opt, batch, cfg)
        assert abs(float(tot_p) - float(tot_r)) <= 1e-3 * max(1.0, abs(float(tot_r))), (it, float(tot_p), float(tot_r))
    assert torch.equal(prod._model.lidar_conv4.weight, w4) and torch.equal(ref._model.lidar_conv4.weight, w4)
    rp = dict(ref.named_parameters())
    diffs = [((p.detach() - rp[n].detach()).abs(), n) for n, p in prod.named_parameters()]
    assert max(d.max().item() for d, _ in diffs) < 4.4e-3
    assert sum(d.sum().item() for d, _ in diffs) / sum(d.numel() for d, _ in diffs) < 2e-5


def test_point_pillars_model_matches_oracle():
    """--use_point_pillars 1 (row H2): raw cloud -> pillar ids (scan, no sort) -> point net -> canvas -> differentiable LiDAR stem;
    the point-net parameters receive their gradients through the whole TransFuser model."""
    cfg = mc.tiny_config(n_layer=1, lidar_res=64)
    cfg.use_point_pillars = True
    cfg.min_x, cfg.max_x, cfg.min_y, cfg.max_y = -4, 4, -8, 0          # 8 px/m -> a 64 x 64 canvas
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
    assert prod._model.lidar_encoder._model.conv1.weight.shape[1] == 33
    batch = mc.small_batch(2, 32, 64, 64, 40)
    g = torch.Generator().manual_seed(5)
    batch["lidar"] = torch.stack([torch.rand(2, 3000, generator=g) * 10 - 5, torch.rand(2, 3000, generator=g) * 10 - 9,
                                  torch.rand(2, 3000, generator=g) * 5 - 4, torch.rand(2, 3000, generator=g)], -1)
    batch["num_points"] = torch.tensor([3000, 2500], dtype=torch.int32)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    assert prod.point_pillar_net.point_net.net[0].weight.grad.abs().max() > 0
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg)


def test_point_pillars_static_shapes_match_the_host_read_path():
    """The static-shape mode of the PointPillars front-end (what train.Engine uses under a captured hipGraph: capacity-sized buffers, the kept-point /
    pillar counts never leave the device, BatchNorm1d with a device-side row count): same losses, the same gradients for every parameter and the same
    running statistics as the mode that reads the two counts on the host - with ragged clouds (a sample with dropped points, one with none kept in
    range being impossible here: the second sample keeps fewer)."""
    from transfuser_amd import ops
    cfg = mc.tiny_config(n_layer=1, lidar_res=64)
    cfg.use_point_pillars = True
    cfg.min_x, cfg.max_x, cfg.min_y, cfg.max_y = -4, 4, -8, 0
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
    batch = mc.small_batch(2, 32, 64, 64, 40)
    g = torch.Generator().manual_seed(5)
    batch["lidar"] = torch.stack([torch.rand(2, 3000, generator=g) * 10 - 5, torch.rand(2, 3000, generator=g) * 10 - 9,
                                  torch.rand(2, 3000, generator=g) * 5 - 4, torch.rand(2, 3000, generator=g)], -1)
    batch["num_points"] = torch.tensor([3000, 1700], dtype=torch.int32)
    state = {k: v.clone() for k, v in prod.state_dict().items()}
    res = {}
    calls = {"n": 0}
    orig = ops.bn_rows_dev_fwd

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    ops.bn_rows_dev_fwd = counted
    try:
        for static in (False, True):
            prod.load_state_dict(state)
            prod.point_pillar_net.static_shapes = static
            lp, _ = mc.run_pair(prod, ref, cfg, batch, "cpu")
            res[static] = ({k: float(v) for k, v in lp.items()}, {n: p.grad.clone() for n, p in prod.named_parameters() if p.grad is not None},
                           {n: b.clone() for n, b in prod.named_buffers() if "running" in n})
            assert calls["n"] == (2 if static else 0), calls
    finally:
        ops.bn_rows_dev_fwd = orig
        prod.point_pillar_net.static_shapes = False
    for k, v in res[True][0].items():
        assert abs(v - res[False][0][k]) <= 2e-5 * max(1.0, abs(v)), (k, v, res[False][0][k])
    for n, gr in res[True][1].items():
        g0 = res[False][1][n]
        assert (gr - g0).abs().max().item() <= 5e-4 * max(g0.abs().max().item(), 1e-3), (n, (gr - g0).abs().max().item(), g0.abs().max().item())
    for n, b in res[True][2].items():
        assert torch.allclose(b, res[False][2][n], rtol=1e-5, atol=1e-6), n


def test_forward_ego_and_reference_checkpoint():
    """(a) a reference-style checkpoint (DDP 'module.' prefix, NCHW-contiguous conv weights) loads with strict key
    matching; (b) forward_ego / control_pid reproduce the oracle's inference outputs."""
    cfg = mc.tiny_config(n_layer=1, lidar_res=128)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
    sd = {"module." + k: v.detach().clone().contiguous() for k, v in ref.state_dict().items()}
    with torch.no_grad():
        for v in sd.values():
            if v.dtype.is_floating_point:
                v.add_(0.01)
    ref.load_state_dict({k[7:]: v for k, v in sd.items()}, strict=True)
    missing, unexpected = prod.load_reference_checkpoint(sd)
    assert not missing and not unexpected
    for k, v in prod.state_dict().items():
        assert torch.equal(v.cpu(), sd["module." + k]), k
    assert prod.pred_bev[0].weight.permute(0, 2, 3, 1).is_contiguous()      # still channels-last storage
    batch = mc.small_batch(2, 64, 128, 128, 40)
    mc.check_forward_ego(prod, ref, cfg, batch, "cpu")


def test_edge_cases_empty_labels_single_sample_empty_cloud_partial_fusion():
    """Ragged / empty inputs of the domain: (1) B = 1 with NO boxes at all (avg_factor falls back to 1, mmdet eps paths), (2) PointPillars
    with a sample whose cloud is empty (num_points = 0), (3) geometric fusion with n_scale = 2 (only stages 3 and 4 fuse)."""
    cfg = mc.tiny_config(n_layer=1)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
    batch = mc.small_batch(1, 32, 64, 64, 40)
    batch["label"].zero_()
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    assert float(lp["loss_wh"]) == 0.0
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg, verbose=False)

    cfg = mc.tiny_config(n_layer=1, lidar_res=64)
    cfg.use_point_pillars = True
    cfg.min_x, cfg.max_x, cfg.min_y, cfg.max_y = -4, 4, -8, 0
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
    batch = mc.small_batch(2, 32, 64, 64, 40)
    g = torch.Generator().manual_seed(5)
    batch["lidar"] = torch.stack([torch.rand(2, 500, generator=g) * 10 - 5, torch.rand(2, 500, generator=g) * 10 - 9,
                                  torch.rand(2, 500, generator=g) * 5 - 4, torch.rand(2, 500, generator=g)], -1)
    batch["num_points"] = torch.tensor([0, 400], dtype=torch.int32)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg, verbose=False)

    cfg = mc.tiny_config(n_layer=1, lidar_res=96)
    cfg.img_vert_anchors, cfg.img_horz_anchors, cfg.lidar_vert_anchors, cfg.lidar_horz_anchors = 2, 3, 3, 3
    cfg.n_embd, cfg.n_scale = 32, 2
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu", backbone="geometric_fusion")
    batch = mc.small_batch(2, 64, 96, 96, 40)
    batch.update(mc.geo_points(2, cfg))
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare_vs_fp64(prod, ref, lp, lr, batch, cfg, verbose=False)
    assert prod._model.image_conv1.weight.grad is None and ref._model.image_conv1.weight.grad is None    # stages 1-2 do not fuse


def test_segmented_backward_equals_single_autograd_graph():
    """train.Engine's backward cuts (the multi-GPU overlap path): with cuts after fusion stages 3, 2, 1 - and the finer ones between the
    trunks and the GPT of a stage and BETWEEN THE BLOCKS of a GPT (the split GPTEmbedFn / GPTBlockFn / GPTOutFn nodes) - the step runs as
    separately enqueued backward pieces restarted from detached boundary tensors, over an arena re-laid-out in backward-ready order.  Same
    kernels, same order => gradients, losses and the parameters after two AdamW steps are IDENTICAL to the uncut engine, and the segment
    ranges tile the arena with every parameter inside the range of its own segment."""
    from transfuser_amd.train import Engine, param_key, cut_key
    cfg = mc.tiny_config(n_layer=2)
    batch = mc.small_batch(2, 32, 64, 64, 40)
    res = []
    for cuts in ((), (2,), ((4, 1, 1), (4, 1, 0), (4, 0, 0), 3, (3, 1, 1), (3, 0, 0), 2, 1), Engine.DEFAULT_CUTS):      # (3, 2, 1) is a subset of the third entry
        prod, _ = mc.build_pair(cfg, "regnety_tiny", "cpu")
        prod.train()
        eng = Engine(prod, cfg, lr=1e-3, cuts=cuts)
        keys = sorted((cut_key(c) for c in cuts), reverse=True)
        keys = [k for k in keys if not (k[1] == 1 and k[2] >= cfg.n_layer)]       # DEFAULT_CUTS names Blocks 1..3 of GPT-4: this model has 2
        assert list(eng.cuts) == keys
        assert eng.n_pieces() == len(keys) + 1 and len(eng.arena.segment_ranges) == len(keys) + 1
        out = eng._fwd_bwd(batch)
        grads = {n: p.grad.detach().clone() for n, p in prod.named_parameters()}
        eng.optimizer.step()
        tot2, _ = eng.train_step(batch)
        res.append((float(out[0]), grads, {n: p.detach().clone() for n, p in prod.named_parameters()}, float(tot2)))
        # layout: ranges are contiguous, ordered, cover the active arena; each parameter lies in the range of its segment
        rr = eng.arena.segment_ranges
        assert rr[0][0] == 0 and rr[-1][1] == eng.arena.active_numel and all(a[1] == b[0] for a, b in zip(rr, rr[1:]))
        for n, p, o in eng.arena.layout:
            k = sum(1 for c in keys if param_key(n) <= c)
            assert rr[k][0] <= o and o + p.numel() <= rr[k][1], (n, k, o, rr)
        if len(keys) >= 3:
            assert all(b - a > 0 for a, b in rr), rr
    for other in res[1:]:
        assert other[0] == res[0][0] and other[3] == res[0][3]
        for n in res[0][1]:
            assert torch.equal(other[1][n], res[0][1][n]), n
            assert torch.equal(other[2][n], res[0][2][n]), n


def test_param_stage_names():
    from transfuser_amd.train import param_stage, param_key
    assert param_stage("_model.image_encoder.features.s3.b2.conv1.conv.weight") == 3
    assert param_stage("_model.lidar_encoder._model.layer4.b1.se.fc1.bias") == 4
    assert param_stage("_model.transformer2.blocks.1.attn.key.weight") == 2
    assert param_stage("_model.image_encoder.features.stem.conv.weight") == 0
    assert param_stage("_model.lidar_encoder._model.conv1.weight") == 0 and param_stage("_model.lidar_encoder._model.bn1.bias") == 0
    assert param_stage("head.heatmap_head.0.weight") == 5 and param_stage("_model.up_conv4.weight") == 5 and param_stage("join.0.weight") == 5
    # the PointPillars point net sits UPSTREAM of the LiDAR stem (model.py:736-738): its gradients are produced by the last backward piece
    assert param_stage("point_pillar_net.point_net.net.0.weight") == 0 and param_stage("module.point_pillar_net.point_net.net.4.bias") == 0
    # forward-order keys inside a GPT: embedding < Block 0 < Block 1 < ... < ln_f; trunk stage i < GPT i < trunk stage i + 1
    ks = [param_key("_model.image_encoder.features.s4.b1.conv3.conv.weight"), param_key("_model.transformer4.pos_emb"), param_key("_model.transformer4.vel_emb.weight"),
          param_key("_model.transformer4.blocks.0.ln1.weight"), param_key("_model.transformer4.blocks.3.mlp.2.bias"), param_key("_model.transformer4.ln_f.weight"),
          param_key("_model.change_channel_conv_image.weight")]
    assert ks == sorted(ks) and ks[1] == ks[2] and len(set(ks)) == 6
    assert param_key("_model.transformer3.ln_f.bias") < param_key("_model.lidar_encoder._model.layer4.b1.conv1.conv.weight")


def test_late_fusion_backbone_matches_oracle():
    """SURVEY.md 8f-4 (late_fusion.py:5-111): trunks without exchange, reducers, FPN, pooled sum (+ velocity embedding); state_dict keys
    are the reference's (features.stem.* / _model.stem.* with in_chans, reduce_channels_conv_*), losses + all gradients vs the oracle."""
    cfg = mc.tiny_config(n_layer=1, lidar_res=128)
    for use_vel in (False, True):
        prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu", backbone="late_fusion", use_velocity=use_vel)
        keys = set(prod.state_dict())
        assert "_model.lidar_encoder._model.stem.conv.weight" in keys and "_model.reduce_channels_conv_image.weight" in keys
        assert not any(".layer1." in k or k.endswith("conv1.weight") and "encoder" in k and ".s" not in k for k in keys)
        assert prod._model.lidar_encoder._model.stem.conv.weight.shape[1] == 3
        batch = mc.small_batch(2, 32, 64, 128, 40)
        lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
        # the un-fused LiDAR trunk sees only the sparse histogram: a few ReLU / SE units sit at round-off distance from 0 and flip between
        # two fp32 summation orders (see mc.compare): relative-L2 per tensor with the tolerance of the GPU tiny-model test
        mc.compare(prod, ref, lp, lr, grad_tol=1e-2, metric="l2")


@pytest.mark.parametrize("arch", ["resnet_tiny", "convnext_mini"])
def test_late_fusion_resnet_and_convnext_trunks(arch):
    """late_fusion.py:5-33,126-132,155-159 with the other two trunk families (round-3 verdict: RegNetY only): ResNet under timm's own names with
    in_chans LiDAR stem (no reducers for a trunk as wide as perception_output_features would be; here 64 -> 512 reducers exist), ConvNeXt with
    ``stem.0/1`` / ``stages.i`` names and LayerNorm(512) behind each pooled vector; keys are the reference's, losses + all gradients vs the oracle."""
    cfg = mc.tiny_config(n_layer=1, lidar_res=128)
    prod, ref = mc.build_pair(cfg, arch, "cpu", backbone="late_fusion", use_velocity=True)
    keys = set(prod.state_dict())
    if arch == "convnext_mini":
        assert {"_model.norm_after_pool_img.weight", "_model.norm_after_pool_lidar.bias", "_model.image_encoder.features.stem.0.weight",
                "_model.lidar_encoder._model.stages.1.downsample.1.weight"} <= keys and not any(".head." in k for k in keys)
        assert prod._model.lidar_encoder._model.stem[0].weight.shape[1] == 3
    else:
        assert {"_model.image_encoder.features.conv1.weight", "_model.lidar_encoder._model.layer2.0.downsample.0.weight"} <= keys
        assert not any(k.endswith(".fc.weight") or "norm_after_pool" in k for k in keys) and prod._model.lidar_encoder._model.conv1.weight.shape[1] == 3
    batch = mc.small_batch(2, 32, 64, 128, 40)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare(prod, ref, lp, lr, grad_tol=1e-2, metric="l2")


def test_dropout_paths_match_oracle_with_the_same_masks():
    """Training-mode dropout (embd / attention / both residual sites: fused dropout+residual and softmax+attn_drop kernels, masks regenerated
    in the backward): the oracle's nn.Dropout modules are replaced by modules that apply the PRODUCT's masks (counter RNG keyed by seed and
    site, laid out like the product's tensors), so all 11 losses and every parameter gradient must agree as in the p = 0 tests.  The same
    check runs on the MI355X (tests/test_model_gpu.py), where the bench times exactly this p = 0.1 path."""
    mc.check_dropout_model("cpu")


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_round5_launch_fusions_are_bitwise_neutral(mode):
    mc.check_round5_fusions_bitwise("cpu", mode)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_lowp_storage_modes_tiny_model(mode):
    """16-bit operand STORAGE (ops.lowp_storage(): the GPT linear layers run on cast16 / gemm16_nt) end to end on the tiny model: the 11 losses
    within 3e-2 of the fp32 oracle, the gradient's global cosine with the fp32 oracle's >= 0.98, and the storage path == the in-register
    rounding path of the same precision (TF_STORE16 off) to fp32 summation accuracy - they round the same operands to the same 16-bit values
    (measured 9e-9 of the gradient norm)."""
    from transfuser_amd import ops
    cfg = mc.tiny_config(n_layer=2)
    batch = mc.small_batch(2, 32, 64, 64, 40)
    res = {}
    for store, conv in ((True, False), (False, False), (True, True)):
        prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
        old, oldc = ops._STORE16, ops.STORE16_CONV
        ops._STORE16, ops.STORE16_CONV = store, conv
        ops.set_precision(mode)
        try:
            assert bool(ops.lowp_storage()) == store and bool(ops.lowp_conv()) == (store and conv)
            lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
        finally:
            ops._STORE16, ops.STORE16_CONV = old, oldc
            ops.set_precision("fp32")
        rp = dict(ref.named_parameters())
        names = [n for n, p in prod.named_parameters() if rp[n].grad is not None]
        gp = torch.cat([dict(prod.named_parameters())[n].grad.detach().double().flatten() for n in names])
        gr = torch.cat([rp[n].grad.double().flatten() for n in names])
        res[(store, conv)] = ({k: float(v) for k, v in lp.items()}, gp)
        for k in lr:
            assert abs(float(lp[k]) - float(lr[k])) <= 3e-2 * max(1.0, abs(float(lr[k]))), (mode, store, conv, k, float(lp[k]), float(lr[k]))
        cos = float(torch.dot(gp, gr) / (gp.norm() * gr.norm()))
        assert cos >= 0.98, (mode, store, conv, cos)
    a, b = res[(True, False)], res[(False, False)]
    for k in a[0]:
        assert abs(a[0][k] - b[0][k]) <= 1e-5 * max(1.0, abs(b[0][k])), (k, a[0][k], b[0][k])
    rel = float((a[1] - b[1]).norm() / b[1].norm())
    assert rel <= 1e-5, rel
    # (True, True): additionally the RegNetY bottlenecks' 1x1 convolutions on stored operands (round 5).  The packed-16 kernels add the same rounded
    # products in another order than the in-register kernels (1e-7 on a layer output, check_lowp16_conv_stage pins that at the block level); this
    # network amplifies such round-off ~1e4x into its gradients, so at the model level: losses within 1e-4, gradient within 2e-2 of the other storage path
    c = res[(True, True)]
    for k in c[0]:
        assert abs(c[0][k] - a[0][k]) <= 1e-4 * max(1.0, abs(a[0][k])), (k, c[0][k], a[0][k])
    relc = float((c[1] - a[1]).norm() / a[1].norm())
    assert relc <= 2e-2, relc


@pytest.mark.parametrize("backbone", ["transFuser", "latentTF"])
def test_lowp_model_level_gate_on_the_tiny_twin(backbone):
    """The model-level 16-bit gate of the MI355X suite (mc.check_lowp_full_size: test_lowp_bench_configuration_parity_B10_H256,
    test_fp16_full_size_forward_parity) run on the tiny twin through the emulator: same code, same assertions, same loss-scaled backward, both
    modes against one oracle run, and the error-scales-with-the-mantissa check (fp16 below bf16 on every late feature map)."""
    cfg = mc.tiny_config(n_layer=2)
    batch = mc.small_batch(2, 32, 64, 64, 40)
    out = {k: 1e-1 for k in mc.LOWP_OUTPUTS}
    res = mc.check_lowp_full_size(backbone, 2, 32, {"bf16": dict(loss=3e-2, out=out, cos=0.97, med=0.6),
                                                    "fp16": dict(loss=3e-2, out=out, cos=0.99, med=0.3, loss_scale=1024.0)}, dev="cpu", tiny=(cfg, batch))
    for k in ("fused_features", "image_features_grid", "p2"):
        assert res["fp16"][1][k] * 2.5 <= res["bf16"][1][k], (k, res["fp16"][1][k], res["bf16"][1][k])


@pytest.mark.parametrize("arch", ["resnet_tiny", "resnet_tiny50"])
def test_resnet_trunks_match_oracle(arch):
    """the ResNet trunks the reference's constructors default to (transfuser.py:15; timm names, no re-labelling: 7x7 / s2 stem +
    3x3 / s2 max pool, BasicBlock / Bottleneck, 1x1-stride downsample) through the whole model: 11 losses, every parameter gradient, strict
    state_dict key parity with the oracle (= timm's names)."""
    cfg = mc.tiny_config(n_layer=1)
    prod, ref = mc.build_pair(cfg, arch, "cpu")
    assert any(k.endswith("image_encoder.features.layer2.0.downsample.0.weight") for k in prod.state_dict())
    batch = mc.small_batch(2, 32, 64, 64, 40)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare(prod, ref, lp, lr)


def test_convnext_trunks_match_oracle():
    """the ConvNeXt re-labelling branch (transfuser.py:395-416, 457-471: patchify stem + LayerNorm2d as conv1 / bn1, stages as
    layer1..4, head as global_pool with LayerNorm((512, 1, 1)); timm block = depthwise 7x7 -> LayerNorm -> Linear -> GELU -> Linear -> layer scale)
    through the whole model: 11 losses, every parameter gradient, strict state_dict key parity (incl. the aliased duplicates)."""
    cfg = mc.tiny_config(n_layer=1)
    prod, ref = mc.build_pair(cfg, "convnext_mini", "cpu")
    assert any(k.endswith("image_encoder.features.global_pool.norm.weight") for k in prod.state_dict())
    batch = mc.small_batch(2, 32, 64, 64, 40)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare(prod, ref, lp, lr)


def test_fp16_dynamic_loss_scale_skips_overflowed_steps():
    """fp16 mode (BASELINE configs[4]) under the device-side dynamic loss scale (tf_adamw_dynscale_f32; GradScaler policy - the reference trains
    fp32 and has none): a clean step == the static-scale step; a step whose gradients overflow leaves parameters, moments and the step counter
    untouched and halves the scale; the scale doubles after ``loss_scale_growth_interval`` clean steps.  All on the device (captured-graph safe)."""
    from transfuser_amd import ops
    from transfuser_amd.train import Engine
    cfg = mc.tiny_config(n_layer=1)
    cfg.loss_scale_growth_interval = 2
    batch = mc.small_batch(2, 32, 64, 64, 40)
    try:
        prod, _ = mc.build_pair(cfg, "regnety_tiny", "cpu", seed=5)
        prod.train()
        eng = Engine(prod, cfg, lr=1e-3, precision="fp16")
        assert eng.ls_state is not None and float(eng.ls_state[0]) == 65536.0
        eng.ls_state[0] = 1024.0
        eng.train_step(batch)
        p1 = eng.arena.params.clone()
        assert float(eng.optimizer.state[0]) == 1.0 and eng.ls_state.tolist()[:3] == [1024.0, 1.0, 0.0]
        ref_m, _ = mc.build_pair(cfg, "regnety_tiny", "cpu", seed=5)
        ref_m.train()
        ref = Engine(ref_m, cfg, lr=1e-3, precision="fp16", loss_scale=1024.0)
        ref.train_step(batch)
        assert torch.equal(ref.arena.params, p1), (ref.arena.params - p1).abs().max()      # same seed, same 1 / 1024: bitwise
        m1, v1 = eng.optimizer.exp_avg.clone(), eng.optimizer.exp_avg_sq.clone()
        eng.ls_state[0] = 2.0 ** 120                                                        # every half-precision dy overflows
        eng.train_step(batch)
        assert not torch.isfinite(eng.arena.grads).all()
        assert torch.equal(eng.arena.params, p1) and torch.equal(eng.optimizer.exp_avg, m1) and torch.equal(eng.optimizer.exp_avg_sq, v1)
        assert float(eng.optimizer.state[0]) == 1.0 and eng.ls_state.tolist()[:3] == [2.0 ** 119, 0.0, 0.0]
        eng.ls_state[0] = 512.0
        eng.train_step(batch); eng.train_step(batch)                                        # two clean steps: the scale doubles, the counter restarts
        assert float(eng.optimizer.state[0]) == 3.0 and eng.ls_state.tolist()[:3] == [1024.0, 0.0, 0.0]
        assert not torch.equal(eng.arena.params, p1) and torch.isfinite(eng.arena.params).all()
        # the scale schedule is part of the optimizer state: a resumed run continues it (in place - the backward's seed is a view of the tensor)
        sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.optimizer.state_dict().items()}
        assert sd["ls_state"].tolist()[:3] == [1024.0, 0.0, 0.0]
        seed_ptr = eng._seed_grad.data_ptr()
        eng.ls_state[0] = 4.0; eng.ls_state[1] = 7.0
        eng.optimizer.load_state_dict(sd)
        assert eng.ls_state.tolist()[:3] == [1024.0, 0.0, 0.0] and eng._seed_grad.data_ptr() == seed_ptr and float(eng._seed_grad) == 1024.0
    finally:
        ops.set_precision("fp32")


def test_bottleneck_bn_apply_folded_into_conv2_matches_the_unfused_block():
    mc.check_bn_conv_fold("cpu", (2, 64, 128, 64, 40))


def test_se_blocks_wider_than_the_fused_excitation_kernels_take_the_generic_path(monkeypatch):
    """csrc/se.cpp keeps a sample's squeezed vector in LDS (C <= ops.SE_FUSED_MAX_C); wider bottlenecks (regnety_320: 3712 channels) must fall back to the
    generic colsum + linear path instead of failing - forced here by lowering the limit under the tiny trunk's widths."""
    from transfuser_amd import ops
    monkeypatch.setattr(ops, "SE_FUSED_MAX_C", 8)
    cfg = mc.tiny_config(n_layer=1)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
    batch = mc.small_batch(2, 32, 64, 64, 40)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare(prod, ref, lp, lr)


def test_direct_stride2_grouped_kernels_match_the_engine_path_inside_the_model():
    mc.check_grouped_s2_switch("cpu", (2, 64, 128, 64, 40))


def test_remaining_block_gradients_within_1e3_tiny():
    """The logic of the -m gpu test of the same name at tiny widths: stride-1 bottleneck, both stems, FPN top_down, join MLP + GRU, one
    geometric-fusion stage - outputs, input gradients and every parameter gradient within 1e-3 (max norm) of PyTorch-CPU autograd."""
    mc.check_remaining_blocks("cpu", full=False)


@pytest.mark.parametrize("weights", mc.MERGED_HEAD_WEIGHTS, ids=str)
def test_engine_merged_head_convolution_gradients(weights):
    mc.check_merged_heads("cpu", weights)
