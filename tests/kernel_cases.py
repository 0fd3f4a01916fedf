"""Shared op-level parity checks: HIP kernel (through transfuser_amd.ops) vs a plain PyTorch fp32
reference / the CPU oracle on the same seeded inputs.  ``dev`` is "cpu" for the emulated build
(tests/test_kernels_emu.py) and "cuda" for the real MI355X run (tests/test_kernels_gpu.py).
Tolerances: fp32, 1e-3 per north_star (most ops are far tighter); integer outputs exact."""
import ctypes
import math

import numpy as np
import torch
import torch.nn.functional as F

from transfuser_amd import ops

TOL = 1e-3


def close(a, b, tol=TOL, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.abs().max().item() if b.numel() else 1.0)
    assert err <= tol * scale, "%s: max err %.3e (scale %.3e)" % (what, err, scale)


def c64(t):
    """The REFERENCE side of an op-level check: a float64 CPU copy - so that the comparison is against PyTorch's CPU arithmetic, never against the vendor
    libraries (rocBLAS / MIOpen) on the GPU under test (round-5 review, weak #4)."""
    return t.detach().to("cpu", torch.float64)


def R(*shape, seed=None, dev="cpu", scale=1.0):
    g = torch.Generator().manual_seed(abs(hash((shape, -1 if seed is None else seed))) % (2 ** 31))   # hash(None) is address-based: per-process data
    return (torch.randn(*shape, generator=g) * scale).to(dev)


# ---------------------------------------------------------------- GEMM
GEMM_CASES = [(100, 72, 72), (130, 216, 40), (70, 24, 100), (33, 300, 18), (200, 90, 64), (1, 3, 64), (257, 129, 33), (10, 192, 4)]


def check_gemm(dev, m, n, k):
    x, w, b, r = R(m, k, dev=dev), R(n, k, dev=dev), R(n, dev=dev), R(m, n, dev=dev)
    y = ops.linear_fwd(x, w, b, relu=True, res=r)
    close(y, torch.relu(c64(x) @ c64(w).t() + c64(b) + c64(r)), what="linear_fwd")
    dy = R(m, n, seed=1, dev=dev)
    close(ops.linear_dgrad(dy, w), c64(dy) @ c64(w), what="linear_dgrad")
    dw0 = R(n, k, seed=2, dev=dev)
    dw = ops.linear_wgrad(dy, x, dw0.clone(), accumulate=True)
    close(dw, c64(dw0) + c64(dy).t() @ c64(x), what="linear_wgrad")
    # strided views (fused qkv buffer)
    big = R(m, 3 * n, seed=3, dev=dev)
    close(ops.linear_dgrad(big[:, n:2 * n], w), c64(big)[:, n:2 * n] @ c64(w), what="dgrad strided")


BATCHED_GEMM_CASES = [(2, 4, 174, 18), (1, 4, 174, 54), (1, 2, 50, 144), (1, 4, 174, 378 // 7)]


def check_attention(dev, B, nh, T, hs):
    """The five batched GEMMs + softmax of one attention layer (transfuser.py:510-527) on a fused qkv buffer."""
    C = nh * hs
    qkv = R(B, T, 3 * C, dev=dev, scale=0.5)
    Tp = (T + 3) // 4 * 4
    att = torch.zeros(B * nh, T, Tp, device=dev)
    alpha = 1.0 / math.sqrt(hs)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    sa = (T * 3 * C, hs)
    ops.gemm(q, k, att, T, T, hs, 3 * C, 3 * C, Tp, alpha=alpha, batch=B * nh, inner=nh, sa=sa, sb=sa, sc=(nh * T * Tp, T * Tp))
    ops.softmax_fwd_(att, B * nh * T, T, Tp)
    y = torch.empty(B, T, C, device=dev)
    ops.gemm(att, v, y, T, hs, T, Tp, 3 * C, C, b_trans=True, batch=B * nh, inner=nh, sa=(nh * T * Tp, T * Tp), sb=sa, sc=(T * C, hs))
    qh, kh, vh = [c64(t).reshape(B, T, nh, hs).transpose(1, 2) for t in (q, k, v)]
    pr = F.softmax((qh @ kh.transpose(-2, -1)) * alpha, dim=-1)
    ref = (pr @ vh).transpose(1, 2).reshape(B, T, C)
    close(att[:, :, :T].reshape(B, nh, T, T), pr, what="att probs")
    close(y, ref, what="att out")
    # backward GEMMs
    dy = R(B, T, C, seed=5, dev=dev)
    dqkv = torch.zeros_like(qkv)
    datt = torch.zeros_like(att)
    ops.gemm(dy, v, datt, T, T, hs, C, 3 * C, Tp, batch=B * nh, inner=nh, sa=(T * C, hs), sb=sa, sc=(nh * T * Tp, T * Tp))  # dP = dY V^T
    ops.gemm(att, dy, dqkv[..., 2 * C:], T, hs, T, Tp, C, 3 * C, a_trans=True, b_trans=True, batch=B * nh, inner=nh,
             sa=(nh * T * Tp, T * Tp), sb=(T * C, hs), sc=sa)  # dV = P^T dY
    ops.softmax_bwd_(att, datt, B * nh * T, T, Tp)
    ops.gemm(datt, k, dqkv[..., :C], T, hs, T, Tp, 3 * C, 3 * C, b_trans=True, alpha=alpha, batch=B * nh, inner=nh,
             sa=(nh * T * Tp, T * Tp), sb=sa, sc=sa)  # dQ = dS K
    ops.gemm(datt, q, dqkv[..., C:2 * C], T, hs, T, Tp, 3 * C, 3 * C, a_trans=True, b_trans=True, alpha=alpha, batch=B * nh, inner=nh,
             sa=(nh * T * Tp, T * Tp), sb=sa, sc=sa)  # dK = dS^T Q
    qr = c64(qkv).requires_grad_(True)
    q2, k2, v2 = [t.reshape(B, T, nh, hs).transpose(1, 2) for t in (qr[..., :C], qr[..., C:2 * C], qr[..., 2 * C:])]
    out = (F.softmax((q2 @ k2.transpose(-2, -1)) * alpha, dim=-1) @ v2).transpose(1, 2).reshape(B, T, C)
    out.backward(c64(dy))
    close(dqkv, qr.grad, what="attention grads")


# ---------------------------------------------------------------- conv
CONV_CASES = [(2, 10, 12, 32, 32, 3, 1, 1), (2, 11, 13, 48, 48, 3, 2, 2), (1, 9, 9, 72, 72, 3, 1, 3), (2, 8, 8, 32, 72, 1, 2, 1),
              (2, 6, 7, 64, 7, 3, 1, 1), (2, 5, 22, 64, 128, 3, 1, 1), (2, 16, 16, 64, 64, 1, 1, 1), (1, 7, 5, 216, 216, 3, 2, 9)]


def cl(w):
    """(Cout, Cin/g, kh, kw) weight in channels_last memory format (what our parameters use)."""
    return w.contiguous(memory_format=torch.channels_last) if w.shape[2] > 1 or w.shape[3] > 1 else w.contiguous()


def check_conv(dev, B, Hi, Wi, Cin, Cout, ks, stride, groups):
    x = R(B, Cin, Hi, Wi, dev=dev)
    w = R(Cout, Cin // groups, ks, ks, dev=dev) * 0.1
    b = R(Cout, dev=dev)
    x64, w64 = c64(x).requires_grad_(True), c64(w).requires_grad_(True)          # reference: float64 on the CPU (c64)
    y_pre = F.conv2d(x64, w64, c64(b), stride, ks // 2, 1, groups)
    y = torch.relu(y_pre)
    dy = R(*y.shape, seed=1, dev=dev)
    xh = x.detach().permute(0, 2, 3, 1).contiguous()
    wh = cl(w.detach())
    yh = ops.conv_fwd(xh, wh, b, stride, None, groups, relu=True)
    close(yh.permute(0, 3, 1, 2), y, what="conv fwd")
    # the reference backward uses OUR ReLU mask: among ~10^6 outputs a few pre-activations lie within round-off of 0 and may take the
    # other side in the reference - one such flip changes a weight-gradient entry by O(|dy * x|), far above the summation tolerance
    mask = c64(yh.permute(0, 3, 1, 2) > 0)
    gx, gw = torch.autograd.grad(y_pre, [x64, w64], c64(dy) * mask)
    y = yh.permute(0, 3, 1, 2)
    dyh = ops.relu_mask(dy.permute(0, 2, 3, 1).contiguous(), yh)
    dxh = ops.conv_dgrad(dyh, wh, xh.shape, stride, None, groups)
    close(dxh.permute(0, 3, 1, 2), gx, what="conv dgrad")
    dw = torch.zeros_like(wh)
    ops.conv_wgrad(dyh, xh, dw, stride, None, groups)
    close(dw, gw, what="conv wgrad")
    db = ops.colsum(dy.permute(0, 2, 3, 1).contiguous(), 1, dyh.numel() // Cout, Cout, mask=yh)
    close(db[0], (c64(dy) * mask).sum((0, 2, 3)), what="bias grad")


DIRECT_CONV_CASES = [(2, 10, 70, 32, 32), (1, 9, 33, 32, 7), (2, 5, 40, 32, 1), (1, 12, 64, 8, 32), (1, 4, 32, 12, 20), (3, 3, 5, 32, 32), (1, 7, 35, 10, 3), (2, 17, 9, 31, 8)]


def check_conv_direct(dev, B, H, W, Cin, Cout):
    """The LDS-tiled direct 3x3 kernels (decoder tail layers): same checks as check_conv with the size threshold lifted, plus the
    accumulate modes of dgrad / wgrad."""
    old = ops._DIRECT_MIN_PIXELS
    ops._DIRECT_MIN_PIXELS = 0
    try:
        assert ops._direct_ok((B, H, W, Cin), Cout, Cin, 3, 1, 1, 1)
        check_conv(dev, B, H, W, Cin, Cout, 3, 1, 1)
        x = R(B, H, W, Cin, seed=3, dev=dev)
        dy = R(B, H, W, Cout, seed=4, dev=dev)
        w = cl(R(Cout, Cin, 3, 3, seed=5, dev=dev) * 0.1)
        base_dx, base_dw = R(B, H, W, Cin, seed=6, dev=dev), cl(R(Cout, Cin, 3, 3, seed=7, dev=dev))
        dx0 = ops.conv_dgrad(dy, w, x.shape, 1, None, 1)
        dx1 = ops.conv_dgrad(dy, w, x.shape, 1, None, 1, out=base_dx.clone(), accumulate=True)
        close(dx1, base_dx + dx0, what="direct dgrad accumulate")
        dw0 = ops.conv_wgrad(dy, x, torch.zeros_like(w), 1, None, 1, accumulate=False)
        dw1 = ops.conv_wgrad(dy, x, base_dw.clone(), 1, None, 1, accumulate=True)
        close(dw1, base_dw + dw0, what="direct wgrad accumulate", tol=2e-5 * max(1.0, B * H * W / 500.0))
    finally:
        ops._DIRECT_MIN_PIXELS = old


THIN_CONV_CASES = [(1, 9, 33, 7), (2, 5, 40, 1), (1, 16, 64, 3), (2, 7, 31, 5), (1, 3, 30, 7), (1, 15, 29, 1)]


def check_conv_thin(dev, B, H, W, Cout):
    """Thin-output 3x3 convolutions (32 -> Cout <= 7, csrc/conv_thin.cpp): forward with bias, input gradient with the fused ReLU mask of the
    preceding layer (+ accumulate), weight gradient with the bias gradient riding along (+ accumulate) vs PyTorch."""
    old = ops._DIRECT_MIN_PIXELS
    ops._DIRECT_MIN_PIXELS = 0
    try:
        Cin = 32
        assert ops._thin_ok((B, H, W, Cin), Cout, Cin, 3, 1, 1, 1)
        x = R(B, Cin, H, W, dev=dev).requires_grad_(True)
        w = (R(Cout, Cin, 3, 3, dev=dev) * 0.1).requires_grad_(True)
        b = R(Cout, dev=dev).requires_grad_(True)
        y = F.conv2d(x, w, b, 1, 1)
        dy = R(*y.shape, seed=1, dev=dev)
        gx, gw, gb = torch.autograd.grad(y, [x, w, b], dy)
        xh = x.detach().permute(0, 2, 3, 1).contiguous()
        wh = cl(w.detach())
        dyh = dy.permute(0, 2, 3, 1).contiguous()
        yh = ops.conv_fwd(xh, wh, b.detach(), 1, None, 1, relu=False)
        close(yh.permute(0, 3, 1, 2), y, what="thin fwd")
        close(ops.conv_fwd(xh, wh, None, 1, None, 1), F.conv2d(x, w, None, 1, 1).permute(0, 2, 3, 1), what="thin fwd no bias")
        dx = ops.conv_dgrad(dyh, wh, xh.shape, 1, None, 1)
        close(dx.permute(0, 3, 1, 2), gx, what="thin dgrad")
        mask = R(B, H, W, Cin, seed=9, dev=dev)
        dxm = ops.conv_dgrad(dyh, wh, xh.shape, 1, None, 1, mask=mask)
        assert torch.equal(dxm, dx * (mask > 0)), "thin dgrad fused mask"
        base = R(B, H, W, Cin, seed=6, dev=dev)
        close(ops.conv_dgrad(dyh, wh, xh.shape, 1, None, 1, out=base.clone(), accumulate=True), base + dx, what="thin dgrad accumulate")
        tol = 2e-5 * max(1.0, B * H * W / 500.0)
        dw = torch.zeros_like(wh)
        db = R(Cout, seed=8, dev=dev)
        db0 = db.clone()
        assert ops.conv_wgrad_takes_bias(xh.shape, Cout, 3)
        ops.conv_wgrad(dyh, xh, dw, 1, None, 1, accumulate=False, dbias=db)
        close(dw, gw, what="thin wgrad", tol=tol)
        close(db, db0 + gb, what="thin bias grad", tol=tol)
        basew = cl(R(Cout, Cin, 3, 3, seed=7, dev=dev))
        dw1 = ops.conv_wgrad(dyh, xh, basew.clone(), 1, None, 1, accumulate=True)
        close(dw1, basew + dw, what="thin wgrad accumulate", tol=tol)
    finally:
        ops._DIRECT_MIN_PIXELS = old


def check_stem(dev, B, H, W):
    rgb = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(0)).float().to(dev)
    w = (R(32, 3, 3, 3, dev=dev) * 0.2).requires_grad_(True)
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
    y = F.conv2d((rgb / 255.0 - mean) / std, w, None, 2, 1)
    dy = R(*y.shape, seed=1, dev=dev)
    (gw,) = torch.autograd.grad(y, [w], dy)
    wh = cl(w.detach())
    yh = ops.stem_conv_fwd(rgb, None, wh, True)
    close(yh.permute(0, 3, 1, 2), y, what="stem fwd")
    dw = torch.zeros_like(wh)
    ops.stem_conv_wgrad(dy.permute(0, 2, 3, 1).contiguous(), rgb, None, dw, True)
    close(dw, gw, what="stem wgrad")
    # lidar stem: 2-channel histogram + 1-channel target point image, never concatenated
    l0, l1 = torch.rand(B, 2, H, H, device=dev), torch.rand(B, 1, H, H, device=dev)
    w2 = (R(32, 3, 3, 3, seed=3, dev=dev) * 0.2)
    y2 = F.conv2d(torch.cat((l0, l1), 1), w2, None, 2, 1)
    close(ops.stem_conv_fwd(l0, l1, cl(w2), False).permute(0, 3, 1, 2), y2, what="lidar stem fwd")
    # its weight gradient (two NCHW sources) accumulated into a non-zero buffer; run twice: the direct kernels' panel reduction is deterministic
    w2r = w2.clone().requires_grad_(True)
    y2r = F.conv2d(torch.cat((l0, l1), 1), w2r, None, 2, 1)
    dy2 = R(*y2r.shape, seed=4, dev=dev)
    (gw2,) = torch.autograd.grad(y2r, [w2r], dy2)
    init = cl(R(32, 3, 3, 3, seed=5, dev=dev))
    outs = []
    for _ in range(2):
        dw2 = init.clone()
        ops.stem_conv_wgrad(dy2.permute(0, 2, 3, 1).contiguous(), l0, l1, dw2, False)
        outs.append(dw2)
    close(outs[0], init + gw2, what="lidar stem wgrad (accumulate)")
    assert torch.equal(outs[0], outs[1]), "stem wgrad: not reproducible"


def check_bn_se_consumer_fusion(dev, B=3, H=6, W=10, C=48, Cr=12):
    """BatchNorm apply folded into its consumers (conv2 -> BN -> ReLU -> SE of a RegNetY bottleneck): the fused ops (statistics finalize ->
    squeeze chunk sums of relu(bn(y)) finished inside the excitation kernel -> BN + ReLU + SE scale in one pass; backward: gate gradient from
    recomputed z, BatchNorm backward with the recomputed mask) against PyTorch autograd of the same sub-graph."""
    y = R(B, H, W, C, dev=dev)
    gamma, beta = (torch.rand(C, device=dev) * 0.5 + 0.75), R(C, seed=2, dev=dev) * 0.3
    w1, b1 = R(Cr, C, 1, 1, seed=3, dev=dev) * 0.3, R(Cr, seed=4, dev=dev) * 0.1
    w2, b2 = R(C, Cr, 1, 1, seed=5, dev=dev) * 0.3, R(C, seed=6, dev=dev) * 0.1
    # reference (PyTorch): batch statistics, biased variance, eps 1e-5
    yr = y.detach().clone().requires_grad_(True)
    prm = [t.detach().clone().requires_grad_(True) for t in (gamma, beta, w1, b1, w2, b2)]
    mean, var = yr.mean((0, 1, 2)), yr.var((0, 1, 2), unbiased=False)
    z = torch.relu((yr - mean) / torch.sqrt(var + 1e-5) * prm[0] + prm[1])
    sq = z.mean((1, 2))
    g1r = torch.relu(sq @ prm[2].view(Cr, C).t() + prm[3])
    gater = g1r @ prm[4].view(C, Cr).t() + prm[5]
    zs = z * torch.sigmoid(gater)[:, None, None, :]
    dzs = R(B, H, W, C, seed=7, dev=dev)
    grads = torch.autograd.grad(zs, [yr] + prm, dzs)
    # product: statistics as a producer's epilogue would have written them (one part = the whole tensor, Welford triple)
    rows = B * H * W
    cs = ops.ColStat(rows, C, dev, max_parts=1)
    y2 = y.reshape(rows, C)
    m = y2.mean(0)
    cs.buf[:C] = rows; cs.buf[C:2 * C] = m; cs.buf[2 * C:] = ((y2 - m) ** 2).sum(0)
    cs.nparts.value = 1
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    coef, sm, si = ops.bn_finalize_parts(cs, gamma, beta, rm, rv)
    close(sm, mean, what="fused bn mean"); close(si, 1.0 / torch.sqrt(var + 1e-5), tol=1e-4, what="fused bn invstd")
    s_, g1, gate = ops.se_squeeze_excite_bn_fwd(y, coef, w1, b1, w2, b2)
    close(s_, sq, what="fused squeeze"); close(g1, g1r, what="fused excite g1"); close(gate, gater, what="fused gate")
    close(ops.se_scale_bn_fwd(y, coef, gate), zs, what="fused bn + relu + se scale")
    dw1, db1, dw2, db2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros_like(b2)
    ds = ops.se_gate_excite_bn_bwd(dzs, y, coef, gate, s_, g1, w1, w2, dw1, db1, dw2, db2)
    for got, ref, what in ((dw1, grads[3], "dW1"), (db1, grads[4], "db1"), (dw2, grads[5], "dW2"), (db2, grads[6], "db2")):
        close(got, ref, tol=1e-4, what="fused excite bwd " + what)
    dz = ops.se_scale_bwd_x(dzs, gate, ds, y.shape)
    dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy = ops.bn_bwd_remask(dz, y, coef, gamma, sm, si, dgm, dbt)
    close(dy, grads[0], tol=1e-4, what="fused bn bwd (recomputed mask)")
    close(dgm, grads[1], tol=1e-4, what="fused bn dgamma"); close(dbt, grads[2], tol=1e-4, what="fused bn dbeta")
    # the SE scale's backward folded into the BatchNorm backward (dz never written)
    dgm2, dbt2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy2 = ops.bn_bwd_remask_se(dzs, gate, ds, y, coef, gamma, sm, si, dgm2, dbt2)
    close(dy2, grads[0], tol=1e-4, what="bn bwd with the SE scale backward folded in")
    close(dgm2, grads[1], tol=1e-4, what="folded bn dgamma"); close(dbt2, grads[2], tol=1e-4, what="folded bn dbeta")


def check_convnext_pieces(dev):
    """ConvNeXt block pieces (csrc/convnext.cpp; timm convnext_*, transfuser.py:395-416): depthwise 7x7 (+ bias) forward / input gradient (flipped
    taps, accumulate) / weight + bias gradient, exact GELU forward / backward, layer scale + shortcut, column sum of a product, the 4x4 / s4 patchify
    stem on NCHW inputs and the 2x2 / s2 stage-entry convolution through the generic engine - all against PyTorch."""
    for (B, H, W, C) in ((2, 9, 11, 16), (1, 5, 20, 72), (3, 4, 4, 8)):
        x = R(B, C, H, W, dev=dev).requires_grad_(True)
        w = (R(C, 1, 7, 7, seed=2, dev=dev) * 0.2).requires_grad_(True)
        b = R(C, seed=3, dev=dev).requires_grad_(True)
        y = F.conv2d(x, w, b, 1, 3, 1, C)
        dy = R(*y.shape, seed=4, dev=dev)
        gx, gw, gb = torch.autograd.grad(y, [x, w, b], dy)
        xh, dyh = x.detach().permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous()
        close(ops.dwconv7(xh, w.detach(), b.detach()).permute(0, 3, 1, 2), y, what="dwconv7 fwd")
        base = R(B, H, W, C, seed=5, dev=dev)
        close(ops.dwconv7(dyh, w.detach(), None, flip=True, out=base.clone(), accumulate=True), base + gx.permute(0, 2, 3, 1), what="dwconv7 dgrad (accumulate)")
        dw, db = torch.zeros_like(w.detach()), torch.zeros(C, device=dev)
        ops.dwconv7_wgrad(dyh, xh, dw, db)
        close(dw, gw, tol=1e-4, what="dwconv7 wgrad"); close(db, gb, tol=1e-4, what="dwconv7 bias grad")
    x = (R(37, 24, dev=dev) * 2).requires_grad_(True)
    y = F.gelu(x)
    dy = R(37, 24, seed=1, dev=dev)
    close(ops.gelu_fwd(x.detach()), y, what="gelu fwd")
    close(ops.gelu_bwd(dy, x.detach()), torch.autograd.grad(y, x, dy)[0], what="gelu bwd")
    g, bt, res = R(24, seed=2, dev=dev), R(24, seed=3, dev=dev), R(37, 24, seed=4, dev=dev)
    close(ops.colscale_add(x.detach(), g, bt, res), res + g * x.detach() + bt, what="colscale_add")
    close(ops.colscale_add(x.detach(), g), g * x.detach(), what="colscale")
    out = R(24, seed=5, dev=dev)
    close(ops.colsum_mul(x.detach(), dy, out.clone()), out + (x.detach() * dy).sum(0), tol=1e-4, what="colsum_mul")
    # patchify stem (4x4 / s4, pad 0) on NCHW + 2x2 / s2 stage entry
    rgb = torch.randint(0, 256, (2, 3, 16, 24), generator=torch.Generator().manual_seed(0)).float().to(dev)
    w4 = (R(16, 3, 4, 4, seed=6, dev=dev) * 0.1).requires_grad_(True)
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
    y = F.conv2d((rgb / 255.0 - mean) / std, w4, None, 4, 0)
    dy = R(*y.shape, seed=7, dev=dev)
    (gw4,) = torch.autograd.grad(y, [w4], dy)
    close(ops.stem_conv_fwd(rgb, None, cl(w4.detach()), True, 4, 0).permute(0, 3, 1, 2), y, what="4x4 s4 stem fwd")
    dw4 = torch.zeros_like(cl(w4.detach()))
    ops.stem_conv_wgrad(dy.permute(0, 2, 3, 1).contiguous(), rgb, None, dw4, True, stride=4, pad=0)
    close(dw4, gw4, tol=1e-4, what="4x4 s4 stem wgrad")
    x = R(2, 16, 8, 12, seed=8, dev=dev).requires_grad_(True)
    w2 = (R(32, 16, 2, 2, seed=9, dev=dev) * 0.2).requires_grad_(True)
    b2 = R(32, seed=10, dev=dev)
    y = F.conv2d(x, w2, b2, 2, 0)
    dy = R(*y.shape, seed=11, dev=dev)
    gx, gw2 = torch.autograd.grad(y, [x, w2], dy)
    xh, dyh, wh = x.detach().permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous(), cl(w2.detach())
    close(ops.conv_fwd(xh, wh, b2, 2, 0, 1).permute(0, 3, 1, 2), y, what="2x2 s2 conv fwd")
    close(ops.conv_dgrad(dyh, wh, xh.shape, 2, 0, 1).permute(0, 3, 1, 2), gx, what="2x2 s2 conv dgrad")
    dw2 = torch.zeros_like(wh)
    ops.conv_wgrad(dyh, xh, dw2, 2, 0, 1)
    close(dw2, gw2, tol=1e-4, what="2x2 s2 conv wgrad")


def check_resnet_stem_and_pool(dev):
    """The ResNet stem pieces (timm resnet18/34/50: the reference's default trunks, transfuser.py:15,136-143): 7x7 / s2 / p3 convolution on the
    NCHW inputs (with normalize_imagenet folded in, or the LiDAR + target-point channels un-concatenated) and nn.MaxPool2d(3, 2, 1) forward /
    backward incl. ties (first maximum wins, as ATen) and odd sizes; a dense 3x3 / s2 and 1x1 / s2 convolution through the generic engine."""
    B, H, W = 2, 20, 26
    rgb = torch.randint(0, 256, (B, 3, H, W), generator=torch.Generator().manual_seed(0)).float().to(dev)
    w = (R(16, 3, 7, 7, dev=dev) * 0.1).requires_grad_(True)
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)
    y = F.conv2d((rgb / 255.0 - mean) / std, w, None, 2, 3)
    dy = R(*y.shape, seed=1, dev=dev)
    (gw,) = torch.autograd.grad(y, [w], dy)
    wh = cl(w.detach())
    close(ops.stem_conv_fwd(rgb, None, wh, True, 2, 3).permute(0, 3, 1, 2), y, what="7x7 stem fwd")
    dw = torch.zeros_like(wh)
    ops.stem_conv_wgrad(dy.permute(0, 2, 3, 1).contiguous(), rgb, None, dw, True, stride=2, pad=3)
    close(dw, gw, what="7x7 stem wgrad", tol=1e-4)
    l0, l1 = torch.rand(B, 2, H, H, device=dev), torch.rand(B, 1, H, H, device=dev)
    w2 = R(16, 3, 7, 7, seed=3, dev=dev) * 0.1
    close(ops.stem_conv_fwd(l0, l1, cl(w2), False, 2, 3).permute(0, 3, 1, 2), F.conv2d(torch.cat((l0, l1), 1), w2, None, 2, 3), what="7x7 lidar stem fwd")
    for (b, h, wd, c) in ((2, 9, 11, 8), (1, 16, 20, 64), (1, 7, 5, 3)):
        x = (torch.randint(-3, 4, (b, c, h, wd), generator=torch.Generator().manual_seed(5)).float() * 0.5).to(dev).requires_grad_(True)     # many ties
        yr = F.max_pool2d(x, 3, 2, 1)
        g = R(*yr.shape, seed=6, dev=dev)
        (gx,) = torch.autograd.grad(yr, [x], g)
        yh, idx = ops.maxpool3x3s2_fwd(x.detach().permute(0, 2, 3, 1).contiguous())
        assert torch.equal(yh.permute(0, 3, 1, 2), yr.detach()), "maxpool fwd"
        dx = ops.maxpool3x3s2_bwd(g.permute(0, 2, 3, 1).contiguous(), idx, (b, h, wd, c))
        close(dx.permute(0, 3, 1, 2), gx, what="maxpool bwd (ties: first maximum)")


# ---------------------------------------------------------------- norms
def check_layernorm(dev, rows, C):
    x = R(rows, C, dev=dev)
    g, b = R(C, seed=1, dev=dev) * 0.3 + 1, R(C, seed=2, dev=dev)
    dy = R(rows, C, seed=3, dev=dev)
    x64, g64, b64 = c64(x).requires_grad_(True), c64(g).requires_grad_(True), c64(b).requires_grad_(True)      # reference: float64 on the CPU
    y = F.layer_norm(x64, (C,), g64, b64, 1e-5)
    gx, gg, gb = torch.autograd.grad(y, [x64, g64, b64], c64(dy))
    yh, mean, rstd = ops.layernorm_fwd(x.detach(), g.detach(), b.detach())
    close(yh, y, what="ln fwd")
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx = ops.layernorm_bwd(dy, x.detach(), g.detach(), mean, rstd, dg, db)
    close(dx, gx, what="ln dx")
    close(dg, gg, what="ln dgamma")
    close(db, gb, what="ln dbeta")
    acc0 = R(rows, C, seed=4, dev=dev)              # dx accumulated in place (the residual stream of a transformer Block)
    acc = acc0.clone()
    ops.layernorm_bwd(dy, x.detach(), g.detach(), mean, rstd, torch.zeros(C, device=dev), torch.zeros(C, device=dev), dx=acc, accumulate=True)
    close(acc, c64(acc0) + gx, what="ln dx accumulate")


def check_softmax(dev, rows, n, ld):
    s = R(rows, ld, dev=dev)
    ref = F.softmax(c64(s)[:, :n], dim=-1)                                  # reference: float64 on the CPU
    p = ops.softmax_fwd_(s.clone(), rows, n, ld)
    close(p[:, :n], ref, what="softmax fwd")
    dp = R(rows, ld, seed=1, dev=dev)
    sr = c64(s)[:, :n].clone().requires_grad_(True)
    (gs,) = torch.autograd.grad(F.softmax(sr, dim=-1), [sr], c64(dp)[:, :n])
    close(ops.softmax_bwd_(p, dp.clone(), rows, n, ld)[:, :n], gs, what="softmax bwd")


def check_bn(dev, B, H, W, C, relu, with_res):
    x = R(B, C, H, W, dev=dev) * 2 + 0.7
    g, b = R(C, seed=1, dev=dev) * 0.3 + 1, R(C, seed=2, dev=dev)
    res = R(B, C, H, W, seed=4, dev=dev) if with_res else None
    x64, g64, b64 = c64(x).requires_grad_(True), c64(g).requires_grad_(True), c64(b).requires_grad_(True)      # reference: float64 on the CPU
    res64 = c64(res).requires_grad_(True) if with_res else None
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    y = F.batch_norm(x64, rm, rv, g64, b64, True, 0.1, 1e-5)
    if with_res:
        y = y + res64
    if relu:
        y = torch.relu(y)
    dy = R(B, C, H, W, seed=3, dev=dev)
    grads = torch.autograd.grad(y, [x64, g64, b64] + ([res64] if with_res else []), c64(dy))
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous()
    rm2, rv2 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    yh, sm, si = ops.bn_fwd(nhwc(x), g.detach(), b.detach(), rm2, rv2, nhwc(res) if with_res else None, relu, True)
    close(yh.permute(0, 3, 1, 2), y, what="bn fwd")
    close(rm2, rm, what="running mean")
    close(rv2, rv, what="running var")
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx, dres = ops.bn_bwd(nhwc(dy), yh if relu else None, nhwc(x), g.detach(), sm, si, dg, db, want_dres=with_res)
    close(dx.permute(0, 3, 1, 2), grads[0], what="bn dx")
    close(dg, grads[1], what="bn dgamma")
    close(db, grads[2], what="bn dbeta")
    if with_res:
        close(dres.permute(0, 3, 1, 2), grads[3], what="bn dres")


def check_bn_eval(dev):
    B, H, W, C = 2, 4, 5, 24
    x = R(B, C, H, W, dev=dev)
    g, b = R(C, seed=1, dev=dev) * 0.3 + 1, R(C, seed=2, dev=dev)
    rm, rv = R(C, seed=5, dev=dev), torch.rand(C, device=dev) + 0.5
    y = torch.relu(F.batch_norm(x, rm.clone(), rv.clone(), g, b, False, 0.1, 1e-5))
    yh, _, _ = ops.bn_fwd(x.permute(0, 2, 3, 1).contiguous(), g, b, rm, rv, None, True, False)
    close(yh.permute(0, 3, 1, 2), y, what="bn eval")


def check_colsum(dev):
    x = R(3, 50, 72, dev=dev)
    close(ops.colsum(x, 3, 50, 72, scale=1 / 50.0), x.mean(1), what="colsum mean")
    x2 = R(1000, 7, dev=dev)
    acc = R(1, 7, seed=1, dev=dev)
    close(ops.colsum(x2, 1, 1000, 7, out=acc.clone(), accumulate=True), acc + x2.sum(0, keepdim=True), what="colsum acc")
    x3 = R(4, 33, 1512, dev=dev)
    close(ops.colsum(x3, 4, 33, 1512), x3.sum(1), what="colsum wide")


def check_fused_finalize(dev):
    """The reduce kernels finish their own reduction (the block that draws a column tile's last ticket sums the chunk partials): many chunks,
    ragged row counts, several segments and column tiles; bitwise run-to-run reproducible; the arrival counters at the end of the workspace
    are back at zero after every launch (so the next launch on the stream starts clean)."""
    L = ops.L()
    L.tf_workspace_bytes.restype = __import__("ctypes").c_long
    nws = L.tf_workspace_bytes() // 4
    for (nseg, rows, C) in [(1, 5000, 576), (3, 777, 72), (2, 1301, 1512), (1, 4099, 7), (5, 64, 216)]:
        x = R(nseg, rows, C, dev=dev)
        m = R(nseg, rows, C, seed=1, dev=dev)
        want = (x.double() * (m > 0)).sum(1)
        outs = [ops.colsum(x, nseg, rows, C, mask=m) for _ in range(3)]
        close(outs[0], want, tol=2e-5, what="fused finalize colsum %s" % ((nseg, rows, C),))
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "fused finalize: not run-to-run reproducible"
        acc = R(nseg, C, seed=2, dev=dev)
        close(ops.colsum(x, nseg, rows, C, scale=0.5, out=acc.clone(), accumulate=True), acc.double() + 0.5 * x.double().sum(1), tol=2e-5, what="fused finalize colsum acc")
        ws = ops.workspace(x.device)
        assert ws.numel() == nws and int(ws[-4096:].view(torch.int32).abs().max()) == 0, "arrival counters not reset"
    # BatchNorm backward through the fused finalize: dgamma / dbeta accumulate, coefficients feed the apply pass
    B, H, W, C = 4, 37, 29, 216
    xb = R(B, H, W, C, dev=dev)
    g, b = R(C, seed=3, dev=dev), R(C, seed=4, dev=dev)
    y, sm, si = ops.bn_fwd(xb, g, b, torch.zeros(C, device=dev), torch.ones(C, device=dev), relu=True)
    dz = R(B, H, W, C, seed=5, dev=dev)
    res = []
    for _ in range(2):
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx, _ = ops.bn_bwd(dz, y, xb, g, sm, si, dg, db)
        res.append((dx, dg, db))
    assert all(torch.equal(a, b_) for a, b_ in zip(res[0], res[1])), "fused finalize BatchNorm backward: not reproducible"
    xr = xb.double().requires_grad_(True); gr = g.double().requires_grad_(True); br = b.double().requires_grad_(True)
    yr = torch.relu(F.batch_norm(xr.permute(0, 3, 1, 2), None, None, gr, br, True, 0.1, 1e-5)).permute(0, 2, 3, 1)
    gx, gg, gb = torch.autograd.grad(yr, [xr, gr, br], dz.double())
    close(res[0][0], gx, tol=2e-4, what="fused finalize bn dx"); close(res[0][1], gg, tol=2e-4, what="fused finalize bn dgamma"); close(res[0][2], gb, tol=2e-4, what="fused finalize bn dbeta")
    assert int(ops.workspace(xb.device)[-4096:].view(torch.int32).abs().max()) == 0


def check_skinny_wgrad(dev):
    """Weight gradients with few output features over many rows (the 1x1 head convolutions, model.py:93-99) take the streaming kernel:
    dW += dY^T X for 1..16 output features, ragged row counts, strided dY; shapes outside its envelope keep the engine path."""
    for (rows, no, C) in [(40960, 1, 64), (40960, 2, 64), (5000, 3, 64), (4099, 12, 64), (8192, 16, 256), (4500, 5, 72), (4096, 7, 16)]:
        dyb = R(rows, no + 3, dev=dev)
        dy = dyb[:, 1:1 + no]                       # row stride > no
        x = R(rows, C, seed=1, dev=dev)
        dw0 = R(no, C, seed=2, dev=dev)
        dw = dw0.clone()
        ops.gemm(dy, x, dw, no, C, rows, dyb.stride(0), x.stride(0), dw.stride(0), a_trans=True, b_trans=True, accumulate=True)
        want = dw0.double() + dy.double().t() @ x.double()
        close(dw, want, tol=3e-5, what="skinny wgrad %s" % ((rows, no, C),))


def check_colsum_multi(dev):
    """Several column sums over the same rows in one single-pass launch (the bias gradients of a transformer Block): ragged strips (C % 32 != 0),
    strided views, accumulation into non-zero destinations, bitwise reproducible; the non-vector layout falls back to colsum."""
    for rows in (1740, 33, 1):
        big = R(rows, 3 * 72 + 40, dev=dev)
        xs = [R(rows, 72, seed=1, dev=dev), R(rows, 288, seed=2, dev=dev), big[:, 40:40 + 216], R(rows, 1512, seed=3, dev=dev), R(rows, 4, seed=4, dev=dev)]
        init = [R(x.shape[1], seed=10 + i, dev=dev) for i, x in enumerate(xs)]
        outs = []
        for _ in range(2):
            o = [t.clone() for t in init]
            ops.colsum_multi(list(zip(xs, o)))
            outs.append(o)
        for x, t, a, b in zip(xs, init, outs[0], outs[1]):
            close(a, t.double() + x.double().sum(0), tol=2e-5, what="colsum_multi rows=%d C=%d" % (rows, x.shape[1]))
            assert torch.equal(a, b), "colsum_multi: not reproducible"
    x5 = R(50, 6, dev=dev)                      # C % 4 != 0: per-pair fallback
    o5 = torch.zeros(6, device=dev)
    ops.colsum_multi([(x5, o5)])
    close(o5, x5.double().sum(0), tol=2e-5, what="colsum_multi fallback")


def check_se(dev, B, H, W, C):
    x = R(B, H, W, C, dev=dev).requires_grad_(True)
    gate = R(B, C, seed=1, dev=dev).requires_grad_(True)
    y = x * torch.sigmoid(gate).view(B, 1, 1, C)
    dy = R(B, H, W, C, seed=2, dev=dev)
    gx, gg = torch.autograd.grad(y, [x, gate], dy)
    close(ops.se_scale_fwd(x.detach(), gate.detach()), y, what="se fwd")
    close(ops.se_scale_bwd_gate(dy, x.detach(), gate.detach()), gg, what="se dgate")
    dmean = R(B, C, seed=3, dev=dev)
    dx = ops.se_scale_bwd_x(dy, gate.detach(), dmean, x.shape)
    close(dx, gx + dmean.view(B, 1, 1, C) / (H * W), what="se dx")
    dx2 = ops.se_scale_bwd_x(None, None, dmean, x.shape)
    close(dx2, (dmean.view(B, 1, 1, C) / (H * W)).expand(B, H, W, C), what="gap bwd")


# ---------------------------------------------------------------- resampling
def check_pool_tokens(dev, B, H, W, C, oh, ow):
    x = R(B, C, H, W, dev=dev).requires_grad_(True)
    T = oh * ow + 7
    pos = R(T, C, seed=1, dev=dev)
    pooled = F.adaptive_avg_pool2d(x, (oh, ow))
    ref = pooled.flatten(2).transpose(1, 2) + pos[3:3 + oh * ow]
    tok = torch.zeros(B, T, C, device=dev)
    ops.pool_tokens_fwd(x.detach().permute(0, 2, 3, 1).contiguous(), oh, ow, pos, tok, 3)
    close(tok[:, 3:3 + oh * ow], ref, what="pool tokens fwd")
    dtok = R(B, T, C, seed=2, dev=dev)
    (gx,) = torch.autograd.grad(ref, [x], dtok[:, 3:3 + oh * ow])
    add = R(B, H, W, C, seed=9, dev=dev)
    dx = ops.pool_tokens_bwd(dtok, (B, H, W, C), oh, ow, 3, add=add)
    close(dx.permute(0, 3, 1, 2), gx + add.permute(0, 3, 1, 2), what="pool tokens bwd")


BILINEAR_CASES = [(2, 12, 5, 22, 40, 176, False, False), (2, 8, 8, 8, 64, 64, False, False), (1, 6, 5, 22, 64, 176, False, False),
                  (2, 4, 8, 8, 16, 16, True, False), (1, 3, 16, 16, 40, 40, True, True), (2, 8, 5, 22, 5, 22, False, False),
                  (1, 5, 8, 22, 16, 44, False, False), (2, 6, 5, 22, 1, 2, False, False), (1, 4, 9, 7, 4, 5, True, True),
                  (2, 8, 8, 22, 64, 176, True, False), (1, 32, 16, 44, 64, 176, True, False), (2, 8, 5, 22, 40, 176, True, False),
                  (1, 12, 16, 16, 40, 40, True, True), (2, 4, 3, 5, 30, 50, True, False), (1, 6, 8, 22, 32, 88, False, False), (2, 10, 8, 8, 32, 32, False, False)]


def check_bilinear(dev, B, C, Hi, Wi, Ho, Wo, in_nhwc, align):
    x = R(B, C, Hi, Wi, dev=dev).requires_grad_(True)
    y = F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=align)
    add = R(B, Ho, Wo, C, seed=1, dev=dev)
    xin = x.detach().permute(0, 2, 3, 1).contiguous() if in_nhwc else x.detach().contiguous()
    yh = ops.bilinear_fwd(xin, B, C, Hi, Wi, Ho, Wo, in_nhwc, True, align, add=add)
    close(yh, y.permute(0, 2, 3, 1) + add, what="bilinear fwd")
    dy = R(B, C, Ho, Wo, seed=2, dev=dev)
    (gx,) = torch.autograd.grad(y, [x], dy)
    dx = ops.bilinear_bwd(dy.permute(0, 2, 3, 1).contiguous(), B, C, Hi, Wi, Ho, Wo, in_nhwc, True, align)
    close(dx.permute(0, 3, 1, 2) if in_nhwc else dx, gx, what="bilinear bwd")


GATHER_CASES = [(2, 5, 22, 8, 64, 5), (3, 8, 8, 512, 110, 5), (1, 2, 3, 4, 9, 5), (2, 4, 4, 36, 7, 3)]


def check_gather_sum(dev, B, Hs, Ws, E, n, K):
    """G1 vs the reference's own formulation: B x B advanced indexing + torch.diagonal (geometric_fusion.py:133-136)."""
    g = torch.Generator().manual_seed(3)
    emb = R(B, E, Hs, Ws, dev="cpu").requires_grad_(True)
    pts = torch.stack((torch.randint(0, Ws, (B, n, K), generator=g), torch.randint(0, Hs, (B, n, K), generator=g)), -1)
    pts[0, 0] = 0                                  # the dataset pads missing correspondences with (0, 0): duplicates
    flat = pts.view(B * n * K, 2)
    enc = emb.permute(0, 2, 3, 1).contiguous()[:, flat[:, 1], flat[:, 0]].view(B, B, n, 1, K, -1)
    ref = torch.sum(torch.diagonal(enc, 0).permute(4, 3, 0, 1, 2).contiguous(), -1)[:, :, :, 0]     # (B, E, n)
    src = emb.detach().permute(0, 2, 3, 1).reshape(B, Hs * Ws, E).contiguous().to(dev)
    out = ops.gather_sum_fwd(src, pts.to(dev), Hs, Ws)
    close(out.cpu(), ref.permute(0, 2, 1), what="gather_sum fwd", tol=1e-6)
    dout = R(B, n, E, seed=5, dev="cpu")
    (gemb,) = torch.autograd.grad(ref, [emb], dout.permute(0, 2, 1))
    want = gemb.permute(0, 2, 3, 1).reshape(B, Hs * Ws, E)
    dsrc = ops.gather_sum_bwd(dout.to(dev), pts.to(dev), Hs, Ws)
    close(dsrc.cpu(), want, what="gather_sum bwd", tol=1e-5)
    acc = R(B, Hs * Ws, E, seed=6, dev=dev)
    dsrc2 = ops.gather_sum_bwd(dout.to(dev), pts.to(dev), Hs, Ws, out=acc.clone(), accumulate=True)
    close(dsrc2.cpu(), want + acc.cpu(), what="gather_sum bwd accumulate", tol=1e-5)


PILLAR_CASES = [(3, 5000, (5000, 4000, 100)), (2, 3000, (0, 3000)), (1, 64, (64,))]


def pillar_cloud(B, N, seed=1):
    g = torch.Generator().manual_seed(seed)
    pts = torch.stack([torch.rand(B, N, generator=g) * 40 - 20, torch.rand(B, N, generator=g) * 40 - 36, torch.rand(B, N, generator=g) * 5 - 4,
                       torch.rand(B, N, generator=g)], -1)
    k = min(50, N // 4)
    pts[0, :k, 0] = torch.nextafter(torch.tensor(16.0), torch.tensor(0.0))      # (x + 16) rounds up to 32.0 -> x_idx = 256 (clamped later)
    pts[0, :k, 1] = -1.0
    pts[0, k:2 * k, :2] = torch.round(pts[0, k:2 * k, :2] * 8) / 8                # points exactly on cell edges
    pts[0, 2 * k:3 * k] = pts[0, 2 * k:2 * k + 1]                                   # exact duplicates (ties in scatter_max)
    return pts


def check_pillars(dev, B, N, npts):
    """H2 vs the oracle restatement of point_pillar.py: pillar ids / inverse indices / unique rows EXACT (integers), decorated
    features and canvas within fp32 tolerance, gradients of the point-net parameters."""
    from oracle import pillars as op
    from transfuser_amd import point_pillar as pp
    kw = dict(min_x=-16, max_x=16, min_y=-32, max_y=0, pixels_per_meter=8)
    pts = pillar_cloud(B, N)
    num = torch.tensor(npts, dtype=torch.int32)
    torch.manual_seed(0)
    o = op.PointPillarNet(9, [32, 32], **kw)
    with torch.no_grad():
        for n_, p_ in o.named_parameters():
            if n_.endswith("1.weight") or n_.endswith("4.weight"):
                p_.copy_(torch.rand(p_.shape) * 0.5 + 0.75)
            if n_.endswith("bias"):
                p_.copy_(torch.randn(p_.shape) * 0.1)
    m = pp.PointPillarNet(9, [32, 32], **kw).to(dev)
    m.load_state_dict(o.state_dict(), strict=True)
    o.train(); m.train()
    kept, uniq, inverse = o.index(pts, num)
    ix = ops.pillar_index(pts.to(dev), num.to(dev), -16, 16, -32, 0, 8)
    assert ix["N"] == kept.shape[0] and ix["P"] == uniq.shape[0]
    assert torch.equal(ix["inv"].cpu().long(), inverse), "inverse indices differ from torch.unique"
    key = (uniq[:, 0] * ix["GX"] + uniq[:, 1]) * ix["GY"] + uniq[:, 2]
    assert torch.equal(ix["cellkey"].cpu().long(), key), "unique pillar rows differ from torch.unique (sorted)"
    assert torch.equal(ix["points"].cpu(), kept), "kept points (stable order)"
    close(ix["feat"], o.decorate(kept, uniq, inverse), what="decorated features", tol=1e-5)
    if ix["N"] == 0:
        return
    extra = (torch.rand(B, 1, 256, 256, generator=torch.Generator().manual_seed(4)) < 0.05).float()
    want = torch.cat((torch.rot90(o(pts, num), -1, dims=(2, 3)), extra), 1)       # model.py:738-742
    got = m.forward_nhwc(pts.to(dev), num.to(dev), extra.to(dev))
    close(got.permute(0, 3, 1, 2), want, what="pillar canvas (rot90 + concat)", tol=2e-4)
    assert torch.equal(got.permute(0, 3, 1, 2).cpu() != 0, want != 0), "canvas occupancy pattern"
    close(m.forward(pts.to(dev), num.to(dev)), o(pts, num), what="reference-API canvas", tol=2e-4)
    wgt = R(B, 33, 256, 256, seed=8)
    (want * wgt).sum().backward()
    (got * wgt.permute(0, 2, 3, 1).to(dev)).sum().backward()
    gmax = max(q.grad.abs().max().item() for q in o.parameters())
    for (n_, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
        err = (p.grad.cpu() - q.grad).abs().max().item()   # (the Linear biases feed a BatchNorm: their true gradient is 0, both sides hold round-off)
        assert err <= 2e-3 * q.grad.abs().max().item() + 1e-6 * gmax, "grad %s: err %.3e scale %.3e" % (n_, err, q.grad.abs().max().item())
    for (n_, p), (_, q) in zip(m.named_buffers(), o.named_buffers()):
        if "running" in n_:
            close(p, q, what=n_, tol=1e-4)


def check_pillar_index_forms(dev):
    """The three-launch pillar index (round 6) against the seven-launch form of rounds 3-5, host-read and static-shape modes: kept points, inverse indices,
    cell keys and features BITWISE equal; the static-shape tails (zero rows, -1 cell keys) written by the gather launch itself (ragged counts, an empty sample, a
    cloud with no kept point at all, point stride 5, a cloud concentrated in sixteen pillars)."""
    cases = [(3, 5000, (5000, 4000, 100), 4), (2, 3000, (0, 3000), 4), (1, 64, (64,), 4), (2, 1030, (1030, 7), 5), (1, 2048, (2048,), 4), (2, 9000, (9000, 8000), 4)]
    for ci, (B, N, npts, stride) in enumerate(cases):
        pts = pillar_cloud(B, N, seed=ci + 1)
        if ci == 4:
            pts[..., 0] += 100.0                     # nothing inside the range: N = P = 0
        if ci == 5:                                  # a cloud concentrated in 4 x 4 pillars of ONE slab: the slab kernel's queue overflows (> 4096 points a trip)
            pts[..., 0] = pts[..., 0].abs() % 0.5 + 1.0
            pts[..., 1] = -(pts[..., 1].abs() % 0.5) - 3.0
        if stride != 4:
            pts = torch.cat((pts, torch.full((B, N, stride - 4), 7.0)), -1)
        num = torch.tensor(npts, dtype=torch.int32)
        a = (pts.to(dev), num.to(dev), -16, 16, -32, 0, 8)
        for static in (False, True):
            want = ops._pillar_index_v1(*a, static=static)
            got = ops.pillar_index(*a, static=static)
            n, p = (int(v) for v in want["totals"].tolist())
            assert [int(v) for v in got["totals"].tolist()] == [n, p], "totals"
            assert got["N"] == want["N"] and got["P"] == want["P"]
            for k in ("points", "inv", "cellkey"):
                assert torch.equal(got[k], want[k]), "case %d static %d: %s differs between the two forms" % (ci, static, k)
            assert torch.equal(got["feat"][:n], want["feat"][:n]), "case %d static %d: features" % (ci, static)
            if static:
                assert bool((got["feat"][n:] == 0).all()) and bool((got["cellkey"][p:] == -1).all()) and bool((got["points"][n:] == 0).all())


SE_EXCITE_CASES = [(10, 576, 144), (2, 72, 8), (16, 1512, 378), (3, 218, 54), (1, 24, 6), (2, 2048, 512), (2, 3072, 64), (12, 216, 54)]


def check_se_excite(dev, B, C, Cr):
    s = R(B, C, dev=dev).requires_grad_(True)
    w1 = R(Cr, C, 1, 1, seed=1, dev=dev, scale=0.1).requires_grad_(True); b1 = R(Cr, seed=2, dev=dev, scale=0.1).requires_grad_(True)
    w2 = R(C, Cr, 1, 1, seed=3, dev=dev, scale=0.1).requires_grad_(True); b2 = R(C, seed=4, dev=dev, scale=0.1).requires_grad_(True)
    h = F.relu(F.linear(s, w1.view(Cr, C), b1))
    ref = F.linear(h, w2.view(C, Cr), b2)
    g1, gate = ops.se_excite_fwd(s.detach(), w1.detach(), b1.detach(), w2.detach(), b2.detach())
    close(g1, h, what="se excite g1")
    close(gate, ref, what="se excite gate")
    dgate = R(B, C, seed=5, dev=dev)
    gs, gw1, gb1, gw2, gb2 = torch.autograd.grad(ref, [s, w1, b1, w2, b2], dgate)
    acc = [R(*t.shape, seed=6 + i, dev=dev) for i, t in enumerate((w1, b1, w2, b2))]
    out = [a.clone() for a in acc]
    ds = ops.se_excite_bwd(dgate, s.detach(), g1, w1.detach(), w2.detach(), *out)
    close(ds, gs, what="se excite ds")
    for o, a, g, n in zip(out, acc, (gw1, gb1, gw2, gb2), ("dW1", "db1", "dW2", "db2")):
        close(o, a + g, what="se excite " + n)
    # a second backward on the same g1 (the forward-cleared scratch is spent): the kernel clears its own scratch
    out2 = [a.clone() for a in acc]
    close(ops.se_excite_bwd(dgate, s.detach(), g1, w1.detach(), w2.detach(), *out2), gs, what="se excite ds (2nd backward)")
    close(out2[0], acc[0] + gw1, what="se excite dW1 (2nd backward)")


# ---------------------------------------------------------------- losses
def check_ce(dev, rows, C, weighted):
    lg = R(rows, C, dev=dev).requires_grad_(True)
    tgt = torch.randint(0, C, (rows,), generator=torch.Generator().manual_seed(1)).to(dev)
    cw = torch.tensor([1., 1., 3.], device=dev) if weighted else None
    ref = F.cross_entropy(lg, tgt, weight=cw)
    (gl,) = torch.autograd.grad(ref, [lg])
    loss, dl, inv = ops.ce_fwd(lg.detach(), tgt, cw)
    close(loss, ref, what="ce loss")
    g = torch.tensor([0.7], device=dev)
    ops.scale_dev_(dl, g, inv, 1.0)
    close(dl, 0.7 * gl, what="ce grad")


def check_l1(dev, n, sig):
    p = R(n, dev=dev).requires_grad_(True)
    t = torch.rand(n, device=dev)
    ref = F.l1_loss(torch.sigmoid(p) if sig else p, t)
    (gp,) = torch.autograd.grad(ref, [p])
    loss, dp = ops.l1_fwd(p.detach(), t, sig)
    close(loss, ref, what="l1 loss")
    close(dp, gp, what="l1 grad")


def check_gru(dev, B, H):
    """Fused waypoint decoder vs the oracle's forward_gru (model.py:611-646) incl. parameter gradients."""
    torch.manual_seed(0)
    gru, outl = torch.nn.GRUCell(4, H).to(dev), torch.nn.Linear(H, 3).to(dev)
    z, tp = R(B, H, dev=dev).requires_grad_(True), R(B, 2, seed=1, dev=dev) * 10
    x = torch.zeros(B, 2, device=dev)
    tpf = tp.clone(); tpf[:, 1] *= -1
    h, wps = z, []
    for _ in range(4):
        h = gru(torch.cat([x, tpf], 1), h)
        x = outl(h)[:, :2] + x
        wps.append(x)
    ref = torch.stack(wps, 1)
    shift = torch.zeros_like(ref); shift[:, :, 0] = 1.3
    ref = ref - shift
    dwp = R(B, 4, 2, seed=2, dev=dev)
    ps = list(gru.parameters()) + list(outl.parameters())
    grads = torch.autograd.grad(ref, [z] + ps, dwp)
    wp, cache = ops.gru_waypoints_fwd(z.detach(), tp, gru, outl, 4, 1.3)
    close(wp, ref, what="gru wp fwd")
    bufs = (torch.zeros_like(gru.weight_ih), torch.zeros_like(gru.weight_hh), torch.zeros_like(gru.bias_ih), torch.zeros_like(gru.bias_hh),
            torch.zeros_like(outl.weight), torch.zeros_like(outl.bias))
    dz = ops.gru_waypoints_bwd(dwp, cache, gru, outl, bufs, B, H, 4)
    close(dz, grads[0], what="gru dz")
    for name, mine, ref_g in zip(("w_ih", "w_hh", "b_ih", "b_hh", "w_out", "b_out"), bufs, grads[1:]):
        close(mine, ref_g, what="gru d" + name)


def check_misc(dev):
    a, b = R(1000, dev=dev), R(1000, seed=1, dev=dev)
    close(ops.axpby(a, b, 2.0, -0.5), 2 * a - 0.5 * b, what="axpby")
    close(ops.relu_mask(a, b), a * (b > 0), what="relu mask")
    seed = torch.tensor([1234], dtype=torch.int32, device=dev)
    x = torch.ones(100000, device=dev)
    y = ops.dropout(x, seed, 3, 0.1)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01, keep
    close(y[y != 0], torch.full_like(y[y != 0], 1 / 0.9), what="dropout scale")
    assert torch.equal(y, ops.dropout(x, seed, 3, 0.1)), "dropout mask must be reproducible (backward regenerates it)"
    assert not torch.equal(y, ops.dropout(x, seed, 4, 0.1))
    r = R(*x.shape, seed=9, dev=dev)
    assert torch.equal(ops.dropout_add(x, r, seed, 3, 0.1), r + y), "dropout_add == residual + dropout with the same mask"
    # softmax + attn_drop fused (forward and backward) == the two separate launches, bitwise on the live columns
    rows, n, ld = 70, 174, 176
    sc = R(rows, ld, seed=11, dev=dev)
    p_ref = ops.softmax_fwd_(sc.clone(), rows, n, ld)
    pd_ref = ops.dropout(p_ref, seed, 7, 0.1)
    p_f = sc.clone()
    pd_f = ops.softmax_dropout_fwd_(p_f, rows, n, ld, seed, 7, 0.1)
    assert torch.equal(p_f[:, :n], p_ref[:, :n]) and torch.equal(pd_f[:, :n], pd_ref[:, :n]), "fused softmax + dropout forward"
    dpd = R(rows, ld, seed=12, dev=dev)
    g_ref = ops.softmax_bwd_(p_ref, ops.dropout(dpd, seed, 7, 0.1), rows, n, ld)
    g_f = ops.softmax_dropout_bwd_(p_ref, dpd.clone(), rows, n, ld, seed, 7, 0.1)
    assert torch.equal(g_f[:, :n], g_ref[:, :n]), "fused softmax + dropout backward"


def check_adamw(dev, n):
    p = torch.nn.Parameter(R(n, dev=dev))
    opt = torch.optim.AdamW([p], lr=1e-2)
    mine = p.detach().clone()
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    state = torch.tensor([0.0, 1e-2], device=dev)
    for it in range(3):
        g = R(n, seed=10 + it, dev=dev)
        p.grad = g.clone()
        opt.step()
        ops.adamw_(mine, g, m, v, state)
    close(mine, p, tol=1e-5, what="adamw")
    assert state[0].item() == 3.0


def check_hist(dev, B, N, stride=4, ragged=None):
    """H1 (data.py:446-470) against the line-by-line oracle, np.array_equal: points on bin edges and on the closed upper borders, > 5 hits
    per cell (the clip), long runs of one cell (the wave-level run merge), ragged ``num_points`` (incl. an empty and a full sample),
    ``stride`` 4 (float4 loads) and 5 (scalar loads)."""
    from oracle import hist
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-20, 20, (B, N)), rng.uniform(-36, 4, (B, N)), rng.uniform(-4, 1, (B, N))] +
                   [rng.uniform(0, 1, (B, N))] * (stride - 3), -1).astype(np.float32)
    pts[0, :40, 0] = 16.0; pts[0, 40:80, 1] = 0.0; pts[0, 80:120, 0] = -16.0; pts[0, 120:160, 1] = -32.0; pts[0, 160:200, 2] = -2.3
    pts[1, :500, :2] = np.round(pts[1, :500, :2] * 8) / 8   # points exactly on bin edges
    pts[1, 500:520] = pts[1, 500]                            # > 5 hits in one bin (clipping)
    pts[1, 600:603] = pts[1, 600]; pts[1, 700:704] = pts[1, 700]; pts[1, 800:805] = pts[1, 800]   # runs of 3 / 4 / 5 (below / at the clip)
    pts[1, 1000:1000 + min(300, N - 1000), :3] = np.float32([3.01, -7.02, 0.5])                    # a run spanning several waves
    pts[1, 62:67, :3] = np.float32([-3.3, -20.1, -3.0])                                             # a run across a wave boundary
    if B > 2:
        pts[2, ::2, :3] = np.float32([1.0, -1.0, 0.0]); pts[2, 1::2, :3] = np.float32([1.0, -1.0, -3.0])   # two cells alternating: no runs, 2 hot counters
    npts = np.full(B, N, np.int32)
    npts[1] = N - 777
    if ragged is not None:
        npts[:] = np.asarray(ragged, np.int32)
    out = ops.lidar_hist(torch.from_numpy(pts).to(dev), torch.from_numpy(npts).to(dev)).cpu().numpy()
    for b in range(B):
        ref = hist.lidar_to_histogram_features(pts[b, :int(npts[b]), :4])
        assert np.array_equal(out[b], ref), "H1 histogram must be bit-exact (sample %d: %d bins differ)" % (b, (out[b] != ref).sum())
    out2 = ops.lidar_hist(torch.from_numpy(pts).to(dev)).cpu().numpy()        # num_points = None: every row counts (and: the previous call left its counter workspace zeroed)
    assert np.array_equal(out2[0], hist.lidar_to_histogram_features(pts[0, :, :4]))
    assert int(ops._hist_ws(B, torch.device(dev)).abs().sum()) == 0, "the counter workspace must be all zero between calls"
    # the three-launch entry point without a workspace (counters in the output buffer)
    from transfuser_amd._lib import ptr, stream_of, check
    pt, nt = torch.from_numpy(pts).to(dev), torch.from_numpy(npts).to(dev)
    out3 = torch.empty(B, 2, 256, 256, dtype=torch.float32, device=dev)
    check(ops.L().tf_lidar_hist_f32(ptr(pt), ptr(nt), B, N, stride, ptr(out3), stream_of(pt)), "tf_lidar_hist_f32")
    assert np.array_equal(out3.cpu().numpy(), out)


def check_correspondences(dev):
    """lidar_bev_cam_correspondences on the GPU (csrc/correspond.cpp) == the oracle (pinned to data.py:632-842 by tests/test_oracle_pinning_data.py),
    integer outputs compared for equality INCLUDING the crowded cells (the kernel and the oracle share the counter-based draw): the golden
    clouds with their edge points, ragged num_points over a padded buffer (keys count in units of the buffer length), the loader's
    y-negated layout, a point stride of 4, several seeds; and an empty sample."""
    import sys as _sys, os as _os
    _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    from oracle import correspondences as oc
    clouds = mg.correspondence_clouds()
    N = 6400
    buf = np.zeros((4, N, 4), np.float32)
    nums = [len(clouds["sparse"]), len(clouds["dense"]), 3000, 0]
    buf[0, :nums[0], :3] = clouds["sparse"]; buf[1, :nums[1], :3] = clouds["dense"]; buf[2, :nums[2], :3] = clouds["dense"][500:3500]
    buf[3, :, :3] = clouds["dense"][:1].repeat(N, 0)          # sample 3 has num_points 0: its padding must be ignored
    buf[..., 3] = 0.5
    for seed, yneg in ((0, False), (12345, True)):
        b2 = buf.copy()
        if yneg:
            b2[..., 1] *= -1
        bev, cam = ops.lidar_cam_correspondences(torch.from_numpy(b2).to(dev), torch.tensor(nums, dtype=torch.int32, device=dev), seed=seed, y_negated=yneg)
        bev, cam = bev.cpu().numpy(), cam.cpu().numpy()
        for s in range(4):
            wb, wc = oc.lidar_bev_cam_correspondences(buf[s, :nums[s], :3], seed=seed, sample=s, key_stride=N)
            assert np.array_equal(bev[s], wb), ("bev_points", seed, s, int((bev[s] != wb).sum()))
            assert np.array_equal(cam[s], wc), ("cam_points", seed, s, int((cam[s] != wc).sum()))
    # stride 3, no num_points: every row counts
    c3 = np.ascontiguousarray(clouds["dense"][:4096])
    bev, cam = ops.lidar_cam_correspondences(torch.from_numpy(c3[None]).to(dev), None, seed=7)
    wb, wc = oc.lidar_bev_cam_correspondences(c3, seed=7, sample=0)
    assert np.array_equal(bev[0].cpu().numpy(), wb) and np.array_equal(cam[0].cpu().numpy(), wc)


# ---------------------------------------------------------------- CenterNet targets + losses vs the oracle
def synthetic_labels(B, seed=0, crowded=False):
    rng = np.random.default_rng(seed)
    label = torch.zeros(B, 20, 7)
    for b in range(B):
        k = int(rng.integers(0, 9)) if not crowded else 20
        if b == 0:
            k = max(k, 3)
        if k:
            label[b, :k, 0:2] = torch.from_numpy(rng.uniform(1, 254, (k, 2))).float()
            label[b, :k, 2:4] = torch.from_numpy(rng.uniform(8, 40, (k, 2))).float()
            label[b, :k, 4] = torch.from_numpy(rng.uniform(-math.pi, math.pi, k)).float()
            label[b, :k, 5] = torch.from_numpy(rng.uniform(0, 8, k)).float()
            label[b, :k, 6] = torch.from_numpy(rng.integers(0, 2, k)).float()
    label[0, 1, 0:2] = label[0, 0, 0:2] + 0.5   # two boxes landing in the same cell (later one wins)
    label[0, 2, 0:2] = torch.tensor([2.0, 253.0])  # splat clipped by the map border
    return label


def check_centernet(dev, B, crowded=False):
    from oracle.centernet import LidarCenterNetHead
    from transfuser_amd.config import GlobalConfig
    cfg = GlobalConfig()
    head = LidarCenterNetHead(64, 64, 1, cfg)
    label = synthetic_labels(B, 0, crowded)
    fh = fw = 64
    tgt, af = head.get_targets(label, torch.zeros_like(label[:, :, 0]), label.sum(-1) == 0., (B, 1, fh, fw))
    tgtf, tgti, cnt = ops.centernet_targets(label.to(dev), fh, fw, fw / 256.0, fh / 256.0, cfg.num_dir_bins)
    tf_, ti = tgtf.cpu(), tgti.cpu()
    assert int(cnt.sum()) == int(tgt['center_heatmap_target'].eq(1).sum()), "avg_factor count"
    close(tf_[..., 0], tgt['center_heatmap_target'][:, 0], tol=1e-6, what="heatmap target")
    assert torch.equal(tf_[..., 0] == 1, tgt['center_heatmap_target'][:, 0] == 1)
    close(tf_[..., 1:3], tgt['wh_target'].permute(0, 2, 3, 1), tol=1e-6, what="wh target")
    close(tf_[..., 3:5], tgt['offset_target'].permute(0, 2, 3, 1), tol=1e-6, what="offset target")
    close(tf_[..., 5], tgt['yaw_res_target'][:, 0], tol=1e-6, what="yaw res target")
    close(tf_[..., 6], tgt['velocity_target'][:, 0], tol=1e-6, what="velocity target")
    assert torch.equal(tf_[..., 7], tgt['wh_offset_target_weight'][:, 0]), "weight map (index scatter) must be exact"
    assert torch.equal(ti[..., 0].long(), tgt['yaw_class_target']), "yaw class (integer) must be exact"
    assert torch.equal(ti[..., 1].long(), tgt['brake_target']), "brake (integer) must be exact"
    # losses + gradients
    P = 9 + cfg.num_dir_bins
    pred = (R(B, fh, fw, P, dev="cpu") * 1.5).requires_grad_(True)
    nchw = lambda a, b_: pred[..., a:b_].permute(0, 3, 1, 2)
    nb = cfg.num_dir_bins
    preds = (nchw(0, 1).sigmoid(), nchw(1, 3), nchw(3, 5), nchw(5, 5 + nb), nchw(5 + nb, 6 + nb), nchw(6 + nb, 7 + nb), nchw(7 + nb, 9 + nb))
    ref = head.loss(preds, label, torch.zeros_like(label[:, :, 0]), label.sum(-1) == 0.)
    names = ['loss_center_heatmap', 'loss_wh', 'loss_offset', 'loss_yaw_class', 'loss_yaw_res', 'loss_velocity', 'loss_brake']
    losses = ops.centernet_loss_fwd(pred.detach().to(dev), tgtf, tgti, cnt, nb)
    for i, k in enumerate(names):
        close(losses[i], ref[k], what=k)
    gup = torch.tensor([1.0, 0.2, 0.2, 0.2, 0.2, 0.5, 0.3])
    total = sum(gup[i] * ref[k] for i, k in enumerate(names))
    (gp,) = torch.autograd.grad(total, [pred])
    dpred = ops.centernet_loss_bwd(pred.detach().to(dev), tgtf, tgti, cnt, gup.to(dev), nb)
    close(dpred, gp, tol=1e-4, what="centernet dpred")


# ---------------------------------------------------------------- every engine tiling (BM x BN x BK, split-K)
ENGINE_PLANS = [(128, 128, 16, 1), (128, 128, 32, 1), (128, 96, 16, 1), (128, 96, 32, 1), (128, 64, 32, 1), (128, 32, 32, 1), (64, 128, 32, 1),
                (64, 64, 32, 1), (64, 64, 16, 3), (128, 32, 16, 2), (128, 96, 32, 2)]


def check_engine_plan(dev, bm, bn, bk, splitk):
    ops.force_plan(bm, bn, bk, splitk)
    try:
        check_gemm(dev, 130, 216, 40)
        check_gemm(dev, 200, 90, 150)
        check_attention(dev, 1, 2, 50, 54)
        check_conv(dev, 2, 11, 13, 48, 48, 3, 2, 2)
        check_conv(dev, 1, 9, 9, 72, 72, 3, 1, 1)
        check_stem(dev, 1, 12, 20)
    finally:
        ops.force_plan(0)


# ---------------------------------------------------------------- LDS-DMA GEMM configurations (tf_gemm_dma.h), kind 1..5 (+ split-K)
DMA_KINDS = [(1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (2, 3), (1, 2), (6, 1), (7, 1), (8, 2), (2, 320), (6, 160)]   # (kind, atomic split-K count); counts >= 100 once collided with the two-pass encoding
DMA_SHAPES_SMALL = [(130, 216, 40), (200, 92, 152), (70, 36, 20), (64, 64, 2600)]
DMA_SHAPES_GPU = [(1740, 1512, 576), (333, 700, 1028), (130, 216, 40), (64, 64, 16), (1740, 216, 864)]


def check_gemm_dma(dev, kind, splitk, shapes):
    """Every operand layout (nt / nn / tn / tt) of the LDS-DMA kernels with ragged M / N / K (k % 16 != 0: zero-filled tails through the
    buffer bounds check), the full epilogue (bias + residual + ReLU, masked store, accumulate, split-K atomics) and batched strides."""
    ops.force_dma(kind, splitk)
    try:
        for (m, n, k) in shapes:
            check_gemm(dev, m, n, k)                       # nt fwd (bias+res+relu), nn dgrad (+ strided view), tn wgrad accumulate (split-K)
            check_gemm_mask(dev, m, n, k)
            a, b = R(m, k, dev=dev), R(k, n, seed=4, dev=dev)
            c = torch.empty(m, n, device=dev)
            ops.gemm(a, b, c, m, n, k, k, n, n, b_trans=True)                                  # nn
            close(c, a @ b, what="dma nn")
            at = a.t().contiguous()                                                            # (k, m)
            ops.gemm(at, b.t().contiguous(), c, m, n, k, m, k, n, a_trans=True)               # tt: A^T stored [k][m], B stored [n][k]
            close(c, a @ b, what="dma tt")
            ops.gemm(at, b, c, m, n, k, m, n, n, a_trans=True, b_trans=True, alpha=0.5)      # tn
            close(c, 0.5 * (a @ b), what="dma tn")
        # batched (outer, inner) strides as the attention GEMMs use them, 16-byte aligned head slices
        B_, nh, T, hs = 2, 2, 52, 24
        C = nh * hs
        qkv = R(B_, T, 3 * C, dev=dev, scale=0.5)
        att = torch.zeros(B_ * nh, T, T, device=dev)
        q, kk = qkv[..., :C], qkv[..., C:2 * C]
        sa = (T * 3 * C, hs)
        ops.gemm(q, kk, att, T, T, hs, 3 * C, 3 * C, T, alpha=0.25, batch=B_ * nh, inner=nh, sa=sa, sb=sa, sc=(nh * T * T, T * T))
        qh, kh = [t.reshape(B_, T, nh, hs).transpose(1, 2) for t in (q, kk)]
        close(att.view(B_, nh, T, T), 0.25 * (qh @ kh.transpose(-2, -1)), what="dma batched")
    finally:
        ops.force_plan(0)


# ---------------------------------------------------------------- inference decode
DECODE_CASES = [(2, 64, 64, 12, 100), (1, 16, 20, 12, 30), (3, 8, 8, 4, 64)]


def check_centernet_decode(dev, B, fh, fw, nbins, k):
    """decode_heatmap kernel vs the oracle restatement (mmdet get_local_maximum / get_topk_from_heatmap / gather): all k rows in
    order wherever the scores are distinct and positive; suppressed cells (score 0) can appear in any order at the tail."""
    from oracle import centernet as oc
    P = 9 + nbins
    pred = R(B, fh, fw, P, dev="cpu", scale=2.0)
    g = torch.Generator().manual_seed(11)
    pred[..., 0] = torch.randn(B, fh, fw, generator=g) * 2.0
    pred[0, :2, :2, 0] = 3.0                          # a plateau of equal maxima: all four survive the NMS (hmax == heat)
    nchw = pred.permute(0, 3, 1, 2)
    preds = (torch.sigmoid(nchw[:, 0:1]), nchw[:, 1:3], nchw[:, 3:5], nchw[:, 5:5 + nbins], nchw[:, 5 + nbins:6 + nbins],
             nchw[:, 6 + nbins:7 + nbins], nchw[:, 7 + nbins:9 + nbins])
    want, _ = oc.decode_heatmap(preds, nbins, k=min(k, fh * fw), kernel=3)
    got = ops.centernet_decode(pred.to(dev), nbins, min(k, fh * fw), 3, 4.0).cpu()
    assert got.shape == want.shape
    for b in range(B):
        close(got[b, :, 7], want[b, :, 7], what="decode scores", tol=1e-6)          # the sorted score lists agree
        sw = want[b, :, 7]
        distinct = torch.ones_like(sw, dtype=torch.bool)
        distinct[1:] &= sw[1:] != sw[:-1]
        distinct[:-1] &= sw[:-1] != sw[1:]
        sel = distinct & (sw > 0)
        assert int(sel.sum()) >= 1
        close(got[b][sel], want[b][sel], what="decoded boxes", tol=1e-5)
        tie = (~distinct) & (sw > 0)              # tied maxima: same set of boxes, any order
        if tie.any():
            a = got[b][tie]; w = want[b][tie]
            assert all(((a - r).abs().sum(1) < 1e-4).any() for r in w), "tied boxes differ as a set"


def check_gemm_mask(dev, M, N, K):
    """dX = (dY W) * [act > 0]: the ReLU mask fused into the dgrad epilogue (tf_gemm_desc.mask)."""
    dy, w, act = R(M, N, dev=dev), R(N, K, seed=1, dev=dev) * 0.1, R(M, K, seed=2, dev=dev)
    want = (dy @ w) * (act > 0)
    got = ops.linear_dgrad(dy, w, mask=act)
    close(got, want, what="masked dgrad")


def check_gemm_dropout(dev, M, N, K):
    """y = res + dropout(x W^T + b) in the GEMM's own epilogue (tf_gemm_desc.drop_seed: the Block's x + resid_drop(proj(.)) / x + mlp(.),
    transfuser.py:543-549) == the separate tf_dropout_add_f32 launch on the same product, BITWISE (same mask, same order of operations), under
    the planner's choice, pinned engine tilings, every LDS-DMA configuration and a two-pass split-K plan (the fix-up pass applies the mask)."""
    x, w, b, r = R(M, K, dev=dev), R(N, K, seed=1, dev=dev) * 0.1, R(N, seed=2, dev=dev), R(M, N, seed=3, dev=dev)
    seed = torch.tensor([4321], dtype=torch.int32, device=dev)
    site, p = 5, 0.1

    def one(what):
        want = ops.dropout_add(ops.linear_fwd(x, w, b), r, seed, site, p)
        got = ops.linear_fwd(x, w, b, res=r, drop=(seed, site, p))
        assert torch.equal(got, want), "dropout epilogue (%s): max diff %.3e" % (what, (got - want).abs().max().item())
        kept = (got != r).float().mean().item()
        assert abs(kept - (1 - p)) < 0.05, kept
    try:
        one("planned")
        for bm, bn, bk in ((64, 64, 16), (128, 32, 32), (64, 128, 16)):
            ops.force_plan(bm, bn, bk, 1)
            one("engine %dx%dx%d" % (bm, bn, bk))
        for kind in (1, 2, 3, 4, 5):
            for sk in (1, 2):
                ops.force_dma(kind, sk)
                one("dma %d split %d" % (kind, sk))
    finally:
        ops.force_plan(0)
    y0 = ops.linear_fwd(x, w, b, res=r, drop=(seed, site, 0.0))
    assert torch.equal(y0, ops.linear_fwd(x, w, b, res=r)), "p = 0 keeps everything"


# ---------------------------------------------------------------- the benchmarked shapes under the shipped (tuned) plans, CPU fp32 reference
BENCH_GEMMS = [(1740, 6048, 1512), (1740, 1512, 6048), (1740, 4536, 1512), (1740, 1512, 1512), (7040, 576, 576), (2560, 576, 576), (28160, 216, 216)]


def check_bench_gemm(dev, m, n, k):
    """GPT-4 / trunk GEMMs of the B=10, 256x704 step exactly as bench.py launches them: the tilings of
    transfuser_amd/plans/mi355x.txt are loaded (the caller does it), and the reference is PyTorch-CPU fp32 - the reference's own
    arithmetic path (north_star) - not rocBLAS.  fwd (bias+ReLU), dgrad (+fused ReLU mask), wgrad (accumulate, split-K where tuned)."""
    x, w, b = R(m, k, dev="cpu"), R(n, k, dev="cpu") * (1.0 / math.sqrt(k)), R(n, dev="cpu")
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    y = ops.linear_fwd(xd, wd, bd, relu=True)
    close(y, torch.relu(x @ w.t() + b), tol=1e-4, what="bench fwd %dx%dx%d" % (m, n, k))
    dy = R(m, n, seed=1, dev="cpu")
    dyd = dy.to(dev)
    act = R(m, k, seed=2, dev="cpu")
    close(ops.linear_dgrad(dyd, wd, mask=act.to(dev)), (dy @ w) * (act > 0), tol=1e-4, what="bench dgrad")
    dw0 = R(n, k, seed=3, dev="cpu") * 0.1
    dw = ops.linear_wgrad(dyd, xd, dw0.to(dev), accumulate=True)
    close(dw, dw0 + dy.t() @ x, tol=1e-4, what="bench wgrad")


BENCH_CONVS = [(10, 256, 704, 32, 32), (10, 256, 704, 32, 7), (10, 256, 704, 32, 1), (10, 64, 176, 64, 32), (10, 16, 44, 576, 576)]


def check_bench_conv(dev, B, H, W, Cin, Cout):
    """Decoder-tail / grouped trunk 3x3 convolutions at the bench batch (B=10, 256x704 maps) vs F.conv2d on the CPU."""
    groups = Cin // 24 if Cin == 576 else 1
    x = R(B, Cin, H, W, dev="cpu").requires_grad_(True)
    w = (R(Cout, Cin // groups, 3, 3, dev="cpu") * 0.1).requires_grad_(True)
    b = R(Cout, dev="cpu")
    y = F.conv2d(x, w, b, 1, 1, 1, groups)
    dy = R(*y.shape, seed=1, dev="cpu")
    gx, gw = torch.autograd.grad(y, [x, w], dy)
    xh, wh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev), cl(w.detach()).to(dev)
    yh = ops.conv_fwd(xh, wh, b.to(dev), 1, None, groups, relu=False)
    close(yh.permute(0, 3, 1, 2), y, tol=1e-4, what="bench conv fwd")
    dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    close(ops.conv_dgrad(dyh, wh, xh.shape, 1, None, groups).permute(0, 3, 1, 2), gx, tol=1e-4, what="bench conv dgrad")
    dw = torch.zeros_like(wh)
    ops.conv_wgrad(dyh, xh, dw, 1, None, groups)
    close(dw, gw, tol=2e-4, what="bench conv wgrad")


# ---------------------------------------------------------------- bf16-MFMA compute mode (tf_set_precision(1), BASELINE configs[2])
BF16_PLANS = [("plan", (64, 64, 16, 1)), ("plan", (128, 128, 32, 1)), ("plan", (128, 32, 16, 2)), ("plan", (64, 128, 32, 1)), ("dma", (1, 1)), ("dma", (2, 2)),
              ("dma", (4, 1)), ("dma", (5, 1)), ("dma", (6, 1)), ("dma", (7, 2)), ("dma", (8, 1))]


def _bf(t):
    return t.bfloat16().float()


def check_bf16_mode(dev, kind, plan, shapes=((130, 216, 40), (200, 92, 152)), mode="bf16"):
    """Engine contractions in bf16-MFMA mode: every operand is rounded to bf16 (round-to-nearest-even) on its way into the MFMA, products
    are exact, accumulation and epilogue stay fp32 - so the result must equal an fp32 GEMM of the bf16-rounded operands to fp32 summation
    accuracy (tolerance 1e-4, NOT a loose 'bf16 tolerance').  All layouts, the masked (im2col) loaders, batched non-vector operands.
    mode="fp16": the same contract with IEEE-half operands (tf_set_precision(3))."""
    ops.set_precision(mode)
    _bf = (lambda t: t.bfloat16().float()) if mode == "bf16" else (lambda t: t.half().float())
    (ops.force_dma if kind == "dma" else ops.force_plan)(*plan)
    try:
        for (m, n, k) in shapes:
            x, w, b, r = R(m, k, dev=dev), R(n, k, dev=dev), R(n, dev=dev), R(m, n, dev=dev)
            close(ops.linear_fwd(x, w, b, relu=True, res=r), torch.relu(_bf(x) @ _bf(w).t() + b + r), tol=1e-4, what="bf16 fwd")
            dy = R(m, n, seed=1, dev=dev)
            close(ops.linear_dgrad(dy, w), _bf(dy) @ _bf(w), tol=1e-4, what="bf16 dgrad")
            dw0 = R(n, k, seed=2, dev=dev)
            close(ops.linear_wgrad(dy, x, dw0.clone(), accumulate=True), dw0 + _bf(dy).t() @ _bf(x), tol=1e-4, what="bf16 wgrad")
            a, bb = R(m, k, dev=dev), R(n, k, seed=4, dev=dev)
            c = torch.empty(m, n, device=dev)
            ops.gemm(a.t().contiguous(), bb, c, m, n, k, m, k, n, a_trans=True)                 # tt
            close(c, _bf(a) @ _bf(bb).t(), tol=1e-4, what="bf16 tt")
        # grouped / strided 3x3 convolution through the im2col loaders
        B, Hi, Wi, Cin, Cout, groups, stride = 2, 9, 11, 48, 48, 2, 2
        x = R(B, Cin, Hi, Wi, dev="cpu").requires_grad_(True)
        w = (R(Cout, Cin // groups, 3, 3, dev="cpu") * 0.1).requires_grad_(True)
        y = F.conv2d(_bf(x), _bf(w), None, stride, 1, 1, groups)
        xh, wh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev), cl(w.detach()).to(dev)
        close(ops.conv_fwd(xh, wh, None, stride, None, groups).permute(0, 3, 1, 2), y, tol=1e-4, what="bf16 conv fwd")
        dy = R(*y.shape, seed=1, dev="cpu")
        gx = torch.autograd.grad(F.conv2d(x, _bf(w), None, stride, 1, 1, groups), x, _bf(dy))[0]
        dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        close(ops.conv_dgrad(dyh, wh, xh.shape, stride, None, groups).permute(0, 3, 1, 2), gx, tol=1e-4, what="bf16 conv dgrad")
        gw = torch.autograd.grad(F.conv2d(_bf(x), w, None, stride, 1, 1, groups), w, _bf(dy))[0]
        dw = torch.zeros_like(wh)
        ops.conv_wgrad(dyh, xh, dw, stride, None, groups)
        close(dw, gw, tol=1e-4, what="bf16 conv wgrad")
        # attention-style batched GEMM with head size 18 (element-wise loaders)
        B_, nh, T, hs = 1, 2, 50, 18
        C = nh * hs
        qkv = R(B_, T, 3 * C, dev=dev, scale=0.5)
        att = torch.zeros(B_ * nh, T, 52, device=dev)
        q, kk = qkv[..., :C], qkv[..., C:2 * C]
        sa = (T * 3 * C, hs)
        ops.gemm(q, kk, att, T, T, hs, 3 * C, 3 * C, 52, alpha=0.5, batch=B_ * nh, inner=nh, sa=sa, sb=sa, sc=(nh * T * 52, T * 52))
        qh, kh = [_bf(t).reshape(B_, T, nh, hs).transpose(1, 2) for t in (q, kk)]
        close(att[:, :, :T].reshape(B_, nh, T, T), 0.5 * (qh @ kh.transpose(-2, -1)), tol=1e-4, what="bf16 batched")
    finally:
        ops.force_plan(0)
        ops.set_precision("fp32")


# ---------------------------------------------------------------- f32x3 compute mode (tf_set_precision(2)): bf16x3 split, fp32-ACCURATE
def check_lowp16_storage(dev, mode):
    """16-bit operand STORAGE path: tf_cast16_f32 (row-major + transposed, zero-padded copies) is torch's round-to-nearest-even cast bit for
    bit; tf_gemm16_nt_f32 equals an fp32 GEMM of the rounded operands (1e-4) for every tile kind, ragged M / N / K tails and every epilogue
    form; the three products of a linear layer (y = x W^T, dx = dy W, dW += dy^T x) through the transposed copies."""
    ops.set_precision(mode)
    t16 = torch.bfloat16 if mode == "bf16" else torch.float16
    rnd = lambda t: t.to(t16).float()
    try:
        assert ops.lowp_storage() == (1 if mode == "bf16" else 2)
        for rows, cols, pad in ((70, 40, 0), (130, 216, 8), (5, 8, 0), (64, 64, 0), (1, 24, 0)):
            x = R(rows, cols + pad, dev=dev, scale=3.0)[:, :cols]              # row stride > cols
            y, yt = ops.cast16(x)
            assert y.dtype == t16 and torch.equal(y.cpu(), x.cpu().to(t16)), "cast16 row-major"
            rows8 = (rows + 7) // 8 * 8
            assert yt.shape == (cols, rows8) and torch.equal(yt[:, :rows].cpu(), x.cpu().to(t16).t()), "cast16 transposed"
            assert bool((yt[:, rows:].float() == 0).all()), "cast16: the pad rows of the transposed copy must be zero"
        for kind in (0, 1, 2, 3, 5):
            for (m, n, k) in ((130, 216, 40), (200, 96, 152), (33, 72, 8), (260, 136, 264)):
                x, w, b, r = R(m, k, dev=dev), R(n, k, seed=3, dev=dev), R(n, seed=4, dev=dev), R(m, n, seed=5, dev=dev)
                x16, x16t = ops.cast16(x)
                w16, w16t = ops.cast16(w)
                ref = rnd(x) @ rnd(w).t()
                out = torch.empty(m, n, device=dev)
                close(ops.gemm16_nt(x16, w16, out, kind=kind), ref, tol=1e-4, what="gemm16 plain kind %d" % kind)
                close(ops.gemm16_nt(x16, w16, torch.empty(m, n, device=dev), bias=b, res=r, relu=True, kind=kind), torch.relu(ref + b + r), tol=1e-4, what="gemm16 bias res relu")
                base = R(m, n, seed=6, dev=dev)
                close(ops.gemm16_nt(x16, w16, base.clone(), accumulate=True, alpha=0.5, kind=kind), base + 0.5 * ref, tol=1e-4, what="gemm16 accumulate alpha")
                msk = R(m, n, seed=7, dev=dev)
                assert torch.equal(ops.gemm16_nt(x16, w16, torch.empty(m, n, device=dev), mask=msk, kind=kind), ops.gemm16_nt(x16, w16, torch.empty(m, n, device=dev), kind=kind) * (msk > 0)), "gemm16 mask"
                # the linear-layer trio
                dy = R(m, n, seed=8, dev=dev)
                d16, d16t = ops.cast16(dy)
                close(ops.gemm16_nt(d16, w16t, torch.empty(m, k, device=dev), k=n, kind=kind), rnd(dy) @ rnd(w), tol=1e-4, what="gemm16 dgrad")
                dw0 = R(n, k, seed=9, dev=dev)
                close(ops.gemm16_nt(d16t, x16t, dw0.clone(), accumulate=True, k=d16t.shape[1], kind=kind), dw0 + rnd(dy).t() @ rnd(x), tol=1e-4, what="gemm16 wgrad")
        # the RegNetY 1x1 convolutions on stored operands: BatchNorm statistics from the epilogue, and the weight gradient over MANY rows
        # (tiny output, long contraction: k-slices that atomically add)
        old_fuse = ops.FUSE_BN_STATS
        ops.FUSE_BN_STATS = True
        try:
            for (m, n, k) in ((2100, 72, 48), (203, 216, 72)):
                x, w = R(m, k, dev=dev) + 2.0, R(n, k, seed=3, dev=dev) * 0.2
                x16, x16t = ops.cast16(x)
                w16, w16t = ops.cast16(w)
                y, cs = ops.gemm16_nt_colstat(x16, w16, torch.empty(m, n, device=dev))
                close(y, rnd(x) @ rnd(w).t(), tol=1e-4, what="gemm16 with colstat")
                assert cs is not None
                _bn_from_parts(dev, y.view(1, 1, m, n), cs, relu=True)
                dy = R(m, n, seed=8, dev=dev)
                d16, d16t = ops.cast16(dy)
                dw0 = R(n, k, seed=9, dev=dev)
                close(ops.gemm16_nt(d16t, x16t, dw0.clone(), accumulate=True, k=d16t.shape[1]), dw0 + rnd(dy).t() @ rnd(x), tol=1e-4, what="gemm16 wgrad (k-split)")
        finally:
            ops.FUSE_BN_STATS = old_fuse
    finally:
        ops.set_precision("fp32")


def check_lowp16_fused_producers(dev, mode):
    """Round 5: the 16-bit operand copies written by their PRODUCERS instead of cast launches - bitwise equal to the separate launches.
    (1) tf_layernorm_fwd16_f32 == tf_cast16_f32(tf_layernorm_fwd_f32(x)): both copies, the zero pad rows of the transposed one, mean / rstd; row counts
        around the 8-row blocks, every register-row width (C up to 2048).  (2) tf_cast16_multi_f32 == one tf_cast16_f32 per matrix (ragged shapes)."""
    ops.set_precision(mode)
    try:
        for rows, C in ((61, 72), (64, 216), (8, 576), (1, 24), (349, 1512), (13, 2048), (20, 1024), (9, 260)):
            x = R(rows, C, dev=dev, scale=2.0) + 0.5
            g, b = R(C, seed=1, dev=dev), R(C, seed=2, dev=dev)
            assert ops.layernorm_fwd16_ok(x, g, b)
            h, m, r = ops.layernorm_fwd(x, g, b, 1e-5)
            y, yt = ops.cast16(h)
            y2, yt2, m2, r2 = ops.layernorm_fwd16(x, g, b, 1e-5)
            assert torch.equal(m, m2) and torch.equal(r, r2), "fwd16 statistics"
            assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), "fwd16 row-major copy (%d, %d)" % (rows, C)
            assert yt2.shape == yt.shape and torch.equal(yt.view(torch.int16), yt2.view(torch.int16)), "fwd16 transposed copy (%d, %d)" % (rows, C)
            assert bool((yt2[:, rows:].float() == 0).all()), "fwd16: the pad rows of the transposed copy must be zero"
        assert not ops.layernorm_fwd16_ok(R(4, 2052, dev=dev), R(2052, dev=dev), R(2052, dev=dev))
        # many matrices, one launch: the Engine's weight refresh
        ws = [R(n, k, seed=i, dev=dev) for i, (n, k) in enumerate(((216, 72), (72, 216), (70, 40), (1, 8), (130, 264), (64, 64)))]
        for w in ws:
            ops.lowp_weight(w)                                 # registers (w, w16, w16t) in the cache
        for w in ws:
            w.mul_(1.5).add_(0.25)                             # "AdamW"
        old = ops.CAST16_MULTI
        try:
            ops.CAST16_MULTI = True
            with ops.lowp_managed():
                ops.lowp_refresh_weights()
                got = [tuple(t.clone() for t in ops.lowp_weight(w)) for w in ws]
        finally:
            ops.CAST16_MULTI = old
        for w, (g16, g16t) in zip(ws, got):
            y, yt = ops.cast16(w)
            assert torch.equal(y.view(torch.int16), g16.view(torch.int16)) and torch.equal(yt.view(torch.int16), g16t.view(torch.int16)), "cast16_multi %s" % (tuple(w.shape),)
    finally:
        ops.set_precision("fp32")


def check_lowp16_conv_producers(dev, mode):
    """Round 5: the element-wise producers of the bottleneck 1x1-convolution operands write the 16-bit copies themselves (tile16_kernel) - every copy is
    bitwise tf_cast16_f32 of the fp32 kernel it stands in for (ragged row counts around the 64-row tiles, C around the 64-column tiles, zero pad rows), the
    fp32 side outputs (block output, shortcut gradient, dgamma / dbeta) are those of the fp32 kernels."""
    ops.set_precision(mode)
    old_dbg = ops._DBG_LEGACY_BNB
    ops._DBG_LEGACY_BNB = True          # fp32 comparison kernels on the chunk-partials + finalize path (no atomics): the path the 16-bit entry points use
    same = lambda a, b, what: (_ for _ in ()).throw(AssertionError(what)) if not torch.equal(a.view(torch.int16), b.view(torch.int16)) else None
    try:
        for (B, H, W, C) in ((2, 5, 7, 72), (1, 8, 8, 216), (3, 4, 11, 24), (1, 1, 1, 8), (2, 16, 5, 136)):
            rows = B * H * W
            x = R(B, H, W, C, dev=dev, scale=1.5)
            g, b = R(C, seed=1, dev=dev).abs() + 0.5, R(C, seed=2, dev=dev)
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            # forward statistics as a producing GEMM would gather them
            xin, w = R(rows, 40, seed=3, dev=dev), R(C, 40, seed=4, dev=dev) * 0.3
            y, cs = ops.linear_fwd(xin, w, colstat=True)
            if cs is None:
                continue
            y = y.view(B, H, W, C)
            res = R(B, H, W, C, seed=5, dev=dev)
            want, sm, si = ops.bn_fwd_parts(y, cs, g, b, rm.clone(), rv.clone(), res, True)
            coef, sm2, si2 = ops.bn_finalize_parts(cs, g, b, rm.clone(), rv.clone())
            assert torch.equal(sm, sm2) and torch.equal(si, si2)
            y32, y16, y16t = ops.bn_apply16(y, coef, res, True)
            assert torch.equal(y32, want), "bn_apply16 fp32 output"
            w16, w16t = ops.cast16(want.view(rows, C))
            same(y16, w16, "bn_apply16 row-major"); same(y16t, w16t, "bn_apply16 transposed")
            _, n16, n16t = ops.bn_apply16(y, coef, None, False, want_f32=False)
            p16, p16t = ops.cast16(ops.bn_fwd_parts(y, cs, g, b, rm.clone(), rv.clone(), None, False)[0].view(rows, C))
            same(n16, p16, "bn_apply16 (no res / relu)"); same(n16t, p16t, "bn_apply16 (no res / relu) transposed")
            # BatchNorm apply + ReLU + SE scale
            gate = R(B, C, seed=6, dev=dev)
            z16, z16t = ops.se_scale_bn16(y, coef, gate)
            q16, q16t = ops.cast16(ops.se_scale_bn_fwd(y, coef, gate).view(rows, C))
            same(z16, q16, "se_scale_bn16 row-major"); same(z16t, q16t, "se_scale_bn16 transposed")
            # BatchNorm backward (ReLU mask from the stored output) + shortcut gradient
            dz = R(B, H, W, C, seed=7, dev=dev)
            dg0, db0, dg1, db1 = (torch.zeros(C, device=dev) for _ in range(4))
            dx, dres = ops.bn_bwd(dz, want, y, g, sm, si, dg0, db0, want_dres=True)
            dx32, d16, d16t, dres2 = ops.bn_bwd16(dz, want, y, g, sm, si, dg1, db1, want_dres=True, want_f32=True)
            assert torch.equal(dx32, dx) and torch.equal(dres2, dres) and torch.equal(dg0, dg1) and torch.equal(db0, db1), "bn_bwd16 fp32 outputs"
            e16, e16t = ops.cast16(dx.view(rows, C))
            same(d16, e16, "bn_bwd16 row-major"); same(d16t, e16t, "bn_bwd16 transposed")
            # BatchNorm backward with the mask recomputed from the raw input
            dg0.zero_(); db0.zero_(); dg1.zero_(); db1.zero_()
            dxr = ops.bn_bwd_remask(dz, y, coef, g, sm, si, dg0, db0)
            _, r16, r16t = ops.bn_bwd_remask16(dz, y, coef, g, sm, si, dg1, db1)
            assert torch.equal(dg0, dg1) and torch.equal(db0, db1), "bn_bwd_remask16 parameter gradients"
            f16_, f16t = ops.cast16(dxr.view(rows, C))
            same(r16, f16_, "bn_bwd_remask16 row-major"); same(r16t, f16t, "bn_bwd_remask16 transposed")
            assert bool((r16t[:, rows:].float() == 0).all()), "pad rows of the transposed copy must be zero"
    finally:
        ops._DBG_LEGACY_BNB = old_dbg
        ops.set_precision("fp32")


def check_lowp16_conv_stage(dev, mode):
    """A RegNetY stage (stride-2 bottleneck + two stride-1 bottlenecks, 48 -> 72 channels, group width 24) in a 16-bit storage mode: the 1x1
    convolutions on STORED operands whose copies their producers write (ops.STORE16_CONV, functions.YBlockFn "lp") against the in-register rounding
    path of the same mode - the same operands rounded to the same 16-bit values, so output and input gradient agree to fp32 summation order (bitwise on
    the emulator) and every parameter gradient to 1e-5; and the copies really are handed from block to block: ONE activation cast launch (the stage input)."""
    from transfuser_amd import regnet
    torch.manual_seed(0)
    st = regnet.RegStage(48, 72, 3, 24, 0.25)
    for m in st.modules():
        if isinstance(m, regnet.Bottleneck):
            torch.nn.init.normal_(m.conv3.bn.weight, 1.0, 0.1)
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    st = st.to(dev).train()
    x0, dy = R(2, 16, 20, 48, dev=dev), R(2, 8, 10, 72, seed=1, dev=dev)
    old, orig = ops.STORE16_CONV, ops.cast16
    res, casts = {}, {}
    try:
        for on in (True, False):
            ops.STORE16_CONV = on
            ops.set_precision(mode)
            n = [0]

            def counting(*a, **k):
                n[0] += 1
                return orig(*a, **k)
            for q in st.parameters():
                q.grad = None
            x = x0.clone().requires_grad_(True)
            with ops.lowp_managed():        # as inside train.Engine: the 16-bit weight copies are made once, not at every use
                for blk in st.children():
                    for w in (blk.conv1.conv.weight, blk.conv3.conv.weight):
                        ops.lowp_weight(w.detach().view(w.shape[0], w.shape[1]))
                ops.cast16 = counting
                y = st(x)
                y.backward(dy)
                ops.cast16 = orig
            res[on], casts[on] = (y.detach().clone(), x.grad.clone(), {k: q.grad.clone() for k, q in st.named_parameters()}), n[0]
    finally:
        ops.cast16, ops.STORE16_CONV = orig, old
        ops.set_precision("fp32")
    assert casts[True] == 1 and casts[False] == 0, casts
    a, b = res[True], res[False]
    rel = lambda u, v: float((u - v).norm() / v.norm().clamp_min(1e-12))
    # MI355X: the two paths run different tile plans and k-split atomics (run-to-run summation order), and a ReLU / 16-bit rounding decision that flips on a
    # 1e-7 difference moves a gradient element by its whole size - the emulator pins exactness, the GPU run checks agreement well below any real defect (O(0.1))
    ty, tg = (1e-6, 1e-5) if dev == "cpu" else (1e-4, 2e-2)      # one flipped ReLU decision among 3e4 elements is ~6e-3 of a gradient's norm
    assert rel(a[0], b[0]) <= ty and rel(a[1], b[1]) <= tg, (rel(a[0], b[0]), rel(a[1], b[1]))
    for k in a[2]:
        assert rel(a[2][k], b[2][k]) <= tg, (k, rel(a[2][k], b[2][k]))


def check_layernorm_bwd_drop(dev):
    """tf_layernorm_bwd_drop_f32: dx as tf_layernorm_bwd_f32 (bitwise, with and without accumulation), dropped == tf_dropout_f32(dx) (same mask)."""
    seed = torch.tensor([99], dtype=torch.int32, device=dev)
    for rows, C in ((61, 72), (174, 216), (9, 1512), (5, 30), (33, 2048)):
        x, dy = R(rows, C, dev=dev), R(rows, C, seed=1, dev=dev)
        g, b = R(C, seed=2, dev=dev), R(C, seed=3, dev=dev)
        _, m, r = ops.layernorm_fwd(x, g, b, 1e-5)
        for acc in (False, True):
            base = R(rows, C, seed=4, dev=dev)
            dg0, db0 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            dg1, db1 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            want = ops.layernorm_bwd(dy, x, g, m, r, dg0, db0, dx=base.clone(), accumulate=acc)
            got, dropped = ops.layernorm_bwd(dy, x, g, m, r, dg1, db1, dx=base.clone(), accumulate=acc, drop=(seed, 6, 0.1))
            assert torch.equal(got, want), "ln bwd drop: dx"
            assert torch.equal(dropped, ops.dropout(want, seed, 6, 0.1)), "ln bwd drop: dropped (%d, %d)" % (rows, C)
            close(dg1, dg0, what="ln bwd drop dgamma"); close(db1, db0, what="ln bwd drop dbeta")


def check_conv1x1_s2_dgrad(dev):
    """Input gradient of a 1x1 / stride-2 convolution in accumulate mode (the RegNet downsample branch): plain GEMM + scatter-add path."""
    for (B, Hi, Wi, Cin, Cout) in ((2, 8, 8, 32, 72), (1, 9, 7, 24, 40), (2, 16, 44, 72, 216)):
        x = R(B, Cin, Hi, Wi, dev=dev).requires_grad_(True)
        w = (R(Cout, Cin, 1, 1, dev=dev) * 0.1).requires_grad_(True)
        y = F.conv2d(x, w, None, 2, 0)
        dy = R(*y.shape, seed=1, dev=dev)
        gx = torch.autograd.grad(y, x, dy)[0]
        base = R(B, Hi, Wi, Cin, seed=2, dev=dev)
        got = ops.conv_dgrad(dy.permute(0, 2, 3, 1).contiguous(), w.detach().contiguous(), (B, Hi, Wi, Cin), 2, 0, 1, out=base.clone(), accumulate=True)
        close(got, base + gx.permute(0, 2, 3, 1), what="1x1 s2 dgrad (gemm + scatter)")


def _err64(got, ref64):
    """max |got - ref| / max |ref| against a float64 reference"""
    got = got.detach().double().cpu()
    return ((got - ref64).abs().max() / ref64.abs().max().clamp_min(1e-30)).item()


def check_f32x3_mode(dev, kind, plan, shapes=((130, 216, 40), (200, 92, 152), (96, 160, 1100))):
    """Engine contractions in bf16x3-split mode: every fp32 operand is split exactly into three bf16 terms (x = h + m + l) and the six leading
    partial products are accumulated in fp32 on the bf16 MFMA.  The claim is fp32 ACCURACY, so the reference is float64 and the bound is the
    one the exact fp32-MFMA path itself meets: relative max error <= 4e-6 (K <= 1100: ~3 sqrt(K) 2^-24), and never worse than 3x the fp32-MFMA path's error
    measured on the same inputs (+ 2e-7).  All operand layouts, masked (im2col) loaders, batched element-wise loaders, wide dynamic range."""
    (ops.force_dma if kind == "dma" else ops.force_plan)(*plan)
    BOUND = 4e-6

    def both(fn):
        ops.set_precision("fp32")
        a = fn()
        ops.set_precision("f32x3")
        b = fn()
        return a, b

    def judge(pair, ref64, what):
        e32, e3 = _err64(pair[0], ref64), _err64(pair[1], ref64)
        assert e3 <= BOUND and e3 <= 3.0 * e32 + 2e-7, "%s: f32x3 err %.3e vs fp32-MFMA err %.3e" % (what, e3, e32)

    try:
        for (m, n, k) in shapes:
            # wide dynamic range: per-row / per-column scales over 2^+-20 (a bf16-rounded operand would be off by 4e-3 relative)
            sx = torch.exp2(torch.randint(-20, 21, (m, 1), generator=torch.Generator().manual_seed(m)).float()).to(dev)
            x, w, b, r = R(m, k, dev=dev) * sx, R(n, k, dev=dev), R(n, dev=dev), R(m, n, dev=dev)
            xd, wd = x.double().cpu(), w.double().cpu()
            judge(both(lambda: ops.linear_fwd(x, w, None)), xd @ wd.t(), "x3 fwd (nt)")
            dy = R(m, n, seed=1, dev=dev)
            dyd = dy.double().cpu()
            judge(both(lambda: ops.linear_dgrad(dy, w)), dyd @ wd, "x3 dgrad (nn)")
            judge(both(lambda: ops.linear_wgrad(dy, x, torch.zeros(n, k, device=dev), accumulate=True)), dyd.t() @ xd, "x3 wgrad (tn)")
            a, bb = R(m, k, dev=dev), R(n, k, seed=4, dev=dev)
            at = a.t().contiguous()

            def tt():
                c = torch.empty(m, n, device=dev)
                ops.gemm(at, bb, c, m, n, k, m, k, n, a_trans=True)
                return c
            judge(both(tt), a.double().cpu() @ bb.double().cpu().t(), "x3 tt")
            # epilogue (bias + residual + ReLU) on top of the split product
            ops.set_precision("f32x3")
            y = ops.linear_fwd(x, w, b, relu=True, res=r)
            ref = torch.relu(xd @ wd.t() + b.double().cpu() + r.double().cpu())
            assert _err64(y, ref) <= BOUND, "x3 fwd epilogue"
        # grouped / strided 3x3 convolution through the im2col loaders
        B, Hi, Wi, Cin, Cout, groups, stride = 2, 9, 11, 48, 48, 2, 2
        x = R(B, Cin, Hi, Wi, dev="cpu").double().requires_grad_(True)
        w = (R(Cout, Cin // groups, 3, 3, dev="cpu") * 0.1).double().requires_grad_(True)
        y = F.conv2d(x, w, None, stride, 1, 1, groups)
        dy = R(*y.shape, seed=1, dev="cpu").double()
        gx, gw = torch.autograd.grad(y, [x, w], dy)
        xh, wh = x.detach().float().permute(0, 2, 3, 1).contiguous().to(dev), cl(w.detach().float()).to(dev)
        dyh = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
        judge(both(lambda: ops.conv_fwd(xh, wh, None, stride, None, groups).permute(0, 3, 1, 2)), y.detach(), "x3 conv fwd")
        judge(both(lambda: ops.conv_dgrad(dyh, wh, xh.shape, stride, None, groups).permute(0, 3, 1, 2)), gx, "x3 conv dgrad")

        def wg():
            dw = torch.zeros_like(wh)
            ops.conv_wgrad(dyh, xh, dw, stride, None, groups)
            return dw
        judge(both(wg), gw, "x3 conv wgrad")
        # attention-style batched GEMM with head size 18 (element-wise loaders)
        B_, nh, T, hs = 1, 2, 50, 18
        C = nh * hs
        qkv = R(B_, T, 3 * C, dev=dev, scale=0.5)
        q, kk = qkv[..., :C], qkv[..., C:2 * C]
        sa = (T * 3 * C, hs)

        def bat():
            att = torch.zeros(B_ * nh, T, 52, device=dev)
            ops.gemm(q, kk, att, T, T, hs, 3 * C, 3 * C, 52, alpha=0.5, batch=B_ * nh, inner=nh, sa=sa, sb=sa, sc=(nh * T * 52, T * 52))
            return att[:, :, :T].reshape(B_, nh, T, T)
        qh, kh = [t.double().cpu().reshape(B_, T, nh, hs).transpose(1, 2) for t in (q, kk)]
        judge(both(bat), 0.5 * (qh @ kh.transpose(-2, -1)), "x3 batched")
    finally:
        ops.force_plan(0)
        ops.set_precision("fp32")


# ---------------------------------------------------------------- grouped 3x3 direct kernels (csrc/conv_grouped.cpp), group width 24
GROUPED_CONV_CASES = [(2, 16, 44, 72), (1, 9, 13, 48), (2, 8, 16, 24), (1, 5, 70, 72), (3, 4, 8, 48), (1, 33, 31, 24)]


def check_conv_grouped(dev, B, H, W, C):
    """fwd (+bias, ReLU), dgrad (store and accumulate), wgrad (store and accumulate) of the per-group direct kernels vs F.conv2d; map sizes
    that pick both tile shapes (8x16 / 4x32), ragged borders in both directions, 1..3 groups."""
    groups = C // 24
    assert ops._grouped_ok((B, H, W, C), C, C, 3, 1, 1, groups) or groups == 1
    x = R(B, C, H, W, dev="cpu").requires_grad_(True)
    w = (R(C, 24, 3, 3, dev="cpu") * 0.1).requires_grad_(True)
    b = R(C, dev="cpu")
    y = F.conv2d(x, w, b, 1, 1, 1, groups)
    dy = R(*y.shape, seed=1, dev="cpu")
    gx, gw = torch.autograd.grad(y, [x, w], dy)
    xh, wh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev), cl(w.detach()).to(dev)
    dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    L = ops.L()
    from transfuser_amd.ops import ptr, wptr, stream_of, check
    yh = torch.empty(B, H, W, C, device=dev)
    check(L.tf_conv3x3_grouped_fwd_f32(ptr(xh), wptr(wh), ptr(b.to(dev)), ptr(yh), B, H, W, C, 1, stream_of(xh)), "grouped fwd")
    close(yh.permute(0, 3, 1, 2), torch.relu(y), what="grouped fwd")
    dx = torch.full((B, H, W, C), 0.5, device=dev)
    check(L.tf_conv3x3_grouped_dgrad_f32(ptr(dyh), wptr(wh), ptr(dx), B, H, W, C, 1, stream_of(xh)), "grouped dgrad")
    close(dx.permute(0, 3, 1, 2), gx + 0.5, what="grouped dgrad (accumulate)")
    check(L.tf_conv3x3_grouped_dgrad_f32(ptr(dyh), wptr(wh), ptr(dx), B, H, W, C, 0, stream_of(xh)), "grouped dgrad")
    close(dx.permute(0, 3, 1, 2), gx, what="grouped dgrad")
    dw = torch.full_like(wh, 0.25)
    ws = ops._grouped_ws(xh.device)
    check(L.tf_conv3x3_grouped_wgrad_f32(ptr(dyh), ptr(xh), wptr(dw), B, H, W, C, 1, ptr(ws), stream_of(xh)), "grouped wgrad")
    close(dw, gw + 0.25, what="grouped wgrad (accumulate)")
    dw0 = torch.full_like(wh, 7.0)               # accumulate = 0 overwrites whatever dW held (the atomically accumulating kernel zero-fills first)
    check(L.tf_conv3x3_grouped_wgrad_f32(ptr(dyh), ptr(xh), wptr(dw0), B, H, W, C, 0, ptr(ws), stream_of(xh)), "grouped wgrad")
    close(dw0, gw, what="grouped wgrad (store)")
    # and through the public ops (dispatch)
    if groups > 1:
        close(ops.conv_fwd(xh, wh, None, 1, None, groups).permute(0, 3, 1, 2), y - b.view(1, -1, 1, 1), what="ops.conv_fwd -> grouped")
        dw2 = torch.zeros_like(wh)
        ops.conv_wgrad(dyh, xh, dw2, 1, None, groups)
        close(dw2, gw, what="ops.conv_wgrad -> grouped")
        close(ops.conv_dgrad(dyh, wh, xh.shape, 1, None, groups).permute(0, 3, 1, 2), gx, what="ops.conv_dgrad -> grouped")
    # BatchNorm apply folded into the consumer: conv(relu(x sc + sh)) and its weight gradient from the RAW x; the zero padding must stay zero
    # (shift > 0 on some channels would otherwise leak relu(shift) in from the border), forward output statistics as the plain colstat forward
    sc, sh = R(C, seed=7, dev="cpu") * 0.5 + 1.0, R(C, seed=8, dev="cpu") * 0.7 + 0.3
    z = torch.relu(x.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    yz = F.conv2d(z, w, None, 1, 1, 1, groups)
    (gwz,) = torch.autograd.grad(yz, [w], dy)
    coef = torch.cat([sc, sh]).to(dev)
    y2 = torch.empty(B, H, W, C, device=dev)
    cs = ops.ColStat(B * H * W, C, xh.device, max_parts=L.tf_conv3x3_grouped_colstat_parts())
    from ctypes import byref
    check(L.tf_conv3x3_grouped_bnrelu_fwd_colstat_f32(ptr(xh), ptr(coef), wptr(wh), ptr(y2), B, H, W, C, ptr(cs.buf), byref(cs.nparts), stream_of(xh)), "grouped bnrelu fwd")
    close(y2.permute(0, 3, 1, 2), yz, what="grouped fwd with the BatchNorm apply of its input folded in")
    g_, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    _, sm, si = ops.bn_finalize_parts(cs, g_, b_, rm, rv)
    y64 = yz.detach().double().permute(0, 2, 3, 1).reshape(-1, C)
    close(sm.cpu(), y64.mean(0).float(), what="grouped bnrelu fwd: output mean", tol=1e-4)
    close(si.cpu(), (1.0 / torch.sqrt(y64.var(0, unbiased=False) + 1e-5)).float(), what="grouped bnrelu fwd: output invstd", tol=1e-4)
    dwz = torch.full_like(wh, 0.25)
    check(L.tf_conv3x3_grouped_bnrelu_wgrad_f32(ptr(dyh), ptr(xh), ptr(coef), wptr(dwz), B, H, W, C, 1, ptr(ws), stream_of(xh)), "grouped bnrelu wgrad")
    close(dwz, cl(gwz) + 0.25, what="grouped wgrad against the recomputed activation (accumulate)")


def check_im2col_gemm_conv(dev):
    """The opt-in im2col + plain-GEMM form of the few-row / deep-K dense 3x3 convolutions (decoder heads; TF_IM2COL_GEMM): the matrix itself (zero
    padding, (kh, kw, c) column order) and the convolution with bias + ReLU through ops.conv_fwd vs F.conv2d."""
    from transfuser_amd.ops import ptr, stream_of, check
    cases = ((2, 8, 22, 128, 32), (1, 5, 7, 116, 24))
    if dev != "cpu":      # the decoders' own first layers at the bench batch (512 -> 128 and 128 -> 64 at 8 x 22, B = 10)
        cases += ((10, 8, 22, 512, 128), (10, 8, 22, 128, 64))
    for (B, H, W, Cin, Cout) in cases:
        x = R(B, Cin, H, W, dev="cpu")
        w = R(Cout, Cin, 3, 3, seed=1, dev="cpu") * 0.05
        b = R(Cout, seed=2, dev="cpu")
        xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
        cols = torch.empty(B * H * W, 9 * Cin, device=dev)
        check(ops.L().tf_im2col3x3_f32(ptr(xh), ptr(cols), B, H, W, Cin, stream_of(xh)), "tf_im2col3x3_f32")
        ref = F.unfold(x, 3, padding=1).view(B, Cin, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * Cin)      # unfold orders (c, tap): -> (tap, c)
        assert torch.equal(cols.cpu(), ref), "im2col matrix"
        prev = ops._IM2COL_GEMM
        ops._IM2COL_GEMM = True
        try:
            g = ops.conv_geom(xh.shape, Cout, 3, 1, 1, 1)
            assert ops._im2col_gemm_ok(g, 3, 1, 1, 1)
            y = ops.conv_fwd(xh, cl(w).to(dev), b.to(dev), 1, 1, 1, relu=True)
        finally:
            ops._IM2COL_GEMM = prev
        close(y.permute(0, 3, 1, 2), torch.relu(F.conv2d(x, w, b, 1, 1)), what="im2col + GEMM convolution (bias, ReLU)")
        close(y, ops.conv_fwd(xh, cl(w).to(dev), b.to(dev), 1, 1, 1, relu=True), what="== the implicit-GEMM path", tol=1e-5)


GROUPED_S2_CASES = [(2, 16, 44, 72), (1, 9, 17, 48), (2, 8, 16, 24), (1, 33, 31, 48), (1, 12, 70, 72)]


def check_conv_grouped_s2(dev, B, H, W, C):
    """The direct STRIDE-2 grouped 3x3 kernels (first block of a RegNetY stage; opt-in TF_GROUPED_S2): forward with output statistics, forward with
    the producer's BatchNorm apply folded in, weight gradient (plain / folded, accumulate) vs F.conv2d(stride 2); even and odd input extents, both
    tile shapes, 1..3 groups; and the ops dispatch with the switch on."""
    groups = C // 24
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = R(B, C, H, W, dev="cpu").requires_grad_(True)
    w = (R(C, 24, 3, 3, dev="cpu") * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, None, 2, 1, 1, groups)
    assert y.shape[2:] == (Ho, Wo)
    dy = R(*y.shape, seed=1, dev="cpu")
    (gw,) = torch.autograd.grad(y, [w], dy)
    xh, wh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev), cl(w.detach()).to(dev)
    dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    L = ops.L()
    from ctypes import byref
    from transfuser_amd.ops import ptr, wptr, stream_of, check, c_p
    yh = torch.empty(B, Ho, Wo, C, device=dev)
    cs = ops.ColStat(B * Ho * Wo, C, xh.device, max_parts=L.tf_conv3x3_grouped_colstat_parts())
    check(L.tf_conv3x3_grouped_s2_fwd_f32(ptr(xh), c_p(0), wptr(wh), ptr(yh), B, H, W, C, ptr(cs.buf), byref(cs.nparts), stream_of(xh)), "grouped s2 fwd")
    close(yh.permute(0, 3, 1, 2), y, what="grouped s2 fwd")
    one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    _, sm, si = ops.bn_finalize_parts(cs, one, zero, zero.clone(), one.clone())
    y64 = y.detach().double().permute(0, 2, 3, 1).reshape(-1, C)
    close(sm.cpu(), y64.mean(0).float(), what="grouped s2 fwd: output mean", tol=1e-4)
    close(si.cpu(), (1.0 / torch.sqrt(y64.var(0, unbiased=False) + 1e-5)).float(), what="grouped s2 fwd: output invstd", tol=1e-4)
    yh2 = torch.empty_like(yh)
    check(L.tf_conv3x3_grouped_s2_fwd_f32(ptr(xh), c_p(0), wptr(wh), ptr(yh2), B, H, W, C, c_p(0), c_p(0), stream_of(xh)), "grouped s2 fwd (no statistics)")
    assert torch.equal(yh2, yh)
    ws = ops._grouped_ws(xh.device)
    dw = torch.full_like(wh, 0.25)
    check(L.tf_conv3x3_grouped_s2_wgrad_f32(ptr(dyh), ptr(xh), c_p(0), wptr(dw), B, H, W, C, 1, ptr(ws), stream_of(xh)), "grouped s2 wgrad")
    close(dw, cl(gw) + 0.25, what="grouped s2 wgrad (accumulate)")
    # the producer's BatchNorm apply folded in: conv(relu(x sc + sh)) from the raw x, zero padding stays zero
    sc, sh = R(C, seed=7, dev="cpu") * 0.5 + 1.0, R(C, seed=8, dev="cpu") * 0.7 + 0.3
    z = torch.relu(x.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    yz = F.conv2d(z, w, None, 2, 1, 1, groups)
    (gwz,) = torch.autograd.grad(yz, [w], dy)
    coef = torch.cat([sc, sh]).to(dev)
    check(L.tf_conv3x3_grouped_s2_fwd_f32(ptr(xh), ptr(coef), wptr(wh), ptr(yh2), B, H, W, C, c_p(0), c_p(0), stream_of(xh)), "grouped s2 bnrelu fwd")
    close(yh2.permute(0, 3, 1, 2), yz, what="grouped s2 fwd with the BatchNorm apply of its input folded in")
    dwz = torch.zeros_like(wh)
    check(L.tf_conv3x3_grouped_s2_wgrad_f32(ptr(dyh), ptr(xh), ptr(coef), wptr(dwz), B, H, W, C, 0, ptr(ws), stream_of(xh)), "grouped s2 bnrelu wgrad")
    close(dwz, cl(gwz), what="grouped s2 wgrad against the recomputed activation")
    # dispatch through the public ops with the switch on
    if groups > 1 and Ho >= 4 and Wo >= 8:
        prev = ops._GROUPED_S2
        ops._GROUPED_S2 = True
        try:
            yo, cso = ops.conv_fwd(xh, wh, None, 2, 1, groups, colstat=True)
            close(yo.permute(0, 3, 1, 2), y, what="ops.conv_fwd -> grouped s2")
            assert cso is not None and cso.nparts.value > 0
            dw2 = torch.zeros_like(wh)
            ops.conv_wgrad(dyh, xh, dw2, 2, 1, groups)
            close(dw2, cl(gw), what="ops.conv_wgrad -> grouped s2")
        finally:
            ops._GROUPED_S2 = prev


def check_bf16_direct(dev):
    """The LDS-tiled direct convolutions (decoder tails, RegNetY grouped 3x3) in bf16-MFMA mode: == fp32 convolution of bf16-rounded operands."""
    old = ops._DIRECT_MIN_PIXELS
    ops._DIRECT_MIN_PIXELS = 0
    ops.set_precision("bf16")
    try:
        for (B, H, W, Cin, Cout, groups) in ((2, 10, 70, 32, 32, 1), (1, 9, 33, 32, 7, 1), (1, 12, 64, 8, 32, 1), (2, 16, 44, 72, 72, 3), (1, 9, 13, 48, 48, 2)):
            x = R(B, Cin, H, W, dev="cpu").requires_grad_(True)
            w = (R(Cout, Cin // groups, 3, 3, dev="cpu") * 0.1).requires_grad_(True)
            dy = R(B, Cout, H, W, seed=1, dev="cpu")
            # the thin-output kernels (32 -> Cout <= 7, bandwidth-bound) multiply exact fp32 in EVERY precision mode: their reference is unrounded
            rd = (lambda t: t) if ops._thin_ok((B, H, W, Cin), Cout, Cin, 3, 1, 1, groups) else _bf
            y = F.conv2d(rd(x), rd(w), None, 1, 1, 1, groups)
            gx = torch.autograd.grad(F.conv2d(x, rd(w), None, 1, 1, 1, groups), x, rd(dy))[0]
            gw = torch.autograd.grad(F.conv2d(rd(x), w, None, 1, 1, 1, groups), w, rd(dy))[0]
            xh, wh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev), cl(w.detach()).to(dev)
            dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
            assert ops._direct_ok(xh.shape, Cout, Cin, 3, 1, 1, groups) or ops._grouped_ok(xh.shape, Cout, Cin, 3, 1, 1, groups)
            close(ops.conv_fwd(xh, wh, None, 1, None, groups).permute(0, 3, 1, 2), y, tol=1e-4, what="bf16 direct fwd")
            close(ops.conv_dgrad(dyh, wh, xh.shape, 1, None, groups).permute(0, 3, 1, 2), gx, tol=1e-4, what="bf16 direct dgrad")
            dw = torch.zeros_like(wh)
            ops.conv_wgrad(dyh, xh, dw, 1, None, groups)
            close(dw, gw, tol=1e-4, what="bf16 direct wgrad")
    finally:
        ops.set_precision("fp32")
        ops._DIRECT_MIN_PIXELS = old


def check_grouped_s2_modes(dev):
    """The opt-in direct stride-2 grouped kernels in the other compute modes: bf16 / fp16 == fp32 convolution of the operands rounded to that type
    (forward and weight gradient), f32x3 within the fp32 bound of a float64 reference."""
    prev = ops._GROUPED_S2
    ops._GROUPED_S2 = True
    try:
        for (B, H, W, C) in ((2, 16, 44, 72), (1, 9, 17, 48)):
            groups = C // 24
            x = R(B, C, H, W, dev="cpu")
            w = R(C, 24, 3, 3, dev="cpu") * 0.1
            y0 = F.conv2d(x, w, None, 2, 1, 1, groups)
            dy = R(*y0.shape, seed=1, dev="cpu")
            xh, wh = x.permute(0, 2, 3, 1).contiguous().to(dev), cl(w).to(dev)
            dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
            for mode, rd in (("bf16", _bf), ("fp16", lambda t: t.half().float())):
                ops.set_precision(mode)
                wr = rd(w).requires_grad_(True)
                yr = F.conv2d(rd(x), wr, None, 2, 1, 1, groups)
                (gw,) = torch.autograd.grad(yr, [wr], rd(dy))
                close(ops.conv_fwd(xh, wh, None, 2, 1, groups).permute(0, 3, 1, 2), yr, tol=1e-4, what=mode + " grouped s2 fwd")
                dw = torch.zeros_like(wh)
                ops.conv_wgrad(dyh, xh, dw, 2, 1, groups)
                close(dw, cl(gw), tol=1e-4, what=mode + " grouped s2 wgrad")
            ops.set_precision("f32x3")
            wd = w.double().requires_grad_(True)
            yd = F.conv2d(x.double(), wd, None, 2, 1, 1, groups)
            (gwd,) = torch.autograd.grad(yd, [wd], dy.double())
            assert _err64(ops.conv_fwd(xh, wh, None, 2, 1, groups).permute(0, 3, 1, 2), yd.detach()) <= 4e-6
            dw = torch.zeros_like(wh)
            ops.conv_wgrad(dyh, xh, dw, 2, 1, groups)
            assert _err64(dw, gwd.detach()) <= 4e-6
    finally:
        ops.set_precision("fp32")
        ops._GROUPED_S2 = prev


def check_f32x3_direct(dev):
    """The LDS-tiled direct convolutions (decoder tails, RegNetY grouped 3x3) in f32x3 mode: float64 reference, the fp32-accuracy bound of
    check_f32x3_mode (<= 2e-6 relative, <= 3x the exact-fp32-MFMA kernels' own error + 2e-7)."""
    old = ops._DIRECT_MIN_PIXELS
    ops._DIRECT_MIN_PIXELS = 0
    try:
        for (B, H, W, Cin, Cout, groups) in ((2, 10, 70, 32, 32, 1), (1, 9, 33, 32, 7, 1), (1, 12, 64, 8, 32, 1), (1, 19, 37, 12, 20, 1), (2, 16, 44, 72, 72, 3),
                                             (1, 9, 13, 48, 48, 2)):
            x = R(B, Cin, H, W, dev="cpu").double().requires_grad_(True)
            w = (R(Cout, Cin // groups, 3, 3, dev="cpu") * 0.1).double().requires_grad_(True)
            dy = R(B, Cout, H, W, seed=1, dev="cpu").double()
            y = F.conv2d(x, w, None, 1, 1, 1, groups)
            gx, gw = torch.autograd.grad(y, [x, w], dy)
            xh, wh = x.detach().float().permute(0, 2, 3, 1).contiguous().to(dev), cl(w.detach().float()).to(dev)
            dyh = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
            assert ops._direct_ok(xh.shape, Cout, Cin, 3, 1, 1, groups) or ops._grouped_ok(xh.shape, Cout, Cin, 3, 1, 1, groups)

            def wg():
                dw = torch.zeros_like(wh)
                ops.conv_wgrad(dyh, xh, dw, 1, None, groups)
                return dw
            for what, fn, ref in (("fwd", lambda: ops.conv_fwd(xh, wh, None, 1, None, groups).permute(0, 3, 1, 2), y.detach()),
                                  ("dgrad", lambda: ops.conv_dgrad(dyh, wh, xh.shape, 1, None, groups).permute(0, 3, 1, 2), gx), ("wgrad", wg, gw)):
                ops.set_precision("fp32")
                e32 = _err64(fn(), ref)
                ops.set_precision("f32x3")
                e3 = _err64(fn(), ref)
                assert e3 <= 2e-6 and e3 <= 3.0 * e32 + 2e-7, "x3 direct %s %s: err %.3e vs fp32-MFMA err %.3e" % (what, (B, H, W, Cin, Cout, groups), e3, e32)
    finally:
        ops.set_precision("fp32")
        ops._DIRECT_MIN_PIXELS = old


# ---------------------------------------------------------------- deterministic two-pass split-K (few-tile outputs, csrc/gemm_fixup.cpp)
TWO_PASS = 1000000       # tf_gemm_engine.h kTwoPass: splitk = TWO_PASS + S
TWO_PASS_CASES = [(1, TWO_PASS + 2), (2, TWO_PASS + 3), (5, TWO_PASS + 4), (6, TWO_PASS + 3), (8, TWO_PASS + 2)]


def check_two_pass_splitk(dev, kind, sk):
    """k-slices store partial tiles into caller scratch, the fix-up pass sums them in order and applies the epilogue: every epilogue form
    (bias + residual + ReLU; ReLU mask; accumulate), all four operand layouts, ragged N (scalar fix-up path), bitwise run-to-run equal;
    once more in the f32x3 compute mode (the X3 kernel instantiations store their slices the same way)."""
    ops.force_dma(kind, sk)
    try:
        ops.set_precision("f32x3")
        x, w, b, r = R(200, 600, dev=dev), R(92, 600, dev=dev), R(92, dev=dev), R(200, 92, dev=dev)
        close(ops.linear_fwd(x, w, b, relu=True, res=r), torch.relu(x @ w.t() + b + r), what="two-pass fwd (f32x3)")
        ops.set_precision("fp32")
        for (m, n, k) in ((200, 92, 600), (130, 216, 1030), (257, 130, 520)):
            x, w, b, r = R(m, k, dev=dev), R(n, k, dev=dev), R(n, dev=dev), R(m, n, dev=dev)
            y = ops.linear_fwd(x, w, b, relu=True, res=r)
            close(y, torch.relu(x @ w.t() + b + r), what="two-pass fwd")
            assert torch.equal(y, ops.linear_fwd(x, w, b, relu=True, res=r)), "two-pass split-K must be run-to-run deterministic"
            dy, act = R(m, n, seed=1, dev=dev), R(m, k, seed=5, dev=dev)
            if n % 4 == 0:
                close(ops.linear_dgrad(dy, w, mask=act), (dy @ w) * (act > 0), what="two-pass dgrad + mask")
            dw0 = R(n, k, seed=2, dev=dev)
            close(ops.linear_wgrad(dy, x, dw0.clone(), accumulate=True), dw0 + dy.t() @ x, what="two-pass wgrad (+=)") if (n % 4 == 0) else None
            a, bb = R(m, k, dev=dev), R(n, k, seed=4, dev=dev)
            if m % 4 == 0:
                c = torch.empty(m, n, device=dev)
                ops.gemm(a.t().contiguous(), bb, c, m, n, k, m, k, n, a_trans=True, alpha=0.5)
                close(c, 0.5 * (a @ bb.t()), what="two-pass tt")
    finally:
        ops.set_precision("fp32")
        ops.force_plan(0)


STREAM_K = 2000000     # GemmPlan.splitk == kStreamK
STREAM_K_KINDS_EMU = [2, 6]
STREAM_K_KINDS_GPU = [1, 2, 3, 4, 5, 6, 7, 8]


def check_stream_k(dev, kind):
    prev, ops.STREAM_K = ops.STREAM_K, True      # opt-in since round 5 (TF_STREAM_K, default off): the kernels stay tested
    try:
        _check_stream_k(dev, kind)
    finally:
        ops.STREAM_K = prev


def _check_stream_k(dev, kind):
    """Stream-K plans of the LDS-DMA kernels (persistent workgroups over the (tile, k-tile) space, a cut tile's k-tail handed over through
    scratch + flag): every operand layout and epilogue form (bias + residual + ReLU; ReLU mask; accumulate; alpha), ragged M / N / K, the
    fused BatchNorm statistics (the owner of a cut tile holds the complete sum), bitwise run-to-run equality (fixed head + tail order), and the
    flags left clear for the next launch.  Sizes: a tile count that does not divide over the device's resident slots (8 emulated CUs on the
    host; on the MI355X the shapes are scaled so that tiles are actually cut)."""
    info = {1: (128, 128), 2: (64, 64), 3: (128, 64), 4: (64, 128), 5: (128, 128), 6: (64, 64), 7: (64, 128), 8: (128, 64)}[kind]
    bm, bn = info
    gpu = str(dev).startswith("cuda")
    # tiles: a little more than one full round of the resident slots (host: 8 CUs x 2..5 workgroups; MI355X: 256 CUs)
    tm, tn = (6, 7) if not gpu else (33, 19)
    m, n = tm * bm - 17, tn * bn - 20            # ragged last tile row / column, n % 4 == 0
    L = ops.L()
    L.tf_streamk_launches.restype = ctypes.c_long
    for k in ((260, 264) if not gpu else (1032, 1512)):
        ops.force_dma(kind, STREAM_K)
        n0 = L.tf_streamk_launches()
        try:
            x, w, b, r = R(m, k, dev=dev), R(n, k, dev=dev) * 0.2, R(n, dev=dev), R(m, n, dev=dev)
            want = torch.relu(x.double() @ w.double().t() + b.double() + r.double()).float()
            y = ops.linear_fwd(x, w, b, relu=True, res=r)
            close(y, want, what="stream-K fwd nt kind %d k %d" % (kind, k))
            assert torch.equal(y, ops.linear_fwd(x, w, b, relu=True, res=r)), "stream-K must be run-to-run deterministic"
            flags = ops._sk_flags(x.device)
            assert not flags.any(), "stream-K flags must be clear between launches"
            dy, act = R(m, n, seed=1, dev=dev), R(m, k, seed=5, dev=dev)
            if k % 4 == 0:
                close(ops.linear_dgrad(dy, w, mask=act), ((dy.double() @ w.double()) * (act > 0)).float(), what="stream-K dgrad nn + mask kind %d" % kind)
                dw0 = R(n, k, seed=2, dev=dev)
                close(ops.linear_wgrad(dy, x, dw0.clone(), accumulate=True), (dw0.double() + dy.double().t() @ x.double()).float(), what="stream-K wgrad tn (+=) kind %d" % kind)
            a, bb = R(m + 17, k, dev=dev), R(n, k, seed=4, dev=dev)
            mm = m + 17                               # tt needs m % 4 == 0
            c = torch.empty(mm, n, device=dev)
            ops.gemm(a.t().contiguous(), bb, c, mm, n, k, mm, k, n, a_trans=True, alpha=0.5)
            close(c, (0.5 * (a.double() @ bb.double().t())).float(), what="stream-K tt kind %d" % kind)
            assert not flags.any()
            # (a layout whose tile count happens to divide over the device's slots runs data-parallel: not every call below is cut on every device)
            assert L.tf_streamk_launches() - n0 >= (5 if not gpu else 2), "the pinned stream-K plan did not run as stream-K (%d launches)" % (L.tf_streamk_launches() - n0)
            # BatchNorm statistics from the epilogue of a stream-K launch
            old = ops.FUSE_BN_STATS
            ops.FUSE_BN_STATS = True
            try:
                if m <= ops._COLSTAT_MAX_ROWS and k % 4 == 0:
                    yy, cs = ops.linear_fwd(x + 2.0, w, colstat=True)
                    close(yy, ((x.double() + 2.0) @ w.double().t()).float(), what="stream-K fwd + colstat")
                    _bn_from_parts(dev, yy.view(1, 1, m, n), cs, relu=True)
            finally:
                ops.FUSE_BN_STATS = old
        finally:
            ops.force_plan(0)


# ---------------------------------------------------------------- BatchNorm statistics fused into the producing convolution
def _bn_from_parts(dev, y_nhwc, cs, relu=True, res=None):
    """BN forward through ops.bn_fwd_parts (statistics from the producer's epilogue) vs F.batch_norm on the same conv output."""
    C = y_nhwc.shape[-1]
    assert cs is not None and cs.nparts.value > 0, "the producer did not write statistics"
    n = cs.nparts.value
    cnt = cs.buf.view(-1, 3, C)[:n, 0].sum(0)
    assert torch.all(cnt == y_nhwc.numel() // C), ("part counts must add up to the row count", cnt[:4], y_nhwc.numel() // C)
    g, b = R(C, seed=11, dev=dev) * 0.3 + 1, R(C, seed=12, dev=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    z, sm, si = ops.bn_fwd_parts(y_nhwc, cs, g, b, rm, rv, res, relu)
    rm2, rv2 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    flat = y_nhwc.reshape(-1, C)
    want = F.batch_norm(flat, rm2, rv2, g, b, True, 0.1, 1e-5)
    if res is not None:
        want = want + res.reshape(-1, C)
    if relu:
        want = torch.relu(want)
    close(z.reshape(-1, C), want, what="bn from fused statistics")
    close(rm, rm2, what="running mean (fused statistics)")
    close(rv, rv2, what="running var (fused statistics)")
    close(sm, flat.mean(0), what="save_mean")
    close(si, 1.0 / torch.sqrt(flat.var(0, unbiased=False) + 1e-5), what="save_invstd")


def check_bn_fused_stats(dev, plans=((0, 64, 64, 16), (0, 128, 32, 16), (0, 128, 128, 32), (0, 64, 128, 16), (2, 0, 0, 0), (1, 0, 0, 0), (6, 0, 0, 0), (8, 0, 0, 0))):
    """The producers of a BatchNorm input write per-part Welford triples from their epilogue: 1x1 conv as GEMM under every engine tiling /
    LDS-DMA kind (rows per wave 32 / 64, ragged M and N, activations with a LARGE mean: the shifted / Welford form must not cancel),
    Linear + bias (PointPillars), strided grouped + 1x1 / s2 convs through the implicit-GEMM engine, the direct grouped 3x3 kernel (both
    tile shapes, ragged borders), and the fall-back when a two-pass split-K plan cannot produce them."""
    old_fuse, old_rows = ops.FUSE_BN_STATS, ops._COLSTAT_MAX_ROWS
    ops.FUSE_BN_STATS = True
    try:
        for kind, bm, bn, bk in plans:
            if kind:
                ops.force_dma(kind, 1)
            else:
                ops.force_plan(bm, bn, bk, 1)
            try:
                for (m, n, k) in ((203, 72, 40), (130, 216, 64), (97, 24, 36)):
                    x = R(m, k, dev=dev) + 3.0                      # mean >> spread on purpose
                    w = R(n, k, seed=5, dev=dev) * 0.2
                    y, cs = ops.linear_fwd(x, w, colstat=True)
                    close(y, x @ w.t(), what="linear with colstat")
                    _bn_from_parts(dev, y.view(1, 1, m, n), cs, relu=(n != 24), res=R(1, 1, m, n, seed=8, dev=dev) if n == 72 else None)
                if (kind, bm) == (0, 64):      # many parts: 288 (one block per channel, one trip) and 1250 (two trips) - the stem / stage-1 / stage-2 layers
                    for m in (9200, 40000):
                        x = R(m, 16, dev=dev) + 3.0
                        w = R(24, 16, seed=5, dev=dev) * 0.2
                        y, cs = ops.linear_fwd(x, w, colstat=True)
                        _bn_from_parts(dev, y.view(1, 1, m, 24), cs, relu=True)
                bias = R(32, seed=6, dev=dev)
                x = R(150, 9, dev=dev)
                w = R(32, 9, seed=7, dev=dev)
                y, cs = ops.linear_fwd(x, w, bias, colstat=True)        # point_pillar.py:15-25: Linear(9, 32) with bias + BatchNorm1d (unaligned K: engine path)
                close(y, x @ w.t() + bias, what="linear + bias with colstat")
                _bn_from_parts(dev, y.view(1, 1, 150, 32), cs)
            finally:
                ops.force_plan(0)
        # implicit-GEMM convolutions: stride-2 grouped 3x3 (first block of a stage), 1x1 / s2 downsample
        for (B, H, W, Cin, Cout, ks, stride, groups) in ((2, 10, 12, 48, 48, 3, 2, 2), (2, 9, 11, 24, 72, 1, 2, 1), (1, 12, 16, 72, 72, 3, 2, 3)):
            xc = R(B, Cin, H, W, dev="cpu") + 1.5
            wc = R(Cout, Cin // groups, ks, ks, seed=3, dev="cpu") * 0.2
            want = F.conv2d(xc, wc, None, stride, ks // 2, 1, groups)
            y, cs = ops.conv_fwd(xc.permute(0, 2, 3, 1).contiguous().to(dev), cl(wc).to(dev), None, stride, ks // 2, groups, colstat=True)
            close(y.permute(0, 3, 1, 2), want, what="conv with colstat")
            _bn_from_parts(dev, y, cs)
        # direct grouped kernel: 8x16 and 4x32 tiles, ragged in both directions, 1..3 groups, several tiles per block
        for (B, H, W, C) in ((2, 9, 44, 72), (3, 16, 16, 48), (2, 5, 37, 24), (2, 20, 70, 72)):
            xc = R(B, C, H, W, dev="cpu") - 2.0
            wc = R(C, 24, 3, 3, seed=3, dev="cpu") * 0.1
            want = F.conv2d(xc, wc, None, 1, 1, 1, C // 24)
            y, cs = ops.conv_fwd(xc.permute(0, 2, 3, 1).contiguous().to(dev), cl(wc).to(dev), None, 1, 1, C // 24, colstat=True)
            close(y.permute(0, 3, 1, 2), want, what="grouped conv with colstat")
            if C // 24 > 1:
                _bn_from_parts(dev, y, cs)
            else:
                assert cs is None or cs.nparts.value > 0
        # maps above the row threshold keep the streaming reduction: no statistics requested
        ops._COLSTAT_MAX_ROWS = 100
        y, cs = ops.linear_fwd(R(203, 40, dev=dev), R(72, 40, seed=5, dev=dev), colstat=True)
        assert cs is None
        ops._COLSTAT_MAX_ROWS = old_rows
        # a two-pass split-K plan cannot fuse them: nparts = 0 -> None, the output is still right
        ops.force_dma(2, 1000002)
        try:
            x, w = R(260, 640, dev=dev), R(136, 640, seed=5, dev=dev) * 0.1
            y, cs = ops.linear_fwd(x, w, colstat=True)
            close(y, x @ w.t(), what="two-pass GEMM with a statistics request")
            assert cs is None
        finally:
            ops.force_plan(0)
    finally:
        ops.FUSE_BN_STATS, ops._COLSTAT_MAX_ROWS = old_fuse, old_rows


# ---------------------------------------------------------------- fused attention (csrc/attention.cpp)
FUSED_ATTENTION_CASES = [(2, 4, 174, 18, 0.1), (1, 4, 174, 54, 0.0), (2, 2, 50, 24, 0.1), (1, 3, 192, 40, 0.1), (2, 1, 33, 6, 0.0),
                         (1, 1, 40, 258, 0.1), (1, 2, 70, 256, 0.0)]      # head sizes >= 256: the backward kernels stage the row operand in LDS (ragged last chunk / none)
FUSED_ATTENTION_CASES_GPU = [(10, 4, 174, 378, 0.1), (3, 4, 174, 144, 0.1), (2, 4, 174, 54, 0.0), (2, 4, 174, 18, 0.1), (1, 1, 192, 384, 0.1), (2, 3, 97, 34, 0.1)]


def check_fused_attention(dev, B, nh, T, hs, p):
    """tf_attention_fwd / bwd vs PyTorch (SelfAttention.forward, transfuser.py:510-527: softmax(q k^T / sqrt(hs)) -> attn_drop -> @ v, heads
    merged) on the same qkv, with the PRODUCT's dropout mask (flat index over (B nh, T, Tp)) applied by the reference; y, and dq / dk / dv."""
    C = nh * hs
    qkv = R(B * T, 3 * C, dev="cpu", scale=0.7)
    qkv.requires_grad_(True)
    k, q, v = [qkv[:, j * C:(j + 1) * C].view(B, T, nh, hs).transpose(1, 2) for j in range(3)]
    att = torch.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hs)), dim=-1)
    seed = torch.full((1,), 777, dtype=torch.int32, device=dev)
    site = 13
    Tp = (T + 3) // 4 * 4
    if p > 0:
        mask = ops.dropout(torch.ones(B * nh, T, Tp, device=dev), seed, site, p)[:, :, :T].reshape(B, nh, T, T).cpu()
        assert 0.5 * p < float((mask == 0).float().mean()) < 1.5 * p + 0.02
        att = att * mask
    y = (att @ v).transpose(1, 2).contiguous().view(B * T, C)
    dy = R(B * T, C, seed=9, dev="cpu")
    (g,) = torch.autograd.grad(y, qkv, dy)
    drop = (seed, site, p) if p > 0 else None
    qd = qkv.detach().to(dev)
    assert ops.attention_supported(T, C, nh)
    yh, lse = ops.attention_fwd(qd, B, T, C, nh, drop)
    close(yh, y, what="fused attention fwd")
    s = (q @ k.transpose(-2, -1)).detach() * (1.0 / math.sqrt(hs))
    close(lse.view(B, nh, T), torch.logsumexp(s, -1), what="fused attention log-sum-exp")
    for form, kw in (("two dependent launches", {}), ("one grid, D = dY . Y", dict(y=yh))):      # rounds 3-5 / round 6 (tf_attention_bwd_y_f32)
        dq = ops.attention_bwd(qd, dy.to(dev), lse, B, T, C, nh, drop, **kw)
        close(dq[:, C:2 * C], g[:, C:2 * C], what="fused attention dQ (%s)" % form)
        close(dq[:, :C], g[:, :C], what="fused attention dK (%s)" % form)
        close(dq[:, 2 * C:], g[:, 2 * C:], what="fused attention dV (%s)" % form)


# ---------------------------------------------------------------- pair launch (csrc/gemm_pair.cpp)
GEMM_PAIR_CASES = [(300, 72, 100, 3), (132, 216, 40, 1), (70, 64, 64, 2), (260, 132, 72, 4), (64, 24, 36, 1)]      # dims % 4 == 0: the joint kernel needs 16-byte operand loads


def check_gemm_pair(dev, M, N, K, splitk, bks=((32, 32), (32, 16), (16, 32), (16, 16))):
    """Weight gradient + input gradient of one layer in ONE grid (ops.gemm_pair): dW (N, K) += dy^T x with an atomic k-split, dx (M, K) = dy W
    (+ residual, ReLU mask) - against fp64, and against the same two calls issued one by one; every BK combination of the joint kernel; the
    library must report a joint launch.  A lone held call and a pair without a joint kernel (two forward products) fall back to plain launches."""
    dy, x, w = R(M, N, dev=dev), R(M, K, seed=1, dev=dev), R(N, K, seed=2, dev=dev) * 0.1
    res, act = R(M, K, seed=3, dev=dev), R(M, K, seed=4, dev=dev)
    dw0 = R(N, K, seed=5, dev=dev) * 0.1
    want_dw = (dw0.double() + dy.double().t() @ x.double()).float()
    want_dx = ((dy.double() @ w.double() + res.double()) * (act > 0)).float()
    for (bkw, bkd) in bks:
        # the plan pin applies to both calls; the weight gradient's BK is pinned first, then re-pinned for the input gradient (plans are resolved per call)
        n0 = ops.gemm_pair_count()
        dw = dw0.clone()
        with ops.gemm_pair(dy) as gp:
            assert gp.on
            ops.force_plan(64, 64, bkw, splitk)
            ops.linear_wgrad(dy, x, dw)
            ops.force_plan(64, 64, bkd, 1)
            dx = ops.linear_dgrad(dy, w, res=res, mask=act)
            ops.force_plan(0)
        assert ops.gemm_pair_count() == n0 + 1, "the joint kernel did not run"
        close(dw, want_dw, tol=2e-5 * max(1, M // 64), what="pair wgrad %s" % ((M, N, K, splitk, bkw, bkd),))
        close(dx, want_dx, tol=2e-5 * max(1, N // 64), what="pair dgrad %s" % ((M, N, K, splitk, bkw, bkd),))
        # the same two calls one by one
        dw1 = dw0.clone()
        ops.force_plan(64, 64, bkw, splitk)
        ops.linear_wgrad(dy, x, dw1)
        ops.force_plan(64, 64, bkd, 1)
        dx1 = ops.linear_dgrad(dy, w, res=res, mask=act)
        ops.force_plan(0)
        assert torch.equal(dx1, dx), "pair dgrad == single dgrad (bitwise)"
        close(dw, dw1, tol=1e-6, what="pair wgrad vs single wgrad")
    # one eligible call inside the bracket: an ordinary launch at the end of the block
    s0 = ops.gemm_pair_count(singles=True)
    ops.force_plan(64, 64, 32, 1)
    with ops.gemm_pair(dy):
        dx = ops.linear_dgrad(dy, w)
    assert ops.gemm_pair_count(singles=True) == s0 + 1
    close(dx, (dy.double() @ w.double()).float(), tol=2e-5 * max(1, N // 64), what="lone held call")
    # two forward products: no joint kernel for that layout pair -> two plain launches in call order (the second reads the first's output)
    with ops.gemm_pair(dy):
        y1 = ops.linear_fwd(x, w)            # (M, N)
        y2 = ops.linear_fwd(x, w, res=dy)
    ops.force_plan(0)
    close(y1, (x.double() @ w.double().t()).float(), tol=2e-5 * max(1, K // 64), what="forward inside the bracket")
    close(y2, (x.double() @ w.double().t() + dy.double()).float(), tol=2e-5 * max(1, K // 64), what="second forward inside the bracket")
    # an ineligible plan (128-row tiles) launches immediately, the eligible partner at the end
    p0, s0 = ops.gemm_pair_count(), ops.gemm_pair_count(singles=True)
    dw = dw0.clone()
    with ops.gemm_pair(dy):
        ops.force_plan(128, 64, 16, 1)
        ops.linear_wgrad(dy, x, dw)
        ops.force_plan(64, 64, 16, 1)
        dx = ops.linear_dgrad(dy, w, res=res, mask=act)
        ops.force_plan(0)
    assert ops.gemm_pair_count() == p0 and ops.gemm_pair_count(singles=True) == s0 + 1
    close(dw, want_dw, tol=2e-5 * max(1, M // 64), what="ineligible wgrad")
    close(dx, want_dx, tol=2e-5 * max(1, N // 64), what="eligible dgrad beside an ineligible wgrad")
