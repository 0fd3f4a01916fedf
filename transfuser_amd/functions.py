"""Block-level ``torch.autograd.Function``s of the TransFuser training path.

Each Function runs a whole reference sub-module (RegNetY bottleneck, a GPT fusion stage, a conv
layer, a loss...) as a hand-ordered sequence of HIP launches and implements its backward the
same way.  PyTorch only links the blocks (a few dozen autograd nodes per step), owns the memory
and provides the stream; no ATen compute kernel is used inside a block.

Parameter gradients: by default every block Function returns them to autograd (plain PyTorch semantics: ``loss.backward()`` fills
``p.grad``, DDP / ZeroRedundancyOptimizer / SyncBatchNorm wrappers work - the reference's unmodified train.py loop).  Inside
``transfuser_amd.train.Engine`` they are ACCUMULATED by the kernels straight into ``param.grad`` (views of the flat gradient arena) and the
Functions return ``None`` for parameter inputs: no per-parameter AccumulateGrad pass, one buffer for the all-reduce and the fused AdamW.
"""
import math

import torch

from . import ops


# ---- where parameter gradients go.
# torch mode (default): a Function's backward RETURNS the gradients of the parameters it received as inputs, like any autograd node, so
#   AccumulateGrad runs for every parameter: ``loss.backward()`` fills ``p.grad``, DistributedDataParallel's reducer hooks fire, ``optim.AdamW``
#   / ``ZeroRedundancyOptimizer`` / ``zero_grad(set_to_none=True)`` behave as with the reference's modules (train.py:132-146,304-316).
# in-place mode (inside train.Engine, ``with inplace_param_grads():``): the kernels accumulate straight into ``p.grad`` - views of the flat
#   gradient arena - and the Functions return None for parameters: no per-parameter AccumulateGrad pass, one buffer for RCCL and AdamW.
_INPLACE = False
_collect = None


class inplace_param_grads:
    def __enter__(self):
        global _INPLACE
        self.prev, _INPLACE = _INPLACE, True

    def __exit__(self, *a):
        global _INPLACE
        _INPLACE = self.prev


def gbuf(p):
    """Gradient accumulation buffer of a parameter: the per-backward collector's zero-filled tensor (torch mode) or ``p.grad`` (in-place mode)."""
    if _collect is not None:
        t = _collect.get(id(p))
        if t is None:
            t = _collect[id(p)] = torch.zeros_like(p)      # keeps the memory format (channels_last conv weights)
        return t
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def routes_param_grads(cls):
    """Class decorator for the block Functions: in torch mode the parameter gradients the kernels accumulated during ``backward`` are handed
    back to autograd in the positions of the nn.Parameter inputs."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._tf_ppos = [(i, a) for i, a in enumerate(args) if isinstance(a, torch.nn.Parameter)]
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        global _collect
        if _INPLACE:
            return bwd(ctx, *grads)
        prev, _collect = _collect, {}
        try:
            out = bwd(ctx, *grads)
            out = list(out) if isinstance(out, tuple) else [out]
            for i, prm in ctx._tf_ppos:
                g = _collect.get(id(prm))
                if g is not None and i < len(out):
                    out[i] = g
            return tuple(out)
        finally:
            _collect = prev

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def w2d(w):
    """(Cout, Cin, 1, 1) -> (Cout, Cin) view of a 1x1 conv weight (or its grad)."""
    return w.view(w.shape[0], w.shape[1])


def bias_grad(dy2d, b, mask2d=None):
    ops.colsum(dy2d, 1, dy2d.shape[0], dy2d.shape[1], 1.0, mask=mask2d, out=gbuf(b).view(1, -1), accumulate=True)


def _bn(x, bn, res=None, relu=False, stat=None):
    """``stat``: ops.ColStat gathered by the epilogue of the kernel that produced x (train mode, local statistics only)."""
    grp = getattr(bn, "_sync_group", None)
    if grp is None and isinstance(bn, torch.nn.SyncBatchNorm) and bn.training:     # torch.nn.SyncBatchNorm.convert_sync_batchnorm(model) (train.py:133)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(bn.process_group) > 1:
            grp = bn.process_group if bn.process_group is not None else dist.group.WORLD
    if grp is not None and bn.training:
        return _bn_sync_fwd(x, bn, res, relu, grp)
    if stat is not None and bn.training:
        y, sm, si = ops.bn_fwd_parts(x, stat, bn.weight, bn.bias, bn.running_mean, bn.running_var, res, relu, bn.momentum, bn.eps)
        return y, (sm, si)
    y, sm, si = ops.bn_fwd(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, res, relu, bn.training, bn.momentum, bn.eps)
    return y, (sm, si)


def _bn_bwd(dz, z, x, bn, st, want_dres=False):
    if len(st) == 4:
        return _bn_sync_bwd(dz, z, x, bn, st, want_dres)
    return ops.bn_bwd(dz, z, x, bn.weight, st[0], st[1], gbuf(bn.weight), gbuf(bn.bias), want_dres)


# ---- SyncBatchNorm (--sync_batch_norm 1, train.py:132-133 = torch.nn.SyncBatchNorm.convert_sync_batchnorm): batch statistics over ALL
# ranks.  Composed from the existing kernels plus two tiny collectives per layer; the per-channel algebra between them runs on (C,)
# tensors.  Not hipGraph-capturable (collectives between kernels): the Engine runs eagerly with this flag.
def convert_sync_batchnorm(model, group=None):
    import torch.distributed as dist
    grp = group if group is not None else dist.group.WORLD
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m._sync_group = grp
    return model


def _bn_sync_fwd(x, bn, res, relu, grp):
    import torch.distributed as dist
    C = x.shape[-1]
    n = x.numel() // C
    _, m_loc, i_loc = ops.bn_fwd(x, bn.weight, bn.bias, None, None, None, False, True, bn.momentum, bn.eps)     # local mean / invstd only
    world = dist.get_world_size(grp)
    mine = torch.cat([m_loc.double(), (1.0 / (i_loc.double() * i_loc.double()) - bn.eps).clamp_(min=0.0), torch.full((1,), float(n), dtype=torch.float64, device=x.device)])
    allst = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allst, mine, group=grp)
    cnt = torch.stack([t[2 * C] for t in allst])                       # (world,)
    means = torch.stack([t[:C] for t in allst]); varis = torch.stack([t[C:2 * C] for t in allst])
    N = cnt.sum()
    mean = (means * cnt[:, None]).sum(0) / N                           # Chan's parallel combination, in fp64
    var = ((varis + (means - mean) ** 2) * cnt[:, None]).sum(0) / N    # biased variance of the global batch
    mean32, var32 = mean.float(), var.float()
    y, _, _ = ops.bn_fwd(x, bn.weight, bn.bias, mean32, var32, res, relu, False, bn.momentum, bn.eps)           # normalise with the global statistics
    if bn.running_mean is not None:
        with torch.no_grad():
            mom = bn.momentum
            Nf = float(N)
            bn.running_mean.mul_(1 - mom).add_(mean32, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var32 * (Nf / (Nf - 1)) if Nf > 1 else var32, alpha=mom)
    invstd = (1.0 / torch.sqrt(var + bn.eps)).float()
    return y, (mean32, invstd, float(N), grp)


def _bn_sync_bwd(dz, z, x, bn, st, want_dres):
    import torch.distributed as dist
    mean, invstd, N, grp = st
    C = x.shape[-1]
    n = x.numel() // C
    loc = torch.zeros(2, C, dtype=torch.float32, device=x.device)      # [sum g * xhat, sum g] of THIS rank
    dx_loc, dres = ops.bn_bwd(dz, z, x, bn.weight, mean, invstd, loc[0], loc[1], want_dres)
    ops.axpby(gbuf(bn.weight), loc[0], 1.0, 1.0, out=gbuf(bn.weight))  # local parameter gradients; the gradient all-reduce averages them
    ops.axpby(gbuf(bn.bias), loc[1], 1.0, 1.0, out=gbuf(bn.bias))
    tot = loc.clone()
    dist.all_reduce(tot, group=grp)
    # dx = A (g - S/N - xhat T/N) with the GLOBAL sums; bn_bwd used the local ones (s/n, t/n): add the per-channel affine correction c1 x + c0
    A = bn.weight.detach() * invstd
    dT = tot[0] / N - loc[0] / n
    dS = tot[1] / N - loc[1] / n
    c1 = -A * invstd * dT
    c0 = -A * dS - c1 * mean
    ones = torch.full_like(c1, 1.0 - bn.eps)
    dx, _, _ = ops.bn_fwd(x, c1.contiguous(), c0.contiguous(), torch.zeros_like(c1), ones, dx_loc, False, False, bn.momentum, bn.eps)
    return dx, dres


# ============================================================================================ stem
@routes_param_grads
class StemFn(torch.autograd.Function):
    """First conv (no bias) + BatchNorm + ReLU on the NCHW model input (transfuser.py:136-143): 3x3 / s2 + BatchNormAct2d for the RegNet trunks,
    7x7 / s2 / p3 + bn1 + act1 + the 3x3 / s2 max pool for the ResNet trunks (``stem.maxpool``)."""

    @staticmethod
    def forward(ctx, s0, s1, stem, w, gamma, beta):
        conv = stem.conv
        stride, pad = conv.stride[0], conv.padding[0]
        y = ops.stem_conv_fwd(s0, s1, w, stem.normalize, stride, pad)
        z, st = _bn(y, stem.bn, relu=True)
        out, idx = ops.maxpool3x3s2_fwd(z) if getattr(stem, "maxpool", False) else (z, None)
        ctx.saved = (s0, s1, stem, w, y, z, st, idx, stride, pad)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("StemFn: trying to backward through the graph a second time: the saved activations are freed by the first backward "
                               "(retain_graph is not supported by the block Functions)")
        s0, s1, stem, w, y, z, st, idx, stride, pad = ctx.saved
        ctx.saved = None        # z is also this node's output: drop the ctx <-> output cycle now instead of waiting for the cyclic GC (~150 MB / step)
        dz = dout.contiguous() if idx is None else ops.maxpool3x3s2_bwd(dout.contiguous(), idx, z.shape)
        dy, _ = _bn_bwd(dz, z, y, stem.bn, st)
        ops.stem_conv_wgrad(dy, s0, s1, gbuf(w), stem.normalize, stride=stride, pad=pad)
        return (None,) * 6


@routes_param_grads
class ConvBnFn(torch.autograd.Function):
    """conv (k x k, stride, no bias, groups 1) -> BatchNorm (+ residual) (+ ReLU) on NHWC: the unit the ResNet trunks are made of
    (timm BasicBlock / Bottleneck / downsample; the reference's default architectures, transfuser.py:15).  The BatchNorm statistics come
    from the convolution's epilogue where the engine can produce them."""

    @staticmethod
    def forward(ctx, x, res, conv, bn, relu, w, gamma, beta):
        k, stride, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        B, H, W, Cin = x.shape
        if k == 1 and stride == 1:
            y, cs = ops.linear_fwd(x.view(-1, Cin), w2d(w), colstat=True)
            y = y.view(B, H, W, w.shape[0])
        else:
            y, cs = ops.conv_fwd(x, w, None, stride, pad, 1, colstat=True)
        z, st = _bn(y, bn, res=res, relu=relu, stat=cs)
        ctx.saved = (x, conv, bn, relu, w, y, z, st, res is not None)
        return z

    @staticmethod
    def backward(ctx, dz):
        if ctx.saved is None:
            raise RuntimeError("ConvBnFn: second backward through the same graph (retain_graph is not supported by the block Functions)")
        x, conv, bn, relu, w, y, z, st, has_res = ctx.saved
        ctx.saved = None
        k, stride, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        dy, dres = _bn_bwd(dz.contiguous(), z if relu else None, y, bn, st, want_dres=has_res)
        if k == 1 and stride == 1:
            ops.linear_wgrad(dy.view(-1, Cout), x.view(-1, Cin), w2d(gbuf(w)))
            dx = ops.linear_dgrad(dy.view(-1, Cout), w2d(w)).view(B, H, W, Cin) if ctx.needs_input_grad[0] else None
        else:
            ops.conv_wgrad(dy, x, gbuf(w), stride, pad, 1)
            dx = ops.conv_dgrad(dy, w, x.shape, stride, pad, 1) if ctx.needs_input_grad[0] else None
        return dx, dres, None, None, None, None, None, None


# ============================================================================================ ConvNeXt (timm 0.5.4; transfuser.py:395-416 re-labelling branch)
def _ln_rows(x, ln):
    """LayerNorm over the channels of an NHWC map (timm LayerNorm2d / the block's channels-last nn.LayerNorm): rows = pixels."""
    C = x.shape[-1]
    y, m, r = ops.layernorm_fwd(x.reshape(-1, C), ln.weight.view(-1), ln.bias.view(-1), ln.eps)
    return y.view(x.shape), m, r


@routes_param_grads
class CnxStemFn(torch.autograd.Function):
    """Patchify stem on the NCHW model input: conv k x k / stride k (bias) -> LayerNorm2d (``stem.0`` / ``stem.1`` = conv1 / bn1 after the re-labelling)."""

    @staticmethod
    def forward(ctx, s0, s1, stem, w, b, g, beta):
        conv, ln = stem.conv, stem.bn
        y = ops.stem_conv_fwd(s0, s1, w, stem.normalize, conv.stride[0], conv.padding[0])
        ops.colscale_add(y, None, b, None, out=y)
        z, m, r = _ln_rows(y, ln)
        ctx.saved = (s0, s1, stem, w, b, y, m, r)
        return z

    @staticmethod
    def backward(ctx, dz):
        s0, s1, stem, w, b, y, m, r = ctx.saved
        ctx.saved = None
        conv, ln = stem.conv, stem.bn
        C = y.shape[-1]
        dy = ops.layernorm_bwd(dz.contiguous().view(-1, C), y.view(-1, C), ln.weight.view(-1), m, r, gbuf(ln.weight).view(-1), gbuf(ln.bias).view(-1))
        bias_grad(dy, b)
        ops.stem_conv_wgrad(dy.view(y.shape), s0, s1, gbuf(w), stem.normalize, stride=conv.stride[0], pad=conv.padding[0])
        return (None,) * 7


@routes_param_grads
class CnxDownFn(torch.autograd.Function):
    """Stage entry: LayerNorm2d -> conv 2x2 / stride 2 (bias)  (``stages.i.downsample``)."""

    @staticmethod
    def forward(ctx, x, ds, lw, lb, w, b):
        ln, conv = ds[0], ds[1]
        n, m, r = _ln_rows(x, ln)
        y = ops.conv_fwd(n, w, b, conv.stride[0], 0, 1)
        ctx.saved = (x, ds, n, m, r, w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ds, n, m, r, w, b = ctx.saved
        ctx.saved = None
        ln, conv = ds[0], ds[1]
        dy = dy.contiguous()
        C = x.shape[-1]
        bias_grad(dy.view(-1, dy.shape[-1]), b)
        ops.conv_wgrad(dy, n, gbuf(w), conv.stride[0], 0, 1)
        dn = ops.conv_dgrad(dy, w, n.shape, conv.stride[0], 0, 1)
        dx = ops.layernorm_bwd(dn.view(-1, C), x.reshape(-1, C), ln.weight, m, r, gbuf(ln.weight), gbuf(ln.bias))
        return dx.view(x.shape), None, None, None, None, None


@routes_param_grads
class CnxBlockFn(torch.autograd.Function):
    """ConvNeXtBlock: depthwise 7x7 (bias) -> LayerNorm (eps 1e-6) -> Linear 4x -> GELU -> Linear -> gamma (layer scale) -> + shortcut."""

    @staticmethod
    def forward(ctx, x, blk, *params):
        B, H, W, C = x.shape
        d = ops.dwconv7(x, blk.conv_dw.weight, blk.conv_dw.bias)
        n, m, r = _ln_rows(d, blk.norm)
        n2 = n.view(-1, C)
        h = ops.linear_fwd(n2, blk.mlp.fc1.weight, blk.mlp.fc1.bias)
        a = ops.gelu_fwd(h)
        o = ops.linear_fwd(a, blk.mlp.fc2.weight, blk.mlp.fc2.bias)
        out = ops.colscale_add(o.view(B, H, W, C), blk.gamma, None, x)
        ctx.saved = (x, blk, d, m, r, n2, h, a, o)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, blk, d, m, r, n2, h, a, o = ctx.saved
        ctx.saved = None
        B, H, W, C = x.shape
        dout = dout.contiguous()
        d2 = dout.view(-1, C)
        ops.colsum_mul(d2, o, gbuf(blk.gamma))
        do = ops.colscale_add(d2, blk.gamma)
        fc1, fc2 = blk.mlp.fc1, blk.mlp.fc2
        ops.linear_wgrad(do, a, gbuf(fc2.weight))
        bias_grad(do, fc2.bias)
        da = ops.linear_dgrad(do, fc2.weight)
        dh = ops.gelu_bwd(da, h, out=da)
        ops.linear_wgrad(dh, n2, gbuf(fc1.weight))
        bias_grad(dh, fc1.bias)
        dn = ops.linear_dgrad(dh, fc1.weight)
        dd = ops.layernorm_bwd(dn, d.view(-1, C), blk.norm.weight, m, r, gbuf(blk.norm.weight), gbuf(blk.norm.bias)).view(B, H, W, C)
        ops.dwconv7_wgrad(dd, x, gbuf(blk.conv_dw.weight), gbuf(blk.conv_dw.bias))
        dx = ops.colscale_add(dout)                     # the shortcut's share (a copy: the depthwise gradient is accumulated into it)
        ops.dwconv7(dd, blk.conv_dw.weight, None, flip=True, out=dx, accumulate=True)
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


@routes_param_grads
class PoolNormFn(torch.autograd.Function):
    """global average pool -> LayerNorm over the pooled channel vector (the re-labelled ConvNeXt ``global_pool`` = head: pool + LayerNorm((512, 1, 1)))."""

    @staticmethod
    def forward(ctx, x, ln, w, b):
        B, H, W, C = x.shape
        p = ops.colsum(x, B, H * W, C, 1.0 / (H * W))
        y, m, r = ops.layernorm_fwd(p, w.view(-1), b.view(-1), ln.eps)
        ctx.saved = (x.shape, w, b, p, m, r)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, w, b, p, m, r = ctx.saved
        ctx.saved = None
        dp = ops.layernorm_bwd(dy.contiguous(), p, w.view(-1), m, r, gbuf(w).view(-1), gbuf(b).view(-1))
        return ops.se_scale_bwd_x(None, None, dp, shape), None, None, None


# ---- 1x1 convolutions of the trunks.  (A 16-bit STORED-operand form of these with a cast16 launch per operand was measured at 39.1 vs 38.3 ms/step
# in bf16 - two cast passes per convolution cost more than the packed GEMMs gain - and removed; YBlockFn's "lp" path below is its successor: the
# copies come out of the element-wise passes that produce the operands anyway.)
def _c1x1_fwd(x2, w, colstat=True):
    """y (M, N) = x2 (M, K) @ w (N, K)^T  (+ BatchNorm statistics of y); returns (y, ColStat or None, saved input)."""
    if colstat:
        y, cs = ops.linear_fwd(x2, w, colstat=True)
        return y, cs, x2
    return ops.linear_fwd(x2, w), None, x2


def _c1x1_bwd(dy2, saved, w, dw, res=None):
    """dW += dy^T x, returns dx = dy W (+ res)."""
    with ops.gemm_pair(dy2):      # weight + input gradient in one grid where both plans are 64 x 64 tilings (csrc/gemm_pair.cpp)
        ops.linear_wgrad(dy2, saved, dw)
        dx = ops.linear_dgrad(dy2, w, res=res)
    return dx


def _c1x1_bwd16(d16, d16t, xt16, w, dw, res=None):
    """16-bit storage form of _c1x1_bwd: dW += dy16^T . x16 (both through their transposed copies, contraction over the rows padded to 8),
    returns dx = dy16 . W16 (+ res) in fp32."""
    ops.gemm16_nt(d16t, xt16, dw, accumulate=True, k=d16t.shape[1])
    _, w16t = ops.lowp_weight(w)
    out = torch.empty(d16.shape[0], w.shape[1], dtype=torch.float32, device=d16.device)
    return ops.gemm16_nt(d16, w16t, out, res=res, k=w.shape[0])


# ============================================================================================ RegNetY block
@routes_param_grads
class YBlockFn(torch.autograd.Function):
    """timm Bottleneck: 1x1 -> BN/ReLU -> grouped 3x3 (stride) -> BN/ReLU -> SE ->
    1x1 -> BN (+ shortcut / 1x1-s2 downsample BN) -> ReLU."""

    @staticmethod
    def forward(ctx, x, blk, *params):
        B, H, W, Cin = x.shape
        C = blk.out_chs
        x2 = x.view(-1, Cin)
        bn1 = blk.conv1.bn
        # 16-bit storage modes (round 5): conv1 / conv3 as packed-16 NT GEMMs on STORED operands whose copies are written by their producers
        # (the previous block's output pass, the SE-scale pass, the two BatchNorm backward applies: ops.bn_apply16 / se_scale_bn16 / bn_bwd16 /
        # bn_bwd_remask16); where a producer variant has no such form, a cast16 launch makes the copies
        lp = (bool(ops.lowp_conv()) and Cin % 8 == 0 and C % 8 == 0 and bn1.training and blk.conv2.bn.training and blk.conv3.bn.training and x.is_contiguous() and
              B * ((H - 1) // blk.stride + 1) * ((W - 1) // blk.stride + 1) >= 32 and      # <= 16 rows: the fp32 small-M kernels (exact, latency-sized) keep the layer
              ops.want_colstat(B * H * W))      # stage 1 (> 40 000 rows: no statistics epilogue, so no folded producers) would need a cast launch per operand
        x16t = z16t = None
        if lp:
            pre = getattr(x, "_lp16", None)
            if pre is not None and (pre[0] != x._version or pre[1].dtype != ops._t16()):      # the tensor was modified in place since its producer wrote the copies
                pre = None                                                                    # (or the storage mode changed): stale copies are never multiplied
            x16, x16t = pre[1:] if pre is not None else ops.cast16(x2)
            y1, cs1 = ops.gemm16_nt_colstat(x16, ops.lowp_weight(w2d(blk.conv1.conv.weight))[0], torch.empty(B * H * W, C, dtype=torch.float32, device=x.device))
            x2s = None
        else:
            y1, cs1, x2s = _c1x1_fwd(x2, w2d(blk.conv1.conv.weight))      # BN statistics gathered by the GEMM epilogue
        y1 = y1.view(B, H, W, C)
        fuse1 = (cs1 is not None and bn1.training and getattr(bn1, "_sync_group", None) is None and not isinstance(bn1, torch.nn.SyncBatchNorm) and
                 ops.grouped_bnrelu_ok(y1.shape, C, blk.groups, blk.stride))
        if fuse1:
            # BatchNorm apply folded into the consumer: the grouped conv2 (and, backward, its weight gradient and the BatchNorm backward's mask)
            # recompute z1 = relu(bn1(y1)) from the raw conv1 output - z1 is never written (st1 = (mean, invstd, [scale | shift]))
            coef1, sm1, si1 = ops.bn_finalize_parts(cs1, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.momentum, bn1.eps)
            st1, z1 = (sm1, si1, coef1), None
            y2, cs2 = ops.grouped_bnrelu_fwd(y1, coef1, blk.conv2.conv.weight, blk.stride)
        else:
            z1, st1 = _bn(y1, bn1, relu=True, stat=cs1)
            y2, cs2 = ops.conv_fwd(z1, blk.conv2.conv.weight, None, blk.stride, 1, blk.groups, colstat=True)
        _, Ho, Wo, _ = y2.shape
        bn2 = blk.conv2.bn
        fuse2 = (ops.FUSE_BN_SE and cs2 is not None and bn2.training and B <= 16 and C <= ops.SE_FUSED_MAX_C and B * blk.se.fc1.weight.shape[0] <= 8192 and
                 getattr(bn2, "_sync_group", None) is None and not isinstance(bn2, torch.nn.SyncBatchNorm))
        if fuse2:
            # BatchNorm apply folded into its consumers: z2 = relu(bn2(y2)) is recomputed by the SE squeeze, the SE scale and (backward) the gate
            # gradient / the BatchNorm backward's mask - never written; the squeeze's chunk sums are finished inside the excitation kernel
            coef2, sm2, si2 = ops.bn_finalize_parts(cs2, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, bn2.momentum, bn2.eps)
            st2, z2 = (sm2, si2, coef2), None
            s, g1, gate = ops.se_squeeze_excite_bn_fwd(y2, coef2, blk.se.fc1.weight, blk.se.fc1.bias, blk.se.fc2.weight, blk.se.fc2.bias)
            if lp:
                z16, z16t = ops.se_scale_bn16(y2, coef2, gate)      # conv3's operand copies straight from the SE-scale pass: no fp32 z2s
                z2s = None
            else:
                z2s = ops.se_scale_bn_fwd(y2, coef2, gate)
        else:
            z2, st2 = _bn(y2, bn2, relu=True, stat=cs2)
            s = ops.colsum(z2, B, Ho * Wo, C, 1.0 / (Ho * Wo))
            if B <= 16 and C <= ops.SE_FUSED_MAX_C:
                g1, gate = ops.se_excite_fwd(s, blk.se.fc1.weight, blk.se.fc1.bias, blk.se.fc2.weight, blk.se.fc2.bias)
            else:
                g1 = ops.linear_fwd(s, w2d(blk.se.fc1.weight), blk.se.fc1.bias, relu=True)
                gate = ops.linear_fwd(g1, w2d(blk.se.fc2.weight), blk.se.fc2.bias)
            z2s = ops.se_scale_fwd(z2, gate)
        if lp:
            if z2s is not None:
                z16, z16t = ops.cast16(z2s.view(-1, C))
            y3, cs3 = ops.gemm16_nt_colstat(z16, ops.lowp_weight(w2d(blk.conv3.conv.weight))[0], torch.empty(B * Ho * Wo, C, dtype=torch.float32, device=x.device))
            z2ss = None
        else:
            y3, cs3, z2ss = _c1x1_fwd(z2s.view(-1, C), w2d(blk.conv3.conv.weight))
        y3 = y3.view(B, Ho, Wo, C)
        yd = std = None
        if blk.downsample is not None:
            if blk.stride == 1:
                yd, csd = ops.linear_fwd(x2, w2d(blk.downsample.conv.weight), colstat=True)
                yd = yd.view(B, H, W, C)
            else:
                yd, csd = ops.conv_fwd(x, blk.downsample.conv.weight, None, blk.stride, 0, 1, colstat=True)
            sc, std = _bn(yd, blk.downsample.bn, relu=False, stat=csd)
        else:
            sc = x
        bn3 = blk.conv3.bn
        if lp and cs3 is not None and getattr(bn3, "_sync_group", None) is None and not isinstance(bn3, torch.nn.SyncBatchNorm):
            # the block output pass also writes the 16-bit copies the NEXT bottleneck's conv1 multiplies (handed over as an attribute of the tensor)
            coef3, sm3, si3 = ops.bn_finalize_parts(cs3, bn3.weight, bn3.bias, bn3.running_mean, bn3.running_var, bn3.momentum, bn3.eps)
            out, o16, o16t = ops.bn_apply16(y3, coef3, sc, True)
            st3 = (sm3, si3)
            out._lp16 = (out._version, o16, o16t)
        else:
            out, st3 = _bn(y3, bn3, res=sc, relu=True, stat=cs3)
        ctx.saved = (x, blk, y1, z1, st1, y2, z2, st2, s, g1, gate, z2ss, y3, st3, yd, std, out, x2s, (lp, x16t, z16t))
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("YBlockFn: trying to backward through the graph a second time: the saved activations are freed by the first backward "
                               "(retain_graph is not supported by the block Functions)")
        x, blk, y1, z1, st1, y2, z2, st2, s, g1, gate, z2ss, y3, st3, yd, std, out, x2s, (lp, x16t, z16t) = ctx.saved
        B, H, W, Cin = x.shape
        _, Ho, Wo, C = out.shape
        x2 = x.view(-1, Cin)
        w3 = blk.conv3.conv.weight
        if lp:
            bn3 = blk.conv3.bn
            if len(st3) == 2:       # the gradient entering conv3 leaves the BatchNorm backward as 16-bit copies (no fp32 dy3)
                _, d16, d16t, dsc = ops.bn_bwd16(dout.contiguous(), out, y3, bn3.weight, st3[0], st3[1], gbuf(bn3.weight), gbuf(bn3.bias), want_dres=True)
            else:
                dy3, dsc = _bn_bwd(dout.contiguous(), out, y3, bn3, st3, want_dres=True)
                d16, d16t = ops.cast16(dy3.view(-1, C))
            dz2s = _c1x1_bwd16(d16, d16t, z16t, w2d(w3), w2d(gbuf(w3))).view(B, Ho, Wo, C)
        else:
            dy3, dsc = _bn_bwd(dout.contiguous(), out, y3, blk.conv3.bn, st3, want_dres=True)
            dy3_2 = dy3.view(-1, C)
            dz2s = _c1x1_bwd(dy3_2, z2ss, w2d(w3), w2d(gbuf(w3))).view(B, Ho, Wo, C)
        # squeeze-excite
        se = blk.se
        if z2 is None:      # forward ran with the BatchNorm apply folded into the consumers (st2 = (mean, invstd, [scale | shift]))
            ds = ops.se_gate_excite_bn_bwd(dz2s, y2, st2[2], gate, s, g1, se.fc1.weight, se.fc2.weight, gbuf(se.fc1.weight), gbuf(se.fc1.bias),
                                           gbuf(se.fc2.weight), gbuf(se.fc2.bias))
            bn2 = blk.conv2.bn
            if ops.FUSE_SE_BN_BWD:      # the SE scale's backward (dz2 = dz2s * sigmoid(gate) + ds / HW) recomputed inside the BatchNorm backward's two passes
                dy2 = ops.bn_bwd_remask_se(dz2s, gate, ds, y2, st2[2], bn2.weight, st2[0], st2[1], gbuf(bn2.weight), gbuf(bn2.bias))
            else:
                dz2 = ops.se_scale_bwd_x(dz2s, gate, ds, y2.shape)
                dy2 = ops.bn_bwd_remask(dz2, y2, st2[2], bn2.weight, st2[0], st2[1], gbuf(bn2.weight), gbuf(bn2.bias))
        else:
            dgate = ops.se_scale_bwd_gate(dz2s, z2, gate)
            if B <= 16 and C <= ops.SE_FUSED_MAX_C and B * g1.shape[1] <= 8192:
                ds = ops.se_excite_bwd(dgate, s, g1, se.fc1.weight, se.fc2.weight, gbuf(se.fc1.weight), gbuf(se.fc1.bias), gbuf(se.fc2.weight),
                                       gbuf(se.fc2.bias))
            else:
                ops.linear_wgrad(dgate, g1, w2d(gbuf(se.fc2.weight)))
                bias_grad(dgate, se.fc2.bias)
                dg1 = ops.linear_dgrad(dgate, w2d(se.fc2.weight))
                dg1 = ops.relu_mask(dg1, g1, out=dg1)
                ops.linear_wgrad(dg1, s, w2d(gbuf(se.fc1.weight)))
                bias_grad(dg1, se.fc1.bias)
                ds = ops.linear_dgrad(dg1, w2d(se.fc1.weight))
            dz2 = ops.se_scale_bwd_x(dz2s, gate, ds, z2.shape)
            dy2, _ = _bn_bwd(dz2, z2, y2, blk.conv2.bn, st2)
        # grouped 3x3
        w2 = blk.conv2.conv.weight
        if z1 is None:      # forward ran with bn1's apply folded into conv2
            bn1 = blk.conv1.bn
            ops.grouped_bnrelu_wgrad(dy2, y1, st1[2], gbuf(w2), stride=blk.stride)
            dz1 = ops.conv_dgrad(dy2, w2, y1.shape, blk.stride, 1, blk.groups)
            if lp:
                _, e16, e16t = ops.bn_bwd_remask16(dz1, y1, st1[2], bn1.weight, st1[0], st1[1], gbuf(bn1.weight), gbuf(bn1.bias))
            else:
                dy1 = ops.bn_bwd_remask(dz1, y1, st1[2], bn1.weight, st1[0], st1[1], gbuf(bn1.weight), gbuf(bn1.bias))
        else:
            ops.conv_wgrad(dy2, z1, gbuf(w2), blk.stride, 1, blk.groups)
            dz1 = ops.conv_dgrad(dy2, w2, z1.shape, blk.stride, 1, blk.groups)
            dy1, _ = _bn_bwd(dz1, z1, y1, blk.conv1.bn, st1)
            if lp:
                e16, e16t = ops.cast16(dy1.view(-1, C))
        w1 = blk.conv1.conv.weight
        if lp:
            c1b = lambda res=None: _c1x1_bwd16(e16, e16t, x16t, w2d(w1), w2d(gbuf(w1)), res=res)
        else:
            dy1_2 = dy1.view(-1, C)
            c1b = lambda res=None: _c1x1_bwd(dy1_2, x2s, w2d(w1), w2d(gbuf(w1)), res=res)
        if blk.downsample is None:
            dx = c1b(dsc.view(-1, Cin))
        else:
            dx = c1b()
            dyd, _ = _bn_bwd(dsc, None, yd, blk.downsample.bn, std)
            wd = blk.downsample.conv.weight
            if blk.stride == 1:
                ops.linear_wgrad(dyd.view(-1, C), x2, w2d(gbuf(wd)))
                ops.linear_dgrad(dyd.view(-1, C), w2d(wd), out=dx, accumulate=True)
            else:
                ops.conv_wgrad(dyd, x, gbuf(wd), blk.stride, 0, 1)
                ops.conv_dgrad(dyd, wd, x.shape, blk.stride, 0, 1, out=dx.view(B, H, W, Cin), accumulate=True)
        ctx.saved = None
        return (dx.view(B, H, W, Cin), None) + (None,) * (len(ctx.needs_input_grad) - 2)


# ============================================================================================ GPT fusion stage
def _attn_fwd(qkv, B, T, C, nh, drop=None):
    """qkv (B*T, 3C) = [key | query | value] (transfuser.py:500-502 order). Returns probs (B*nh, T, Tp), Tp [, dropped probs when
    ``drop`` = (seed, site, p): softmax + attn_drop in one launch]."""
    hs = C // nh
    Tp = (T + 3) // 4 * 4
    att = torch.empty(B * nh, T, Tp, dtype=torch.float32, device=qkv.device)
    k, q, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    sq, sp = (T * 3 * C, hs), (nh * T * Tp, T * Tp)
    ops.gemm(q, k, att, T, T, hs, 3 * C, 3 * C, Tp, alpha=1.0 / math.sqrt(hs), batch=B * nh, inner=nh, sa=sq, sb=sq, sc=sp)
    if drop is not None:
        return att, Tp, ops.softmax_dropout_fwd_(att, B * nh * T, T, Tp, *drop)
    ops.softmax_fwd_(att, B * nh * T, T, Tp)
    return att, Tp


def _attn_ctx(att, qkv, B, T, C, nh, Tp):
    hs = C // nh
    y = torch.empty(B * T, C, dtype=torch.float32, device=qkv.device)
    ops.gemm(att, qkv[:, 2 * C:], y, T, hs, T, Tp, 3 * C, C, b_trans=True, batch=B * nh, inner=nh, sa=(nh * T * Tp, T * Tp),
             sb=(T * 3 * C, hs), sc=(T * C, hs))
    return y


# The stage is written as three helpers per direction (embed, one Block, output) that BOTH the one-node GPTStageFn (single GPU: the whole
# stage is one autograd node) and the split Functions below (multi-GPU: train.Engine may cut the backward between two Blocks of a
# stage, so that the 27.5 M-parameter Blocks of GPT-4 are all-reduced one by one while the next one is differentiated) run.
def _gpt_dims(gpt, s_img, s_lid):
    cfg = gpt.geom
    n_img, n_lid = cfg.ih * cfg.iw, cfg.lh * cfg.lw
    return cfg, s_img[0], s_img[3], n_img, n_lid, n_img + n_lid


def _gpt_embed_fwd(gpt, x_img, x_lid, velocity):
    cfg, B, C, n_img, n_lid, T = _gpt_dims(gpt, x_img.shape, x_lid.shape)
    tok = torch.empty(B, T, C, dtype=torch.float32, device=x_img.device)
    bvec = None
    if gpt.use_velocity:
        bvec = ops.linear_fwd(velocity, gpt.vel_emb.weight, gpt.vel_emb.bias)
    ops.pool_tokens_fwd(x_img, cfg.ih, cfg.iw, gpt.pos_emb, tok, 0, bvec)
    ops.pool_tokens_fwd(x_lid, cfg.lh, cfg.lw, gpt.pos_emb, tok, n_img, bvec)
    x = tok.view(B * T, C)
    if gpt.training and gpt.pdrop_any and gpt.embd_pdrop > 0:
        x = ops.dropout(x, gpt.seed, gpt.site(0), gpt.embd_pdrop)
    return x


def _gpt_embed_bwd(gpt, dx, s_img, s_lid, velocity, drop, add_img=None, add_lid=None):
    """dx (B*T, C) is consumed (modified in place).  Returns the gradients of the two feature maps: add_* + pool^T(dtok)."""
    cfg, B, C, n_img, n_lid, T = _gpt_dims(gpt, s_img, s_lid)
    if drop and gpt.embd_pdrop > 0:
        ops.dropout(dx, gpt.seed, gpt.site(0), gpt.embd_pdrop, out=dx)
    dtok = dx.view(B, T, C)
    ops.colsum(dtok, 1, B, T * C, 1.0, out=gbuf(gpt.pos_emb).view(1, -1), accumulate=True)
    if gpt.use_velocity:
        db = ops.colsum(dtok, B, T, C, 1.0)
        ops.linear_wgrad(db, velocity, gbuf(gpt.vel_emb.weight))
        bias_grad(db, gpt.vel_emb.bias)
    dx_img = ops.pool_tokens_bwd(dtok, s_img, cfg.ih, cfg.iw, 0, add=add_img)
    dx_lid = ops.pool_tokens_bwd(dtok, s_lid, cfg.lh, cfg.lw, n_img, add=add_lid)
    return dx_img, dx_lid


# ---- 16-bit operand STORAGE for the linear layers of a Block (ops.lowp_storage(): precision "bf16" / "fp16").  Every contraction becomes an
# NT product of 16-bit matrices (ops.gemm16_nt): the forward multiplies the cast activation with the cached 16-bit weight, the backward
# casts dy once into (dy16, dy16^T) and multiplies dy16^T with the SAVED transposed activation copy (weight gradient) and dy16 with the
# transposed weight copy (input gradient).  What a Block keeps for its backward is therefore the transposed 16-bit copy, not the fp32 tensor.
class A16:
    """Saved-for-backward stand-in of an activation: its transposed 16-bit copy (K, M8) [+ the fp32 tensor when a ReLU mask needs it]."""
    __slots__ = ("t", "f32")

    def __init__(self, t, f32=None):
        self.t, self.f32 = t, f32


def _lin16_fwd(x16, w, bias=None, relu=False, res=None, out=None):
    w16, _ = ops.lowp_weight(w)
    if out is None:
        out = torch.empty(x16.shape[0], w.shape[0], dtype=torch.float32, device=x16.device)
    return ops.gemm16_nt(x16, w16, out, bias=bias, res=res, relu=relu)


def _lin16_bwd(dy, xa, w, dw, mask=None, out=None, accumulate=False):
    """dW += dy^T x (x = the A16 saved by the forward), returns dx (+)= dy W (masked by ``mask`` > 0)."""
    d16, d16t = ops.cast16(dy)
    ops.gemm16_nt(d16t, xa.t, dw, accumulate=True, k=d16t.shape[1])
    _, w16t = ops.lowp_weight(w)
    if out is None:
        out = torch.empty(dy.shape[0], w.shape[1], dtype=torch.float32, device=dy.device)
    return ops.gemm16_nt(d16, w16t, out, mask=mask, accumulate=accumulate, k=w.shape[0])


def _lowp_block(gpt, blk, C):
    return bool(ops.lowp_storage()) and C % 8 == 0 and blk.mlp[0].weight.shape[0] % 8 == 0


def _gpt_block_fwd(gpt, li, x, B, T, drop):
    """One transformer Block (transfuser.py:545-549) on x (B*T, C); returns (x_out, saved)."""
    blk = gpt.blocks[li]
    C, nh, dev = x.shape[1], gpt.n_head, x.device
    lowp = _lowp_block(gpt, blk, C)
    ln16 = lowp and ops.LN_FWD16 and ops.layernorm_fwd16_ok(x, blk.ln1.weight, blk.ln1.bias) and ops.layernorm_fwd16_ok(x, blk.ln2.weight, blk.ln2.bias)      # 16-bit storage: LayerNorm writes the operand copies itself (no fp32 h, no cast launch)
    if ln16:
        h1_16, h1_t, m1, r1 = ops.layernorm_fwd16(x, blk.ln1.weight, blk.ln1.bias, blk.ln1.eps)
    else:
        h1, m1, r1 = ops.layernorm_fwd(x, blk.ln1.weight, blk.ln1.bias, blk.ln1.eps)
    qkv = torch.empty(B * T, 3 * C, dtype=torch.float32, device=dev)
    fw = blk.attn.fused()
    if lowp:
        if not ln16:
            h1_16, h1_t = ops.cast16(h1)
        if fw is not None:
            _lin16_fwd(h1_16, fw[0], fw[1], out=qkv)
        else:
            for j, l3 in enumerate((blk.attn.key, blk.attn.query, blk.attn.value)):
                _lin16_fwd(h1_16, l3.weight, l3.bias, out=qkv[:, j * C:(j + 1) * C])
        h1 = A16(h1_t)
    elif fw is not None:
        ops.linear_fwd(h1, fw[0], fw[1], out=qkv)
    else:
        for j, lin in enumerate((blk.attn.key, blk.attn.query, blk.attn.value)):
            ops.linear_fwd(h1, lin.weight, lin.bias, out=qkv[:, j * C:(j + 1) * C])
    adrop = (gpt.seed, gpt.site(4 * li + 1), gpt.attn_pdrop) if (drop and gpt.attn_pdrop > 0) else None
    if ops.attention_supported(T, C, nh):
        # one launch: QK^T -> softmax -> attn_drop -> PV per (sample, head, 32-query tile); only the row-wise log-sum-exp is kept
        y_att, att = ops.attention_fwd(qkv, B, T, C, nh, adrop)
        att_d, Tp = None, 0
    elif adrop is not None:
        att, Tp, att_d = _attn_fwd(qkv, B, T, C, nh, drop=adrop)
        y_att = _attn_ctx(att_d, qkv, B, T, C, nh, Tp)
    else:
        att, Tp = _attn_fwd(qkv, B, T, C, nh)
        att_d = att
        y_att = _attn_ctx(att_d, qkv, B, T, C, nh, Tp)
    rdrop = drop and gpt.resid_pdrop > 0
    if lowp:
        ya_16, ya_t = ops.cast16(y_att)
        y_att = A16(ya_t, y_att if att_d is None else None)      # fused attention: its backward forms D = dY . Y from the fp32 output
        lin = lambda a16, layer, **kw: _lin16_fwd(a16, layer.weight, layer.bias, **kw)
    else:
        ya_16 = y_att
        lin = lambda a, layer, **kw: ops.linear_fwd(a, layer.weight, layer.bias, **kw)
    fdrop = rdrop and not lowp and ops.FUSE_DROPOUT      # fp32-stored operands: resid_drop + the residual add ride in the GEMM's epilogue (same mask)
    if fdrop:
        x_mid = lin(ya_16, blk.attn.proj, res=x, drop=(gpt.seed, gpt.site(4 * li + 2), gpt.resid_pdrop))
    elif rdrop:
        pr = lin(ya_16, blk.attn.proj)
        x_mid = ops.dropout_add(pr, x, gpt.seed, gpt.site(4 * li + 2), gpt.resid_pdrop, out=pr)
    else:
        x_mid = lin(ya_16, blk.attn.proj, res=x)
    if ln16:
        h2_16, h2_t, m2, r2 = ops.layernorm_fwd16(x_mid, blk.ln2.weight, blk.ln2.bias, blk.ln2.eps)
    else:
        h2, m2, r2 = ops.layernorm_fwd(x_mid, blk.ln2.weight, blk.ln2.bias, blk.ln2.eps)
    if lowp:
        if not ln16:
            h2_16, h2_t = ops.cast16(h2)
        a1 = lin(h2_16, blk.mlp[0], relu=True)
        h2 = A16(h2_t)
        a1_16, a1_t = ops.cast16(a1)
        a1s = A16(a1_t, a1)
    else:
        a1 = a1_16 = a1s = lin(h2, blk.mlp[0], relu=True)
    if fdrop:
        x_out = lin(a1_16, blk.mlp[2], res=x_mid, drop=(gpt.seed, gpt.site(4 * li + 3), gpt.resid_pdrop))
    elif rdrop:
        f2 = lin(a1_16, blk.mlp[2])
        x_out = ops.dropout_add(f2, x_mid, gpt.seed, gpt.site(4 * li + 3), gpt.resid_pdrop, out=f2)
    else:
        x_out = lin(a1_16, blk.mlp[2], res=x_mid)
    return x_out, (x, h1, m1, r1, qkv, att, att_d, Tp, y_att, x_mid, h2, m2, r2, a1s)


def _gpt_block_bwd(gpt, li, saved, dx, B, T, drop):
    """Backward of one Block: dx (B*T, C) = gradient of the Block's output, updated IN PLACE to the gradient of its input."""
    blk = gpt.blocks[li]
    x, h1, m1, r1, qkv, att, att_d, Tp, y_att, x_mid, h2, m2, r2, a1 = saved
    C, nh = x.shape[1], gpt.n_head
    hs = C // nh
    alpha = 1.0 / math.sqrt(hs)
    fc1, fc2, proj = blk.mlp[0], blk.mlp[2], blk.attn.proj
    # The four bias gradients of the Block are column sums over the same B*T rows: they are collected and reduced by ONE single-pass launch
    # at the end (ops.colsum_multi) instead of a reduce + finalize pair each.  A gradient that aliases dx (no residual dropout) is summed
    # right away: dx is updated in place further down.
    pending = []

    def bias_later(dy2d, b, fresh=True):
        if fresh and ops.COLSUM_MULTI:
            pending.append((dy2d, gbuf(b).view(-1)))
        else:
            bias_grad(dy2d, b)

    # ---- MLP: x_out = x_mid + drop(fc2(relu(fc1(ln2(x_mid)))))
    dres = dx
    if drop and gpt.resid_pdrop > 0:
        dres = ops.dropout(dx, gpt.seed, gpt.site(4 * li + 3), gpt.resid_pdrop)
    lowp = isinstance(a1, A16)      # the forward ran on 16-bit stored operands: so does the backward
    if lowp:
        bias_later(dres, fc2.bias, dres is not dx)
        da1 = _lin16_bwd(dres, a1, fc2.weight, gbuf(fc2.weight), mask=a1.f32)     # ReLU backward fused into the dgrad epilogue
        bias_later(da1, fc1.bias)
        dh2 = _lin16_bwd(da1, h2, fc1.weight, gbuf(fc1.weight))
    else:
        # (weight gradient, input gradient) of a layer: one grid where both plans are 64 x 64 tilings (the narrow stages; ops.gemm_pair)
        with ops.gemm_pair(dres):
            ops.linear_wgrad(dres, a1, gbuf(fc2.weight))
            da1 = ops.linear_dgrad(dres, fc2.weight, mask=a1)     # ReLU backward fused into the dgrad epilogue
        bias_later(dres, fc2.bias, dres is not dx)
        with ops.gemm_pair(da1):
            ops.linear_wgrad(da1, h2, gbuf(fc1.weight))
            dh2 = ops.linear_dgrad(da1, fc1.weight)
        bias_later(da1, fc1.bias)
    # dx_mid = dx + ln2_bwd(dh2): accumulate in place into dx
    # ---- attention: x_mid = x + drop(proj(att @ v)); the launch that finishes dx_mid also writes resid_drop's gradient of it (same mask as the forward)
    if drop and gpt.resid_pdrop > 0 and ops.FUSE_DROPOUT:
        _, dres = ops.layernorm_bwd(dh2, x_mid, blk.ln2.weight, m2, r2, gbuf(blk.ln2.weight), gbuf(blk.ln2.bias), dx=dx, accumulate=True,
                                    drop=(gpt.seed, gpt.site(4 * li + 2), gpt.resid_pdrop))
    else:
        ops.layernorm_bwd(dh2, x_mid, blk.ln2.weight, m2, r2, gbuf(blk.ln2.weight), gbuf(blk.ln2.bias), dx=dx, accumulate=True)
        dres = dx
        if drop and gpt.resid_pdrop > 0:
            dres = ops.dropout(dx, gpt.seed, gpt.site(4 * li + 2), gpt.resid_pdrop)
    bias_later(dres, proj.bias, dres is not dx)
    if lowp:
        dy = _lin16_bwd(dres, y_att, proj.weight, gbuf(proj.weight))
    else:
        with ops.gemm_pair(dres):
            ops.linear_wgrad(dres, y_att, gbuf(proj.weight))
            dy = ops.linear_dgrad(dres, proj.weight)
    if att_d is None:       # fused attention: att is the log-sum-exp; probabilities are recomputed inside the two backward kernels
        adrop = (gpt.seed, gpt.site(4 * li + 1), gpt.attn_pdrop) if (drop and gpt.attn_pdrop > 0) else None
        dqkv = ops.attention_bwd(qkv, dy, att, B, T, C, nh, adrop, y=y_att.f32 if isinstance(y_att, A16) else y_att)
    else:
        dqkv = torch.empty_like(qkv)
        datt = torch.empty_like(att)
        k, q, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        sq, sp, sy = (T * 3 * C, hs), (nh * T * Tp, T * Tp), (T * C, hs)
        ops.gemm(dy, v, datt, T, T, hs, C, 3 * C, Tp, batch=B * nh, inner=nh, sa=sy, sb=sq, sc=sp)                       # dP = dY V^T
        ops.gemm(att_d, dy, dqkv[:, 2 * C:], T, hs, T, Tp, C, 3 * C, a_trans=True, b_trans=True, batch=B * nh, inner=nh,
                 sa=sp, sb=sy, sc=sq)                                                                                     # dV = P^T dY
        if drop and gpt.attn_pdrop > 0:
            ops.softmax_dropout_bwd_(att, datt, B * nh * T, T, Tp, gpt.seed, gpt.site(4 * li + 1), gpt.attn_pdrop)
        else:
            ops.softmax_bwd_(att, datt, B * nh * T, T, Tp)
        ops.gemm(datt, k, dqkv[:, C:2 * C], T, hs, T, Tp, 3 * C, 3 * C, b_trans=True, alpha=alpha, batch=B * nh, inner=nh,
                 sa=sp, sb=sq, sc=sq)                                                                                     # dQ = dS K
        ops.gemm(datt, q, dqkv[:, :C], T, hs, T, Tp, 3 * C, 3 * C, a_trans=True, b_trans=True, alpha=alpha, batch=B * nh, inner=nh,
                 sa=sp, sb=sq, sc=sq)                                                                                     # dK = dS^T Q
    fw = blk.attn.fused()
    if lowp and fw is not None:
        if ops.COLSUM_MULTI:
            pending.append((dqkv, fw[3].view(-1)))
        else:
            ops.colsum(dqkv, 1, B * T, 3 * C, 1.0, out=fw[3].view(1, -1), accumulate=True)
        dh1 = _lin16_bwd(dqkv, h1, fw[0], fw[2])
    elif lowp:
        dh1 = None
        for j, l3 in enumerate((blk.attn.key, blk.attn.query, blk.attn.value)):
            dh1 = _lin16_bwd(dqkv[:, j * C:(j + 1) * C], h1, l3.weight, gbuf(l3.weight), out=dh1, accumulate=dh1 is not None)
        b3 = ops.colsum(dqkv, 1, B * T, 3 * C, 1.0)
        for j, l3 in enumerate((blk.attn.key, blk.attn.query, blk.attn.value)):
            ops.axpby(gbuf(l3.bias), b3[0, j * C:(j + 1) * C], 1.0, 1.0, out=gbuf(l3.bias))
    elif fw is not None:
        with ops.gemm_pair(dqkv):
            ops.linear_wgrad(dqkv, h1, fw[2])
            dh1 = ops.linear_dgrad(dqkv, fw[0])
        if ops.COLSUM_MULTI:
            pending.append((dqkv, fw[3].view(-1)))
        else:
            ops.colsum(dqkv, 1, B * T, 3 * C, 1.0, out=fw[3].view(1, -1), accumulate=True)
    else:
        dh1 = None
        for j, lin in enumerate((blk.attn.key, blk.attn.query, blk.attn.value)):
            dj = dqkv[:, j * C:(j + 1) * C]
            ops.linear_wgrad(dj, h1, gbuf(lin.weight))
            dh1 = ops.linear_dgrad(dj, lin.weight, out=dh1, accumulate=dh1 is not None)
        b3 = ops.colsum(dqkv, 1, B * T, 3 * C, 1.0)
        for j, lin in enumerate((blk.attn.key, blk.attn.query, blk.attn.value)):
            ops.axpby(gbuf(lin.bias), b3[0, j * C:(j + 1) * C], 1.0, 1.0, out=gbuf(lin.bias))
    ops.layernorm_bwd(dh1, x, blk.ln1.weight, m1, r1, gbuf(blk.ln1.weight), gbuf(blk.ln1.bias), dx=dx, accumulate=True)
    if pending:
        ops.colsum_multi(pending)
    return dx


def _gpt_out_fwd(gpt, x, x_img, x_lid):
    """ln_f -> raw view (quirk Q1) -> bilinear up-sample -> residual add, for both branches."""
    cfg, B, C, n_img, n_lid, T = _gpt_dims(gpt, x_img.shape, x_lid.shape)
    _, Hi, Wi, _ = x_img.shape
    _, Hl, Wl, _ = x_lid.shape
    xf, mf, rf = ops.layernorm_fwd(x, gpt.ln_f.weight, gpt.ln_f.bias, gpt.ln_f.eps)
    # Q1: token memory (hw, C) of each sample is re-read as (C, h, w); strides relative to the slice start
    out_img = ops.bilinear_fwd(xf, B, C, cfg.ih, cfg.iw, Hi, Wi, add=x_img, in_strides=(T * C, cfg.ih * cfg.iw, cfg.iw, 1))
    out_lid = ops.bilinear_fwd(xf[n_img:], B, C, cfg.lh, cfg.lw, Hl, Wl, add=x_lid, in_strides=(T * C, cfg.lh * cfg.lw, cfg.lw, 1))
    return out_img, out_lid, mf, rf


def _gpt_out_bwd(gpt, d_img, d_lid, x_last, mf, rf, s_img, s_lid):
    """(d_img, d_lid contiguous) -> gradient of the token matrix entering ln_f (a fresh tensor the Block backwards update in place)."""
    cfg, B, C, n_img, n_lid, T = _gpt_dims(gpt, s_img, s_lid)
    _, Hi, Wi, _ = s_img
    _, Hl, Wl, _ = s_lid
    dxf = torch.empty(B * T, C, dtype=torch.float32, device=d_img.device)
    ops.bilinear_bwd(d_img, B, C, cfg.ih, cfg.iw, Hi, Wi, out=dxf, in_strides=(T * C, cfg.ih * cfg.iw, cfg.iw, 1), in_nhwc=False)
    ops.bilinear_bwd(d_lid, B, C, cfg.lh, cfg.lw, Hl, Wl, out=dxf[n_img:], in_strides=(T * C, cfg.lh * cfg.lw, cfg.lw, 1), in_nhwc=False)
    return ops.layernorm_bwd(dxf, x_last, gpt.ln_f.weight, mf, rf, gbuf(gpt.ln_f.weight), gbuf(gpt.ln_f.bias))


@routes_param_grads
class GPTStageFn(torch.autograd.Function):
    """One fusion stage (transfuser.py:150-157 + GPT.forward :333-366): adaptive pools -> tokens + pos_emb
    -> n_layer Blocks -> ln_f -> raw view (quirk Q1) -> bilinear up-sample -> residual add, for both branches."""

    @staticmethod
    def forward(ctx, x_img, x_lid, gpt, velocity, *params):
        B, T = x_img.shape[0], gpt.geom.ih * gpt.geom.iw + gpt.geom.lh * gpt.geom.lw
        drop = gpt.training and gpt.pdrop_any
        x = _gpt_embed_fwd(gpt, x_img, x_lid, velocity)
        saved = []
        for li in range(len(gpt.blocks)):
            x, sv = _gpt_block_fwd(gpt, li, x, B, T, drop)
            saved.append(sv)
        out_img, out_lid, mf, rf = _gpt_out_fwd(gpt, x, x_img, x_lid)
        ctx.saved = (gpt, x_img.shape, x_lid.shape, saved, x, mf, rf, drop, velocity)
        return out_img, out_lid

    @staticmethod
    def backward(ctx, d_img, d_lid):
        gpt, s_img, s_lid, saved, x_last, mf, rf, drop, velocity = ctx.saved
        B, T = s_img[0], gpt.geom.ih * gpt.geom.iw + gpt.geom.lh * gpt.geom.lw
        d_img, d_lid = d_img.contiguous(), d_lid.contiguous()
        dx = _gpt_out_bwd(gpt, d_img, d_lid, x_last, mf, rf, s_img, s_lid)
        for li in range(len(gpt.blocks) - 1, -1, -1):
            _gpt_block_bwd(gpt, li, saved[li], dx, B, T, drop)
        dx_img, dx_lid = _gpt_embed_bwd(gpt, dx, s_img, s_lid, velocity, drop, add_img=d_img, add_lid=d_lid)
        ctx.saved = None
        return (dx_img, dx_lid, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


# ---- the same stage as separate autograd nodes (embed | Block x n | output): used for a stage that train.Engine cuts BETWEEN Blocks
@routes_param_grads
class GPTEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_img, x_lid, gpt, velocity, *params):
        ctx.saved = (gpt, x_img.shape, x_lid.shape, velocity, gpt.training and gpt.pdrop_any)
        return _gpt_embed_fwd(gpt, x_img, x_lid, velocity)

    @staticmethod
    def backward(ctx, dx):
        gpt, s_img, s_lid, velocity, drop = ctx.saved
        dx_img, dx_lid = _gpt_embed_bwd(gpt, dx.contiguous(), s_img, s_lid, velocity, drop)
        return (dx_img, dx_lid, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


@routes_param_grads
class GPTBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gpt, li, B, T, *params):
        drop = gpt.training and gpt.pdrop_any
        x_out, sv = _gpt_block_fwd(gpt, li, x, B, T, drop)
        ctx.saved = (gpt, li, sv, B, T, drop)
        return x_out

    @staticmethod
    def backward(ctx, dx):
        if ctx.saved is None:
            raise RuntimeError("GPTBlockFn: backward through the graph a second time (the saved activations were freed after the first backward)")
        gpt, li, sv, B, T, drop = ctx.saved
        ctx.saved = None
        # the incoming gradient is owned by this chain (produced by the next Block's / the output node's backward, one consumer): updated in place
        dx = _gpt_block_bwd(gpt, li, sv, dx.contiguous(), B, T, drop)
        return (dx, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


@routes_param_grads
class GPTOutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_img, x_lid, gpt, *params):
        out_img, out_lid, mf, rf = _gpt_out_fwd(gpt, x, x_img, x_lid)
        ctx.saved = (gpt, x, mf, rf, x_img.shape, x_lid.shape)
        return out_img, out_lid

    @staticmethod
    def backward(ctx, d_img, d_lid):
        gpt, x_last, mf, rf, s_img, s_lid = ctx.saved
        d_img, d_lid = d_img.contiguous(), d_lid.contiguous()
        dx = _gpt_out_bwd(gpt, d_img, d_lid, x_last, mf, rf, s_img, s_lid)
        return (dx, d_img, d_lid, None) + (None,) * (len(ctx.needs_input_grad) - 4)     # residual adds: the maps' gradients pass through


def gpt_stage(gpt, x_img, x_lid, velocity, cut=None):
    """Run one fusion stage.  ``cut(j, tensors, leaves=None) -> tensors`` is the backbone's hook for a backward cut in front of Block j
    (j = 0: between the embedding and Block 0; ``cut(j, None)`` only asks whether j is a cut); None, a stage without inner cuts or a
    forward that records no graph runs as the single node GPTStageFn.

    The residual adds at the end of the stage (out = up(tokens) + x) connect the stage's output straight to its input maps, so cutting
    the token chain alone would not sever the autograd graph: the maps reach GPTOutFn through detached leaves, and the FIRST inner cut
    (the last one the backward passes) hands their gradients back together with the token gradient - the trunks before the stage are
    differentiated once, with the sum."""
    inner = cut is not None and torch.is_grad_enabled() and x_img.requires_grad and any(cut(j, None) for j in range(len(gpt.blocks)))
    if not inner:
        return GPTStageFn.apply(x_img, x_lid, gpt, velocity, *gpt.parameters())
    B, T = x_img.shape[0], gpt.geom.ih * gpt.geom.iw + gpt.geom.lh * gpt.geom.lw
    emb = [gpt.pos_emb] + (list(gpt.vel_emb.parameters()) if gpt.use_velocity else [])
    first = min(j for j in range(len(gpt.blocks)) if cut(j, None))
    r_img, r_lid = x_img.detach().requires_grad_(True), x_lid.detach().requires_grad_(True)
    x = GPTEmbedFn.apply(x_img, x_lid, gpt, velocity, *emb)
    for j, blk in enumerate(gpt.blocks):
        if j == first:
            x = cut(j, (x, x_img, x_lid), (None, r_img, r_lid))[0]
        elif cut(j, None):
            (x,) = cut(j, (x,))
        x = GPTBlockFn.apply(x, gpt, j, B, T, *blk.parameters())
    return GPTOutFn.apply(x, r_img, r_lid, gpt, *gpt.ln_f.parameters())


# ============================================================================================ geometric-fusion stage (C4)
def _mlp_fwd(h, seq):
    """3 x [Linear + ReLU] (geometric_fusion.py:56-64): returns the activation list [h0, h1, h2, h3]."""
    acts = [h]
    for lin in (seq[0], seq[2], seq[4]):
        acts.append(ops.linear_fwd(acts[-1], lin.weight, lin.bias, relu=True))
    return acts


def _mlp_bwd(dh, acts, seq):
    for i, lin in reversed(list(enumerate((seq[0], seq[2], seq[4])))):
        g = ops.relu_mask(dh, acts[i + 1])
        ops.linear_wgrad(g, acts[i], gbuf(lin.weight))
        bias_grad(g, lin.bias)
        dh = ops.linear_dgrad(g, lin.weight)
    return dh


@routes_param_grads
class GeoStageFn(torch.autograd.Function):
    """One geometric-fusion stage for both branches (geometric_fusion.py:125-166 and the three copies below it):
    1x1 conv C->E + adaptive pool, gather the 5 correspondences of every cell from the OTHER branch (kernel G1), sum, 3-layer
    MLP, bilinear up-sample by the stage's fixed factor, 1x1 conv E->C, residual add (+ velocity embedding).

    Both 1x1 convs are evaluated at the pooled resolution: average pooling and bilinear interpolation are affine maps whose
    weights sum to one, so conv(pool(x)) == pool(conv(x)) and conv(up(z)) == up(conv(z)) in exact arithmetic (differences are
    fp32 round-off); this removes ~95 % of the stage's FLOPs.  Quirk Q4: stage 4's image side gathers from stage 3's pooled
    LiDAR embedding (``prev_lid_e``); ``lidar_conv4`` therefore never receives a gradient."""

    @staticmethod
    def forward(ctx, x_img, x_lid, prev_lid_e, st, velocity, bev_idx, img_idx, *params):
        ctx.set_materialize_grads(False)
        g = st.geom
        B, Hi, Wi, C = x_img.shape
        _, Hl, Wl, _ = x_lid.shape
        assert (Hi, Wi, Hl, Wl) == (g.ih * st.scale, g.iw * st.scale, g.lh * st.scale, g.lw * st.scale), \
            "geometric fusion: feature maps %s / %s do not match the fixed scale factor %d (reference shape error at geometric_fusion.py:154)" % (
                (Hi, Wi), (Hl, Wl), st.scale)
        E = st.image_conv.weight.shape[0]
        n_img, n_lid = g.ih * g.iw, g.lh * g.lw
        dev = x_img.device
        pi = ops.pool_tokens_fwd(x_img, g.ih, g.iw, None, torch.empty(B, n_img, C, dtype=torch.float32, device=dev), 0)
        img_e = ops.linear_fwd(pi.view(-1, C), w2d(st.image_conv.weight), st.image_conv.bias)
        pl = lid_e = None
        if not st.use_prev:
            pl = ops.pool_tokens_fwd(x_lid, g.lh, g.lw, None, torch.empty(B, n_lid, C, dtype=torch.float32, device=dev), 0)
            lid_e = ops.linear_fwd(pl.view(-1, C), w2d(st.lidar_conv.weight), st.lidar_conv.bias)
        src = prev_lid_e if st.use_prev else lid_e
        vel_l = vel_i = vrep_l = vrep_i = None
        if st.vel_emb is not None:   # vel_emb(velocity)[:, :, None, None] added to both maps == added to the low-res increment
            vrep_l = velocity.reshape(B, 1).repeat_interleave(n_lid, 0).contiguous()
            vrep_i = velocity.reshape(B, 1).repeat_interleave(n_img, 0).contiguous()
            vel_l = ops.linear_fwd(vrep_l, st.vel_emb.weight, st.vel_emb.bias)
            vel_i = ops.linear_fwd(vrep_i, st.vel_emb.weight, st.vel_emb.bias)
        # image -> BEV
        gb = ops.gather_sum_fwd(img_e.view(B, n_img, E), bev_idx, g.ih, g.iw)
        hb = _mlp_fwd(gb.view(-1, E), st.image_projection)
        inc_l = ops.linear_fwd(hb[-1], w2d(st.lidar_deconv.weight), st.lidar_deconv.bias, res=vel_l)
        out_lid = ops.bilinear_fwd(inc_l, B, C, g.lh, g.lw, Hl, Wl, add=x_lid)
        # BEV -> image
        gi = ops.gather_sum_fwd(src.reshape(B, n_lid, E), img_idx, g.lh, g.lw)
        hi = _mlp_fwd(gi.view(-1, E), st.lidar_projection)
        inc_i = ops.linear_fwd(hi[-1], w2d(st.image_deconv.weight), st.image_deconv.bias, res=vel_i)
        out_img = ops.bilinear_fwd(inc_i, B, C, g.ih, g.iw, Hi, Wi, add=x_img)
        ctx.saved = (st, x_img.shape, x_lid.shape, pi, pl, hb, hi, bev_idx, img_idx, vrep_l, vrep_i)
        if lid_e is not None:
            lid_e = lid_e.view(B, n_lid, E)
        return out_img, out_lid, lid_e

    @staticmethod
    def backward(ctx, d_img, d_lid, d_lid_e_out):
        st, s_img, s_lid, pi, pl, hb, hi, bev_idx, img_idx, vrep_l, vrep_i = ctx.saved
        g = st.geom
        B, Hi, Wi, C = s_img
        _, Hl, Wl, _ = s_lid
        E = st.image_conv.weight.shape[0]
        n_img, n_lid = g.ih * g.iw, g.lh * g.lw
        d_img, d_lid = d_img.contiguous(), d_lid.contiguous()

        def side(d_map, oh, ow, Ho, Wo, deconv, acts, proj, vrep):
            dinc = ops.bilinear_bwd(d_map, B, C, oh, ow, Ho, Wo).view(-1, C)
            ops.linear_wgrad(dinc, acts[-1], w2d(gbuf(deconv.weight)))
            bias_grad(dinc, deconv.bias)
            if vrep is not None:
                ops.linear_wgrad(dinc, vrep, gbuf(st.vel_emb.weight))
                bias_grad(dinc, st.vel_emb.bias)
            return _mlp_bwd(ops.linear_dgrad(dinc, w2d(deconv.weight)), acts, proj)

        dgb = side(d_lid, g.lh, g.lw, Hl, Wl, st.lidar_deconv, hb, st.image_projection, vrep_l)
        d_img_e = ops.gather_sum_bwd(dgb.view(B, n_lid, E), bev_idx, g.ih, g.iw).view(-1, E)
        dgi = side(d_img, g.ih, g.iw, Hi, Wi, st.image_deconv, hi, st.lidar_projection, vrep_i)
        d_src = ops.gather_sum_bwd(dgi.view(B, n_img, E), img_idx, g.lh, g.lw)
        d_prev = None
        if st.use_prev:
            d_prev, d_lid_e = d_src, d_lid_e_out
        else:
            d_lid_e = d_src if d_lid_e_out is None else ops.axpby(d_src, d_lid_e_out.contiguous().view_as(d_src), 1.0, 1.0, out=d_src)
        ops.linear_wgrad(d_img_e, pi.view(-1, C), w2d(gbuf(st.image_conv.weight)))
        bias_grad(d_img_e, st.image_conv.bias)
        dpi = ops.linear_dgrad(d_img_e, w2d(st.image_conv.weight))
        dx_img = ops.pool_tokens_bwd(dpi.view(B, n_img, C), s_img, g.ih, g.iw, 0, add=d_img)
        dx_lid = d_lid
        if d_lid_e is not None and pl is not None:
            d2 = d_lid_e.contiguous().view(-1, E)
            ops.linear_wgrad(d2, pl.view(-1, C), w2d(gbuf(st.lidar_conv.weight)))
            bias_grad(d2, st.lidar_conv.bias)
            dpl = ops.linear_dgrad(d2, w2d(st.lidar_conv.weight))
            dx_lid = ops.pool_tokens_bwd(dpl.view(B, n_lid, C), s_lid, g.lh, g.lw, 0, add=d_lid)
        ctx.saved = None
        return (dx_img, dx_lid, d_prev, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 7)


# ============================================================================================ generic conv / resample
@routes_param_grads
class ConvFn(torch.autograd.Function):
    """conv (1x1 or 3x3, stride 1, bias) (+ReLU) on NHWC: decoders, heads, FPN, channel reducers.
    ``link``: 0 plain; 1 = this (ReLU) layer's ONLY consumer is a ConvFn with link 2, which hands back an already ReLU-masked gradient;
    2 = the input is the output of a link-1 layer: the input gradient is masked with it (fused into the dgrad epilogue where the kernel can)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, link=0):
        B, H, W, Cin = x.shape
        if w.shape[2] == 1:
            y = ops.linear_fwd(x.view(-1, Cin), w2d(w), b, relu=relu).view(B, H, W, w.shape[0])
        else:
            y = ops.conv_fwd(x, w, b, 1, 1, 1, relu)
        ctx.saved = (x, w, b, relu, y, link)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, relu, y, link = ctx.saved
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        dy = dy.contiguous()
        g = ops.relu_mask(dy, y) if (relu and link != 1) else dy
        fused_bias = b is not None and w.shape[2] == 3 and ops.conv_wgrad_takes_bias(x.shape, Cout, 3)
        if b is not None and not fused_bias:
            bias_grad(g.view(-1, Cout), b)
        dx = None
        if w.shape[2] == 1:
            ops.linear_wgrad(g.view(-1, Cout), x.view(-1, Cin), w2d(gbuf(w)))
            if ctx.needs_input_grad[0]:
                dx = ops.linear_dgrad(g.view(-1, Cout), w2d(w), mask=x.view(-1, Cin) if link == 2 else None).view(B, H, W, Cin)
        else:
            ops.conv_wgrad(g, x, gbuf(w), 1, 1, 1, dbias=gbuf(b) if fused_bias else None)
            if ctx.needs_input_grad[0]:
                dx = ops.conv_dgrad(g, w, x.shape, 1, 1, 1, mask=x if link == 2 else None)
        ctx.saved = None
        return dx, None, None, None, None


class UpsampleFn(torch.autograd.Function):
    """bilinear NHWC -> NHWC (nn.Upsample x2 / F.interpolate scale_factor / size, either align mode)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo, align_corners):
        B, H, W, C = x.shape
        ctx.meta = (B, C, H, W, Ho, Wo, align_corners)
        return ops.bilinear_fwd(x, B, C, H, W, Ho, Wo, align_corners=align_corners)

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, Ho, Wo, align = ctx.meta
        return ops.bilinear_bwd(dy.contiguous(), B, C, H, W, Ho, Wo, align_corners=align), None, None, None


class GlobalPoolAddFn(torch.autograd.Function):
    """fused_features = flatten(gap(img)) + flatten(gap(lidar)) (transfuser.py:203-208)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.shapes = (a.shape, b.shape)
        pa = ops.colsum(a, a.shape[0], a.shape[1] * a.shape[2], a.shape[3], 1.0 / (a.shape[1] * a.shape[2]))
        return ops.colsum(b, b.shape[0], b.shape[1] * b.shape[2], b.shape[3], 1.0 / (b.shape[1] * b.shape[2]), out=pa, accumulate=True)

    @staticmethod
    def backward(ctx, d):
        sa, sb = ctx.shapes
        d = d.contiguous()
        return ops.se_scale_bwd_x(None, None, d, sa), ops.se_scale_bwd_x(None, None, d, sb)


# ============================================================================================ losses
class CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy over NHWC logits (model.py:763,783)."""

    @staticmethod
    def forward(ctx, logits, target, class_w):
        loss, dl, inv = ops.ce_fwd(logits, target, class_w)
        ctx.saved = (dl, inv)
        return loss

    @staticmethod
    def backward(ctx, g):
        dl, inv = ctx.saved
        ctx.saved = None
        return ops.scale_dev_(dl, g.contiguous(), inv, 1.0), None, None


class L1Fn(torch.autograd.Function):
    """mean |f(pred) - target| (model.py:765; :784 with the DepthDecoder sigmoid folded in)."""

    @staticmethod
    def forward(ctx, pred, target, use_sigmoid):
        loss, dp = ops.l1_fwd(pred, target, use_sigmoid)
        ctx.saved = dp
        return loss

    @staticmethod
    def backward(ctx, g):
        dp = ctx.saved
        ctx.saved = None
        return ops.scale_dev_(dp, g.contiguous(), None, 1.0), None, None


class CenterNetLossFn(torch.autograd.Function):
    """get_targets + the 7 CenterNet losses (model.py:150-248,285-374) -> 7 scalars (views of one (7,) tensor: indexing a (7,) output
    instead cost a select_backward - zeros + copy - and an accumulation launch per loss in the backward)."""

    @staticmethod
    def forward(ctx, pred, label, nbins, ratio_w, ratio_h):
        B, fh, fw, _ = pred.shape
        tgtf, tgti, cnt = ops.centernet_targets(label, fh, fw, ratio_w, ratio_h, nbins)
        ctx.saved = (pred, tgtf, tgti, cnt, nbins)
        out = ops.centernet_loss_fwd(pred, tgtf, tgti, cnt, nbins)
        return tuple(out[i] for i in range(out.shape[0]))

    @staticmethod
    def backward(ctx, *gs):
        pred, tgtf, tgti, cnt, nbins = ctx.saved
        ctx.saved = None
        g0 = gs[0]
        if all(t.is_contiguous() and t.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr() and t.storage_offset() == g0.storage_offset() + i
               for i, t in enumerate(gs)):       # WeightedSumFn hands back adjacent elements of one tensor: no gather
            g = torch.as_strided(g0, (len(gs),), (1,), g0.storage_offset())
        else:
            g = torch.stack([t.reshape(()) for t in gs])
        return ops.centernet_loss_bwd(pred, tgtf, tgti, cnt, g.contiguous(), nbins), None, None, None, None


# ============================================================================================ waypoint head
@routes_param_grads
class WaypointFn(torch.autograd.Function):
    """join MLP 512->256->128->64 (+ReLU) on the MFMA engine, then the fused auto-regressive GRU decoder
    (model.py:592-605,611-646) -> pred_wp (B, pred_len, 2)."""

    @staticmethod
    def forward(ctx, fused, target_point, head, *params):
        j0, j1, j2 = head.join[0], head.join[2], head.join[4]
        a0 = ops.linear_fwd(fused, j0.weight, j0.bias, relu=True)
        a1 = ops.linear_fwd(a0, j1.weight, j1.bias, relu=True)
        z = ops.linear_fwd(a1, j2.weight, j2.bias, relu=True)
        wp, cache = ops.gru_waypoints_fwd(z, target_point.contiguous(), head.decoder, head.output, head.pred_len, float(head.config.lidar_pos[0]))
        ctx.saved = (fused, head, a0, a1, z, cache)
        return wp

    @staticmethod
    def backward(ctx, dwp):
        fused, head, a0, a1, z, cache = ctx.saved
        gru, outl = head.decoder, head.output
        j0, j1, j2 = head.join[0], head.join[2], head.join[4]
        grads = (gbuf(gru.weight_ih), gbuf(gru.weight_hh), gbuf(gru.bias_ih), gbuf(gru.bias_hh), gbuf(outl.weight), gbuf(outl.bias))
        dz = ops.gru_waypoints_bwd(dwp.contiguous(), cache, gru, outl, grads, z.shape[0], z.shape[1], head.pred_len)
        ops.relu_mask(dz, z, out=dz)
        ops.linear_wgrad(dz, a1, gbuf(j2.weight))
        bias_grad(dz, j2.bias)
        da1 = ops.linear_dgrad(dz, j2.weight)
        ops.relu_mask(da1, a1, out=da1)
        ops.linear_wgrad(da1, a0, gbuf(j1.weight))
        bias_grad(da1, j1.bias)
        da0 = ops.linear_dgrad(da1, j1.weight)
        ops.relu_mask(da0, a0, out=da0)
        ops.linear_wgrad(da0, fused, gbuf(j0.weight))
        bias_grad(da0, j0.bias)
        dfused = ops.linear_dgrad(da0, j0.weight)
        ctx.saved = None
        return (dfused, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)



class WeightedSumFn(torch.autograd.Function):
    """total = sum_i w_i * loss_i (train.py:307-311) over the detailed losses' 0-dim tensors: one launch forward, one backward, instead of the
    0-dim ATen multiplies / adds of the Python loop and their autograd nodes.  ``weights`` is a tuple of Python floats."""

    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.weights = tuple(float(w) for w in weights)
        ctx.dev = terms[0].device
        return ops.weighted_sum([t.detach() for t in terms], ctx.weights)

    @staticmethod
    def backward(ctx, dtotal):
        g = ops.weighted_sum_bwd(dtotal.contiguous() if dtotal is not None else None, ctx.weights, ctx.dev)
        return (None,) + tuple(g[i] for i in range(len(ctx.weights)))

