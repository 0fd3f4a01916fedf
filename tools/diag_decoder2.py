"""Per-operation isolation of the decoder backward at the full 256x704 size: every backward op of the product is fed the CPU oracle's EXACT
incoming gradient and saved activations, its outputs are compared with the oracle's (fp64 truth)."""
import sys, os, torch
import torch.nn.functional as F
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
l2 = lambda a, b: ((a.detach().cpu().double() - b.detach().double()).norm() / max(b.detach().double().norm().item(), 1e-30)).item()
nh = lambda t: t.permute(0, 2, 3, 1).contiguous().float().cuda()
cl = lambda w: w.float().contiguous(memory_format=torch.channels_last).cuda()
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (Cin, Cout, H, W) in [(32, 7, 256, 704), (32, 32, 256, 704), (64, 32, 64, 176), (64, 64, 64, 176), (256, 64, 8, 22)]:
    x = torch.randn(B, Cin, H, W, dtype=torch.float64).relu()            # a post-ReLU input (about half zeros), like the decoder's
    w = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64) * (1.0 / (3 * Cin ** 0.5))
    xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True)
    y = F.conv2d(xg, wg, None, 1, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    xh, wh, dyh = nh(x), cl(w), nh(dy)
    dx = ops.conv_dgrad(dyh, wh, xh.shape, 1, 1, 1)
    dxm = ops.conv_dgrad(dyh, wh, xh.shape, 1, 1, 1, mask=xh)
    dw = torch.zeros_like(wh); ops.conv_wgrad(dyh, xh, dw, 1, 1, 1)
    yh = ops.conv_fwd(xh, wh, None, 1, 1, 1, False)
    print("conv3x3 %3d -> %2d at %dx%dx%d: fwd %.1e dgrad %.1e dgrad+mask %.1e wgrad %.1e" % (Cin, Cout, B, H, W, l2(yh.permute(0, 3, 1, 2), y), l2(dx.permute(0, 3, 1, 2), xg.grad),
          l2(dxm.permute(0, 3, 1, 2), xg.grad * (x > 0)), l2(dw, wg.grad)), flush=True)
for (C, H, W, s) in [(32, 64, 176, 4), (64, 8, 22, 8)]:
    x = torch.randn(B, C, H, W, dtype=torch.float64, requires_grad=True)
    y = F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
    dy = torch.randn_like(y); y.backward(dy)
    yh = ops.bilinear_fwd(nh(x.detach()), B, C, H, W, H * s, W * s, align_corners=False)
    dxh = ops.bilinear_bwd(nh(dy), B, C, H, W, H * s, W * s, align_corners=False)
    print("bilinear x%d C=%d %dx%d: fwd %.1e bwd %.1e" % (s, C, H, W, l2(yh.permute(0, 3, 1, 2), y), l2(dxh.permute(0, 3, 1, 2), x.grad)), flush=True)
y = torch.randn(B * 256 * 704, 7, dtype=torch.float64); 
print("colsum bias grad 7 x %d rows: %.1e" % (y.shape[0], l2(ops.colsum(y.float().cuda(), 1, y.shape[0], 7)[0], y.sum(0))))
y = torch.randn(B * 256 * 704, 32, dtype=torch.float64)
print("colsum bias grad 32 x %d rows: %.1e" % (y.shape[0], l2(ops.colsum(y.float().cuda(), 1, y.shape[0], 32)[0], y.sum(0))))
