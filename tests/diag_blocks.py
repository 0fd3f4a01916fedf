"""Block-level parity at the REAL shapes of regnety_032 / GPT1-4 on the GPU (diagnostic, not a pytest)."""
import sys, os, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import model_cases as mc
from transfuser_amd import regnet as PR, transfuser as PT
from oracle import regnet as OR, transfuser_cpu as OT

dev = "cuda"
torch.manual_seed(0)

def cmp(prod, ref, tag):
    rp = dict(ref.named_parameters()); worst = []
    for n, p in prod.named_parameters():
        g = rp[n].grad; e = (p.grad.cpu() - g).abs().max().item(); worst.append((e / max(g.abs().max().item(), 1e-4), n))
    worst.sort(reverse=True); print("   params", tag, ["%.1e %s" % w for w in worst[:3]], flush=True)

def cl(mod):
    for m in mod.modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size != (1, 1):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)

B = 2
for (cin, cout, stride, H, W) in [(32, 72, 2, 80, 352), (72, 72, 1, 40, 176), (72, 216, 2, 40, 176), (216, 216, 1, 20, 88), (216, 576, 2, 20, 88),
                                  (576, 576, 1, 10, 44), (576, 1512, 2, 10, 44), (576, 576, 1, 16, 16), (576, 1512, 2, 16, 16)]:
    pb = PR.Bottleneck(cin, cout, stride, 24, 0.25); mc.randomize(pb); cl(pb)
    ob = OR.Bottleneck(cin, cout, stride, 24, 0.25); ob.load_state_dict(pb.state_dict()); pb = pb.to(dev)
    x = torch.randn(B, cin, H, W, requires_grad=True); xh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    yo = ob(x); yp = pb(xh)
    dy = torch.randn_like(yo); yo.backward(dy); yp.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev))
    print("block", (cin, cout, stride, H, W), "fwd %.1e" % (yp.detach().cpu().permute(0, 3, 1, 2) - yo).abs().max().item(),
          "dx %.1e / %.1e" % ((xh.grad.cpu().permute(0, 3, 1, 2) - x.grad).abs().max().item(), x.grad.abs().max().item()), flush=True)
    cmp(pb, ob, "")

cfg = mc.full_config()
for (C, Hi, Wi, Hl, Wl) in [(72, 40, 176, 64, 64), (216, 20, 88, 32, 32), (576, 10, 44, 16, 16), (1512, 5, 22, 8, 8), (1512, 8, 22, 8, 8)]:
    pg = PT.GPT(C, 4, 4, 4, 5, 22, 8, 8, 1, 0., 0., 0., cfg, use_velocity=False); mc.randomize(pg)
    og = OT.GPT(C, cfg, False); og.load_state_dict(pg.state_dict()); pg = pg.to(dev)
    pg.seed = torch.zeros(1, dtype=torch.int32, device=dev)
    xi = torch.randn(B, C, Hi, Wi, requires_grad=True); xl = torch.randn(B, C, Hl, Wl, requires_grad=True)
    fi, fl = og(F.adaptive_avg_pool2d(xi, (5, 22)), F.adaptive_avg_pool2d(xl, (8, 8)), None)
    yi = xi + F.interpolate(fi, size=(Hi, Wi), mode='bilinear', align_corners=False); yl = xl + F.interpolate(fl, size=(Hl, Wl), mode='bilinear', align_corners=False)
    xih = xi.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True); xlh = xl.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    pi, pl = pg(xih, xlh, None)
    di, dl = torch.randn_like(yi), torch.randn_like(yl)
    torch.autograd.backward([yi, yl], [di, dl]); torch.autograd.backward([pi, pl], [di.permute(0, 2, 3, 1).contiguous().to(dev), dl.permute(0, 2, 3, 1).contiguous().to(dev)])
    print("gpt", (C, Hi, Wi), "fwd %.1e %.1e" % ((pi.detach().cpu().permute(0, 3, 1, 2) - yi).abs().max().item(), (pl.detach().cpu().permute(0, 3, 1, 2) - yl).abs().max().item()),
          "dx %.1e %.1e / %.1e" % ((xih.grad.cpu().permute(0, 3, 1, 2) - xi.grad).abs().max().item(), (xlh.grad.cpu().permute(0, 3, 1, 2) - xl.grad).abs().max().item(), xi.grad.abs().max().item()), flush=True)
    cmp(pg, og, "")
print("done", flush=True)
