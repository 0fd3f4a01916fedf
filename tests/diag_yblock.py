"""Intermediate-gradient probes inside the failing RegNetY block (diagnostic)."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import model_cases as mc
from transfuser_amd import regnet as PR
from oracle import regnet as OR
dev = "cuda"
torch.manual_seed(0)
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    d = (a - b).abs()
    return "maxerr %.2e rms_ref %.2e relmax %.2e  nbad(>1e-3*rms) %d/%d" % (d.max(), b.pow(2).mean().sqrt(), d.max() / b.pow(2).mean().sqrt(), int((d > 1e-3 * b.pow(2).mean().sqrt()).sum()), d.numel())
for (cin, cout, stride, H, W) in [(576, 1512, 2, 16, 16), (576, 1512, 2, 10, 44)]:
    pb = PR.Bottleneck(cin, cout, stride, 24, 0.25); mc.randomize(pb)
    for m in pb.modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size != (1, 1):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    ob = OR.Bottleneck(cin, cout, stride, 24, 0.25); ob.load_state_dict(pb.state_dict()); pb = pb.to(dev)
    cap = {}
    def hook(name):
        def f(mod, gin, gout):
            cap[name + ".gout"] = gout[0]
            if gin[0] is not None: cap[name + ".gin"] = gin[0]
        return f
    for name in ["conv1.conv", "conv1.bn", "conv2.conv", "conv2.bn", "se", "conv3.conv", "conv3.bn", "downsample.conv", "downsample.bn"]:
        mod = ob
        for part in name.split("."): mod = getattr(mod, part)
        mod.register_full_backward_hook(hook(name))
    x = torch.randn(2, cin, H, W, requires_grad=True); xh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    yo = ob(x); pb._dbg = {}; yp = pb(xh)
    dy = torch.randn_like(yo); yo.backward(dy); yp.backward(dy.permute(0, 2, 3, 1).contiguous().to(dev))
    d = pb._dbg
    nchw = lambda t: t.permute(0, 3, 1, 2)
    print("== block", (cin, cout, stride, H, W))
    print("dy3  (grad into conv3.bn input)   ", rel(nchw(d["dy3"]), cap["conv3.conv.gout"]))
    print("dz2s (grad out of conv3 wrt input) ", rel(nchw(d["dz2s"]), cap["se.gout"]))
    print("dz2  (grad into SE input)          ", rel(nchw(d["dz2"]), cap["conv2.bn.gout"]))
    print("dy2  (grad into conv2.bn input)    ", rel(nchw(d["dy2"]), cap["conv2.conv.gout"]))
    print("dz1  (conv2 dgrad)                 ", rel(nchw(d["dz1"]), cap["conv1.bn.gout"]))
    print("dy1                                ", rel(nchw(d["dy1"]), cap["conv1.conv.gout"]))
    print("dsc                                ", rel(nchw(d["dsc"]), cap["downsample.bn.gout"]))
    print("dx                                 ", rel(nchw(xh.grad), x.grad))
    e = (nchw(d["dz1"]).cpu() - cap["conv1.bn.gout"]).abs()
    bad = (e > 1e-3 * cap["conv1.bn.gout"].pow(2).mean().sqrt()).nonzero()
    print("dz1 bad idx sample (b,c,h,w):", bad[:12].tolist(), " unique h:", sorted(set(bad[:, 2].tolist()))[:20], " unique w:", sorted(set(bad[:, 3].tolist()))[:20], "unique c count", len(set(bad[:,1].tolist())))
