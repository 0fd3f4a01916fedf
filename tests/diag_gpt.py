"""GPT C=1512 stage under TF_CHECK=1 after warming the allocator with other work (diagnostic)."""
import sys, os
os.environ["TF_CHECK"] = "1"
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import model_cases as mc
from transfuser_amd import transfuser as PT
from oracle import transfuser_cpu as OT
dev = "cuda"; torch.manual_seed(0)
cfg = mc.full_config(); B = 2
junk = [torch.randn(1 << 22, device=dev) * 1e3 for _ in range(64)]; del junk   # poison the caching allocator's free blocks
for (C, Hi, Wi, Hl, Wl) in [(576, 10, 44, 16, 16), (1512, 5, 22, 8, 8), (1512, 8, 22, 8, 8)]:
    pg = PT.GPT(C, 4, 4, 4, 5, 22, 8, 8, 1, 0., 0., 0., cfg, use_velocity=False); mc.randomize(pg)
    og = OT.GPT(C, cfg, False); og.load_state_dict(pg.state_dict()); pg = pg.to(dev)
    pg.seed = torch.zeros(1, dtype=torch.int32, device=dev)
    xi = torch.randn(B, C, Hi, Wi, requires_grad=True); xl = torch.randn(B, C, Hl, Wl, requires_grad=True)
    xih = xi.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True); xlh = xl.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    try:
        pi, pl = pg(xih, xlh, None)
        di, dl = torch.randn(B, Hi, Wi, C), torch.randn(B, Hl, Wl, C)
        torch.autograd.backward([pi, pl], [di.to(dev), dl.to(dev)])
        print("C=%d: all GEMMs verified OK" % C, flush=True)
    except Exception as e:
        print("C=%d: %s" % (C, str(e)[:1500]), flush=True)
