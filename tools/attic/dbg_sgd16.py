import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd
kw = dict(lr=1e-2, momentum=0.9, nesterov=True, weight_decay=1e-3)
for shape in ((70,), (70, 300), (64,), (72,), (4096,)):
    torch.manual_seed(3)
    w0 = (torch.randn(shape, device="cuda") * 0.05).half()
    p = torch.nn.Parameter(w0.float().clone()); p._eoe_fp16_weight = True
    rr = torch.nn.Parameter(w0.clone())
    o1, o2 = eoe_amd.FusedSGD([p], **kw), torch.optim.SGD([rr], **kw)
    for step in range(3):
        g = (torch.randn_like(p) * (0.5 + step)).half().float()
        p.grad, rr.grad = g.clone(), g.half()
        pb = p.detach().clone()
        o1.step(); o2.step()
        d = (p.detach() - rr.detach().float())
        bd = (o1.state[p]["momentum_buffer"] - o2.state[rr]["momentum_buffer"].float())
        print(shape, "step", step, "param mismatches", (d != 0).float().mean().item(), "buf mismatches", (bd != 0).float().mean().item())
        if (d != 0).any() and step == 0:
            for j in (d != 0).flatten().nonzero()[:3].flatten().tolist():
                print("   idx", j, "p0", pb.flatten()[j].item(), "g", g.flatten()[j].item(), "ours", p.flatten()[j].item(), "torch", rr.float().flatten()[j].item())
