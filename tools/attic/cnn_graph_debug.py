"""one bench-like CNN32 run with graph replay and per-step checks: on the first non-finite loss say whether the inputs /
parameters going INTO that replay were finite, which outputs are not, and what an eager step on the same state gives"""
import copy, sys
import torch
sys.path.insert(0, ".")
import eoe_amd
from eoe_amd import parallel
from eoe_amd.models import CNN32

dev = torch.device("cuda")
nb = 128
torch.manual_seed(0)
model = CNN32(bias=True).to(dev).train()
opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
arena = parallel.GradArena(model)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
imgs = torch.randn((2 * nb, 3, 32, 32), generator=gen, device=dev)
imgs[nb:] += 0.5 * torch.randn((1, 3, 32, 32), generator=torch.Generator(device=dev).manual_seed(7), device=dev)
lbls = torch.cat([torch.zeros(nb, dtype=torch.int64), torch.ones(nb, dtype=torch.int64)]).to(dev)
gs = eoe_amd.GraphedStep(model, lambda f, y: eoe_amd.hsc_loss(f, y, 0, 1.0 / (2 * nb)), eoe_amd.hsc_score, imgs, lbls)
check_every = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sleep_cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # GPU-side delay per step: lets the host run far ahead
prev = None
for i in range(60):
    opt.zero_grad()
    if i % check_every == 0:
        prev = {k: v.detach().clone() for k, v in model.state_dict().items()}      # state going into this replay
    if sleep_cycles:
        torch.cuda._sleep(sleep_cycles)
    loss, sc = gs(imgs, lbls)
    if i % check_every == 0:
        lv = loss.item()
        if lv != lv:
            pin = [k for k, v in prev.items() if v.dtype.is_floating_point and not torch.isfinite(v).all()]
            gbad = [n for n, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
            print(f"step {i}: loss NaN; non-finite state going in: {pin}; static imgs finite {torch.isfinite(gs.imgs).all().item()}; "
                  f"non-finite grads {gbad}; scores finite {torch.isfinite(sc).all().item()}")
            # the same step eagerly on the saved state
            m2 = CNN32(bias=True).to(dev).train()
            m2.load_state_dict(prev)
            l2 = eoe_amd.hsc_loss(m2(imgs), lbls, 0, 1.0 / (2 * nb))
            l2.backward()
            g2 = [n for n, p in m2.named_parameters() if not torch.isfinite(p.grad).all()]
            print(f"   eager on the same state: loss {l2.item()}, non-finite grads {g2}")
            # replay again on the same (restored) state
            model.load_state_dict(prev)
            opt.zero_grad()
            l3, _ = gs(imgs, lbls)
            print(f"   graph replay again on the restored state: loss {l3.item()}")
            sys.exit(1)
    opt.step()
lv = loss.item()
if lv != lv:
    print("final loss NaN")
    sys.exit(2)
print("ok")
