"""how close does the CNN32 + HSC synthetic run come to fp16's range? per step: min OE distance, max |df| at the loss head"""
import sys
import torch
sys.path.insert(0, ".")
import eoe_amd
from eoe_amd.models import CNN32
dev = torch.device("cuda")
nb = 128
gen = torch.Generator(device=dev); gen.manual_seed(1234)
imgs = torch.randn((2 * nb, 3, 32, 32), generator=gen, device=dev)
imgs[nb:] += 0.5 * torch.randn((1, 3, 32, 32), generator=torch.Generator(device=dev).manual_seed(7), device=dev)
lbls = torch.cat([torch.zeros(nb, dtype=torch.int64), torch.ones(nb, dtype=torch.int64)]).to(dev)
worst_d, worst_g = 1e9, 0.0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    torch.manual_seed(0)
    model = CNN32(bias=True).to(dev).train()
    opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
    for i in range(70):
        opt.zero_grad()
        f = model(imgs)
        f.retain_grad()
        loss = eoe_amd.hsc_loss(f, lbls, 0, 1.0 / (2 * nb))
        loss.backward()
        d = (f.detach()[nb:].norm(dim=1) ** 2 + 1).sqrt() - 1
        worst_d = min(worst_d, d.min().item())
        worst_g = max(worst_g, f.grad.abs().max().item())
        opt.step()
print(f"min OE distance over all steps {worst_d:.3e}; max |dloss/dfeature| {worst_g:.3e} (x 256 = per-sample {worst_g * 256:.3e}); fp16 max 65504")
