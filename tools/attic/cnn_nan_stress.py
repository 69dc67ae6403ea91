"""stress: many CNN32 training steps under flag combinations, counting non-finite losses (hunting an intermittent NaN)"""
import sys, itertools
import torch
sys.path.insert(0, ".")
import eoe_amd
import eoe_amd.ops as ops
from eoe_amd import _lib
from eoe_amd.models import CNN32

dev = torch.device("cuda")
nb = 128
gen = torch.Generator(device=dev); gen.manual_seed(1234)
imgs = torch.randn((2 * nb, 3, 32, 32), generator=gen, device=dev)
imgs[nb:] += 0.5 * torch.randn((1, 3, 32, 32), generator=torch.Generator(device=dev).manual_seed(7), device=dev)
lbls = torch.cat([torch.zeros(nb, dtype=torch.int64), torch.ones(nb, dtype=torch.int64)]).to(dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for graph, tnf, implicit in itertools.product((False, True), (0, 1), (True, False)):
    _lib.check(_lib.lib.eoe_set_option(b"tn_flags", tnf), "opt")
    ops.set_implicit_conv(implicit)
    bad = 0
    first_bad = None
    for rep in range(reps):
        torch.manual_seed(0)
        model = CNN32(bias=True).to(dev).train()
        opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
        if graph:
            gs = eoe_amd.GraphedStep(model, lambda f, y: eoe_amd.hsc_loss(f, y, 0, 1.0 / (2 * nb)), eoe_amd.hsc_score, imgs, lbls)
        losses = []
        for i in range(70):
            opt.zero_grad()
            if graph:
                loss, _ = gs(imgs, lbls)
            else:
                loss = eoe_amd.hsc_loss(model(imgs), lbls, 0, 1.0 / (2 * nb))
                loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        l = torch.stack(losses).cpu()
        if not torch.isfinite(l).all():
            bad += 1
            if first_bad is None:
                first_bad = int((~torch.isfinite(l)).nonzero()[0])
    print(f"graph={graph} tn_flags={tnf} implicit={implicit}: {bad}/{reps} runs with a non-finite loss (first at step {first_bad}); last loss {l[-1].item():.5f}", flush=True)
