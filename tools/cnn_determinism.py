"""the CNN32 step is built from kernels without atomics: two runs from the same state must agree BITWISE.  Any mismatch is a
race.  usage: python tools/cnn_determinism.py <reps> graph|eager"""
import sys
import torch
sys.path.insert(0, ".")
import eoe_amd
from eoe_amd import parallel
from eoe_amd.models import CNN32

dev = torch.device("cuda")
nb = 128
gen = torch.Generator(device=dev); gen.manual_seed(1234)
imgs = torch.randn((2 * nb, 3, 32, 32), generator=gen, device=dev)
imgs[nb:] += 0.5 * torch.randn((1, 3, 32, 32), generator=torch.Generator(device=dev).manual_seed(7), device=dev)
lbls = torch.cat([torch.zeros(nb, dtype=torch.int64), torch.ones(nb, dtype=torch.int64)]).to(dev)
reps, graph = int(sys.argv[1]), sys.argv[2] == "graph"
ref = None
mism = 0
for rep in range(reps):
    torch.manual_seed(0)
    model = CNN32(bias=True).to(dev).train()
    opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
    arena = parallel.GradArena(model)
    if graph:
        gs = eoe_amd.GraphedStep(model, lambda f, y: eoe_amd.hsc_loss(f, y, 0, 1.0 / (2 * nb)), eoe_amd.hsc_score, imgs, lbls)
    losses = []
    for i in range(60):
        opt.zero_grad()
        if graph:
            loss, sc = gs(imgs, lbls)
        else:
            loss = eoe_amd.hsc_loss(model(imgs), lbls, 0, 1.0 / (2 * nb))
            loss.backward()
        opt.step()
        losses.append(loss.detach().clone())
    l = torch.stack(losses).cpu()
    if ref is None:
        ref = l
    elif not torch.equal(l, ref):
        d = (l != ref).nonzero().flatten()
        mism += 1
        print(f"rep {rep}: {len(d)} of 60 losses differ from rep 0, first at step {int(d[0])}: {l[d[0]].item()} vs {ref[d[0]].item()}", flush=True)
    del model, opt, arena
import hashlib
print(f"{'graph' if graph else 'eager'}: {mism}/{reps - 1} repetitions differ bitwise from the first; final loss {ref[-1].item():.6f}; "
      f"trajectory sha1 {hashlib.sha1(ref.numpy().tobytes()).hexdigest()[:12]}")
sys.exit(1 if mism else 0)
