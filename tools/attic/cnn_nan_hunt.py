"""hunt the intermittent NaN of the CNN32 bench configuration: bench-like loop (grad arena, graph replay), many fresh runs in one
process; on the first non-finite loss report the step and which tensors are non-finite"""
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
reps, graph, arena_on = int(sys.argv[1]), sys.argv[2] == "graph", sys.argv[3] == "arena"
bad = 0
poison = len(sys.argv) > 4 and sys.argv[4] == "poison"


def poison_allocator():
    """fill the caching allocator's free blocks with NaN bit patterns: a kernel that reads memory nobody wrote then sees NaN"""
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    big = [torch.full((64 << 20,), float("nan"), device=dev) for _ in range(8)]          # 8 x 256 MB (large pool)
    mid = [torch.full((1 << 18,), float("nan"), device=dev) for _ in range(256)]         # 256 x 1 MB
    small = [torch.full((n,), float("nan"), device=dev) for n in (64, 256, 1024, 4096, 16384, 65536) for _ in range(64)]
    torch.cuda.synchronize()
    del big, mid, small


for rep in range(reps):
    if poison:
        poison_allocator()
    torch.manual_seed(0)
    model = CNN32(bias=True).to(dev).train()
    opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
    arena = parallel.GradArena(model) if arena_on else None
    if graph:
        gs = eoe_amd.GraphedStep(model, lambda f, y: eoe_amd.hsc_loss(f, y, 0, 1.0 / (2 * nb)), eoe_amd.hsc_score, imgs, lbls)
    for i in range(60):
        opt.zero_grad()
        if graph:
            loss, sc = gs(imgs, lbls)
        else:
            f = model(imgs)
            loss = eoe_amd.hsc_loss(f, lbls, 0, 1.0 / (2 * nb))
            loss.backward()
        gbad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        lossv = loss.item()
        if gbad or lossv != lossv:
            pbad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
            bbad = [n for n, b in model.named_buffers() if b.dtype.is_floating_point and not torch.isfinite(b).all()]
            print(f"rep {rep} step {i}: loss {lossv}; non-finite grads {gbad}; params {pbad}; buffers {bbad}", flush=True)
            bad += 1
            break
        opt.step()
    del model, opt, arena
    if graph:
        del gs
print(f"graph={graph} arena={arena_on}: {bad}/{reps} runs hit a non-finite value", flush=True)
