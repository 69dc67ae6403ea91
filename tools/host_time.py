"""Host-side enqueue time of each phase of the ViT training step (no device synchronisation inside the loop): where the Python side
spends its share of the step.  python tools/host_time.py [steps]"""
import sys
import time

import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as bench.py and the tests: kernel arguments in device memory (read when libamdhip64 loads)
import torch

sys.path.insert(0, ".")
import eoe_amd                                         # noqa: E402
from eoe_amd import parallel                           # noqa: E402
from eoe_amd.models import ClipViTB32Custom            # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
eoe_amd.set_compute_dtype("fp16")
eoe_amd.set_grad_scale(eoe_amd.default_grad_scale())
torch.manual_seed(0)
model = ClipViTB32Custom().to(dev).train()
opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-3)
arena = parallel.GradArena(model, comm=None)
imgs = torch.randn(256, 3, 224, 224, device=dev)
lbls = torch.cat([torch.zeros(128, dtype=torch.int64), torch.ones(128, dtype=torch.int64)]).to(dev)
acc = {}


def lap(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t


for i in range(steps + 10):
    if i == 10:
        torch.cuda.synchronize()
        acc.clear()
        wall0 = time.perf_counter()
    t = time.perf_counter()
    opt.zero_grad()
    t = lap("zero_grad", t)
    feats = model(imgs)
    t = lap("forward", t)
    loss = eoe_amd.hsc_loss(feats, lbls, 0, 1.0 / 256)
    t = lap("loss", t)
    loss.backward()
    t = lap("backward", t)
    arena.finish()
    t = lap("arena.finish", t)
    opt.step()
    t = lap("opt.step", t)
    s = eoe_amd.hsc_score(feats)
    t = lap("score", t)
torch.cuda.synchronize()
wall = (time.perf_counter() - wall0) / steps * 1e3
print(f"wall {wall:.3f} ms per step; host enqueue time per step:")
for k, v in acc.items():
    print(f"  {k:14s} {v / steps * 1e3:7.3f} ms")
print(f"  {'sum':14s} {sum(acc.values()) / steps * 1e3:7.3f} ms")
