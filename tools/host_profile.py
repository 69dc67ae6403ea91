"""cProfile of the host side of the ViT training step (what tools/host_time.py sums up): python tools/host_profile.py"""
import cProfile
import pstats
import sys

import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as bench.py and the tests: kernel arguments in device memory (read when libamdhip64 loads)
import torch

sys.path.insert(0, ".")
import eoe_amd                                         # noqa: E402
from eoe_amd import parallel                           # noqa: E402
from eoe_amd.models import ClipViTB32Custom            # noqa: E402

dev = torch.device("cuda:0")
eoe_amd.set_compute_dtype("fp16")
eoe_amd.set_grad_scale(eoe_amd.default_grad_scale())
model = ClipViTB32Custom().to(dev).train()
opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-3)
arena = parallel.GradArena(model, comm=None)
imgs = torch.randn(256, 3, 224, 224, device=dev)
lbls = torch.cat([torch.zeros(128, dtype=torch.int64), torch.ones(128, dtype=torch.int64)]).to(dev)


def step():
    opt.zero_grad()
    feats = model(imgs)
    loss = eoe_amd.hsc_loss(feats, lbls, 0, 1.0 / 256)
    loss.backward()
    arena.finish()
    opt.step()
    return eoe_amd.hsc_score(feats)


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
