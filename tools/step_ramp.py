#!/usr/bin/env python
"""per-step wall time of the first 120 ViT training steps after an idle period (hipEvents around every step): how long does the
GPU take to reach its steady clock, and what do 5 warm-up + 20 timed steps measure?"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd
from eoe_amd.models import ClipViTB32Custom

eoe_amd.set_compute_dtype("fp16")
eoe_amd.set_grad_scale(256.0)
dev = torch.device("cuda")
torch.manual_seed(0)
model = ClipViTB32Custom(prediction_head=True, clf=False, freeze=False).to(dev).train()
opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-3)
imgs = torch.randn(256, 3, 224, 224, device=dev)
lbls = torch.cat([torch.zeros(128, dtype=torch.int64), torch.ones(128, dtype=torch.int64)]).to(dev)


def step():
    opt.zero_grad()
    loss = eoe_amd.hsc_loss(model(imgs), lbls, 0)
    loss.backward()
    opt.step()


for idle in (0.0, 3.0):
    step(); torch.cuda.synchronize()
    time.sleep(idle)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(121)]
    ev[0].record()
    for i in range(120):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(120)])
    print(f"idle {idle:.0f} s before: steps 0-4 {t[:5].mean():.2f} ms, 5-24 {t[5:25].mean():.2f}, 25-59 {t[25:60].mean():.2f}, 60-119 {t[60:].mean():.2f}   first ten: {np.round(t[:10], 2)}")
