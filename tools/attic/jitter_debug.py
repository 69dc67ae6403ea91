"""diagnostic: each colour op of eoe_color_jitter_u8 alone against oracle/augment.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eoe_amd import data
from oracle import augment, fill
imgs = (fill.fill("cj/img", (4, 20, 28, 3), std=0.7) * 128 + 128).clip(0, 255).astype(np.uint8)
for op, facs in ((0, (0.99, 1.01, 0.5, 1.7)), (1, (0.99, 1.01, 0.5, 1.7)), (2, (0.99, 1.01, 0.5, 1.7)), (3, (0.0071, -0.0093, 0.31, 0.0))):
    f = np.ones((4, 4), np.float32)
    f[:, 3] = 0
    f[:, op] = facs
    order = np.full((4, 4), 9, np.int32)
    order[:, 0] = op
    got = data.color_jitter_u8(torch.from_numpy(imgs).cuda(), torch.arange(4), torch.from_numpy(f), torch.from_numpy(order)).cpu().numpy()
    want = np.stack([augment.color_jitter(im, ff, [op]) for im, ff in zip(imgs, f)])
    d = np.abs(got.astype(int) - want.astype(int))
    print("op", op, "max diff per image", d.reshape(4, -1).max(1), "count", (d > 0).reshape(4, -1).sum(1))
    if d.max() > 0:
        i = np.argwhere(d > 0)[0]
        print("   first:", i, "src", imgs[i[0], i[1], i[2]], "got", got[i[0], i[1], i[2]], "want", want[i[0], i[1], i[2]], "factor", f[i[0], op])
