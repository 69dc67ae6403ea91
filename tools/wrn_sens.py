"""diagnostic: per-tensor gradient-norm deviation of the HIP WideResNet vs the golden vectors, next to the deviation of the
fp32 oracle from the same vectors (conditioning).  python tools/wrn_sens.py [fp16|bf16]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import eoe_amd
from eoe_amd.models import WideResNet
from eoe_amd.ops import hsc_loss
from oracle import models as omodels, objectives, trainer as otrainer

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
eoe_amd.set_compute_dtype(dt)
g = np.load("tests/golden/g5_wideresnet_hsc.npz")
ref = omodels.deterministic_init(omodels.WideResNet(), tag="wrn")
m = WideResNet()
m.load_state_dict(ref.state_dict())
m = m.cuda().train()
x, y = otrainer.synthetic_batch("g5/b0", 2, 2, 224)
ref.train()
objectives.hsc_loss(ref(x), y, 0).backward()
f = m(x.cuda())
l = hsc_loss(f, y.cuda(), 0)
l.backward()
print("loss", l.item(), float(g["losses"][0]))
rows = []
for (n, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
    gn = float(g[f"gnorm/{n}"])
    rows.append((abs(p.grad.double().norm().item() - gn) / (gn + 1e-5), abs(pr.grad.double().norm().item() - gn) / (gn + 1e-5), gn, n))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.3e  oracle32 %.3e  gnorm %.3e  %s" % r)
d = sorted(r[0] for r in rows); s = sorted(r[1] for r in rows)
print("median hip", d[len(d) // 2], "median oracle32", s[len(s) // 2], "p90 hip", d[int(len(d) * .9)], "p90 oracle", s[int(len(s) * .9)])
