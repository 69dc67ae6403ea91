"""CPU restatement of the optimiser and LR schedule of the hot path (test infrastructure).

Follows reference `src/eoe/training/ad_trainer.py:383-384` -- `torch.optim.Adam(params, lr, weight_decay=wdk,
amsgrad=False)` (betas (0.9, 0.999), eps 1e-8, L2-in-gradient weight decay, bias-corrected) and
`MultiStepLR(opt, milestones, 0.1)` stepped once per epoch (`ad_trainer.py:468`).  Third-party arithmetic:
torch >= 2.3.1 (`src/requirements.txt:7`), single-tensor Adam update order.
"""
import math
import numpy as np
from typing import List, Optional
import torch


class AdamState:
    def __init__(self, params: List[torch.Tensor]):
        self.step = [0 for _ in params]
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]


def adam_step(params: List[torch.Tensor], grads: List[Optional[torch.Tensor]], state: AdamState,
              lr: float, weight_decay: float = 0.0, beta1: float = 0.9, beta2: float = 0.999,
              eps: float = 1e-8) -> None:
    """in-place Adam update; params whose grad is None are skipped and their step does not advance
    (SURVEY.md section 7 'Adam details')."""
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            if g is None:
                continue
            state.step[i] += 1
            t = state.step[i]
            if weight_decay != 0:
                g = g + weight_decay * p
            m, v = state.m[i], state.v[i]
            m.add_((g - m) * (1 - beta1))                    # exp_avg.lerp_(grad, 1 - beta1)
            v.mul_(beta2).add_(g * g * (1 - beta2))          # exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
            bc1 = 1 - beta1 ** t
            bc2 = 1 - beta2 ** t
            step_size = lr / bc1
            denom = v.sqrt() / math.sqrt(bc2) + eps
            p.add_(m / denom * (-step_size))


def multistep_lr(base_lr: float, milestones: List[int], epoch: int, gamma: float = 0.1) -> float:
    """learning rate in effect during `epoch` (0-based) when sched.step() is called at each epoch end."""
    k = sum(1 for m in milestones if epoch >= m)
    return base_lr * (gamma ** k)


class SgdState:
    def __init__(self, params):
        self.buf = [None for _ in params]


def sgd_step(params, grads, state: SgdState, lr: float, momentum: float = 0.9, weight_decay: float = 0.0,
             nesterov: bool = True) -> None:
    """in-place `torch.optim.SGD` update (dampening 0) as constructed for CLIP models, `ad_trainer.py:380-381`;
    third-party arithmetic: torch's single-tensor SGD (first step: buf = grad)."""
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            if g is None:
                continue
            if weight_decay != 0:
                g = g + weight_decay * p
            if momentum != 0:
                if state.buf[i] is None:
                    state.buf[i] = g.clone()
                else:
                    state.buf[i].mul_(momentum).add_(g)
                g = g + momentum * state.buf[i] if nesterov else state.buf[i]
            p.add_(g * (-lr))


def _r16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).to(torch.float32)


def sgd_step_fp16(p: torch.Tensor, g: torch.Tensor, buf: Optional[torch.Tensor], lr: float, momentum: float = 0.9, weight_decay: float = 0.0,
                  nesterov: bool = True, alpha_fp16: bool = False):
    """one step of `torch.optim.SGD` on an fp16 parameter with an fp16 gradient (the reference's fp16-weights mode: `convert_weights`,
    clip/model.py:371-392 + ad_trainer.py:380-381), restated on fp32 tensors that hold fp16 values: torch's update is a chain of
    elementwise ops -- grad.add(param, alpha=wd); buf.mul_(momentum); buf.add_(grad); grad.add(buf, alpha=momentum); param.add_(grad,
    alpha=-lr) -- each computed in fp32 and rounded to fp16 once.  `alpha_fp16`: torch's CPU kernels round the `alpha` scalar of add to
    fp16 first (fixture g16 is made on the CPU); the GPU kernels keep it in fp32 (what `eoe_sgd_multi` implements).  Returns (p, buf)."""
    a = (lambda v: float(torch.tensor(v, dtype=torch.float16))) if alpha_fp16 else (lambda v: float(np.float32(v)))
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)      # noqa: E731
    g = _r16(g)
    if weight_decay != 0:
        g = _r16(g + f32(a(weight_decay)) * p)
    step = g
    if momentum != 0:
        buf = g.clone() if buf is None else _r16(_r16(buf * f32(float(np.float32(momentum)))) + g)
        step = _r16(g + f32(a(momentum)) * buf) if nesterov else buf
    return _r16(p + f32(a(-lr)) * step), buf
