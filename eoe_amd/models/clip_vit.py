"""Drop-in for the image tower of the reference's vendored CLIP (`src/eoe/models/clip_official/clip/model.py`):
`VisualTransformer` (:202-236), `ResidualAttentionBlock` (:167-188), `Transformer` (:191-199).

Same constructor signatures, same parameter names and shapes (so a reference `state_dict` / snapshot
`{'net': state_dict}` loads unchanged, `logger.py:334-337`), same initial distributions and
`reset_parameters()` behaviour under `ADTrainer.run.copy_model`'s `weight_reset` (`ad_trainer.py:31-34,237-239`).
The torch.nn modules below are parameter CONTAINERS only: their `forward` is never called -- all arithmetic runs
in the HIP kernels of libeoe_hip.so via `eoe_amd.ops`.
"""
import math
from collections import OrderedDict

import torch
from torch import nn

from .. import ops


class _PackedAttention(nn.Module):
    """parameter container with nn.MultiheadAttention's names: in_proj_weight [3D,D], in_proj_bias [3D],
    out_proj.{weight,bias} (model.py:171)"""

    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        # nn.MultiheadAttention._reset_parameters (not `reset_parameters`: weight_reset does not touch it)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)


class ResidualAttentionBlock(nn.Module):
    """model.py:167-188.  forward takes the batch-major residual stream [n, L, D] (fp32) and returns the same."""

    def __init__(self, d_model: int, n_head: int, attn_mask: torch.Tensor = None):
        super().__init__()
        if attn_mask is not None:
            raise NotImplementedError("attention masks belong to the text tower, which is out of scope")
        if d_model != 64 * n_head:
            raise NotImplementedError("the attention kernel is written for head dim 64")
        self.n_head = n_head
        self.attn = _PackedAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", nn.Identity()),            # QuickGELU (model.py:162-164) is fused into the c_fc GEMM epilogue
            ("c_proj", nn.Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = nn.LayerNorm(d_model)
        self.attn_mask = None

    def _params(self):
        return (self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight, self.attn.in_proj_bias,
                self.attn.out_proj.weight, self.attn.out_proj.bias, self.ln_2.weight, self.ln_2.bias,
                self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias)

    def forward_tokens(self, x2d: torch.Tensor, n: int, cls_only: bool = False) -> torch.Tensor:
        """`cls_only`: return the [n, D] class-token rows only (the caller reads nothing else of this block's output); the rows nobody
        reads are then not computed (ops.VitBlockFunction)"""
        return ops.VitBlockFunction.apply(x2d, n, self.n_head, *self._params(), cls_only)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, L, D = x.shape
        return self.forward_tokens(x.reshape(n * L, D).float(), n).reshape(n, L, D)


class Transformer(nn.Module):
    # model.py:191-199
    def __init__(self, width: int, layers: int, heads: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.resblocks(x)


class VisualTransformer(nn.Module):
    """model.py:202-236.  `normalize=(mean, std)` optionally fuses the trainer's per-channel Normalize
    (`ad_trainer.py:413-425`, `transformations.py:126-138`) into the patch-extraction kernel."""

    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, output_dim
        self.conv1 = nn.Conv2d(in_channels=3, out_channels=width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self.normalize = None           # optional (mean[3], std[3]) device tensors

    def set_normalize(self, mean, std):
        if mean is None:
            self.normalize = None
        else:
            dev = self.proj.device
            self.normalize = (torch.as_tensor(mean, dtype=torch.float32, device=dev).contiguous(),
                              torch.as_tensor(std, dtype=torch.float32, device=dev).contiguous())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("eoe_amd.VisualTransformer runs on the GPU only (no CPU fallback)")
        n = x.shape[0]
        mean, std = self.normalize if self.normalize is not None else (None, None)
        # all 16-bit weight copies that the optimiser step made stale, in one launch (48 matrices for 12 blocks)
        ops.shadow.refresh([w for blk in self.transformer.resblocks
                            for w in (blk.attn.in_proj_weight, blk.attn.out_proj.weight, blk.mlp.c_fc.weight, blk.mlp.c_proj.weight)]
                           + [self.proj])
        tok = ops.VitEmbedFunction.apply(x, self.conv1.weight, self.class_embedding, self.positional_embedding,
                                         self.ln_pre.weight, self.ln_pre.bias, self.patch_size, mean, std)
        # ln_post reads x[:, 0, :] only (model.py:231-232): the last block is asked for its class-token rows alone -- [n, D], which the
        # head takes as a one-token sequence -- and skips the out-projection, LayerNorm-2 and MLP of the other L - 1 tokens
        blocks = list(self.transformer.resblocks)
        for i, blk in enumerate(blocks):
            tok = blk.forward_tokens(tok, n, cls_only=ops.VIT_CLS_ONLY_LAST and i == len(blocks) - 1)
        return ops.VitHeadFunction.apply(tok, n, self.ln_post.weight, self.ln_post.bias, self.proj)


def convert_weights(model: nn.Module) -> nn.Module:
    """The reference's fp16-weights mode (`clip/model.py:371-392`; `build_model` applies it to every CLIP model, `:430`, and `clip.load`
    undoes it with `model.float()` on the CPU only, `clip.py:116`): the weight and bias of every convolution / linear layer, the packed
    attention projections (`in_proj_weight`, `in_proj_bias`) and the `proj` / `text_projection` matrices become fp16 tensors; LayerNorm
    parameters, `class_embedding` and `positional_embedding` stay fp32.  Here the storage stays fp32 (the kernels' 16-bit operand copies
    are then exact) and holds the fp16-rounded values; the parameters are marked so that `eoe_amd.FusedSGD` -- the optimiser the
    reference builds for CLIP models, `ad_trainer.py:380-381` -- updates them with torch's fp16 arithmetic (`eoe_sgd_multi`,
    EOE_CHUNK_FP16).  Returns the model."""
    def mark(p):
        if p is None:
            return
        with torch.no_grad():
            p.copy_(p.to(torch.float16).to(torch.float32))
        p._eoe_fp16_weight = True

    for m in model.modules():
        if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
            mark(m.weight)
            mark(m.bias)
        if isinstance(m, (_PackedAttention, nn.MultiheadAttention)):
            for name in ("in_proj_weight", "q_proj_weight", "k_proj_weight", "v_proj_weight", "in_proj_bias", "bias_k", "bias_v"):
                mark(getattr(m, name, None))
        for name in ("text_projection", "proj"):
            attr = getattr(m, name, None)
            if isinstance(attr, nn.Parameter):
                mark(attr)
    return model
