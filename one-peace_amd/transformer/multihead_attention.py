"""Mirror of one_peace/models/transformer/multihead_attention.py: same constructor, parameters
(q/v/out_proj with bias, k_proj without, optional per-head scale ``c_attn`` and sub-LayerNorm ``ln``) and
``forward(x, key_padding_mask, attn_mask)`` contract (time-major x, additive dense mask, key_padding_mask unused).

This module's own forward is the torch-op path (CPU / fp32 / dense-mask callers).  On MI355X the encoder layer does
not call it: the projections, the attention core and the sub-LN are part of the fused HIP attention branch (ops.AttnBranchFn)."""
import torch
import torch.nn as nn

from ..components import FairseqDropout, LayerNorm, Linear


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0, scale_heads=False, magneto_scale_attn=False):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.dropout_p = dropout
        self.dropout_module = FairseqDropout(dropout, module_name=type(self).__name__)
        self.c_attn = nn.Parameter(torch.ones(num_heads)) if scale_heads else None
        self.ln = LayerNorm(embed_dim) if magneto_scale_attn else None
        self.k_proj = Linear(embed_dim, embed_dim, bias=False)
        self.v_proj = Linear(embed_dim, embed_dim, bias=True)
        self.q_proj = Linear(embed_dim, embed_dim, bias=True)
        self.out_proj = Linear(embed_dim, embed_dim, bias=True)

    def forward(self, x, key_padding_mask=None, attn_mask=None):
        T, B, C = x.shape
        nh, hd = self.num_heads, self.head_dim

        def split(t):  # [T, B, C] -> [B, nh, T, hd]
            return t.view(T, B, nh, hd).permute(1, 2, 0, 3)

        q = split(self.q_proj(x)) * self.scaling
        k, v = split(self.k_proj(x)), split(self.v_proj(x))
        scores = torch.matmul(q, k.transpose(-1, -2))
        if attn_mask is not None:
            scores = scores + attn_mask.view(B, nh, T, T)
        probs = torch.softmax(scores, dim=-1, dtype=torch.float32).to(scores.dtype)
        ctx = torch.matmul(self.dropout_module(probs), v)  # [B, nh, T, hd]
        if self.c_attn is not None:
            ctx = ctx * self.c_attn.view(1, nh, 1, 1)
        ctx = ctx.permute(2, 0, 1, 3).reshape(T, B, C)
        if self.ln is not None:
            ctx = self.ln(ctx)
        return self.out_proj(ctx)
