"""Operator seam: the factories the reference routes every LayerNorm / Linear / Embedding through
(one_peace/models/components.py:23-44).  Same names, same parameter names/shapes/initialisation; on a bf16
MI355X tensor the forward/backward run the HIP kernels, on anything else the plain torch ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def trunc_normal_(tensor, mean=0.0, std=0.02):
    # components.py:19-20: truncation at +-std (timm semantics = nn.init.trunc_normal_ with absolute bounds)
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=-std, b=std)


class HipLayerNorm(nn.LayerNorm):
    def forward(self, x):
        if ops.hip_eligible(x) and x.shape[-1] % 8 == 0 and x.shape[-1] <= 8192:
            return ops.layer_norm(x, self.weight, self.bias, self.eps)
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)


class HipLinear(nn.Linear):
    def forward(self, x):
        if ops.hip_eligible(x):
            return ops.linear(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


def LayerNorm(normalized_shape, eps=1e-5, elementwise_affine=True):
    return HipLayerNorm(normalized_shape, eps, elementwise_affine)


def Linear(in_features, out_features, bias=True):
    m = HipLinear(in_features, out_features, bias)
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.zeros_(m.bias)
    return m


def Embedding(num_embeddings, embedding_dim, padding_idx=None, zero_init=False):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    if zero_init:
        nn.init.zeros_(m.weight)
    else:
        nn.init.normal_(m.weight, mean=0.0, std=embedding_dim ** -0.5)
        if padding_idx is not None:
            with torch.no_grad():
                m.weight[padding_idx].zero_()
    return m


class FairseqDropout(nn.Module):
    """fairseq/modules/fairseq_dropout.py:16-27 behaviour (train-only dropout with probability p)."""

    def __init__(self, p, module_name=None):
        super().__init__()
        self.p = p
        self.module_name = module_name
        self.apply_during_inference = False

    def forward(self, x, inplace: bool = False):
        if self.p > 0 and (self.training or self.apply_during_inference):
            return F.dropout(x, p=self.p, training=True, inplace=inplace)
        return x
