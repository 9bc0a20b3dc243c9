"""Mirror of one_peace/models/adapter/image.py (ImageAdapter): hMLP patch stem, CLS token, learned absolute
positions (bicubic resize when the patch grid differs from the bucket), Swin-style 2-D relative-position tables.

MI355X path: the three non-overlapping-patch convolutions of the stem are exactly GEMMs over patch vectors
(adapter/image.py:66-75), so on bf16 device tensors they run through the HIP MFMA GEMM with the LayerNorm2D+GELU
pairs fused into one HIP kernel on channels-last rows; the weights stay ``nn.Conv2d`` parameters (state-dict
contract) and are viewed as [out, patch] matrices on the fly."""
import logging

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..components import Embedding, FairseqDropout, LayerNorm, trunc_normal_
from ..relpos import RelPosSpec, make_image_bucket_position
from . import common

logger = logging.getLogger(__name__)


class LayerNorm2D(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.layer_norm = LayerNorm(embed_dim)

    def forward(self, x):  # [B, C, H, W]
        return self.layer_norm(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


def _patch_rows(x_bhwc, k):
    """channels-last [B, Hh, Ww, C] -> rows of k x k patches [B*(Hh/k)*(Ww/k), k*k*C] ordered (kh, kw, c)."""
    B, Hh, Ww, C = x_bhwc.shape
    p = x_bhwc.view(B, Hh // k, k, Ww // k, k, C).permute(0, 1, 3, 2, 4, 5)
    return p.reshape(B * (Hh // k) * (Ww // k), k * k * C)


def _conv_as_matrix(conv):
    """Conv2d weight [out, in, k, k] -> [out, k*k*in] matching _patch_rows' (kh, kw, c) order."""
    w = conv.weight
    return w.permute(0, 2, 3, 1).reshape(w.size(0), -1)


class ImageAdapter(nn.Module):
    def __init__(self, cfg, embed_dim, attention_heads, num_layers=None):
        super().__init__()
        self.dropout_module = FairseqDropout(cfg.dropout, module_name=type(self).__name__)
        self.alpha = cfg.shrink_alpha
        q = embed_dim // 4
        if cfg.vision_encoder_type == "mlp":
            self.embed_images = nn.Conv2d(3, embed_dim, kernel_size=16, stride=16, bias=False)
        elif cfg.vision_encoder_type == "hmlp":
            self.embed_images = nn.Sequential(nn.Conv2d(3, q, kernel_size=4, stride=4), LayerNorm2D(q), nn.GELU(),
                                              nn.Conv2d(q, q, kernel_size=2, stride=2), LayerNorm2D(q), nn.GELU(),
                                              nn.Conv2d(q, embed_dim, kernel_size=2, stride=2))
        else:
            self.embed_images = None
        self.layernorm_embedding = LayerNorm(embed_dim) if cfg.layernorm_embedding else None
        self.cls_embedding = nn.Parameter(torch.zeros(1, 1, embed_dim))
        if cfg.add_type_embedding:
            self.type_embedding = nn.Parameter(torch.zeros(1, 1, embed_dim))
            self.type_embedding_2 = nn.Parameter(torch.zeros(1, 1, embed_dim))
        else:
            self.type_embedding = self.type_embedding_2 = None
        self.bucket_size = cfg.bucket_size
        self.pos_embed = nn.Parameter(torch.zeros(self.bucket_size ** 2 + 1, embed_dim))
        self.register_buffer("position_idx", torch.arange(self.bucket_size ** 2 + 1))
        if cfg.use_attn_bias:
            self.rel_bucket_size = cfg.rel_bucket_size
            n_rel = (2 * self.rel_bucket_size - 1) ** 2 + 3
            self.register_buffer("rp_bucket", make_image_bucket_position(self.rel_bucket_size, n_rel))
            self.rel_pos_table_list = nn.ModuleList(
                Embedding(n_rel, attention_heads, zero_init=True) for _ in range(num_layers or 1))
        else:
            self.rel_pos_table_list = None
        self._buckets = common.BucketCache()
        trunc_normal_(self.cls_embedding)
        trunc_normal_(self.pos_embed)

    # ---------------------------------------------------------------------------------------------
    def get_rel_pos_bias(self, bsz):
        S = self.rp_bucket.size(0)
        b32 = self._buckets.get(self.rp_bucket, S) if self.rp_bucket.is_cuda else None
        return [RelPosSpec(t.weight, self.rp_bucket, b32) for t in self.rel_pos_table_list]

    def get_embed_positions(self, bsz, window_size):
        pos = self.pos_embed
        if window_size != self.bucket_size:
            grid = pos[1:].view(1, self.bucket_size, self.bucket_size, -1).permute(0, 3, 1, 2)
            grid = F.interpolate(grid.float(), size=(window_size, window_size), mode="bicubic").type_as(pos)
            pos = torch.cat([pos[:1], grid.permute(0, 2, 3, 1).reshape(window_size ** 2, -1)], dim=0)
        return pos.unsqueeze(0).expand(bsz, -1, -1)

    def _stem(self, src_images):
        seq = self.embed_images
        if not (ops.hip_eligible(src_images) and isinstance(seq, nn.Sequential)):
            return seq(src_images).flatten(2).transpose(1, 2)
        B, _, R, _ = src_images.shape
        x = src_images.permute(0, 2, 3, 1).contiguous()  # channels-last
        x = ops.linear(_patch_rows(x, 4), _conv_as_matrix(seq[0]), seq[0].bias)
        x = ops.layer_norm(x, seq[1].layer_norm.weight, seq[1].layer_norm.bias, seq[1].layer_norm.eps, gelu=True)
        x = ops.linear(_patch_rows(x.view(B, R // 4, R // 4, -1), 2), _conv_as_matrix(seq[3]), seq[3].bias)
        x = ops.layer_norm(x, seq[4].layer_norm.weight, seq[4].layer_norm.bias, seq[4].layer_norm.eps, gelu=True)
        x = ops.linear(_patch_rows(x.view(B, R // 8, R // 8, -1), 2), _conv_as_matrix(seq[6]), seq[6].bias)
        return x.view(B, (R // 16) ** 2, -1)

    def forward(self, src_images, preserve_ids=None, preserve_embed=None, mask_token=None, is_second_image=False):
        """-> x [B, (R/16)^2+1, H], padding_mask (all False unless masked-pretraining ids are given), bias list."""
        bsz, win = src_images.size(0), src_images.size(2) // 16
        n = win * win + 1
        padding_mask = torch.zeros(bsz, n, dtype=torch.bool, device=src_images.device)
        padding_mask._all_false = True
        pos = self.get_embed_positions(bsz, win)
        biases = self.get_rel_pos_bias(bsz) if self.rel_pos_table_list is not None else None
        if preserve_embed is not None:
            emb = common.scatter_preserved(preserve_ids, preserve_embed, mask_token, bsz, n)
        else:
            emb = common.prepend_token(self.cls_embedding, self._stem(src_images))
            if preserve_ids is not None:
                padding_mask = preserve_ids.eq(-1)
                ids = preserve_ids.masked_fill(padding_mask, preserve_ids.size(1) - 1)
                emb, pos, biases = common.take_rows(emb, ids), common.take_rows(pos, ids), common.take_bias(biases, ids, bsz)
            if self.layernorm_embedding is not None:
                emb = self.layernorm_embedding(emb)
            if self.alpha != 1.0:
                emb = emb * self.alpha + emb.detach() * (1 - self.alpha)
        x = emb + pos
        if self.type_embedding is not None:
            x = x + self.type_embedding
            if is_second_image:
                x = x + self.type_embedding_2
        return self.dropout_module(x), padding_mask, biases

    # ---------------------------------------------------------------------------------------------
    # load-time only: resize checkpoints trained at another resolution (adapter/image.py:115-162,262-312)
    @staticmethod
    def _geometric_resample(src_size, dst_size, table, heads):
        from scipy import interpolate
        lo, hi = 1.01, 1.5
        half = src_size // 2
        while hi - lo > 1e-6:
            r = (lo + hi) / 2.0
            if (1.0 - r ** half) / (1.0 - r) > dst_size // 2:
                hi = r
            else:
                lo = r
        steps, cur = [], 1.0
        for i in range(half):
            steps.append(cur)
            cur += r ** (i + 1)
        axis = [-s for s in reversed(steps)] + [0] + steps
        tgt = np.arange(-(dst_size // 2.0), dst_size // 2.0 + 0.1, 1.0)
        cols = []
        for h in range(heads):
            z = table[:, h].view(src_size, src_size).float().numpy()
            f = interpolate.interp2d(axis, axis, z, kind="cubic")
            cols.append(torch.Tensor(f(tgt, tgt)).contiguous().view(-1, 1).to(table))
        return torch.cat(cols, dim=-1)

    def upgrade_state_dict_named(self, state_dict, name):
        prefix = name + "." if name != "" else ""
        old = prefix + "rel_pos_table.weight"
        if old in state_dict:
            state_dict[prefix + "rel_pos_table_list.0.weight"] = state_dict.pop(old)
        k0 = prefix + "rel_pos_table_list.0.weight"
        if k0 in state_dict and (2 * self.rel_bucket_size - 1) ** 2 + 3 > state_dict[k0].size(0):
            logger.info("interpolate relative position embedding")
            w = state_dict[k0]
            extra, body = w[-3:], w[:-3]
            src = int(body.size(0) ** 0.5)
            new = self._geometric_resample(src, 2 * self.rel_bucket_size - 1, body.cpu(), w.size(-1)).to(extra)
            state_dict[k0] = torch.cat([new, extra], dim=0)
            state_dict[prefix + "rp_bucket"] = self.state_dict()["rp_bucket"]
        common.upgrade_rel_pos_tables(self, state_dict, prefix)
        kp = prefix + "pos_embed"
        if kp in state_dict and self.bucket_size ** 2 + 1 > state_dict[kp].size(0):
            logger.info("interpolate absolute position embedding")
            w = state_dict[kp]
            side = int((w.size(0) - 1) ** 0.5)
            grid = F.interpolate(w[1:].view(1, side, side, -1).permute(0, 3, 1, 2), size=(self.bucket_size,) * 2,
                                 mode="bicubic")
            state_dict[kp] = torch.cat([w[:1], grid.permute(0, 2, 3, 1).reshape(self.bucket_size ** 2, -1)], dim=0)
            state_dict[prefix + "position_idx"] = self.state_dict()["position_idx"]
        common.fill_missing(self, state_dict, prefix)
        return state_dict
