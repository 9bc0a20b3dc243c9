"""Mirror of one_peace/models/adapter/audio.py (AudioAdapter): wav2vec2-style strided Conv1d feature extractor
(7 x [conv -> LayerNorm over channels -> GELU]), LN + Linear to the model width, a convolutional positional
encoder (5 x [grouped conv -> LayerNorm without affine -> GELU]), CLS token and 1-D relative-position tables.

MI355X path (``audio_ops.py``): the whole stem stays channels-last; every convolution is a HIP MFMA GEMM over strided
views of the activation buffer (no im2col, no transposes, no MIOpen), every LayerNorm(+GELU) pair is the fused HIP
kernel on [rows, C], and the 512->H projection is the HIP GEMM."""
from typing import List, Tuple

import torch
import torch.nn as nn

from .. import audio_ops, ops
from ..components import Embedding, FairseqDropout, LayerNorm, Linear, trunc_normal_
from ..relpos import RelPosSpec, add_cls_buckets, make_token_bucket_position
from . import common


class TransposeLast(nn.Module):
    def __init__(self, tranpose_dim=-2):
        super().__init__()
        self.tranpose_dim = tranpose_dim

    def forward(self, x):
        return x.transpose(self.tranpose_dim, -1)


class SamePad(nn.Module):
    def __init__(self, kernel_size):
        super().__init__()
        self.remove = 1 if kernel_size % 2 == 0 else 0

    def forward(self, x):
        return x[:, :, :-self.remove] if self.remove > 0 else x


def _ln_gelu_channels(x_bct, ln):
    """LayerNorm over C followed by GELU on a [B, C, T] tensor."""
    xt = x_bct.transpose(1, 2)
    if ops.hip_eligible(xt) and xt.shape[-1] % 8 == 0:
        y = ops.layer_norm(xt, ln.weight, ln.bias, ln.eps, gelu=True)
    else:
        y = nn.functional.gelu(nn.functional.layer_norm(xt, ln.normalized_shape, ln.weight, ln.bias, ln.eps))
    return y.transpose(1, 2)


class ConvFeatureExtractionModel(nn.Module):
    def __init__(self, conv_layers: List[Tuple[int, int, int]], dropout: float = 0.0, conv_bias: bool = False):
        super().__init__()
        self.conv_layers = nn.ModuleList()
        in_d = 1
        for spec in conv_layers:
            assert len(spec) == 3, "invalid conv definition: " + str(spec)
            dim, k, stride = spec
            conv = nn.Conv1d(in_d, dim, k, stride=stride, bias=conv_bias)
            nn.init.kaiming_normal_(conv.weight)
            self.conv_layers.append(nn.Sequential(
                conv, nn.Dropout(p=dropout), nn.Sequential(TransposeLast(), LayerNorm(dim), TransposeLast()), nn.GELU()))
            in_d = dim

    def forward(self, x):  # [B, T_wav] -> [B, C, T]
        x = x.unsqueeze(1)
        for block in self.conv_layers:
            x = block[1](block[0](x))
            x = _ln_gelu_channels(x, block[2][1])
        return x


class AudioAdapter(nn.Module):
    def __init__(self, cfg, embed_dim, attention_heads, num_layers=None):
        super().__init__()
        self.dropout_module = FairseqDropout(cfg.dropout, module_name=type(self).__name__)
        self.alpha = cfg.shrink_alpha
        if cfg.feature_encoder_spec is not None:
            spec = eval(cfg.feature_encoder_spec)
            feat = spec[-1][0]
            self.embed_audios = nn.Sequential(ConvFeatureExtractionModel(spec, dropout=0.0, conv_bias=cfg.conv_bias),
                                              TransposeLast(), LayerNorm(feat), Linear(feat, embed_dim))
        if cfg.abs_pos_type == "conv":
            depth = cfg.conv_pos_depth
            k = max(3, cfg.conv_pos_width // depth)
            blocks = [nn.Sequential(nn.Conv1d(embed_dim, embed_dim, kernel_size=k, padding=k // 2, groups=cfg.conv_pos_groups),
                                    SamePad(k), TransposeLast(), nn.LayerNorm(embed_dim, elementwise_affine=False),
                                    TransposeLast(), nn.GELU()) for _ in range(depth)]
            self.embed_positions = nn.Sequential(TransposeLast(), *blocks, TransposeLast())
            if cfg.conv_pos_pre_ln:
                self.embed_positions = nn.Sequential(LayerNorm(embed_dim), self.embed_positions)
            self.cls_pos_embed = nn.Parameter(torch.zeros(1, 1, embed_dim))
            trunc_normal_(self.cls_pos_embed)
        elif cfg.abs_pos_type == "fixed":
            self.embed_positions = Embedding(1024 + 2, embed_dim)
        else:
            raise NotImplementedError(cfg.abs_pos_type)
        self.abs_pos_type, self.conv_pos_pre_ln = cfg.abs_pos_type, cfg.conv_pos_pre_ln
        self.layernorm_embedding = LayerNorm(embed_dim) if cfg.layernorm_embedding else None
        self.cls_embedding = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.type_embedding = nn.Parameter(torch.zeros(1, 1, embed_dim)) if cfg.add_type_embedding else None
        if cfg.use_attn_bias:
            n_rel = 2 * cfg.bucket_size - 1
            self.register_buffer("rp_bucket", add_cls_buckets(make_token_bucket_position(cfg.bucket_size, 1024), n_rel))
            self.rel_pos_table_list = nn.ModuleList(
                Embedding(n_rel + 3, attention_heads, zero_init=True) for _ in range(num_layers or 1))
        else:
            self.rel_pos_table_list = None
        self.mask_embedding = nn.Parameter(torch.zeros(1, embed_dim))
        self._buckets = common.BucketCache()
        trunc_normal_(self.cls_embedding)
        trunc_normal_(self.mask_embedding)
        if cfg.abs_pos_type == "fixed":
            trunc_normal_(self.embed_positions.weight)

    def get_rel_pos_bias(self, bsz, seq_len):
        b64 = self.rp_bucket[:seq_len, :seq_len]
        b32 = self._buckets.get(self.rp_bucket, seq_len) if self.rp_bucket.is_cuda else None
        return [RelPosSpec(t.weight, b64, b32) for t in self.rel_pos_table_list]

    def _hip_stem_ok(self, src_audios):
        if not (ops.hip_eligible(src_audios) and hasattr(self, "embed_audios")):
            return False
        blocks = self.embed_audios[0].conv_layers
        c0 = blocks[0][0]
        ok = c0.kernel_size[0] == 10 and c0.stride[0] == 5 and c0.in_channels == 1
        for blk in blocks[1:]:
            c = blk[0]
            ok = ok and c.stride[0] == 2 and c.kernel_size[0] in (2, 3) and c.bias is None and c.in_channels % 32 == 0
        return ok and all(blk[1].p == 0.0 for blk in blocks)

    def _frames(self, src_audios):
        """embed_audios (adapter/audio.py:46-55): conv stack -> LayerNorm(512) -> Linear(512 -> H); [B, T, H]."""
        if self._hip_stem_ok(src_audios):
            feats = audio_ops.feature_extractor(src_audios, self.embed_audios[0].conv_layers)
            return self.embed_audios[3](self.embed_audios[2](feats))
        return self.embed_audios(src_audios)

    def _positions(self, frames):
        """Convolutional positional encoding of the frame embeddings [B, T, H] (adapter/audio.py:57-84)."""
        if self.abs_pos_type != "conv":
            return self.embed_positions.weight[: frames.size(1)].unsqueeze(0).expand(frames.size(0), -1, -1)
        seq = self.embed_positions
        x = frames
        if self.conv_pos_pre_ln:
            x, seq = seq[0](x), seq[1]
        blocks = list(seq)[1:-1]
        conv0 = blocks[0][0]
        cg = conv0.in_channels // conv0.groups
        if (ops.hip_eligible(x) and conv0.kernel_size[0] % 2 == 1 and cg % 8 == 0
                and (conv0.kernel_size[0] + 1) * cg >= (conv0.kernel_size[0] * cg + 63) // 64 * 64):
            for block in blocks:  # channels-last grouped-conv GEMMs + fused LN(no affine)+GELU
                conv = block[0]
                x = audio_ops.grouped_conv1d_same(x, conv.weight, conv.bias, conv.groups)
                x = ops.layer_norm(x, None, None, block[3].eps, gelu=True)
            return x
        x = x.transpose(1, 2)
        for block in blocks:
            x = block[1](block[0](x))
            x = nn.functional.gelu(block[3](x.transpose(1, 2))).transpose(1, 2)
        return x.transpose(1, 2)

    def forward(self, src_audios, padding_mask, preserve_ids=None, preserve_embed=None, mask_token=None):
        """-> x [B, T+1, H], padding_mask [B, T+1] (from the data layer), bias list."""
        bsz, n = padding_mask.size(0), padding_mask.size(1)
        biases = self.get_rel_pos_bias(bsz, n) if self.rel_pos_table_list is not None else None
        if preserve_embed is not None:
            pos = self.embed_positions.weight[:n].unsqueeze(0).expand(bsz, -1, -1)
            emb = common.scatter_preserved(preserve_ids, preserve_embed, mask_token, bsz, n)
        else:
            frames = self._frames(src_audios)
            if preserve_ids is not None:
                padding_mask = preserve_ids.eq(-1)
                ids = preserve_ids.masked_fill(padding_mask, preserve_ids.size(1) - 1)
                frames = common.take_rows(frames, ids[:, 1:] - 1)
                biases = common.take_bias(biases, ids, bsz)
            if self.layernorm_embedding is None and self.alpha == 1.0:
                # emb + pos with emb = [cls | frames], pos = [cls_pos | positions] (adapter/audio.py:186-203) as ONE add over the frames and
                # one over the token: the same sums bit for bit, without the two [B, T + 1, H] concatenations (torch.cat's batched 2-byte
                # copy kernel moved the 98 MB position matrix at 0.07 TB/s: 1.4 ms of the headline step) and the full-size add behind them
                x = common.prepend_token(self.cls_embedding + self.cls_pos_embed, frames + self._positions(frames))
                if self.type_embedding is not None:
                    x = x + self.type_embedding
                return self.dropout_module(x), padding_mask, biases
            pos = common.prepend_token(self.cls_pos_embed, self._positions(frames))
            emb = common.prepend_token(self.cls_embedding, frames)
            if self.layernorm_embedding is not None:
                emb = self.layernorm_embedding(emb)
            if self.alpha != 1.0:
                emb = emb * self.alpha + emb.detach() * (1 - self.alpha)
        x = emb + pos
        if self.type_embedding is not None:
            x = x + self.type_embedding
        return self.dropout_module(x), padding_mask, biases

    def upgrade_state_dict_named(self, state_dict, name):
        prefix = name + "." if name != "" else ""
        common.upgrade_rel_pos_tables(self, state_dict, prefix)
        common.fill_missing(self, state_dict, prefix)
        return state_dict
