"""Mirror of one_peace/models/adapter/text.py (TextAdapter): token embedding + CLS + learned absolute positions,
T5-style log-bucketed relative-position tables, padding mask with the CLS slot unmasked."""
import torch
import torch.nn as nn

from ..components import Embedding, FairseqDropout, LayerNorm, trunc_normal_
from ..relpos import RelPosSpec, add_cls_buckets, make_token_bucket_position
from . import common


class TextAdapter(nn.Module):
    def __init__(self, cfg, embed_dim, attention_heads, src_dict=None, num_layers=None):
        super().__init__()
        self.dropout_module = FairseqDropout(cfg.dropout, module_name=type(self).__name__)
        self.alpha = cfg.shrink_alpha
        if src_dict is not None:
            self.padding_idx = src_dict.pad()
            self.embed_tokens = Embedding(len(src_dict), embed_dim, self.padding_idx)
        else:
            self.padding_idx, self.embed_tokens = 1, None
        self.layernorm_embedding = LayerNorm(embed_dim) if cfg.layernorm_embedding else None
        self.cls_embedding = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.type_embedding = nn.Parameter(torch.zeros(1, 1, embed_dim)) if cfg.add_type_embedding else None
        self.embed_positions = Embedding(512 + 2, embed_dim)
        if cfg.use_attn_bias:
            n_rel = 2 * cfg.bucket_size - 1
            self.register_buffer("rp_bucket", add_cls_buckets(make_token_bucket_position(cfg.bucket_size, 1024), n_rel))
            self.rel_pos_table_list = nn.ModuleList(
                Embedding(n_rel + 3, attention_heads, zero_init=True) for _ in range(num_layers or 1))
        else:
            self.rel_pos_table_list = None
        self._buckets = common.BucketCache()
        trunc_normal_(self.cls_embedding)
        trunc_normal_(self.embed_positions.weight)
        if self.embed_tokens is not None:
            trunc_normal_(self.embed_tokens.weight)
            with torch.no_grad():
                self.embed_tokens.weight[self.padding_idx].zero_()

    def get_rel_pos_bias(self, bsz, seq_len):
        b64 = self.rp_bucket[:seq_len, :seq_len]
        b32 = self._buckets.get(self.rp_bucket, seq_len) if self.rp_bucket.is_cuda else None
        return [RelPosSpec(t.weight, b64, b32) for t in self.rel_pos_table_list]

    def forward(self, src_tokens, preserve_ids=None, preserve_embed=None, mask_token=None):
        """-> x [B, T+1, H], padding_mask [B, T+1] (bool), list of relative-position biases (or None)."""
        bsz, n = src_tokens.size(0), src_tokens.size(1) + 1
        padding_mask = torch.cat([src_tokens.new_zeros(bsz, 1, dtype=torch.bool), src_tokens.eq(self.padding_idx)], dim=1)
        pos = self.embed_positions.weight[:n].unsqueeze(0).expand(bsz, -1, -1)
        biases = self.get_rel_pos_bias(bsz, n) if self.rel_pos_table_list is not None else None
        if preserve_embed is not None:
            emb = common.scatter_preserved(preserve_ids, preserve_embed, mask_token, bsz, n)
        else:
            emb = common.prepend_token(self.cls_embedding, self.embed_tokens(src_tokens))
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
        return self.dropout_module(x), padding_mask, biases

    def upgrade_state_dict_named(self, state_dict, name):
        prefix = name + "." if name != "" else ""
        common.upgrade_rel_pos_tables(self, state_dict, prefix)
        common.fill_missing(self, state_dict, prefix)
        return state_dict
