"""Mirror of one_peace/models/transformer/transformer_layer.py (GeGLU, TransformerEncoderLayer).

Parameter names/shapes are the reference's (state-dict contract): ``self_attn.*``,
``self_attn_layer_norm``, ``final_layer_norm``, ``{text,image,audio}_ffn.{0.wi_0,0.wi_1,2,3}``, ``gamma_1/2``,
optional ``attn_ln``.  ``forward`` keeps the reference signature (time-major x, dense additive bias) and runs torch
ops; ``forward_fused`` is the MI355X path: one HIP-backed autograd function per branch (attention over the whole --
possibly joint -- stream, then each modality's rows through its own GeGLU FFN)."""
import logging

import torch
import torch.nn as nn

from .. import ops
from ..components import FairseqDropout, LayerNorm, Linear
from .multihead_attention import MultiheadAttention

logger = logging.getLogger(__name__)


class GeGLU(nn.Module):
    def __init__(self, embed_dim, ffn_dim):
        super().__init__()
        self.wi_0 = Linear(embed_dim, ffn_dim, bias=False)
        self.wi_1 = Linear(embed_dim, ffn_dim, bias=False)
        self.act = nn.GELU()

    def forward(self, x):
        return self.act(self.wi_0(x)) * self.wi_1(x)


def sample_path_scale(batch, drop_prob, training, device):
    """Per-sample stochastic-depth multiplier: 0 with prob p else 1/(1-p) (transformer_layer.py:78-85)."""
    if not training or drop_prob <= 0.0:
        return None
    keep = 1.0 - drop_prob
    return torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, cfg, drop_path_rate=0.0):
        super().__init__()
        self.cfg = cfg
        self.embed_dim, self.ffn_embed_dim = cfg.embed_dim, cfg.ffn_embed_dim
        self.self_attn = MultiheadAttention(self.embed_dim, cfg.attention_heads, dropout=cfg.attention_dropout,
                                            scale_heads=cfg.scale_heads, magneto_scale_attn=cfg.magneto_scale_attn)
        self.self_attn_layer_norm = LayerNorm(self.embed_dim)
        self.dropout_module = FairseqDropout(cfg.dropout, module_name=type(self).__name__)
        self.activation_dropout_module = FairseqDropout(float(cfg.activation_dropout), module_name=type(self).__name__)
        self.dropout_prob, self.drop_path_prob = cfg.dropout, drop_path_rate
        for name, on in (("text", cfg.use_text_moe), ("image", cfg.use_image_moe), ("audio", cfg.use_audio_moe)):
            if on:
                setattr(self, name + "_ffn", self._make_ffn(cfg))
        self.attn_ln = LayerNorm(self.embed_dim) if cfg.scale_attn else None
        self.final_layer_norm = LayerNorm(self.embed_dim)
        if cfg.use_layer_scale:
            self.gamma_1 = nn.Parameter(cfg.layer_scale_init_value * torch.ones(self.embed_dim))
            self.gamma_2 = nn.Parameter(cfg.layer_scale_init_value * torch.ones(self.embed_dim))
        else:
            self.gamma_1 = self.gamma_2 = None

    def _make_ffn(self, cfg):
        return nn.Sequential(GeGLU(self.embed_dim, self.ffn_embed_dim), self.activation_dropout_module,
                             LayerNorm(self.ffn_embed_dim) if cfg.scale_fc else nn.Identity(),
                             Linear(self.ffn_embed_dim, self.embed_dim))

    # ------------------------------------------------------------------ torch-op path (reference signature)
    def _branch(self, y, gamma, residual):
        if self.training and self.dropout_prob > 0.0:
            y = torch.nn.functional.dropout(y, self.dropout_prob)
        if gamma is not None:
            y = gamma * y
        ps = sample_path_scale(y.shape[1], self.drop_path_prob, self.training, y.device)
        if ps is not None:
            y = y * ps.to(y.dtype).view(1, -1, 1)
        return residual + y

    def forward(self, x, encoder_padding_mask, self_attn_bias=None, encoder_type=None, text_seq_len=None,
                image_seq_len=None, audio_seq_len=None):
        y = self.self_attn(self.self_attn_layer_norm(x), key_padding_mask=encoder_padding_mask, attn_mask=self_attn_bias)
        if self.attn_ln is not None:
            y = self.attn_ln(y)
        x = self._branch(y, self.gamma_1, x)
        h = self.final_layer_norm(x)
        if encoder_type in ("text", "image", "audio"):
            y = getattr(self, encoder_type + "_ffn")(h)
        elif encoder_type in ("vl", "al"):
            other, n_other = ("image", image_seq_len) if encoder_type == "vl" else ("audio", audio_seq_len)
            y = torch.cat([self.text_ffn(h[:text_seq_len]), getattr(self, other + "_ffn")(h[-n_other:])], dim=0)
        else:
            raise NotImplementedError(encoder_type)
        return self._branch(y, self.gamma_2, x)

    # ------------------------------------------------------------------ MI355X path
    _STREAMS = {"text": ("text",), "image": ("image",), "audio": ("audio",), "vl": ("text", "image"),
                "al": ("text", "audio")}

    def fused_supported(self, encoder_type):
        streams = self._STREAMS.get(encoder_type)
        return (streams is not None and all(hasattr(self, s + "_ffn") for s in streams)
                and self.self_attn.head_dim == 64 and self.attn_ln is None
                and self.self_attn.c_attn is None and not (self.training and self.dropout_prob > 0.0)
                and self.self_attn.dropout_p == 0.0 and float(self.cfg.activation_dropout) == 0.0
                and self.embed_dim % 64 == 0 and self.ffn_embed_dim % 64 == 0)

    def attn_params(self):
        a, sub = self.self_attn, self.self_attn.ln
        return (self.self_attn_layer_norm.weight, self.self_attn_layer_norm.bias, a.q_proj.weight, a.q_proj.bias,
                a.k_proj.weight, a.v_proj.weight, a.v_proj.bias, sub.weight if sub is not None else None,
                sub.bias if sub is not None else None, a.out_proj.weight, a.out_proj.bias, self.gamma_1)

    def ffn_params(self, modality):
        ffn = getattr(self, modality + "_ffn")
        fln = ffn[2] if isinstance(ffn[2], nn.LayerNorm) else None
        return (self.final_layer_norm.weight, self.final_layer_norm.bias, ffn[0].wi_0.weight, ffn[0].wi_1.weight,
                fln.weight if fln is not None else None, fln.bias if fln is not None else None, ffn[3].weight, ffn[3].bias,
                self.gamma_2)

    def fused_params(self, encoder_type):
        return self.attn_params() + self.ffn_params(encoder_type)

    def forward_fused(self, x_bsh, bias_handle, key_pad_u8, encoder_type, seg_lens=None, path_scales=None):
        """x_bsh: [B, S, H] bf16 batch-major; bias_handle: ops.RelPosBias or None; key_pad_u8: [B, Spad] or None.
        For 'vl'/'al' S = seg_lens[0] (text) + seg_lens[1] (image|audio) and the two row ranges take their own FFN
        (transformer_layer.py:206-216) under ONE shared drop-path draw, as in the reference.  path_scales: optional
        (ps1, ps2) fp32 [B] multipliers (None entries = no drop-path) drawn by the caller."""
        B = x_bsh.shape[0]
        keep = not getattr(self.cfg, "checkpoint_activations", False)
        if path_scales is not None:  # drawn for the whole stack by the encoder (one launch instead of two per layer)
            ps1, ps2 = path_scales
        else:
            ps1 = sample_path_scale(B, self.drop_path_prob, self.training, x_bsh.device)
            ps2 = sample_path_scale(B, self.drop_path_prob, self.training, x_bsh.device)
        x = ops.attn_branch(x_bsh, bias_handle, key_pad_u8, ps1, self.self_attn.num_heads, self.attn_params(), keep)
        streams = self._STREAMS[encoder_type]
        if len(streams) == 1:
            return ops.ffn_branch(x, ps2, self.ffn_params(streams[0]), keep)
        n_text = seg_lens[0]
        parts = (x[:, :n_text].contiguous(), x[:, n_text:].contiguous())
        return torch.cat([ops.ffn_branch(part, ps2, self.ffn_params(m), keep) for part, m in zip(parts, streams)], dim=1)

    def forward_fused_multi(self, x2, segs, ps1_rows, ps2s, kept=(None, None)):
        """Lock-step pass over several single-modality streams: x2 [sum rows, H] packs the rows of the streams in `segs`
        (ops.StreamSeg, named by modality).  The attention branch is modality-shared, so it runs once over all rows; every segment
        then goes through its own FFN.  ps1_rows: fp32 [sum rows] per-row drop-path multipliers of the attention branch or None;
        ps2s: per segment fp32 [B] multipliers of the FFN branch or None.  kept: per branch a hip.KeptRows (or None): the branch is
        computed for the samples stochastic depth keeps only (the reference multiplies the others' branch output by zero,
        transformer_layer.py:78-88); its ps argument is then ignored."""
        keep = not getattr(self.cfg, "checkpoint_activations", False)
        k1, k2 = kept
        if k1 is not None:
            csegs, vec = ops.kept_segments(k1, segs, x2.device)
            x2 = ops.attn_branch_multi(x2, csegs, vec, self.self_attn.num_heads, self.attn_params(), keep, kept=k1)
        else:
            x2 = ops.attn_branch_multi(x2, segs, ps1_rows, self.self_attn.num_heads, self.attn_params(), keep)
        own = [self.ffn_params(sg.name)[2:8] for sg in segs]
        shared = (self.final_layer_norm.weight, self.final_layer_norm.bias, self.gamma_2)
        if k2 is not None:
            csegs, vec = ops.kept_segments(k2, segs, x2.device)
            return ops.ffn_branch_multi(x2, csegs, [vec] * len(segs), shared, own, keep, kept=k2)
        return ops.ffn_branch_multi(x2, segs, ps2s, shared, own, keep)

    def upgrade_state_dict_named(self, state_dict, name):
        """Legacy key renames + fill-in of missing keys (reference transformer_layer.py:230-248)."""
        for old, new in (("0", "self_attn_layer_norm"), ("1", "final_layer_norm")):
            for leaf in ("weight", "bias"):
                k = "%s.layer_norms.%s.%s" % (name, old, leaf)
                if k in state_dict:
                    state_dict["%s.%s.%s" % (name, new, leaf)] = state_dict.pop(k)
        prefix = name + "." if name != "" else ""
        for k, v in self.state_dict().items():
            if prefix + k not in state_dict:
                logger.info("%s not exists, re-initialized", prefix + k)
                state_dict[prefix + k] = v
