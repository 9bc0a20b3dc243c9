"""Mirror of one_peace/models/one_peace/one_peace_base.py: ``ModelWrapper`` (adapters + shared encoder, routing by
``encoder_type``), the registered base model and the parameter initialiser.  Fine-tune heads
(MultiheadAttentionPooling / OnePeaceClassifyHead) are out of the hot-path scope."""
import logging
from typing import Optional

import torch
import torch.nn as nn

from ..adapter.audio import AudioAdapter
from ..adapter.image import ImageAdapter
from ..adapter.text import TextAdapter
from ..components import trunc_normal_
from ..registry import BaseFairseqModel, register_model
from ..transformer.transformer_encoder import TransformerEncoder
from ..unify_model_config import UnifyModelConfig

logger = logging.getLogger(__name__)

_TEXT, _IMAGE, _AUDIO = ("text", "vl", "al", "val"), ("image", "vl", "val"), ("audio", "al", "val")


class ModelWrapper(nn.Module):
    def __init__(self, cfg, src_dict=None, use_text_norm=True, use_image_norm=True, use_audio_norm=True, num_layers=None):
        super().__init__()
        H, heads = cfg.embed_dim, cfg.attention_heads
        if cfg.use_text_moe:
            self.text_adapter = TextAdapter(cfg.text_adapter, H, heads, src_dict, num_layers)
        if cfg.use_image_moe:
            self.image_adapter = ImageAdapter(cfg.image_adapter, H, heads, num_layers)
        if cfg.use_audio_moe:
            self.audio_adapter = AudioAdapter(cfg.audio_adapter, H, heads, num_layers)
        self.fusion_model = TransformerEncoder(cfg, src_dict, use_text_norm=use_text_norm, use_image_norm=use_image_norm,
                                               use_audio_norm=use_audio_norm)

    def forward(self, src_tokens: Optional[torch.Tensor] = None, text_preserve_ids=None, text_preserve_embed=None,
                text_mask_token=None, src_images: Optional[torch.Tensor] = None, image_preserve_ids=None,
                image_preserve_embed=None, image_mask_token=None, is_second_image: bool = False,
                src_audios: Optional[torch.Tensor] = None, audio_padding_masks=None, audio_preserve_ids=None,
                audio_preserve_embed=None, audio_mask_token=None, encoder_type: Optional[str] = None,
                return_padding_mask: bool = False):
        t = i = a = None
        if encoder_type in _TEXT:
            t = self.text_adapter(src_tokens, text_preserve_ids, text_preserve_embed, text_mask_token)
        if encoder_type in _IMAGE:
            i = self.image_adapter(src_images, image_preserve_ids, image_preserve_embed, image_mask_token, is_second_image)
        if encoder_type in _AUDIO:
            a = self.audio_adapter(src_audios, audio_padding_masks, preserve_ids=audio_preserve_ids,
                                   preserve_embed=audio_preserve_embed, mask_token=audio_mask_token)
        out = self.fusion_model(t, i, a, encoder_type=encoder_type)
        feats = out["encoder_out"][0].transpose(0, 1)  # B x S x H (text stream first, the other stream last)
        pad = out["encoder_padding_mask"]
        tf = imf = af = tp = ip = ap = None
        if t is not None:
            n = t[0].size(1)
            tf, tp = feats[:, :n, :], pad[:, :n]
        if i is not None:
            n = i[0].size(1)
            imf, ip = feats[:, -n:, :], pad[:, -n:]
        if a is not None:
            n = a[0].size(1)
            af, ap = feats[:, -n:, :], pad[:, -n:]
        if return_padding_mask:
            return tf, imf, af, tp, ip, ap
        return tf, imf, af


def _wrapper_forward_multi(self, src_tokens=None, src_images=None, src_audios=None, audio_padding_masks=None):
    """ModelWrapper.forward_multi: the single-modality passes of the given inputs as one lock-step pass (see
    TransformerEncoder.forward_multi); returns {modality: features [B, S, H]}, or None when the configuration / inputs do not
    qualify (the caller then runs one forward per modality)."""
    probe = next(self.fusion_model.parameters())
    if not self.fusion_model.multi_possible(probe.device, probe.dtype):  # before the adapters run: a fallback must not run them twice
        return None
    infos = {}
    if src_tokens is not None and hasattr(self, "text_adapter"):
        infos["text"] = self.text_adapter(src_tokens, None, None, None)
    if src_images is not None and hasattr(self, "image_adapter"):
        infos["image"] = self.image_adapter(src_images, None, None, None, False)
    if src_audios is not None and hasattr(self, "audio_adapter"):
        infos["audio"] = self.audio_adapter(src_audios, audio_padding_masks, preserve_ids=None, preserve_embed=None, mask_token=None)
    if not self.fusion_model.multi_ok(infos):
        return None
    return self.fusion_model.forward_multi(infos)


ModelWrapper.forward_multi = _wrapper_forward_multi


@register_model("one_peace_base", dataclass=UnifyModelConfig)
class OnePeaceBaseModel(BaseFairseqModel):
    def __init__(self, cfg, src_dict):
        super().__init__()
        self.cfg, self.src_dict = cfg, src_dict

    @classmethod
    def build_model(cls, cfg, task):
        return cls(cfg, task.source_dictionary)

    def no_weight_decay(self):
        # one_peace_base.py:251-259 (including its missing comma, which fuses two of the decoder names)
        return {
            "encoder_wrapper.text_adapter.embed_positions.weight", "encoder_wrapper.text_adapter.cls_embedding",
            "encoder_wrapper.image_adapter.pos_embed", "encoder_wrapper.image_adapter.cls_embedding",
            "encoder_wrapper.audio_adapter.cls_embedding",
            "decoder_wrapper.text_adapter.embed_positions.weight",
            "decoder_wrapper.text_adapter.cls_embeddingdecoder_wrapper.image_adapter.pos_embed",
            "decoder_wrapper.image_adapter.cls_embedding",
            "decoder_wrapper.audio_adapter.embed_positions.weight", "decoder_wrapper.audio_adapter.cls_embedding",
        }


def init_one_peace_params(module):
    """one_peace_base.py:262-274: trunc-normal Linear weights, zero biases, unit LayerNorm."""
    if isinstance(module, nn.Linear):
        trunc_normal_(module.weight)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.LayerNorm):
        if module.elementwise_affine:
            nn.init.zeros_(module.bias)
            nn.init.ones_(module.weight)
    elif isinstance(module, nn.Conv2d) and module.bias is not None:
        nn.init.zeros_(module.bias)
