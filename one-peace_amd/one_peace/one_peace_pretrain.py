"""Mirror of one_peace/models/one_peace/one_peace_pretrain.py.

The contrastive branch (``forward(..., encoder_type in {text,image,audio})`` -> ``(normalised CLS projection, features)``
and ``return_logit_scale``) is the hot path and runs on the HIP kernels.  The masked-feature (DCL) branch with the small
decoder -- ``*_preserve_ids`` -- keeps the reference semantics through the torch-op path ("next")."""
import logging
import math
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn as nn

from ..components import Linear, trunc_normal_
from ..registry import register_model
from ..unify_model_config import AdjustEncDecConfig, UnifyModelConfig
from .one_peace_base import ModelWrapper, OnePeaceBaseModel, init_one_peace_params
from .one_peace_retrieval import normalized_projection

logger = logging.getLogger(__name__)


@dataclass
class OnePeacePretrainConfig(UnifyModelConfig):
    decoder: Optional[AdjustEncDecConfig] = None
    reset_logit_scale: bool = False
    logit_scale_init: float = 1 / 0.07
    stage2_pretrain: bool = False


@register_model("one_peace_pretrain", dataclass=OnePeacePretrainConfig)
class OnePeacePretrainModel(OnePeaceBaseModel):
    def __init__(self, cfg, src_dict):
        super().__init__(cfg, src_dict)
        enc, dec = cfg.encoder, getattr(cfg, "decoder", None)
        He = enc.embed_dim
        self.encoder_wrapper = ModelWrapper(enc, src_dict)
        if dec is not None:
            self.decoder_wrapper = ModelWrapper(dec)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(cfg.logit_scale_init))
        for m, on in (("text", enc.use_text_moe), ("image", enc.use_image_moe), ("audio", enc.use_audio_moe)):
            if on:
                setattr(self, m + "_proj", Linear(He, He))
        for m in ("text", "image", "audio"):
            setattr(self, m + "_mask_token", None)
            if dec is not None and getattr(enc, "use_%s_moe" % m) and getattr(dec, "use_%s_moe" % m):
                setattr(self, "decoder_%s_embed" % m, Linear(He, dec.embed_dim))
                tok = nn.Parameter(torch.zeros(1, dec.embed_dim))
                trunc_normal_(tok)
                setattr(self, m + "_mask_token", tok)
                setattr(self, m + "_mask_head", Linear(dec.embed_dim, He))
        self.apply(init_one_peace_params)
        if cfg.stage2_pretrain:  # audio-language stage: only the audio branch trains (reference :98-104)
            self.text_proj.requires_grad_(False)
            self.encoder_wrapper.requires_grad_(False)
            self.encoder_wrapper.audio_adapter.requires_grad_(True)
            self.encoder_wrapper.fusion_model.audio_layer_norm.requires_grad_(True)
            for layer in self.encoder_wrapper.fusion_model.layers:
                layer.audio_ffn.requires_grad_(True)

    @classmethod
    def build_model(cls, cfg, task):
        return cls(cfg, task.source_dictionary)

    def forward(self, src_tokens: Optional[torch.Tensor] = None, text_preserve_ids=None,
                src_images: Optional[torch.Tensor] = None, image_preserve_ids=None,
                src_audios: Optional[torch.Tensor] = None, audio_padding_masks=None, audio_preserve_ids=None,
                encoder_type: str = None, return_logit_scale: bool = False):
        if return_logit_scale:
            with torch.no_grad():
                self.logit_scale.clamp_(0, math.log(100))
            return self.logit_scale.exp()
        tf, imf, af = self.encoder_wrapper(
            src_tokens=src_tokens, text_preserve_ids=text_preserve_ids, src_images=src_images,
            image_preserve_ids=image_preserve_ids, src_audios=src_audios, audio_padding_masks=audio_padding_masks,
            audio_preserve_ids=audio_preserve_ids, encoder_type=encoder_type)
        if text_preserve_ids is not None or image_preserve_ids is not None or audio_preserve_ids is not None:
            te = self.decoder_text_embed(tf) if tf is not None else None
            ie = self.decoder_image_embed(imf) if imf is not None else None
            ae = self.decoder_audio_embed(af) if af is not None else None
            dt, di, da = self.decoder_wrapper(
                src_tokens=src_tokens, text_preserve_ids=text_preserve_ids, text_preserve_embed=te,
                text_mask_token=self.text_mask_token, src_images=src_images, image_preserve_ids=image_preserve_ids,
                image_preserve_embed=ie, image_mask_token=self.image_mask_token, src_audios=src_audios,
                audio_padding_masks=audio_padding_masks, audio_preserve_ids=audio_preserve_ids, audio_preserve_embed=ae,
                audio_mask_token=self.audio_mask_token, encoder_type=encoder_type)
            return (self.text_mask_head(dt) if dt is not None else None,
                    self.image_mask_head(di) if di is not None else None,
                    self.audio_mask_head(da) if da is not None else None)
        if encoder_type == "text":
            return normalized_projection(self.text_proj, tf[:, 0, :]), tf
        if encoder_type == "image":
            return normalized_projection(self.image_proj, imf[:, 0, :]), imf
        if encoder_type == "audio":
            return normalized_projection(self.audio_proj, af[:, 0, :]), af
        if encoder_type == "vl":
            return tf, imf
        if encoder_type == "al":
            return tf, af
        raise NotImplementedError(encoder_type)

    def upgrade_state_dict_named(self, state_dict, name):
        super().upgrade_state_dict_named(state_dict, name)
        if self.cfg.reset_logit_scale:
            state_dict.pop("logit_scale", None)
        if self.cfg.stage2_pretrain:
            for k in list(state_dict.keys()):
                if "image_" in k:
                    del state_dict[k]
        prefix = name + "." if name != "" else ""
        for k, v in self.state_dict().items():
            if prefix + k not in state_dict:
                logger.info("%s not exists, re-initialized", prefix + k)
                state_dict[prefix + k] = v
