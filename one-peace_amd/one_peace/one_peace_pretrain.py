"""Mirror of one_peace/models/one_peace/one_peace_pretrain.py.

The contrastive branch (``forward(..., encoder_type in {text,image,audio})`` -> ``(normalised CLS projection, features)``
and ``return_logit_scale``) is the hot path and runs on the HIP kernels.  The masked-feature (DCL) branch -- ``*_preserve_ids``
gathers in the adapters, the small decoder -- runs on the same fused HIP layers (per-sample bias images, SURVEY.md 8f rank 2;
tests/test_model_gpu.py::test_full_pretraining_objective_on_hip).

Deviation from the reference's state-dict: ``cfg.decoder`` is Optional here (a contrastive-only model has no
``decoder_wrapper.*`` / ``decoder_*`` keys); the reference always builds ``decoder_wrapper`` (one_peace_pretrain.py:57-66).
With a decoder config the key set is identical (tests/test_model_cpu.py)."""
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
from .one_peace_retrieval import MODALITIES, clamped_logit_scale, normalized_projection

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
            return clamped_logit_scale(self.logit_scale)
        preserve = {"text": text_preserve_ids, "image": image_preserve_ids, "audio": audio_preserve_ids}
        feats = dict(zip(MODALITIES, self.encoder_wrapper(
            src_tokens=src_tokens, text_preserve_ids=text_preserve_ids, src_images=src_images,
            image_preserve_ids=image_preserve_ids, src_audios=src_audios, audio_padding_masks=audio_padding_masks,
            audio_preserve_ids=audio_preserve_ids, encoder_type=encoder_type)))
        if any(v is not None for v in preserve.values()):
            # masked-feature branch: encoder features of the kept tokens -> small decoder -> per-modality mask heads
            embeds = {m: (getattr(self, "decoder_%s_embed" % m)(f) if f is not None else None) for m, f in feats.items()}
            dec = self.decoder_wrapper(
                src_tokens=src_tokens, text_preserve_ids=text_preserve_ids, text_preserve_embed=embeds["text"],
                text_mask_token=self.text_mask_token, src_images=src_images, image_preserve_ids=image_preserve_ids,
                image_preserve_embed=embeds["image"], image_mask_token=self.image_mask_token, src_audios=src_audios,
                audio_padding_masks=audio_padding_masks, audio_preserve_ids=audio_preserve_ids,
                audio_preserve_embed=embeds["audio"], audio_mask_token=self.audio_mask_token, encoder_type=encoder_type)
            return tuple(getattr(self, m + "_mask_head")(d) if d is not None else None for m, d in zip(MODALITIES, dec))
        if encoder_type in MODALITIES:
            f = feats[encoder_type]
            return normalized_projection(getattr(self, encoder_type + "_proj"), f[:, 0, :]), f
        if encoder_type == "vl":
            return feats["text"], feats["image"]
        if encoder_type == "al":
            return feats["text"], feats["audio"]
        raise NotImplementedError(encoder_type)

    def forward_multi(self, src_tokens=None, src_images=None, src_audios=None, audio_padding_masks=None):
        """{modality: (normalised CLS embedding, features)} of the given UNMASKED inputs -- what one `forward(..., encoder_type=m)` per
        modality returns (lines 91-93 above) -- from ONE lock-step pass through the shared encoder (MI355X path); None when that pass
        does not apply.  The criterion's first two passes (image_text_pretrain_loss.py:76-83) qualify."""
        feats = self.encoder_wrapper.forward_multi(src_tokens=src_tokens, src_images=src_images, src_audios=src_audios,
                                                   audio_padding_masks=audio_padding_masks)
        if feats is None:
            return None
        return {m: (normalized_projection(getattr(self, m + "_proj"), f[:, 0, :]), f) for m, f in feats.items()}

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
