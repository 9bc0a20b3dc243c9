"""Mirror of one_peace/models/one_peace/one_peace_retrieval.py: encoder-only contrastive model returning the
L2-normalised CLS projection per modality; builds only the branches ``head_type`` needs."""
import logging
import math
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..components import Linear
from ..registry import register_model
from ..unify_model_config import AdjustEncDecConfig, UnifyModelConfig
from .one_peace_base import ModelWrapper, OnePeaceBaseModel, init_one_peace_params

logger = logging.getLogger(__name__)


@dataclass
class OnePeaceRetrievalConfig(UnifyModelConfig):
    copy_rel_pos_table: bool = False


def normalized_projection(proj, cls_features):
    """F.normalize(proj(cls), dim=1) (one_peace_retrieval.py:110-121) - HIP GEMM + HIP L2-normalise on bf16 device."""
    y = proj(cls_features)
    if ops.hip_eligible(y) and y.shape[-1] % 8 == 0:
        return ops.l2_normalize(y)
    return F.normalize(y, dim=1)


@register_model("one_peace_retrieval", dataclass=OnePeaceRetrievalConfig)
class OnePeaceRetrievalModel(OnePeaceBaseModel):
    def __init__(self, cfg, src_dict, head_type):
        super().__init__(cfg, src_dict)
        enc = cfg.encoder
        self.head_type = head_type
        enc.use_text_moe = head_type in ("text", "vl", "al", "val")
        enc.use_image_moe = head_type in ("image", "vl", "val")
        enc.use_audio_moe = head_type in ("audio", "al", "val")
        self.encoder_wrapper = ModelWrapper(enc, src_dict, use_text_norm=enc.use_text_moe, use_image_norm=enc.use_image_moe,
                                            use_audio_norm=enc.use_audio_moe,
                                            num_layers=enc.layers if getattr(cfg, "copy_rel_pos_table", False) else None)
        H = enc.embed_dim
        if enc.use_text_moe:
            self.text_proj = Linear(H, H)
        if enc.use_image_moe:
            self.image_proj = Linear(H, H)
        if enc.use_audio_moe:
            self.audio_proj = Linear(H, H)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self.apply(init_one_peace_params)
        # checkpoint_activations / fsdp_wrap (reference :72-79): the fused HIP layer recomputes its activations in
        # backward by construction, so there is nothing to wrap.

    def set_num_updates(self, num_updates):
        super().set_num_updates(num_updates)
        self.num_updates = num_updates

    def forward(self, src_tokens: Optional[torch.Tensor] = None, src_images: Optional[torch.Tensor] = None,
                src_audios: Optional[torch.Tensor] = None, audio_padding_masks: Optional[torch.Tensor] = None,
                return_logit_scale: bool = False, encoder_type: Optional[str] = None):
        if return_logit_scale:
            with torch.no_grad():
                self.logit_scale.clamp_(0, math.log(100))
            return self.logit_scale.exp()
        tf, imf, af = self.encoder_wrapper(src_tokens=src_tokens, src_images=src_images, src_audios=src_audios,
                                           audio_padding_masks=audio_padding_masks, encoder_type=encoder_type)
        if encoder_type == "text":
            return normalized_projection(self.text_proj, tf[:, 0, :])
        if encoder_type == "image":
            return normalized_projection(self.image_proj, imf[:, 0, :])
        if encoder_type == "audio":
            return normalized_projection(self.audio_proj, af[:, 0, :])
        raise NotImplementedError(encoder_type)

    @classmethod
    def build_model(cls, cfg, task):
        cfg.encoder.image_adapter.rel_bucket_size = task.cfg.patch_image_size // 16
        return cls(cfg, task.source_dictionary, task.cfg.head_type)

    def upgrade_state_dict_named(self, state_dict, name):
        super().upgrade_state_dict_named(state_dict, name)
        self.remove_pretraining_modules(state_dict)
        prefix = name + "." if name != "" else ""
        for k, v in self.state_dict().items():
            if prefix + k not in state_dict:
                logger.info("%s not exists, re-initialized", prefix + k)
                state_dict[prefix + k] = v

    def remove_pretraining_modules(self, state_dict):
        keep = {"text_": self.head_type in ("text", "vl", "al", "val"), "image_": self.head_type in ("image", "vl", "val"),
                "audio_": self.head_type in ("audio", "al", "val")}
        for k in list(state_dict.keys()):
            for tag, on in keep.items():
                if not on and tag in k:
                    del state_dict[k]
                    break
