"""Encoder-only contrastive model (registry name ``one_peace_retrieval``).

API parity target: one_peace/models/one_peace/one_peace_retrieval.py -- same constructor ``(cfg, src_dict, head_type)``,
``build_model(cfg, task)``, ``forward(src_tokens, src_images, src_audios, audio_padding_masks, return_logit_scale,
encoder_type)`` returning the L2-normalised CLS projection, same parameter names (``encoder_wrapper.*``,
``{text,image,audio}_proj``, ``logit_scale``) and the same checkpoint upgrade rules.  Only the branches that
``head_type`` needs are built."""
import logging
import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..components import Linear
from ..registry import register_model
from ..unify_model_config import UnifyModelConfig
from .one_peace_base import ModelWrapper, OnePeaceBaseModel, init_one_peace_params

logger = logging.getLogger(__name__)

MODALITIES = ("text", "image", "audio")
# which modality towers a head type instantiates (reference :43-50)
_HEAD_USES = {"text": ("text",), "image": ("image",), "audio": ("audio",), "vl": ("text", "image"),
              "al": ("text", "audio"), "val": MODALITIES}
LOGIT_SCALE_MAX = math.log(100)


@dataclass
class OnePeaceRetrievalConfig(UnifyModelConfig):
    copy_rel_pos_table: bool = False


def normalized_projection(proj, cls_features):
    """``F.normalize(proj(cls), dim=1)``: HIP GEMM + HIP L2-normalise for bf16 device tensors, torch ops otherwise."""
    y = proj(cls_features)
    if ops.hip_eligible(y) and y.shape[-1] % 8 == 0:
        return ops.l2_normalize(y)
    return F.normalize(y, dim=1)


def clamped_logit_scale(param):
    """exp(logit_scale) after the in-place clamp to [0, ln 100] the reference applies on every call."""
    with torch.no_grad():
        param.clamp_(0, LOGIT_SCALE_MAX)
    return param.exp()


@register_model("one_peace_retrieval", dataclass=OnePeaceRetrievalConfig)
class OnePeaceRetrievalModel(OnePeaceBaseModel):
    def __init__(self, cfg, src_dict, head_type):
        super().__init__(cfg, src_dict)
        self.head_type = head_type
        used = _HEAD_USES[head_type]
        enc = cfg.encoder
        for m in MODALITIES:
            setattr(enc, "use_%s_moe" % m, m in used)
        per_layer_tables = enc.layers if getattr(cfg, "copy_rel_pos_table", False) else None
        self.encoder_wrapper = ModelWrapper(enc, src_dict, use_text_norm="text" in used, use_image_norm="image" in used,
                                            use_audio_norm="audio" in used, num_layers=per_layer_tables)
        for m in used:
            setattr(self, m + "_proj", Linear(enc.embed_dim, enc.embed_dim))
        self.logit_scale = nn.Parameter(torch.full((), math.log(1 / 0.07)))
        self.apply(init_one_peace_params)
        # The reference wraps every layer in fsdp_wrap(checkpoint_wrapper(...)) here; the fused HIP layer implements both
        # activation policies itself (cfg.encoder.checkpoint_activations), so there is nothing to wrap.

    @classmethod
    def build_model(cls, cfg, task):
        cfg.encoder.image_adapter.rel_bucket_size = task.cfg.patch_image_size // 16
        return cls(cfg, task.source_dictionary, task.cfg.head_type)

    def set_num_updates(self, num_updates):
        super().set_num_updates(num_updates)
        self.num_updates = num_updates

    def forward(self, src_tokens: Optional[torch.Tensor] = None, src_images: Optional[torch.Tensor] = None,
                src_audios: Optional[torch.Tensor] = None, audio_padding_masks: Optional[torch.Tensor] = None,
                return_logit_scale: bool = False, encoder_type: Optional[str] = None):
        if return_logit_scale:
            return clamped_logit_scale(self.logit_scale)
        if encoder_type not in MODALITIES:
            raise NotImplementedError(encoder_type)
        feats = self.encoder_wrapper(src_tokens=src_tokens, src_images=src_images, src_audios=src_audios,
                                     audio_padding_masks=audio_padding_masks, encoder_type=encoder_type)
        tower = feats[MODALITIES.index(encoder_type)]
        return normalized_projection(getattr(self, encoder_type + "_proj"), tower[:, 0, :])

    def forward_multi(self, src_tokens=None, src_images=None, src_audios=None, audio_padding_masks=None):
        """{modality: normalised CLS embedding} of the given inputs -- the same results as one `forward(..., encoder_type=m)` per
        modality -- from ONE lock-step pass through the shared encoder (MI355X path); None when that pass does not apply."""
        feats = self.encoder_wrapper.forward_multi(src_tokens=src_tokens, src_images=src_images, src_audios=src_audios,
                                                   audio_padding_masks=audio_padding_masks)
        if feats is None:
            return None
        return {m: normalized_projection(getattr(self, m + "_proj"), f[:, 0, :]) for m, f in feats.items()}

    # ---- checkpoints ------------------------------------------------------------------------------------------------
    def remove_pretraining_modules(self, state_dict):
        """Drop the towers this head type does not have (a pretraining checkpoint carries all of them)."""
        unused = [m + "_" for m in MODALITIES if m not in _HEAD_USES[self.head_type]]
        for key in [k for k in state_dict if any(tag in k for tag in unused)]:
            del state_dict[key]

    def upgrade_state_dict_named(self, state_dict, name):
        super().upgrade_state_dict_named(state_dict, name)
        self.remove_pretraining_modules(state_dict)
        prefix = name + "." if name != "" else ""
        for key, value in self.state_dict().items():
            if prefix + key not in state_dict:
                logger.info("%s not exists, re-initialized", prefix + key)
                state_dict[prefix + key] = value
