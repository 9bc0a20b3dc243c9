"""Config tree of the hot path: the field names are the attributes the reference modules read
(one_peace/models/unify_model_config.py:14-218, EncDecBaseConfig fields from
fairseq/models/transformer/transformer_config.py:26-50).  Plain dataclasses, so they work with or without fairseq /
omegaconf; the reference's YAML values (run_scripts/pretrain/pretrain_vl_3B.yaml:89-130) map 1:1."""
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class TextAdapterConfig:
    bucket_size: int = 256
    layernorm_embedding: bool = False
    add_type_embedding: bool = False
    shrink_alpha: float = 1.0
    dropout: float = 0.0
    use_attn_bias: bool = False


@dataclass
class ImageAdapterConfig:
    bucket_size: int = 16
    rel_bucket_size: int = 16
    layernorm_embedding: bool = False
    add_type_embedding: bool = False
    vision_encoder_type: str = "hmlp"  # mlp | hmlp | none
    shrink_alpha: float = 1.0
    dropout: float = 0.0
    use_attn_bias: bool = False


@dataclass
class AudioAdapterConfig:
    feature_embed_dim: int = 512
    feature_encoder_spec: Optional[str] = "[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] + [(512,2,2)]"
    abs_pos_type: str = "conv"
    conv_pos_depth: int = 5
    conv_pos_width: int = 95
    conv_pos_groups: int = 16
    conv_pos_pre_ln: bool = False
    bucket_size: int = 256
    layernorm_embedding: bool = False
    add_type_embedding: bool = False
    shrink_alpha: float = 1.0
    dropout: float = 0.0
    use_attn_bias: bool = False
    conv_bias: bool = False
    freeze_extractor: bool = False


@dataclass
class AdjustEncDecConfig:
    embed_path: Optional[str] = None
    embed_dim: int = 512
    ffn_embed_dim: int = 2048
    layers: int = 6
    attention_heads: int = 8
    normalize_before: bool = False
    learned_pos: bool = False
    layerdrop: float = 0.0
    text_adapter: TextAdapterConfig = field(default_factory=TextAdapterConfig)
    image_adapter: ImageAdapterConfig = field(default_factory=ImageAdapterConfig)
    audio_adapter: AudioAdapterConfig = field(default_factory=AudioAdapterConfig)
    drop_path_rate: float = 0.0
    magneto_scale_attn: bool = False
    scale_attn: bool = True
    scale_fc: bool = True
    scale_heads: bool = True
    use_text_moe: bool = True
    use_image_moe: bool = True
    use_audio_moe: bool = True
    use_layer_scale: bool = True
    layer_scale_init_value: float = 1e-2
    activation_fn: str = "gelu"
    dropout: float = 0.1
    attention_dropout: float = 0.0
    activation_dropout: float = 0.0
    max_positions: int = 1024
    checkpoint_activations: bool = False
    fsdp_checkpoint_wrap_layer_preserve_frequency: Optional[int] = 1
    fsdp_checkpoint_wrap_layer_skip_frequency: Optional[int] = 1000
    offload_activations: bool = False


@dataclass
class UnifyModelConfig:
    encoder: AdjustEncDecConfig = field(default_factory=AdjustEncDecConfig)


def one_peace_encoder_config(embed_dim=1536, ffn_embed_dim=6144, layers=40, attention_heads=24, drop_path_rate=0.4,
                             layer_scale_init_value=1e-6, image_bucket_size=16, image_rel_bucket_size=16,
                             text_bucket_size=256, audio_bucket_size=512, use_text_moe=True, use_image_moe=True,
                             use_audio_moe=True, checkpoint_activations=True):
    """The shipped ONE-PEACE-4B encoder settings (pretrain_vl_3B.yaml:89-130, pretrain_al_3B.yaml) with overridable
    dimensions; the tiny/micro test models use the same switches at smaller sizes."""
    return AdjustEncDecConfig(
        embed_dim=embed_dim, ffn_embed_dim=ffn_embed_dim, layers=layers, attention_heads=attention_heads,
        normalize_before=True, learned_pos=True, drop_path_rate=drop_path_rate, use_text_moe=use_text_moe,
        use_image_moe=use_image_moe, use_audio_moe=use_audio_moe, attention_dropout=0.0, dropout=0.0,
        magneto_scale_attn=True, scale_attn=False, scale_fc=True, scale_heads=False, use_layer_scale=True,
        layer_scale_init_value=layer_scale_init_value, checkpoint_activations=checkpoint_activations,
        text_adapter=TextAdapterConfig(bucket_size=text_bucket_size, use_attn_bias=True),
        image_adapter=ImageAdapterConfig(bucket_size=image_bucket_size, rel_bucket_size=image_rel_bucket_size,
                                         vision_encoder_type="hmlp", use_attn_bias=True),
        audio_adapter=AudioAdapterConfig(bucket_size=audio_bucket_size, use_attn_bias=True))
