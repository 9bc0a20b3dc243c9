"""TEST INFRASTRUCTURE ONLY -- loader that runs the *unmodified* reference sources on CPU.

The reference (`/root/reference`, OFA-Sys/ONE-PEACE @ 2024-10-08) cannot be imported as a package in
this image (omegaconf / timm / torchvision / librosa are absent).  Its hot-path
modules, however, only need torch plus a handful of symbols from fairseq/timm.  This module
registers *bare* package shells (so no reference `__init__.py` runs), loads two small fairseq files
verbatim by path, stubs the few remaining symbols, and then imports the reference files as they lie
on disk.  Nothing is copied into this repository.

Used by: `tests/golden/make_golden.py` (fixture generation) and the `-m "not gpu"` tests that pin
`oracle/onepeace_oracle.py` against the reference when `/root/reference` is present.  `/root/reference`
does not exist on the GPU box, so nothing that runs there may import this file.

Reference files executed through this shim:
  one_peace/models/components.py
  one_peace/models/transformer/{multihead_attention,transformer_layer,transformer_encoder}.py
  one_peace/models/adapter/{text,image,audio}.py
  one_peace/models/one_peace/{one_peace_base,one_peace_retrieval,one_peace_pretrain}.py
  one_peace/criterions/{image_text,audio_text}_{retrieval,pretrain}_loss.py
  fairseq/fairseq/modules/{fairseq_dropout,layer_drop}.py
"""
import importlib
import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("ONE_PEACE_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "one_peace", "models"))


def _shell(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    m.__package__ = name
    sys.modules[name] = m
    return m


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_INSTALLED = False


def install():
    """Install the shells/stubs once; idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    if "fairseq" in sys.modules or "one_peace" in sys.modules:
        raise RuntimeError("a real fairseq/one_peace is already imported; refusing to shadow it")
    fs_root = os.path.join(REFERENCE_ROOT, "fairseq", "fairseq")
    op_root = os.path.join(REFERENCE_ROOT, "one_peace")

    # ---- fairseq shell -------------------------------------------------------------------
    fairseq = _shell("fairseq")
    utils = _shell("fairseq.utils")
    # fairseq/fairseq/utils.py:516-527 and :707-714 (three one-liners restated)
    utils.softmax = lambda x, dim, onnx_trace=False: F.softmax(x, dim=dim, dtype=torch.float32)
    utils.log_softmax = lambda x, dim, onnx_trace=False: F.log_softmax(x, dim=dim, dtype=torch.float32)

    def new_arange(x, *size):
        if len(size) == 0:
            size = x.size()
        return torch.arange(size[-1], device=x.device).expand(*size).contiguous()

    utils.new_arange = new_arange
    utils.get_available_activation_fns = lambda: ["relu", "gelu"]
    fairseq.utils = utils

    modules = _shell("fairseq.modules")
    fd = _load_file("fairseq.modules.fairseq_dropout", os.path.join(fs_root, "modules", "fairseq_dropout.py"))
    ld = _load_file("fairseq.modules.layer_drop", os.path.join(fs_root, "modules", "layer_drop.py"))
    modules.FairseqDropout = fd.FairseqDropout
    modules.LayerDropModuleList = ld.LayerDropModuleList
    ca = _shell("fairseq.modules.checkpoint_activations")
    ca.checkpoint_wrapper = lambda m, *a, **k: m
    fairseq.modules = modules

    models = _shell("fairseq.models")

    class FairseqEncoder(nn.Module):
        def __init__(self, dictionary):
            super().__init__()
            self.dictionary = dictionary

    class BaseFairseqModel(nn.Module):
        def __init__(self):
            super().__init__()

        def set_num_updates(self, n):
            pass

        def upgrade_state_dict_named(self, sd, name):
            pass

    def register_model(name, dataclass=None):
        def deco(cls):
            return cls
        return deco

    models.FairseqEncoder = FairseqEncoder
    models.BaseFairseqModel = BaseFairseqModel
    models.register_model = register_model
    fairseq.models = models

    dist_m = _shell("fairseq.distributed")
    dist_m.fsdp_wrap = lambda m, *a, **k: m
    fairseq.distributed = dist_m

    crit = _shell("fairseq.criterions")

    class FairseqCriterion(nn.Module):
        def __init__(self, task):
            super().__init__()
            self.task = task

    def register_criterion(name, dataclass=None):
        def deco(cls):
            return cls
        return deco

    crit.FairseqCriterion = FairseqCriterion
    crit.register_criterion = register_criterion
    fairseq.criterions = crit

    dc = _shell("fairseq.dataclass")

    class FairseqDataclass:
        pass

    dc.FairseqDataclass = FairseqDataclass
    dc.ChoiceEnum = lambda choices: str
    fairseq.dataclass = dc

    metrics = _shell("fairseq.metrics")
    metrics.log_scalar = lambda *a, **k: None
    fairseq.metrics = metrics

    # ---- timm.models.layers.trunc_normal_ (requirements.txt:8 pins timm==0.6.11; init only) -----
    if "timm" not in sys.modules:
        _shell("timm")
        _shell("timm.models")
        tl = _shell("timm.models.layers")
        tl.trunc_normal_ = lambda t, mean=0.0, std=1.0, a=-2.0, b=2.0: nn.init.trunc_normal_(t, mean, std, a, b)

    # ---- one_peace shells (no __init__.py executed) ------------------------------------------------
    _shell("one_peace", op_root)
    _shell("one_peace.models", os.path.join(op_root, "models"))
    _shell("one_peace.models.transformer", os.path.join(op_root, "models", "transformer"))
    _shell("one_peace.models.adapter", os.path.join(op_root, "models", "adapter"))
    _shell("one_peace.models.one_peace", os.path.join(op_root, "models", "one_peace"))
    _shell("one_peace.criterions", os.path.join(op_root, "criterions"))
    umc = _shell("one_peace.models.unify_model_config")

    class UnifyModelConfig:  # placeholder: configs are SimpleNamespace trees (see make_cfg)
        pass

    umc.UnifyModelConfig = UnifyModelConfig
    _INSTALLED = True


def ref_optim():
    """The reference's optimiser leg, loaded verbatim: one_peace/optim/adam.py (class Adam, :124-253),
    one_peace/utils/layer_decay.py (LayerDecayValueAssigner, get_parameter_groups, :8-77) and
    fairseq/fairseq/utils.py (clip_grad_norm_, :349-398).  adam.py's module level needs fairseq.optim / omegaconf names for
    its *config* classes only; they are stubbed (the Adam class itself is plain torch)."""
    install()
    if "one_peace.optim.adam" not in sys.modules:
        fo = _shell("fairseq.optim")

        class FairseqOptimizer:
            def __init__(self, cfg):
                self.cfg = cfg

        fo.FairseqOptimizer = FairseqOptimizer
        fo.register_optimizer = lambda name, dataclass=None: (lambda cls: cls)
        if "omegaconf" not in sys.modules:
            oc = _shell("omegaconf")
            oc.II = lambda key: None
            oc.OmegaConf = type("OmegaConf", (), {"to_container": staticmethod(lambda x: x)})
        _shell("one_peace.optim", os.path.join(REFERENCE_ROOT, "one_peace", "optim"))
        _shell("one_peace.utils", os.path.join(REFERENCE_ROOT, "one_peace", "utils"))
    adam = importlib.import_module("one_peace.optim.adam")
    ld = importlib.import_module("one_peace.utils.layer_decay")
    fu = sys.modules.get("_ref_fairseq_utils_full") or _load_file(
        "_ref_fairseq_utils_full", os.path.join(REFERENCE_ROOT, "fairseq", "fairseq", "utils.py"))
    return SimpleNamespace(Adam=adam.Adam, LayerDecayValueAssigner=ld.LayerDecayValueAssigner,
                           get_parameter_groups=ld.get_parameter_groups, clip_grad_norm_=fu.clip_grad_norm_)


def ref(modname):
    """Import a reference module, e.g. ref('one_peace.models.transformer.transformer_layer')."""
    install()
    return importlib.import_module(modname)


# ------------------------------------------------------------------------------------------------
# Config trees (values follow one_peace/run_scripts/pretrain/pretrain_vl_3B.yaml:89-130 unless
# overridden; field names = the attributes the reference reads)
# ------------------------------------------------------------------------------------------------
def make_cfg(embed_dim=256, ffn_embed_dim=1024, layers=4, attention_heads=4, drop_path_rate=0.0,
             use_text_moe=True, use_image_moe=True, use_audio_moe=True, layer_scale_init_value=1e-2,
             image_bucket_size=16, image_rel_bucket_size=None, text_bucket_size=256, audio_bucket_size=512,
             use_attn_bias=True, vision_encoder_type="hmlp", checkpoint_activations=False):
    text = SimpleNamespace(bucket_size=text_bucket_size, layernorm_embedding=False, add_type_embedding=False,
                           shrink_alpha=1.0, dropout=0.0, use_attn_bias=use_attn_bias)
    image = SimpleNamespace(bucket_size=image_bucket_size, rel_bucket_size=image_rel_bucket_size or image_bucket_size,
                            vision_encoder_type=vision_encoder_type, layernorm_embedding=False,
                            add_type_embedding=False, shrink_alpha=1.0, dropout=0.0, use_attn_bias=use_attn_bias)
    audio = SimpleNamespace(feature_embed_dim=512,
                            feature_encoder_spec='[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] + [(512,2,2)]',
                            abs_pos_type="conv", conv_pos_depth=5, conv_pos_width=95, conv_pos_groups=16,
                            conv_pos_pre_ln=False, bucket_size=audio_bucket_size, layernorm_embedding=False,
                            add_type_embedding=False, shrink_alpha=1.0, dropout=0.0, use_attn_bias=use_attn_bias,
                            conv_bias=False, freeze_extractor=False)
    enc = SimpleNamespace(embed_dim=embed_dim, ffn_embed_dim=ffn_embed_dim, layers=layers,
                          attention_heads=attention_heads, normalize_before=True, learned_pos=True,
                          layerdrop=0.0, drop_path_rate=drop_path_rate, use_text_moe=use_text_moe,
                          use_image_moe=use_image_moe, use_audio_moe=use_audio_moe, attention_dropout=0.0,
                          dropout=0.0, activation_dropout=0.0, activation_fn="gelu", magneto_scale_attn=True,
                          scale_attn=False, scale_fc=True, scale_heads=False, use_layer_scale=True,
                          layer_scale_init_value=layer_scale_init_value, max_positions=1024,
                          checkpoint_activations=checkpoint_activations,
                          fsdp_checkpoint_wrap_layer_preserve_frequency=1,
                          fsdp_checkpoint_wrap_layer_skip_frequency=1000, offload_activations=False,
                          text_adapter=text, image_adapter=image, audio_adapter=audio)
    return SimpleNamespace(encoder=enc, copy_rel_pos_table=False)


class TinyDictionary:
    """Stand-in for fairseq Dictionary: only len() and pad() are read (adapter/text.py:40-44)."""

    def __init__(self, n=50265, pad=1):
        self.n, self._pad = n, pad

    def __len__(self):
        return self.n

    def pad(self):
        return self._pad
