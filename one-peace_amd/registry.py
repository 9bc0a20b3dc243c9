"""Registration into fairseq's model / criterion registries when fairseq is importable, stand-alone registries
otherwise (this image has no omegaconf, so ``import fairseq`` fails: SURVEY.md section 7).

The reference registers with ``@register_model(name, dataclass=...)`` (fairseq/models/__init__.py:109-153) and
``@register_criterion`` (fairseq/criterions/__init__.py).  The same decorators are used here; with a real fairseq the
classes land in its registries (so ``--user-dir`` + ``_name: one_peace_retrieval`` builds THIS implementation), without
it they land in the dictionaries below."""
import torch.nn as nn

MODEL_REGISTRY, MODEL_DATACLASS_REGISTRY, CRITERION_REGISTRY = {}, {}, {}

try:  # pragma: no cover - exercised only where fairseq is installed
    from fairseq.models import BaseFairseqModel, FairseqEncoder  # type: ignore
    from fairseq.models import register_model as _fs_register_model  # type: ignore
    from fairseq.criterions import FairseqCriterion  # type: ignore
    from fairseq.criterions import register_criterion as _fs_register_criterion  # type: ignore
    HAVE_FAIRSEQ = True
except Exception:  # ImportError or fairseq's own missing dependencies
    HAVE_FAIRSEQ = False

    class BaseFairseqModel(nn.Module):
        """The slice of fairseq_model.py:39-161 the hot path touches."""

        def __init__(self):
            super().__init__()

        @classmethod
        def build_model(cls, cfg, task):
            raise NotImplementedError

        def set_num_updates(self, num_updates):
            for m in self.modules():
                if m is not self and hasattr(m, "set_num_updates"):
                    m.set_num_updates(num_updates)

        def upgrade_state_dict(self, state_dict):
            self.upgrade_state_dict_named(state_dict, "")

        def upgrade_state_dict_named(self, state_dict, name):
            def walk(mod, prefix):
                if len(prefix) > 0:
                    prefix += "."
                for n, c in mod.named_children():
                    if hasattr(c, "upgrade_state_dict_named"):
                        c.upgrade_state_dict_named(state_dict, prefix + n)
                    walk(c, prefix + n)
            walk(self, name)

        def load_state_dict(self, state_dict, strict=True, model_cfg=None, args=None):
            self.upgrade_state_dict(state_dict)
            return super().load_state_dict(state_dict, strict)

        def max_positions(self):
            return None

    class FairseqEncoder(nn.Module):
        def __init__(self, dictionary):
            super().__init__()
            self.dictionary = dictionary

    class FairseqCriterion(nn.Module):
        def __init__(self, task):
            super().__init__()
            self.task = task

        @staticmethod
        def logging_outputs_can_be_summed() -> bool:
            return False


def register_model(name, dataclass=None):
    def deco(cls):
        MODEL_REGISTRY[name] = cls
        MODEL_DATACLASS_REGISTRY[name] = dataclass
        if HAVE_FAIRSEQ:
            from fairseq.models import MODEL_REGISTRY as FS  # type: ignore
            if name not in FS:  # the reference's own class may already hold the name; do not double-register
                return _fs_register_model(name, dataclass=dataclass)(cls)
        return cls
    return deco


def register_criterion(name, dataclass=None):
    def deco(cls):
        CRITERION_REGISTRY[name] = cls
        if HAVE_FAIRSEQ:
            from fairseq.criterions import CRITERION_REGISTRY as FS  # type: ignore
            if name not in FS:
                return _fs_register_criterion(name, dataclass=dataclass)(cls)
        return cls
    return deco


def build_model(name, cfg, task):
    return MODEL_REGISTRY[name].build_model(cfg, task)
