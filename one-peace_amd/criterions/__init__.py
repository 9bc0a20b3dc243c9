from . import contrastive, pretrain  # noqa: F401  (fills the criterion registries)
