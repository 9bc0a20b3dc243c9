from . import contrastive  # noqa: F401  (fills the criterion registries)
