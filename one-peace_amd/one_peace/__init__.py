from . import one_peace_base, one_peace_retrieval, one_peace_pretrain, hub_interface  # noqa: F401  (fills the registries)
