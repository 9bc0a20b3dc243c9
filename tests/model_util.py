"""Builders shared by the CPU and GPU model tests."""
from types import SimpleNamespace

import torch

from oracle import synth


class TinyDictionary:
    def __init__(self, n=50265, pad=1):
        self.n, self._pad = n, pad

    def __len__(self):
        return self.n

    def pad(self):
        return self._pad


def build_retrieval(cfg_kw, vocab, head_type="val", copy_rel_pos_table=False):
    from one_peace_amd.one_peace.one_peace_retrieval import OnePeaceRetrievalModel
    from one_peace_amd.unify_model_config import one_peace_encoder_config
    kw = dict(cfg_kw)
    for k in ("use_text_moe", "use_image_moe", "use_audio_moe"):
        kw.pop(k, None)
    enc = one_peace_encoder_config(drop_path_rate=0.0, layer_scale_init_value=1e-2, **kw)
    cfg = SimpleNamespace(encoder=enc, copy_rel_pos_table=copy_rel_pos_table)
    torch.manual_seed(0)
    return OnePeaceRetrievalModel(cfg, TinyDictionary(vocab), head_type)


def load_synth(model, shapes=None):
    own = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    if shapes is not None:
        assert own == {k: tuple(v) for k, v in shapes.items()}, "state-dict keys/shapes differ from the reference's"
    sd = synth.synth_state_dict(own)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.split(".")[-1] in synth.NON_SYNTH for k in missing), (missing, unexpected)
    return model
