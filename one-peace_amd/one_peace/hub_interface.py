"""Mirror of the feature-extraction surface of one_peace/models/one_peace/hub_interface.py:
``from_pretrained(...) -> OnePeaceHubInterface`` with ``extract_{text,image,audio,vl}_features`` (:206-225) and the dtype
cast of :107-122.  Raw-data pre-processing (BPE, CLIP resize, 16 kHz layer-normed waveforms, :134-193) stays with the
reference's Python (PIL / torchvision / librosa are not part of the hot path); the ``process_*`` methods here accept
already-tokenised / already-decoded tensors and do the collation only."""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from ..unify_model_config import one_peace_encoder_config
from .one_peace_retrieval import OnePeaceRetrievalModel

_DTYPES = {"float32": torch.float32, "fp32": torch.float32, "fp16": torch.float16, "float16": torch.float16,
           "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


class _Dictionary:
    def __init__(self, n=50265, pad=1, bos=0, eos=2):
        self.n, self._pad, self._bos, self._eos = n, pad, bos, eos

    def __len__(self):
        return self.n

    def pad(self):
        return self._pad

    def bos(self):
        return self._bos

    def eos(self):
        return self._eos


def _cfg_get(tree, key, default=None):
    if isinstance(tree, dict):
        return tree.get(key, default)
    return getattr(tree, key, default)


def build_from_checkpoint_cfg(model_cfg, head_type="vl", patch_image_size=256, vocab_size=50265):
    """Build the retrieval model from the `cfg.model` tree of a reference checkpoint (dict / namespace / omegaconf)."""
    enc = _cfg_get(model_cfg, "encoder")
    kw = {}
    for k in ("embed_dim", "ffn_embed_dim", "layers", "attention_heads", "drop_path_rate", "layer_scale_init_value"):
        v = _cfg_get(enc, k)
        if v is not None:
            kw[k] = v
    cfg_enc = one_peace_encoder_config(**kw)
    for k in ("magneto_scale_attn", "scale_attn", "scale_fc", "scale_heads", "use_layer_scale", "checkpoint_activations"):
        v = _cfg_get(enc, k)
        if v is not None:
            setattr(cfg_enc, k, v)
    for adapter in ("text_adapter", "image_adapter", "audio_adapter"):
        sub = _cfg_get(enc, adapter)
        if sub is not None:
            tgt = getattr(cfg_enc, adapter)
            for f in vars(tgt):
                v = _cfg_get(sub, f)
                if v is not None:
                    setattr(tgt, f, v)
    cfg_enc.image_adapter.rel_bucket_size = patch_image_size // 16
    cfg = SimpleNamespace(encoder=cfg_enc, copy_rel_pos_table=bool(_cfg_get(model_cfg, "copy_rel_pos_table", False)))
    return OnePeaceRetrievalModel(cfg, _Dictionary(vocab_size), head_type)


def from_pretrained(model_name_or_path, model_type="one_peace_retrieval", device="cuda", dtype="float32",
                    download_root=None, head_type="vl", patch_image_size=256):
    """hub_interface.py:53-73.  `model_name_or_path` must be a local checkpoint file ({"cfg": {"model": ...}, "model":
    state_dict}); downloading by name needs network access and the reference's URL table."""
    ckpt = torch.load(model_name_or_path, map_location="cpu", weights_only=False)
    model_cfg = _cfg_get(_cfg_get(ckpt, "cfg"), "model")
    model = build_from_checkpoint_cfg(model_cfg, head_type=head_type, patch_image_size=patch_image_size)
    model.load_state_dict(ckpt["model"], strict=True)
    return OnePeaceHubInterface(model, device=device, dtype=dtype)


class OnePeaceHubInterface:
    def __init__(self, model, device="cuda", dtype="float32"):
        self.model = model.to(device).eval()
        self.device = device
        self.dtype = _DTYPES[dtype]
        self.dict = model.src_dict
        if self.dtype != torch.float32:  # hub_interface.py:107-114
            self.model.to(self.dtype)

    def cast_data_dtype(self, t):
        return t.to(self.dtype) if t.is_floating_point() else t

    # ---- collation of already pre-processed inputs -------------------------------------------------------------
    def process_text(self, token_id_lists):
        """list of 1-D LongTensors (BPE ids incl. EOS) -> right-padded [B, T] batch (collate_tokens semantics)."""
        T = max(len(t) for t in token_id_lists)
        out = torch.full((len(token_id_lists), T), self.dict.pad(), dtype=torch.long)
        for i, t in enumerate(token_id_lists):
            out[i, : len(t)] = t
        return out.to(self.device)

    def process_image(self, images):
        return self.cast_data_dtype(torch.as_tensor(images).to(self.device))

    def _feature_encoder_spec(self):
        """The conv stack [(dim, kernel, stride), ...] of the model's audio adapter (hub_interface.py:116-118)."""
        cfg = getattr(getattr(self.model, "cfg", None), "encoder", None)
        spec = getattr(getattr(cfg, "audio_adapter", None), "feature_encoder_spec", None)
        return eval(spec) if isinstance(spec, str) else (spec or [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2)

    def _frames(self, n):
        """hub_interface.py:124-132 (_get_mask_indices_dims with padding 0, dilation 1)."""
        for _, k, s in self._feature_encoder_spec():
            n = 1 + (n - (k - 1) - 1) // s
        return n

    def process_audio(self, wav_list, sample_rate=16000):
        """list of 1-D float waveforms @16 kHz -> (src_audios [B, T], audio_padding_masks [B, frames+1]), as
        hub_interface.py:170-193 after librosa.load: per-waveform layer norm, crop to 15 s, tile up to 1 s, an all-False
        frame mask of each clip's OWN length, then right-padding of waveforms with 0 and of masks with True."""
        feats, masks = [], []
        for w in wav_list:
            w = torch.as_tensor(w, dtype=torch.float32)
            w = F.layer_norm(w, w.shape)
            if w.numel() > sample_rate * 15:
                w = w[: sample_rate * 15]
            if w.numel() < sample_rate:
                w = w.repeat(math.ceil(sample_rate / w.numel()))[:sample_rate]
            feats.append(w)
            masks.append(torch.zeros(self._frames(w.numel()) + 1, dtype=torch.bool))
        T, Fm = max(w.numel() for w in feats), max(m.numel() for m in masks)
        wavs = torch.zeros(len(feats), T)
        pad = torch.ones(len(feats), Fm, dtype=torch.bool)
        for i, (w, m) in enumerate(zip(feats, masks)):
            wavs[i, : w.numel()] = w
            pad[i, : m.numel()] = m
        return self.cast_data_dtype(wavs.to(self.device)), pad.to(self.device)

    # ---- hipGraph replay of the extract_* calls (MI355X serving path; no reference counterpart) -----------------------
    def enable_graphs(self, on=True):
        """Capture each extract_* call into one hipGraph per input shape (one_peace_amd/graphs.py) and replay it: at batch
        1-8 the 40-layer forward is bound by the host's launch path.  Weights are captured by address: call again after
        loading new weights or moving the model."""
        self._graphs = {} if on else None
        return self

    def _run(self, tag, **kw):
        graphs = getattr(self, "_graphs", None)
        if graphs is None or not all(v.is_cuda for v in kw.values() if torch.is_tensor(v)):
            return self._eager(tag, **kw)
        if tag not in graphs:
            from ..graphs import GraphCache
            graphs[tag] = GraphCache(lambda _tag=tag, **k: self._eager(_tag, **k))
        return graphs[tag](**kw)

    def _eager(self, tag, **kw):
        if tag == "vl":
            tf, _, _ = self.model.encoder_wrapper(encoder_type="vl", **kw)
            return tf[:, 0, :]
        return self.model(encoder_type=tag, **kw)

    # ---- hub_interface.py:206-225 ---------------------------------------------------------------------------------
    @torch.no_grad()
    def extract_text_features(self, src_tokens):
        return self._run("text", src_tokens=src_tokens)

    @torch.no_grad()
    def extract_image_features(self, src_images):
        return self._run("image", src_images=self.cast_data_dtype(src_images))

    @torch.no_grad()
    def extract_audio_features(self, src_audios, audio_padding_masks):
        return self._run("audio", src_audios=self.cast_data_dtype(src_audios), audio_padding_masks=audio_padding_masks)

    @torch.no_grad()
    def extract_vl_features(self, src_images, src_tokens):
        return self._run("vl", src_tokens=src_tokens, src_images=self.cast_data_dtype(src_images))
