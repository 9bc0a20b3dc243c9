"""TEST INFRASTRUCTURE ONLY -- deterministic synthetic weights / inputs shared by the golden-vector
generator, the parity tests and the CPU-baseline leg of bench.py.

Weights are a pure function of (tensor name, shape, seed) through numpy's PCG64 stream, so a fixture
only has to store inputs and expected outputs, never the weights.  Values are chosen so that every
term of the path is *visible* in the outputs: the reference zero-initialises the relative-position
tables and sets layer-scale gamma to 1e-6 (pretrain_vl_3B.yaml:130), which would hide bias and
residual-branch bugs behind the residual stream (SURVEY.md 8d "Synthetic inputs").
"""
import zlib

import numpy as np
import torch

SEED = 3407  # the reference's own seed, run_scripts/pretrain/pretrain_vl_3B.yaml:75


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def synth_tensor(name, shape, seed=SEED):
    r = _rng(name, seed)
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    z = r.standard_normal(n).astype(np.float32).reshape(shape)
    leaf = name.split(".")[-1]
    if "gamma_" in name:                       # layer-scale: 0.1 .. 1.0
        v = 0.1 + 0.9 * r.random(n).astype(np.float32).reshape(shape)
    elif name == "logit_scale":
        v = np.full(shape, np.log(1 / 0.07), dtype=np.float32)
    elif "rel_pos_table_list" in name:         # visible relative-position bias
        v = 0.5 * z
    elif leaf == "weight" and len(shape) == 1:  # LayerNorm gain
        v = 1.0 + 0.1 * z
    elif leaf == "bias":
        v = 0.02 * z
    elif leaf == "weight" and len(shape) >= 2:  # Linear / Conv / Embedding: ~unit-variance outputs
        fan_in = int(np.prod(shape[1:]))
        if "embed_tokens" in name or "embed_positions.weight" in name:
            v = 0.5 * z
        else:
            v = z / np.sqrt(fan_in)
    else:                                      # cls_embedding, pos_embed, cls_pos_embed, mask tokens
        v = 0.5 * z
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


NON_SYNTH = ("rp_bucket", "position_idx", "version")


def synth_state_dict(shapes, seed=SEED):
    """shapes: {name: shape}; integer buffers (rp_bucket, position_idx, version) are skipped."""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items() if k.split(".")[-1] not in NON_SYNTH}


def synth_inputs(batch, text_len=None, image_res=None, audio_samples=None, vocab=50265, pad_idx=1,
                 seed=SEED, pad_pattern=True):
    """tokens uniform in [4, vocab), row r gets (r mod 8) trailing pads; images and
    waveforms ~ N(0, 1); audio has no padding."""
    out = {}
    r = _rng("inputs", seed)
    if text_len is not None:
        tok = r.integers(4, vocab, size=(batch, text_len)).astype(np.int64)
        if pad_pattern:
            for i in range(batch):
                k = i % 8
                if k:
                    tok[i, text_len - k:] = pad_idx
        out["src_tokens"] = torch.from_numpy(tok)
    if image_res is not None:
        out["src_images"] = torch.from_numpy(
            r.standard_normal((batch, 3, image_res, image_res)).astype(np.float32))
    if audio_samples is not None:
        out["src_audios"] = torch.from_numpy(r.standard_normal((batch, audio_samples)).astype(np.float32))
        frames = audio_frames(audio_samples)
        out["audio_padding_masks"] = torch.zeros(batch, frames + 1, dtype=torch.bool)
    return out


def audio_frames(n_samples):
    """Output length of the 7-layer conv stack (adapter/audio.py:254-311): k/stride (10,5),(3,2)x4,(2,2)x2."""
    n = n_samples
    for k, s in [(10, 5)] + [(3, 2)] * 4 + [(2, 2)] * 2:
        n = (n - k) // s + 1
    return n


def optim_grad(name, shape, step):
    """Deterministic synthetic gradient of parameter `name` at optimiser step `step` (bf16), shared by the golden generator
    (tests/golden/make_golden.py::optim_fixture) and the optimiser parity tests."""
    g = synth_tensor("grad%d/%s" % (step, name), shape, seed=99)
    return (g * 0.05).to(torch.bfloat16).reshape(tuple(shape))
