"""Pins oracle/onepeace_oracle.py (the CPU restatement) to outputs of the UNMODIFIED reference.

tests/golden/*.pt were produced by tests/golden/make_golden.py running the reference sources on CPU.
fp32 tolerance: 2e-5 absolute on O(1) values (different summation order only).
"""
import os

import pytest
import torch

from oracle import onepeace_oracle as O
from oracle import synth

ATOL = 2e-5


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def _micro_sd(fx):
    sd = synth.synth_state_dict(fx["shapes"])
    p = "encoder_wrapper."
    sd[p + "text_adapter.rp_bucket"] = O.token_bucket_position(fx["cfg"]["text_bucket_size"])
    sd[p + "audio_adapter.rp_bucket"] = O.token_bucket_position(fx["cfg"]["audio_bucket_size"])
    rb = fx["cfg"]["image_rel_bucket_size"]
    sd[p + "image_adapter.rp_bucket"] = O.image_bucket_position(rb, (2 * rb - 1) ** 2 + 3)
    return sd


def test_bucket_tables(golden_dir):
    fx = _load(golden_dir, "micro_retrieval.pt")
    sd = _micro_sd(fx)
    assert torch.equal(sd["encoder_wrapper.image_adapter.rp_bucket"], fx["image_rp_bucket"])
    tb = sd["encoder_wrapper.text_adapter.rp_bucket"]
    assert tb.sum() == fx["text_rp_bucket_sum"]
    assert torch.equal(tb[:40, :40], fx["text_rp_bucket_corner"])
    ab = sd["encoder_wrapper.audio_adapter.rp_bucket"]
    assert ab.sum() == fx["audio_rp_bucket_sum"]
    assert torch.equal(ab[700], fx["audio_rp_bucket_row"])


def test_micro_embeddings_losses_and_grads(golden_dir):
    fx = _load(golden_dir, "micro_retrieval.pt")
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in _micro_sd(fx).items()}
    inp = fx["inputs"]
    heads, L = fx["cfg"]["attention_heads"], fx["cfg"]["layers"]
    t, tf = O.contrastive_embed(sd, heads, L, "text", src_tokens=inp["src_tokens"])
    i, imf = O.contrastive_embed(sd, heads, L, "image", src_images=inp["src_images"])
    a, _ = O.contrastive_embed(sd, heads, L, "audio", src_audios=inp["src_audios"],
                               audio_padding_masks=inp["audio_padding_masks"])
    for got, key in ((t, "text_logits"), (i, "image_logits"), (a, "audio_logits"), (tf, "text_feats"),
                     (imf, "image_feats")):
        assert torch.allclose(got, fx[key], atol=ATOL, rtol=1e-4), key
    scale = O.logit_scale_exp(sd["logit_scale"])
    l_it, i2t, t2i = O.itc_loss(i, t, i, t, scale)
    l_at, a2t, t2a = O.itc_loss(a, t, a, t, scale, label_smoothing=0.1)
    assert torch.allclose(l_it, fx["itc_loss"], atol=ATOL)
    assert torch.allclose(l_at, fx["atc_loss"], atol=ATOL)
    assert (i2t, t2i, a2t, t2a) == (fx["i2t"], fx["t2i"], fx["a2t"], fx["t2a"])
    (l_it + l_at).backward()
    checked = 0
    for k, g in fx["grads"].items():
        if k.endswith("#norm"):
            n = k[:-5]
            got = sd[n].grad.double().norm().float()
            assert torch.allclose(got, g, rtol=2e-4, atol=1e-6), k
        elif k.endswith("#rows4"):
            assert torch.allclose(sd[k[:-6]].grad[:4], g, atol=ATOL, rtol=2e-4), k
        else:
            assert torch.allclose(sd[k].grad, g, atol=ATOL, rtol=2e-4), k
        checked += 1
    assert checked > 50


def test_micro_joint_streams(golden_dir):
    fx = _load(golden_dir, "micro_retrieval.pt")
    sd = _micro_sd(fx)
    inp = fx["inputs"]
    heads, L = fx["cfg"]["attention_heads"], fx["cfg"]["layers"]
    vl, _ = O.model_wrapper_forward(sd, "encoder_wrapper", heads, L, "vl", src_tokens=inp["src_tokens"],
                                    src_images=inp["src_images"])
    al, _ = O.model_wrapper_forward(sd, "encoder_wrapper", heads, L, "al", src_tokens=inp["src_tokens"],
                                    src_audios=inp["src_audios"], audio_padding_masks=inp["audio_padding_masks"])
    assert torch.allclose(vl["text"], fx["vl_text"], atol=ATOL, rtol=1e-4)
    assert torch.allclose(vl["image"], fx["vl_image"], atol=ATOL, rtol=1e-4)
    assert torch.allclose(al["text"], fx["al_text"], atol=ATOL, rtol=1e-4)
    assert torch.allclose(al["audio"], fx["al_audio"], atol=ATOL, rtol=1e-4)


def test_tiny_text_config1(golden_dir):
    """BASELINE.json configs[0]: H=256, L=4 text-only extract_text_features, bs=8, seq=64."""
    fx = _load(golden_dir, "tiny_text.pt")
    sd = synth.synth_state_dict(fx["shapes"])
    sd["encoder_wrapper.text_adapter.rp_bucket"] = O.token_bucket_position(256)
    out, _ = O.contrastive_embed(sd, 4, 4, "text", src_tokens=fx["inputs"]["src_tokens"])
    assert out.shape == (8, 256)
    assert torch.allclose(out, fx["text_logits"], atol=ATOL, rtol=1e-4)


@pytest.mark.parametrize("et", ["text", "image", "audio"])
def test_layer_forward_backward(golden_dir, et):
    fx = _load(golden_dir, "layer.pt")
    sd = {"L." + k: v.clone().requires_grad_(True) for k, v in synth.synth_state_dict(fx["shapes"]).items()}
    x = fx["x"].clone().requires_grad_(True)
    y = O.encoder_layer(x, sd, "L", fx["heads"], et, fx["bias"])
    ref = fx["out"][et]
    assert torch.allclose(y, ref["y"], atol=ATOL, rtol=1e-4)
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    (y * w).sum().backward()
    assert torch.allclose(x.grad, ref["dx"], atol=1e-4, rtol=2e-4)
    for k, g in ref["grads"].items():
        if k.endswith("#norm"):
            assert torch.allclose(sd["L." + k[:-5]].grad.double().norm().float(), g, rtol=2e-4), k
        elif k.endswith("#rows4"):
            assert torch.allclose(sd["L." + k[:-6]].grad[:4], g, atol=1e-4, rtol=2e-4), k
        else:
            assert torch.allclose(sd["L." + k].grad, g, atol=1e-4, rtol=2e-4), k


def test_adamw_matches_reference_formula():
    """one_peace/optim/adam.py:186-253 restated; cross-check against torch.optim.AdamW semantics is NOT
    valid (the reference adds eps to sqrt(v) before bias correction), so check a hand computation."""
    p = torch.tensor([1.0, -2.0]); g = torch.tensor([0.5, 0.25])
    m = torch.zeros(2); v = torch.zeros(2)
    O.adamw_step(p, g, m, v, step=1, lr=0.1, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01)
    m1 = 0.1 * g; v1 = 0.001 * g * g
    step_size = 0.1 * (1 - 0.999) ** 0.5 / (1 - 0.9)
    want = torch.tensor([1.0, -2.0]) * (1 - 0.01 * 0.1) - step_size * m1 / (v1.sqrt() + 1e-8)
    assert torch.allclose(p, want, atol=1e-6)


def test_optimizer_leg_matches_reference_adam_clip_and_layer_decay(golden_dir):
    """tests/golden/optim.pt: the reference's Adam (optim/adam.py:124-253) + clip_grad_norm_ (fairseq/utils.py:349-398) +
    layer-decay param groups (utils/layer_decay.py, trainer.py:265-278) executed through ref_shim on the bf16 micro model for
    three steps.  The oracle restatement must reproduce group assignment, gradient norms and every parameter bit for bit."""
    from tests.model_util import build_retrieval
    fx = _load(golden_dir, "optim.pt")
    cfg, oc = fx["cfg"], fx["optim"]
    model = build_retrieval(dict(cfg), fx["vocab"])  # the mirror: same parameter names / no_weight_decay() as the reference
    sd = synth.synth_state_dict(fx["shapes"])
    params = {n: sd[n].to(torch.bfloat16) for n, _ in model.named_parameters()}
    groups = O.param_groups(list(params.items()), oc["weight_decay"], model.no_weight_decay(), cfg["layers"], oc["layer_decay"])
    assert set(groups) == set(fx["assign"])
    for n, (scale, wd) in fx["assign"].items():
        assert abs(groups[n][0] - scale) < 1e-12 and groups[n][1] == wd, n
    state = {n: (torch.zeros(p.shape), torch.zeros(p.shape)) for n, p in params.items()}
    for step, lr in enumerate(oc["lr"], start=1):
        grads = {n: synth.optim_grad(n, p.shape, step) for n, p in params.items()}
        total = O.optimizer_step(params, grads, state, groups, step, lr, oc["betas"], oc["eps"], oc["clip_norm"])
        assert torch.allclose(total.float(), fx["grad_norms"][step - 1], rtol=1e-6)
        snap = fx["params_after"][step - 1]
        for n, p in params.items():
            assert torch.allclose(p.double().norm().float(), snap[n + "#norm"], rtol=1e-6), (step, n)
            if n in snap:
                assert torch.equal(p.reshape(snap[n].shape), snap[n]), (step, n)
            else:
                assert torch.equal(p.reshape(-1)[:2048], snap[n + "#head"]), (step, n)


def test_deep_vision_branch_against_reference(golden_dir):
    """tests/golden/deep_vision.pt (reference, H=1536 / F=6144 / 24 heads, 8 layers, 2 images): the oracle reproduces the
    embeddings and feature rows; forward only (the backward of 302 M parameters is covered at 2-4 layers by the other fixtures
    and would double this test's two minutes)."""
    fx = _load(golden_dir, "deep_vision.pt")
    sd = synth.synth_state_dict(fx["shapes"])
    rb = fx["cfg"]["image_rel_bucket_size"]
    sd["encoder_wrapper.image_adapter.rp_bucket"] = O.image_bucket_position(rb, (2 * rb - 1) ** 2 + 3)
    imgs = synth.synth_inputs(fx["batch"], image_res=fx["image_res"], vocab=fx["vocab"])["src_images"]
    with torch.no_grad():
        feats = O.model_wrapper_forward(sd, "encoder_wrapper", fx["cfg"]["attention_heads"], fx["cfg"]["layers"], "image",
                                        src_images=imgs)[0]["image"]
        logits = O.l2_normalize(O.linear(feats[:, 0], sd["image_proj.weight"], sd["image_proj.bias"]))
    assert torch.allclose(feats[:, :4], fx["feats_head"], atol=2e-4, rtol=1e-4)
    assert torch.allclose(logits, fx["logits"], atol=ATOL, rtol=1e-4)


def test_deep_text_and_audio_towers_with_backward_against_reference(golden_dir):
    """tests/golden/deep_text_audio.pt (round 4; reference, H=1536 / F=6144 / 24 heads, 8 layers, 3 captions + 3 two-second clips):
    the oracle reproduces both embeddings and the feature rows, and -- the first DEEP fixture with a backward -- every parameter
    gradient norm of loss = sum(text_logits * w_t) + sum(audio_logits * w_a), the small gradients in full and row probes of the
    large ones in the first and last layer."""
    fx = _load(golden_dir, "deep_text_audio.pt")
    cfg = fx["cfg"]
    sd = synth.synth_state_dict(fx["shapes"])
    sd["encoder_wrapper.text_adapter.rp_bucket"] = O.token_bucket_position(cfg["text_bucket_size"])
    sd["encoder_wrapper.audio_adapter.rp_bucket"] = O.token_bucket_position(cfg["audio_bucket_size"])
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    inp = synth.synth_inputs(fx["batch"], text_len=fx["text_len"], audio_samples=fx["audio_samples"], vocab=fx["vocab"])
    heads, L = cfg["attention_heads"], cfg["layers"]
    t, tf = O.contrastive_embed(sd, heads, L, "text", src_tokens=inp["src_tokens"])
    a, af = O.contrastive_embed(sd, heads, L, "audio", src_audios=inp["src_audios"], audio_padding_masks=inp["audio_padding_masks"])
    assert torch.allclose(t, fx["text_logits"], atol=ATOL, rtol=1e-4) and torch.allclose(a, fx["audio_logits"], atol=ATOL, rtol=1e-4)
    assert torch.allclose(tf[:, :4], fx["text_feats_head"], atol=2e-4, rtol=1e-4)
    assert torch.allclose(af[:, :4], fx["audio_feats_head"], atol=2e-4, rtol=1e-4)
    wt = synth.synth_tensor("deep_ta/wt", t.shape, seed=6)
    wa = synth.synth_tensor("deep_ta/wa", a.shape, seed=7)
    ((t * wt).sum() + (a * wa).sum()).backward()
    checked = 0
    for k, g in fx["grads"].items():
        if k.endswith("#norm"):
            got = sd[k[:-5]].grad.double().norm().float()
            assert torch.allclose(got, g, rtol=5e-4, atol=1e-6), (k, float(got), float(g))
        elif k.endswith("#rows4"):
            assert torch.allclose(sd[k[:-6]].grad[:4], g, atol=1e-4, rtol=5e-4), k
        else:
            assert torch.allclose(sd[k].grad, g, atol=1e-4, rtol=5e-4), k
        checked += 1
    assert checked > 300


def test_layerdrop_and_all_hiddens_against_reference(golden_dir):
    """transformer_encoder.py:48-51,186-199 + fairseq/modules/layer_drop.py:13-44 on the unmodified reference (make_golden.py
    layerdrop_fixture): the oracle draws the same layerdrop mask from the seeded CPU generator, runs the same layers and returns the same
    per-layer states, for a joint text+image stream and a text-only stream, in eval and training mode."""
    fx = _load(golden_dir, "layerdrop_hiddens.pt")
    sd = _micro_sd(fx)
    heads, L = fx["cfg"]["attention_heads"], fx["cfg"]["layers"]
    inp = fx["inputs"]
    p = "encoder_wrapper"
    ti = O.text_adapter(sd, p + ".text_adapter", inp["src_tokens"])
    ii = O.image_adapter(sd, p + ".image_adapter", inp["src_images"])
    n_states = 0
    for case in fx["cases"]:
        torch.manual_seed(case["seed"])
        mask = O.layerdrop_mask(L, fx["layerdrop"], training=case["train"])
        assert mask == case["ran"], (case["seed"], mask, case["ran"])
        for et, infos in (("vl", (ti, ii)), ("text", (ti, None))):
            x, _, states = O.encoder_forward(sd, p + ".fusion_model", heads, L, et, infos[0], infos[1], None, layer_mask=mask,
                                             return_all_hiddens=True)
            want = case[et]
            assert torch.allclose(x.transpose(0, 1), want["encoder_out"], atol=ATOL, rtol=1e-4), (case["seed"], et)
            assert len(states["text"]) == len(want["text_states"]) == sum(mask) and states["audio"] == []
            assert len(states["image"]) == len(want["image_states"]) == (sum(mask) if et == "vl" else 0)
            for got, ref in zip(states["text"] + states["image"], want["text_states"] + want["image_states"]):
                assert got.shape == ref.shape and torch.allclose(got, ref, atol=ATOL, rtol=1e-4), (case["seed"], et)
                n_states += 1
    assert n_states == 3 * sum(sum(c["ran"]) for c in fx["cases"])
