"""Host-logic tests on CPU: the mirrors of the reference modules (their torch-op path) must reproduce the
reference-generated golden fixtures with the reference's own state-dict keys -- this is the drop-in contract
 checked without a GPU."""
import os

import pytest
import torch

from tests.model_util import build_retrieval, load_synth

ATOL = 3e-5


def _fx(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


@pytest.fixture(scope="module")
def micro(golden_dir):
    fx = _fx(golden_dir, "micro_retrieval.pt")
    m = load_synth(build_retrieval(fx["cfg"], fx["vocab"]), fx["shapes"]).eval()
    return fx, m


def test_state_dict_contract_and_buckets(micro):
    fx, m = micro
    assert torch.equal(m.encoder_wrapper.image_adapter.rp_bucket, fx["image_rp_bucket"])
    assert m.encoder_wrapper.text_adapter.rp_bucket.sum() == fx["text_rp_bucket_sum"]
    assert torch.equal(m.encoder_wrapper.text_adapter.rp_bucket[:40, :40], fx["text_rp_bucket_corner"])
    assert torch.equal(m.encoder_wrapper.audio_adapter.rp_bucket[700], fx["audio_rp_bucket_row"])


def test_micro_forward_losses_grads(micro):
    from one_peace_amd.criterions.contrastive import AudioTextRetrievalCriterion, ImageTextRetrievalCriterion
    fx, m = micro
    inp = fx["inputs"]
    t = m(src_tokens=inp["src_tokens"], encoder_type="text")
    i = m(src_images=inp["src_images"], encoder_type="image")
    a = m(src_audios=inp["src_audios"], audio_padding_masks=inp["audio_padding_masks"], encoder_type="audio")
    for got, key in ((t, "text_logits"), (i, "image_logits"), (a, "audio_logits")):
        assert torch.allclose(got, fx[key], atol=ATOL, rtol=1e-4), key
    scale = m(return_logit_scale=True)
    itc = ImageTextRetrievalCriterion(None, 0.0)
    atc = AudioTextRetrievalCriterion(None, 0.1)
    l1, i2t, t2i = itc.compute_itc_loss(i, t, i.data, t.data, scale)
    l2, a2t, t2a = atc.compute_atc_loss(a, t, a.data, t.data, scale)
    assert torch.allclose(l1, fx["itc_loss"], atol=ATOL) and torch.allclose(l2, fx["atc_loss"], atol=ATOL)
    assert (i2t, t2i, a2t, t2a) == (fx["i2t"], fx["t2i"], fx["a2t"], fx["t2a"])
    m.zero_grad()
    (l1 + l2).backward()
    params = dict(m.named_parameters())
    n = 0
    for k, g in fx["grads"].items():
        if k.endswith("#norm"):
            assert torch.allclose(params[k[:-5]].grad.double().norm().float(), g, rtol=3e-4, atol=1e-6), k
        elif k.endswith("#rows4"):
            assert torch.allclose(params[k[:-6]].grad[:4], g, atol=ATOL, rtol=3e-4), k
        else:
            assert torch.allclose(params[k].grad, g, atol=ATOL, rtol=3e-4), k
        n += 1
    assert n > 50


def test_micro_criterion_forward(micro):
    from one_peace_amd.criterions.contrastive import ImageTextRetrievalCriterion, TriModalContrastiveCriterion
    fx, m = micro
    sample = {"net_input": dict(fx["inputs"]), "nsentences": 4}
    loss, ss, log = ImageTextRetrievalCriterion(None, 0.0)(m, sample)
    assert ss == 1 and torch.allclose(loss, fx["itc_loss"], atol=ATOL)
    loss3, _, log3 = TriModalContrastiveCriterion(None, 0.0)(m, sample)
    assert torch.allclose(log3["itc_loss"], fx["itc_loss"], atol=ATOL)


def test_micro_joint_streams(micro):
    fx, m = micro
    inp = fx["inputs"]
    with torch.no_grad():
        vt, vi, _ = m.encoder_wrapper(src_tokens=inp["src_tokens"], src_images=inp["src_images"], encoder_type="vl")
        at, _, aa = m.encoder_wrapper(src_tokens=inp["src_tokens"], src_audios=inp["src_audios"],
                                      audio_padding_masks=inp["audio_padding_masks"], encoder_type="al")
    assert torch.allclose(vt, fx["vl_text"], atol=ATOL, rtol=1e-4) and torch.allclose(vi, fx["vl_image"], atol=ATOL, rtol=1e-4)
    assert torch.allclose(at, fx["al_text"], atol=ATOL, rtol=1e-4) and torch.allclose(aa, fx["al_audio"], atol=ATOL, rtol=1e-4)


def test_tiny_text_config1(golden_dir):
    fx = _fx(golden_dir, "tiny_text.pt")
    cfg = {k: v for k, v in fx["cfg"].items()}
    m = load_synth(build_retrieval(cfg, 50265, head_type="text"), fx["shapes"]).eval()
    with torch.no_grad():
        out = m(src_tokens=fx["inputs"]["src_tokens"], encoder_type="text")
    assert torch.allclose(out, fx["text_logits"], atol=ATOL, rtol=1e-4)


def test_pretrain_model_contrastive_branch_and_registry():
    from types import SimpleNamespace
    from one_peace_amd.one_peace.one_peace_pretrain import OnePeacePretrainModel
    from one_peace_amd.registry import MODEL_REGISTRY
    from one_peace_amd.unify_model_config import one_peace_encoder_config
    from tests.model_util import TinyDictionary
    assert MODEL_REGISTRY["one_peace_pretrain"] is OnePeacePretrainModel
    enc = one_peace_encoder_config(embed_dim=128, ffn_embed_dim=256, layers=2, attention_heads=2, drop_path_rate=0.0,
                                   image_rel_bucket_size=4, use_audio_moe=False)
    dec = one_peace_encoder_config(embed_dim=64, ffn_embed_dim=128, layers=1, attention_heads=1, drop_path_rate=0.0,
                                   use_audio_moe=False)
    dec.text_adapter.use_attn_bias = dec.image_adapter.use_attn_bias = False
    dec.image_adapter.vision_encoder_type = "none"
    cfg = SimpleNamespace(encoder=enc, decoder=dec, reset_logit_scale=False, logit_scale_init=1 / 0.07, stage2_pretrain=False)
    m = OnePeacePretrainModel(cfg, TinyDictionary(1000)).eval()
    tok = torch.randint(4, 1000, (2, 9))
    logits, feats = m(src_tokens=tok, encoder_type="text")
    assert logits.shape == (2, 128) and feats.shape == (2, 10, 128)
    assert torch.allclose(logits.norm(dim=1), torch.ones(2), atol=1e-5)
    keys = set(m.state_dict())
    for k in ("decoder_text_embed.weight", "text_mask_token", "image_mask_head.bias", "logit_scale",
              "decoder_wrapper.fusion_model.layers.0.gamma_1"):
        assert k in keys
    # masked (DCL) branch keeps working through the torch-op path
    ids = torch.tensor([[0, 1, 3, 5, -1], [0, 2, 4, -1, -1]])
    dt, di, da = m(src_tokens=tok, text_preserve_ids=ids, encoder_type="text")
    assert dt.shape == (2, 10, 128) and di is None and da is None


def test_hub_interface_from_pretrained_roundtrip(golden_dir, tmp_path):
    """from_pretrained / extract_text_features (BASELINE config 1 surface) on a checkpoint in the reference's format."""
    from one_peace_amd.one_peace import hub_interface
    fx = _fx(golden_dir, "tiny_text.pt")
    m = load_synth(build_retrieval(dict(fx["cfg"]), 50265, head_type="text"), fx["shapes"])
    ckpt = {"cfg": {"model": {"encoder": {"embed_dim": 256, "ffn_embed_dim": 1024, "layers": 4, "attention_heads": 4,
                                          "drop_path_rate": 0.0, "layer_scale_init_value": 1e-2}}},
            "model": m.state_dict()}
    path = tmp_path / "tiny.pt"
    torch.save(ckpt, path)
    hub = hub_interface.from_pretrained(str(path), device="cpu", dtype="float32", head_type="text")
    toks = hub.process_text([row[row != 1] for row in fx["inputs"]["src_tokens"]])
    assert torch.equal(toks.cpu(), fx["inputs"]["src_tokens"])
    out = hub.extract_text_features(toks)
    assert torch.allclose(out, fx["text_logits"], atol=ATOL, rtol=1e-4)


def test_hub_interface_all_modalities_and_graph_switch_on_cpu(golden_dir):
    """extract_{text,image,audio,vl}_features of the hub mirror reproduce the reference's outputs (fp32, CPU torch path);
    enable_graphs() is a device feature: CPU tensors keep running eagerly, and a GraphedCall over CPU tensors fails loudly."""
    from one_peace_amd.graphs import GraphedCall
    from one_peace_amd.one_peace.hub_interface import OnePeaceHubInterface
    fx = _fx(golden_dir, "micro_retrieval.pt")
    hub = OnePeaceHubInterface(load_synth(build_retrieval(fx["cfg"], fx["vocab"]), fx["shapes"]), device="cpu", dtype="float32")
    inp = fx["inputs"]
    toks = hub.process_text([row[row != 1] for row in inp["src_tokens"]])
    refs = {"text": fx["text_logits"], "image": fx["image_logits"], "audio": fx["audio_logits"], "vl": fx["vl_text"][:, 0]}
    for graphs in (False, True):
        hub.enable_graphs(graphs)
        outs = {"text": hub.extract_text_features(toks),
                "image": hub.extract_image_features(hub.process_image(inp["src_images"])),
                "audio": hub.extract_audio_features(inp["src_audios"], inp["audio_padding_masks"]),
                "vl": hub.extract_vl_features(inp["src_images"], toks)}
        for k, ref in refs.items():
            assert torch.allclose(outs[k], ref, atol=ATOL, rtol=1e-4), k
    with pytest.raises(RuntimeError, match="device tensors"):
        GraphedCall(lambda x: x + 1, {"x": torch.zeros(2)})


def test_hub_process_audio_follows_reference_collation(golden_dir):
    """hub_interface.py:170-193: layer-normed waveforms, 15 s crop, tiling up to 1 s, per-clip all-False frame masks of the
    clip's own frame count (from the model's feature_encoder_spec), waveforms right-padded with 0 and masks with True."""
    from one_peace_amd.one_peace.hub_interface import OnePeaceHubInterface
    fx = _fx(golden_dir, "micro_retrieval.pt")
    hub = OnePeaceHubInterface(load_synth(build_retrieval(fx["cfg"], fx["vocab"]), fx["shapes"]), device="cpu", dtype="float32")
    spec = hub._feature_encoder_spec()

    def frames(n):
        for _, k, s in spec:
            n = (n - k) // s + 1
        return n
    g = torch.Generator().manual_seed(0)
    clips = [torch.randn(n, generator=g) * 3 + 1 for n in (5000, 80000, 16000 * 16)]
    wavs, masks = hub.process_audio(clips)
    assert wavs.shape == (3, 16000 * 15) and masks.shape == (3, frames(16000 * 15) + 1)
    lens = [16000, 80000, 16000 * 15]  # tiled to 1 s, kept, cropped to 15 s
    for i, n in enumerate(lens):
        assert not masks[i, : frames(n) + 1].any() and masks[i, frames(n) + 1:].all()
        assert wavs[i, n:].abs().max() == 0 if n < wavs.shape[1] else True
    normed = torch.nn.functional.layer_norm(clips[0], clips[0].shape)
    assert torch.allclose(wavs[0, :16000], normed.repeat(4)[:16000])
    assert torch.allclose(wavs[2], torch.nn.functional.layer_norm(clips[2], clips[2].shape)[: 16000 * 15])


def _build_pretrain(fx, audio_language=False, stage2=False):
    from types import SimpleNamespace
    from one_peace_amd.one_peace.one_peace_pretrain import OnePeacePretrainModel
    from one_peace_amd.unify_model_config import one_peace_encoder_config
    from tests.model_util import TinyDictionary, load_synth
    enc = one_peace_encoder_config(drop_path_rate=0.0, **fx["enc"])
    dec_kw = dict(fx["dec"])
    use_attn_bias, vis = dec_kw.pop("use_attn_bias"), dec_kw.pop("vision_encoder_type", "none")
    dec = one_peace_encoder_config(drop_path_rate=0.0, **dec_kw)
    dec.text_adapter.use_attn_bias = dec.image_adapter.use_attn_bias = dec.audio_adapter.use_attn_bias = use_attn_bias
    dec.image_adapter.vision_encoder_type = vis
    if audio_language:  # pretrain_al_3B.yaml:138-176
        dec.audio_adapter.feature_encoder_spec = None
        dec.audio_adapter.abs_pos_type = "fixed"
        dec.audio_adapter.bucket_size = 256
        dec.use_layer_scale = False
    cfg = SimpleNamespace(encoder=enc, decoder=dec, reset_logit_scale=False, logit_scale_init=1 / 0.07, stage2_pretrain=stage2)
    return load_synth(OnePeacePretrainModel(cfg, TinyDictionary(fx["vocab"])), fx["shapes"]).eval()


def test_full_pretraining_objective_matches_reference(golden_dir):
    """ITC + four masked-token (DCL) terms, six forward passes incl. the per-sample preserve-id gathers and the small
    decoder (image_text_pretrain_loss.py:76-160): losses, student features and gradients of the mirror against the
    fixture written by the unmodified reference (tests/golden/make_golden.py::pretrain_fixture)."""
    from one_peace_amd.criterions.pretrain import ImageTextPretrainLossCriterion
    fx = torch.load(os.path.join(golden_dir, "micro_pretrain.pt"), weights_only=False)
    m = _build_pretrain(fx)
    ni = fx["net_input"]
    with torch.no_grad():
        st = m(src_tokens=ni["src_tokens"], text_preserve_ids=ni["text_preserve_ids"], encoder_type="text")[0]
        svt, svi, _ = m(src_tokens=ni["src_tokens"], text_preserve_ids=ni["vl_text_preserve_ids"], src_images=ni["src_images"],
                        image_preserve_ids=ni["vl_image_preserve_ids"], encoder_type="vl")
    assert torch.allclose(st, fx["student_text"], atol=ATOL, rtol=1e-4)
    assert torch.allclose(svt, fx["student_vl_text"], atol=ATOL, rtol=1e-4) and torch.allclose(svi, fx["student_vl_image"], atol=ATOL, rtol=1e-4)
    crit = ImageTextPretrainLossCriterion(None, 0.5, 1.0, 0.5, 0.5, 2.5, label_smoothing=0.0)
    loss, _, log = crit(m, {"net_input": ni, "nsentences": 4})
    for k, v in fx["log"].items():
        if "loss" in k:
            assert abs(float(log[k]) - float(v)) < 2e-5 * max(1.0, abs(float(v))), (k, float(log[k]), float(v))
    assert float(log["i2t_ncorrect"]) == float(fx["log"]["i2t_ncorrect"])
    m.zero_grad()
    loss.backward()
    named = dict(m.named_parameters())
    checked = 0
    for k, v in fx["grads"].items():
        if k.endswith("#norm"):
            n = k[:-5]
            if named[n].grad is not None:
                assert abs(float(named[n].grad.double().norm()) - float(v)) <= 2e-4 * max(float(v), 1e-3), (n,)
                checked += 1
        elif not k.endswith("#rows4"):
            assert torch.allclose(named[k].grad, v, atol=2e-5, rtol=2e-4), k
    assert checked > 60


def test_audio_language_pretraining_objective_matches_reference(golden_dir):
    """ATC + three DCL terms of the audio-language stage (audio_text_pretrain_loss.py:73-157; frozen text teacher, joint
    'al' teacher, decoder with the fixed-position spec-less audio adapter and no layer scale) against the reference fixture."""
    from one_peace_amd.criterions.pretrain import AudioTextPretrainLossCriterion
    fx = torch.load(os.path.join(golden_dir, "micro_pretrain_al.pt"), weights_only=False)
    m = _build_pretrain(fx, audio_language=True)
    crit = AudioTextPretrainLossCriterion(None, 1.0, 0.5, 0.5, 2.5, label_smoothing=0.0)
    loss, _, log = crit(m, {"net_input": fx["net_input"], "nsentences": 4})
    for k, v in fx["log"].items():
        if "loss" in k:
            assert abs(float(log[k]) - float(v)) < 2e-5 * max(1.0, abs(float(v))), (k, float(log[k]), float(v))
    m.zero_grad()
    loss.backward()
    named = dict(m.named_parameters())
    checked = 0
    for k, v in fx["grads"].items():
        if k.endswith("#norm"):
            n = k[:-5]
            if named[n].grad is not None:
                assert abs(float(named[n].grad.double().norm()) - float(v)) <= 2e-4 * max(float(v), 1e-3), (n,)
                checked += 1
        elif not k.endswith("#rows4"):
            assert torch.allclose(named[k].grad, v, atol=2e-5, rtol=2e-4), k
    assert checked > 40


def test_stage2_audio_language_pretraining_freezes_what_the_reference_freezes(golden_dir):
    """`stage2_pretrain: true` (one_peace_pretrain.py:98-104): text_proj and the whole encoder frozen except the audio adapter,
    audio_layer_norm and every layer's audio_ffn.  tests/golden/micro_pretrain_al_stage2.pt (written by the unmodified reference):
    the same set of parameters is trainable here, the same set receives a gradient, the loss is unchanged and every stored
    gradient matches."""
    from one_peace_amd.criterions.pretrain import AudioTextPretrainLossCriterion
    fx = torch.load(os.path.join(golden_dir, "micro_pretrain_al_stage2.pt"), weights_only=False)
    m = _build_pretrain(fx, audio_language=True, stage2=True)
    assert sorted(n for n, q in m.named_parameters() if q.requires_grad) == fx["trainable"]
    frozen = [n for n, q in m.named_parameters() if not q.requires_grad]
    assert any("self_attn.q_proj" in n for n in frozen) and any("text_ffn" in n for n in frozen) and "text_proj.weight" in frozen
    assert not any("audio_ffn" in n or "audio_adapter" in n or n.startswith("decoder_") for n in frozen)
    crit = AudioTextPretrainLossCriterion(None, 1.0, 0.5, 0.5, 2.5, label_smoothing=0.0)
    loss, _, log = crit(m, {"net_input": fx["net_input"], "nsentences": 4})
    assert abs(float(loss) - float(fx["loss"])) < 2e-5 * abs(float(fx["loss"]))
    m.zero_grad()
    loss.backward()
    named = dict(m.named_parameters())
    assert sorted(n for n, q in named.items() if q.grad is not None) == fx["with_grad"]
    checked = 0
    for k, v in fx["grads"].items():
        if k.endswith("#norm"):
            assert abs(float(named[k[:-5]].grad.double().norm()) - float(v)) <= 2e-4 * max(float(v), 1e-3), k
            checked += 1
        elif not k.endswith("#rows4"):
            assert torch.allclose(named[k].grad, v, atol=2e-5, rtol=2e-4), k
    assert checked > 30


def test_flat_parameter_groups_follow_reference_param_groups(golden_dir):
    """distributed.FlatParameters + optim.reference_param_groups lay the micro model out as the reference's optimiser groups it
    (trainer.py:265-278, utils/layer_decay.py:34-77; tests/golden/optim.pt holds the reference's own assignment): every group
    is one contiguous, 8-aligned range, and parameter views keep their values."""
    from one_peace_amd.distributed import FlatParameters
    from one_peace_amd.optim import reference_param_groups
    fx = _fx(golden_dir, "optim.pt")
    cfg, oc = fx["cfg"], fx["optim"]
    m = load_synth(build_retrieval(dict(cfg), fx["vocab"]), fx["shapes"])
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    no_decay, lr_scale = reference_param_groups(m, cfg["layers"], oc["layer_decay"])
    flat = FlatParameters(m, no_decay=no_decay, lr_scale=lr_scale)
    assert flat.groups[0][0] == 0 and flat.groups[-1][1] == flat.numel
    assert all(a[1] == b[0] and a[1] % 8 == 0 for a, b in zip(flat.groups, flat.groups[1:]))
    seen = {}
    for n, p, o, k in flat.entries:
        grp = [g for g in flat.groups if g[0] <= o < g[1]]
        assert len(grp) == 1 and o + k <= grp[0][1]
        seen[n] = (grp[0][2], oc["weight_decay"] if grp[0][3] else 0.0)
        assert torch.equal(p.detach(), before[n]) and p.data_ptr() == flat.params[o:].data_ptr()
    assert seen.keys() == fx["assign"].keys()
    for n, (scale, wd) in fx["assign"].items():
        assert abs(seen[n][0] - scale) < 1e-12 and seen[n][1] == wd, n
    assert len(flat.groups) == len(set(fx["assign"].values())) == 8
    # without layer decay: the two-range layout (all decayed parameters, then the rest)
    m2 = load_synth(build_retrieval(dict(cfg), fx["vocab"]), fx["shapes"])
    nd, sc = reference_param_groups(m2)
    flat2 = FlatParameters(m2, no_decay=nd, lr_scale=sc)
    assert sc is None and len(flat2.groups) == 2 and flat2.decay_range == flat2.groups[0][:2]


def test_layerdrop_encoder_bookkeeping_does_not_go_through_the_layerdrop_iterator():
    """With layerdrop > 0, iterating `encoder.layers` in training mode DRAWS a mask (fairseq/modules/layer_drop.py:13-44): everything that
    needs every layer -- the stochastic-depth rates of the stack, upgrade_state_dict_named (transformer_encoder.py:238-244), the fused-path
    eligibility checks -- must index the list instead; return_all_hiddens returns one state per layer that ran."""
    from types import SimpleNamespace
    from one_peace_amd.one_peace.one_peace_retrieval import OnePeaceRetrievalModel
    from one_peace_amd.unify_model_config import one_peace_encoder_config
    from oracle import synth
    from tests.model_util import TinyDictionary
    enc = one_peace_encoder_config(embed_dim=128, ffn_embed_dim=256, layers=4, attention_heads=2, image_rel_bucket_size=4,
                                   text_bucket_size=256, audio_bucket_size=512, drop_path_rate=0.2, layer_scale_init_value=1e-1,
                                   checkpoint_activations=False)
    enc.layerdrop = 0.9
    torch.manual_seed(0)
    m = load_synth(OnePeaceRetrievalModel(SimpleNamespace(encoder=enc, copy_rel_pos_table=False), TinyDictionary(1000), "val")).train()
    F = m.encoder_wrapper.fusion_model
    assert len(F.all_layers()) == 4
    for seed in range(4):
        torch.manual_seed(seed)
        scales = F._draw_path_scales(3, torch.device("cpu"))
        assert scales is not None and len(scales) == 4 and scales[0] == (None, None) and scales[3][0].shape == (3,)
    sd = {}
    F.upgrade_state_dict_named(sd, "enc")
    assert all(any(k.startswith("enc.layers.%d." % i) for k in sd) for i in range(4))
    assert F.multi_possible() is False  # (training with layerdrop: one pass per modality, each of which may take the fused layers)
    inp = synth.synth_inputs(2, text_len=15, image_res=64, audio_samples=8000, vocab=1000)
    W = m.encoder_wrapper
    torch.manual_seed(5)  # uniform_(4) > 0.9 ... the mask of this pass
    keep = int((torch.empty(4).uniform_() > 0.9).sum())
    torch.manual_seed(5)
    out = F(W.text_adapter(inp["src_tokens"], None, None, None), None, None, return_all_hiddens=True, encoder_type="text")
    assert len(out["text_encoder_states"]) == keep and out["image_encoder_states"] == []


def _layerdrop_model(fx):
    from types import SimpleNamespace
    from one_peace_amd.one_peace.one_peace_retrieval import OnePeaceRetrievalModel
    from one_peace_amd.unify_model_config import one_peace_encoder_config
    from tests.model_util import TinyDictionary
    enc = one_peace_encoder_config(drop_path_rate=0.0, checkpoint_activations=False, **fx["cfg"])
    enc.layerdrop = fx["layerdrop"]
    torch.manual_seed(0)
    return load_synth(OnePeaceRetrievalModel(SimpleNamespace(encoder=enc, copy_rel_pos_table=False), TinyDictionary(fx["vocab"]), "val"),
                      fx["shapes"])


def test_layerdrop_and_all_hiddens_match_the_reference_fixture(golden_dir):
    """The mirror's torch path against what the UNMODIFIED reference returned (tests/golden/make_golden.py layerdrop_fixture;
    transformer_encoder.py:48-51,186-199, fairseq/modules/layer_drop.py:13-44): same seed of the CPU generator -> the same layers run, the
    same per-layer states (T x B x C slices per modality) and the same encoder output, eval and training, joint and single stream."""
    fx = _fx(golden_dir, "layerdrop_hiddens.pt")
    m = _layerdrop_model(fx)
    W = m.encoder_wrapper
    inp = fx["inputs"]
    for case in fx["cases"]:
        m.train(case["train"])
        with torch.no_grad():
            t = W.text_adapter(inp["src_tokens"], None, None, None)
            i = W.image_adapter(inp["src_images"], None, None, None, False)
            for et, infos in (("vl", (t, i, None)), ("text", (t, None, None))):
                torch.manual_seed(case["seed"])
                out = W.fusion_model(*infos, return_all_hiddens=True, encoder_type=et)
                want = case[et]
                assert torch.allclose(out["encoder_out"][0], want["encoder_out"], atol=ATOL, rtol=1e-4), (case["seed"], et)
                assert len(out["text_encoder_states"]) == sum(case["ran"]) and out["audio_encoder_states"] == []
                assert len(out["image_encoder_states"]) == (sum(case["ran"]) if et == "vl" else 0)
                for got, ref in zip(out["text_encoder_states"] + out["image_encoder_states"], want["text_states"] + want["image_states"]):
                    assert got.shape == ref.shape and torch.allclose(got, ref, atol=ATOL, rtol=1e-4), (case["seed"], et)


def test_prepend_token_promotes_like_torch_cat():
    """adapter/text.py:110-113 (and siblings) concatenate the cls token with torch.cat, which PROMOTES mixed dtypes (an fp32 token in front
    of a bf16 body under autocast gives fp32); the slice-copy form of the mirrors must do the same, values and gradients."""
    from one_peace_amd.adapter.common import prepend_token
    g = torch.Generator().manual_seed(0)
    for td, bd in ((torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)):
        tok = torch.randn(1, 1, 8, generator=g).to(td).requires_grad_(True)
        body = torch.randn(3, 5, 8, generator=g).to(bd).requires_grad_(True)
        got = prepend_token(tok, body)
        want = torch.cat([tok.expand(3, -1, -1), body], dim=1)
        assert got.dtype == want.dtype and torch.equal(got, want), (td, bd, got.dtype, want.dtype)
        w = torch.randn(got.shape, generator=g)
        g1 = torch.autograd.grad((got.float() * w).sum(), (tok, body))
        g2 = torch.autograd.grad((want.float() * w).sum(), (tok, body))
        assert all(a.dtype == b.dtype and torch.equal(a, b) for a, b in zip(g1, g2))
