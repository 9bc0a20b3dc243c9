"""Generate the golden fixtures from the UNMODIFIED reference, run on CPU through oracle/ref_shim.py.

    python tests/golden/make_golden.py        # needs /root/reference; writes tests/golden/*.pt

The reference has no golden vectors of its own for this path, so these files pin the
oracle (and through it the HIP path) to what the reference code itself computes.  Weights are the
deterministic synthetic weights of oracle/synth.py, so only inputs and expected outputs are stored.
"""
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_shim as R  # noqa: E402
from oracle import synth  # noqa: E402

MICRO = dict(embed_dim=128, ffn_embed_dim=256, layers=2, attention_heads=2, image_rel_bucket_size=4,
             text_bucket_size=256, audio_bucket_size=512)
TINY = dict(embed_dim=256, ffn_embed_dim=1024, layers=4, attention_heads=4)  # BASELINE config 1


def build_ref_model(cfg_kw, vocab, head_type="val"):
    rt = R.ref("one_peace.models.one_peace.one_peace_retrieval")
    cfg = R.make_cfg(**cfg_kw)
    torch.manual_seed(0)
    m = rt.OnePeaceRetrievalModel(cfg, R.TinyDictionary(vocab), head_type)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.split(".")[-1] in synth.NON_SYNTH for k in missing), (missing, unexpected)
    m.eval()
    return m, shapes


def grads_summary(model, names):
    out = {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        out[n + "#norm"] = p.grad.double().norm().float()
        if n in names or p.grad.numel() <= 4096:
            out[n] = p.grad.clone()
        elif p.grad.dim() == 2:
            out[n + "#rows4"] = p.grad[:4].clone()
    return out


def micro_fixture():
    vocab = 1000
    m, shapes = build_ref_model(MICRO, vocab)
    B = 4
    inp = synth.synth_inputs(B, text_len=15, image_res=64, audio_samples=8000, vocab=vocab)
    crit = R.ref("one_peace.criterions.image_text_retrieval_loss")
    acrit = R.ref("one_peace.criterions.audio_text_retrieval_loss")
    itc = crit.ImageTextRetrievalCriterion(None, label_smoothing=0.0)
    atc = acrit.AudioTextRetrievalCriterion(None, label_smoothing=0.1)

    t = m(src_tokens=inp["src_tokens"], encoder_type="text")
    i = m(src_images=inp["src_images"], encoder_type="image")
    a = m(src_audios=inp["src_audios"], audio_padding_masks=inp["audio_padding_masks"], encoder_type="audio")
    scale = m(return_logit_scale=True)
    l_it, i2t, t2i = itc.compute_itc_loss(i, t, i.data, t.data, scale)
    l_at, a2t, t2a = atc.compute_atc_loss(a, t, a.data, t.data, scale)
    loss = l_it + l_at
    m.zero_grad()
    loss.backward()
    keep = {"logit_scale", "encoder_wrapper.fusion_model.layers.0.self_attn.q_proj.weight",
            "encoder_wrapper.fusion_model.layers.1.image_ffn.0.wi_0.weight",
            "encoder_wrapper.fusion_model.layers.1.text_ffn.3.weight",
            "encoder_wrapper.fusion_model.layers.0.audio_ffn.0.wi_1.weight"}
    grads = grads_summary(m, keep)

    # joint streams through ModelWrapper (encoder_type vl / al), eval mode
    with torch.no_grad():
        vl_t, vl_i, _ = m.encoder_wrapper(src_tokens=inp["src_tokens"], src_images=inp["src_images"], encoder_type="vl")
        al_t, _, al_a = m.encoder_wrapper(src_tokens=inp["src_tokens"], src_audios=inp["src_audios"],
                                          audio_padding_masks=inp["audio_padding_masks"], encoder_type="al")
        feats_t = m.encoder_wrapper(src_tokens=inp["src_tokens"], encoder_type="text")[0]
        feats_i = m.encoder_wrapper(src_images=inp["src_images"], encoder_type="image")[1]
    fx = dict(cfg=MICRO, vocab=vocab, shapes=shapes, inputs=inp,
              text_logits=t.detach(), image_logits=i.detach(), audio_logits=a.detach(),
              text_feats=feats_t, image_feats=feats_i, scale=scale.detach(),
              itc_loss=l_it.detach(), atc_loss=l_at.detach(), i2t=i2t, t2i=t2i, a2t=a2t, t2a=t2a,
              vl_text=vl_t, vl_image=vl_i, al_text=al_t, al_audio=al_a, grads=grads,
              text_rp_bucket_sum=m.encoder_wrapper.text_adapter.rp_bucket.sum(),
              image_rp_bucket=m.encoder_wrapper.image_adapter.rp_bucket.clone(),
              audio_rp_bucket_sum=m.encoder_wrapper.audio_adapter.rp_bucket.sum(),
              text_rp_bucket_corner=m.encoder_wrapper.text_adapter.rp_bucket[:40, :40].clone(),
              audio_rp_bucket_row=m.encoder_wrapper.audio_adapter.rp_bucket[700, :].clone())
    torch.save(fx, os.path.join(HERE, "micro_retrieval.pt"))
    print("micro: itc %.6f atc %.6f" % (l_it.item(), l_at.item()))


def tiny_text_fixture():
    """BASELINE.json configs[0]: tiny encoder (H=256, L=4) text-only extract_text_features, bs=8 seq=64."""
    m, shapes = build_ref_model(dict(TINY, use_image_moe=False, use_audio_moe=False), 50265, head_type="text")
    inp = synth.synth_inputs(8, text_len=63)
    with torch.no_grad():
        out = m(src_tokens=inp["src_tokens"], encoder_type="text")
    shapes = {k: v for k, v in shapes.items()}
    torch.save(dict(cfg=dict(TINY, use_image_moe=False, use_audio_moe=False), shapes=shapes,
                    inputs=inp, text_logits=out), os.path.join(HERE, "tiny_text.pt"))
    print("tiny text:", out.shape, out[0, :4])


def layer_fixture():
    """One encoder layer at a 4B-like aspect (hd=64) incl. drop-path, padded keys, all three FFNs + grads."""
    tl = R.ref("one_peace.models.transformer.transformer_layer")
    cfg = R.make_cfg(embed_dim=192, ffn_embed_dim=384, layers=1, attention_heads=3).encoder
    torch.manual_seed(0)
    layer = tl.TransformerEncoderLayer(cfg, drop_path_rate=0.0)
    shapes = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
    sd = synth.synth_state_dict(shapes)
    layer.load_state_dict(sd)
    layer.eval()
    S, B, H = 37, 3, 192
    g = torch.Generator().manual_seed(1)
    x = torch.randn(S, B, H, generator=g)
    bias = 0.5 * torch.randn(B, 3, S, S, generator=g)
    bias[1, :, :, 30:] = float("-inf")
    out = {}
    for et in ("text", "image", "audio"):
        xi = x.clone().requires_grad_(True)
        y = layer(xi, None, self_attn_bias=bias, encoder_type=et)
        w = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
        layer.zero_grad()
        (y * w).sum().backward()
        out[et] = dict(y=y.detach(), dx=xi.grad.clone(), grads=grads_summary(layer, ()))
    torch.save(dict(shapes=shapes, x=x, bias=bias, heads=3, out=out), os.path.join(HERE, "layer.pt"))
    print("layer:", out["text"]["y"].norm().item())


PRETRAIN_ENC = dict(embed_dim=128, ffn_embed_dim=256, layers=2, attention_heads=2, image_rel_bucket_size=4,
                    text_bucket_size=256, use_audio_moe=False)
PRETRAIN_DEC = dict(embed_dim=64, ffn_embed_dim=128, layers=1, attention_heads=1, use_audio_moe=False, use_attn_bias=False,
                    vision_encoder_type="none")


def masks_for(valid, keep_fraction, gen):
    """valid [B, S] bool (False = padding).  Position 0 (CLS) is always kept; returns preserve_ids [B, K] (-1 padded)
    and mask_indices [B, S] bool (True = masked out), the two forms the dataset hands over
    (data/pretrain_data/image_text_pretrain_dataset.py:85-104)."""
    B, S = valid.shape
    keep_rows, mask = [], torch.zeros(B, S, dtype=torch.bool)
    for b in range(B):
        cand = [i for i in range(1, S) if valid[b, i]]
        n_keep = max(1, int(round(len(cand) * keep_fraction)))
        perm = torch.randperm(len(cand), generator=gen).tolist()
        kept = sorted(cand[i] for i in perm[:n_keep])
        for i in cand:
            if i not in kept:
                mask[b, i] = True
        keep_rows.append([0] + kept)
    K = max(len(r) for r in keep_rows)
    ids = torch.full((B, K), -1, dtype=torch.long)
    for b, r in enumerate(keep_rows):
        ids[b, :len(r)] = torch.tensor(r)
    return ids, mask


def pretrain_fixture():
    """The full image-text pretraining objective (ITC + four DCL terms, image_text_pretrain_loss.py:76-160) on a micro
    model with the small decoder: six forward passes incl. the masked ones with per-sample preserve ids."""
    from types import SimpleNamespace
    pm = R.ref("one_peace.models.one_peace.one_peace_pretrain")
    crit_mod = R.ref("one_peace.criterions.image_text_pretrain_loss")
    vocab = 1000
    cfg = SimpleNamespace(encoder=R.make_cfg(**PRETRAIN_ENC).encoder, decoder=R.make_cfg(**PRETRAIN_DEC).encoder,
                          copy_rel_pos_table=False, reset_logit_scale=False, logit_scale_init=1 / 0.07, stage2_pretrain=False)
    torch.manual_seed(0)
    m = pm.OnePeacePretrainModel(cfg, R.TinyDictionary(vocab))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.split(".")[-1] in synth.NON_SYNTH for k in missing), (missing, unexpected)
    m.eval()
    B = 4
    inp = synth.synth_inputs(B, text_len=15, image_res=64, vocab=vocab)
    g = torch.Generator().manual_seed(11)
    text_valid = torch.cat([torch.ones(B, 1, dtype=torch.bool), inp["src_tokens"].ne(1)], dim=1)   # CLS + tokens
    image_valid = torch.ones(B, 17, dtype=torch.bool)
    ni = dict(inp)
    ni["text_preserve_ids"], ni["text_mask_indices"] = masks_for(text_valid, 0.6, g)
    ni["image_preserve_ids"], ni["image_mask_indices"] = masks_for(image_valid, 0.35, g)
    ni["vl_text_preserve_ids"], ni["vl_text_mask_indices"] = masks_for(text_valid, 0.6, g)
    ni["vl_image_preserve_ids"], ni["vl_image_mask_indices"] = masks_for(image_valid, 0.35, g)
    crit = crit_mod.ImageTextPretrainLossCriterion(None, 0.5, 1.0, 0.5, 0.5, 2.5, label_smoothing=0.0)
    sample = {"net_input": ni, "nsentences": B}
    loss, _, log = crit(m, sample)
    m.zero_grad()
    loss.backward()
    keep = {"logit_scale", "text_mask_token", "image_mask_token",
            "encoder_wrapper.fusion_model.layers.0.self_attn.q_proj.weight",
            "decoder_wrapper.fusion_model.layers.0.image_ffn.0.wi_0.weight"}
    with torch.no_grad():
        st = m(src_tokens=ni["src_tokens"], text_preserve_ids=ni["text_preserve_ids"], encoder_type="text")[0]
        si = m(src_images=ni["src_images"], image_preserve_ids=ni["image_preserve_ids"], encoder_type="image")[1]
        svt, svi, _ = m(src_tokens=ni["src_tokens"], text_preserve_ids=ni["vl_text_preserve_ids"], src_images=ni["src_images"],
                        image_preserve_ids=ni["vl_image_preserve_ids"], encoder_type="vl")
    fx = dict(enc=PRETRAIN_ENC, dec=PRETRAIN_DEC, vocab=vocab, shapes=shapes, net_input=ni,
              loss=loss.detach(), log={k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in log.items()},
              student_text=st, student_image=si, student_vl_text=svt, student_vl_image=svi, grads=grads_summary(m, keep))
    torch.save(fx, os.path.join(HERE, "micro_pretrain.pt"))
    print("pretrain: loss %.6f " % loss.item(), {k: round(float(v), 5) for k, v in log.items() if "loss" in k})


PRETRAIN_AL_ENC = dict(embed_dim=128, ffn_embed_dim=256, layers=2, attention_heads=2, text_bucket_size=256, audio_bucket_size=512,
                       use_image_moe=False)
PRETRAIN_AL_DEC = dict(embed_dim=64, ffn_embed_dim=128, layers=1, attention_heads=1, use_image_moe=False, use_attn_bias=False)


def pretrain_al_fixture(stage2=False):
    """The audio-language pretraining objective (ATC + three DCL terms, audio_text_pretrain_loss.py:73-157): frozen text
    teacher, audio / joint 'al' students with preserve ids, decoder with the spec-less fixed-position audio adapter and
    without layer scale (pretrain_al_3B.yaml:138-176).  stage2: the same with `stage2_pretrain: true`
    (one_peace_pretrain.py:98-104: text_proj and the encoder frozen except the audio adapter, audio_layer_norm and every
    layer's audio_ffn) -> micro_pretrain_al_stage2.pt: which parameters receive a gradient, and those gradients."""
    from types import SimpleNamespace
    pm = R.ref("one_peace.models.one_peace.one_peace_pretrain")
    crit_mod = R.ref("one_peace.criterions.audio_text_pretrain_loss")
    vocab = 1000
    dec = R.make_cfg(**PRETRAIN_AL_DEC).encoder
    dec.audio_adapter.feature_encoder_spec = None
    dec.audio_adapter.abs_pos_type = "fixed"
    dec.audio_adapter.bucket_size = 256
    dec.use_layer_scale = False
    cfg = SimpleNamespace(encoder=R.make_cfg(**PRETRAIN_AL_ENC).encoder, decoder=dec, copy_rel_pos_table=False,
                          reset_logit_scale=False, logit_scale_init=1 / 0.07, stage2_pretrain=bool(stage2))
    torch.manual_seed(0)
    m = pm.OnePeacePretrainModel(cfg, R.TinyDictionary(vocab))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.split(".")[-1] in synth.NON_SYNTH for k in missing), (missing, unexpected)
    m.eval()
    B = 4
    inp = synth.synth_inputs(B, text_len=15, audio_samples=8000, vocab=vocab)
    frames = inp["audio_padding_masks"].shape[1]           # CLS + frames
    inp["audio_padding_masks"][1, frames - 3:] = True      # two samples with padded audio tails
    inp["audio_padding_masks"][3, frames - 6:] = True
    g = torch.Generator().manual_seed(13)
    text_valid = torch.cat([torch.ones(B, 1, dtype=torch.bool), inp["src_tokens"].ne(1)], dim=1)
    audio_valid = ~inp["audio_padding_masks"]
    ni = dict(inp)
    ni["audio_preserve_ids"], ni["audio_mask_indices"] = masks_for(audio_valid, 0.45, g)
    ni["al_text_preserve_ids"], ni["al_text_mask_indices"] = masks_for(text_valid, 0.6, g)
    ni["al_audio_preserve_ids"], ni["al_audio_mask_indices"] = masks_for(audio_valid, 0.45, g)
    crit = crit_mod.AudioTextPretrainLossCriterion(None, 1.0, 0.5, 0.5, 2.5, label_smoothing=0.0)
    loss, _, log = crit(m, {"net_input": ni, "nsentences": B})
    m.zero_grad()
    loss.backward()
    keep = {"logit_scale", "text_mask_token", "audio_mask_token",
            "encoder_wrapper.fusion_model.layers.0.self_attn.q_proj.weight"}
    fx = dict(enc=PRETRAIN_AL_ENC, dec=PRETRAIN_AL_DEC, vocab=vocab, shapes=shapes, net_input=ni, loss=loss.detach(),
              log={k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in log.items()}, grads=grads_summary(m, keep))
    if stage2:
        fx["trainable"] = sorted(n for n, q in m.named_parameters() if q.requires_grad)
        fx["with_grad"] = sorted(n for n, q in m.named_parameters() if q.grad is not None)
    torch.save(fx, os.path.join(HERE, "micro_pretrain_al_stage2.pt" if stage2 else "micro_pretrain_al.pt"))
    print("pretrain al%s: loss %.6f " % (" (stage 2)" if stage2 else "", loss.item()),
          {k: round(float(v), 5) for k, v in log.items() if "loss" in k})


DEEP = dict(embed_dim=1536, ffn_embed_dim=6144, layers=8, attention_heads=24, image_rel_bucket_size=16,
            text_bucket_size=256, audio_bucket_size=512)


def deep_vision_fixture():
    """Vision branch at the 4B layer dimensions (H=1536, F=6144, 24 heads), EIGHT layers deep, 256^2 images (257 tokens), b=2:
    the reference's image-only retrieval model (head_type 'image').  Stored: the normalised CLS embeddings, the first feature
    rows, and -- for loss = sum(logits * w) -- every parameter gradient's norm (+ small gradients in full).  Weights are the
    deterministic synthetic ones of oracle/synth.py (302 M parameters, never stored)."""
    m, shapes = build_ref_model(DEEP, 1000, head_type="image")
    B = 2
    imgs = synth.synth_inputs(B, image_res=256, vocab=1000)["src_images"]
    logits = m(src_images=imgs, encoder_type="image")
    feats = m.encoder_wrapper(src_images=imgs, encoder_type="image")[1]
    w = synth.synth_tensor("deep/w", logits.shape, seed=5)
    m.zero_grad()
    (logits * w).sum().backward()
    grads = {}
    for n, q in m.named_parameters():  # the inputs / weights are regenerated from oracle/synth.py: store results only
        if q.grad is not None:
            grads[n + "#norm"] = q.grad.double().norm().float()
            if q.grad.numel() <= 1536:
                grads[n] = q.grad.clone()
    fx = dict(cfg=DEEP, vocab=1000, shapes=shapes, batch=B, image_res=256, logits=logits.detach(),
              feats_head=feats[:, :4].detach().clone(), grads=grads)
    torch.save(fx, os.path.join(HERE, "deep_vision.pt"))
    print("deep vision: logits norm %.4f, %d grads" % (float(logits.norm()), len(fx["grads"])))


def deep_text_audio_fixture():
    """Round 4: the text and audio towers at the 4B layer dimensions (H=1536, F=6144, 24 heads), EIGHT layers deep, WITH a backward
    (the deep fixtures pinned the image tower only): the reference's retrieval model on b = 3 captions of 24 tokens
    (padded rows) and 2 s waveforms (100 frames + CLS, no padding), loss = sum(text_logits * w_t) + sum(audio_logits * w_a).
    Stored: both normalised embeddings, the first feature rows of both towers, every parameter gradient's norm, small gradients in
    full and the first four rows of the large ones (grads_summary).  Weights: oracle/synth.py, never stored."""
    m, shapes = build_ref_model(DEEP, 1000, head_type="al")
    B = 3
    inp = synth.synth_inputs(B, text_len=24, audio_samples=32000, vocab=1000)
    t = m(src_tokens=inp["src_tokens"], encoder_type="text")
    a = m(src_audios=inp["src_audios"], audio_padding_masks=inp["audio_padding_masks"], encoder_type="audio")
    with torch.no_grad():
        feats_t = m.encoder_wrapper(src_tokens=inp["src_tokens"], encoder_type="text")[0]
        feats_a = m.encoder_wrapper(src_audios=inp["src_audios"], audio_padding_masks=inp["audio_padding_masks"], encoder_type="audio")[2]
    wt = synth.synth_tensor("deep_ta/wt", t.shape, seed=6)
    wa = synth.synth_tensor("deep_ta/wa", a.shape, seed=7)
    m.zero_grad()
    ((t * wt).sum() + (a * wa).sum()).backward()
    grads = {k: v for k, v in grads_summary(m, set()).items()  # row probes of the first and last layer only (file size)
             if not k.endswith("#rows4") or ".layers.0." in k or ".layers.7." in k}
    fx = dict(cfg=DEEP, vocab=1000, shapes=shapes, batch=B, text_len=24, audio_samples=32000, text_logits=t.detach(), audio_logits=a.detach(),
              text_feats_head=feats_t[:, :4].detach().clone(), audio_feats_head=feats_a[:, :4].detach().clone(), grads=grads)
    torch.save(fx, os.path.join(HERE, "deep_text_audio.pt"))
    print("deep text+audio: logits norms %.4f / %.4f, %d gradient entries" % (float(t.norm()), float(a.norm()), len(grads)))


DEEP40 = dict(DEEP, layers=40)


def deep_vision40_fixture():
    """The reference's FULL-DEPTH image tower: 40 layers at the 4B layer dimensions (the vision branch of ONE-PEACE-4B, 1.5 B
    parameters, BASELINE configs[1]), one 256^2 image, forward only (2 s on CPU): the normalised CLS embedding, three probe rows
    of the final features and the residual-stream norm after the last layer.  Weights are the deterministic synthetic ones of
    oracle/synth.py and are never stored."""
    m, shapes = build_ref_model(DEEP40, 1000, head_type="image")
    imgs = synth.synth_inputs(1, image_res=256, vocab=1000)["src_images"]
    with torch.no_grad():
        logits = m(src_images=imgs, encoder_type="image")
        feats = m.encoder_wrapper(src_images=imgs, encoder_type="image")[1]
    fx = dict(cfg=DEEP40, vocab=1000, nparams=sum(int(torch.tensor(v).prod()) for v in shapes.values()), batch=1, image_res=256,
              logits=logits.detach(), feats_rows=feats[0, [0, 1, 128, 256]].detach().clone(), feats_norm=feats.double().norm().float())
    torch.save(fx, os.path.join(HERE, "deep_vision40.pt"))
    print("deep vision 40: logits norm %.4f, feats norm %.4f" % (float(logits.norm()), float(fx["feats_norm"])))


OPTIM = dict(lr=[1e-3, 2e-3, 1.5e-3], betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05, layer_decay=0.8, clip_norm=0.7)


optim_grad = synth.optim_grad


def optim_fixture():
    """The reference's optimiser leg on the micro model in bf16, three steps with layer-wise lr decay (trainer.py:265-278 ->
    utils/layer_decay.py:34-77 param groups with lr_scale; optim/base_optimizer.py:8-14 set_lr), global-norm clipping
    (fairseq/utils.py:349-398, applied IN PLACE to the bf16 gradients) and the in-repo Adam (optim/adam.py:124-253: fp32 moments,
    fp32 math on bf16 parameters).  Stored: the group of every parameter, the gradient norms, and all parameters after each step."""
    ro = R.ref_optim()
    m, shapes = build_ref_model(MICRO, 1000)
    m = m.to(torch.bfloat16)
    L = MICRO["layers"]
    assigner = ro.LayerDecayValueAssigner([OPTIM["layer_decay"] ** (L + 1 - i) for i in range(L + 2)])
    groups = ro.get_parameter_groups(m, OPTIM["weight_decay"], m.no_weight_decay(), assigner.get_layer_id, assigner.get_scale)
    opt = ro.Adam(groups, lr=OPTIM["lr"][0], betas=OPTIM["betas"], eps=OPTIM["eps"], weight_decay=OPTIM["weight_decay"])
    names = {id(p): n for n, p in m.named_parameters()}
    assign = {}
    for gr in opt.param_groups:
        for p in gr["params"]:
            assign[names[id(p)]] = (float(gr["lr_scale"]), float(gr["weight_decay"]))
    params_after, norms = [], []
    for step, lr in enumerate(OPTIM["lr"], start=1):
        for gr in opt.param_groups:  # optim/base_optimizer.py:8-14
            gr["lr"] = lr * gr["lr_scale"]
        for n, p in m.named_parameters():
            p.grad = optim_grad(n, p.shape, step)
        norms.append(ro.clip_grad_norm_(list(m.parameters()), OPTIM["clip_norm"]).float().clone())
        opt.step()
        snap = {}
        for n, p in m.named_parameters():  # every parameter's norm; small ones in full, the first rows of the big ones
            d = p.detach()
            snap[n + "#norm"] = d.double().norm().float()
            if d.numel() <= 4096:
                snap[n] = d.clone()
            else:
                snap[n + "#head"] = d.reshape(-1)[:2048].clone()
        params_after.append(snap)
    fx = dict(cfg=MICRO, vocab=1000, shapes=shapes, optim=OPTIM, assign=assign, grad_norms=norms, params_after=params_after)
    torch.save(fx, os.path.join(HERE, "optim.pt"))
    print("optim: %d groups, grad norms %s" % (len(opt.param_groups), [round(float(x), 4) for x in norms]))


LAYERDROP = dict(embed_dim=128, ffn_embed_dim=256, layers=4, attention_heads=2, image_rel_bucket_size=4, text_bucket_size=256,
                 audio_bucket_size=512, layer_scale_init_value=1e-1)
LAYERDROP_CASES = ((False, 1), (True, 1), (True, 5), (True, 3), (True, 11))  # (training, seed of the CPU generator before the encoder call)


def layerdrop_fixture():
    """return_all_hiddens + layerdrop on the UNMODIFIED reference encoder (transformer_encoder.py:48-51,186-199;
    fairseq/modules/layer_drop.py:13-44): a joint text+image stream and a text-only stream through four layers with layerdrop 0.5, eval and
    training mode; the layerdrop mask of a pass is what `torch.empty(4).uniform_()` draws from the seeded CPU generator at the start of
    the layer loop (the adapters run before the seed is set; drop-path and dropout are 0: nothing else consumes random numbers).
    Stored per case: which layers ran, every returned state and the encoder output.

    The model is BUILT with layerdrop 1e-9 and the list's probability set to 0.5 afterwards: the reference's constructors enumerate
    `fusion_model.layers` in training mode to wrap the layers (one_peace_retrieval.py: `for i, layer in enumerate(...layers):
    ...layers[i] = fsdp_wrap(layer)`), which with layerdrop > 0 DRAWS a mask -- a dropped layer shifts the index and the loop then
    stores a later layer object into an earlier slot (seed 0, p = 0.5: slots 2 and 3 end up the same module, the original layer 2 is
    gone).  That is a construction accident of the reference, not part of the forward semantics pinned here."""
    rt = R.ref("one_peace.models.one_peace.one_peace_retrieval")
    cfg = R.make_cfg(**LAYERDROP)
    cfg.encoder.layerdrop = 1e-9
    torch.manual_seed(0)
    m = rt.OnePeaceRetrievalModel(cfg, R.TinyDictionary(1000), "val")
    layers = m.encoder_wrapper.fusion_model.layers
    assert len({id(x) for x in layers._modules.values()}) == 4
    layers.p = 0.5
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    missing, unexpected = m.load_state_dict(synth.synth_state_dict(shapes), strict=False)
    assert not unexpected and all(k.split(".")[-1] in synth.NON_SYNTH for k in missing), (missing, unexpected)
    inp = synth.synth_inputs(3, text_len=15, image_res=64, audio_samples=8000, vocab=1000)
    W = m.encoder_wrapper
    assert type(W.fusion_model.layers).__name__ == "LayerDropModuleList"
    cases = []
    for train, seed in LAYERDROP_CASES:
        m.train(train)
        rec = dict(train=train, seed=seed)
        with torch.no_grad():
            t = W.text_adapter(inp["src_tokens"], None, None, None)
            i = W.image_adapter(inp["src_images"], None, None, None, False)
            torch.manual_seed(seed)
            ran = (torch.empty(4).uniform_() > 0.5).tolist() if train else [True] * 4
            for et, infos in (("vl", (t, i, None)), ("text", (t, None, None))):
                torch.manual_seed(seed)
                out = W.fusion_model(*infos, return_all_hiddens=True, encoder_type=et)
                assert len(out["text_encoder_states"]) == sum(ran) and out["audio_encoder_states"] == []
                rec[et] = dict(encoder_out=out["encoder_out"][0].clone(), text_states=[x.clone() for x in out["text_encoder_states"]],
                               image_states=[x.clone() for x in out["image_encoder_states"]])
        rec["ran"] = ran
        cases.append(rec)
    torch.save(dict(cfg=LAYERDROP, layerdrop=0.5, vocab=1000, shapes=shapes, inputs=inp, cases=cases), os.path.join(HERE, "layerdrop_hiddens.pt"))
    print("layerdrop:", [(c["train"], c["seed"], c["ran"]) for c in cases])


if __name__ == "__main__":
    assert R.reference_available(), "needs /root/reference"
    if len(sys.argv) > 1 and sys.argv[1] in ("optim", "deep", "stage2", "deep40", "deep_ta", "layerdrop"):
        {"layerdrop": layerdrop_fixture, "optim": optim_fixture, "deep": deep_vision_fixture, "stage2": lambda: pretrain_al_fixture(stage2=True),
         "deep40": deep_vision40_fixture, "deep_ta": deep_text_audio_fixture}[sys.argv[1]]()
        sys.exit(0)
    micro_fixture()
    tiny_text_fixture()
    layer_fixture()
    pretrain_fixture()
    pretrain_al_fixture()
    pretrain_al_fixture(stage2=True)
    optim_fixture()
    deep_vision_fixture()
    deep_vision40_fixture()
    deep_text_audio_fixture()
    layerdrop_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
