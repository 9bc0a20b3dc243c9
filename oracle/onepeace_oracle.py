"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the ONE-PEACE hot path (the parity oracle).

This file restates, as small pure functions over a flat ``{name: tensor}`` state dict (the
reference's own state-dict key names), the algorithm of the reference path that
``BASELINE.json.north_star`` names: modality adapters -> shared-attention / per-modality GeGLU-FFN
pre-LN Transformer -> CLS projection + L2-normalise -> all-gather + InfoNCE.  It is written from the
reference's behaviour, not its code: every function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module -- and only as the checker or the timed CPU baseline, never as a product path.  The product
(``one-peace_amd``) never imports it and fails loudly when its HIP library is missing.

Pinning: the reference holds NO golden vectors / known-answer tests for this path, so
the oracle is pinned against *outputs of the reference itself*, executed unmodified on CPU through
``oracle/ref_shim.py`` in the authoring container: ``tests/golden/make_golden.py`` wrote
``tests/golden/*.pt`` and ``tests/test_oracle_golden.py`` checks this file against them (and, when
``/root/reference`` is present, against the live reference modules).

All math is torch-on-CPU; pass fp32 tensors for the canonical oracle (fp64 also works).  Gradients
come from torch autograd over these functions, which is exactly how the reference differentiates
(SURVEY.md 3.4).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

LN_EPS = 1e-5  # one_peace/models/components.py:23 (torch.nn.LayerNorm default eps passed through)


# ------------------------------------------------------------------------------------------------
# primitive ops
# ------------------------------------------------------------------------------------------------
def layer_norm(x, w=None, b=None, eps=LN_EPS):
    """components.py:23-26 -> torch.nn.LayerNorm over the last dim, biased variance, eps inside sqrt."""
    xf = x
    mu = xf.mean(dim=-1, keepdim=True)
    xc = xf - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    y = xc * torch.rsqrt(var + eps)
    if w is not None:
        y = y * w
    if b is not None:
        y = y + b
    return y


def gelu_erf(x):
    """transformer_layer.py:61 uses nn.GELU() = exact erf form 0.5 x (1 + erf(x / sqrt 2))."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x, w, b=None):
    """components.py:29-34: nn.Linear, weight layout [out, in]."""
    y = x @ w.t()
    return y if b is None else y + b


def geglu_ffn(x, sd, p, scale_fc=True):
    """transformer_layer.py:54-67 (GeGLU) + :149-157 (Sequential: GeGLU, dropout(p=0), LN(F), Linear).

    keys: p.0.wi_0.weight, p.0.wi_1.weight, p.2.{weight,bias}, p.3.{weight,bias}
    """
    g = gelu_erf(linear(x, sd[p + ".0.wi_0.weight"])) * linear(x, sd[p + ".0.wi_1.weight"])
    if scale_fc:
        g = layer_norm(g, sd[p + ".2.weight"], sd[p + ".2.bias"])
    return linear(g, sd[p + ".3.weight"], sd[p + ".3.bias"])


def self_attention(x, sd, p, num_heads, bias=None):
    """multihead_attention.py:102-124, fallback branch (x is time-major [S, B, H]).

    q,v,out have biases, k has none (:63-66); q is scaled by head_dim**-0.5 (:106); the additive
    bias [B, heads, S, S] carries rel-pos bias and -inf on padded keys (:108-109); softmax is taken in
    fp32 (:111, fairseq/utils.py:520); sub-LayerNorm sits between P.V and out_proj (:122-124).
    """
    S, B, H = x.shape
    hd = H // num_heads
    q = linear(x, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"])
    k = linear(x, sd[p + ".k_proj.weight"])
    v = linear(x, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"])

    def heads(t):  # [S, B, H] -> [B, heads, S, hd]
        return t.reshape(S, B, num_heads, hd).permute(1, 2, 0, 3)

    q, k, v = heads(q) * (hd ** -0.5), heads(k), heads(v)
    s = q @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias
    pr = torch.softmax(s.float(), dim=-1).to(s.dtype)
    a = (pr @ v).permute(2, 0, 1, 3).reshape(S, B, H)
    if (p + ".ln.weight") in sd:
        a = layer_norm(a, sd[p + ".ln.weight"], sd[p + ".ln.bias"])
    return linear(a, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def residual_scale(x, gamma, residual, path_scale=None):
    """transformer_layer.py:70-88 fused_dropout_res with dropout p = 0: residual + droppath(gamma * x).

    `path_scale` is the per-sample drop-path multiplier, shape [B] holding 0 or 1/keep_prob (:80-85);
    None = evaluation / rate 0.
    """
    y = x if gamma is None else gamma * x
    if path_scale is not None:
        y = y * path_scale.view(1, -1, 1)
    return y + residual


def encoder_layer(x, sd, p, num_heads, encoder_type, bias=None, text_seq_len=0, image_seq_len=0,
                  audio_seq_len=0, path_scale=None):
    """transformer_layer.py:165-228.  x: [S, B, H] time-major.  path_scale: None, one [B] multiplier vector for both residual
    branches, or a pair (attention branch, FFN branch) -- the reference draws a fresh mask per fused_dropout_res call (:78-85)."""
    g1, g2 = sd.get(p + ".gamma_1"), sd.get(p + ".gamma_2")
    ps_attn, ps_ffn = path_scale if isinstance(path_scale, (tuple, list)) else (path_scale, path_scale)
    res = x
    h = layer_norm(x, sd[p + ".self_attn_layer_norm.weight"], sd[p + ".self_attn_layer_norm.bias"])
    h = self_attention(h, sd, p + ".self_attn", num_heads, bias)
    if (p + ".attn_ln.weight") in sd:  # cfg.scale_attn (:139); off in the shipped configs
        h = layer_norm(h, sd[p + ".attn_ln.weight"], sd[p + ".attn_ln.bias"])
    x = residual_scale(h, g1, res, ps_attn)
    res = x
    h = layer_norm(x, sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"])
    scale_fc = (p + ".text_ffn.2.weight") in sd or (p + ".image_ffn.2.weight") in sd or \
               (p + ".audio_ffn.2.weight") in sd
    if encoder_type in ("text", "image", "audio"):
        h = geglu_ffn(h, sd, p + "." + encoder_type + "_ffn", scale_fc)
    elif encoder_type == "vl":  # :210-213 -- split on the sequence axis, per-modality FFN, concat
        h = torch.cat([geglu_ffn(h[:text_seq_len], sd, p + ".text_ffn", scale_fc),
                       geglu_ffn(h[-image_seq_len:], sd, p + ".image_ffn", scale_fc)], dim=0)
    elif encoder_type == "al":  # :214-217
        h = torch.cat([geglu_ffn(h[:text_seq_len], sd, p + ".text_ffn", scale_fc),
                       geglu_ffn(h[-audio_seq_len:], sd, p + ".audio_ffn", scale_fc)], dim=0)
    else:
        raise NotImplementedError(encoder_type)
    return residual_scale(h, g2, res, ps_ffn)


# ------------------------------------------------------------------------------------------------
# relative-position buckets
# ------------------------------------------------------------------------------------------------
def image_bucket_position(bucket_size, num_rel_dis):
    """adapter/image.py:19-34: Swin/BEiT 2-D relative index on a bucket_size^2 grid plus 3 CLS buckets."""
    n = bucket_size
    idx = torch.arange(n * n)
    r, c = idx // n, idx % n
    dr = r[:, None] - r[None, :] + (n - 1)
    dc = c[:, None] - c[None, :] + (n - 1)
    out = torch.zeros(n * n + 1, n * n + 1, dtype=torch.long)
    out[1:, 1:] = dr * (2 * n - 1) + dc
    out[0, :] = num_rel_dis - 3
    out[:, 0] = num_rel_dis - 2
    out[0, 0] = num_rel_dis - 1
    return out


def token_bucket_position(bucket_size, max_position=1024):
    """adapter/text.py:18-29 (= adapter/audio.py:20-32): T5-style log bucketing of i-j, then the three
    CLS buckets written by the adapter constructors (text.py:65-67, audio.py:104-106)."""
    pos = torch.arange(max_position, dtype=torch.long)
    rel = pos[:, None] - pos[None, :]
    mid = bucket_size // 2
    near = (rel < mid) & (rel > -mid)
    a = torch.where(near, torch.full_like(rel, mid - 1), rel.abs())
    logp = mid + torch.ceil(torch.log(a / mid) / math.log((max_position - 1) / mid) * (mid - 1)).long()
    bucket = torch.where(a <= mid, rel, logp * torch.sign(rel)).long() + bucket_size - 1
    n = 2 * bucket_size - 1
    bucket[0, :] = n
    bucket[:, 0] = n + 1
    bucket[0, 0] = n + 2
    return bucket


def rel_pos_bias(table, rp_bucket):
    """adapter/*.py get_rel_pos_bias: table[rp_bucket] -> [heads, S, S] (the reference then expands over B)."""
    return table[rp_bucket].permute(2, 0, 1)


# ------------------------------------------------------------------------------------------------
# adapters (unmasked path: preserve_ids / preserve_embed are None)
# ------------------------------------------------------------------------------------------------
def text_adapter(sd, p, src_tokens, pad_idx=1):
    """adapter/text.py:111-164.  Returns x [B, S+1, H], padding_mask [B, S+1], bias list ([heads,S,S])."""
    B, T = src_tokens.shape
    pad = torch.zeros(B, T + 1, dtype=torch.bool)
    pad[:, 1:] = src_tokens.eq(pad_idx)
    tok = sd[p + ".embed_tokens.weight"][src_tokens]
    cls = sd[p + ".cls_embedding"].expand(B, -1, -1)
    x = torch.cat([cls, tok], dim=1) + sd[p + ".embed_positions.weight"][: T + 1].unsqueeze(0)
    biases = None
    if (p + ".rp_bucket") in sd:
        rp = sd[p + ".rp_bucket"][: T + 1, : T + 1]
        biases = [rel_pos_bias(sd[k], rp) for k in _table_keys(sd, p)]
    return x, pad, biases


def _table_keys(sd, p):
    ks = [k for k in sd if k.startswith(p + ".rel_pos_table_list.")]
    return sorted(ks, key=lambda k: int(k.split(".")[-2]))


def image_stem(sd, p, src_images):
    """adapter/image.py:66-75 hMLP stem: Conv(k4,s4) LN2D GELU Conv(k2,s2) LN2D GELU Conv(k2,s2)."""
    x = F.conv2d(src_images, sd[p + ".embed_images.0.weight"], sd[p + ".embed_images.0.bias"], stride=4)
    x = layer_norm(x.permute(0, 2, 3, 1), sd[p + ".embed_images.1.layer_norm.weight"],
                   sd[p + ".embed_images.1.layer_norm.bias"]).permute(0, 3, 1, 2)
    x = gelu_erf(x)
    x = F.conv2d(x, sd[p + ".embed_images.3.weight"], sd[p + ".embed_images.3.bias"], stride=2)
    x = layer_norm(x.permute(0, 2, 3, 1), sd[p + ".embed_images.4.layer_norm.weight"],
                   sd[p + ".embed_images.4.layer_norm.bias"]).permute(0, 3, 1, 2)
    x = gelu_erf(x)
    x = F.conv2d(x, sd[p + ".embed_images.6.weight"], sd[p + ".embed_images.6.bias"], stride=2)
    return x.flatten(2).transpose(1, 2)  # [B, (R/16)^2, H]


def image_adapter(sd, p, src_images):
    """adapter/image.py:206-260 (unmasked), incl. bicubic resize of pos_embed when R/16 != bucket (:173-186)."""
    B = src_images.shape[0]
    win = src_images.shape[2] // 16
    pos = sd[p + ".pos_embed"]
    bucket = int(round(math.sqrt(pos.shape[0] - 1)))
    if win != bucket:
        grid = pos[1:].reshape(1, bucket, bucket, -1).permute(0, 3, 1, 2).float()
        grid = F.interpolate(grid, size=(win, win), mode="bicubic").to(pos.dtype)
        pos = torch.cat([pos[:1], grid.permute(0, 2, 3, 1).reshape(win * win, -1)], dim=0)
    x = torch.cat([sd[p + ".cls_embedding"].expand(B, -1, -1), image_stem(sd, p, src_images)], dim=1)
    x = x + pos.unsqueeze(0)
    pad = torch.zeros(B, win * win + 1, dtype=torch.bool)
    biases = None
    if (p + ".rp_bucket") in sd:
        biases = [rel_pos_bias(sd[k], sd[p + ".rp_bucket"]) for k in _table_keys(sd, p)]
    return x, pad, biases


AUDIO_CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2  # unify_model_config.py:75


def audio_adapter(sd, p, src_audios, padding_mask, conv_pos_groups=16):
    """adapter/audio.py:150-210 (unmasked, abs_pos_type='conv').

    Feature extractor :254-311: 7x [Conv1d(no bias) -> LN over channels -> GELU]; then LN(512) and
    Linear(512->H) (:46-55).  Positional encoder :57-84: 5x [grouped Conv1d(k, pad k//2) -> LN without
    affine -> GELU] applied to the frame embeddings; CLS gets cls_pos_embed (:192-195).
    """
    B = src_audios.shape[0]
    x = src_audios.unsqueeze(1)
    for i, (_, _, stride) in enumerate(AUDIO_CONV_SPEC):
        q = "%s.embed_audios.0.conv_layers.%d" % (p, i)
        x = F.conv1d(x, sd[q + ".0.weight"], None, stride=stride)
        x = layer_norm(x.transpose(1, 2), sd[q + ".2.1.weight"], sd[q + ".2.1.bias"]).transpose(1, 2)
        x = gelu_erf(x)
    x = layer_norm(x.transpose(1, 2), sd[p + ".embed_audios.2.weight"], sd[p + ".embed_audios.2.bias"])
    x = linear(x, sd[p + ".embed_audios.3.weight"], sd[p + ".embed_audios.3.bias"])  # [B, T, H]
    pe = x.transpose(1, 2)
    i = 1
    while ("%s.embed_positions.%d.0.weight" % (p, i)) in sd:
        w, b = sd["%s.embed_positions.%d.0.weight" % (p, i)], sd["%s.embed_positions.%d.0.bias" % (p, i)]
        k = w.shape[-1]
        pe = F.conv1d(pe, w, b, padding=k // 2, groups=conv_pos_groups)
        if k % 2 == 0:  # SamePad (:240-248)
            pe = pe[:, :, :-1]
        pe = gelu_erf(layer_norm(pe.transpose(1, 2)).transpose(1, 2))
        i += 1
    pe = torch.cat([sd[p + ".cls_pos_embed"].expand(B, -1, -1), pe.transpose(1, 2)], dim=1)
    x = torch.cat([sd[p + ".cls_embedding"].expand(B, -1, -1), x], dim=1) + pe
    T1 = padding_mask.shape[1]
    biases = None
    if (p + ".rp_bucket") in sd:
        rp = sd[p + ".rp_bucket"][:T1, :T1]
        biases = [rel_pos_bias(sd[k], rp) for k in _table_keys(sd, p)]
    return x, padding_mask, biases


# ------------------------------------------------------------------------------------------------
# encoder + models
# ------------------------------------------------------------------------------------------------
def layerdrop_mask(num_layers, p, training=True):
    """fairseq/modules/layer_drop.py:38-44: one uniform draw per layer from the CPU generator when the layer list is iterated; layer i
    runs iff not training or u_i > p."""
    u = torch.empty(num_layers).uniform_()
    return [(not training) or bool(u[i] > p) for i in range(num_layers)]


def encoder_forward(sd, p, num_heads, num_layers, encoder_type, text_info=None, image_info=None,
                    audio_info=None, path_scales=None, layer_mask=None, return_all_hiddens=False):
    """transformer_encoder.py:73-232.  *_info = (x [B,S,H], pad [B,S], [bias [heads,S,S], ...] or None).

    Stream concat (:116-137), zeroing of padded positions (:139-142), block-diagonal bias with -inf on
    padded keys (:144-162; cross-modal blocks stay 0), layer loop with one shared or per-layer bias
    (:172-188), per-modality final LayerNorm (:201-220).  Returns x batch-major [B, S, H] and the mask.
    layer_mask: the layers a LayerDropModuleList pass runs (layerdrop_mask; :48-51 -- the bias index stays the position in the
    ITERATION, :172-176).  return_all_hiddens (:186-199): also {modality: [time-major slice of every executed layer's output]}.
    """
    infos = {"text": [text_info], "image": [image_info], "audio": [audio_info],
             "vl": [text_info, image_info], "al": [text_info, audio_info]}[encoder_type]
    x = torch.cat([i[0] for i in infos], dim=1)
    pad = torch.cat([i[1] for i in infos], dim=1)
    lens = [i[0].shape[1] for i in infos]
    nbias = len(infos[0][2]) if infos[0][2] is not None else 0
    if bool(pad.any()):
        x = x * (1 - pad.unsqueeze(-1).to(x.dtype))
    B, S, H = x.shape
    biases = []
    for li in range(nbias):
        bias = x.new_zeros(B, num_heads, S, S)
        off = 0
        for info, n in zip(infos, lens):
            if info[2] is not None:
                bias[:, :, off:off + n, off:off + n] += info[2][li].unsqueeze(0)
            off += n
        if bool(pad.any()):
            bias = bias.masked_fill(pad.view(B, 1, 1, S), float("-inf"))
        biases.append(bias)
    x = x.transpose(0, 1)
    tl = text_info[0].shape[1] if text_info is not None else 0
    il = image_info[0].shape[1] if image_info is not None else 0
    al = audio_info[0].shape[1] if audio_info is not None else 0
    states = {"text": [], "image": [], "audio": []}
    idx = 0  # position in the iteration over the executed layers (the reference enumerates the LayerDropModuleList iterator)
    for li in range(num_layers):
        if layer_mask is not None and not layer_mask[li]:
            continue
        bias = None if not biases else (biases[0] if len(biases) == 1 else biases[idx])
        ps = None if path_scales is None else path_scales[li]
        x = encoder_layer(x, sd, "%s.layers.%d" % (p, li), num_heads, encoder_type, bias, tl, il, al, ps)
        idx += 1
        if return_all_hiddens:
            if text_info is not None:
                states["text"].append(x[:tl])
            if image_info is not None:
                states["image"].append(x[tl:tl + il])
            if audio_info is not None:
                states["audio"].append(x[tl + il:tl + il + al])

    def final(t, m):
        k = "%s.%s_layer_norm.weight" % (p, m)
        return layer_norm(t, sd[k], sd[k[:-6] + "bias"]) if k in sd else t

    if encoder_type in ("text", "image", "audio"):
        x = final(x, encoder_type)
    else:
        second = "image" if encoder_type == "vl" else "audio"
        n2 = il if encoder_type == "vl" else al
        x = torch.cat([final(x[:tl], "text"), final(x[-n2:], second)], dim=0)
    if return_all_hiddens:
        return x.transpose(0, 1), pad, states
    return x.transpose(0, 1), pad


def model_wrapper_forward(sd, p, num_heads, num_layers, encoder_type, src_tokens=None, src_images=None,
                          src_audios=None, audio_padding_masks=None, path_scales=None):
    """one_peace_base.py:68-129: adapters -> fusion_model -> per-modality slices (text first, other last)."""
    ti = ii = ai = None
    if encoder_type in ("text", "vl", "al"):
        ti = text_adapter(sd, p + ".text_adapter", src_tokens)
    if encoder_type in ("image", "vl"):
        ii = image_adapter(sd, p + ".image_adapter", src_images)
    if encoder_type in ("audio", "al"):
        ai = audio_adapter(sd, p + ".audio_adapter", src_audios, audio_padding_masks)
    x, pad = encoder_forward(sd, p + ".fusion_model", num_heads, num_layers, encoder_type, ti, ii, ai, path_scales)
    out = {}
    if ti is not None:
        out["text"] = x[:, : ti[0].shape[1]]
    if ii is not None:
        out["image"] = x[:, -ii[0].shape[1]:]
    if ai is not None:
        out["audio"] = x[:, -ai[0].shape[1]:]
    return out, pad


def l2_normalize(x, eps=1e-12):
    """F.normalize(dim=1): x / max(||x||_2, eps) (one_peace_retrieval.py:112,116,120)."""
    return x / x.norm(dim=1, keepdim=True).clamp_min(eps)


def contrastive_embed(sd, num_heads, num_layers, encoder_type, path_scales=None, **inputs):
    """one_peace_retrieval.py:96-123 / one_peace_pretrain.py:162-173: normalise(proj(CLS))."""
    feats, _ = model_wrapper_forward(sd, "encoder_wrapper", num_heads, num_layers, encoder_type,
                                     path_scales=path_scales, **inputs)
    f = feats[encoder_type]
    proj = linear(f[:, 0, :], sd[encoder_type + "_proj.weight"], sd[encoder_type + "_proj.bias"])
    return l2_normalize(proj), f


def logit_scale_exp(logit_scale):
    """one_peace_pretrain.py:118-122: clamp to [0, ln 100] (in place in the reference), then exp."""
    return logit_scale.clamp(0, math.log(100)).exp()


# ------------------------------------------------------------------------------------------------
# contrastive head
# ------------------------------------------------------------------------------------------------
def smoothed_nll(lprobs, target, epsilon=0.0):
    """image_text_retrieval_loss.py:15-25 adjust_label_smoothed_nll_loss (mean over rows)."""
    nll = -lprobs.gather(-1, target.unsqueeze(-1)).squeeze(-1)
    if epsilon != 0:
        smooth = -lprobs.sum(dim=-1)
        eps_i = epsilon / (lprobs.size(-1) - 1)
        nll = (1.0 - epsilon - eps_i) * nll + eps_i * smooth
    return nll.mean()


def itc_loss(a_local, b_local, a_all, b_all, scale, rank=0, label_smoothing=0.0):
    """image_text_pretrain_loss.py:164-185 / image_text_retrieval_loss.py:92-112 (ATC: audio_text_*:160-181).

    a = image (or audio) embeddings, b = text embeddings; *_all are the rank-major all-gathered
    copies WITHOUT gradient (gather_without_grad :30-39).  sim = scale * local @ all^T, fp32
    log-softmax, targets rank*bsz + i, mean of the two directions; also the argmax hit counts.
    """
    bsz = a_local.shape[0]
    tgt = torch.arange(bsz) + bsz * rank
    sim_a2b = scale * a_local @ b_all.detach().t()
    sim_b2a = scale * b_local @ a_all.detach().t()
    la = F.log_softmax(sim_a2b.float(), dim=-1).to(sim_a2b.dtype)
    lb = F.log_softmax(sim_b2a.float(), dim=-1).to(sim_b2a.dtype)
    loss = (smoothed_nll(la, tgt, label_smoothing) + smoothed_nll(lb, tgt, label_smoothing)) / 2
    a_ok = (sim_a2b.argmax(dim=1) == tgt).float().sum()
    b_ok = (sim_b2a.argmax(dim=1) == tgt).float().sum()
    return loss, a_ok, b_ok


# ------------------------------------------------------------------------------------------------
# optimiser + data-parallel gradient reduction (callers of the hot path)
# ------------------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """one_peace/optim/adam.py:186-253: fp32 math on (possibly bf16) params; decoupled decay applied to
    the parameter BEFORE the Adam update; eps added to sqrt(v) (not bias-corrected v)."""
    g = g.float()
    pf = p.float()
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    if weight_decay != 0:
        pf = pf + pf * (-weight_decay * lr)
    pf = pf - step_size * (m / denom)
    p.copy_(pf)
    return p


def clip_coef(grads, max_norm):
    """fairseq/fairseq/utils.py:349-397 (clip_grad_norm_): total L2 norm over all gradients in fp32;
    clip_coef = clamp(max_norm / (total_norm + 1e-6), max=1).  Returns (total_norm, clip_coef)."""
    total = torch.norm(torch.stack([torch.norm(g.float(), p=2) for g in grads]))
    coef = (float(max_norm) / (total + 1e-6)).clamp(max=1.0) if max_norm > 0 else torch.tensor(1.0)
    return total, coef


def layer_id_of(name, num_max_layer):
    """one_peace/utils/layer_decay.py:8-21 (get_num_layer) on a parameter name without the 'encoder_wrapper.' prefix."""
    for ad in ("text_adapter", "image_adapter", "audio_adapter"):
        if name.startswith(ad):
            rest = name[len(ad) + 1:]
            return int(rest.split(".")[1]) + 1 if rest.startswith("rel_pos_table") else 0
    if name.startswith("fusion_model.layers"):
        return int(name.split(".")[2]) + 1
    return num_max_layer - 1


def param_groups(named_params, weight_decay, skip_list, num_layers=None, layer_decay=1.0):
    """trainer.py:265-278 + utils/layer_decay.py:34-77: {name: (lr_scale, weight_decay)}.  No decay for ndim <= 1, '.bias' and
    the model's no_weight_decay() names; with layer_decay < 1 the lr of layer id i is scaled by layer_decay ** (L + 1 - i)."""
    out = {}
    for name, p in named_params:
        wd = 0.0 if (p.ndim <= 1 or name.endswith(".bias") or name in skip_list) else weight_decay
        scale = 1.0
        if num_layers is not None and layer_decay < 1.0:
            values = [layer_decay ** (num_layers + 1 - i) for i in range(num_layers + 2)]
            var = name[len("encoder_wrapper."):] if name.startswith("encoder_wrapper.") else name
            scale = values[layer_id_of(var, len(values))]
        out[name] = (scale, wd)
    return out


def clip_grad_norm_(grads, max_norm):
    """fairseq/fairseq/utils.py:349-398: returns the total norm and scales the gradients IN PLACE (in their own dtype: bf16
    gradients are rounded after the multiplication)."""
    total, coef = clip_coef(grads, max_norm)
    if max_norm > 0:
        for g in grads:
            g.mul_(coef)
    return total


def optimizer_step(params, grads, state, groups, step, lr, betas, eps, max_norm=0.0):
    """One trainer step of the optimiser leg: clip (in place), then Adam per parameter with its group's lr * lr_scale and
    weight decay (optim/base_optimizer.py:8-14, optim/adam.py:186-253).  params / grads / state: dicts by name; state[name] =
    (exp_avg, exp_avg_sq) fp32."""
    total = clip_grad_norm_([grads[n] for n in params], max_norm)
    for n, p in params.items():
        scale, wd = groups[n]
        m, v = state[n]
        adamw_step(p, grads[n], m, v, step, lr * scale, betas[0], betas[1], eps, wd)
    return total


def dp_mean_grads(per_rank_grads):
    """fairseq legacy_distributed_data_parallel.py:76-165: each rank divides by world size, then SUM."""
    w = len(per_rank_grads)
    return sum(g / w for g in per_rank_grads)


def model_dims(sd):
    """Infer (embed_dim, layers) from a state dict (test helper)."""
    H = sd["encoder_wrapper.fusion_model.layers.0.self_attn.q_proj.weight"].shape[0]
    L = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("encoder_wrapper.fusion_model.layers."))
    return SimpleNamespace(embed_dim=H, layers=L)
