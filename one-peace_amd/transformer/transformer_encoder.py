"""Mirror of one_peace/models/transformer/transformer_encoder.py: the modality-shared encoder.

``forward(text_info, image_info, audio_info, return_all_hiddens, encoder_type)`` takes the adapters' tuples
``(x [B,S,H], padding_mask [B,S], bias_list)`` and returns the reference's dict (``encoder_out[0]`` is T x B x C).
bias_list entries may be dense tensors (reference adapters) or ``RelPosSpec`` (our adapters).

MI355X path (bf16; single-modality and joint vl/al streams): activations stay batch-major, the dense
``[B, heads, S, S]`` bias of the reference (transformer_encoder.py:144-162) is never built -- each table (or, for a
joint stream, the block-diagonal pair of tables) becomes one ``[heads, S, Spad]`` image and key padding a ``[B, Spad]``
byte mask -- and every block is two fused HIP functions (attention branch, per-modality FFN branch)."""
import logging
import os

import torch
import torch.nn as nn

from .. import hip, ops
from ..components import FairseqDropout, LayerNorm
from ..adapter.common import PackRowsFn, split_rows
from ..relpos import RelPosSpec, joint_handle
from .transformer_layer import TransformerEncoderLayer

logger = logging.getLogger(__name__)


class LayerDropModuleList(nn.ModuleList):
    """fairseq/modules/layer_drop.py:13-44: skip each layer with probability p while training."""

    def __init__(self, p, modules=None):
        super().__init__(modules)
        self.p = p

    def __iter__(self):
        probs = torch.empty(len(self)).uniform_()
        for i, m in enumerate(super().__iter__()):
            if not self.training or probs[i] > self.p:
                yield m


class TransformerEncoder(nn.Module):
    def __init__(self, cfg, dictionary, use_text_norm, use_image_norm, use_audio_norm):
        super().__init__()
        self.cfg, self.dictionary = cfg, dictionary
        self.register_buffer("version", torch.Tensor([3]))
        self.dropout_module = FairseqDropout(cfg.dropout, module_name=type(self).__name__)
        self.encoder_layerdrop = cfg.layerdrop
        self.max_positions = cfg.max_positions
        self.num_attention_heads = cfg.attention_heads
        self.layers = LayerDropModuleList(p=cfg.layerdrop) if cfg.layerdrop > 0.0 else nn.ModuleList()
        rates = torch.linspace(0, cfg.drop_path_rate, cfg.layers).tolist()
        self.layers.extend(self.build_encoder_layer(cfg, drop_path_rate=r) for r in rates)
        self.num_layers = len(self.layers)
        self.text_layer_norm = LayerNorm(cfg.embed_dim) if cfg.use_text_moe and use_text_norm else None
        self.image_layer_norm = LayerNorm(cfg.embed_dim) if cfg.use_image_moe and use_image_norm else None
        self.audio_layer_norm = LayerNorm(cfg.embed_dim) if cfg.use_audio_moe and use_audio_norm else None
        # Stochastic depth in the lock-step pass: compute a residual branch only for the samples it keeps (the reference multiplies the
        # branch output of the others by zero, transformer_layer.py:78-88; linspace(0, drop_path_rate) over the layers = a fifth of all
        # branch work at 0.4).  Off by default: the default step performs the reference's arithmetic, zeros included.
        self.skip_dropped_branches = os.environ.get("ONEPEACE_SKIP_DROPPED", "0") == "1"
        self.kept_rows_pad = 256  # packed rows per segment are rounded up to whole 256-row GEMM tiles (zero rows)
        # below these drop rates a branch (attention, FFN) keeps the multiplier form: the two packing passes + the zero rows cost more than
        # the dropped samples' share of the branch (headline step: 0.3 ms against 5.7 ms resp. 12 ms per branch, forward + backward)
        # (round 6, with the row tables: thresholds of (0.03, 0.02) and (0.04, 0.02) measured the same 580.6 ms as these over three alternations)
        self.pack_min_drop = tuple(float(v) for v in os.environ.get("ONEPEACE_PACK_MIN_DROP", "0.055,0.03").split(","))

    def build_encoder_layer(self, cfg, drop_path_rate=0.0):
        return TransformerEncoderLayer(cfg, drop_path_rate=drop_path_rate)

    def all_layers(self):
        """Every layer, in order -- iterating `self.layers` itself draws the layerdrop mask of a training pass (LayerDropModuleList)."""
        return list(self.layers._modules.values())

    # ------------------------------------------------------------------------------------------------------
    def forward(self, text_info, image_info, audio_info, return_all_hiddens: bool = False, encoder_type=None):
        streams = {"text": ("text",), "image": ("image",), "audio": ("audio",), "vl": ("text", "image"),
                   "al": ("text", "audio")}.get(encoder_type)
        if streams is None:
            raise NotImplementedError(encoder_type)
        infos = dict(text=text_info, image=image_info, audio=audio_info)
        parts = [infos[s] for s in streams]
        if ops.hip_eligible(parts[0][0]) and self._fused_ok(encoder_type, parts):
            return self._forward_fused(encoder_type, streams, parts, return_all_hiddens)
        return self._forward_torch(encoder_type, streams, infos, return_all_hiddens)

    def _fused_ok(self, encoder_type, parts):
        n_bias = None
        for x, _, biases in parts:
            if biases is not None:
                for b in biases:  # lazy (table, bucket) specs, or dense per-sample tensors from the masked-token gather
                    if not (isinstance(b, RelPosSpec) or (torch.is_tensor(b) and b.dim() == 4 and b.is_cuda and b.dtype == x.dtype)):
                        return False
                if n_bias is not None and len(biases) != n_bias:
                    return False
                n_bias = len(biases)
        if len(parts) > 1 and any(p[0].dtype != parts[0][0].dtype or not p[0].is_cuda for p in parts):
            return False
        return all(getattr(layer, "fused_supported", lambda e: False)(encoder_type) for layer in self.all_layers())

    def _forward_fused(self, encoder_type, streams, parts, return_all_hiddens=False):
        if (len(parts) == 1 and self.skip_dropped_branches and self.training and self.multi_possible() and not return_all_hiddens
                and max((float(getattr(layer, "drop_path_prob", 0.0)) for layer in self.all_layers()), default=0.0) > 0.0
                and not any(torch.is_tensor(b) or getattr(b, "ids", None) is not None for b in (parts[0][2] or ()))):
            # a single-modality training pass with skip_dropped_branches: the lock-step pass with ONE segment (per row the same
            # arithmetic; it is the form that packs the kept samples of every branch)
            feats = self.forward_multi({encoder_type: parts[0]})[encoder_type]
            return {"encoder_out": [feats.transpose(0, 1)], "encoder_padding_mask": parts[0][1], "text_encoder_states": [],
                    "image_encoder_states": [], "audio_encoder_states": []}
        lens = [p[0].shape[1] for p in parts]
        dense = any(torch.is_tensor(b) for p in parts if p[2] is not None for b in p[2])
        if len(parts) == 1:
            x, pad, biases = parts[0]
            no_pads = getattr(pad, "_all_false", False)
            handles = [(ops.DenseBias(b) if torch.is_tensor(b) else b.handle()) for b in biases] if biases else []
        else:
            x = torch.cat([p[0] for p in parts], dim=1)
            pad = torch.cat([p[1] for p in parts], dim=1)
            no_pads = all(getattr(p[1], "_all_false", False) for p in parts)
            n_bias = max((len(p[2]) for p in parts if p[2] is not None), default=0)
            cache = {}
            if dense:  # per-sample blocks: assemble the block-diagonal bias densely (transformer_encoder.py:144-158)
                handles = [ops.DenseBias(self._dense_joint_bias(parts, lens, i, x)) for i in range(n_bias)]
            else:
                handles = [joint_handle([p[2][i] if p[2] is not None else None for p in parts], lens, cache)
                           for i in range(n_bias)]
        B, S, _ = x.shape
        key_pad = None
        if not no_pads:
            x = x * (~pad).unsqueeze(-1).to(x.dtype)  # transformer_encoder.py:141-142
            # The reference masks padded KEYS only through the bias tensor (masked_fill_ of the assembled bias,
            # transformer_encoder.py:144-162; MultiheadAttention.forward never looks at key_padding_mask): a model without
            # attention bias (the pretraining decoder, use_attn_bias: false) attends to its zeroed pad rows.  Same here.
            if handles:
                key_pad = torch.ones(B, hip.attn_spad(S), dtype=torch.uint8, device=x.device)
                key_pad[:, :S] = pad.to(torch.uint8)
        x = x.contiguous()
        scales = self._draw_path_scales(B, x.device)
        states = {"text": [], "image": [], "audio": []}
        position = {id(layer): i for i, layer in enumerate(self.all_layers())}
        # iterating self.layers draws the layerdrop mask of a training pass (fairseq/modules/layer_drop.py:13-44); like the reference
        # (transformer_encoder.py:164-190) the bias of a per-layer bias list is picked by the running index of the layers that RUN,
        # the stochastic-depth rate is the layer's own
        for idx, layer in enumerate(self.layers):
            h = None if not handles else (handles[0] if len(handles) == 1 else handles[idx])
            x = layer.forward_fused(x, h, key_pad, encoder_type, lens, scales[position[id(layer)]] if scales is not None else None)
            if return_all_hiddens:  # transformer_encoder.py:186-190: every layer's output per modality, T x B x C (views, no copies)
                off = 0
                for s_, n in zip(streams, lens):
                    states[s_].append(x[:, off:off + n].transpose(0, 1))
                    off += n
        if len(parts) == 1:
            norm = getattr(self, encoder_type + "_layer_norm")
            x = norm(x) if norm is not None else x
        else:  # transformer_encoder.py:196-207: each modality's rows through its own final norm
            segs, off = [], 0
            for s, n in zip(streams, lens):
                norm, seg = getattr(self, s + "_layer_norm"), x[:, off:off + n]
                segs.append(norm(seg.contiguous()) if norm is not None else seg)
                off += n
            x = torch.cat(segs, dim=1)
        return {"encoder_out": [x.transpose(0, 1)], "encoder_padding_mask": pad, "text_encoder_states": states["text"],
                "image_encoder_states": states["image"], "audio_encoder_states": states["audio"]}

    # ------------------------------------------------------------------------------------------------------
    def multi_possible(self, device=None, dtype=None):
        """The preconditions of a lock-step pass that do not need the adapters' outputs (ModelWrapper.forward_multi asks BEFORE it runs
        the adapters: otherwise a pass that does not qualify -- CPU, fp32, layerdrop -- ran every adapter twice): the HIP
        path for the activations' device / dtype, no layerdrop in training, fused layers; with the opt-in fp8 FFN (ops.set_fp8_ffn) only
        TRAINING passes whose FFN takes the split form (plain up-projection + op_ln_geglu_fwd: the form the lock-step fp8 launches exist
        in since round 5) -- any other pass runs stream by stream on the fp8 kernels instead of silently in bf16."""
        if device is not None and not (torch.device(device).type == "cuda" and dtype == torch.bfloat16):
            return False
        if self.encoder_layerdrop > 0.0 and self.training:
            return False
        if ops.FP8_FFN:
            l0 = self.all_layers()[0] if len(self.layers) else None
            if not (self.training and torch.is_grad_enabled() and l0 is not None and getattr(l0.cfg, "scale_fc", False) and ops.GEGLU_SPLIT
                    and l0.embed_dim % 256 == 0 and l0.ffn_embed_dim % 256 == 0):
                return False
        return all(any(getattr(layer, "fused_supported", lambda e: False)(m) for m in ("text", "image", "audio")) for layer in self.all_layers())

    def multi_ok(self, infos):
        """Can the single-modality streams `infos` ({modality: (x, pad, biases)}) run as ONE lock-step pass?"""
        if len(infos) < 2 or not self.multi_possible():
            return False
        xs = [p[0] for p in infos.values()]
        if not all(ops.hip_eligible(x) and x.device == xs[0].device for x in xs):
            return False
        return all(self._fused_ok(m, [p]) and not any(torch.is_tensor(b) or getattr(b, "ids", None) is not None for b in (p[2] or ()))
                   for m, p in infos.items())

    def forward_multi(self, infos):
        """The single-modality passes of `infos` ({modality: adapter tuple (x [B,S,H], pad [B,S], bias list)}) advanced through the
        layers in LOCK-STEP on one packed activation matrix [sum B*S, H]: what the reference computes with one forward per
        modality (image_text_pretrain_loss.py:76-105 calls the model once per stream), with every modality-shared GEMM / LayerNorm
        of the attention branch launched once over all rows and the per-modality FFN weights applied to their own row ranges.
        Returns {modality: features [B, S, H]} -- per row the same arithmetic as `forward(..., encoder_type=modality)`."""
        segs, xs, row0, samples = [], [], 0, 0
        per_layer = []
        for m, (x, pad, biases) in infos.items():
            B, S, H = x.shape
            handles = [b.handle() for b in biases] if biases else []
            key_pad = None
            if not getattr(pad, "_all_false", False):
                x = x * (~pad).unsqueeze(-1).to(x.dtype)  # transformer_encoder.py:141-142
                if handles:
                    key_pad = torch.ones(B, hip.attn_spad(S), dtype=torch.uint8, device=x.device)
                    key_pad[:, :S] = pad.to(torch.uint8)
            xs.append(x.reshape(B * S, H))
            segs.append((m, B, S, row0, key_pad, samples))
            per_layer.append(handles)
            row0 += B * S
            samples += B
        assert all(xi.dtype == xs[0].dtype for xi in xs), "forward_multi: the streams of a lock-step pass share one dtype (bf16 on the HIP path)"
        x2 = PackRowsFn.apply(*xs)  # (slice copies, not torch.cat: see adapter.common.prepend_token)
        dev = x2.device
        packable = all(getattr(h, "ids", None) is None and not isinstance(h, ops.DenseBias) for hs in per_layer for h in hs)
        if self.skip_dropped_branches and packable:
            scales, plans = self._draw_kept_plans(segs, samples, row0, dev)
        else:
            scales, plans = self._draw_path_scales(samples, dev), None
        row2sample = None
        if scales is not None:
            row2sample = torch.cat([torch.arange(B, device=dev).repeat_interleave(S) + s0 for (_, B, S, _, _, s0) in segs])
        for idx, layer in enumerate(self.layers):
            lsegs = []
            for (m, B, S, r0, key_pad, _), handles in zip(segs, per_layer):
                h = None if not handles else (handles[0] if len(handles) == 1 else handles[idx])
                lsegs.append(ops.StreamSeg(m, B, S, r0, h, key_pad))
            ps1 = ps2 = None
            if scales is not None:
                ps1, ps2 = scales[idx]
            ps1_rows = ps1.index_select(0, row2sample) if ps1 is not None else None
            ps2s = [ps2[s0:s0 + B] if ps2 is not None else None for (_, B, _, _, _, s0) in segs]
            x2 = layer.forward_fused_multi(x2, lsegs, ps1_rows, ps2s, kept=plans[idx] if plans is not None else (None, None))
        out = {}
        pieces = split_rows(x2, [(r0, r0 + B * S) for (_, B, S, r0, _, _) in segs])
        for (m, B, S, _, _, _), rows in zip(segs, pieces):
            norm = getattr(self, m + "_layer_norm")
            out[m] = (norm(rows) if norm is not None else rows).view(B, S, -1)
        return out

    def _draw_kept_plans(self, segs, samples, rows, device):
        """Stochastic depth for `skip_dropped_branches`: the Bernoulli masks of the whole stack are drawn on the HOST (the packed
        row counts size every launch of a layer, and a device-side draw would cost a synchronisation per step), per layer and branch
        the kept sample numbers of every segment go to the device in ONE int32 copy, and each branch gets a hip.KeptRows.
        A branch in which some segment keeps no sample at all falls back to the multiplier form (scales).  Returns
        (scales or None, [(KeptRows | None, KeptRows | None)] per layer) -- (None, None) outside training / without drop-path."""
        if not self.training:
            return None, None
        probs = [float(getattr(layer, "drop_path_prob", 0.0)) for layer in self.all_layers()]
        if max(probs, default=0.0) <= 0.0:
            return None, None
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("skip_dropped_branches sizes its launches by the masks of the step: it cannot be captured in a hipGraph "
                               "(graphs.TrainStepGraph needs the default multiplier form)")
        mask = self._draw_keep_mask(probs, samples)
        todo, dense = [], {}
        for i, p in enumerate(probs):
            for b in range(2):
                if p <= 0.0:
                    continue
                per_seg = [(r0, S, B, torch.nonzero(mask[i, b, s0:s0 + B]).flatten().tolist()) for (_, B, S, r0, _, s0) in segs]
                if p >= self.pack_min_drop[b] and all(len(t[3]) > 0 for t in per_seg):
                    todo.append(((i, b), per_seg))
                else:
                    dense[(i, b)] = mask[i, b].float() / (1.0 - p)
        lists, bases = hip.pack_kept_lists([t[1] for t in todo])
        if device.type == "cuda":  # pinned staging + asynchronous copies: a pageable-memory copy would make the host wait for the stream
            lists = lists.pin_memory().to(device, non_blocking=True)
            if dense:
                keys = sorted(dense)
                allps = torch.stack([dense[k] for k in keys]).pin_memory().to(device, non_blocking=True)
                dense = {k: allps[j] for j, k in enumerate(keys)}
        else:
            lists = lists.to(device)
            dense = {k: v.to(device) for k, v in dense.items()}
        kept = {key: hip.KeptRows(per_seg, lists, base, rows, 1.0 / (1.0 - probs[key[0]]), pad=self.kept_rows_pad)
                for (key, per_seg), base in zip(todo, bases)}
        plans = [(kept.get((i, 0)), kept.get((i, 1))) for i in range(len(probs))]
        scales = [(dense.get((i, 0)), dense.get((i, 1))) for i in range(len(probs))] if dense else None
        return scales, plans

    @staticmethod
    def _draw_keep_mask(probs, samples):
        """bool [layers, 2 branches, samples] on the host: True = the sample keeps the branch (CPU generator: reproducible under
        torch.manual_seed)."""
        keep = 1.0 - torch.tensor(probs, dtype=torch.float32).view(-1, 1, 1)
        return torch.bernoulli(keep.expand(-1, 2, samples)).bool()

    def _draw_path_scales(self, B, device):
        """Per-sample stochastic-depth multipliers of the whole stack in ONE draw (transformer_layer.py:78-85 draws a fresh
        Bernoulli mask per residual branch: 2 per layer; same distribution, 2 launches instead of 4 per layer)."""
        if not self.training:
            return None
        probs = [float(getattr(layer, "drop_path_prob", 0.0)) for layer in self.all_layers()]
        if max(probs, default=0.0) <= 0.0:
            return None
        cached = getattr(self, "_keep_probs", None)  # device-resident: no host-to-device copy per forward (hipGraph capture)
        if cached is None or cached[0] != (tuple(probs), device):
            cached = ((tuple(probs), device), 1.0 - torch.tensor(probs, dtype=torch.float32, device=device).view(-1, 1, 1))
            self._keep_probs = cached
        keep = cached[1]
        draw = torch.bernoulli(keep.expand(-1, 2, B)) / keep
        return [(None, None) if p <= 0.0 else (draw[i, 0], draw[i, 1]) for i, p in enumerate(probs)]

    def _dense_joint_bias(self, parts, lens, i, x):
        B, S = x.shape[0], sum(lens)
        full = x.new_zeros(B, self.num_attention_heads, S, S)
        off = 0
        for p, n in zip(parts, lens):
            if p[2] is not None:
                blk = p[2][i]
                full[:, :, off:off + n, off:off + n] += blk.dense(B).to(x.dtype) if isinstance(blk, RelPosSpec) else blk
            off += n
        return full

    def _forward_torch(self, encoder_type, streams, infos, return_all_hiddens):
        parts = [infos[s] for s in streams]
        lens = {s: infos[s][0].size(1) for s in streams}
        x = torch.cat([p[0] for p in parts], dim=1) if len(parts) > 1 else parts[0][0]
        pad = torch.cat([p[1] for p in parts], dim=1) if len(parts) > 1 else parts[0][1]
        n_bias = len(parts[0][2]) if parts[0][2] is not None else 0
        has_pads = bool(pad.any())
        if has_pads:
            x = x * (1 - pad.unsqueeze(-1).type_as(x))
        B, S, _ = x.shape
        dense = []
        for i in range(n_bias):
            full = x.new_zeros(B, self.num_attention_heads, S, S)
            off = 0
            for s, p in zip(streams, parts):
                n = lens[s]
                if p[2] is not None:
                    blk = p[2][i]
                    full[:, :, off:off + n, off:off + n] += blk.dense(B).to(x.dtype) if isinstance(blk, RelPosSpec) else blk
                off += n
            if has_pads:
                full = full.masked_fill(pad.view(B, 1, 1, S), float("-inf"))
            dense.append(full)
        x = x.transpose(0, 1)
        states = {"text": [], "image": [], "audio": []}
        tl, il, al = lens.get("text", 0), lens.get("image", 0), lens.get("audio", 0)
        for idx, layer in enumerate(self.layers):
            bias = None if not dense else (dense[0] if len(dense) == 1 else dense[idx])
            x = layer(x, encoder_padding_mask=pad, self_attn_bias=bias, encoder_type=encoder_type, text_seq_len=tl,
                      image_seq_len=il, audio_seq_len=al)
            if return_all_hiddens:
                off = 0
                for s in streams:
                    states[s].append(x[off:off + lens[s]])
                    off += lens[s]

        def final(t, s):
            norm = getattr(self, s + "_layer_norm")
            return norm(t) if norm is not None else t

        if len(streams) == 1:
            x = final(x, streams[0])
        else:
            x = torch.cat([final(x[:tl], "text"), final(x[-lens[streams[1]]:], streams[1])], dim=0)
        return {"encoder_out": [x], "encoder_padding_mask": pad, "text_encoder_states": states["text"],
                "image_encoder_states": states["image"], "audio_encoder_states": states["audio"]}

    def upgrade_state_dict_named(self, state_dict, name):
        for i, layer in enumerate(self.all_layers()):  # (by index, transformer_encoder.py:240-244: not through the layerdrop iterator)
            layer.upgrade_state_dict_named(state_dict, "%s.layers.%d" % (name, i))
        prefix = name + "." if name != "" else ""
        for k, v in self.state_dict().items():
            if prefix + k not in state_dict:
                logger.info("%s not exists, re-initialized", prefix + k)
                state_dict[prefix + k] = v
        return state_dict
