"""Pieces shared by the three modality adapters (masked-token gather/scatter of the pretraining objective and
state-dict upgrades).  Reference: adapter/{text,image,audio}.py gather_features / preserve_embed branches."""
import logging

import torch

from ..relpos import RelPosSpec

logger = logging.getLogger(__name__)


class _PrependTokenFn(torch.autograd.Function):
    """prepend_token with its own backward: autograd's CopySlices node of an `out[:, 1:] = body` assignment CLONES the incoming gradient
    (two 100 MB copies per adapter call in the headline step); here the body's gradient is a view of it."""

    @staticmethod
    def forward(ctx, tok, body):
        B, S, H = body.shape
        out = body.new_empty(B, S + 1, H, dtype=torch.result_type(tok, body))
        out[:, :1] = tok.expand(B, -1, -1)
        out[:, 1:] = body
        ctx.tok_shape, ctx.tok_dtype, ctx.body_dtype = tok.shape, tok.dtype, body.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        d_tok = g[:, :1].to(ctx.tok_dtype).sum_to_size(ctx.tok_shape) if ctx.needs_input_grad[0] else None
        d_body = g[:, 1:].to(ctx.body_dtype) if ctx.needs_input_grad[1] else None
        return d_tok, d_body


def prepend_token(tok, body):
    """[tok | body] along the sequence: tok [1, 1, H] (a parameter, broadcast over the batch), body [B, S, H] -> [B, S + 1, H].
    Same values and gradients as torch.cat([tok.expand(B, -1, -1), body], dim=1) (adapter/text.py:110-113 and its siblings); two slice
    copies into one allocation instead -- the batched 2-byte copy kernel behind torch.cat moves these 100 MB matrices at 0.3-0.6 TB/s
    (0.8 ms per call in the headline step, profiles/r5_bench_last_step_final2_b128.txt).  The result has torch.cat's PROMOTED dtype
    (an fp32 cls token in front of a bf16 body under autocast gives fp32, as in the reference), not the body's."""
    return _PrependTokenFn.apply(tok, body)


class PackRowsFn(torch.autograd.Function):
    """Row matrices [n_i, H] -> one [sum n_i, H] allocation (what torch.cat(dim=0) returns; slice copies, see prepend_token); the gradients
    are VIEWS of the incoming gradient (slice assignments under autograd clone it once per assignment)."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.rows = [x.shape[0] for x in xs]
        out = xs[0].new_empty(sum(ctx.rows), xs[0].shape[1])
        r = 0
        for x in xs:
            out[r:r + x.shape[0]] = x
            r += x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        out, r = [], 0
        for i, n in enumerate(ctx.rows):
            out.append(g[r:r + n] if ctx.needs_input_grad[i] else None)
            r += n
        return tuple(out)


class SplitRowsFn(torch.autograd.Function):
    """x [rows, H] -> the row ranges x[r0:r1] of `bounds` (views).  Backward writes the pieces' gradients into ONE buffer (autograd's
    slice nodes build a zero-filled full-size matrix per piece and add them up: three fills, three copies and two full-size adds for
    the three streams of the headline step)."""

    @staticmethod
    def forward(ctx, x, *bounds):
        ctx.bounds, ctx.shape = bounds, x.shape
        return tuple(x[r0:r1] for r0, r1 in bounds)

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        out = ref.new_empty(ctx.shape)
        done = 0
        for (r0, r1), g in zip(ctx.bounds, gs):  # (bounds ascending and disjoint: asserted by split_rows)
            if r0 > done:
                out[done:r0].zero_()
            if g is None:
                out[r0:r1].zero_()
            else:
                out[r0:r1] = g
            done = r1
        if done < ctx.shape[0]:
            out[done:].zero_()
        return (out,) + (None,) * len(ctx.bounds)


def split_rows(x, bounds):
    bounds = [tuple(b) for b in bounds]
    assert all(a[1] <= b[0] for a, b in zip(bounds, bounds[1:])) and all(0 <= r0 <= r1 <= x.shape[0] for r0, r1 in bounds)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return tuple(x[r0:r1] for r0, r1 in bounds)
    return SplitRowsFn.apply(x, *bounds)


def take_rows(t, ids):
    """t [B, S, H], ids [B, K] -> t[b, ids[b, k], :]."""
    return torch.gather(t, 1, ids.unsqueeze(-1).expand(-1, -1, t.size(-1)))


def take_bias(bias_list, ids, bsz):
    """Select rows and columns ids[b] of each bias (adapter/image.py:188-204).  A lazy RelPosSpec stays lazy (the HIP path builds
    the per-sample images straight from the table, relpos.RelPosSpec.with_ids); dense tensors are gathered densely."""
    if bias_list is None:
        return None
    out = []
    for bias in bias_list:
        if isinstance(bias, RelPosSpec) and bias.ids is None:
            out.append(bias.with_ids(ids))
            continue
        d = bias.dense(bsz) if isinstance(bias, RelPosSpec) else bias
        heads, full = d.size(1), d.size(-1)
        k = ids.size(1)
        rows = torch.gather(d, 2, ids[:, None, :, None].expand(-1, heads, -1, full))
        out.append(torch.gather(rows, 3, ids[:, None, None, :].expand(-1, heads, k, -1)))
    return out


def scatter_preserved(preserve_ids, preserve_embed, mask_token, bsz, seq_len):
    """Decoder input: mask_token everywhere except positions preserve_ids[b, k] != -1, which get preserve_embed[b, k]."""
    dim = preserve_embed.size(-1)
    flat = mask_token.repeat(bsz * seq_len, 1)
    keep = torch.nonzero(preserve_ids.ne(-1).flatten(), as_tuple=False).flatten()
    dest = (preserve_ids + (torch.arange(bsz, device=preserve_ids.device) * seq_len).unsqueeze(1)).flatten()[keep]
    flat[dest] = preserve_embed.reshape(-1, dim)[keep]
    return flat.view(bsz, seq_len, dim)


def upgrade_rel_pos_tables(module, state_dict, prefix):
    """`rel_pos_table.weight` -> `rel_pos_table_list.0.weight`, and replicate table 0 when per-layer tables are new."""
    old = prefix + "rel_pos_table.weight"
    if old in state_dict:
        state_dict[prefix + "rel_pos_table_list.0.weight"] = state_dict.pop(old)
    tables = getattr(module, "rel_pos_table_list", None)
    if tables is not None and len(tables) > 1 and prefix + "rel_pos_table_list.1.weight" not in state_dict:
        logger.info("copy rel_pos_weight to each layer")
        w = state_dict[prefix + "rel_pos_table_list.0.weight"]
        for i in range(len(tables)):
            state_dict[prefix + "rel_pos_table_list.%d.weight" % i] = w.clone()


def fill_missing(module, state_dict, prefix):
    for k, v in module.state_dict().items():
        if prefix + k not in state_dict:
            logger.info("%s not exists, re-initialized", prefix + k)
            state_dict[prefix + k] = v


class BucketCache:
    """int32, device-resident, contiguous copies of rp_bucket[:S, :S] for the HIP bias-image kernel."""

    def __init__(self):
        self._c = {}

    def get(self, rp_bucket, S):
        key = (S, rp_bucket.device, rp_bucket.data_ptr())
        if key not in self._c:
            self._c = {key: rp_bucket[:S, :S].to(torch.int32).contiguous()}
        return self._c[key]
