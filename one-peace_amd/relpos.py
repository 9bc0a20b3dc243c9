"""Relative-position buckets and the lazy bias object the adapters hand to the encoder.

The reference adapters return dense ``[B, heads, S, S]`` tensors (adapter/image.py:164-171).  Here an adapter
returns ``RelPosSpec`` objects instead: the encoder turns them into the per-table ``[heads, S, Spad]`` image for the
HIP attention kernels (``joint_handle`` for the block-diagonal bias of a joint vl/al stream), or into the dense tensor
when it has to run the torch ops (CPU, fp32)."""
import math

import torch

from . import ops


def make_image_bucket_position(bucket_size, num_relative_distance):
    """2-D relative index over a bucket_size x bucket_size grid + 3 CLS buckets (adapter/image.py:19-34)."""
    n = bucket_size
    cell = torch.arange(n * n)
    row, col = torch.div(cell, n, rounding_mode="floor"), cell % n
    d_row = row.unsqueeze(1) - row.unsqueeze(0) + (n - 1)
    d_col = col.unsqueeze(1) - col.unsqueeze(0) + (n - 1)
    idx = torch.empty(n * n + 1, n * n + 1, dtype=torch.long)
    idx[1:, 1:] = d_row * (2 * n - 1) + d_col
    idx[0, :] = num_relative_distance - 3
    idx[:, 0] = num_relative_distance - 2
    idx[0, 0] = num_relative_distance - 1
    return idx


def make_token_bucket_position(bucket_size, max_position=1024):
    """T5-style log bucketing of (i - j) (adapter/text.py:18-29, adapter/audio.py:20-32)."""
    ids = torch.arange(max_position, dtype=torch.long)
    rel = ids.unsqueeze(1) - ids.unsqueeze(0)
    mid = bucket_size // 2
    dist = torch.where(rel.abs() < mid, torch.full_like(rel, mid - 1), rel.abs())
    growth = torch.log(dist / mid) / math.log((max_position - 1) / mid) * (mid - 1)
    far = (mid + torch.ceil(growth).long()) * torch.sign(rel)
    return torch.where(dist <= mid, rel, far).long() + bucket_size - 1


def add_cls_buckets(rp_bucket, num_rel_dis):
    """adapter/text.py:65-67: row 0 / column 0 / corner get three dedicated buckets."""
    rp_bucket[0, :] = num_rel_dis
    rp_bucket[:, 0] = num_rel_dis + 1
    rp_bucket[0, 0] = num_rel_dis + 2
    return rp_bucket


class RelPosSpec:
    """table: nn.Embedding weight [num_rel, heads]; bucket: long [S, S] (already sliced to the sequence).
    ids (optional): long [B, K] position ids of the K tokens each sample keeps (masked pretraining,
    adapter/image.py:188-204,229-246): the bias of sample b is then rows AND columns ids[b] of table[bucket]."""

    def __init__(self, table, bucket, bucket_i32=None, ids=None):
        self.table, self.bucket, self.bucket_i32, self.ids = table, bucket, bucket_i32, ids

    def with_ids(self, ids):
        return RelPosSpec(self.table, self.bucket, self.bucket_i32, ids)

    def dense(self, bsz):
        v = self.table[self.bucket]  # [S, S, heads]
        d = v.permute(2, 0, 1).unsqueeze(0).expand(bsz, -1, -1, -1)
        if self.ids is None:
            return d
        heads, full, k = d.size(1), d.size(-1), self.ids.size(1)  # the reference's two gathers (adapter/image.py:196-200)
        rows = torch.gather(d, 2, self.ids[:, None, :, None].expand(-1, heads, -1, full))
        return torch.gather(rows, 3, self.ids[:, None, None, :].expand(-1, heads, k, -1))

    def _b32(self):
        return self.bucket_i32 if self.bucket_i32 is not None else self.bucket.to(torch.int32).contiguous()

    def handle(self):
        if self.ids is not None:
            return ops.RelPosBias(self.table, self._b32(), self.ids.size(1), ids=self.ids.to(torch.int32).contiguous())
        return ops.RelPosBias(self.table, self._b32(), self.bucket.shape[0])


def joint_handle(specs, lens, bucket_cache=None):
    """Block-diagonal bias of a joint stream (transformer_encoder.py:144-158: zeros, then each modality's block added on
    its own diagonal square) as ONE table + bucket: rows of the per-modality tables stacked, plus a constant zero row
    that every cross-modal (off-diagonal) position points at.  ``torch.cat`` routes the table gradient back to each
    modality's own embedding.  specs: RelPosSpec or None per segment; lens: segment lengths.
    Segments with per-sample position ids (masked students): the joint bucket is built over the FULL positions of every
    segment and the ids are concatenated with the segments' offsets -- the joint stream is then one ids-indexed bias."""
    live = [sp for sp in specs if sp is not None]
    ref = live[0].table
    with_ids = any(sp is not None and sp.ids is not None for sp in specs)
    full = [(sp.bucket.shape[0] if sp is not None else n) for sp, n in zip(specs, lens)] if with_ids else list(lens)
    S = sum(full)
    key = tuple(id(sp.bucket) if sp is not None else None for sp in specs) + tuple(full)
    bucket = bucket_cache.get(key) if bucket_cache is not None else None
    n_rows = sum(sp.table.shape[0] for sp in live)
    if bucket is None:
        bucket = torch.full((S, S), n_rows, dtype=torch.int32, device=ref.device)
        off = base = 0
        for sp, n in zip(specs, full):
            if sp is not None:
                bucket[off:off + n, off:off + n] = sp.bucket.to(torch.int32) + base
                base += sp.table.shape[0]
            off += n
        if bucket_cache is not None:
            bucket_cache[key] = bucket
    table = torch.cat([sp.table for sp in live] + [ref.new_zeros(1, ref.shape[1])], dim=0)
    if not with_ids:
        return ops.RelPosBias(table, bucket, S)
    B = next(sp.ids.shape[0] for sp in live if sp.ids is not None)
    parts, off = [], 0
    for sp, n_full, n in zip(specs, full, lens):
        if sp is not None and sp.ids is not None:
            parts.append(sp.ids.to(torch.int32) + off)
        else:  # a segment that keeps all its positions (or has no bias): its own positions in order
            parts.append(torch.arange(off, off + n, dtype=torch.int32, device=ref.device).unsqueeze(0).expand(B, -1))
        off += n_full
    ids = torch.cat(parts, dim=1).contiguous()
    return ops.RelPosBias(table, bucket, ids.shape[1], ids=ids)
