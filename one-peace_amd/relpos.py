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
    """table: nn.Embedding weight [num_rel, heads]; bucket: long [S, S] (already sliced to the sequence)."""

    def __init__(self, table, bucket, bucket_i32=None):
        self.table, self.bucket, self.bucket_i32 = table, bucket, bucket_i32

    def dense(self, bsz):
        v = self.table[self.bucket]  # [S, S, heads]
        return v.permute(2, 0, 1).unsqueeze(0).expand(bsz, -1, -1, -1)

    def handle(self):
        S = self.bucket.shape[0]
        b32 = self.bucket_i32 if self.bucket_i32 is not None else self.bucket.to(torch.int32).contiguous()
        return ops.RelPosBias(self.table, b32, S)


def joint_handle(specs, lens, bucket_cache=None):
    """Block-diagonal bias of a joint stream (transformer_encoder.py:144-158: zeros, then each modality's block added on
    its own diagonal square) as ONE table + bucket: rows of the per-modality tables stacked, plus a constant zero row
    that every cross-modal (off-diagonal) position points at.  ``torch.cat`` routes the table gradient back to each
    modality's own embedding.  specs: RelPosSpec or None per segment; lens: segment lengths."""
    live = [sp for sp in specs if sp is not None]
    ref = live[0].table
    S = sum(lens)
    key = tuple(id(sp.bucket) if sp is not None else None for sp in specs) + tuple(lens)
    bucket = bucket_cache.get(key) if bucket_cache is not None else None
    n_rows = sum(sp.table.shape[0] for sp in live)
    if bucket is None:
        bucket = torch.full((S, S), n_rows, dtype=torch.int32, device=ref.device)
        off = base = 0
        for sp, n in zip(specs, lens):
            if sp is not None:
                bucket[off:off + n, off:off + n] = sp.bucket.to(torch.int32) + base
                base += sp.table.shape[0]
            off += n
        if bucket_cache is not None:
            bucket_cache[key] = bucket
    table = torch.cat([sp.table for sp in live] + [ref.new_zeros(1, ref.shape[1])], dim=0)
    return ops.RelPosBias(table, bucket, S)
