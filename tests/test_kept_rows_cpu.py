"""Host logic of "stochastic depth on the kept samples only" (TransformerEncoder.skip_dropped_branches): the per-branch row packing
tables (hip.KeptRows / pack_kept_lists) and the encoder's plan for a whole stack.  No GPU, no library call."""
import torch

from one_peace_amd import hip
from one_peace_amd.transformer.transformer_encoder import TransformerEncoder
from one_peace_amd.unify_model_config import one_peace_encoder_config


def test_kept_rows_layout():
    segs = [(0, 64, 4, [1, 3]), (256, 257, 4, [0, 1, 2, 3]), (256 + 4 * 257, 250, 4, [2])]
    lists, bases = hip.pack_kept_lists([segs, segs])
    assert bases == [0, len(lists) // 2]
    kr = hip.KeptRows(segs, lists, bases[1], 256 + 4 * 257 + 4 * 250, 1.0 / 0.8, pad=256)
    assert kr.n_kept == [2, 4, 1]
    assert kr.dst_rows == [256, 1280, 256] and kr.dst_row0 == [0, 256, 1536] and kr.total == 1792
    assert all(r % 64 == 0 for r in kr.dst_rows)                       # K of every weight-gradient GEMM
    base = bases[1]
    assert lists[kr.off_kept[0]:kr.off_kept[0] + 2].tolist() == [1, 3]
    assert lists[kr.off_inv[0]:kr.off_inv[0] + 4].tolist() == [-1, 0, -1, 1]
    assert lists[kr.off_inv[1]:kr.off_inv[1] + 4].tolist() == [0, 1, 2, 3]
    assert lists[kr.off_inv[2]:kr.off_inv[2] + 4].tolist() == [-1, -1, 0, -1]
    assert kr.off_kept[0] == base and kr.list_end == len(lists)
    assert kr.kept_list(2).tolist() == [2]


def test_encoder_plans_cover_every_branch_with_drop_path():
    cfg = one_peace_encoder_config(embed_dim=128, ffn_embed_dim=256, layers=4, attention_heads=2, drop_path_rate=0.3)
    enc = TransformerEncoder(cfg, None, True, True, True).train()
    B = 5
    segs = [("text", B, 16, 0, None, 0), ("image", B, 17, B * 16, None, B), ("audio", B, 25, B * 33, None, 2 * B)]
    rows = B * (16 + 17 + 25)
    mask = torch.ones(4, 2, 3 * B, dtype=torch.bool)
    mask[1, 0, 0] = False
    mask[2, 1, B:2 * B] = False          # a whole segment dropped: that branch falls back to multipliers
    mask[3, 0, [1, 7, 8, 14]] = False
    enc._draw_keep_mask = lambda probs, n: mask.clone()
    scales, plans = enc._draw_kept_plans(segs, 3 * B, rows, torch.device("cpu"))
    assert plans[0] == (None, None)                                     # linspace(0, 0.3, 4)[0] = 0: no drop-path in layer 0
    assert plans[2][1] is None and scales[2][1] is not None and scales[2][0] is None
    assert torch.allclose(scales[2][1], mask[2, 1].float() / (1 - 0.2))
    k = plans[3][0]
    assert k.n_kept == [4, 3, 4] and abs(k.scale - 1 / 0.7) < 1e-6 and k.full_rows == rows
    assert k.kept_list(1).tolist() == [0, 1, 4]
    assert all(r % 256 == 0 for r in k.dst_rows)
    k = plans[1][0]
    assert k.n_kept == [4, 5, 5]
    # a branch whose drop rate is below the break-even of packing keeps the multiplier form
    enc.pack_min_drop = (0.15, 0.05)          # layer 1 has p = 0.1: its attention branch stays dense, its FFN branch packs
    scales, plans = enc._draw_kept_plans(segs, 3 * B, rows, torch.device("cpu"))
    assert plans[1][0] is None and plans[1][1] is not None
    assert torch.allclose(scales[1][0], mask[1, 0].float() / 0.9) and scales[1][1] is None
    enc.eval()
    assert enc._draw_kept_plans(segs, 3 * B, rows, torch.device("cpu")) == (None, None)


def test_qkv_launch_is_cut_to_whole_rounds_only_when_a_small_second_launch_saves_one():
    """ops._qkv_rows_that_fill_whole_rounds (round 5): rows of the main q|k|v launch such that its 256 x 256 tiles are whole rounds of 256
    workgroups -- only when that saves a round and leaves at most 512 rows for the second launch."""
    from one_peace_amd import ops
    f = ops._qkv_rows_that_fill_whole_rounds
    assert f(73088, 4608) == 72704          # headline: 286 x 18 = 5148 tiles = 20.1 rounds -> 284 row tiles (20 rounds) + 384 rows
    assert f(73088, 1536) == 73088          # 1716 tiles = 6.7 rounds: 30 row tiles would be left over
    assert f(72704, 4608) == 72704          # already whole rounds
    assert f(32896, 12288) == 32768         # 129 x 48 = 24.2 rounds -> 128 row tiles + the 128-row tail the planner splits off anyway
    assert f(2048, 4608) == 2048            # less than one round: nothing to cut
    for rows in range(256, 80000, 1777):
        for n in (1536, 4608, 12288):
            full = f(rows, n)
            assert full == rows or (full % 256 == 0 and 0 < rows - full <= 512 and (full // 256) * (n // 256) % 256 <= 255)
            if full != rows:  # the main launch is within its last whole round, and one round fewer than the unsplit launch
                tiles, tiles_all = (full // 256) * (n // 256), -(-rows // 256) * (n // 256)
                assert -(-tiles // 256) == -(-tiles_all // 256) - 1
