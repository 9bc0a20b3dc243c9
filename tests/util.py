"""Shared helpers for the parity tests."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def out_dir():
    os.makedirs(OUT, exist_ok=True)
    return OUT


def rel_fro(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def max_rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


# Stated floating-point tolerances (bf16 storage, fp32 accumulation) against the fp32 oracle:
#   single fused op with one bf16 rounding of the output:  rel-Frobenius <= 4e-3, max-abs/max|ref| <= 1.2e-2
#   (bf16 has 8 significand bits: worst-case rounding 2^-9 = 1.95e-3 relative, RMS ~1.1e-3)
BF16_FRO = 4e-3
BF16_MAX = 1.2e-2


def assert_close(got, ref, fro=BF16_FRO, mx=BF16_MAX, what=""):
    f, m = rel_fro(got, ref), max_rel(got, ref)
    assert f <= fro and m <= mx, "%s: rel-fro %.3e (<= %.1e)  max-rel %.3e (<= %.1e)" % (what, f, fro, m, mx)
    return f, m


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)
