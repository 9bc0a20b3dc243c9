"""Measures the gfx950 lane<->element maps the kernels rely on (MFMA fragments, ds_read_b64_tr_b16, LDS-DMA)
by running raw instructions on known register images.  Raw dumps go to gpurun_out/ for offline analysis."""
import os

import numpy as np
import pytest
import torch

from tests.util import out_dir

pytestmark = pytest.mark.gpu


def _hip():
    from one_peace_amd import hip
    return hip


def test_mfma16_fragment_maps():
    hip = _hip()
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-3, 4, (16, 32), generator=g).float()
    B = torch.randint(-3, 4, (32, 16), generator=g).float()
    lanes = torch.arange(64)
    row, kb = lanes & 15, (lanes >> 4) * 8
    a_raw = torch.stack([A[row, kb + e] for e in range(8)], dim=1)          # lane: A[i=l&15][k=(l>>4)*8+e]
    b_raw = torch.stack([B[kb + e, row] for e in range(8)], dim=1)          # lane: B[k][j=l&15]
    # extra runs: one-hot scans for offline decoding
    runs_a, runs_b = [a_raw], [b_raw]
    for la in range(64):
        x = torch.zeros(64, 8); x[la, 0] = 1.0
        runs_a.append(x); runs_b.append(torch.ones(64, 8))
    for lb in range(64):
        x = torch.zeros(64, 8); x[lb, 0] = 1.0
        runs_a.append(torch.ones(64, 8)); runs_b.append(x)
    a = torch.stack(runs_a).to(torch.bfloat16).to(dev).contiguous()
    b = torch.stack(runs_b).to(torch.bfloat16).to(dev).contiguous()
    d = torch.empty(a.shape[0], 64, 4, dtype=torch.float32, device=dev)
    hip._check_probe(hip.probe_lib().op_probe_mfma16(hip.ptr(a), hip.ptr(b), hip.ptr(d), a.shape[0], hip.stream()), "probe_mfma16")
    torch.cuda.synchronize()
    d = d.cpu()
    np.save(os.path.join(out_dir(), "probe_mfma16.npy"), d.numpy())
    D = torch.zeros(16, 16)
    for l in range(64):
        for r in range(4):
            D[(l >> 4) * 4 + r, l & 15] = d[0, l, r]                        # D[i=(l>>4)*4+r][j=l&15]
    assert torch.equal(D, A @ B), "mfma 16x16x32 fragment map differs from the assumed one"


def test_mfma32_fragment_maps():
    hip = _hip()
    dev = "cuda"
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-3, 4, (32, 16), generator=g).float()
    B = torch.randint(-3, 4, (16, 32), generator=g).float()
    lanes = torch.arange(64)
    row, kb = lanes & 31, (lanes >> 5) * 8
    a_raw = torch.stack([A[row, kb + e] for e in range(8)], dim=1)
    b_raw = torch.stack([B[kb + e, row] for e in range(8)], dim=1)
    a = a_raw[None].to(torch.bfloat16).to(dev).contiguous()
    b = b_raw[None].to(torch.bfloat16).to(dev).contiguous()
    d = torch.empty(1, 64, 16, dtype=torch.float32, device=dev)
    hip._check_probe(hip.probe_lib().op_probe_mfma32(hip.ptr(a), hip.ptr(b), hip.ptr(d), 1, hip.stream()), "probe_mfma32")
    torch.cuda.synchronize()
    d = d.cpu()
    np.save(os.path.join(out_dir(), "probe_mfma32.npy"), d.numpy())
    D = torch.zeros(32, 32)
    for l in range(64):
        for r in range(16):
            D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = d[0, l, r]
    assert torch.equal(D, A @ B), "mfma 32x32x16 fragment map differs from the assumed one"


def test_tr16_read_semantics():
    """lane t of a 16-lane group supplies the address of 4 contiguous bf16 X[t][0..3] and must receive
    X[4j + t/4][t%4], j = 0..3  (= column t of the [4][16] block when X rows tile it row-major)."""
    hip = _hip()
    dev = "cuda"
    img = torch.arange(8192, dtype=torch.float32)  # values < 256 exact in bf16 only; use int16 bit patterns instead
    img_bits = torch.arange(8192, dtype=torch.int16)
    img_dev = img_bits.view(torch.bfloat16).to(dev)
    runs = []
    lanes = torch.arange(64)
    g, t = lanes >> 4, lanes & 15
    stride = 80  # elements per row (160 B), as the attention V tile
    # run 0: attention-style block [4 keys][16 d] per group: key0 = g*4, d0 = 0
    runs.append(((g * 4 + (t >> 2)) * stride + (t & 3) * 4) * 2)
    # run 1: dense 64-element blocks per group (guide formula)
    runs.append((g * 64 + t * 4) * 2)
    # run 2: same address in every lane
    runs.append(torch.full((64,), 1000 * 2))
    addr = torch.stack(runs).to(torch.int32).to(dev).contiguous()
    out = torch.empty(len(runs), 64, 4, dtype=torch.bfloat16, device=dev)
    hip._check_probe(hip.probe_lib().op_probe_tr16(hip.ptr(img_dev), hip.ptr(addr), hip.ptr(out), len(runs), hip.stream()), "probe_tr16")
    torch.cuda.synchronize()
    got = out.view(torch.int16).cpu().long()
    np.save(os.path.join(out_dir(), "probe_tr16.npy"), got.numpy())
    # expectation for run 0: lane (g,t) receives V[key0 + j][d0 + t] = img[(g*4 + j)*stride + t]
    exp0 = torch.stack([(g * 4 + j) * stride + t for j in range(4)], dim=1)
    assert torch.equal(got[0], exp0), "ds_read_b64_tr_b16 semantics differ (run 0)\n%s\n%s" % (got[0][:20], exp0[:20])
    exp1 = torch.stack([g * 64 + t + 16 * j for j in range(4)], dim=1)
    assert torch.equal(got[1], exp1), "ds_read_b64_tr_b16 semantics differ (run 1)"


def test_global_load_lds_semantics():
    """LDS destination must be wave-uniform base + lane*16; the global source is per lane."""
    hip = _hip()
    dev = "cuda"
    src = torch.arange(8192, dtype=torch.int16).to(dev)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(3))
    src_off = (perm * 16 * 3).to(torch.int32).to(dev)  # 16-byte chunks, scattered
    dump = torch.empty(8192, dtype=torch.int16, device=dev)
    base = 512
    hip._check_probe(hip.probe_lib().op_probe_glds(hip.ptr(src), hip.ptr(src_off), base, hip.ptr(dump), hip.stream()), "probe_glds")
    torch.cuda.synchronize()
    got = dump.cpu().long()
    np.save(os.path.join(out_dir(), "probe_glds.npy"), got.numpy())
    exp = torch.full((8192,), -1, dtype=torch.long)
    for l in range(64):
        s0 = int(perm[l]) * 16 * 3 // 2
        exp[base // 2 + l * 8: base // 2 + l * 8 + 8] = torch.arange(s0, s0 + 8)
    assert torch.equal(got, exp), "global_load_lds placement differs from base + lane*16"


def test_mfma_rate_probe_reports_a_sane_rate_and_clock():
    """op_probe_mfma_rate (bench.py's roofline.power_limited_peak): a register-only MFMA loop.  A short run says nothing about the
    power limit, only that the accounting is right: the rate can never exceed the data-sheet figure at the measured clock, and a
    full chip of back-to-back MFMAs must get well past half of it."""
    hip = _hip()
    r = hip.mfma_rate_probe(seconds=0.05, waves_per_cu=8, data="zeros")
    mhz = r["mhz"]["mean"]
    assert 500.0 < mhz < 2600.0, r
    at_clock = 2500.0 * mhz / 2400.0
    assert 0.5 * at_clock < r["tflops"] <= 1.02 * at_clock, r
