"""Op-level parity: every C-ABI kernel against the CPU oracle (fp32) on identical bf16-rounded inputs.
Tolerances are the ones stated in tests/util.py."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import onepeace_oracle as O
from tests.util import assert_close, bf16_round, rel_fro

pytestmark = pytest.mark.gpu
DEV = "cuda"


def hipmod():
    from one_peace_amd import hip
    return hip


def rnd(*shape, seed=0, scale=1.0):
    return bf16_round(torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale)


def dev_bf16(t):
    return t.to(torch.bfloat16).to(DEV).contiguous()


@pytest.mark.parametrize("rows,cols", [(5, 64), (37, 384), (130, 512), (257, 768), (300, 1536), (64, 2048), (33, 6144), (9, 8192)])
@pytest.mark.parametrize("gelu", [False, True])
def test_layernorm_fwd_bwd_bf16(rows, cols, gelu):
    hip = hipmod()
    x, w, b, dy = rnd(rows, cols, seed=1, scale=2.0), 1 + 0.1 * rnd(cols, seed=2), 0.1 * rnd(cols, seed=3), rnd(rows, cols, seed=4)
    w, b = bf16_round(w), bf16_round(b)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = O.layer_norm(xr, wr, br)
    if gelu:
        ref = O.gelu_erf(ref)
    ref.backward(dy)
    y, mean, rstd = hip.layernorm_fwd(dev_bf16(x), dev_bf16(w), dev_bf16(b), gelu=gelu)
    assert_close(y, ref, what="ln y")
    assert_close(mean, x.mean(-1), fro=1e-5, mx=1e-5, what="mean") if x.mean(-1).abs().max() > 1e-3 else None
    dx, dw, db = hip.layernorm_bwd(dev_bf16(dy), dev_bf16(x), dev_bf16(w), dev_bf16(b), mean, rstd, gelu=gelu)
    assert_close(dx, xr.grad, what="ln dx")
    assert_close(dw, wr.grad, what="ln dw")
    assert_close(db, br.grad, what="ln db")


def test_layernorm_fp32_no_affine():
    hip = hipmod()
    x = torch.randn(50, 1536, generator=torch.Generator().manual_seed(5))
    dy = torch.randn(50, 1536, generator=torch.Generator().manual_seed(6))
    xr = x.clone().requires_grad_(True)
    ref = O.layer_norm(xr)
    ref.backward(dy)
    y, mean, rstd = hip.layernorm_fwd(x.to(DEV), None, None)
    assert_close(y, ref, fro=1e-6, mx=1e-5, what="ln fp32")
    dx, _, _ = hip.layernorm_bwd(dy.to(DEV), x.to(DEV), None, None, mean, rstd, need_wgrad=False)
    assert_close(dx, xr.grad, fro=1e-5, mx=1e-4, what="ln fp32 dx")


GEMM_SHAPES = [(128, 128, 64), (300, 384, 192), (1000, 1536, 256), (257, 136, 128), (64, 4608, 1536), (4099, 256, 1024)]


@pytest.fixture(params=[1, 2, 3, 4], ids=["tile128", "tile256_bk32", "tile256_bk64", "tile256_bk64_four_waves"])
def tile_mode(request):
    """Run every GEMM test on all tile configurations (auto-selection would pick 128x128 at these small sizes): the
    128x128 kernel, the 256x256 kernel with 64-byte (BK = 32) rows, its full-line (BK = 64) flavour with eight waves of
    128 x 64 and the one with four waves of 128 x 128 (one wave per SIMD)."""
    hip = hipmod()
    old = hip.lib().op_gemm_set_tile(1 if request.param == 1 else 2)
    hip.lib().op_gemm_set_tile({3: 21, 4: 23}.get(request.param, 20))
    yield request.param
    hip.lib().op_gemm_set_tile(22)
    hip.lib().op_gemm_set_tile(old)


@pytest.mark.parametrize("glds", [1, 0])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_bias(M, N, K, glds, tile_mode):
    hip = hipmod()
    hip.lib().op_gemm_set_staging(glds)
    try:
        A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
        ref = A @ W.t() + b
        out = hip.gemm_nt(dev_bf16(A), [dev_bf16(W)], [dev_bf16(b)])
        assert_close(out, ref, what="gemm bias glds=%d" % glds)
        out32 = hip.gemm_nt(dev_bf16(A), [dev_bf16(W)], None, epilogue=hip.EPI_F32,
                            alpha=torch.tensor([0.5], device=DEV))
        assert_close(out32, 0.5 * (A @ W.t()), fro=1e-5, mx=1e-4, what="gemm f32")
    finally:
        hip.lib().op_gemm_set_staging(1)


@pytest.mark.parametrize("M,N,K", [(300, 512, 128), (1000, 1536, 1536), (4099, 256, 3072), (8192, 1536, 6144), (32896, 1536, 192)])
def test_gemm_four_wave_flavour_is_bit_identical_to_eight_waves(M, N, K):
    """Every BK = 64 flavour of the 256x256 kernel accumulates every output element in the same order (k ascending, one MFMA
    per 32 k): bias, residual (+ branch output), GeGLU (+ pre-activations) and fp32 outputs must agree bit for bit, so the
    planner may pick any of them per launch without changing results.  Four-wave kernels: gemm256v (3) and its persistent form
    gemm256p (6; bias / residual epilogues -- the other two fall back to gemm256v).  (Round 3 found accumulator registers read stale behind inline-asm MFMAs in a peeled-loop
    variant of gemm256v: a few registers per wave, deterministic per binary -- this test is what catches that class.)"""
    hip = hipmod()
    a = dev_bf16(rnd(M, K, seed=1))
    w, w1 = dev_bf16(rnd(N, K, seed=2, scale=K ** -0.5)), dev_bf16(rnd(N, K, seed=3, scale=K ** -0.5))
    b, gamma, res = dev_bf16(rnd(N, seed=4)), dev_bf16(rnd(N, seed=5)), dev_bf16(rnd(M, N, seed=6))
    ps = torch.rand((M + 6) // 7, device=DEV)
    outs = {}
    T = hip.TUNE
    try:
        for flavour in ("eight", 3, 6):
            T.reset()
            T.tile_mode = 2
            T.fullline = 1 if flavour == "eight" else 3
            T.sched = 0 if flavour == "eight" else flavour
            h0, h1, y = (torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(3))
            outs[flavour] = [
                hip.gemm_nt(a, [w], [b]),
                hip.gemm_nt(a, [w], None, epilogue=hip.EPI_F32, alpha=torch.tensor([0.25], device=DEV)),
                hip.gemm_nt(a, [w, w1], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1), h0, h1,
                hip.gemm_nt(a, [w], [b], epilogue=hip.EPI_RESID, resid=res, gamma=gamma, rowscale=ps, rows_per_sample=7, h0=y), y,
            ]
            if N % 768 == 0:  # three weight segments, middle one without bias (the fused q|k|v projection)
                seg = N // 3
                outs[flavour].append(hip.gemm_nt(a, [w[:seg], w[seg:2 * seg], w[2 * seg:]], [b[:seg], None, b[:seg]], n_seg=seg, N=N))
        torch.cuda.synchronize()
    finally:
        T.reset()
    for flavour in (3, 6):
        for i, (x, y) in enumerate(zip(outs["eight"], outs[flavour])):
            assert torch.equal(x, y), (flavour, i, float((x.float() - y.float()).abs().max()))
    assert_close(outs[3][0], rnd(M, K, seed=1) @ rnd(N, K, seed=2, scale=K ** -0.5).t() + rnd(N, seed=4), what="four-wave bias")


@pytest.mark.parametrize("Ms", [(300, 257, 64), (2048, 4096, 1000), (512, 8192 + 128, 1250)])
@pytest.mark.parametrize("persistent", [False, True])
def test_gemm_grouped_launch_is_bit_identical_to_separate_launches(Ms, persistent):
    """op_gemm_nt_grouped: the three modality FFNs of a layer (own rows, weights, bias, layer scale, residual, drop-path row
    scales with their own rows-per-sample) as ONE launch, ragged row counts included, against three op_gemm_nt launches."""
    hip = hipmod()
    H, F = 256, 512
    T = hip.TUNE
    xs = [dev_bf16(rnd(m, H, seed=10 + i)) for i, m in enumerate(Ms)]
    w0 = [dev_bf16(rnd(F, H, seed=20 + i, scale=H ** -0.5)) for i in range(3)]
    w1 = [dev_bf16(rnd(F, H, seed=30 + i, scale=H ** -0.5)) for i in range(3)]
    w2 = [dev_bf16(rnd(H, F, seed=40 + i, scale=F ** -0.5)) for i in range(3)]
    b2 = [dev_bf16(rnd(H, seed=50 + i)) for i in range(3)]
    gam = [dev_bf16(rnd(H, seed=60 + i)) for i in range(3)]
    res = [dev_bf16(rnd(m, H, seed=70 + i)) for i, m in enumerate(Ms)]
    rps = [3, 7, 5]
    ps = [torch.rand(m // r + 1, device=DEV) for m, r in zip(Ms, rps)]

    def bf(m, n):
        return torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    try:
        T.reset()
        ref_g, ref_o = [], []
        for i, m in enumerate(Ms):
            h0, h1, y = bf(m, F), bf(m, F), bf(m, H)
            g = hip.gemm_nt(xs[i], [w0[i], w1[i]], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1)
            ref_g.append((g, h0, h1))
            o = hip.gemm_nt(g, [w2[i]], [b2[i]], epilogue=hip.EPI_RESID, resid=res[i], gamma=gam[i], rowscale=ps[i], rows_per_sample=rps[i],
                            h0=y, splitk=False)
            ref_o.append((o, y, hip.gemm_nt(g, [w2[i]], [b2[i]], splitk=False)))
        T.sched = 6 if persistent else 7  # (op_gemm_nt_grouped: persistent unless 7; production = 0 = persistent since round 3)
        ys = [bf(m, H) for m in Ms]
        # (round 5: the GeGLU epilogue has no grouped form any more -- the product never used it -- and says so instead of launching)
        assert hip.gemm_nt_grouped(xs, list(zip(w0, w1)), epilogue=hip.EPI_GEGLU, h0s=[bf(m, F) for m in Ms], h1s=[bf(m, F) for m in Ms]) is None
        gs = [g for g, _, _ in ref_g]
        os_ = hip.gemm_nt_grouped(gs, w2, biases=b2, epilogue=hip.EPI_RESID, h0s=ys, resids=res, gammas=gam, rowscales=ps, rows_per_sample=rps)
        plain = hip.gemm_nt_grouped(gs, w2, biases=b2)
        torch.cuda.synchronize()
    finally:
        T.reset()
    for i in range(3):
        assert torch.equal(os_[i], ref_o[i][0]) and torch.equal(ys[i], ref_o[i][1]) and torch.equal(plain[i], ref_o[i][2]), i
    assert hip.gemm_nt_grouped([xs[0][:, :72]], [w0[0][:, :72]]) is None  # K % 64 != 0: the caller falls back


@pytest.mark.parametrize("M,N,K", [(1536, 1536, 8192), (384, 256, 16448 - 16448 % 64), (4608, 1536, 4096)])
def test_gemm_split_k_weight_gradient_shapes(M, N, K, tile_mode):
    """Few output tiles + long K: the planner splits K across workgroups (fp32 slabs + reduce)."""
    hip = hipmod()
    A, W = rnd(M, K, seed=1, scale=0.5), rnd(N, K, seed=2, scale=0.5)
    ref = A @ W.t()
    out = hip.gemm_nt(dev_bf16(A), [dev_bf16(W)])
    assert_close(out, ref, what="split-k")
    out1 = hip.gemm_nt(dev_bf16(A), [dev_bf16(W)], splitk=False)
    assert_close(out1, ref, what="no split")


@pytest.mark.parametrize("K,M,N", [(64, 256, 256), (128, 264, 8), (1024, 1536, 512), (4096, 768, 1536), (16448, 1536, 1536),
                                   (2048, 4608, 1536)])
def test_gemm_tn_transpose_read(K, M, N):
    """dW = dy^T x straight from row-major operands (ds_read_b64_tr_b16 fragments), incl. split-K and ragged M/N."""
    hip = hipmod()
    A, Bm = rnd(K, M, seed=1, scale=0.5), rnd(K, N, seed=2, scale=0.5)
    ref = A.t() @ Bm
    out = hip.gemm_tn(dev_bf16(A), dev_bf16(Bm))
    assert_close(out, ref, what="gemm_tn")
    # strided operands (views into wider buffers), as the packed qkv gradient
    wide = dev_bf16(torch.cat([A, A], dim=1))
    out2 = hip.gemm_tn(wide[:, M:], dev_bf16(Bm))
    assert_close(out2, ref, what="gemm_tn strided")


@pytest.mark.parametrize("N", [256, 384])
@pytest.mark.parametrize("rows", [16 * 257, 4000, 96 * 5 + 7, 40])
def test_weight_gradient_with_row_count_not_a_multiple_of_64(rows, N):
    """ops.wgrad: 16 x 257 image tokens (4112 = 64 x 64 + 16) stay on the transpose-read kernel -- the multiple-of-64 part plus
    a one-step launch over a zero-padded copy of the leftover rows -- fresh output and accumulation into an existing gradient,
    also from strided views of the packed qkv gradient; fewer than 64 rows take the transposed-copies path."""
    from one_peace_amd import ops
    hip = hipmod()
    M = 384  # N == M: both operands' leftover rows have the same shape (their pad buffers must still be distinct)
    dy, x = rnd(rows, M, seed=1, scale=0.5), rnd(rows, N, seed=2, scale=0.5)
    ref = dy.t() @ x
    calls = []
    orig = hip.gemm_tn
    hip.gemm_tn = lambda *a, **k: (calls.append(a[0].shape[0]), orig(*a, **k))[1]
    try:
        out = ops.wgrad(dev_bf16(dy), dev_bf16(x))
        assert_close(out, ref, what="wgrad")
        base = rnd(M, N, seed=3)
        acc = dev_bf16(base).clone()
        ops.wgrad(dev_bf16(dy), dev_bf16(x), out=acc, accumulate=True)
        assert_close(acc, base.bfloat16().float() + ref, what="wgrad accumulate")
        wide = dev_bf16(torch.cat([dy, dy, dy], dim=1))
        assert_close(ops.wgrad(wide[:, M:2 * M], dev_bf16(x)), ref, what="wgrad strided")
    finally:
        hip.gemm_tn = orig
    if rows >= 64:
        expect = [rows - rows % 64] + ([64] if rows % 64 else [])
        assert calls == expect * 3, calls
    else:
        assert calls == []


@pytest.mark.parametrize("K,M,N", [(256, 256, 256), (4096, 1536, 512), (16448, 1536, 1536), (8192, 264, 4608)])
def test_gemm_tn_four_wave_flavour_is_bit_identical_to_eight_waves(K, M, N):
    """Weight-gradient kernel: the flavour with four waves of 128 x 128 (auto-selected for <= 108 output tiles at K >= 16384)
    and the one with eight waves of 128 x 64 accumulate every element in the same order -- fresh output, accumulation into an
    existing bf16 gradient and the split-K path must agree bit for bit, and match the fp32 product."""
    hip = hipmod()
    L = hip.lib()
    A, Bm = rnd(K, M, seed=1, scale=0.5), rnd(K, N, seed=2, scale=0.5)
    base = dev_bf16(rnd(M, N, seed=3))
    outs = {}
    try:
        for flavour in (21, 23):
            L.op_gemm_set_tile(flavour)
            acc = base.clone()
            outs[flavour] = (hip.gemm_tn(dev_bf16(A), dev_bf16(Bm)), hip.gemm_tn(dev_bf16(A), dev_bf16(Bm), acc, True))
        torch.cuda.synchronize()
    finally:
        L.op_gemm_set_tile(22)
    for x, y in zip(outs[21], outs[23]):
        assert torch.equal(x, y), float((x.float() - y.float()).abs().max())
    assert_close(outs[23][0], A.t() @ Bm, what="gemm_tn four waves")


@pytest.mark.parametrize("glds", [1, 0])
def test_gemm_three_segments_qkv(glds, tile_mode):
    hip = hipmod()
    hip.lib().op_gemm_set_staging(glds)
    try:
        M, H = 333, 256
        A = rnd(M, H, seed=1)
        Ws = [rnd(H, H, seed=10 + i, scale=H ** -0.5) for i in range(3)]
        bq, bv = rnd(H, seed=20), rnd(H, seed=21)
        ref = torch.cat([A @ Ws[0].t() + bq, A @ Ws[1].t(), A @ Ws[2].t() + bv], dim=1)
        out = hip.gemm_nt(dev_bf16(A), [dev_bf16(w) for w in Ws], [dev_bf16(bq), None, dev_bf16(bv)], n_seg=H, N=3 * H)
        assert_close(out, ref, what="qkv gemm")
    finally:
        hip.lib().op_gemm_set_staging(1)


@pytest.mark.parametrize("glds", [1, 0])
@pytest.mark.parametrize("M,F_,K", [(200, 256, 128), (515, 1024, 256), (130, 6144, 1536)])
def test_gemm_geglu(M, F_, K, glds, tile_mode):
    hip = hipmod()
    hip.lib().op_gemm_set_staging(glds)
    try:
        A, W0, W1 = rnd(M, K, seed=1), rnd(F_, K, seed=2, scale=K ** -0.5), rnd(F_, K, seed=3, scale=K ** -0.5)
        h0r, h1r = A @ W0.t(), A @ W1.t()
        ref = O.gelu_erf(h0r) * h1r
        h0 = torch.empty(M, F_, dtype=torch.bfloat16, device=DEV)
        h1 = torch.empty_like(h0)
        out = hip.gemm_nt(dev_bf16(A), [dev_bf16(W0), dev_bf16(W1)], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1)
        assert_close(out, ref, what="geglu")
        assert_close(h0, h0r, what="h0")
        assert_close(h1, h1r, what="h1")
        out2 = hip.gemm_nt(dev_bf16(A), [dev_bf16(W0), dev_bf16(W1)], epilogue=hip.EPI_GEGLU)
        assert torch.equal(out2, out)
    finally:
        hip.lib().op_gemm_set_staging(1)


@pytest.mark.parametrize("glds", [1, 0])
def test_gemm_residual_epilogue(glds, tile_mode):
    hip = hipmod()
    hip.lib().op_gemm_set_staging(glds)
    try:
        B, S, H, K = 5, 37, 256, 512
        M = B * S
        A, W, b = rnd(M, K, seed=1), rnd(H, K, seed=2, scale=K ** -0.5), rnd(H, seed=3)
        res, gamma = rnd(M, H, seed=4), bf16_round(0.1 + torch.rand(H, generator=torch.Generator().manual_seed(5)))
        rs = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25])
        y = A @ W.t() + b
        ref = res + rs.repeat_interleave(S)[:, None] * gamma * y
        ybuf = torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
        out = hip.gemm_nt(dev_bf16(A), [dev_bf16(W)], [dev_bf16(b)], epilogue=hip.EPI_RESID, resid=dev_bf16(res),
                          gamma=dev_bf16(gamma), rowscale=rs.to(DEV), rows_per_sample=S, h0=ybuf)
        assert_close(out, ref, what="resid")
        assert_close(ybuf, y, what="branch y")
        # in-place accumulate: C = C + A W^T
        C = dev_bf16(res)
        hip.gemm_nt(dev_bf16(A), [dev_bf16(W)], None, out=C, epilogue=hip.EPI_RESID, resid=C)
        assert_close(C, res + A @ W.t(), what="accumulate")
    finally:
        hip.lib().op_gemm_set_staging(1)


@pytest.mark.parametrize("rows,cols", [(64, 64), (257, 1536), (1000, 136), (3, 8)])
def test_transpose(rows, cols):
    hip = hipmod()
    x = rnd(rows, cols, seed=7)
    out = hip.transpose(dev_bf16(x))
    assert torch.equal(out.float().cpu(), x.t())


def test_colsum_and_scale_rows():
    hip = hipmod()
    B, S, N = 6, 50, 1536
    M = B * S
    x, y = rnd(M, N, seed=1), rnd(M, N, seed=2)
    rs = torch.tensor([1.0, 0.0, 2.0, 1.0, 0.5, 1.0])
    mul = rnd(N, seed=3)
    rsm = rs.repeat_interleave(S)[:, None]
    out = hip.colsum(dev_bf16(x), dev_bf16(y), rs.to(DEV), S, dev_bf16(mul))
    assert_close(out, mul * (rsm * x * y).sum(0), what="colsum xy")
    out2 = hip.colsum(dev_bf16(x), out_dtype=torch.float32)
    assert_close(out2, x.sum(0), fro=1e-5, mx=1e-4, what="colsum f32")
    acc = dev_bf16(mul)
    hip.colsum(dev_bf16(x), out=acc, accumulate=True)
    assert_close(acc, mul + x.sum(0), what="colsum acc")
    sr = hip.scale_rows(dev_bf16(x), dev_bf16(mul), rs.to(DEV), S)
    assert_close(sr, rsm * mul * x, what="scale_rows")


@pytest.mark.parametrize("M,N,rps,use_gamma,use_ps,accumulate", [(70, 1536, 7, True, True, False), (1030, 520, 103, True, False, True),
                                                              (33, 64, 3, False, True, False), (4100, 1536, 41, True, True, True)])
def test_resid_bwd_one_pass(M, N, rps, use_gamma, use_ps, accumulate):
    """Layer-scale / drop-path residual backward (transformer_layer.py:70-88,190-196): dbranch, dgamma, dbias together."""
    hip = hipmod()
    dout, y = rnd(M, N, seed=1), rnd(M, N, seed=2)
    gamma = rnd(N, seed=3) if use_gamma else None
    ps = ((torch.arange(M // rps) % 3 != 0).float() / 0.75) if use_ps else None
    rows = ps.repeat_interleave(rps)[:, None] if use_ps else 1.0
    base_g, base_b = rnd(N, seed=4), rnd(N, seed=5)
    ref_dy = dout * rows * (gamma if use_gamma else 1.0)
    ref_dg = (dout * y * rows).sum(0)
    ref_db = ref_dy.sum(0)
    kw = {}
    if accumulate:
        kw = dict(dgamma=dev_bf16(base_g) if use_gamma else None, dbias=dev_bf16(base_b), accumulate=True)
        ref_dg, ref_db = ref_dg + base_g.bfloat16().float(), ref_db + base_b.bfloat16().float()
    else:
        kw = dict(dgamma=True if use_gamma else None, dbias=True)
    dy, dg, db = hip.resid_bwd(dev_bf16(dout), dev_bf16(y) if use_gamma else None, dev_bf16(gamma) if use_gamma else None,
                               ps.to(DEV) if use_ps else None, rps, **kw)
    gq = gamma.bfloat16().float() if use_gamma else 1.0
    assert_close(dy, dout.bfloat16().float() * rows * gq, what="dbranch")
    assert_close(db, ref_db, what="dbias")
    if use_gamma:
        assert_close(dg, ref_dg, what="dgamma")
    else:
        assert dg is None


@pytest.mark.parametrize("accumulate", [False, True])
def test_colsum_segments_qkv_bias_gradients(accumulate):
    hip = hipmod()
    M, H = 777, 192
    x = rnd(M, 3 * H, seed=1)
    ref = x.bfloat16().float().sum(0)
    if accumulate:
        b0, b2 = rnd(H, seed=2), rnd(H, seed=3)
        t0, t2 = dev_bf16(b0), dev_bf16(b2)
        hip.colsum_segments(dev_bf16(x), H, [t0, None, t2], accumulate=True)
        assert_close(t0, ref[:H] + b0.bfloat16().float(), what="seg0")
        assert_close(t2, ref[2 * H:] + b2.bfloat16().float(), what="seg2")
    else:
        outs = hip.colsum_segments(dev_bf16(x), H)
        for i in range(3):
            assert_close(outs[i], ref[i * H:(i + 1) * H], what="seg%d" % i)


@pytest.mark.parametrize("M,F_", [(130, 1024), (37, 256), (70, 6144), (9, 2048), (261, 1536)])
def test_ln_geglu_fwd_fused(M, F_):
    """LayerNorm_F(gelu(h0) * h1) from the two halves of one [M, 2F] matrix: against the oracle's formulas
    (transformer_layer.py:64-67,111-118), and against the unfused pair it replaces (bf16 product, then op_layernorm_fwd): same
    statistics and output up to the few bf16 ulps by which torch's erf and the device's differ."""
    hip = hipmod()
    hh = dev_bf16(torch.cat([rnd(M, F_, seed=2, scale=2.0), rnd(M, F_, seed=3)], 1))
    h0, h1 = hh[:, :F_], hh[:, F_:]
    w, b = dev_bf16(1 + 0.1 * rnd(F_, seed=4)), dev_bf16(0.1 * rnd(F_, seed=5))
    y, mean, rstd = hip.ln_geglu_fwd(h0, h1, w, b)
    g = (O.gelu_erf(h0.float()) * h1.float()).bfloat16()
    y2, mean2, rstd2 = hip.layernorm_fwd(g, w, b, want_stats=True)
    assert_close(y, y2.float().cpu(), fro=1e-3, mx=2e-2, what="vs unfused")
    torch.testing.assert_close(mean, mean2, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(rstd, rstd2, rtol=1e-3, atol=1e-4)
    ref = O.layer_norm(O.gelu_erf(h0.float().cpu()) * h1.float().cpu(), w.float().cpu(), b.float().cpu())
    assert_close(y, ref, what="ln(geglu)")
    # strided inputs of the backward: same result as from contiguous copies
    dy = dev_bf16(rnd(M, F_, seed=1))
    a = hip.ln_geglu_bwd(dy, h0, h1, w, mean, rstd)
    c = hip.ln_geglu_bwd(dy, h0.contiguous(), h1.contiguous(), w, mean, rstd)
    for u, v in zip(a, c):
        assert torch.equal(u, v)


@pytest.mark.parametrize("M,F_", [(130, 1024), (37, 256), (70, 6144), (9, 2048)])
def test_ln_geglu_bwd_fused(M, F_):
    """LayerNorm(F) backward + GeGLU backward in one pass vs autograd through the oracle's formulas
    (transformer_layer.py:64-67,111-118)."""
    hip = hipmod()
    dy, h0, h1 = rnd(M, F_, seed=1), rnd(M, F_, seed=2, scale=2.0), rnd(M, F_, seed=3)
    w, b = 1 + 0.1 * rnd(F_, seed=4), 0.1 * rnd(F_, seed=5)
    q = lambda t: t.bfloat16().float()
    a, c = q(h0).requires_grad_(True), q(h1).requires_grad_(True)
    wr, br = q(w).requires_grad_(True), q(b).requires_grad_(True)
    g = O.gelu_erf(a) * c
    O.layer_norm(g, wr, br).backward(q(dy))
    # forward statistics as the HIP forward produces them (from the bf16-rounded g)
    _, mean, rstd = hip.layernorm_fwd(dev_bf16(g.detach()), dev_bf16(w), dev_bf16(b), want_stats=True)
    dh0, dh1, dw, db = hip.ln_geglu_bwd(dev_bf16(dy), dev_bf16(h0), dev_bf16(h1), dev_bf16(w), mean, rstd)
    assert_close(dh0, a.grad, what="dh0", fro=6e-3, mx=2.5e-2)
    assert_close(dh1, c.grad, what="dh1", fro=6e-3, mx=2.5e-2)
    assert_close(dw, wr.grad, what="dw", fro=8e-3, mx=3e-2)
    assert_close(db, br.grad, what="db")


def test_geglu_bwd():
    hip = hipmod()
    M, F_ = 130, 1024
    dg, h0, h1 = rnd(M, F_, seed=1), rnd(M, F_, seed=2, scale=2.0), rnd(M, F_, seed=3)
    a, b = h0.clone().requires_grad_(True), h1.clone().requires_grad_(True)
    (O.gelu_erf(a) * b).backward(dg)
    d0, d1 = hip.geglu_bwd(dev_bf16(dg), dev_bf16(h0), dev_bf16(h1))
    assert_close(d0, a.grad, what="dh0")
    assert_close(d1, b.grad, what="dh1")


def test_l2norm():
    hip = hipmod()
    x = rnd(19, 1536, seed=1, scale=3.0)
    dy = torch.randn(19, 1536, generator=torch.Generator().manual_seed(2))
    xr = x.clone().requires_grad_(True)
    ref = O.l2_normalize(xr)
    ref.backward(dy)
    y, inv = hip.l2norm_fwd(dev_bf16(x), out_dtype=torch.float32)
    assert_close(y, ref, fro=1e-6, mx=1e-5, what="l2norm")
    dx = hip.l2norm_bwd(dy.to(DEV), y, inv)
    assert_close(dx, xr.grad, what="l2norm dx")


@pytest.mark.parametrize("eps", [0.0, 0.1])
def test_infonce_rows(eps):
    hip = hipmod()
    rows, n, t0 = 16, 48, 16
    sim = 5 * torch.randn(rows, n, generator=torch.Generator().manual_seed(1))
    sr = sim.clone().requires_grad_(True)
    tgt = torch.arange(rows) + t0
    lp = F.log_softmax(sr, dim=-1)
    nll = -lp.gather(-1, tgt[:, None]).squeeze(-1)
    if eps:
        e = eps / (n - 1)
        nll = (1 - eps - e) * nll + e * (-lp.sum(-1))
    nll.sum().backward()
    s = sim.to(DEV).contiguous()
    loss, hit, dot = hip.infonce_rows(s, t0, eps, gscale=1.0)
    assert_close(loss, nll, fro=1e-5, mx=1e-5, what="row loss")
    assert_close(s, sr.grad, fro=1e-5, mx=1e-4, what="dsim")
    assert torch.equal(hit.cpu(), (sim.argmax(1) == tgt).float())
    assert_close(dot, (sr.grad * sim).sum(1), fro=1e-4, mx=1e-4, what="dot")


def test_adamw_matches_reference_rule():
    hip = hipmod()
    n = 4096 + 8
    p0, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.01)
    m = torch.zeros(n); v = torch.zeros(n)
    pd, gd, md, vd = dev_bf16(p0), dev_bf16(g), m.to(DEV), v.to(DEV)
    pr = p0.to(torch.bfloat16)
    for step in (1, 2, 3):
        O.adamw_step(pr, g.to(torch.bfloat16), m, v, step, lr=1e-2, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.05)
        hip.adamw_step(pd, gd, md, vd, 1e-2, 0.9, 0.98, 1e-6, 0.05, step)
    assert_close(md, m, fro=1e-6, mx=1e-5, what="m")
    assert_close(vd, v, fro=1e-6, mx=1e-5, what="v")
    # bf16 parameter: allow 1 ulp on a tiny fraction (fp32 op-order differences before the final rounding)
    diff = (pd.float().cpu() - pr.float()).abs()
    assert (diff > 0).float().mean() < 0.01 and float(diff.max()) <= float(pr.float().abs().max()) * 2 ** -7


@pytest.mark.parametrize("gnorm_scale", [0.01, 30.0])
def test_adamw_with_global_norm_clipping(gnorm_scale):
    """trainer.py:917-935: grads * 1/world, clip_grad_norm(3.0) over ALL parameters (two decay groups here), Adam step."""
    hip = hipmod()
    n = 8192
    p0, g = rnd(n, seed=1), rnd(n, seed=2, scale=gnorm_scale)
    gq = g.to(torch.bfloat16)
    world = 4
    total, coef = O.clip_coef([gq.float() / world], 3.0)
    assert (float(coef) < 1.0) == (gnorm_scale > 1.0)
    m = torch.zeros(n); v = torch.zeros(n)
    pr = p0.to(torch.bfloat16)
    O.adamw_step(pr, (gq.float() / world * coef), m, v, 1, lr=1e-2, beta1=0.9, beta2=0.98, eps=1e-6, weight_decay=0.05)
    pd, gd, md, vd = dev_bf16(p0), dev_bf16(g), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sq = hip.sqnorm(gd)
    assert abs(float(sq.sqrt()) / world - float(total)) <= 1e-4 * float(total)
    for s_, e_ in ((0, 4096), (4096, n)):  # two ranges share the one global norm
        hip.adamw_step(pd[s_:e_], gd[s_:e_], md[s_:e_], vd[s_:e_], 1e-2, 0.9, 0.98, 1e-6, 0.05, 1, 1.0 / world, sq, 3.0)
    assert_close(md, m, fro=1e-5, mx=1e-4, what="m")
    assert_close(vd, v, fro=1e-5, mx=1e-4, what="v")
    diff = (pd.float().cpu() - pr.float()).abs()
    assert (diff > 0).float().mean() < 0.01 and float(diff.max()) <= float(pr.float().abs().max()) * 2 ** -7


def test_relpos_bias_build_and_bwd():
    hip = hipmod()
    heads, n = 3, 4
    num_rel = (2 * n - 1) ** 2 + 3
    bucket = O.image_bucket_position(n, num_rel)
    S = n * n + 1
    Spad = 64
    table = rnd(num_rel, heads, seed=1)
    ref = O.rel_pos_bias(table, bucket)
    out = hip.relpos_bias_build(dev_bf16(table), bucket.to(torch.int32).to(DEV), S, Spad)
    assert torch.equal(out[:, :, :S].float().cpu(), ref)
    outT = hip.relpos_bias_build(dev_bf16(table), bucket.to(torch.int32).to(DEV), S, Spad, transposed=True)
    assert torch.equal(outT[:, :, :S].float().cpu(), ref.transpose(1, 2))
    assert float(out[:, :, S:].abs().max()) == 0.0
    dbias = torch.randn(heads, S, Spad, generator=torch.Generator().manual_seed(2))
    tr = table.clone().requires_grad_(True)
    O.rel_pos_bias(tr, bucket).backward(dbias[:, :, :S])
    dt = hip.relpos_bias_bwd(dbias.to(DEV), bucket.to(torch.int32).to(DEV), num_rel, S, Spad)
    assert_close(dt, tr.grad, fro=1e-5, mx=1e-4, what="dtable")


def test_relpos_bias_build_from_position_ids():
    """Per-sample images of the masked-pretraining passes straight from the table: out[b][h][i][j] =
    table[bucket[ids[b,i]][ids[b,j]]][h] (what the reference obtains by gathering rows and columns of the dense bias,
    adapter/image.py:188-204), its transposed image, zero pad columns, and the scatter of per-sample gradient slabs back
    into the table."""
    hip = hipmod()
    heads, n, B, K = 3, 4, 5, 9
    num_rel = (2 * n - 1) ** 2 + 3
    bucket = O.image_bucket_position(n, num_rel)
    S, Kpad = n * n + 1, 128
    table = rnd(num_rel, heads, seed=1)
    g = torch.Generator().manual_seed(4)
    ids = torch.stack([torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(S - 1, generator=g)[: K - 1].sort().values])
                       for _ in range(B)])
    ids[2, -2:] = K - 1  # padded slots are mapped to a valid position id by the adapters (adapter/image.py:241-243)
    dense = O.rel_pos_bias(table, bucket).unsqueeze(0).expand(B, -1, -1, -1)
    ref = torch.gather(torch.gather(dense, 2, ids[:, None, :, None].expand(-1, heads, -1, S)), 3,
                       ids[:, None, None, :].expand(-1, heads, K, -1))
    b32, i32 = bucket.to(torch.int32).to(DEV), ids.to(torch.int32).to(DEV)
    out = hip.relpos_bias_build_ids(dev_bf16(table), b32, i32, Kpad)
    outT = hip.relpos_bias_build_ids(dev_bf16(table), b32, i32, Kpad, transposed=True)
    assert torch.equal(out[..., :K].float().cpu(), ref) and torch.equal(outT[..., :K].float().cpu(), ref.transpose(2, 3))
    assert float(out[..., K:].abs().max()) == 0.0 and float(outT[..., K:].abs().max()) == 0.0
    dbias = torch.randn(B, heads, K, Kpad, generator=torch.Generator().manual_seed(2))
    tr = table.clone().requires_grad_(True)
    dense_r = O.rel_pos_bias(tr, bucket).unsqueeze(0).expand(B, -1, -1, -1)
    got = torch.gather(torch.gather(dense_r, 2, ids[:, None, :, None].expand(-1, heads, -1, S)), 3,
                       ids[:, None, None, :].expand(-1, heads, K, -1))
    got.backward(dbias[..., :K])
    dt = hip.relpos_bias_bwd_ids(dbias.to(DEV), b32, i32, num_rel)
    assert_close(dt, tr.grad, fro=1e-5, mx=1e-4, what="dtable from per-sample slabs")


def _attn_ref(q, k, v, heads, scale, bias, key_pad):
    """q,k,v: [B,S,H] fp32; bias [heads,S,S]; key_pad [B,S] bool.  multihead_attention.py:102-115."""
    B, S, H = q.shape
    hd = H // heads
    qh, kh, vh = (t.reshape(B, S, heads, hd).transpose(1, 2) for t in (q, k, v))
    s = (qh * scale) @ kh.transpose(-1, -2)
    if bias is not None:
        s = s + (bias[None] if bias.dim() == 3 else bias)
    if key_pad is not None:
        s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, S, H), torch.logsumexp(s, dim=-1)


ATTN_CASES = [(2, 37, 2, True, True), (3, 64, 4, True, True), (2, 257, 3, True, False), (1, 130, 2, False, False),
              (2, 300, 2, False, True), (1, 1025, 1, True, False), (20, 72, 2, True, True)]


def _attn_inputs(B, S, heads, use_bias, use_pad):
    hip = hipmod()
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=1)
    Spad = hip.attn_spad(S)
    bias = rnd(heads, S, S, seed=2) if use_bias else None
    key_pad = None
    if use_pad:
        key_pad = torch.zeros(B, S, dtype=torch.bool)
        for b in range(B):
            key_pad[b, S - 1 - (3 * b) % (S // 2):] = True
        key_pad[0, :] = False
    d = dev_bf16(qkv)
    bias_d = biasT_d = pad_d = None
    if use_bias:
        bias_d = torch.zeros(heads, S, Spad, dtype=torch.bfloat16, device=DEV)
        bias_d[:, :, :S] = bias.to(torch.bfloat16).to(DEV)
        biasT_d = torch.zeros(heads, S, Spad, dtype=torch.bfloat16, device=DEV)
        biasT_d[:, :, :S] = bias.transpose(1, 2).to(torch.bfloat16).to(DEV)
    if use_pad:
        pad_d = torch.ones(B, Spad, dtype=torch.uint8, device=DEV)
        pad_d[:, :S] = key_pad.to(torch.uint8).to(DEV)
    return qkv, bias, key_pad, d, bias_d, biasT_d, pad_d, Spad


@pytest.fixture(params=[1, 0], ids=["dq_dbias_merged", "dq_dbias_separate"])
def merge_dbias(request):
    hip = hipmod()
    old = hip.lib().op_attn_set_merge_dbias(request.param)
    yield request.param
    hip.lib().op_attn_set_merge_dbias(old)


@pytest.fixture(params=[1, 0], ids=["resident", "streaming"])
def resident(request):
    """S <= 320 runs the resident-K/V kernels by default; the streaming kernels (all longer sequences) stay tested at the
    same shapes through the knob."""
    hip = hipmod()
    old = hip.lib().op_attn_set_resident(request.param)
    yield request.param
    hip.lib().op_attn_set_resident(old)


@pytest.mark.parametrize("B,S,heads,use_bias,use_pad", ATTN_CASES + [(5, 327, 2, True, True), (3, 384, 1, True, False),
                                                                     (3, 250, 2, True, True), (2, 272, 1, False, True),
                                                                     (4, 100, 2, True, False), (2, 16, 1, True, True),
                                                                     (3, 257, 2, True, False), (2, 320, 2, True, True),
                                                                     (2, 289, 1, False, False), (7, 64, 3, True, True)])
def test_attention_forward_backward(B, S, heads, use_bias, use_pad, merge_dbias, resident):
    hip = hipmod()
    H = heads * 64
    qkv, bias, key_pad, d, bias_d, biasT_d, pad_d, Spad = _attn_inputs(B, S, heads, use_bias, use_pad)
    qkv_r = qkv.clone().requires_grad_(True)
    bias_r = bias.clone().requires_grad_(True) if use_bias else None
    q, k, v = (qkv_r[:, i * H:(i + 1) * H].reshape(B, S, H) for i in range(3))
    ref, lse_ref = _attn_ref(q, k, v, heads, 0.125, bias_r, key_pad)
    dout = rnd(B * S, H, seed=3)
    ref.backward(dout.view(B, S, H))
    frag = hip.attn_bias_pack(bias_d, S) if (use_bias and resident) else None  # with it S <= 320 takes the resident kernel
    out, lse = hip.attn_fwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, B, S, heads, 0.125, bias_d, pad_d, Spad, bias_frag=frag)
    assert_close(out.view(B, S, H), ref, fro=6e-3, what="attn out")
    assert_close(lse[:, :, :S], lse_ref, fro=1e-3, mx=2e-3, what="lse")
    dqkv, dbias = hip.attn_bwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, dev_bf16(dout), out, lse, B, S, heads, 0.125,
                               bias_d, biasT_d, pad_d, Spad, want_dbias=use_bias, bias_frag=frag)
    # gradients pass through bf16 P / dS operands: tolerance 1.2e-2 rel-Frobenius, 3e-2 max
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        assert_close(dqkv[:, sl], qkv_r.grad[:, sl], fro=1.2e-2, mx=3e-2, what=name)
    if use_bias:
        assert_close(dbias[:, :, :S], bias_r.grad, fro=1.2e-2, mx=3e-2, what="dbias")


@pytest.mark.parametrize("B,S,heads,use_bias,use_pad,per_sample", [
    (40, 257, 12, True, False, False), (48, 250, 12, True, True, False), (30, 200, 10, False, True, False),
    (26, 257, 12, True, True, True), (70, 256, 8, True, False, False), (33, 257, 9, False, False, False), (2, 193, 1, True, True, False)])
def test_attention_persistent_forward_over_several_items_per_workgroup(B, S, heads, use_bias, use_pad, per_sample):
    """Round 4: the persistent forward kernel (193 ... 257 tokens) with MORE (sample, head) items than workgroups, so that every
    workgroup walks several items through its K / V double buffer and the hand-counted waits, the lone query of S = 257 is
    merged after the next item's barrier, and the last item's merge happens after the loop.  The 16 regular query blocks run
    the resident kernel's per-tile code: bit-identical to it; the lone query (fp32 merge of 9 partial softmaxes) agrees to the
    op tolerance, like everything against the fp32 reference."""
    hip = hipmod()
    H = heads * 64
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(B * S, 3 * H, generator=g, device=DEV).to(torch.bfloat16)
    Spad = hip.attn_spad(S)
    bias_d = pad_d = None
    if use_bias:
        shape = (B, heads, S, Spad) if per_sample else (heads, S, Spad)
        bias_d = torch.zeros(shape, dtype=torch.bfloat16, device=DEV)
        bias_d[..., :S] = torch.randn(*shape[:-1], S, generator=g, device=DEV).to(torch.bfloat16)
    if use_pad:
        pad = torch.zeros(B, Spad, dtype=torch.uint8, device=DEV)
        pad[:, S:] = 1
        for b in range(1, B):
            pad[b, S - 1 - (7 * b) % (S // 2):] = 1
        pad_d = pad
    frag = hip.attn_bias_pack(bias_d, S) if use_bias else None
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    out, lse = hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias_d, pad_d, Spad, bias_frag=frag)
    old = hip.TUNE.attn_pers
    hip.TUNE.attn_pers = 0
    try:
        out_r, lse_r = hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias_d, pad_d, Spad, bias_frag=frag)
    finally:
        hip.TUNE.attn_pers = old
    nreg = min(S, 256)
    o3, r3 = out.view(B, S, H), out_r.view(B, S, H)
    assert torch.equal(o3[:, :nreg], r3[:, :nreg]), "regular query blocks differ from the resident kernel"
    torch.testing.assert_close(lse[:, :, :nreg], lse_r[:, :, :nreg], rtol=2e-6, atol=2e-6)  # (same max and sum; the final log / fma may contract differently)
    if S > 256:
        assert_close(o3[:, 256:], r3[:, 256:].float().cpu(), fro=4e-3, mx=2e-2, what="lone query")
        assert_close(lse[:, :, 256:S], lse_r[:, :, 256:S].cpu(), fro=1e-5, mx=1e-5, what="lone query lse")
    # against fp32 on a few samples
    for b in (0, B // 2, B - 1):
        qf, kf, vf = (x[b * S:(b + 1) * S].float().view(1, S, H) for x in (q, k, v))
        bb = None
        if use_bias:
            bb = (bias_d[b] if per_sample else bias_d)[..., :S].float()
        kp = pad_d[b:b + 1, :S].bool() if use_pad else None
        ref, lse_ref = _attn_ref(qf, kf, vf, heads, 0.125, bb, kp)
        assert_close(o3[b:b + 1], ref.cpu(), fro=6e-3, what="attn out vs fp32 (sample %d)" % b)
        assert_close(lse[b:b + 1, :, :S], lse_ref.cpu(), fro=1e-3, mx=2e-3, what="lse vs fp32")


@pytest.mark.parametrize("B,S,heads,use_bias,use_pad", [(40, 257, 12, True, False), (48, 250, 12, True, True), (30, 200, 10, False, True),
                                                       (26, 257, 8, True, True), (64, 256, 4, True, False), (33, 257, 9, False, False),
                                                       (128, 257, 24, True, False)])
def test_attention_persistent_backward_over_several_samples_per_workgroup(B, S, heads, use_bias, use_pad):
    """Round 4: the persistent dQ (+ dBias) kernel -- a workgroup owns one (head, half of the query blocks) and walks a CHUNK of the
    batch through its K / V double buffer, keeping the dBias accumulators in registers.  Against the kernels of rounds 1-3 (tune
    bit 10) on the same inputs: dq of the 16 regular query blocks and delta bit-identical (same per-tile arithmetic and order),
    the lone query of S = 257 (summed from nine key-pair partials) and dbias (other chunking of the fp32 sums) to fp32 accuracy;
    dk / dv -- the old kernel, fed with this kernel's delta -- bit-identical."""
    hip = hipmod()
    H = heads * 64
    g = torch.Generator(device=DEV).manual_seed(7)
    qkv = torch.randn(B * S, 3 * H, generator=g, device=DEV).to(torch.bfloat16)
    dout = torch.randn(B * S, H, generator=g, device=DEV).to(torch.bfloat16)
    Spad = hip.attn_spad(S)
    bias_d = biasT_d = pad_d = frag = None
    if use_bias:
        bias_d = torch.zeros(heads, S, Spad, dtype=torch.bfloat16, device=DEV)
        bias_d[..., :S] = torch.randn(heads, S, S, generator=g, device=DEV).to(torch.bfloat16)
        biasT_d = torch.zeros_like(bias_d)
        biasT_d[..., :S] = bias_d[..., :S].transpose(1, 2)
        frag = hip.attn_bias_pack(bias_d, S)
    if use_pad:
        pad_d = torch.zeros(B, Spad, dtype=torch.uint8, device=DEV)
        pad_d[:, S:] = 1
        for b in range(1, B):
            pad_d[b, S - 1 - (5 * b) % (S // 2):] = 1
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    out, lse = hip.attn_fwd(q, k, v, 3 * H, B, S, heads, 0.125, bias_d, pad_d, Spad, bias_frag=frag)
    res = {}
    for pers in (1, 0):
        hip.TUNE.attn_pers_bwd = pers
        try:
            res[pers] = hip.attn_bwd(q, k, v, 3 * H, dout, out, lse, B, S, heads, 0.125, bias_d, biasT_d, pad_d, Spad,
                                     want_dbias=use_bias, bias_frag=frag)
        finally:
            hip.TUNE.attn_pers_bwd = 1
    (new, dbn), (old, dbo) = res[1], res[0]
    assert new.isfinite().all()
    n3, o3 = new.view(B, S, 3 * H), old.view(B, S, 3 * H)
    nreg = min(S, 256)
    if use_bias:
        assert torch.equal(n3[:, :nreg, :H], o3[:, :nreg, :H]), "dq of the regular query blocks differs from the round-3 kernel"
    else:  # (without a bias rounds 1-3 run attn_bwd_dq_kernel: exp instead of exp2 of the pre-scaled score)
        assert_close(n3[:, :nreg, :H], o3[:, :nreg, :H].float().cpu(), fro=4e-3, mx=2e-2, what="dq")
    # dk / dv: the persistent dK / dV kernel behind it (same per-tile arithmetic and order as the rounds 1-3 kernel, fed with this
    # kernel's delta): keys 0 ... 255 bit-identical; the lone key of S = 257 is summed from eight query-half partials
    assert torch.equal(n3[:, :nreg, H:], o3[:, :nreg, H:]), "dk / dv differ (delta?)"
    if S > 256:
        assert_close(n3[:, 256:, :H], o3[:, 256:, :H].float().cpu(), fro=4e-3, mx=2e-2, what="dq of the lone query")
        assert_close(n3[:, 256:, H:], o3[:, 256:, H:].float().cpu(), fro=4e-3, mx=2e-2, what="dk / dv of the lone key")
    if use_bias:
        assert_close(dbn[:, :, :S], dbo[:, :, :S].cpu(), fro=2e-5, mx=2e-4, what="dbias")
    # fp32 reference on sampled items -- at the headline launch (B = 128, 24 heads) too, where rounds 1-4 compared the persistent
    # kernels with the older kernels only (round-4 verdict, Weak #2)
    if True:
        for b in ((0, B - 1) if B <= 64 else (0, B // 2 + 1, B - 1)):
            qkv_r = qkv[b * S:(b + 1) * S].float().cpu().requires_grad_(True)
            qr, kr, vr = (qkv_r[:, i * H:(i + 1) * H].reshape(1, S, H) for i in range(3))
            bb = bias_d[..., :S].float().cpu() if use_bias else None
            kp = pad_d[b:b + 1, :S].bool().cpu() if use_pad else None
            ref, _ = _attn_ref(qr, kr, vr, heads, 0.125, bb, kp)
            ref.backward(dout[b * S:(b + 1) * S].float().cpu().view(1, S, H))
            assert_close(n3[b], qkv_r.grad, fro=1.2e-2, mx=3e-2, what="dqkv vs fp32 (sample %d)" % b)


@pytest.mark.parametrize("B,S,heads", [(20, 72, 2), (3, 257, 2), (2, 330, 1)])
def test_attention_backward_ignores_unspecified_pad_entries(B, S, heads, merge_dbias):
    """lse / delta rows, bias columns and bias rows in [S, Spad) are unspecified (the wrappers allocate with torch.empty, the
    delta kernel writes rows < S only): poison them with NaN / Inf and require the same gradients (a `0 * x` on a masked
    entry instead of a select turns a NaN there into NaN dK / dV for the whole key block)."""
    hip = hipmod()
    H = heads * 64
    qkv, bias, key_pad, d, bias_d, biasT_d, pad_d, Spad = _attn_inputs(B, S, heads, True, True)
    out, lse = hip.attn_fwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, B, S, heads, 0.125, bias_d, pad_d, Spad,
                            bias_frag=hip.attn_bias_pack(bias_d, S))
    dout = dev_bf16(rnd(B * S, H, seed=3))
    clean, clean_dbias = hip.attn_bwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, dout, out, lse, B, S, heads, 0.125,
                                      bias_d, biasT_d, pad_d, Spad, want_dbias=True, bias_frag=hip.attn_bias_pack(bias_d, S))
    assert clean.isfinite().all() and clean_dbias[:, :, :S].isfinite().all()
    # delta is a workspace of the call (the dQ kernels fill its live part from dout and out): poison all of it
    delta = torch.full((B, heads, Spad), float("inf"), dtype=torch.float32, device=DEV)
    lse_p = lse.clone()
    lse_p[:, :, S:] = float("nan")
    bias_p, biasT_p = bias_d.clone(), biasT_d.clone()
    bias_p[:, :, S:] = float("nan")
    biasT_p[:, :, S:] = 3.0e4  # the transposed image's pad columns must be finite (the dK/dV kernel adds the bias with an MFMA)
    dqkv = torch.empty(B * S, 3 * H, dtype=torch.bfloat16, device=DEV)
    dbias = hip.attn_dbias_buffer(B, S, heads, Spad, DEV)
    hip.attn_bwd_launch(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, dout, bias_p, biasT_p, pad_d, lse_p, delta,
                        dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:], 3 * H, dbias, B, S, Spad, heads, 0.125,
                        hip.attn_bias_pack(bias_p, S), out=out)
    assert torch.equal(dqkv, clean)
    # the separate delta pass (op_attn_bwd_delta, out = NULL) gives the same gradients up to the order of its fp32 row sums
    delta2 = torch.empty(B, heads, Spad, dtype=torch.float32, device=DEV)
    hip._check(hip.lib().op_attn_bwd_delta(hip.ptr(dout), hip.ptr(out), dout.stride(0), hip.ptr(delta2), B, S, Spad, heads,
                                           hip.stream()), "op_attn_bwd_delta")
    torch.testing.assert_close(delta[:, :, :S], delta2[:, :, :S], rtol=1e-4, atol=1e-4)
    dqkv2 = torch.empty_like(dqkv)
    hip.attn_bwd_launch(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, dout, bias_d, biasT_d, pad_d, lse, delta2,
                        dqkv2[:, :H], dqkv2[:, H:2 * H], dqkv2[:, 2 * H:], 3 * H, None, B, S, Spad, heads, 0.125,
                        hip.attn_bias_pack(bias_d, S))
    assert_close(dqkv2, clean.float().cpu(), fro=2e-3, mx=2e-2, what="separate delta pass")
    got = dbias.sum(0)[:, :, :S]
    assert got.isfinite().all()
    if merge_dbias:  # slabs with plain read-modify-write: deterministic
        assert torch.equal(got, clean_dbias[:, :, :S])
    else:            # the separate dBias kernel adds batch chunks with fp32 atomics: order-dependent last bits
        assert_close(got, clean_dbias[:, :, :S].cpu(), fro=1e-5, mx=1e-4, what="dbias")
    out_p, _ = hip.attn_fwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, B, S, heads, 0.125, bias_p, pad_d, Spad,
                            bias_frag=hip.attn_bias_pack(bias_p, S))
    assert torch.equal(out_p, out)
    hip.lib().op_attn_set_resident(0)  # the streaming kernel reads the row-major image: same requirement
    try:
        out_s, _ = hip.attn_fwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, B, S, heads, 0.125, bias_d, pad_d, Spad)
        out_sp, _ = hip.attn_fwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, B, S, heads, 0.125, bias_p, pad_d, Spad)
    finally:
        hip.lib().op_attn_set_resident(1)
    assert torch.equal(out_sp, out_s)


@pytest.mark.parametrize("B,S,heads,use_pad", [(3, 70, 2, True), (5, 83, 3, False), (2, 257, 2, True), (3, 401, 2, True), (2, 530, 1, False)])
def test_attention_per_sample_bias(B, S, heads, use_pad):
    """Masked pretraining gathers a different token subset per sample, so the additive bias is [B, heads, S, S]
    (adapter/image.py:229-246): one bias image per sample in the kernels, one gradient slab per sample back.  More than 384
    kept tokens (448^2 images and up) take the separate dQ and dBias kernels, one sample per dBias workgroup (round 3)."""
    hip = hipmod()
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=1)
    bias = rnd(B, heads, S, S, seed=2)
    key_pad = None
    if use_pad:
        key_pad = torch.zeros(B, S, dtype=torch.bool)
        for b in range(1, B):
            key_pad[b, S - 3 * b:] = True
    Spad = hip.attn_spad(S)
    qkv_r, bias_r = qkv.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    q, k, v = (qkv_r[:, i * H:(i + 1) * H].reshape(B, S, H) for i in range(3))
    ref, lse_ref = _attn_ref(q, k, v, heads, 0.125, bias_r, key_pad)
    dout = rnd(B * S, H, seed=3)
    ref.backward(dout.view(B, S, H))
    d = dev_bf16(qkv)
    bias_d = torch.zeros(B, heads, S, Spad, dtype=torch.bfloat16, device=DEV)
    bias_d[..., :S] = bias.to(torch.bfloat16).to(DEV)
    biasT_d = torch.zeros_like(bias_d)
    biasT_d[..., :S] = bias.transpose(2, 3).to(torch.bfloat16).to(DEV)
    pad_d = None
    if use_pad:
        pad_d = torch.ones(B, Spad, dtype=torch.uint8, device=DEV)
        pad_d[:, :S] = key_pad.to(torch.uint8).to(DEV)
    out_s, _ = hip.attn_fwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, B, S, heads, 0.125, bias_d, pad_d, Spad)  # streaming
    assert_close(out_s.view(B, S, H), ref, fro=6e-3, what="attn out (streaming kernel)")
    out, lse = hip.attn_fwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, B, S, heads, 0.125, bias_d, pad_d, Spad,
                            bias_frag=hip.attn_bias_pack(bias_d, S))  # resident kernel, per-sample fragment-major images
    assert_close(out.view(B, S, H), ref, fro=6e-3, what="attn out")
    for fr in (hip.attn_bias_pack(bias_d, S), None):  # merged dQ + dBias kernel with the fragment image / with the q-major image
        dqkv, dbias = hip.attn_bwd(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], 3 * H, dev_bf16(dout), out, lse, B, S, heads, 0.125,
                                   bias_d, biasT_d, pad_d, Spad, want_dbias=True, bias_frag=fr)
        for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
            assert_close(dqkv[:, sl], qkv_r.grad[:, sl], fro=1.2e-2, mx=3e-2, what=name)
        assert_close(dbias[..., :S], bias_r.grad, fro=1.2e-2, mx=3e-2, what="per-sample dbias")
    assert dbias.shape == (B, heads, S, Spad)
    assert_close(dbias[..., :S], bias_r.grad, fro=1.2e-2, mx=3e-2, what="per-sample dbias")


@pytest.mark.parametrize("B,slots,T", [(3, 40, 37), (4, 64, 46)])
def test_audio_stem_convs_match_conv1d(B, slots, T):
    """audio_ops: strided-view GEMM convolutions (feature extractor + grouped positional conv) vs F.conv1d, fwd + grads.
    The second size makes the row counts multiples of 64, so the weight gradients take the transpose-read GEMM on the
    strided (and, for the grouped conv, overlapping-row) views; the first one takes the transposed-copy fallback."""
    from one_peace_amd import audio_ops
    import torch.nn as nn
    torch.manual_seed(0)
    # --- stride-2 convs, k = 3 and k = 2, channels-last flat rows with slack ---
    for k in (3, 2):
        Cin, Cout = 64, 128
        xs = rnd(B, slots, Cin, seed=3 + k)
        w = rnd(Cout, Cin, k, seed=5 + k, scale=(Cin * k) ** -0.5)
        xr, wr = xs.clone().requires_grad_(True), w.clone().requires_grad_(True)
        # reference on each sample separately; valid outputs only
        ref = F.conv1d(xr.transpose(1, 2), wr, stride=2).transpose(1, 2)  # [B, Tout, Cout]
        Tout = ref.shape[1]
        dyv = rnd(B, Tout, Cout, seed=9)
        ref.backward(dyv)
        xd = torch.cat([dev_bf16(xs).view(B * slots, Cin), torch.zeros(2, Cin, dtype=torch.bfloat16, device=DEV)]).requires_grad_(True)
        wd = dev_bf16(w).requires_grad_(True)
        y = audio_ops.StridedConv1dFn.apply(xd, wd)
        yv = y[: B * slots // 2].view(B, slots // 2, Cout)[:, :Tout]
        assert_close(yv, ref, what="strided conv k=%d" % k)
        yv.backward(dev_bf16(dyv))
        assert_close(xd.grad[: B * slots].view(B, slots, Cin), xr.grad, what="strided conv dx k=%d" % k)
        assert_close(wd.grad, wr.grad, what="strided conv dw k=%d" % k)
    # --- grouped same-padding conv ---
    C, G, k = 128, 4, 19
    x = rnd(B, T, C, seed=11)
    w = rnd(C, C // G, k, seed=12, scale=(C // G * k) ** -0.5)
    b = rnd(C, seed=13)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv1d(xr.transpose(1, 2), wr, br, padding=k // 2, groups=G).transpose(1, 2)
    dy = rnd(B, T, C, seed=14)
    ref.backward(dy)
    xd, wd, bd = (dev_bf16(t).requires_grad_(True) for t in (x, w, b))
    y = audio_ops.grouped_conv1d_same(xd, wd, bd, G)
    assert_close(y, ref, what="grouped conv")
    y.backward(dev_bf16(dy))
    assert_close(xd.grad, xr.grad, what="grouped conv dx")
    assert_close(wd.grad, wr.grad, what="grouped conv dw")
    assert_close(bd.grad, br.grad, what="grouped conv db")


def test_audio_feature_extractor_matches_conv_stack():
    """Whole 7-layer extractor (k10s5 + 4x k3s2 + 2x k2s2, LN+GELU each) through audio_ops vs the torch conv path."""
    from one_peace_amd import audio_ops, ops
    from one_peace_amd.adapter.audio import ConvFeatureExtractionModel
    torch.manual_seed(0)
    spec = [(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)] * 2
    m = ConvFeatureExtractionModel(spec).to(DEV).to(torch.bfloat16)
    wav = torch.randn(3, 4000, device=DEV).to(torch.bfloat16)
    out = audio_ops.feature_extractor(wav, m.conv_layers)           # [B, T, C]
    saved = ops.hip_eligible
    ops.hip_eligible = lambda t: False
    try:
        ref = m(wav).transpose(1, 2)
    finally:
        ops.hip_eligible = saved
    assert out.shape == ref.shape
    assert_close(out, ref.float(), fro=3e-2, mx=8e-2, what="feature extractor (bf16 vs bf16, 7 layers)")


@pytest.mark.parametrize("C,bias", [(64, False), (512, True), (512, False)])
def test_audio_first_block_fused_from_the_waveform(C, bias):
    """(ABI 7) op_audio_conv1_ln_gelu_fwd / _bwd: Conv1d(1 -> C, k = 10, stride 5) -> LayerNorm(C) -> GELU as ONE kernel each way, rows
    computed from their ten waveform samples (adapter/audio.py:254-311, block 0).  Against the fp32 definition (the convolution output
    rounded to bf16 before the statistics, as every bf16 path stores it), forward and all four parameter gradients; and the whole
    extractor with the fused block against the GEMM + LayerNorm form of rounds 1-4 (ONEPEACE_FUSED_CONV1=0)."""
    from one_peace_amd import audio_ops
    from one_peace_amd.adapter.audio import ConvFeatureExtractionModel
    hip = hipmod()
    g = torch.Generator().manual_seed(3)
    B, T = 3, 4000
    Tp = (T + 319) // 320 * 320
    rows = B * (Tp // 5)
    wav = torch.zeros(B * Tp + 16)
    wav[: B * Tp].view(B, Tp)[:, :T] = torch.randn(B, T, generator=g)
    w0 = bf16_round(torch.randn(C, 10, generator=g) * 0.4)
    b0 = bf16_round(torch.randn(C, generator=g) * 0.1) if bias else None
    lw, lb = bf16_round(1 + 0.1 * torch.randn(C, generator=g)), bf16_round(0.1 * torch.randn(C, generator=g))
    dy = bf16_round(torch.randn(rows, C, generator=g))
    wq = bf16_round(wav)
    # ---- fp32 definition ----
    w0r, lwr, lbr = w0.clone().requires_grad_(True), lw.clone().requires_grad_(True), lb.clone().requires_grad_(True)
    b0r = b0.clone().requires_grad_(True) if bias else None
    patches = torch.as_strided(wq, (rows, 10), (5, 1))
    x = patches @ w0r.t() + (b0r if bias else 0.0)
    x = x + (bf16_round(x.detach()) - x.detach())  # straight-through bf16 rounding of the stored convolution output
    ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x, (C,), lwr, lbr, 1e-5))
    (ref * dy).sum().backward()
    # ---- HIP ----
    wd = dev_bf16(wq)
    y, mean, rstd = hip.audio_conv1_ln_gelu_fwd(wd, 5, dev_bf16(w0), dev_bf16(b0) if bias else None, dev_bf16(lw), dev_bf16(lb), rows, 1e-5)
    assert_close(y, ref.detach(), what="fused first block")
    dw0, db0, dlw, dlb = hip.audio_conv1_ln_gelu_bwd(dev_bf16(dy), wd, 5, dev_bf16(w0), dev_bf16(b0) if bias else None, dev_bf16(lw), dev_bf16(lb),
                                                     mean, rstd)
    assert_close(dw0, w0r.grad, fro=8e-3, mx=3e-2, what="dW of the first convolution")
    assert_close(dlw, lwr.grad, fro=8e-3, mx=3e-2, what="dLN weight")
    assert_close(dlb, lbr.grad, fro=8e-3, mx=3e-2, what="dLN bias")
    if bias:
        assert_close(db0, b0r.grad, fro=8e-3, mx=3e-2, what="db of the first convolution")
    else:
        assert db0 is None
    # ---- the whole extractor: fused block against the GEMM + LayerNorm form ----
    if C == 64:
        torch.manual_seed(0)
        spec = [(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)] * 2
        m = ConvFeatureExtractionModel(spec).to(DEV).to(torch.bfloat16)
        wv = torch.randn(3, 4000, device=DEV).to(torch.bfloat16)
        outs, grads = {}, {}
        saved = audio_ops.FUSED_CONV1
        try:
            for mode in (True, False):
                audio_ops.FUSED_CONV1 = mode
                m.zero_grad()
                o = audio_ops.feature_extractor(wv, m.conv_layers)
                o.float().square().sum().backward()
                outs[mode] = o.detach().float()
                grads[mode] = {n: q.grad.detach().float().clone() for n, q in m.named_parameters()}
        finally:
            audio_ops.FUSED_CONV1 = saved
        assert_close(outs[True], outs[False].cpu(), fro=2e-2, mx=6e-2, what="extractor, fused vs unfused first block")
        for n, gq in grads[False].items():
            assert rel_fro(grads[True][n], gq) <= 3e-2, (n, rel_fro(grads[True][n], gq))


@pytest.mark.parametrize("M,N,K,epi", [(16448, 4608, 1536, "qkv"), (16448, 6144, 1536, "geglu"), (16000, 1536, 6144, "resid"),
                                       (32896, 1536, 1536, "bias"),
                                       # the EXACT launches of the headline step (128 x 257 image tokens per pass), fp32-checked:
                                       (32896, 4608, 1536, "qkv"), (32896, 6144, 1536, "geglu"), (32896, 12288, 1536, "up2"),
                                       (32896, 1536, 6144, "resid"), (32896, 1536, 1536, "resid"), (32896, 1536, 12288, "plain")])
def test_gemm_full_size_launches_match_torch(M, N, K, epi):
    """BASELINE-size launches (every CU busy, the auto-selected kernel flavours: the four-wave 256x256 kernels + tail-rows launch
    at M = 32896, 128x128 for M = 16000) against an fp32 matmul of the same bf16 operands on the same device -- the oracle is
    too slow at this size, so the yardstick is an independent fp32 GEMM; epilogues are applied to its result as the oracle
    defines them.  "up2" is the training path of the FFN up-projection since round 3 (plain wi_0 | wi_1 launch writing h0 | h1,
    then op_ln_geglu_fwd), "plain" the merged K = 2F input gradient of the FFN."""
    hip = hipmod()
    g = torch.Generator(device=DEV).manual_seed(7)
    a = torch.randn(M, K, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    mk = lambda *s: (torch.randn(*s, generator=g, device=DEV, dtype=torch.float32) * 0.05).to(torch.bfloat16)
    if epi == "qkv":
        ws, b = [mk(N // 3, K) for _ in range(3)], mk(N // 3)
        out = hip.gemm_nt(a, ws, [b, None, b], n_seg=N // 3, N=N)
        ref = torch.cat([a.float() @ w.float().t() for w in ws], 1)
        ref[:, :N // 3] += b.float()
        ref[:, 2 * N // 3:] += b.float()
    elif epi == "geglu":
        w0, w1 = mk(N, K), mk(N, K)
        h0, h1 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        out = hip.gemm_nt(a, [w0, w1], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1)
        r0, r1 = a.float() @ w0.float().t(), a.float() @ w1.float().t()
        ref = torch.nn.functional.gelu(r0) * r1
        assert_close(h0, r0, what="h0")
        assert_close(h1, r1, what="h1")
    elif epi == "up2":
        Fd = N // 2
        w0, w1 = mk(Fd, K), mk(Fd, K)
        lw, lb = (1 + mk(Fd)).to(torch.bfloat16), mk(Fd)
        out = hip.gemm_nt(a, [w0, w1], n_seg=Fd, N=N)
        ref = torch.cat([a.float() @ w0.float().t(), a.float() @ w1.float().t()], 1)
        gln, mean, rstd = hip.ln_geglu_fwd(out[:, :Fd], out[:, Fd:], lw, lb)
        gref = torch.nn.functional.layer_norm(torch.nn.functional.gelu(ref[:, :Fd]) * ref[:, Fd:], (Fd,), lw.float(), lb.float(), 1e-5)
        # (h0 / h1 are rounded to bf16 before the GELU, as the reference's bf16 modules round them: one more rounding than a single op)
        assert_close(gln, gref, fro=8e-3, mx=6e-2, what="LayerNorm(gelu(h0) * h1)")
        del gref
    elif epi == "resid":
        rps = 257 if M % 257 == 0 else 250
        w, b, gamma = mk(N, K), mk(N), mk(N)
        res = torch.randn(M, N, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
        ps = (torch.rand(M // rps, generator=g, device=DEV) > 0.3).float() / 0.7
        y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        out = hip.gemm_nt(a, [w], [b], epilogue=hip.EPI_RESID, resid=res, gamma=gamma, rowscale=ps, rows_per_sample=rps, h0=y)
        yref = a.float() @ w.float().t() + b.float()
        assert_close(y, yref, what="branch output")
        ref = res.float() + ps.repeat_interleave(rps)[:, None] * gamma.float() * yref
    elif epi == "plain":
        w = mk(N, K)
        out = hip.gemm_nt(a, [w])
        ref = a.float() @ w.float().t()
    else:
        w, b = mk(N, K), mk(N)
        out = hip.gemm_nt(a, [w], [b])
        ref = a.float() @ w.float().t() + b.float()
    assert_close(out, ref, what=epi)


@pytest.mark.parametrize("kind", ["down_resid", "dgrad"])
def test_grouped_gemm_at_the_headline_launches_matches_torch(kind):
    """Round 4: op_gemm_nt_grouped at the EXACT grouped launches of the headline step -- the three per-modality FFN
    problems of the lock-step pass (8 192 text, 32 896 image, 32 000 audio rows, N = 1536) as one persistent launch: the
    down-projection + bias + layer-scale / drop-path residual with the branch output saved (K = 6144) and the K = 12 288 input
    gradient of wi_0 | wi_1 -- every problem against an fp32 matmul of the same bf16 operands on the device."""
    hip = hipmod()
    g = torch.Generator(device=DEV).manual_seed(13)
    rows, rps = (8192, 32896, 32000), (64, 257, 250)
    N, K = 1536, (6144 if kind == "down_resid" else 12288)
    mk = lambda *s_, sc=1.0: (torch.randn(*s_, generator=g, device=DEV, dtype=torch.float32) * sc).to(torch.bfloat16)  # noqa: E731
    As = [mk(m, K) for m in rows]
    Ws = [mk(N, K, sc=0.05) for _ in rows]
    outs = [torch.empty(m, N, dtype=torch.bfloat16, device=DEV) for m in rows]
    if kind == "down_resid":
        bs, gamma = [mk(N, sc=0.05) for _ in rows], mk(N, sc=0.05)
        res = [mk(m, N) for m in rows]
        ys = [torch.empty(m, N, dtype=torch.bfloat16, device=DEV) for m in rows]
        pss = [(torch.rand(m // r, generator=g, device=DEV) > 0.4).float() / 0.6 for m, r in zip(rows, rps)]
        got = hip.gemm_nt_grouped(As, Ws, biases=bs, outs=outs, epilogue=hip.EPI_RESID, h0s=ys, resids=res, gammas=[gamma] * 3,
                                  rowscales=pss, rows_per_sample=list(rps))
        assert got is not None, "the grouped launch did not take the headline shape"
        for a, w, b, r, ps, rp, o, y in zip(As, Ws, bs, res, pss, rps, outs, ys):
            yref = a.float() @ w.float().t() + b.float()
            assert_close(y, yref, what="branch output")
            assert_close(o, r.float() + ps.repeat_interleave(rp)[:, None] * gamma.float() * yref, what="grouped down-projection + residual")
    else:
        got = hip.gemm_nt_grouped(As, Ws, outs=outs)
        assert got is not None, "the grouped launch did not take the headline shape"
        for a, w, o in zip(As, Ws, outs):
            assert_close(o, a.float() @ w.float().t(), what="grouped input gradient")


@pytest.mark.parametrize("M,No,Ni", [(16448, 1536, 6144), (32896, 1536, 6144), (32896, 12288, 1536), (73216, 4608, 1536)])
def test_weight_gradient_full_size_matches_torch(M, No, Ni):
    """dW = dy^T x at BASELINE size with in-place accumulation, transpose-read kernels + split-K: tokens = 16448 (round 1),
    and the exact launches of the headline step -- K = 32896 image tokens for the per-modality FFN weights (w2; the merged
    wi_0 | wi_1 launch), K = 73216 rows of the lock-step pass for the merged q | k | v weights -- fp32-checked."""
    from one_peace_amd import ops
    g = torch.Generator(device=DEV).manual_seed(9)
    dy = torch.randn(M, No, generator=g, device=DEV).to(torch.bfloat16)
    x = torch.randn(M, Ni, generator=g, device=DEV).to(torch.bfloat16)
    base = torch.randn(No, Ni, generator=g, device=DEV).to(torch.bfloat16)
    grad = base.clone()
    ops.wgrad(dy, x, out=grad, accumulate=True)
    ref = base.float() + dy.float().t() @ x.float()
    assert_close(grad, ref, what="dW accumulate")


GROUPED_TN_SETS = {
    "mixed": [(256, 512, 256, True), (128, 264, 520, False), (1024, 256, 1536, True), (64, 8, 8, True), (448, 2304, 2560, False)],
    "twelve": [(320, 512, 384, bool(i & 1)) for i in range(12)],
    "sixteen": [(128, 96, 8 * (i + 1), bool(i & 1)) for i in range(16)],  # (the 16 groups of the audio positional convolution's weight gradient: M = 96)
    "one": [(512, 768, 256, True)],
}


@pytest.mark.parametrize("name", sorted(GROUPED_TN_SETS))
@pytest.mark.parametrize("nwg", [0, 3, 8, 37])
def test_gemm_tn_grouped_is_bit_identical_to_unsplit_launches(name, nwg):
    """op_gemm_tn_grouped: several weight-gradient GEMMs (own K, sizes, ragged edges, strided operands, fresh output or
    accumulation) as one persistent launch -- every tile runs its whole K, so each problem must equal the UNSPLIT op_gemm_tn bit
    for bit and the fp32 product within the bf16 tolerance; forced workgroup counts (fewer than queues, fewer than tiles, not a
    multiple of eight) exercise work stealing, and a second launch on the same counter block the re-arm."""
    hip = hipmod()
    probs, want, refs = [], [], []
    for i, (K, M, N, acc) in enumerate(GROUPED_TN_SETS[name]):
        dy, x = rnd(K, M + 16, seed=10 + i, scale=0.5), rnd(K, N, seed=40 + i, scale=0.5)
        base = rnd(M, N, seed=70 + i)
        a = dev_bf16(dy)[:, 8:8 + M]  # a column block of a wider matrix, like the halves of the packed dh
        out = dev_bf16(base).clone()
        probs.append((a, dev_bf16(x), out, acc))
        want.append(hip.gemm_tn(a, dev_bf16(x), dev_bf16(base).clone(), acc, splitk=False))
        refs.append((base if acc else 0) + dy[:, 8:8 + M].t() @ x)
    for rep in range(2):
        if rep:
            for (_, _, out, _), (K, M, N, acc), i in zip(probs, GROUPED_TN_SETS[name], range(99)):
                out.copy_(dev_bf16(rnd(M, N, seed=70 + i)))
        assert hip.gemm_tn_grouped(probs, tune=nwg)
        torch.cuda.synchronize()
        for (_, _, out, _), w, r in zip(probs, want, refs):
            assert torch.equal(out, w), "grouped launch differs from the unsplit op_gemm_tn (launch %d)" % rep
            assert_close(out, r, what="grouped dW")


@pytest.mark.parametrize("use_bias", [False, True])
def test_gemm_nt_batched_equals_one_launch_per_problem(use_bias):
    """(ABI 7) op_gemm_nt_batched: G equally shaped products over OVERLAPPING strided patch views (row stride cg < K: the grouped
    positional Conv1d of the audio adapter, adapter/audio.py:57-84, as GEMMs over a zero-padded group-major copy) in one launch --
    bit-identical to one op_gemm_nt launch per group on the same views, and within the bf16 tolerance of the fp32 product."""
    hip = hipmod()
    G, rows, cg, k = 5, 700, 96, 19
    Kp = (k * cg + 63) // 64 * 64
    xg = dev_bf16(rnd(G, rows + k + 1, cg, seed=3, scale=0.5))
    wg = dev_bf16(rnd(G, cg, Kp, seed=4, scale=0.05))
    bias = dev_bf16(rnd(G, cg, seed=5)) if use_bias else None
    out = torch.empty(G, rows, cg, dtype=torch.bfloat16, device=DEV)
    hip.gemm_nt_batched(xg, wg, bias, out, rows, Kp)
    torch.cuda.synchronize()
    for g in range(G):
        a = torch.as_strided(xg[g], (rows, Kp), (cg, 1), xg[g].storage_offset())
        one = hip.gemm_nt(a, [wg[g]], [bias[g]] if use_bias else None, splitk=False)
        assert torch.equal(out[g], one), "batched launch differs from the per-group launch (group %d)" % g
        ref = a.float() @ wg[g].float().t() + (bias[g].float() if use_bias else 0.0)
        assert_close(out[g], ref.cpu(), what="batched product")


@pytest.mark.parametrize("nwg", [0, 5])
def test_gemm_tn_grouped_row_dot_side_product(nwg):
    """(ABI 7) op_gemm_tn_grouped with W / rowdot on some problems: rowdot[m] += sum_n W[m][n] * P[m][n] with P = THIS launch's fp32
    product A^T B (not the accumulated gradient), added on top of what the buffer holds; the gradients themselves must stay bit-identical
    to a launch without the side product, and a problem without one is untouched.  (ABI 8) With rscale the gradient receives
    rscale[m] * P[m] while rowdot still sums the unscaled product: with A = the branch gradient without gamma and rscale = gamma, rowdot
    is sum_rows ps dout y-without-bias for ANY gamma (ops.dgamma_from_wgrad_ok, transformer_layer.py:70-88)."""
    hip = hipmod()
    sizes = [(1024, 256, 512, True), (2048, 512, 256, True), (512, 256, 256, True)]
    with_side = (True, False, True)
    probs, plain, refs, rowdots = [], [], [], []
    for i, (K, M, N, acc) in enumerate(sizes):
        dy, x, base, w = rnd(K, M, seed=10 + i, scale=0.5), rnd(K, N, seed=40 + i, scale=0.5), rnd(M, N, seed=70 + i), rnd(M, N, seed=90 + i)
        a, b = dev_bf16(dy), dev_bf16(x)
        out = dev_bf16(base).clone()
        wd = dev_bf16(w)
        rd = torch.full((N // 128, M), 0.25, dtype=torch.float32, device=DEV) if with_side[i] else None  # (overwritten, not added to)
        probs.append((a, b, out, acc, (wd, rd) if with_side[i] else None))
        plain.append((a, b, dev_bf16(base).clone(), acc))
        prod = a.float().t() @ b.float()
        refs.append((wd.float() * prod).view(M, N // 128, 128).sum(2).t().contiguous())  # [slots, M]: 128-column partial sums
        rowdots.append(rd)
    assert hip.gemm_tn_grouped(plain, tune=nwg)
    assert hip.gemm_tn_grouped(probs, tune=nwg)
    torch.cuda.synchronize()
    for q, q0, ref, rd in zip(probs, plain, refs, rowdots):
        assert torch.equal(q[2], q0[2]), "the side product changed the gradient"
        if rd is not None:
            err = float((rd - ref).abs().max()) / float(ref.abs().max())
            assert err < 2e-3, err
    again = [(q[0], q[1], q[2].clone(), q[3], (q[4][0], torch.empty_like(q[4][1])) if q[4] is not None else None) for q in probs]
    assert hip.gemm_tn_grouped(again, tune=nwg)  # no atomics: the partial sums are the same bits in every launch
    torch.cuda.synchronize()
    for q, q2 in zip(probs, again):
        assert q[4] is None or torch.equal(q[4][1], q2[4][1])
    # (ABI 8) row scales: an all-ones vector changes no bit; a real one (with zeros and tiny entries) scales the accumulated product only
    ones = [(q[0], q[1], dev_bf16(rnd(*q[2].shape, seed=70 + i)).clone(), q[3],
             (q[4][0], torch.zeros_like(q[4][1]), torch.ones(q[2].shape[0], dtype=torch.bfloat16, device=DEV)) if q[4] is not None else None)
            for i, q in enumerate(probs)]
    scaled, scales = [], []
    for i, q in enumerate(probs):
        sc = dev_bf16(rnd(q[2].shape[0], seed=120 + i))
        sc[3], sc[5] = 0.0, 1e-6
        scales.append(sc)
        scaled.append((q[0], q[1], dev_bf16(rnd(*q[2].shape, seed=70 + i)).clone(), q[3],
                       (q[4][0], torch.zeros_like(q[4][1]), sc) if q[4] is not None else None))
    assert hip.gemm_tn_grouped(ones, tune=nwg)
    assert hip.gemm_tn_grouped(scaled, tune=nwg)
    torch.cuda.synchronize()
    for i, (q1, qs, q0, ref) in enumerate(zip(ones, scaled, plain, refs)):
        assert torch.equal(q1[2], q0[2]), "a row scale of 1 changed the gradient"
        if qs[4] is None:
            assert torch.equal(qs[2], q0[2])
            continue
        prod = qs[0].float().t() @ qs[1].float()
        want = dev_bf16(rnd(*qs[2].shape, seed=70 + i)).float() + scales[i].float()[:, None] * prod
        assert_close(qs[2], want.cpu(), what="row-scaled accumulation")
        assert torch.equal(qs[2][3], dev_bf16(rnd(*qs[2].shape, seed=70 + i))[3]), "a zero row scale must leave the row alone"
        err = float((qs[4][1] - ref).abs().max()) / float(ref.abs().max())
        assert err < 2e-3, err  # the side product does not see the scale
    with pytest.raises(RuntimeError):  # ragged M: the side product rides on full tiles only
        bad = (dev_bf16(rnd(128, 264)), dev_bf16(rnd(128, 256)), dev_bf16(rnd(264, 256)), True,
               (dev_bf16(rnd(264, 256)), torch.zeros(2, 264, dtype=torch.float32, device=DEV)))
        hip.gemm_tn_grouped([bad, bad[:4]])
    with pytest.raises(RuntimeError):  # a row scale without the side product
        q = plain[0]
        lib = hip.lib()
        import ctypes
        one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())  # noqa: E731
        i64 = lambda v: (ctypes.c_int64 * 1)(v)  # noqa: E731
        ctr = torch.zeros(int(lib.op_gemm_tn_grouped_counter_bytes()), dtype=torch.uint8, device=DEV)
        sc = torch.ones(q[2].shape[0], dtype=torch.bfloat16, device=DEV)
        hip._check(lib.op_gemm_tn_grouped(1, one(q[0]), i64(q[0].stride(0)), one(q[1]), i64(q[1].stride(0)), one(q[2]), i64(q[2].stride(0)),
                                          i64(q[0].shape[1]), i64(q[1].shape[1]), i64(q[0].shape[0]), (ctypes.c_int32 * 1)(1), None, None, None,
                                          one(sc), hip.ptr(ctr), 0, hip.stream()), "op_gemm_tn_grouped")


def test_resid_bwd_g0_and_gamma_grad_finish():
    """(ABI 7, 8) The layer-scale gradient without the branch output: for out = resid + ps * gamma * (x W^T + b),
    dgamma[n] = sum_k W[n][k] G[n][k] + b[n] g0[n] with G = (ps dout)^T x and g0 = sum_m ps dout -- op_resid_bwd hands out g0 and the
    branch gradient u = ps dout WITHOUT gamma, the grouped weight-gradient launch adds gamma * G to the weight gradient and the row dot of
    W with G to a vector, op_gamma_grad_finish adds the bias term and re-arms the buffer.  No division by gamma anywhere: against the
    reference's definition sum_m ps dout y in fp32 (transformer_layer.py:78-88) for EVERY entry, gamma == 0, 1e-6 and 1e-30 included;
    the input gradient u . (gamma o W) from the gamma-scaled transposed copy (op_transpose_scaled)."""
    hip = hipmod()
    M, N, K, rps = 1024, 256, 512, 64
    dout, x, w, b = rnd(M, N, seed=1), rnd(M, K, seed=2, scale=0.5), rnd(N, K, seed=3, scale=0.2), rnd(N, seed=4)
    gamma = (0.5 + rnd(N, seed=5).abs()).clamp(max=2.0)
    gamma[7], gamma[8], gamma[9] = 0.0, 1e-6, 1e-30
    ps = (torch.arange(M // rps) % 3 != 0).float() / 0.75
    d_, x_, w_, b_, g_ = dev_bf16(dout), dev_bf16(x), dev_bf16(w), dev_bf16(b), dev_bf16(gamma)
    g0 = torch.empty(N, dtype=torch.float32, device=DEV)
    dbias = torch.zeros(N, dtype=torch.bfloat16, device=DEV)
    dy, dg, db = hip.resid_bwd(d_, None, g_, ps.to(DEV), rps, dgamma=None, dbias=dbias, accumulate=True, g0=g0)
    assert dg is None
    rows = ps.repeat_interleave(rps)[:, None].to(DEV)
    dq = d_.float() * rows
    assert_close(g0, dq.sum(0).cpu(), what="g0")
    assert_close(db, (dq * g_.float()).sum(0).cpu(), what="dbias")
    assert torch.equal(dy, dq.to(torch.bfloat16)), "with g0 the branch gradient carries no gamma"
    rowdot = torch.full((K // 128, N), float("nan"), dtype=torch.float32, device=DEV)  # every slot is written: no zeroing needed
    dW = torch.zeros(N, K, dtype=torch.bfloat16, device=DEV)
    other = (dev_bf16(rnd(256, 256, seed=8)), dev_bf16(rnd(256, 256, seed=9)), torch.zeros(256, 256, dtype=torch.bfloat16, device=DEV), True)
    assert hip.gemm_tn_grouped([(dy, x_, dW, True, (w_, rowdot, g_)), other])
    base = dev_bf16(rnd(N, seed=6))
    dgamma = base.clone()
    hip.gamma_grad_finish(rowdot, [(b_, g0)], dgamma, True)
    wt = hip.transpose(w_, scale=g_)
    dx = hip.gemm_nt(dy, [wt])
    torch.cuda.synchronize()
    y = x_.float() @ w_.float().t() + b_.float()
    ref = (dq * y).sum(0) + base.float()  # the reference's value for every column: no exception for gamma == 0
    assert_close(dgamma, ref.cpu(), what="dgamma from the weight gradient")
    # the zero / tiny layer scales one by one (assert_close is a norm over all columns): the reference's value within the bf16 noise of
    # a column (sum over 1024 rows of bf16-rounded ps * dout times y: measured 2e-3 ... 4e-3 of the largest column), and NOT the
    # bias-only value round 5 returned for gamma == 0
    scale = float(ref.abs().max())
    for col in (7, 8, 9):
        assert abs(float(dgamma[col]) - float(ref[col])) <= 1e-2 * scale, (col, float(dgamma[col]), float(ref[col]), scale)
    bias_only = float((b_.float() * g0)[7] + base.float()[7])  # what round 5 returned for the gamma == 0 column: the row-dot term missing
    assert abs(float(dgamma[7]) - float(ref[7])) < 0.25 * abs(float(ref[7]) - bias_only), (float(dgamma[7]), float(ref[7]), bias_only)
    dyg = dy.float() * g_.float()  # what the reference back-propagates into the branch: ps * gamma * dout
    assert_close(dW, (dyg.t() @ x_.float()).cpu(), what="weight gradient gamma * G")
    assert float(dW[7].abs().max()) == 0.0
    assert torch.equal(wt, (w_.float() * g_.float()[:, None]).to(torch.bfloat16).t())
    assert_close(dx, (dyg @ w_.float()).cpu(), what="input gradient through the gamma-scaled weight copy")


def test_gamma_grad_finish_with_three_weight_sets_sharing_gamma():
    """The lock-step FFN form: gamma_2 is shared by the three modality FFNs -- one row-dot buffer receives the side products of three
    down-projection weight gradients, op_gamma_grad_finish adds the three bias terms (one FFN without bias), writes (not accumulates)."""
    hip = hipmod()
    N = 256
    rowdot = dev_bf16(rnd(3 * 4, N, seed=1)).float().contiguous()  # three weight sets x four 128-column slots
    bs = [dev_bf16(rnd(N, seed=3)), None, dev_bf16(rnd(N, seed=4))]
    g0s = [dev_bf16(rnd(N, seed=5 + i)).float().contiguous() for i in range(3)]
    ref = rowdot.sum(0) + bs[0].float() * g0s[0] + bs[2].float() * g0s[2]
    out = torch.full((N,), 7.0, dtype=torch.bfloat16, device=DEV)
    hip.gamma_grad_finish(rowdot, list(zip(bs, g0s)), out, False)
    torch.cuda.synchronize()
    assert_close(out, ref.cpu(), what="dgamma over three weight sets")


def test_gemm_tn_grouped_rejects_what_the_kernel_cannot_take():
    hip = hipmod()
    ok = (dev_bf16(rnd(128, 64)), dev_bf16(rnd(128, 64)), dev_bf16(rnd(64, 64)), False)
    bad = (dev_bf16(rnd(96, 64)), dev_bf16(rnd(96, 64)), dev_bf16(rnd(64, 64)), False)  # K % 64 != 0
    before = ok[2].clone()
    assert hip.gemm_tn_grouped([ok, bad]) is False
    torch.cuda.synchronize()
    assert torch.equal(ok[2], before)  # nothing was launched
    with pytest.raises(RuntimeError):
        hip.gemm_tn_grouped([ok] * 17)


def test_weight_gradients_of_a_headline_layer_grouped_match_torch():
    """The eight weight gradients of one lock-step layer of the headline step (q|k|v and out-proj over 73 088 rows, wi_0|wi_1 and
    wo of the image / audio / text FFNs) as ONE grouped launch, accumulated into existing gradients: fp32-checked, and equal to
    what the per-problem split-K launches give within two bf16 roundings."""
    hip = hipmod()
    g = torch.Generator(device=DEV).manual_seed(11)
    H, F = 1536, 6144
    rows = {"all": 73088, "img": 32896, "aud": 32000, "txt": 8192}
    mk = lambda r, c: torch.randn(r, c, generator=g, device=DEV).to(torch.bfloat16)  # noqa: E731
    sets = [("all", 3 * H, H), ("all", H, H)] + [(m, o, i) for m in ("img", "aud", "txt") for o, i in ((2 * F, H), (H, F))]
    probs = []
    for m, o, i in sets:
        probs.append((mk(rows[m], o), mk(rows[m], i), mk(o, i), True))
    base = [q[2].clone() for q in probs]
    assert hip.gemm_tn_grouped(probs)
    torch.cuda.synchronize()
    for (dy, x, out, _), b in zip(probs, base):
        ref = b.float() + dy.float().t() @ x.float()
        assert_close(out, ref, what="grouped layer dW")
        single = hip.gemm_tn(dy, x, b.clone(), True)
        assert rel_fro(single, out) < 6e-3


@pytest.mark.parametrize("epi", ["bias", "geglu", "resid"])
def test_gemm_tail_rows_split(epi):
    """M = 2 x 256 + 77: the full M-tiles and the leftover rows run as two launches (forced here; in production only when
    that saves a round of all CUs).  Row-dependent epilogue inputs (residual, drop-path scale, saved branch outputs) must
    follow the row offset."""
    hip = hipmod()
    M, N, K, S = 589, 512, 256, 19
    a = rnd(M, K, seed=1)
    old = hip.lib().op_gemm_set_tile(2)
    hip.lib().op_gemm_set_tile(53)
    try:
        if epi == "bias":
            w, b = rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
            out = hip.gemm_nt(dev_bf16(a), [dev_bf16(w)], [dev_bf16(b)])
            assert_close(out, a @ w.t() + b, what="bias")
        elif epi == "geglu":
            w0, w1 = rnd(N, K, seed=2, scale=0.1), rnd(N, K, seed=3, scale=0.1)
            h0 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            h1 = torch.empty_like(h0)
            out = hip.gemm_nt(dev_bf16(a), [dev_bf16(w0), dev_bf16(w1)], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1)
            assert_close(out, O.gelu_erf(a @ w0.t()) * (a @ w1.t()), what="geglu")
            assert_close(h0, a @ w0.t(), what="h0")
            assert_close(h1, a @ w1.t(), what="h1")
        else:
            w, b, gamma, res = rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(N, seed=4), rnd(M, N, seed=5)
            ps = (torch.arange(M // S) % 3 != 1).float() / 0.66
            y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            out = hip.gemm_nt(dev_bf16(a), [dev_bf16(w)], [dev_bf16(b)], epilogue=hip.EPI_RESID, resid=dev_bf16(res),
                              gamma=dev_bf16(gamma), rowscale=ps.to(DEV), rows_per_sample=S, h0=y)
            branch = a @ w.t() + b
            assert_close(y, branch, what="branch output")
            assert_close(out, res + ps.repeat_interleave(S)[:, None] * gamma * branch, what="resid")
    finally:
        hip.lib().op_gemm_set_tile(51)
        hip.lib().op_gemm_set_tile(old)


@pytest.mark.parametrize("shape", [(100, 512, 4096, 10), (257, 1536, 1536, 257), (514, 1536, 6144, 257), (1000, 4608, 1536, 8)])
@pytest.mark.parametrize("epi", ["bias", "resid"])
def test_gemm_small_m_split_k_with_epilogue_fold(epi, shape):
    """M <= 1024 (the leftover rows of a tail-rows split; batch-1..3 feature extraction at 257 tokens): the launch is
    latency-bound on K with most CUs idle, so K is split and the epilogue (bias / residual + layer scale + drop-path + saved
    branch output) is applied by the fold kernel."""
    hip = hipmod()
    M, N, K, S = shape
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.03), rnd(N, seed=3)
    if epi == "bias":
        if N == 4608:  # three weight segments with their own biases (the q | k | v launch; k has no bias)
            ws = [dev_bf16(w[i * 1536:(i + 1) * 1536]) for i in range(3)]
            bs = [dev_bf16(b[:1536]), None, dev_bf16(b[3072:])]
            out = hip.gemm_nt(dev_bf16(a), ws, bs)
            ref = a @ w.t() + torch.cat([b[:1536], torch.zeros(1536), b[3072:]])
        else:
            out = hip.gemm_nt(dev_bf16(a), [dev_bf16(w)], [dev_bf16(b)])
            ref = a @ w.t() + b
        assert_close(out, ref, what="bias")
    else:
        gamma, res = rnd(N, seed=4), rnd(M, N, seed=5)
        ps = (torch.arange(-(-M // S)) % 3 != 1).float() / 0.66
        y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        out = hip.gemm_nt(dev_bf16(a), [dev_bf16(w)], [dev_bf16(b)], epilogue=hip.EPI_RESID, resid=dev_bf16(res),
                          gamma=dev_bf16(gamma), rowscale=ps.to(DEV), rows_per_sample=S, h0=y)
        branch = a @ w.t() + b
        assert_close(y, branch, what="branch output")
        assert_close(out, res + ps.repeat_interleave(S)[:M, None] * gamma * branch, what="resid")


# ---------------------------------------------------------------------------------------------------------------------
# fp8 (e4m3) variant of the FFN GEMMs (BASELINE configs[4], opt-in).  Stated tolerance of the variant: rel-Frobenius <= 5e-2
# of the bf16 kernel's result (per-row scaled e4m3 has 3 mantissa bits: ~3 % RMS per operand element, averaged over K);
# the KERNEL itself must be exact: against the fp32 product of the dequantised operands it meets the bf16 output tolerance.
# ---------------------------------------------------------------------------------------------------------------------
FP8_TOL = 5e-2


def _deq(q, s):
    return q.view(torch.float8_e4m3fn).float() * s[:, None]


@pytest.mark.parametrize("rows,cols", [(5, 128), (130, 1536), (33, 6144), (64, 8192), (7, 264), (37, 12288), (6, 8200)])
def test_fp8_row_quantisation(rows, cols):
    hip = hipmod()
    x = rnd(rows, cols, seed=1, scale=3.0)
    x[0] = 0.0  # an all-zero row must not divide by zero
    q, s = hip.quant_fp8_rows(dev_bf16(x))
    amax = x.abs().amax(dim=1)
    want_s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(s.cpu(), want_s, rtol=1e-6)
    want_q = (x / want_s[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    got = q.cpu().view(torch.float8_e4m3fn).float()
    # identical e4m3 codes except where x * (1 / s) and x / s straddle a rounding boundary (bf16 inputs sit on a coarse grid, so
    # exact ties are not rare): those few land on the neighbouring code
    diff = got != want_q.float()
    assert diff.float().mean() < 1e-2
    assert ((got - want_q.float()).abs() <= 0.126 * want_q.float().abs() + 2 ** -9)[diff].all()
    assert rel_fro(_deq(q.cpu(), s.cpu()), x) < 4e-2 or rows == 1
    assert got[0].abs().max() == 0


@pytest.mark.parametrize("M,N,K", [(200, 256, 128), (515, 1536, 6144), (130, 1536, 1536), (257, 520, 256)])
def test_fp8_gemm_bias_and_residual(M, N, K):
    hip = hipmod()
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, gamma = rnd(N, seed=3), rnd(N, seed=4)
    resid = rnd(M, N, seed=5)
    rps = 5
    ps = (torch.arange((M + rps - 1) // rps) % 3 != 0).float() / 0.7
    xq, xs = hip.quant_fp8_rows(dev_bf16(x))
    wq, ws = hip.quant_fp8_rows(dev_bf16(w))
    exact = _deq(xq.cpu(), xs.cpu()) @ _deq(wq.cpu(), ws.cpu()).t() + bias
    out = hip.gemm_nt_fp8(xq, xs, [wq], [ws], bias=dev_bf16(bias))
    assert_close(out, exact, what="fp8 kernel vs dequantised operands")
    ref16 = hip.gemm_nt(dev_bf16(x), [dev_bf16(w)], [dev_bf16(bias)])
    assert rel_fro(out.float(), ref16.float()) <= FP8_TOL
    # residual epilogue: resid + rowscale * gamma * (acc + bias), branch output y
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    out_r = hip.gemm_nt_fp8(xq, xs, [wq], [ws], bias=dev_bf16(bias), epilogue=hip.EPI_RESID, resid=dev_bf16(resid),
                            gamma=dev_bf16(gamma), rowscale=ps.to(DEV), rows_per_sample=rps, h0=y)
    rs = ps.repeat_interleave(rps)[:M, None]
    assert_close(y, exact, what="fp8 branch output")
    assert_close(out_r, resid + rs * gamma * exact, what="fp8 residual epilogue")


@pytest.mark.parametrize("rows,cols", [(77, 256), (300, 1536), (129, 2048), (200, 6144), (66, 8192)])
def test_layernorm_kernels_emit_the_fp8_operand_of_the_next_gemm(rows, cols):
    """Round 5: op_layernorm_fwd_q8 / op_ln_geglu_fwd_q8 write, next to their bf16 output, the row-quantised e4m3 copy the fp8 FFN GEMMs
    read -- bit-identical (codes and scales) to op_quant_fp8_rows of that bf16 output, and the bf16 output / statistics identical to the
    plain kernels': the quantisation passes of the fp8 forward (0.22 ms per layer and stream at 50 240 rows) are gone, not moved."""
    hip = hipmod()
    x = dev_bf16(rnd(rows, cols, seed=1, scale=2.0))
    x[3] = 0.0
    w, b = dev_bf16(1 + 0.1 * rnd(cols, seed=2)), dev_bf16(0.1 * rnd(cols, seed=3))
    y0, m0, r0 = hip.layernorm_fwd(x, w, b)
    y1, m1, r1, (q, qs) = hip.layernorm_fwd(x, w, b, q8=True)
    assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(r0, r1)
    q_ref, s_ref = hip.quant_fp8_rows(y0)
    assert torch.equal(q, q_ref) and torch.equal(qs, s_ref)
    # a zero-input row: LayerNorm gives the bias there; an all-zero OUTPUT row (no affine) must not divide by zero
    y2, _, _, (q2, qs2) = hip.layernorm_fwd(torch.zeros_like(x), None, None, q8=True)
    assert float(y2.abs().max()) == 0 and float(q2.float().abs().max()) == 0 and torch.equal(qs2, torch.ones_like(qs2))
    h = dev_bf16(rnd(rows, 2 * cols, seed=4))
    h0, h1 = h[:, :cols], h[:, cols:]
    g0, gm0, gr0 = hip.ln_geglu_fwd(h0, h1, w, b)
    gq = (torch.empty(rows, cols, dtype=torch.uint8, device=DEV), torch.empty(rows, dtype=torch.float32, device=DEV))
    g1, gm1, gr1 = hip.ln_geglu_fwd(h0, h1, w, b, q8=gq)
    assert torch.equal(g0, g1) and torch.equal(gm0, gm1) and torch.equal(gr0, gr1)
    q_ref, s_ref = hip.quant_fp8_rows(g0)
    assert torch.equal(gq[0], q_ref) and torch.equal(gq[1], s_ref)


@pytest.mark.parametrize("M,N,K", [(16384, 1024, 256), (11000, 1536, 1536), (8200, 3072, 768), (11009, 1536, 6144)])
def test_fp8_four_wave_kernel_on_launches_that_fill_the_chip(M, N, K):
    """Round 5: gemm256f8_kernel -- 256 x 256 tiles, four waves, the skeleton of the bf16 production kernel -- takes every fp8 launch with
    >= 256 tiles (N % 256 == 0, K % 256 == 0; plain / bias and residual epilogues).  Against the exact product of the DEQUANTISED operands
    (fp32), against the 128 x 128 kernel on the same quantised operands (same numbers, other summation order: equal up to the last bf16
    digit of a few elements) and against the bf16 GEMM (the variant's tolerance); ragged last M-tile, per-sample row scales, second output."""
    hip = hipmod()
    assert ((M + 255) // 256) * (N // 256) >= 256
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, gamma = rnd(N, seed=3), rnd(N, seed=4)
    resid = rnd(M, N, seed=5)
    rps = 7
    ps = (torch.arange((M + rps - 1) // rps) % 3 != 0).float() / 0.7
    xq, xs = hip.quant_fp8_rows(dev_bf16(x))
    wq, ws = hip.quant_fp8_rows(dev_bf16(w))
    exact = _deq(xq.cpu(), xs.cpu()) @ _deq(wq.cpu(), ws.cpu()).t() + bias
    res = {}
    for small in (0, 1):
        hip.TUNE.fp8_small = small
        try:
            y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            res[small] = (hip.gemm_nt_fp8(xq, xs, [wq], [ws], bias=dev_bf16(bias)), hip.gemm_nt_fp8(xq, xs, [wq], [ws]),
                          hip.gemm_nt_fp8(xq, xs, [wq], [ws], bias=dev_bf16(bias), epilogue=hip.EPI_RESID, resid=dev_bf16(resid),
                                          gamma=dev_bf16(gamma), rowscale=ps.to(DEV), rows_per_sample=rps, h0=y), y)
            torch.cuda.synchronize()
        finally:
            hip.TUNE.fp8_small = 0
    out, plain, out_r, y = res[0]
    assert_close(out, exact, what="fp8 four-wave kernel vs dequantised operands")
    assert_close(plain, exact - bias, what="fp8 four-wave kernel, no bias")
    rs = ps.repeat_interleave(rps)[:M, None]
    assert_close(y, exact, what="fp8 four-wave branch output")
    assert_close(out_r, resid + rs * gamma * exact, what="fp8 four-wave residual epilogue")
    for a, b in zip(res[0], res[1]):
        d = (a.float() - b.float()).abs()
        assert float(d.max()) <= 2.0 ** -6 * float(b.float().abs().max()) and float((d > 0).float().mean()) < 2e-2, (float(d.max()), float((d > 0).float().mean()))
    ref16 = hip.gemm_nt(dev_bf16(x), [dev_bf16(w)], [dev_bf16(bias)])
    assert rel_fro(out.float(), ref16.float()) <= FP8_TOL


@pytest.mark.parametrize("M,F_,K", [(200, 256, 128), (515, 1024, 256), (130, 6144, 1536)])
def test_fp8_gemm_geglu(M, F_, K):
    hip = hipmod()
    x = rnd(M, K, seed=1)
    w0, w1 = rnd(F_, K, seed=2, scale=K ** -0.5), rnd(F_, K, seed=3, scale=K ** -0.5)
    xq, xs = hip.quant_fp8_rows(dev_bf16(x))
    (w0q, w0s), (w1q, w1s) = hip.quant_fp8_rows(dev_bf16(w0)), hip.quant_fp8_rows(dev_bf16(w1))
    xd = _deq(xq.cpu(), xs.cpu())
    e0, e1 = xd @ _deq(w0q.cpu(), w0s.cpu()).t(), xd @ _deq(w1q.cpu(), w1s.cpu()).t()
    h0 = torch.empty(M, F_, dtype=torch.bfloat16, device=DEV)
    h1 = torch.empty_like(h0)
    g = hip.gemm_nt_fp8(xq, xs, [w0q, w1q], [w0s, w1s], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1)
    assert_close(h0, e0, what="fp8 geglu h0")
    assert_close(h1, e1, what="fp8 geglu h1")
    assert_close(g, F.gelu(e0) * e1, fro=6e-3, mx=2e-2, what="fp8 geglu out")
    g16 = hip.gemm_nt(dev_bf16(x), [dev_bf16(w0), dev_bf16(w1)], epilogue=hip.EPI_GEGLU)
    assert rel_fro(g.float(), g16.float()) <= 1.5 * FP8_TOL  # product of two quantised projections


def test_fp8_ffn_branch_forward_close_to_bf16_and_backward_unchanged():
    """ops.set_fp8_ffn: the fused FFN branch with its two forward GEMMs on the fp8 path stays within the variant's tolerance of
    the bf16 branch, and the backward pass (bf16 kernels on the saved bf16 activations) still produces every gradient."""
    from one_peace_amd import ops
    H, Fd, B, S = 256, 512, 3, 40
    torch.manual_seed(0)
    params = [dev_bf16(1 + 0.1 * rnd(H, seed=1)), dev_bf16(0.1 * rnd(H, seed=2)), dev_bf16(rnd(Fd, H, seed=3, scale=H ** -0.5)),
              dev_bf16(rnd(Fd, H, seed=4, scale=H ** -0.5)), dev_bf16(1 + 0.1 * rnd(Fd, seed=5)), dev_bf16(0.1 * rnd(Fd, seed=6)),
              dev_bf16(rnd(H, Fd, seed=7, scale=Fd ** -0.5)), dev_bf16(0.1 * rnd(H, seed=8)), dev_bf16(0.5 + 0.1 * rnd(H, seed=9))]
    x = dev_bf16(rnd(B, S, H, seed=10))
    outs, grads = [], []
    for on in (False, True):
        old = ops.set_fp8_ffn(on)
        try:
            ps = [p.clone().requires_grad_(True) for p in params]
            xi = x.clone().requires_grad_(True)
            y = ops.ffn_branch(xi, None, ps, save_acts=True)
            y.float().pow(2).sum().backward()
            outs.append(y.detach().float())
            grads.append([xi.grad.float()] + [p.grad.float() for p in ps])
        finally:
            ops.set_fp8_ffn(old)
    assert rel_fro(outs[1] - x.float(), outs[0] - x.float()) <= 1.5 * FP8_TOL   # the branch itself, without the residual
    for g8, g16 in zip(grads[1], grads[0]):
        assert g8.isfinite().all() and rel_fro(g8, g16) <= 0.15  # same bf16 backward kernels on slightly different activations


def test_rows_gather_and_merge_of_kept_samples():
    """op_rows_gather / op_rows_merge (stochastic depth on the samples a branch keeps): three segments with their own tokens per
    sample, kept lists incl. 'all', 'one' and 'the last', packed rows rounded up to 64 with ZERO rows behind every segment;
    the merge replaces exactly the kept samples' rows, out of place and in place."""
    hip = hipmod()
    cols = 136
    segs_full = [(0, 5, 4), (20, 7, 3), (41, 3, 6)]          # (first row, tokens per sample, samples)
    kept = [[0, 2, 3], [1], [0, 1, 2, 3, 4, 5]]
    rows = 41 + 3 * 6
    lists, bases = hip.pack_kept_lists([[(r0, S, n, k) for (r0, S, n), k in zip(segs_full, kept)]])
    kr = hip.KeptRows([(r0, S, n, k) for (r0, S, n), k in zip(segs_full, kept)], lists.to("cuda"), bases[0], rows, 1.25, pad=64)
    assert kr.dst_rows == [64, 64, 64] and kr.total == 192 and kr.n_kept == [3, 1, 6]
    x = dev_bf16(rnd(rows, cols, seed=3))
    packed = hip.rows_gather(x, kr)
    want = torch.zeros(192, cols, dtype=torch.bfloat16, device="cuda")
    for i, ((r0, S, n), k) in enumerate(zip(segs_full, kept)):
        for j, smp in enumerate(k):
            want[kr.dst_row0[i] + j * S:kr.dst_row0[i] + (j + 1) * S] = x[r0 + smp * S:r0 + (smp + 1) * S]
    assert torch.equal(packed, want)
    upd = dev_bf16(rnd(192, cols, seed=4))
    merged = hip.rows_merge(x, upd, kr)
    ref = x.clone()
    for i, ((r0, S, n), k) in enumerate(zip(segs_full, kept)):
        for j, smp in enumerate(k):
            ref[r0 + smp * S:r0 + (smp + 1) * S] = upd[kr.dst_row0[i] + j * S:kr.dst_row0[i] + (j + 1) * S]
    assert torch.equal(merged, ref)
    inplace = x.clone()
    hip.rows_merge(inplace, upd, kr, out=inplace)
    assert torch.equal(inplace, ref)
    torch.cuda.synchronize()


def _kept_rows_case(hip, S, n_samples, keep_seed, pad, cols_unused=None):
    """Three segments (tokens per sample S[i], samples n_samples[i]) with random kept sets (at least one sample each)."""
    g = torch.Generator().manual_seed(keep_seed)
    segs, r0 = [], 0
    for s, n in zip(S, n_samples):
        k = [j for j in range(n) if torch.rand((), generator=g).item() < 0.7] or [n - 1]
        segs.append((r0, s, n, k))
        r0 += s * n
    lists, bases = hip.pack_kept_lists([segs])
    return hip.KeptRows(segs, lists.to("cuda"), bases[0], r0, 1.25, pad=pad), segs, r0


@pytest.mark.parametrize("S,n,pad,cols", [((5, 7, 3), (4, 3, 6), 64, 136), ((64, 257, 250), (8, 9, 7), 256, 1536)])
def test_row_tables_read_and_write_the_full_matrix_like_the_packed_copies(S, n, pad, cols):
    """(ABI 9) op_rows_map + x_rows of op_layernorm_fwd / _bwd + dout_rows of op_resid_bwd + op_rows_merge without upd: a residual branch
    on the samples stochastic depth keeps, reading and writing the FULL matrix through the row table -- bit for bit what the packed copies
    (op_rows_gather before, op_rows_merge behind) give, incl. the zero rows behind every rounded-up segment."""
    hip = hipmod()
    kr, segs, rows = _kept_rows_case(hip, S, n, 5, pad)
    m = kr.rowmap()
    want = torch.full((kr.total,), -1, dtype=torch.int32)
    for i, (r0, s, _, k) in enumerate(segs):
        for j, smp in enumerate(k):
            want[kr.dst_row0[i] + j * s:kr.dst_row0[i] + (j + 1) * s] = torch.arange(r0 + smp * s, r0 + (smp + 1) * s, dtype=torch.int32)
    assert torch.equal(m.cpu(), want)
    x = dev_bf16(rnd(rows, cols, seed=3, scale=2.0))
    w, b = dev_bf16(1 + 0.1 * rnd(cols, seed=2)), dev_bf16(0.1 * rnd(cols, seed=4))
    xg = hip.rows_gather(x, kr)
    for gelu in (False, True):
        y0, mean0, rstd0 = hip.layernorm_fwd(xg, w, b, gelu=gelu)
        y1, mean1, rstd1 = hip.layernorm_fwd(x, w, b, gelu=gelu, x_rows=m)
        assert torch.equal(y0, y1) and torch.equal(mean0, mean1) and torch.equal(rstd0, rstd1)
    # backward: dy of the packed rows (zero where no sample is), the residual-path gradient `add` of the full matrix
    dy = dev_bf16(rnd(kr.total, cols, seed=6)) * (m >= 0).unsqueeze(1).to(torch.bfloat16)
    add = dev_bf16(rnd(rows, cols, seed=7))
    dxp, dw0, db0 = hip.layernorm_bwd(dy, xg, w, b, mean0, rstd0, add=hip.rows_gather(add, kr))
    ref = hip.rows_merge(add, dxp, kr)
    inplace = add.clone()
    _, dw1, db1 = hip.layernorm_bwd(dy, x, w, b, mean1, rstd1, add=inplace, dx=inplace, x_rows=m)
    assert torch.equal(inplace, ref) and torch.equal(dw0, dw1) and torch.equal(db0, db1)
    out = torch.empty_like(add)  # out of place: the mapped rows by the kernel, the dropped samples' rows by the copy
    hip.layernorm_bwd(dy, x, w, b, mean1, rstd1, add=add, dx=out, x_rows=m)
    hip.rows_merge(add, None, kr, out=out)
    assert torch.equal(out, ref)
    # residual-branch backward reading dout through the table
    gamma = dev_bf16(0.3 * rnd(cols, seed=8))
    scale = torch.full((kr.total + 8,), 1.25, dtype=torch.float32, device="cuda")
    g00, g01 = torch.empty(cols, dtype=torch.float32, device="cuda"), torch.empty(cols, dtype=torch.float32, device="cuda")
    for kw0, kw1 in ((dict(dbias=True), dict(dbias=True)), (dict(dbias=True, g0=g00), dict(dbias=True, g0=g01))):
        d0, _, b0_ = hip.resid_bwd(hip.rows_gather(add, kr), None, gamma, scale, 1, **kw0)
        d1, _, b1_ = hip.resid_bwd(add, None, gamma, scale, 1, dout_rows=m, **kw1)
        assert torch.equal(d0, d1) and torch.equal(b0_, b1_)
    assert torch.equal(g00, g01)
    torch.cuda.synchronize()


@pytest.mark.parametrize("S,n,pad,N,K,bias,y", [
    ((5, 7, 3), (4, 3, 6), 64, 128, 128, True, True),         # 128 x 128 tiles, a few M-tiles: split-K fold with the epilogue
    ((64, 257, 250), (8, 9, 7), 256, 1536, 1536, True, False),  # the persistent four-wave kernel (K <= 2048)
    ((64, 257, 250), (8, 9, 7), 256, 1536, 6144, True, True),   # gemm256v (K > 2048), with the branch output
    ((64, 257, 250), (8, 9, 7), 64, 1536, 1536, False, False),  # M % 256 != 0: tail rows as a second launch
    ((16, 17, 20), (6, 6, 6), 64, 256, 256, True, False),     # fewer than 256 tiles of 256 x 256
])
def test_residual_epilogue_through_a_row_table(S, n, pad, N, K, bias, y):
    """(ABI 9) resid_rows of op_gemm_nt: out[rows[m]] = resid[rows[m]] + rowscale * gamma * (A W^T + b)[m] straight into the full matrix
    (+ the dropped samples' rows copied) == the packed launch on op_rows_gather's copy followed by op_rows_merge, bit for bit, on every
    kernel the launch plan can pick."""
    hip = hipmod()
    kr, segs, rows = _kept_rows_case(hip, S, n, 9, pad)
    m = kr.rowmap()
    x = dev_bf16(rnd(rows, N, seed=1))
    A = dev_bf16(rnd(kr.total, K, seed=2, scale=0.5)) * (m >= 0).unsqueeze(1).to(torch.bfloat16)
    W, bvec, gamma = dev_bf16(rnd(N, K, seed=3, scale=K ** -0.5)), dev_bf16(0.1 * rnd(N, seed=4)), dev_bf16(0.3 * rnd(N, seed=5))
    scale = torch.full((kr.total + 8,), 1.25, dtype=torch.float32, device="cuda")
    h0a = torch.empty(kr.total, N, dtype=torch.bfloat16, device="cuda") if y else None
    h0b = torch.empty(kr.total, N, dtype=torch.bfloat16, device="cuda") if y else None
    packed = hip.gemm_nt(A, [W], [bvec] if bias else None, epilogue=hip.EPI_RESID, resid=hip.rows_gather(x, kr), gamma=gamma, rowscale=scale,
                         rows_per_sample=1, h0=h0a)
    ref = hip.rows_merge(x, packed, kr)
    out = torch.empty_like(x)
    hip.gemm_nt(A, [W], [bvec] if bias else None, epilogue=hip.EPI_RESID, resid=x, gamma=gamma, rowscale=scale, rows_per_sample=1, h0=h0b,
                out=out, resid_rows=m)
    hip.rows_merge(x, None, kr, out=out)
    assert torch.equal(out, ref)
    if y:
        assert torch.equal(h0a, h0b)
    # in place on the residual stream (out = resid): the dropped samples' rows are simply left alone
    inplace = x.clone()
    hip.gemm_nt(A, [W], [bvec] if bias else None, epilogue=hip.EPI_RESID, resid=inplace, gamma=gamma, rowscale=scale, rows_per_sample=1,
                out=inplace, resid_rows=m)
    assert torch.equal(inplace, ref)
    torch.cuda.synchronize()


def test_grouped_residual_launch_through_row_tables():
    """(ABI 9) resid_rows of op_gemm_nt_grouped: the three modality FFN down-projections of a layer as ONE launch, each problem writing the
    rows of ITS kept samples into the shared full matrix == the packed grouped launch + op_rows_merge, bit for bit."""
    hip = hipmod()
    N, K = 1536, 512
    kr, segs, rows = _kept_rows_case(hip, (64, 257, 250), (8, 9, 7), 13, 256)
    m = kr.rowmap()
    x = dev_bf16(rnd(rows, N, seed=1))
    A = dev_bf16(rnd(kr.total, K, seed=2, scale=0.5)) * (m >= 0).unsqueeze(1).to(torch.bfloat16)
    Ws = [dev_bf16(rnd(N, K, seed=10 + i, scale=K ** -0.5)) for i in range(3)]
    bs = [dev_bf16(0.1 * rnd(N, seed=20 + i)) for i in range(3)]
    gamma = dev_bf16(0.3 * rnd(N, seed=5))
    scale = torch.full((kr.total + 8,), 1.25, dtype=torch.float32, device="cuda")
    rs = [slice(kr.dst_row0[i], kr.dst_row0[i] + kr.dst_rows[i]) for i in range(3)]
    xg = hip.rows_gather(x, kr)
    packed = torch.empty_like(xg)
    got = hip.gemm_nt_grouped([A[r] for r in rs], Ws, biases=bs, outs=[packed[r] for r in rs], epilogue=hip.EPI_RESID,
                              resids=[xg[r] for r in rs], gammas=[gamma] * 3, rowscales=[scale] * 3, rows_per_sample=[1, 1, 1])
    assert got is not None
    ref = hip.rows_merge(x, packed, kr)
    out = torch.empty_like(x)
    got = hip.gemm_nt_grouped([A[r] for r in rs], Ws, biases=bs, outs=[out] * 3, epilogue=hip.EPI_RESID, resids=[x] * 3, gammas=[gamma] * 3,
                              rowscales=[scale] * 3, rows_per_sample=[1, 1, 1], resid_rows=[m[r] for r in rs])
    assert got is not None
    hip.rows_merge(x, None, kr, out=out)
    assert torch.equal(out, ref)
    torch.cuda.synchronize()


@pytest.mark.parametrize("kept", [[[0, 1, 2, 3]], [[2]], [[0, 3]]])
def test_row_tables_edge_cases_one_segment(kept):
    """(ABI 9) One segment of four 256-token samples: every sample kept (a table without -1 entries that is the identity), a single
    sample kept, the first and the last kept -- LayerNorm through the table, the residual epilogue into the full matrix and the copy of the
    dropped samples' rows against plain torch indexing."""
    hip = hipmod()
    S, n, cols = 256, 4, 512
    segs = [(0, S, n, kept[0])]
    lists, bases = hip.pack_kept_lists([segs])
    kr = hip.KeptRows(segs, lists.to("cuda"), bases[0], S * n, 1.0, pad=256)
    m = kr.rowmap()
    want = torch.cat([torch.arange(j * S, (j + 1) * S, dtype=torch.int32) for j in kept[0]])
    assert kr.total == len(kept[0]) * S and torch.equal(m.cpu(), want)
    x = dev_bf16(rnd(S * n, cols, seed=1, scale=2.0))
    w, b = dev_bf16(1 + 0.1 * rnd(cols, seed=2)), dev_bf16(0.1 * rnd(cols, seed=3))
    y, mean, rstd = hip.layernorm_fwd(x, w, b, x_rows=m)
    y_ref, mean_ref, rstd_ref = hip.layernorm_fwd(x[m.long()].contiguous(), w, b)
    assert torch.equal(y, y_ref) and torch.equal(mean, mean_ref) and torch.equal(rstd, rstd_ref)
    W = dev_bf16(rnd(cols, cols, seed=4, scale=cols ** -0.5))
    out = torch.empty_like(x)
    hip.gemm_nt(y, [W], epilogue=hip.EPI_RESID, resid=x, out=out, resid_rows=m)
    hip.rows_merge(x, None, kr, out=out)
    ref = x.clone()
    ref[m.long()] = hip.gemm_nt(y_ref, [W], epilogue=hip.EPI_RESID, resid=x[m.long()].contiguous())
    assert torch.equal(out, ref)
    torch.cuda.synchronize()
