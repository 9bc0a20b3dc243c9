"""The four-wave flavour of the 256x256 NT kernel (tune bits 2-3 = 3) against the eight-wave full-line flavour and the planner's own choice: bit-equality and time, all four epilogues.  MS=8192,16000 selects other row counts."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip
from tools.bench_ops import timeit

bf = dict(dtype=torch.bfloat16, device="cuda")
IT = int(os.environ.get("ITERS", "100"))
torch.manual_seed(0)
L = hip.lib()
L.op_gemm_set_tile(2)


def run(flavour, fn):
    L.op_gemm_set_tile(20 + flavour)
    out = fn()
    torch.cuda.synchronize()
    return out


def case(name, M, N, K, fn, flops, time_it=True):
    a = run(1, fn)
    b = run(3, fn)
    a = a if isinstance(a, (tuple, list)) else (a,)
    b = b if isinstance(b, (tuple, list)) else (b,)
    same = all(torch.equal(x, y) for x, y in zip(a, b))
    fin = all(bool(torch.isfinite(y.float()).all()) for y in b)
    line = "%-34s M=%6d N=%5d K=%5d  bit-identical=%s finite=%s" % (name, M, N, K, same, fin)
    if time_it:
        L.op_gemm_set_tile(21)
        t1 = timeit(fn, iters=IT, warmup=20)
        L.op_gemm_set_tile(23)
        t3 = timeit(fn, iters=IT, warmup=20)
        L.op_gemm_set_tile(21)
        t1b = timeit(fn, iters=IT, warmup=5)
        L.op_gemm_set_tile(0)
        L.op_gemm_set_tile(22)
        ta = timeit(fn, iters=IT, warmup=5)
        L.op_gemm_set_tile(2)
        line += "  eight waves %.3f / %.3f ms (%.0f TF/s)  four waves %.3f ms (%.0f TF/s)  %+.1f%%" % (
            t1, t1b, flops / min(t1, t1b) / 1e9, t3, flops / t3 / 1e9, 100.0 * (min(t1, t1b) / t3 - 1.0))
        line += "  auto dispatch %.3f" % ta
    print(line, flush=True)
    if not same:
        d = (a[0].float() - b[0].float()).abs()
        print("   max abs diff %.4g at %s" % (float(d.max()), [int(v) for v in torch.unravel_index(d.argmax(), d.shape)]))


def _with_h(fn, M, N):
    h0 = torch.empty(M, N, **bf)
    return fn(h0), h0


def _with_h2(fn, M, N):
    h0, h1 = torch.empty(M, N, **bf), torch.empty(M, N, **bf)
    return fn(h0, h1), h0, h1


H, F = 1536, 6144
for M in ([int(v) for v in os.environ["MS"].split(",")] if os.environ.get("MS") else (300, 2048, 128 * 257, 128 * 256)):
    big = M > 4096
    x = torch.randn(M, H, **bf)
    xf = torch.randn(M, F, **bf)
    wqkv = [torch.randn(H, H, **bf) * 0.03 for _ in range(3)]
    bq = [torch.randn(H, **bf), None, torch.randn(H, **bf)]
    w0, w1 = torch.randn(F, H, **bf) * 0.03, torch.randn(F, H, **bf) * 0.03
    w2 = torch.randn(H, F, **bf) * 0.02
    b2, gamma = torch.randn(H, **bf), torch.randn(H, **bf)
    res = torch.randn(M, H, **bf)
    ps = torch.rand(M // 2 + 1, device="cuda")
    case("qkv (3 segments, bias)", M, 3 * H, H, lambda: hip.gemm_nt(x, wqkv, bq, n_seg=H, N=3 * H), 2.0 * M * 3 * H * H, big)
    case("dgrad K=1536 N=1536 bias only", M, H, H, lambda: hip.gemm_nt(x, [wqkv[0]], [b2]), 2.0 * M * H * H, big)
    case("out-proj residual", M, H, H, lambda: _with_h(lambda h0: hip.gemm_nt(x, [wqkv[0]], [b2], epilogue=hip.EPI_RESID, resid=res,
                                                                                   gamma=gamma, rowscale=ps, rows_per_sample=2, h0=h0), M, H),
         2.0 * M * H * H, big)
    case("GeGLU up-projection", M, F, H, lambda: _with_h2(lambda h0, h1: hip.gemm_nt(x, [w0, w1], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1), M, F), 4.0 * M * F * H, big)
    case("down-projection residual K=6144", M, H, F, lambda: hip.gemm_nt(xf, [w2], [b2], epilogue=hip.EPI_RESID, resid=res, gamma=gamma),
         2.0 * M * H * F, big)
    case("dgrad K=6144 N=1536 no bias", M, H, F, lambda: hip.gemm_nt(xf, [w2]), 2.0 * M * H * F, big)
    if M <= 4096:
        wk = torch.randn(512, 768, **bf) * 0.05
        xk = torch.randn(M, 768, **bf)
        case("K=768 (shortest K), fp32 out", M, 512, 768, lambda: hip.gemm_nt(xk, [wk], epilogue=hip.EPI_F32, alpha=torch.full((1,), 0.5, device="cuda")),
             2.0 * M * 512 * 768, False)
L.op_gemm_set_tile(22)
L.op_gemm_set_tile(0)
