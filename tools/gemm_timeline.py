"""Round 3: where a one-tile workgroup of gemm256v_kernel spends its time.  Builds an instrumented copy of the library
(csrc/gemm.hip with -DOP_GEMM_TIMELINE: five s_memrealtime stamps per workgroup, 100 MHz) next to the production one, runs
single launches, and reports per shape the medians of

    set-up   kernel entry -> first operand tiles requested (index math, descriptors, 32 LDS-DMA ops, accumulator clear)
    wait     -> first tiles have landed, barrier passed
    loop     -> last MFMA retired
    epilogue -> last store ISSUED;   drain -> last store completed (extra s_waitcnt, instrumented build only)
    gap      end of a workgroup -> entry of the next workgroup on the SAME CU (dispatch)

    python tools/gemm_timeline.py build      (container or GPU box: compiles lib/libonepeace_hip_timeline.so)
    python tools/gemm_timeline.py            (GPU)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "one-peace_amd")
TL_LIB = os.environ.get("TL_LIB") or os.path.join(PKG, "lib", "libonepeace_hip_timeline.so")


def build():
    sys.path.insert(0, PKG)
    import build as B
    objdir = os.path.join(PKG, "build")
    B.build(verbose=False)  # production objects of every other source
    obj = os.path.join(objdir, "gemm_timeline.o")
    extra = ["-D" + d for d in os.environ.get("DEFS", "").split(",") if d]  # e.g. DEFS=OP_EXP_EPI=1 (epilogue ablations)
    subprocess.run([B.HIPCC] + B.FLAGS + ["-DOP_GEMM_TIMELINE"] + extra + ["-c", os.path.join(B.CSRC, "gemm.hip"), "-o", obj], check=True)
    objs = [os.path.join(objdir, f[:-4] + ".o") for f in B.sources() if f != "gemm.hip"] + [obj]
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TL_LIB] + objs, check=True)
    print("built", TL_LIB)


if len(sys.argv) > 1 and sys.argv[1] == "build":
    build()
    sys.exit(0)

os.environ["ONEPEACE_HIP_LIB"] = TL_LIB
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from one_peace_amd import hip  # noqa: E402

bf = dict(dtype=torch.bfloat16, device="cuda")
L = hip.lib()
raw = ctypes.CDLL(TL_LIB)
raw.op_debug_gemm_timeline.argtypes = [ctypes.c_void_p]
T = hip.TUNE
H, F = 1536, 6144
torch.manual_seed(0)


def run(name, M, N, K, fn):
    tiles = ((M + 255) // 256) * (N // 256)
    buf = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
    T.reset()
    T.tile_mode, T.fullline, T.sched = 2, 3, 3
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    raw.op_debug_gemm_timeline(ctypes.c_void_p(buf.data_ptr()))
    fn()
    torch.cuda.synchronize()
    raw.op_debug_gemm_timeline(None)
    T.reset()
    d = buf.view(tiles, 8).cpu()
    d = d[d[:, 1] != 0]
    hw = d[:, 0]
    cu = ((hw >> 32) & 0xf) * 65536 + (hw & 0xff00)  # XCC id | SE / SH / CU id bits of HW_ID (wave / simd / pipe dropped)
    if os.environ.get('RAW'):
        print([hex(int(v)) for v in hw[:12]])
    t = d[:, 1:7].double() * 0.01  # us
    index_math = float(((d[:, 7] - d[:, 1]).double() * 0.01).median())  # kernel entry -> first LDS-DMA instruction
    t0 = t[:, 0].min()
    med = lambda x: float(x.median())
    seg = dict(setup=med(t[:, 1] - t[:, 0]), wait=med(t[:, 2] - t[:, 1]), loop=med(t[:, 3] - t[:, 2]), epilogue=med(t[:, 4] - t[:, 3]),
               drain=med(t[:, 5] - t[:, 4]))
    gaps = []
    for c in cu.unique():
        rows = t[cu == c]
        rows = rows[rows[:, 0].argsort()]
        if rows.shape[0] > 1:
            gaps.append(rows[1:, 0] - rows[:-1, 5])
    gaps = torch.cat(gaps) if gaps else torch.zeros(1, dtype=torch.double)
    total = float(t[:, 5].max() - t0)
    print("%-28s M=%6d N=%5d K=%5d | %4d workgroups on %3d CUs, launch %.1f us | set-up %.2f (of which kernel arguments + index math %.2f)  wait %.2f  loop %.2f  epilogue %.2f  drain %.2f | "
          "gap to next workgroup on the CU: median %.2f  p90 %.2f us | first entry spread %.2f us" % (
              name, M, N, K, d.shape[0], len(cu.unique()), total, seg["setup"], index_math, seg["wait"], seg["loop"], seg["epilogue"], seg["drain"],
              med(gaps), float(gaps.quantile(0.9)), float(t[:, 0].sort().values[min(255, d.shape[0] - 1)] - t0)), flush=True)


for M in [int(v) for v in os.environ.get('MS', '32768,32000').split(',')]:
    x = torch.randn(M, H, **bf)
    xf = torch.randn(M, F, **bf)
    w = torch.randn(H, H, **bf) * 0.03
    wf = torch.randn(F, H, **bf) * 0.03
    w2 = torch.randn(H, F, **bf) * 0.02
    b2, gamma, res = torch.randn(H, **bf), torch.randn(H, **bf), torch.randn(M, H, **bf)
    ps = torch.rand(M // 2 + 1, device="cuda")
    y, o_h, o_f, o_ff = torch.empty(M, H, **bf), torch.empty(M, H, **bf), torch.empty(M, F, **bf), torch.empty(M, 2 * F, **bf)
    run("plain N=1536 K=1536", M, H, H, lambda: hip.gemm_nt(x, [w], out=o_h, splitk=False))
    run("plain N=6144 K=1536", M, F, H, lambda: hip.gemm_nt(x, [wf], out=o_f, splitk=False))
    run("two-segment N=12288 K=1536", M, 2 * F, H, lambda: hip.gemm_nt(x, [wf, wf], n_seg=F, N=2 * F, out=o_ff, splitk=False))
    run("plain N=1536 K=6144", M, H, F, lambda: hip.gemm_nt(xf, [w2], out=o_h, splitk=False))
    run("resid N=1536 K=1536", M, H, H, lambda: hip.gemm_nt(x, [w], [b2], epilogue=hip.EPI_RESID, resid=res, gamma=gamma, rowscale=ps,
                                                            rows_per_sample=2, h0=y, out=o_h))
    run("resid N=1536 K=6144", M, H, F, lambda: hip.gemm_nt(xf, [w2], [b2], epilogue=hip.EPI_RESID, resid=res, gamma=gamma, rowscale=ps,
                                                            rows_per_sample=2, h0=y, out=o_h))
    del x, xf, y, o_h, o_f, o_ff, res
    torch.cuda.empty_cache()
