"""What do CUs held by ANOTHER kernel cost the NT GEMM launches of the step -- persistent workgroups against one tile per workgroup?

    python tools/cu_contention_ab.py [--held 0,8,16,32] [--reps 12] > profiles/r6_cu_contention_ab.txt

On a multi-GPU node RCCL's all-reduce kernels run beside backward (distributed.BucketedGradReducer).  A four-wave GEMM workgroup owns
its CU's whole register file and LDS, so a CU that holds a collective's workgroup is lost to the GEMM until that workgroup ends.  The
persistent NT kernel (gemm256p_kernel: one workgroup per CU walks tiles b, b + 256, ...) then has workgroups that cannot start before
another one ENDS; one-tile launches (tune sched 7) flow onto the CUs that are free.  distributed.share_cus_with_collectives() picks the
launch rule for world > 1 from THIS measurement: a stand-in for the collective (op_probe_occupy: k workgroups of 256 threads and 96 KiB
of LDS copying memory, on a stream whose CU mask allows exactly k CUs spread over the XCDs) is started, then the headline launches are
timed with both rules while it runs.  Single GPU: no RCCL involved, only the dispatch behaviour."""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402


def masked_stream(cus, total=256):
    """A HIP stream restricted to `cus` CUs spread evenly over the index range (hipExtStreamCreateWithCUMask)."""
    rt = ctypes.CDLL("libamdhip64.so")
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(cus):
        c = (i * total) // cus
        mask[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, "hipExtStreamCreateWithCUMask failed: %d" % rc
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--held", default="0,8,16,32")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--occupy-us", type=int, default=6000)
    a = ap.parse_args()
    dev = torch.device("cuda")
    P = hip.probe_lib()
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda r, c: torch.randn(r, c, generator=g, device=dev).to(torch.bfloat16)  # noqa: E731
    H, F = 1536, 6144
    # (name, M, N, K, epilogue): the NT launches of a lock-step headline layer that take the persistent kernel by default (K <= 2048)
    # plus the long-K ones (gemm256v_kernel under both rules: the control)
    shapes = [("q|k|v            K=1536", 73088, 3 * H, H), ("out-proj dgrad   K=1536", 73088, H, H), ("FFN up (image)   K=1536", 32896, 2 * F, H),
              ("FFN down dgrad   K=1536", 32896, F, H), ("FFN down (image) K=6144", 32896, H, F)]
    slab = 1 << 20
    buf = torch.zeros(64 * slab, dtype=torch.float32, device=dev)
    # the GEMMs go on a NON-BLOCKING stream: torch's default stream is the legacy null stream, which serialises with every blocking
    # stream -- the CU-masked one included (the first version of this tool measured the GEMMs AFTER the stand-in had ended)
    main = torch.cuda.Stream()
    when = torch.zeros(64 * 2, dtype=torch.int64, device=dev)
    print("# CUs held by a stand-in collective kernel vs the NT GEMM launch rule; ms per launch, median of %d (min)" % a.reps)
    print("# %-26s %6s %22s %22s %8s" % ("launch", "held", "persistent (sched 0)", "one tile/wg (sched 7)", "ratio"))
    for name, M, N, K in shapes:
        A, W, out = mk(M, K), mk(N, K), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for held in [int(x) for x in a.held.split(",")]:
            side = masked_stream(held) if held else None
            res, overlap = {}, []
            with torch.cuda.stream(main):
                for sched in (0, 7):
                    hip.TUNE.sched = sched
                    ts = []
                    for rep in range(a.reps + 2):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        if held:
                            hip._check_probe(P.op_probe_occupy(hip.ptr(buf), slab, held, a.occupy_us, hip.ptr(when), side), "op_probe_occupy")
                        time.sleep(0.0005)  # the stand-in is resident on its CUs before the GEMM is enqueued (same pause without one)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        hip.gemm_nt(A, [W], out=out)
                        e1.record()
                        e1.synchronize()
                        t1 = time.perf_counter()  # the GEMM is done; with a stand-in of occupy_us that must be BEFORE the stand-in ends
                        torch.cuda.synchronize()
                        if rep >= 2:
                            ts.append(e0.elapsed_time(e1))
                            overlap.append((t1 - t0) * 1e6 < a.occupy_us)
                    ts.sort()
                    res[sched] = (ts[len(ts) // 2], ts[0])
            hip.TUNE.sched = 0
            print("  %-26s %6d %12.4f (%7.4f) %12.4f (%7.4f) %8.3f   %s" % (name, held, res[0][0], res[0][1], res[7][0], res[7][1], res[7][0] / res[0][0],
                  "" if not held else "GEMM finished while the stand-in ran: %d / %d reps" % (sum(overlap), len(overlap))))
    print("# ratio < 1: the one-tile rule is faster under that contention.  A launch with `held` CUs taken for its whole duration:")
    print("# persistent = up to 2 T (the late workgroups start when the first ones end), one tile = T * 256 / (256 - held).")


if __name__ == "__main__":
    main()
