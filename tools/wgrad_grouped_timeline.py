"""Round 4: what the workgroups of ONE grouped weight-gradient launch (op_gemm_tn_grouped, a lock-step layer at b = 128) do and when.
Instrumented library (tools/gemm_timeline.py build; -DOP_GEMM_TIMELINE): per tile the ticket, its start, the end of its main loop
and its end (s_memrealtime, 100 MHz).  Reports tile durations per problem, the start skew inside a group of tiles that share an
operand panel, idle time, and the makespan against the sum of the work.

    python tools/gemm_timeline.py build && python tools/wgrad_grouped_timeline.py [nwg]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TL_LIB = os.environ.get("TL_LIB") or os.path.join(ROOT, "one-peace_amd", "lib", "libonepeace_hip_timeline.so")
os.environ["ONEPEACE_HIP_LIB"] = TL_LIB
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from one_peace_amd import hip  # noqa: E402

nwg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
bf = dict(dtype=torch.bfloat16, device="cuda")
hip.lib()
raw = ctypes.CDLL(TL_LIB)
raw.op_debug_gemm_timeline.argtypes = [ctypes.c_void_p]
H, F = 1536, 6144
rows = {"all": 73088, "img": 32896, "aud": 32000, "txt": 8192}
names = [("q|k|v", "all", 3 * H, H), ("out-proj", "all", H, H)]
for m in ("img", "aud", "txt"):
    names += [("%s wi" % m, m, 2 * F, H), ("%s wo" % m, m, H, F)]
probs = [(torch.randn(rows[m], o, **bf), torch.randn(rows[m], i, **bf), torch.zeros(o, i, **bf), True) for _, m, o, i in names]
order = sorted(range(len(probs)), key=lambda i: -probs[i][0].shape[0])  # the library sorts by K (stable): its problem index -> ours
for _ in range(3):
    hip.gemm_tn_grouped(probs, tune=nwg)
torch.cuda.synchronize()
NW = (nwg & 1023) or 256
buf = torch.zeros(NW * 128, dtype=torch.int64, device="cuda")
raw.op_debug_gemm_timeline(ctypes.c_void_p(buf.data_ptr()))
hip.gemm_tn_grouped(probs, tune=nwg)
torch.cuda.synchronize()
raw.op_debug_gemm_timeline(None)
d = buf.view(NW, 32, 4).cpu()
t0 = int(d[:, :, 1][d[:, :, 1] > 0].min())
recs = []  # (wg, slot, queue, qslot, prob, start, loop_end, end) in us
for w in range(NW):
    for s in range(32):
        code, a, b, c = [int(v) for v in d[w, s]]
        if a == 0:
            continue
        pr = ((code >> 32) & 15) - 1
        recs.append((w, s, (code >> 24) & 0xff, code & 0xffffff, pr, (a - t0) * 0.01, (b - t0) * 0.01 if b else None, (c - t0) * 0.01,
                     (code >> 36) & 0xfffffff))  # last: s_memtime ticks over the main loop
end = max(r[7] for r in recs)
print("workgroups %d, tiles %d (+ %d empty slots), makespan %.1f us" % (NW, sum(1 for r in recs if r[4] >= 0), sum(1 for r in recs if r[4] < 0), end))
busy = {}
for r in recs:
    busy[r[0]] = busy.get(r[0], 0.0) + (r[7] - r[5])
tot = sum(busy.values())
print("sum of tile times %.1f us = %.3f of workgroups x makespan; workgroup busy min / median / max = %.1f / %.1f / %.1f us" % (
    tot, tot / (NW * end), min(busy.values()), sorted(busy.values())[len(busy) // 2], max(busy.values())))
last = sorted(max(r[7] for r in recs if r[0] == w) for w in busy)
print("workgroup finish times: 10%% %.1f  50%% %.1f  90%% %.1f  max %.1f us" % (last[len(last) // 10], last[len(last) // 2], last[9 * len(last) // 10], last[-1]))
for pi, oi in enumerate(order):
    rs = [r for r in recs if r[4] == pi]
    if not rs:
        continue
    dur = sorted(r[7] - r[5] for r in rs)
    loop = sorted(r[6] - r[5] for r in rs)
    epi = sorted(r[7] - r[6] for r in rs)
    K = probs[oi][0].shape[0]
    fl = 2.0 * 256 * 256 * K
    print("%-10s K=%6d tiles %4d  tile time min %.1f med %.1f max %.1f us (%.2f TF/s per CU at the median)  loop med %.1f  epilogue med %.1f  start of first / last tile %.1f / %.1f us" % (
        names[oi][0], K, len(rs), dur[0], dur[len(dur) // 2], dur[-1], fl / dur[len(dur) // 2] / 1e6, loop[len(loop) // 2], epi[len(epi) // 2],
        min(r[5] for r in rs), max(r[5] for r in rs)))
    mhz = sorted(r[8] / (r[6] - r[5]) for r in rs if r[6] and r[6] > r[5])
    print("           s_memtime ticks per us of the main loop (the clock the shader actually ran at, MHz): min %.0f  median %.0f  max %.0f" % (
        mhz[0], mhz[len(mhz) // 2], mhz[-1]))
    # groups: consecutive runs of 6 slots in one queue
    skews = []
    for x in range(8):
        q = sorted((r[3], r[5]) for r in rs if r[2] == x)
        for i in range(0, len(q) - 5, 6):
            st = [v[1] for v in q[i:i + 6]]
            skews.append(max(st) - min(st))
    if skews:
        skews.sort()
        print("           start skew inside a 6-tile group: median %.1f  90%% %.1f  max %.1f us" % (skews[len(skews) // 2], skews[9 * len(skews) // 10], skews[-1]))
gaps = []
for w in busy:
    rs = sorted((r for r in recs if r[0] == w), key=lambda r: r[5])
    for a, b in zip(rs, rs[1:]):
        gaps.append(b[5] - a[7])
gaps.sort()
if gaps:
    print("gap between a tile's end and the next tile's start on a workgroup: median %.2f  max %.2f us" % (gaps[len(gaps) // 2], gaps[-1]))
