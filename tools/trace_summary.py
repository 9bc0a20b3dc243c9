"""Summarise the LAST training step of a rocprofv3 --kernel-trace CSV (steps are delimited by the AdamW launches).

    python tools/trace_summary.py /tmp/prof/bench_kernel_trace.csv out.json [adamw_launches_per_step]
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    path, out = sys.argv[1], sys.argv[2]
    per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "gemm" in name and "Grid_Size_X" in r:  # tell GEMM shapes apart by their launch geometry
                name = "%s grid=%sx%s wg=%s" % (name[:70], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X"))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adamw" in r[2]]
    if len(adam) < 2 * per_step:
        lo, hi = 0, len(rows)
    else:
        lo, hi = adam[-per_step - 1] + 1, adam[-1] + 1
    sel = rows[lo:hi]
    wall = (sel[-1][1] - sel[0][0]) / 1e6
    agg = defaultdict(lambda: [0.0, 0])
    busy = 0.0
    for s, e, n in sel:
        agg[n][0] += (e - s) / 1e6
        agg[n][1] += 1
        busy += (e - s) / 1e6
    top = sorted(((v[0], v[1], k) for k, v in agg.items()), reverse=True)
    res = {"step_wall_ms": wall, "kernel_busy_ms": busy, "launches": len(sel),
           "kernels": [{"ms": round(t, 3), "calls": c, "avg_us": round(1e3 * t / c, 2), "name": n[:160]} for t, c, n in top[:60]]}
    json.dump(res, open(out, "w"), indent=1)
    print("last step: wall %.1f ms, kernel busy %.1f ms, %d launches" % (wall, busy, len(sel)))
    for t, c, n in top[:40]:
        print("%8.2f ms %6d  %s" % (t, c, n[:120]))


if __name__ == "__main__":
    main()
