"""Compact per-stream timeline of ONE training step from a rocprofv3 kernel-trace CSV: kernels of at least
`min_us` are listed one per line, runs of shorter ones on the same queue are folded into one line.

    python tools/trace_timeline.py k_kernel_trace.csv [step_index=4] [min_us=40] [from_ms to_ms: list every launch]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "opt_adam" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
seg = rows[adam[k] + 1:adam[k + 1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
queues = sorted({r["Queue_Id"] for r in seg})
print("step wall %.2f ms, %d launches, queues %s" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e6, len(seg), queues))
for q in queues:
    print("---- queue", q)
    run = None            # [first start, last end, count, busy]
    for r in [x for x in seg if x["Queue_Id"] == q]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        if (e - s) / 1e3 < min_us:
            if run is None:
                run = [s, e, 0, 0]
            run[1], run[2], run[3] = e, run[2] + 1, run[3] + (e - s)
            continue
        if run is not None:
            print("  %8.3f .. %8.3f ms   %4d short kernels, busy %.0f us" % (run[0] / 1e6, run[1] / 1e6, run[2], run[3] / 1e3))
            run = None
        print("  %8.3f .. %8.3f ms   %7.0f us  %s" % (s / 1e6, e / 1e6, (e - s) / 1e3, r["Kernel_Name"][:70]))
    if run is not None:
        print("  %8.3f .. %8.3f ms   %4d short kernels, busy %.0f us" % (run[0] / 1e6, run[1] / 1e6, run[2], run[3] / 1e3))
if len(sys.argv) > 5:          # every launch of a window [from_ms, to_ms] of the step, all queues
    lo, hi = float(sys.argv[4]) * 1e6, float(sys.argv[5]) * 1e6
    print("---- every launch between %.2f and %.2f ms" % (lo / 1e6, hi / 1e6))
    for r in seg:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        if e >= lo and s <= hi:
            print("  q%s %8.3f .. %8.3f ms  %6.1f us  grid %s wg %s lds %s  %s" % (
                r["Queue_Id"], s / 1e6, e / 1e6, (e - s) / 1e3, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"),
                r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:60]))
