"""Where a stream of decoding batches leaves the GPU idle: from a rocprofv3 kernel-trace CSV (all queues together),
the union of the kernel intervals against the wall clock, and every idle gap of at least `min_us` with the kernels
on either side.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/decode_profile.py --mode greedy --batches 6
    python tools/trace_gaps.py DIR/*/*_kernel_trace.csv [min_us=8] [skip_fraction=0.5]

`skip_fraction`: the leading part of the trace (model set-up, eager and capture passes) to ignore."""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
t_first, t_last = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
cut = t_first + skip * (t_last - t_first)
seg = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
wall = (t1 - t0) / 1e3
busy_q = collections.Counter()
for r in seg:
    busy_q[r["Queue_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# union of intervals
ivs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in seg)
union, gaps = 0.0, []
cur_s, cur_e, cur_name = ivs[0]
for s, e, name in ivs[1:]:
    if s > cur_e:
        union += (cur_e - cur_s) / 1e3
        gaps.append(((s - cur_e) / 1e3, cur_name, name, (cur_e - t0) / 1e6))
        cur_s, cur_e, cur_name = s, e, name
    elif e > cur_e:
        cur_e, cur_name = e, name
union += (cur_e - cur_s) / 1e3
print("window %.2f ms, %d launches on queues %s" % (wall / 1e3, len(seg), dict((q, round(b / 1e3, 2)) for q, b in busy_q.items())))
print("some kernel running: %.2f ms (%.1f %%), idle %.2f ms in %d gaps" % (union / 1e3, 100 * union / wall, (wall - union) / 1e3, len(gaps)))
small = [g for g in gaps if g[0] < min_us]
print("gaps below %.0f us: %d, together %.2f ms (mean %.2f us)" % (min_us, len(small), sum(g[0] for g in small) / 1e3,
                                                                  sum(g[0] for g in small) / max(len(small), 1)))
big = [g for g in gaps if g[0] >= min_us]
print("gaps of at least %.0f us: %d, together %.2f ms" % (min_us, len(big), sum(g[0] for g in big) / 1e3))
by_pair = collections.defaultdict(list)
for g in big:
    by_pair[(g[1][:48], g[2][:48])].append(g[0])
for (a, b), v in sorted(by_pair.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("  %3d x %7.1f us (sum %7.2f ms)  after %-48s before %s" % (len(v), sum(v) / len(v), sum(v) / 1e3, a, b))
