"""Kernel breakdown of the LAST `frac` of a rocprofv3 kernel-trace CSV (steady state), with the
GPU-idle time between kernels:  python tools/trace_window.py k_kernel_trace.csv [frac=0.5] [top=25]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
seg = rows[int(len(rows) * (1 - frac)):]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
cov, cs, ce = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > ce:
        cov += ce - cs
        gaps.append(s - ce)
        cs, ce = s, e
    else:
        ce = max(ce, e)
cov += ce - cs
busy = sum(e - s for s, e in iv)
print("launches %d  wall %.2f ms  sum-of-kernels %.2f ms  covered %.2f ms  idle %.2f ms" % (
    len(seg), (t1 - t0) / 1e6, busy / 1e6, cov / 1e6, (t1 - t0 - cov) / 1e6))
big = [g for g in gaps if g > 8000]
print("gaps: %d of <= 8 us (total %.2f ms, avg %.2f us); %d longer (host round trips: total %.2f ms, avg %.1f us)" % (
    len(gaps) - len(big), (sum(gaps) - sum(big)) / 1e6, (sum(gaps) - sum(big)) / 1e3 / max(1, len(gaps) - len(big)),
    len(big), sum(big) / 1e6, sum(big) / 1e3 / max(1, len(big))))
# the longest gaps and the kernels around them
order = sorted(seg, key=lambda r: int(r["Start_Timestamp"]))
found, end_so_far, last = [], int(order[0]["End_Timestamp"]), order[0]
for r in order[1:]:
    s0 = int(r["Start_Timestamp"])
    if s0 > end_so_far:
        found.append((s0 - end_so_far, last["Kernel_Name"][:36], r["Kernel_Name"][:36]))
    if int(r["End_Timestamp"]) >= end_so_far:
        end_so_far, last = int(r["End_Timestamp"]), r
kinds = collections.Counter()
for g_ns, a, b in found:
    if g_ns > 8000:
        kinds[(a, b)] += g_ns
for (a, b), tot in kinds.most_common(8):
    n = sum(1 for g_ns, x, y in found if g_ns > 8000 and (x, y) == (a, b))
    print("  long gaps %3d x avg %6.1f us   after %-36s before %s" % (n, tot / 1e3 / n, a, b))
agg = collections.OrderedDict()
for r in seg:
    c = agg.setdefault(r["Kernel_Name"][:60], [0, 0])
    c[0] += 1
    c[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%-60s n %5d  tot_ms %8.2f  avg_us %8.1f  %5.1f%%" % (k, v[0], v[1] / 1e6, v[1] / 1e3 / v[0], 100.0 * v[1] / (t1 - t0)))
