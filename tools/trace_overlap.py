"""How much do the hardware queues of a rocprofv3 kernel trace really run side by side?  For the last `frac` of the
trace: per queue the launches, busy time and the share of that busy time during which another queue was busy too.

    python tools/trace_overlap.py k_kernel_trace.csv [frac=0.4]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
seg = rows[int(len(rows) * (1 - frac)):]
queues = sorted({r["Queue_Id"] for r in seg})
iv = {q: [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg if r["Queue_Id"] == q] for q in queues}


def merged(spans):
    out = []
    for s, e in sorted(spans):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def common(a, b):
    i = j = tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


m = {q: merged(v) for q, v in iv.items()}
wall = max(e for v in iv.values() for _, e in v) - min(s for v in iv.values() for s, _ in v)
print("window %.2f ms, queues %s" % (wall / 1e6, queues))
for q in queues:
    busy = sum(e - s for s, e in m[q])
    others = merged([tuple(x) for p in queues if p != q for x in m[p]])
    names = {}
    for r in seg:
        if r["Queue_Id"] == q:
            k = r["Kernel_Name"][:48]
            names[k] = names.get(k, 0) + 1
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print("queue %s: %d launches, busy %.2f ms, of which %.2f ms beside another queue; mostly %s" % (
        q, len(iv[q]), busy / 1e6, common(m[q], others) / 1e6, ", ".join("%dx %s" % (n, k) for k, n in top)))
