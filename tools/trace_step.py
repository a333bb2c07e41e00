"""Per-step kernel breakdown from a rocprofv3 kernel-trace CSV:
   rocprofv3 --kernel-trace --output-format csv -d DIR -o k -- python bench.py ...
   python tools/trace_step.py DIR/k_kernel_trace.csv [step_index]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "opt_adam" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seg = rows[adam[k] + 1:adam[k + 1] + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
# union of busy intervals (streams overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
cov, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print("steps %d  launches/step %d  wall %.2f ms  sum-of-kernels %.2f ms  covered %.2f ms (idle %.2f)" % (
    len(adam), len(seg), (t1 - t0) / 1e6, busy / 1e6, cov / 1e6, (t1 - t0 - cov) / 1e6))
agg = collections.OrderedDict()
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    key = (r["Kernel_Name"][:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    c = agg.setdefault(key, [0, 0])
    c[0] += 1
    c[1] += d
for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-44s grid %8s %5s %3s wg %5s  n %4d  tot_us %9.1f  avg_us %8.1f" % (key + (v[0], v[1] / 1e3, v[1] / 1e3 / v[0])))
