"""Matrix-core utilisation per kernel from ONE rocprofv3 counter pass over a training run:

    cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
        --output-format csv -d /tmp/mfma -- python /root/repo/tools/train_profile.py --steps 6
    python tools/mfma_pmc.py /tmp/mfma profiles/r05_train_mfma_pmc.json

SQ_VALU_MFMA_BUSY_CYCLES adds up the cycles in which a SIMD's matrix pipe was busy, over all 1024 SIMDs
(MI355X_MICROARCH.md, per-instruction constants: 64 cycles per v_mfma_f32_32x32x2_f32, 32 per 16x16x4); GRBM_GUI_ACTIVE
is reported summed over the 8 XCDs, so a dispatch lasted GRBM_GUI_ACTIVE / 8 shader cycles.  MFMA-busy share of a kernel =
MFMA_BUSY / (1024 x GRBM_GUI_ACTIVE / 8).  (Counter collection serialises the dispatches: kernels that overlap in a
real step are measured alone here.)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

csv.field_size_limit(sys.maxsize)


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    src, out = sys.argv[1], sys.argv[2]
    per = defaultdict(lambda: defaultdict(dict))          # kernel -> dispatch id -> counter -> value
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                per[short(row["Kernel_Name"])][row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    kernels = {}
    for name, dispatches in per.items():
        rows = [d for d in dispatches.values() if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE")]
        if not rows:
            continue
        mfma = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in rows)
        active = sum(d["GRBM_GUI_ACTIVE"] for d in rows)
        busy = sum(d.get("SQ_BUSY_CYCLES", 0.0) for d in rows)
        us = dur.get(name, [])
        kernels[name] = {"launches": len(rows), "mfma_busy_share": mfma / (1024.0 * active / 8.0),
                         "avg_us_under_the_counters": (sum(us) / len(us)) if us else None,
                         "total_us_under_the_counters": sum(us) if us else None,
                         "avg_mfma_busy_cycles": mfma / len(rows), "avg_gui_active_per_xcd": active / 8.0 / len(rows),
                         "avg_sq_busy_cycles": busy / len(rows)}
    ranked = sorted(kernels.items(), key=lambda kv: -(kv[1]["total_us_under_the_counters"] or 0.0))
    doc = {"what": __doc__.split("\n\n")[-1].replace("\n", " "), "kernels": dict(ranked)}
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("{:<74} {:>6} {:>9} {:>8}".format("kernel", "calls", "avg us", "MFMA %"))
    for name, k in ranked[:24]:
        print("{:<74} {:>6} {:>9.1f} {:>8.1f}".format(name[:74], k["launches"], k["avg_us_under_the_counters"] or 0.0,
                                                       100.0 * k["mfma_busy_share"]))


if __name__ == "__main__":
    main()
