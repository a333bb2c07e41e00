"""Summarise rocprofv3 counter / kernel-trace CSVs for the kernels whose names contain given substrings.

    python tools/pmc_summary.py --out profiles/attn_step_pmc.json --match attn_partial attn_combine \
        --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write [--trace gpurun_out/attn_trace]

Each directory is the ``-d`` output of ONE rocprofv3 pass (FETCH_SIZE and WRITE_SIZE need separate passes:
MI355X_MICROARCH.md, rocprofv3 PMC slots).  HBM bytes per launch follow that guide's HBM section: the
counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B / lane) coalesced
streaming read, so reads = 2 x FETCH_SIZE x 1024; WRITE_SIZE is taken as is (uncalibrated).  The first
``--skip`` launches of every kernel (warm-up) are dropped.
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

csv.field_size_limit(sys.maxsize)


def rows(directory, suffix):
    for path in glob.glob(os.path.join(directory, "**", "*" + suffix), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                yield row


def short(name):
    """Kernel name without its argument list / return type."""
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def counter_avgs(directory, counter, match, skip):
    per = defaultdict(list)
    for row in rows(directory, "counter_collection.csv"):
        if row.get("Counter_Name") != counter:
            continue
        name = short(row["Kernel_Name"])
        if any(m in name for m in match):
            per[name].append(float(row["Counter_Value"]))
    return {k: (sum(v[skip:]) / max(1, len(v[skip:])), len(v[skip:])) for k, v in per.items()}


def trace_avgs(directory, match, skip):
    per = defaultdict(list)
    for row in rows(directory, "kernel_trace.csv"):
        name = short(row["Kernel_Name"])
        if any(m in name for m in match):
            per[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return {k: (sum(v[skip:]) / max(1, len(v[skip:])), len(v[skip:])) for k, v in per.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--match", nargs="+", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--trace")
    ap.add_argument("--skip", type=int, default=3)
    ap.add_argument("--algorithmic-bytes", type=int, default=53528576)
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    out = {"kernels": {}, "correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are KB; on gfx950 "
           "FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; "
           "WRITE_SIZE taken as is (uncalibrated)", "algorithmic_bytes_per_launch": args.algorithmic_bytes,
           "note": args.note}
    fetch = counter_avgs(args.fetch, "FETCH_SIZE", args.match, args.skip) if args.fetch else {}
    write = counter_avgs(args.write, "WRITE_SIZE", args.match, args.skip) if args.write else {}
    trace = trace_avgs(args.trace, args.match, args.skip) if args.trace else {}
    total = 0.0
    total_us = 0.0
    for name in sorted(set(fetch) | set(write) | set(trace)):
        rec = {}
        if name in fetch:
            rec["FETCH_SIZE_raw_KB_per_launch"], rec["launches"] = fetch[name]
            rec["read_bytes_corrected"] = 2.0 * fetch[name][0] * 1024.0
            total += rec["read_bytes_corrected"]
        if name in write:
            rec["WRITE_SIZE_raw_KB_per_launch"] = write[name][0]
            rec["write_bytes"] = write[name][0] * 1024.0
            total += rec["write_bytes"]
        if name in trace:
            rec["avg_us"], rec["trace_launches"] = trace[name]
            total_us += trace[name][0]
        out["kernels"][name] = rec
    out["hbm_bytes_per_launch"] = int(total) if (fetch or write) else None
    if trace:
        out["sum_avg_us"] = total_us
        out["algorithmic_GBps"] = args.algorithmic_bytes / (total_us * 1e-6) / 1e9
        out["frac_of_8TBps"] = out["algorithmic_GBps"] / 8000.0
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
