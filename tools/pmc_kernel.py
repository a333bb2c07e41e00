"""Average of every collected counter per (kernel, grid size) from the counter_collection.csv files of rocprofv3
--pmc passes:  python tools/pmc_kernel.py DIR [DIR...] --match step_group_medium"""
import argparse
import csv
import glob
import os
import sys
from collections import defaultdict

csv.field_size_limit(sys.maxsize)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--match", nargs="+", required=True)
    args = ap.parse_args()
    acc = defaultdict(lambda: defaultdict(list))
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    if not any(m in name for m in args.match):
                        continue
                    grid = row.get("Grid_Size") or row.get("Grid_Size_X") or "?"
                    acc[(name, grid)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for (name, grid), ctrs in sorted(acc.items()):
        print("{}  grid {}".format(name, grid))
        for c, v in sorted(ctrs.items()):
            print("    {:<28} {:>16.1f}  (n={})".format(c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
