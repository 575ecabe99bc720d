#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.
Usage: python3 tools/pmc_summary.py DIR [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict

def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_"
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            if want not in name or "at::" in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[name][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    for name, ctrs in sorted(acc.items()):
        d = list(dur[name].values())
        print(f"{name}  dispatches {len(d)}  avg {sum(d)/len(d):.1f} us (under counters)")
        for c, v in sorted(ctrs.items()):
            print(f"    {c:28s} avg {sum(v)/len(v):16.0f}  max {max(v):16.0f}")

if __name__ == "__main__":
    main()
