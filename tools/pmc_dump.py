#!/usr/bin/env python3
"""Per-kernel averages (per launch) of every counter found in one or more rocprofv3 --pmc output directories.

    python tools/pmc_dump.py gpurun_out/pmc_a gpurun_out/pmc_b [kernel-name-substring ...]
"""
import collections
import csv
import glob
import os
import sys


def main():
    dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
    pick = [a for a in sys.argv[1:] if not os.path.isdir(a)]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path, newline="")):
                k = r["Kernel_Name"].replace("void ", "")
                k = k[:k.index("(")] if "(" in k else k
                if pick and not any(p in k for p in pick):
                    continue
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]].add((d, r["Dispatch_Id"]))
    for k in sorted(tot):
        print(k)
        for c in sorted(tot[k]):
            print("    %-32s %.4g" % (c, tot[k][c] / max(1, len(launches[k][c]))))


if __name__ == "__main__":
    main()
