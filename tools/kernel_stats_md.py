#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into a short markdown table.

    python tools/kernel_stats_md.py gpurun_out/r01_prof/r01_kernel_stats.csv "<command that was profiled>" > profiles/x.md
"""
import csv
import sys


def main():
    path, cmd = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(path, newline="")))
    print("rocprofv3 --kernel-trace --stats -- %s\n" % cmd)
    print("| kernel | calls | avg us | min us | max us | % of GPU time |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        name = r["Name"].replace("void ", "")
        for cut in ("(",):
            if cut in name:
                name = name[:name.index(cut)]
        if len(name) > 60:
            name = name[:57] + "..."
        print("| `%s` | %s | %.1f | %.1f | %.1f | %.2f |" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                       float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                       float(r["Percentage"])))


if __name__ == "__main__":
    main()
