#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into per-kernel HBM traffic per launch.

    python tools/pmc_summary.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write \\
        --config c4 --out profiles/hbm_traffic.json --raw profiles/r02_c4_pmc_traffic.json

FETCH_SIZE and WRITE_SIZE are collected in separate passes (they do not fit the
TCC counter slots together) and are reported by rocprofv3 in KiB.  On gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so
the read figure is doubled; WRITE_SIZE is exact for 16-byte streaming stores and
for float atomics.  The correction is exact only for wide streaming reads; for
gather-heavy kernels the doubled figure is an upper bound on the read bytes.
"""
import argparse
import collections
import csv
import glob
import json
import os

# stage name used by bench.py -> kernels that make up the stage
STAGE_KERNELS = {
    "preprocess_fwd": ["preprocess_fwd_kernel"],
    "scan_block_sums": ["scan_block_sums_kernel"],
    "duplicate_with_keys": ["duplicate_kernel"],
    "radix_sort_tile_bits": ["sort_hist_kernel", "sort_scan_kernel", "sort_scatter_kernel"],
    "tile_ranges": ["tile_ranges_kernel"],
    "segment_sort_depth": ["segment_sort_bucket_kernel", "segment_split_kernel", "segment_sort_sub_kernel",
                           "segment_sort_long_kernel"],
    "depth_sort_gaussians": ["depth_order_kernel"],  # its radix passes are counted under radix_sort_tile_bits
    "render_fwd": ["render_fwd_kernel"],
    "render_bwd": ["render_bwd_kernel"],
    "preprocess_bwd": ["preprocess_bwd_kernel"],
}


def short(name):
    name = name.replace("void ", "")
    for cut in ("<", "("):
        if cut in name:
            name = name[:name.index(cut)]
    return name.strip()


def per_kernel(directory, counter):
    """-> {kernel: (sum of counter over launches, launches)}"""
    tot = collections.defaultdict(float)
    cnt = collections.Counter()
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return {k: (tot[k], cnt[k]) for k in tot}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--config", default="c4")
    ap.add_argument("--fwd-runs", type=int, default=0, help="forward passes the profiled command ran (default: the "
                    "number of render_fwd_kernel launches in the fetch pass)")
    ap.add_argument("--bwd-runs", type=int, default=0, help="backward passes it ran (default: render_bwd_kernel "
                    "launches)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--raw", required=True)
    a = ap.parse_args()
    rd = per_kernel(a.fetch, "FETCH_SIZE")
    wr = per_kernel(a.write, "WRITE_SIZE")
    a.fwd_runs = a.fwd_runs or rd["render_fwd_kernel"][1]
    a.bwd_runs = a.bwd_runs or rd["render_bwd_kernel"][1]
    raw = {}
    for k in sorted(set(rd) | set(wr)):
        r, rn = rd.get(k, (0.0, 0))
        w, wn = wr.get(k, (0.0, 0))
        raw[k] = {"launches": max(rn, wn), "FETCH_SIZE_KiB_per_launch": r / rn if rn else None,
                  "WRITE_SIZE_KiB_per_launch": w / wn if wn else None}
    stages = {}
    for stage, kernels in STAGE_KERNELS.items():
        # a stage may launch several kernels per pass (sort passes): sum over all launches, divide by the passes
        read = sum(rd.get(k, (0.0, 0))[0] for k in kernels) * 1024.0
        write = sum(wr.get(k, (0.0, 0))[0] for k in kernels) * 1024.0
        runs = a.bwd_runs if stage.endswith("_bwd") else a.fwd_runs
        stages[stage] = {"read_bytes_raw": read / runs, "read_bytes_corrected": 2.0 * read / runs,
                         "write_bytes": write / runs, "hbm_bytes": (2.0 * read + write) / runs}
    with open(a.raw, "w") as f:
        json.dump({"unit": "KiB per launch as printed by rocprofv3", "kernels": raw, "stages_per_step": stages,
                   "forward_runs": a.fwd_runs, "backward_runs": a.bwd_runs}, f, indent=1)
    from _source import merged_table, source_record
    table = merged_table(a.out, source_record(a.raw))
    table[a.config] = {s: int(v["hbm_bytes"]) for s, v in stages.items()}
    with open(a.out, "w") as f:
        json.dump(table, f, indent=1)
    for s, v in stages.items():
        print("%-22s read(raw) %8.1f MB  read(x2) %8.1f MB  write %8.1f MB" %
              (s, v["read_bytes_raw"] / 1e6, v["read_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
