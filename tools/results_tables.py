#!/usr/bin/env python3
"""Prints the markdown tables of profiles/README.md from the bench lines and the traffic file kept next to it.

    python tools/results_tables.py r01 > /tmp/tables.md
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = [("preprocess_fwd", "preprocess_fwd"), ("scan_block_sums", "scan_block_sums"),
          ("duplicate_with_keys", "duplicate_with_keys"), ("radix_sort_tile_bits", "radix_sort_tile_bits (2 passes)"),
          ("tile_ranges", "tile_ranges"), ("depth_sort_gaussians", "depth_sort_gaussians"), ("segment_sort_depth", "segment_sort_depth"), ("render_fwd", "render_fwd"),
          ("zero_accumulator", "zero_accumulator"), ("render_bwd", "render_bwd"), ("preprocess_bwd", "preprocess_bwd")]
CONFIGS = {"c1": "C1 256 / 64×64 / SH0", "c2": "C2 100 k / 800×800 / SH0", "c3": "C3 300 k / 800×800 / SH3, train loop",
           "c4": "C4 3 M / 1080p / SH3", "c5": "C5 6 M / 1080p / SH3, one view"}


def load(tag, c):
    with open(os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (tag, c))) as f:
        return json.load(f)


def spaced(n):
    return "{:,}".format(n).replace(",", " ")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    d = load(tag, "c4")
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
        traffic = json.load(f)["c4"]
    print("| stage | ms | algorithmic GB | GB/s | PMC HBM bytes (read×2 + write) |")
    print("|---|---|---|---|---|")
    for key, label in STAGES:
        if key not in d["stages"]:  # a stage of the depth-order route this run did not take
            continue
        v = d["stages"][key]
        t = traffic.get(key)
        tiny = v["algorithmic_GB"] <= 0.001
        print("| {} | {:.3f} | {} | {} | {} |".format(
            label, v["ms"], "~0" if tiny else "{:.3f}".format(v["algorithmic_GB"]),
            "–" if tiny else "{:d}".format(round(v["GB_per_s"])),
            "{:.2f} GB".format(t / 1e9) if t and t > 1e6 else "–"))
    print("| **step (fwd+bwd)** | **{:.3f}** | | | **{:d} Mpixels/s, {:d} it/s** |".format(
        d["ms_per_step"], round(d["value"]), round(d["iters_per_s"])))
    print()
    t = d["train"]
    print("train: {:.3f} ms per iteration = {:.1f} it/s; stages: {}".format(
        t["ms_per_iter"], t["iters_per_s"],
        ", ".join("{} {:.3f}".format(k, v["ms_per_iter"]) for k, v in t["stages"].items())))
    print("idxmap: {}".format({k: v["ms"] for k, v in d["idxmap"]["stages"].items()}))
    print("scene_init: {}".format(d["scene_init"]))
    print("cpu_baseline: {}".format(d["cpu_baseline"]))
    print("roofline: {}".format(d["roofline"]))
    print()
    print("| Config | Device | cores / GPUs | fwd+bwd ms | fwd+bwd Mpix/s | rasterizer it/s | train it/s | R | "
          "dominant kernel GB/s (frac of 8 TB/s) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for c, label in CONFIGS.items():
        b = load(tag, c)
        cpu = b["cpu_baseline"]
        W, H, R = b["config"]["width"], b["config"]["height"], b["config"]["instances_R"]
        print("| {} | CPU checker | {} | {:.0f} | {:.2f} | – | – | {} | – |".format(
            label, cpu["cores"], W * H / 1e6 / cpu["value"] * 1e3, cpu["value"], spaced(R)))
        dom = b["roofline"]
        domtxt = "launch-latency bound" if c == "c1" else "{} {:d} ({:.2f})".format(dom["kernel"], round(dom["achieved"]),
                                                                                  dom["frac"])
        dev = "1× MI355X (per GPU of the 8-view batch)" if c == "c5" else "1× MI355X"
        print("| {} | {} | 1 | {:.3f} | {:.1f} | {:d} | {:d} | {} | {} |".format(
            label.split(" ")[0], dev, b["ms_per_step"], b["value"], round(b["iters_per_s"]),
            round(b["train"]["iters_per_s"]), spaced(R), domtxt))


if __name__ == "__main__":
    main()
