#!/usr/bin/env python3
"""The headline benchmark at three splat footprints (about 4, 8 and 16 instances per Gaussian): cost is linear in the
instance count and in the compositing depth, so one operating point says little about scenes with larger splats.

    python tools/footprint_sweep.py profiles/r02_c4_footprint.json [--config c4]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = sys.argv[1]
    config = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "c4"
    rows = []
    for off in (0.0, 0.6, 1.1):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "40", "--warmup", "5",
               "--train-steps", "10", "--no-idxmap", "--no-cpu-baseline", "--log-scale-offset", str(off)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            print(p.stdout[-2000:], p.stderr[-3000:], file=sys.stderr)
            return 1
        d = json.loads(line[-1])
        rows.append({"log_scale_offset": off, "instances_per_gaussian": d["config"]["instances_per_gaussian"],
                     "instances_R": d["config"]["instances_R"],
                     "instances_per_tile_mean": d["config"]["instances_per_tile_mean"],
                     "last_contributor_per_pixel_mean": d["config"]["last_contributor_per_pixel_mean"],
                     "ms_per_step": d["ms_per_step"], "Mpixels_per_s": d["value"], "iters_per_s": d["iters_per_s"],
                     "train_iters_per_s": d["train"]["iters_per_s"] if d.get("train") else None,
                     "depth_order": d.get("depth_order"),
                     "stages_ms": {k: v["ms"] for k, v in d["stages"].items()}})
        print(json.dumps(rows[-1]), flush=True)
    with open(out, "w") as f:
        json.dump({"config": config, "what": "python bench.py --log-scale-offset X (40 steps, 5 warm-up)", "rows": rows}, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
