#!/usr/bin/env python3
"""Vector-ALU and LDS load of each kernel from one rocprofv3 --pmc pass over
SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE.

    python tools/pmc_valu_summary.py gpurun_out/f_pmc_valu [profiles/valu_issue.json c4] > profiles/r01_c4_valu.md

With the two optional arguments the issue fractions are also stored in that JSON file under the given configuration
name (bench.py quotes the dominant kernel's figure next to its HBM roofline).

Units on gfx950: SQ_INSTS_* count wave-instructions; SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over waves;
SQ_LDS_IDX_ACTIVE counts the cycles the CUs' LDS arrays are busy, summed over the 256 CUs; GRBM_GUI_ACTIVE is the sum of
the 8 XCDs' busy cycles.
    valu_issue_fraction = SQ_INSTS_VALU * 2 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
is the fraction of a peak of one wave64 FP32 instruction per 2 cycles per SIMD.  That peak is nominal: a loop of
independent v_fma_f32 at full occupancy sustains one per 2.7 cycles (tools/probes/valu_rate_probe.hip), transcendental,
DPP, lane-swap and packed instructions cost 1.5 to 3 times that, and no kernel of this library, whatever its mix and
occupancy, has been seen above one vector instruction per 3.6 cycles per SIMD (fraction 0.55).
    lds_busy_fraction = SQ_LDS_IDX_ACTIVE / (256 CUs * GRBM_GUI_ACTIVE / 8)
"""
import collections
import csv
import glob
import json
import os
import sys

SIMDS = 1024


def main():
    d = sys.argv[1]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            k = r["Kernel_Name"].replace("void ", "")
            k = k[:k.index("(")] if "(" in k else k
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
    print("| kernel | launches | VALU wave-instr / launch | SALU | LDS | cycles (per XCD) | VALU issue fraction | "
          "LDS busy fraction | waves resident per SIMD | wave time waiting on memory/LDS counters | "
          "wave time stalled at issue |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    fractions = {}
    for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", 0)):
        c, n = tot[k], len(launches[k])
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8 / n
        if cyc <= 0 or c.get("SQ_INSTS_VALU", 0) <= 0 or k.startswith("__amd") or k.startswith("at::"):
            continue
        valu = c["SQ_INSTS_VALU"] / n
        wave_q = c.get("SQ_WAVE_CYCLES", 0) / n
        fractions[k.split("<")[0].replace("_kernel", "")] = round(valu * 2 / (SIMDS * cyc), 4)
        print("| `%s` | %d | %.3g | %.3g | %.3g | %.3g | %.2f | %.2f | %.1f | %.0f %% | %.0f %% |" % (
            k, n, valu, c.get("SQ_INSTS_SALU", 0) / n, c.get("SQ_INSTS_LDS", 0) / n, cyc,
            valu * 2 / (SIMDS * cyc), c.get("SQ_LDS_IDX_ACTIVE", 0) / n / (256 * cyc), wave_q * 4 / (SIMDS * cyc),
            100 * c.get("SQ_WAIT_ANY", 0) / n / wave_q if wave_q else 0,
            100 * c.get("SQ_WAIT_INST_ANY", 0) / n / wave_q if wave_q else 0))
    if len(sys.argv) > 3:
        from _source import merged_table, source_record
        table = merged_table(sys.argv[2], source_record(sys.argv[4] if len(sys.argv) > 4 else "rocprofv3 --pmc SQ_* pass"))
        table[sys.argv[3]] = fractions
        with open(sys.argv[2], "w") as f:
            json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()
