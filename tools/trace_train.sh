# Kernel trace of the training loop: per-iteration launch sequence with the idle time before each kernel.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_train
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-idxmap --no-stage-timing --train-steps 8 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_train/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 40 launches: name, duration, gap to the previous kernel's end
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((r["Kernel_Name"][:60], (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = max(prev_end or 0, e)
for name, dur, gap in out[-45:]:
    print("%-60s %9.1f us   idle before %8.1f us" % (name, dur, gap))
PY
