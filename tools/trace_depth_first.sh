# Kernel statistics of the forward pass with the depth order taken from a sort of the Gaussians (c4).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_df
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from neuralgaussiansplatting_amd import synthetic
from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _C, _lib
_lib.lib().ngs_debug_depth_order(1)
dev = torch.device("cuda:0")
P, W, H, D, M, ls, ext = synthetic.CONFIGS["c4"]
sc = synthetic.make_scene(P, D, ls, ext, 1234, M).to(dev)
cam = synthetic.make_view(W, H, index=0, count=8, device=dev)
tanx, tany = synthetic.tan_half_fov(cam)
bg = torch.zeros(3, device=dev); e = torch.empty(0)
for _ in range(20):
    _C.rasterize_gaussians(bg, sc.means3D, e, sc.opacities, sc.scales, sc.rotations, 1.0, e, cam.world_view_transform,
                           cam.full_proj_transform, tanx, tany, H, W, sc.shs, D, cam.camera_center, False, False)
torch.cuda.synchronize()
PY
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) "forward, depth order from the Gaussians" | head -24
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_df/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("render_fwd")]
a, b = idx[-2], idx[-1]
prev = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-46s %8.1f us  idle before %6.1f" % (r["Kernel_Name"][:46], (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
PY
