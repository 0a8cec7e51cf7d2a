#!/usr/bin/env python3
"""A/B comparison of library builds on one configuration: per-stage HIP-event times of forward+backward, the
whole-step time with stage timing off, and a parity check of every build against the first one (image bit-equal or
close, gradients close).  Each build runs in its own process (the library path is fixed at import time).

    python tools/ab_stage.py --config c4 --steps 40 lib/libngs_rasterizer.so lib/variants/libngs_x.so ...
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    import gc
    import time

    import torch
    sys.path.insert(0, ROOT)
    from neuralgaussiansplatting_amd import synthetic, view_parallel
    from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _lib as _C
    dev = torch.device("cuda:0")
    _C.lib().ngs_debug_depth_order({"auto": -1, "tile": 0, "gaussian": 1}[args.depth_order])
    P, W, H, D, M, ls, ext = synthetic.CONFIGS[args.config]
    if args.log_scale is not None:
        ls = tuple(args.log_scale)
    scene = synthetic.make_scene(P, D, ls, ext, 1234, M).to(dev)
    cam = synthetic.make_view(W, H, index=args.view, count=8, device=dev)
    tanx, tany = synthetic.tan_half_fov(cam)
    bg = torch.zeros(3, device=dev)
    dL = (torch.randn(3, H, W, generator=torch.Generator().manual_seed(1234)) / float(W * H)).to(dev)
    bucket = view_parallel.GradientBucket(P, M, dev)

    def step():
        return view_parallel.forward_backward(scene, cam, bg, lambda color: dL, bucket, tanx, tany)

    for _ in range(args.warmup):
        color, radii, aux = step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / args.steps * 1e3
    _C.profile_enable(True)
    _C.profile_read(reset=True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    st = _C.profile_read(reset=True)
    _C.profile_enable(False)
    color, radii, aux = step()
    torch.cuda.synchronize()
    out = {"lib": os.environ.get("NGS_RASTERIZER_LIB", "default"), "ms_per_step": plain, "R": aux["num_rendered"],
           "stages": {k: tot / cnt for k, (tot, cnt) in st.items() if cnt}}
    if args.dump:
        torch.save({"color": color.cpu(), "grads": {k: v.cpu().clone() for k, v in bucket.shaped().items()}}, args.dump)
    print("ABJSON " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--config", default="c4")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--view", type=int, default=0)
    ap.add_argument("--log-scale", type=float, nargs=2, default=None)
    ap.add_argument("--dump", default=None)
    ap.add_argument("--depth-order", default="auto", choices=("auto", "tile", "gaussian"),
                    help="where the depth order comes from: the library's choice, the per-tile sort, or the sort of "
                         "the Gaussians")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    import torch
    results, ref = [], None
    for i, lib in enumerate(args.libs or ["default"]):
        env = dict(os.environ)
        if lib != "default":
            env["NGS_RASTERIZER_LIB"] = os.path.abspath(lib)
        dump = "/tmp/ab_dump_%d.pt" % i
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--config", args.config, "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--view", str(args.view), "--dump", dump, "--depth-order",
               args.depth_order]
        if args.log_scale is not None:
            cmd += ["--log-scale", str(args.log_scale[0]), str(args.log_scale[1])]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("ABJSON ")]
        if p.returncode != 0 or not line:
            print("FAILED %s rc=%d\n%s\n%s" % (lib, p.returncode, p.stdout[-2000:], p.stderr[-3000:]), flush=True)
            if p.returncode in (-6, -11, 134, 139, 124, 137):
                print("stopping: a GPU step died", flush=True)
                return 1
            continue
        r = json.loads(line[0][7:])
        d = torch.load(dump, weights_only=True)
        if ref is None:
            ref = d
            r["parity"] = "reference build"
        else:
            worst = {"color": float((d["color"] - ref["color"]).abs().max())}
            for k, v in d["grads"].items():
                big = float(ref["grads"][k].abs().max())
                worst[k] = float((v - ref["grads"][k]).abs().max()) / (big if big > 0 else 1.0)
            r["parity"] = {k: "%.1e" % v for k, v in worst.items()}
        results.append(r)
    names = []
    for r in results:
        for k in r["stages"]:
            if k not in names:
                names.append(k)
    print("%-28s" % "stage (ms)" + "".join("%14s" % os.path.basename(r["lib"])[-13:] for r in results))
    for k in names:
        print("%-28s" % k + "".join("%14.4f" % r["stages"].get(k, float("nan")) for r in results))
    print("%-28s" % "sum of stages" + "".join("%14.4f" % sum(r["stages"].values()) for r in results))
    print("%-28s" % "step, timing off" + "".join("%14.4f" % r["ms_per_step"] for r in results))
    for r in results:
        print(os.path.basename(r["lib"]), "R=%d" % r["R"], "parity vs first:", r["parity"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
