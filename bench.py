#!/usr/bin/env python3
"""Benchmark of the rasterizer hot path: forward + backward of one camera view
per GPU on a synthetic scene, through the C ABI of libngs_rasterizer.so.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A step = preprocess -> scan -> key emission -> radix sort -> tile ranges ->
compositing forward, then compositing backward -> per-Gaussian backward, on
inputs already resident in HBM; with N > 1 each rank renders its own view of
the replicated scene and the step ends with the exchange (RCCL) that leaves the
sum over views of the per-Gaussian gradients on every rank (see
neuralgaussiansplatting_amd/view_parallel.py).  Rank 0 prints one JSON line.
The timed loop of exactly --steps steps (barrier + device synchronisation on
both sides, slowest rank) is run --repeats times; ``value`` and ``ms_per_step``
are those of the median loop, ``repeats`` holds the minimum and maximum.

Side loops reported in the same line: whole training iterations (at most 20 from the initial scene unless
--train-steps says otherwise: fitting the random target changes the scene and with it the work per pixel), the
index-map operator and the nearest-neighbour initialisation.  Python's garbage collector is held off during the
timed loops; its full collections take tens of milliseconds and would land in whichever loop is running.
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from neuralgaussiansplatting_amd import build_id, synthetic, view_parallel  # noqa: E402
from neuralgaussiansplatting_amd.debug_state import decode_state  # noqa: E402
from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _lib as _C  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def stage_bytes(P, R, W, H, D, M, vis, views=1):
    """Algorithmic HBM bytes of one launch of each stage: every array a stage must read or write counted once,
    intermediates that fused kernels never move already left out."""
    c = (D + 1) ** 2
    T = ((W + 15) // 16) * ((H + 15) // 16)
    tile_passes = (T.bit_length() + 7) // 8  # bit_length(T) is the reference's getHigherMsb(T)
    vP = vis * P
    return {
        "preprocess_fwd": P * (12 + 8) + vP * (32 + 12 * c + 61),
        "scan_block_sums": 8 * ((P + 255) // 256),
        # the tile-level sort moves (tile id, Gaussian index) pairs of 8 bytes; the depth joins in the per-tile sort
        # tile count; rectangle, offset share and depth; key + list word; the tile ranges zeroed on the side
        "duplicate_with_keys": 4 * P + 16 * vP + 8 * R + 8 * T,
        "radix_sort_tile_bits": tile_passes * 16 * R,
        "tile_ranges": 4 * R + 8 * T,
        # list words and keys in (both carry the depth rank, so no depth is gathered), list out
        "segment_sort_depth": 4 * R + 4 * R + 4 * R + 8 * T,
        # four 8-bit passes over P (depth bits, index) pairs (histogram read + pair read + pair write), then the
        # gather of tile counts and rectangles into depth order
        "depth_sort_gaussians": 4 * (4 + 8 + 8) * P + (4 + 4 + 8 + 12) * P,
        # the compositing kernel also clears the backward pass's accumulator rows (64 bytes per Gaussian)
        "render_fwd": 40 * R + 8 * T + 20 * W * H + 64 * P,
        "zero_accumulator": 4,  # a flag read: the rows are clear already
        "render_bwd": 112 * R + 8 * T + 20 * W * H,
        "preprocess_bwd": 4 * P + vP * (105 + 12 * c) + P * (92 + 12 * M),
        # training-step companions: per iteration (the Adam stages are several launches)
        "sh_gradient_from_views": 25 * P + 12 * P + views * 12 * P + 12 * M * P,
        "l1_ssim_loss": 11 * 12 * W * H,
        "adam_identity_groups": 28 * (3 * P + 3 * M * P),
        # per-Gaussian backward with the SH coefficients' Adam step fused in (single-rank training iterations):
        # coefficients and both moments read and written once, no gradient rows
        "preprocess_bwd_sh_adam": 4 * P + vP * 105 + 92 * P + 72 * M * P,
        "adam_positions_only": 28 * 3 * P,
        # the same kernel stepping all five parameter groups (raw parameters in, raw parameters out): per Gaussian the
        # 11 geometry floats and 48 SH floats with both moments read and written once, no gradient at all; per
        # rendered Gaussian its accumulator row, clamp flags and the three densification statistics
        "preprocess_bwd_train": 4 * P + vP * (37 + 24) + 24 * 11 * P + 72 * M * P,
        "adam_activated_groups": 32 * 8 * P + 40 * P,
    }


def cpu_baseline(scene, cam, W, H, D, tanx, tany, dL):
    """The CPU checker (a port of the reference's algorithm) timed on this box's cores: one forward+backward
    of the same workload."""
    from oracle import oracle as orc
    orc.build()
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    orc.set_threads(cores)
    a = lambda t: t.detach().cpu().numpy()
    t0 = time.time()
    st = orc.forward(a(scene.means3D), a(scene.opacities), a(cam.world_view_transform), a(cam.full_proj_transform),
                     a(cam.camera_center), np.zeros(3, np.float32), W, H, tanx, tany, sh_degree=D, shs=a(scene.shs),
                     scales=a(scene.scales), rotations=a(scene.rotations))
    t1 = time.time()
    orc.backward(st, a(dL), a(scene.means3D), a(cam.world_view_transform), a(cam.full_proj_transform),
                 a(cam.camera_center), np.zeros(3, np.float32), tanx, tany, shs=a(scene.shs), scales=a(scene.scales),
                 rotations=a(scene.rotations))
    t2 = time.time()
    return {"value": W * H / (t2 - t0) / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "1 forward+backward of the same view and scene (fwd %.2f s, bwd %.2f s, R=%d); gcc -O3, "
                      "OpenMP over Gaussians / tiles, backward sums in per-thread copies (no atomics)"
                      % (t1 - t0, t2 - t1, st["num_rendered"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c4", choices=sorted(synthetic.CONFIGS) + sorted(synthetic.CLUSTERED_CONFIGS),
                    help="c1..c5: uniform clouds of small splats; c4s: c4's counts with the shape of a trained scene "
                         "(clustered positions, log-normal sizes, bimodal opacities)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed loop of --steps steps is run this many times; the line reports the median loop "
                         "(value, ms_per_step) and the minimum / maximum beside it")
    ap.add_argument("--exchange", default="factored", choices=("factored", "dense"),
                    help="gradient exchange at N > 1: all-gather of masked colour gradients + all-reduce of the other "
                         "11 floats per Gaussian (default), or one dense all-reduce of all 59")
    ap.add_argument("--chunks", type=int, default=None,
                    help="runs of Gaussians the per-Gaussian stage of the backward pass is queued in at N > 1, each "
                         "run's gradients leaving in a collective of their own while the next is computed (default 4)")
    ap.add_argument("--views-per-gpu", type=int, default=1,
                    help="views each GPU renders per step (the step's gradient is the sum over all GPUs' views): with "
                         "more than one, a view's collectives travel while the next view is rendered and only the "
                         "last view's exchange is exposed; a step then covers that many images per GPU")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo lets several ranks share one GPU to rehearse the N > 1 path on a one-GPU box")
    ap.add_argument("--unfused-train", action="store_true",
                    help="training iterations with separate optimiser kernels (activated copies, gradients through "
                         "memory) instead of the optimiser step inside the backward pass")
    ap.add_argument("--train-steps", type=int, default=None,
                    help="full training iterations (render + L1/D-SSIM loss + backward + exchange + Adam) timed in a "
                         "second loop for the train iters/s figure; default: min(--steps, 20), 0 to skip")
    ap.add_argument("--no-idxmap", action="store_true",
                    help="skip the side measurements: the index-map operator of the reference's second rasterizer (forward+backward) and the nearest-neighbour scene initialisation")
    ap.add_argument("--log-scale-offset", type=float, default=0.0,
                    help="added to the configuration's log-scale range: larger splats, more instances per Gaussian "
                         "(the configurations give about 4; +0.6 about 8, +1.1 about 16 at c4); reported in config")
    ap.add_argument("--stage-steps", type=int, default=None,
                    help="steps of the separate loop that times the stages with HIP events (default min(--steps, 30)); "
                         "the headline loop runs without them")
    ap.add_argument("--depth-order", default="auto", choices=("auto", "per_tile_sort", "gaussians_first"),
                    help="where the depth order of the tile lists comes from: chosen by the library from what either "
                         "has cost (default), or forced (counter and trace runs, whose few frames would otherwise "
                         "mix the two while the library is still trying both)")
    ap.add_argument("--verbose", action="store_true", help="progress lines on stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the rasterizer has no CPU path)")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    # same seed everywhere: replicated parameters
    if args.config in synthetic.CLUSTERED_CONFIGS:
        P, W, H, D, M, mean, ext = synthetic.CLUSTERED_CONFIGS[args.config]
        ls = (mean + args.log_scale_offset, 1.0)  # mean and sigma of the log-size
        scene_cpu = synthetic.make_clustered_scene(P, D, ls[0], ls[1], ext, 1234, M)
    else:
        P, W, H, D, M, ls, ext = synthetic.CONFIGS[args.config]
        ls = (ls[0] + args.log_scale_offset, ls[1] + args.log_scale_offset)
        scene_cpu = synthetic.make_scene(P, D, ls, ext, 1234, M)
    scene = scene_cpu.to(dev)
    V = max(args.views_per_gpu, 1)
    # orbit views: rank r renders views r*V .. r*V+V-1 of the orbit (view 0 of rank 0 is the single-GPU workload)
    cams = [synthetic.make_view(W, H, index=rank * V + k, count=8 * V if V > 1 else 8, device=dev) for k in range(V)]
    cam = cams[0]
    tanx, tany = synthetic.tan_half_fov(cam)
    bg = torch.zeros(3, device=dev)
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    dL = (torch.randn(3, H, W, generator=gen) / float(W * H)).to(dev)
    def make_exchange(form, chunks=None):
        if (world > 1 or V > 1) and form == "factored":
            return view_parallel.FactoredGradientExchange(P, M, dev, views_per_rank=V, chunks=chunks)
        return view_parallel.GradientBucket(P, M, dev, views_per_rank=V, chunks=chunks)

    bucket = make_exchange(args.exchange, args.chunks)
    exchanged_floats = bucket.bytes_sent_per_rank() // 4

    def step(b=None):
        b = bucket if b is None else b
        for k in range(V):
            out = view_parallel.forward_backward(scene, cams[k], bg, lambda color: dL, b, tanx, tany, slot=k)
            b.post_view(k, scene, cams[k], out[2])
        b.finish(scene)
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    class quiet_host:
        """Keeps the Python garbage collector out of a timed loop: a full collection walks the whole heap of the
        imported frameworks (tens of milliseconds) whenever the allocation counters say so, which would be charged
        to whichever loop happens to be running."""

        def __enter__(self):
            gc.collect()
            gc.disable()

        def __exit__(self, *exc):
            gc.enable()

    def note(msg):
        if args.verbose:
            print("[bench rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)

    if args.depth_order != "auto":
        _C.lib().ngs_debug_depth_order({"per_tile_sort": 0, "gaussians_first": 1}[args.depth_order])
    note("scene and view resident")
    for i in range(max(args.warmup, 1)):
        color, radii, aux = step()
        torch.cuda.synchronize(dev)
        note("warm-up step %d done" % i)
    # workload statistics of this rank's view (cost is linear in them)
    R = aux["num_rendered"]
    vis = float((radii > 0).float().mean())
    # the headline loop runs without per-stage events; the stages are timed in a second, shorter loop
    _C.profile_enable(False)
    loops = []
    for _ in range(max(args.repeats, 1)):
        with quiet_host():
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            loops.append(time.perf_counter() - t0)
    if world > 1:  # per loop, the slowest rank
        t = torch.tensor(loops, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        loops = [float(v) for v in t.tolist()]
    elapsed = sorted(loops)[len(loops) // 2]
    note("timed steps done")
    # which of the two sources of the depth order the library took for this workload, and what the other one costs
    depth_order = None
    if world == 1 and V == 1 and not args.no_stage_timing and args.depth_order == "auto":
        L = _C.lib()
        L.ngs_debug_depth_order_taken.restype = ctypes.c_int
        taken = int(L.ngs_debug_depth_order_taken())
        depth_order = {"taken": {0: "per-tile sort of the instances", 1: "sort of the Gaussians first"}.get(taken),
                       "how": "by the cost of either on this process's earlier frames, clocked on the device"}
        for choice, key in ((0, "ms_per_step_per_tile_sort"), (1, "ms_per_step_gaussians_first")):
            L.ngs_debug_depth_order(choice)
            n = max(min(args.steps, 20), 1)
            for _ in range(2):
                step()
            with quiet_host():
                fence()
                t0 = time.perf_counter()
                for _ in range(n):
                    step()
                fence()
                depth_order[key] = round((time.perf_counter() - t0) / n * 1e3, 4)
        L.ngs_debug_depth_order(-1)
        for _ in range(2):  # the capacity estimate and the caches back to the automatic choice
            step()
        torch.cuda.synchronize(dev)
        # what the choice went by: the binning stages as the device clocked them on the last frame with either source
        ua, ub = ctypes.c_double(), ctypes.c_double()
        L.ngs_debug_stage_clock.restype = ctypes.c_int
        if L.ngs_debug_stage_clock(ctypes.byref(ua), ctypes.byref(ub)) == 0:
            depth_order["device_clocked_binning_us"] = {"per_tile_sort": round(ua.value, 1),
                                                        "gaussians_first": round(ub.value, 1)}
    stages = {}
    if not args.no_stage_timing:
        stage_steps = min(args.steps, 30) if args.stage_steps is None else args.stage_steps
        _C.profile_enable(True)
        _C.profile_read(reset=True)
        with quiet_host():
            for _ in range(stage_steps):
                step()
            torch.cuda.synchronize(dev)
        stages = _C.profile_read(reset=True)
        _C.profile_enable(False)
        note("stage timing done")

    # ---- the exchange on its own clock, and the other form of it, so that one run records both ----
    exchange = None
    if world > 1:
        def clocked(b, n):
            for _ in range(2):
                step(b)
            b.enable_clock(True)
            with quiet_host():
                fence()
                t0 = time.perf_counter()
                for _ in range(n):
                    step(b)
                fence()
                t = time.perf_counter() - t0
            total, exposed = b.clock_read() if args.backend == "nccl" else (None, None)
            b.enable_clock(False)
            vals = torch.tensor([t, -1.0 if total is None else total, -1.0 if exposed is None else exposed],
                                dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(vals, op=dist.ReduceOp.MAX)
            t, total, exposed = (float(v) for v in vals.tolist())
            r4 = lambda v: None if v < 0 else round(v, 4)
            return {"ms_per_step": round(t / n * 1e3, 4), "exchange_ms_total": r4(total),
                    "exchange_ms_exposed": r4(exposed), "runs_of_gaussians": b.K,
                    "floats_per_rank_and_step": b.bytes_sent_per_rank() // 4}

        n_ex = max(min(args.steps, 20), 1)
        exchange = {"headline": dict(clocked(bucket, n_ex), form=args.exchange),
                    "what": "exchange_ms_total: from the moment the step's first collective is handed its data until "
                            "the last one has been waited for; exchange_ms_exposed: the part of that after this "
                            "rank's own kernels of the step have run (HIP events on the compute stream, slowest rank, "
                            "%d steps)" % n_ex}
        other_form = "dense" if args.exchange == "factored" else "factored"
        variants = [(other_form, args.chunks)]
        if bucket.K > 1:
            variants.append((args.exchange, 1))  # the headline form sent in one piece
        for form, chunks in variants:
            b2 = make_exchange(form, chunks)
            exchange["%s%s" % (form, "" if chunks is None else "_unchunked")] = dict(clocked(b2, n_ex), form=form)
            del b2
        note("exchange variants done")

    # ---- second loop: whole training iterations ----
    # The training loop fits a random target, which changes the scene (opacities fall, every pixel composites more
    # splats): a short loop from the initial scene keeps the workload the one the first loop measured.
    train_steps = min(args.steps, 20) if args.train_steps is None else args.train_steps
    train_warmup = max(min(args.warmup, 3), 1)
    train = None
    if train_steps > 0:
        from neuralgaussiansplatting_amd import train_step
        trainer = train_step.GaussianTrainer(scene, dev, factored_exchange=args.exchange == "factored",
                                             fused=not args.unfused_train, views_per_rank=V)
        gt = torch.rand(3, H, W, generator=torch.Generator(device="cpu").manual_seed(4321 + rank)).to(dev)
        train_args = (cams, [gt] * V) if V > 1 else (cam, gt)
        for _ in range(train_warmup):
            trainer.step(*train_args, bg, tanx, tany)
        _C.profile_enable(not args.no_stage_timing)
        _C.profile_read(reset=True)
        with quiet_host():
            fence()
            t0 = time.perf_counter()
            for _ in range(train_steps):
                trainer.step(*train_args, bg, tanx, tany)
            fence()
            t_train = time.perf_counter() - t0
        train_stages = _C.profile_read(reset=True)
        _C.profile_enable(False)
        note("training iterations done")
        in_sync = None
        if world > 1:
            cdev = dev if args.backend == "nccl" else "cpu"
            t = torch.tensor([t_train], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_train = float(t.item())
            # the replicas must have stayed bit-identical: every rank applied the same all-view gradients
            sums = torch.stack([p.double().sum() for p in (trainer.xyz, trainer.features, trainer.opacity_raw,
                                                           trainer.scaling_raw, trainer.rotation_raw)]).to(cdev)
            hi, lo = sums.clone(), sums.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            in_sync = bool(torch.equal(hi, lo))
        train = (t_train, train_stages, in_sync, trainer.fused)

    # ---- third loop: the index-map operator (nearest centre per pixel) on the same scene and view ----
    idxmap = None
    if not args.no_idxmap and rank == 0:
        from neuralgaussiansplatting_amd.rasterizer2 import _C as r2c
        fvec = torch.randn(P, r2c.NUM_FEATURES, generator=torch.Generator(device="cpu").manual_seed(99)).to(dev)
        gfeat = torch.randn(H, W, r2c.NUM_FEATURES, generator=torch.Generator(device="cpu").manual_seed(98)).to(dev)
        e0 = torch.empty(0, device=dev)

        def idx_step():
            o = r2c.get_idxmap2(e0, scene.means3D, e0, e0, e0, cam.world_view_transform, cam.full_proj_transform,
                                tanx, tany, H, W, fvec, e0, cam.camera_center, False, count_entries=False)
            r2c.get_idxmap_backward2(scene.means3D, H, W, cam.camera_center, o[1], gfeat, False,
                                     viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform)
            return o

        for _ in range(max(args.warmup, 1)):
            o = idx_step()
        _C.profile_enable(not args.no_stage_timing)
        _C.profile_read(reset=True)
        with quiet_host():
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                idx_step()
            torch.cuda.synchronize(dev)
            t_idx = time.perf_counter() - t0
        st = _C.profile_read(reset=True)
        _C.profile_enable(False)
        hit = int((o[1] >= 0).sum())
        nb = {"idxmap_forward": 12 * P + 16 * W * H + 276 * W * H + 168 * hit,
              "idxmap_backward": 256 * P + 164 * W * H + 312 * hit}
        idxmap = {"ms_per_step": round(t_idx / args.steps * 1e3, 4),
                  "Mpixels_per_s": round(W * H * args.steps / t_idx / 1e6, 1), "pixels_hit": hit,
                  "what": "get_idxmap2 + get_idxmap_backward2, %d Gaussians with 64 features, %dx%d" % (P, W, H),
                  "stages": {k: {"ms": round(tot / cnt, 4), "algorithmic_GB": round(nb[k] / 1e9, 4),
                                 "GB_per_s": round(nb[k] / (tot / cnt * 1e-3) / 1e9, 1)}
                             for k, (tot, cnt) in st.items() if cnt and k in nb}}
    # ---- scene initialisation: mean squared distance to the three nearest neighbours of the scene's centres ----
    scene_init = None
    if not args.no_idxmap and rank == 0:
        from neuralgaussiansplatting_amd.simple_knn._C import distCUDA2
        distCUDA2(scene.means3D)
        torch.cuda.synchronize(dev)
        reps = max(1, min(args.steps, 5))
        t0 = time.perf_counter()
        for _ in range(reps):
            distCUDA2(scene.means3D)
        torch.cuda.synchronize(dev)
        t_knn = (time.perf_counter() - t0) / reps
        scene_init = {"knn_ms": round(t_knn * 1e3, 3), "Mpoints_per_s": round(P / t_knn / 1e6, 1),
                      "what": "distCUDA2 on the scene's %d centres (once per scene, not per iteration)" % P}
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * V * W * H * args.steps / elapsed / 1e6
        # statistics that need the state buffers: one extra untimed forward
        e = torch.empty(0)
        Rr, _, radii2, geom, binning, img = _C.rasterize_gaussians(
            bg, scene.means3D, e, scene.opacities, scene.scales, scene.rotations, 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, tanx, tany, H, W, scene.shs, D, cam.camera_center, False, False)
        stt = decode_state(_C, P, Rr, W, H, geom, binning, img)
        per_tile = (stt["ranges"][:, 1] - stt["ranges"][:, 0]).astype(np.int64)
        nbytes = stage_bytes(P, R, W, H, D, M, vis, views=world)
        stage_out, dom, dom_ms = {}, None, 0.0
        for name, (tot, cnt) in stages.items():
            if cnt == 0:
                continue
            avg = tot / cnt
            gbs = nbytes[name] / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
            stage_out[name] = {"ms": round(avg, 4), "launches": cnt, "algorithmic_GB": round(nbytes[name] / 1e9, 4),
                               "GB_per_s": round(gbs, 1)}
            if avg > dom_ms:
                dom, dom_ms = name, avg
        roofline = None
        if dom is not None:
            ach = nbytes[dom] / (dom_ms * 1e-3) / 1e9
            # Counter-derived figures come from separate rocprofv3 --pmc passes (tools/collect_profiles.sh), not
            # from this run: they are quoted only if the profiled build had the kernel sources this build has, and
            # the line says which file and commit they come from.
            here = build_id.kernel_sources_sha()

            def profiled(fname):
                path = os.path.join(ROOT, "profiles", fname)
                if not os.path.exists(path) or args.log_scale_offset != 0.0:
                    return None, None
                with open(path) as f:
                    doc = json.load(f)
                src = doc.get("_source", {})
                if src.get("kernel_sources_sha") != here:
                    return None, "stale: profiles/%s was measured on kernel sources %s, this build is %s" % (
                        fname, src.get("kernel_sources_sha", "unrecorded"), here)
                return doc.get(args.config, {}).get(dom), "profiles/%s (%s) at commit %s, kernel sources %s" % (
                    fname, src.get("raw", "rocprofv3 --pmc passes"), src.get("commit", "unrecorded"), here)

            traffic, traffic_source = profiled("hbm_traffic.json")
            # the compositing kernels are limited by vector-ALU issue and LDS, not by HBM: quote the counter-derived
            # issue fraction as well (share of a nominal peak of one wave64 FP32 instruction per 2 cycles per SIMD;
            # no kernel here has been seen above 0.55 of it, see tools/pmc_valu_summary.py)
            valu, valu_source = profiled("valu_issue.json")
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "traffic_source": traffic_source, "avg_launch_ms": round(dom_ms, 4),
                        "algorithmic_bytes": int(nbytes[dom]), "valu_issue_frac": valu, "valu_source": valu_source,
                        "limiter": "vector-ALU issue (valu_issue_frac; the practical ceiling is about 0.55) and LDS, "
                                   "not HBM: the kernel moves less than its algorithmic bytes"
                                   if dom.startswith("render_") else "HBM"}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(scene_cpu, synthetic.make_view(W, H, index=0, count=8), W, H, D, tanx, tany, dL)
        train_out = None
        if train is not None:
            t_train, train_stages, in_sync, all_fused = train
            per_iter = {}
            fused = world == 1 and M == 16
            for name, (tot, cnt) in train_stages.items():
                if cnt:
                    ms = tot / train_steps
                    bkey = {"preprocess_bwd": "preprocess_bwd_train" if all_fused else "preprocess_bwd_sh_adam",
                            "adam_identity_groups": "adam_positions_only"}.get(name, name) if fused else name
                    per_iter[name] = {"ms_per_iter": round(ms, 4), "launches_per_iter": round(cnt / train_steps, 2),
                                      "GB_per_s": round(nbytes[bkey] / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0}
            train_out = {"iters_per_s": round(train_steps / t_train, 2),
                         "views_per_s": round(world * V * train_steps / t_train, 2),
                         "ms_per_iter": round(t_train / train_steps * 1e3, 4), "steps": train_steps,
                         "what": ("lr schedule + render from the raw parameters + fused L1/D-SSIM loss + backward with "
                                  "the activations' backward, Adam of all five parameter groups and the "
                                  "densification statistics inside its per-Gaussian kernel, fixed random target "
                                  "image") if all_fused else
                                 "lr schedule + render + fused L1/D-SSIM loss + backward%s + %sfused Adam with "
                                 "activations and densification statistics, fixed random target image"
                                 % (" with the SH coefficients' Adam step inside" if fused else "",
                                    "gradient exchange + " if world > 1 else ""),
                         "replicas_in_sync": in_sync, "stages": per_iter}
        line = {
            "metric": "fwd+bwd Mpixels/s (rasterizer forward+backward, 1080p / 3M Gaussians per GPU)"
                      if args.config == "c4" else "fwd+bwd Mpixels/s (rasterizer forward+backward)",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "iters_per_s": round(args.steps / elapsed, 2), "views_per_s": round(world * V * args.steps / elapsed, 2),
            "repeats": {"loops": len(loops), "steps_per_loop": args.steps, "reported": "median loop",
                        "ms_per_step_min": round(min(loops) / args.steps * 1e3, 4),
                        "ms_per_step_median": round(elapsed / args.steps * 1e3, 4),
                        "ms_per_step_max": round(max(loops) / args.steps * 1e3, 4)},
            "depth_order": depth_order, "exchange": exchange,
            "config": {"workload": "%s%s: %d Gaussians, %dx%d, SH degree %d (M=%d), %s per GPU and step, "
                                   "fwd+bwd%s" % (args.config,
                                                  "" if args.log_scale_offset == 0.0 else
                                                  " with log-scale offset %+.2f" % args.log_scale_offset,
                                                  P, W, H, D, M,
                                                  "one orbit view" if V == 1 else "%d orbit views" % V,
                                                  " + %s gradient exchange of %d floats per rank and step in %d "
                                                  "collective(s) per view"
                                                  % (args.exchange, exchanged_floats,
                                                     bucket.K + (1 if args.exchange == "factored" else 0))
                                                  if world > 1 else ""),
                       "views_per_gpu": V,
                       "gaussians": P, "width": W, "height": H, "sh_degree": D, "instances_R": int(R),
                       "instances_per_gaussian": round(R / P, 2), "log_scale": [round(v, 3) for v in ls],
                       "kernel_sources_sha": build_id.kernel_sources_sha(),
                       "visible_fraction": round(vis, 4), "instances_per_tile_mean": round(float(per_tile.mean()), 1),
                       "instances_per_tile_max": int(per_tile.max()),
                       "last_contributor_per_pixel_mean": round(float(stt["n_contrib"].mean()), 1),
                       "parallelism": "view-sharded dp%d" % world},
            "roofline": roofline, "cpu_baseline": cpu, "stages": stage_out, "train": train_out, "idxmap": idxmap,
            "scene_init": scene_init,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
