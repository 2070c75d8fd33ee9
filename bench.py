#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json: Mrays/s (primary + secondary) of the PT_RGB
ray loop on the synthetic 100k-triangle scene at 1024x1024 (configs[2]; 256 spp = the
default 8 steps x 32 frames).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch of pixel-samples: `--frames-per-step`
(default 32) consecutive frames of the full 1024^2 film, i.e. 33.5 M paths (one wavefront batch).  With N > 1 the
film is sharded by pixel tiles (linear pixel index, tiles of 4096, round-robin over ranks,
replicated scene + BVH, no collective on the data path) and the tiles are summed into rank
0's film with ONE RCCL reduce at the end -- total work is fixed, so "scaling" is "strong".
A ray = one closet_hit or closet_hit_shadow call of the reference
(integrator/PT_RGB.py:65,104); rays are counted by device counters.

Timing: W untimed warm-up steps, then exactly K steps bracketed by barrier +
torch.cuda.synchronize() + tirt_sync on both sides; MAX over ranks; rank 0 prints one JSON
line.  Inputs (scene, BVH) are resident in HBM before the timed region.

Extra objects:
  roofline      dominant kernel = closest-hit traversal (k_trace).  achieved = ALGORITHMIC
                bytes per launch / mean launch duration, both measured live: bytes from the
                reference-semantics pop counts (32 B x N_box + 36 B x N_leaf + 48 B per ray,
                SURVEY.md 8d) gathered by an untimed exhaustive counting pass over the same
                frames, duration from HIP events on the library's stream around every
                closest-hit launch of an untimed instrumented pass.
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference algorithm: AoS rows,
                exhaustive unordered traversal, per-pixel loop) on all host cores over a
                bounded pixel sample of the same scene/frame ("kind": "port"; the reference's
                own ti.cpu path cannot run: Taichi is not installable here).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per render lane; must be set before the HIP runtime starts

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one HW queue per render lane; before HIP initialises

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames-per-step", type=int, default=32)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--ntri", type=int, default=100000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tile-size", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-target-s", type=float, default=15.0, help="seconds of CPU-oracle work for cpu_baseline")
    ap.add_argument("--save-png", default="")
    ap.add_argument("--emulate-world", type=int, default=0, help="render only rank 0's tiles of an N-rank job (scaling study on one GPU)")
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to tirt_set_option (tuning)")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs one process per GPU: launch with torch.distributed.run" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # self-test knobs (not used by the driver): TIRT_FORCE_DIST=1 exercises the RCCL path with one rank;
    # TIRT_BENCH_ONE_DEVICE=1 + TIRT_BENCH_BACKEND=gloo lets N ranks share cuda:0 on a 1-GPU box, which runs the
    # whole N-rank flow (tile split, merged submission, film reduce, max over ranks) on real hardware
    if os.environ.get("TIRT_BENCH_ONE_DEVICE", "0") == "1":
        local_rank = 0
    backend = os.environ.get("TIRT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("TIRT_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from ti_raytrace_amd import scenes, _native
    from ti_raytrace_amd import distributed as tdist

    W = H = args.size
    total_frames = (args.warmup + args.steps) * args.frames_per_step
    ex = scenes.synthetic(W, H, max(total_frames, 4), ntri=args.ntri, device_id=local_rank, seed=args.seed,
                          tile_rank=rank, tile_count=(args.emulate_world if args.emulate_world > 0 and world == 1 else world),
                          tile_size=args.tile_size)
    t0 = time.time()
    ex.build_scene()
    ctx = ex.scene.ctx
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, float(v))
    ctx.sync()
    build_wall = time.time() - t0
    build_ms = ctx.stats()["ms_build"]
    fps = args.frames_per_step

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(k):
        for _ in range(k):
            ex.integrator.render_frames(fps)
            ex.cam.update_frame(fps)

    run_steps(args.warmup)
    tdist.warmup(ctx, W, H, force=force_dist)
    barrier()
    ctx.stats_reset()
    t_begin = time.perf_counter()
    run_steps(args.steps)
    film = tdist.reduce_film(ctx, W, H, dst=0, force=force_dist)   # one RCCL reduce of the framebuffer (world > 1)
    barrier()
    elapsed = time.perf_counter() - t_begin
    st = ctx.stats()

    # MAX over ranks of the elapsed time, SUM over ranks of the rays
    el_t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    rays_t = torch.tensor([float(st["rays_closest"]), float(st["rays_shadow"]), float(st["paths"])], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
    elapsed = float(el_t.item())
    rays_closest, rays_shadow, paths = [float(x) for x in rays_t.tolist()]
    mrays = (rays_closest + rays_shadow) / elapsed / 1e6

    result = {
        "metric": "Mrays/s (primary+secondary) at 1024\u00b2 100k-tri, 1/2/4/8 GPU + HBM GB/s",
        "value": round(mrays, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "synthetic %d-tri random mesh (scene seed 1234, s=0.03), PT_RGB %dx%d, %d spp timed "
                        "(%d steps x %d frames), max_depth 15, render seed %d" %
                        (args.ntri, W, H, args.steps * fps, args.steps, fps, args.seed),
            "parallelism": "pixel tiles of %d round-robin over %d GPU(s), replicated BVH, one RCCL film reduce" % (args.tile_size, world),
            "traversal": "ordered+t-culled (bit-identical hits to the reference's exhaustive order)",
        },
        "rays": {"closest": int(rays_closest), "shadow": int(rays_shadow), "paths": int(paths),
                 "rays_per_path": round((rays_closest + rays_shadow) / max(paths, 1.0), 3)},
        "lbvh_build_ms": round(build_ms, 3),
        "scene_setup_wall_s": round(build_wall, 3),
    }

    if rank == 0 and args.save_png:
        from ti_raytrace_amd.Example import write_png
        ctx.film_import_device(film.data_ptr()) if world > 1 else None
        ctx.tone_map(0.5)
        write_png(ctx.film_download(W, H, want_hdr=False, want_rgb=True)[1], args.save_png)

    # ---- roofline for the dominant kernel (rank 0's shard, untimed extra passes) -----------------
    if not args.no_roofline:
        probe_frames = fps
        f0 = ex.cam.frame
        ctx.stats_reset()
        ctx.pt_rgb_render(f0, probe_frames, args.seed, 15, 64, _native.TRAVERSE_EXHAUSTIVE | _native.COUNT_NODES)
        ctx.sync()
        c = ctx.stats()
        alg_closest = 32.0 * c["box_closest"] + 36.0 * c["leaf_closest"] + 48.0 * c["rays_closest"]
        alg_shadow = 32.0 * c["box_shadow"] + 36.0 * c["leaf_shadow"] + 48.0 * c["rays_shadow"]
        ctx.stats_reset()
        ctx.pt_rgb_render(f0, probe_frames, args.seed, 15, 64, _native.TRAVERSE_ORDERED | _native.COUNT_NODES)
        ctx.sync()
        co = ctx.stats()
        ctx.set_option("time_kernels", 1)            # per-kernel HIP events; runs the batches on ONE lane (no overlap)
        ctx.stats_reset()
        ctx.pt_rgb_render(f0, probe_frames, args.seed, 15, 64, 0)
        ctx.sync()
        t = ctx.stats()
        ctx.set_option("time_kernels", 0)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and args.size == 1024 and args.ntri == 100000 and fps == 32 and world == 1:
            tj = json.load(open(tpath)).get("k_trace", {})
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), tj.get("source")
        # dominant kernel = k_trace (closest hits of bounce b + NEE shadow rays of bounce b-1 share a launch)
        n_launch = max(t["launches_trace_closest"] + t["launches_trace_shadow"], 1)
        alg_trace = alg_closest + alg_shadow
        avg_ms = (t["ms_trace_closest"] + t["ms_trace_shadow"]) / n_launch
        achieved = (alg_trace / n_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        gather_bytes = (112.0 / 4.0 * (co["box_closest"] + co["box_shadow"]) + 48.0 * (co["leaf_closest"] + co["leaf_shadow"]) +
                        24.0 * (co["rays_closest"] + co["rays_shadow"]))
        result["roofline"] = {
            "bound": "hbm", "kernel": "k_trace<ordered> (closest-hit + NEE shadow rays)",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "hbm_GBps_measured": (round(traffic / (avg_ms * 1e-3) / 1e9, 1) if (traffic and avg_ms > 0) else None),
            "alg_bytes_per_launch": round(alg_trace / n_launch, 1),
            "avg_launch_ms": round(avg_ms, 5), "launches": int(n_launch),
            "alg_bytes_per_closest_ray": round(alg_closest / max(c["rays_closest"], 1), 1),
            "alg_bytes_per_shadow_ray": round(alg_shadow / max(c["rays_shadow"], 1), 1),
            "n_box_per_closest_ray": round(c["box_closest"] / max(c["rays_closest"], 1), 2),
            "n_leaf_per_closest_ray": round(c["leaf_closest"] / max(c["rays_closest"], 1), 2),
            "n_box_ordered_per_closest_ray": round(co["box_closest"] / max(co["rays_closest"], 1), 2),
            "n_leaf_ordered_per_closest_ray": round(co["leaf_closest"] / max(co["rays_closest"], 1), 2),
            "kernel_ms": {"trace_closest": round(t["ms_trace_closest"], 3), "trace_shadow": round(t["ms_trace_shadow"], 3),
                          "shade": round(t["ms_shade"], 3), "render_total": round(t["ms_render"], 3)},
            "wave_diag_ordered": {"node_iters_per_ray": round(co["diag_it_node"] * 64.0 / max(co["rays_closest"] + co["rays_shadow"], 1), 2),
                                  "node_lane_util": round(co["diag_lanes_node"] / max(co["diag_it_node"] * 64.0, 1), 4),
                                  "leaf_iters_per_ray": round(co["diag_it_leaf"] * 64.0 / max(co["rays_closest"] + co["rays_shadow"], 1), 2),
                                  "leaf_lane_util": round(co["diag_lanes_leaf"] / max(co["diag_it_leaf"] * 64.0, 1), 4),
                                  "top_node_visits_per_ray": round(co["diag_it_outer"] / max(co["rays_closest"] + co["rays_shadow"], 1), 2),
                                  "refills_per_wave_ray": round(co["diag_refills"] * 64.0 / max(co["rays_closest"] + co["rays_shadow"], 1), 3)},
            # what actually limits the kernel: 16-byte-per-lane gathers of node / triangle records through the L1
            # (TA/TD) path -- measured ceiling for scattered 64-byte records on MI355X (tools/micro/gather_rate.hip):
            # 14.0 TB/s from an L2-resident array, 7.7 TB/s from an 8 MB one; rocprofv3: TA busy 94 % (profiles/)
            "l1_gather": {"GBps": round(gather_bytes / n_launch / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None,
                          "bytes_per_launch": round(gather_bytes / n_launch, 1),
                          "measured_ceiling_GBps": {"l2_resident_records": 14000.0, "8MB_working_set": 7700.0},
                          "def": "112 B per 4-wide node visit + 48 B per primitive test + 24 B per ray, ordered traversal counts"},
            "whole_job_alg_GBps": round((alg_closest + alg_shadow + 260.0 * c["shaded"] + 24.0 * c["paths"]) /
                                        max(t["ms_render"], 1e-9) / 1e6, 2),
            "note": "the ~11 MB traversal data is L2/Infinity-Cache resident: HBM traffic is far below the algorithmic bytes; the kernel is bound by the L1 gather path (see l1_gather)",
        }

    # ---- CPU baseline: oracle on the host cores, bounded sample (rank 0, N = 1 only) ---------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api
        orc = oracle_api.OracleScene(ex.scene, ex.cam)
        tb = time.perf_counter()
        orc.lbvh_build()
        cpu_build_s = time.perf_counter() - tb
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        # probe 1/64 of one frame, then size the sample for ~15 s of CPU work (whole frames)
        tp = time.perf_counter()
        _, pst = orc.render(W, H, 1, 1, seed=args.seed, tile_rank=0, tile_count=64, tile_size=args.tile_size, nthreads=cores)
        probe_s = max(time.perf_counter() - tp, 1e-3)
        frame_s = probe_s * 64.0
        # (the probe over-estimates on many-core hosts: thread start-up dominates its 16k paths)
        nframes = int(min(max(round(1.7 * args.cpu_target_s / frame_s), 1.0), 16.0))
        tc = time.perf_counter()
        _, ost = orc.render(W, H, 1, nframes, seed=args.seed, nthreads=cores)
        cpu_s = time.perf_counter() - tc
        cpu_rays = ost["rays_closest"] + ost["rays_shadow"]
        result["cpu_baseline"] = {
            "value": round(cpu_rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "frames 1..%d of the same scene at the full 1024^2 (%d paths, %d rays) in %.2f s on %d threads; "
                      "CPU oracle LBVH build %.3f s (GPU %.3f ms)" %
                      (nframes, ost["paths"], cpu_rays, cpu_s, cores, cpu_build_s, build_ms),
        }

    # RCCL writes its version banner to C stdio (block-buffered when piped): every rank pushes its buffer out, then
    # rank 0 prints the JSON line as the LAST line on stdout (no os._exit here: rocprofv3 writes its output from
    # exit handlers of this process)
    def flush_c():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    flush_c(); sys.stdout.flush()
    if world > 1 or force_dist:
        dist.barrier()
        torch.cuda.synchronize()
    if rank == 0:
        print(json.dumps(result), flush=True)
    sys.stdout.flush(); sys.stderr.flush()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
