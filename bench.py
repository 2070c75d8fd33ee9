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
  roofline      dominant kernel = the traversal kernel k_trace (closest hits of bounce b + NEE shadow
                rays of bounce b-1 in one launch).  rocprofv3 PMC (profiles/) shows it bound by the
                L1 gather path (texture-address unit), not by HBM, so:
                  achieved = bytes of node / primitive / ray records the launch gathers from global memory
                             (device counters of an untimed counting pass over the same frames)
                             / mean launch duration (HIP events on the library's stream, untimed pass)
                  peak     = the ceiling of exactly that access pattern (random 64-byte records,
                             4 x 16-byte loads per lane) measured in THIS run on an array of the BVH's
                             actual byte size (tirt_micro_gather_rate)
                  frac     = achieved / peak
                  traffic  = HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md)
                             from two rocprofv3 --pmc passes this script runs itself on a child process;
                             hbm_frac = traffic / duration / 8 TB/s
                  alg_reference_semantics = SURVEY.md 8d's figure (32 B x N_box + 36 B x N_leaf + 48 B per
                             ray with the REFERENCE's exhaustive pop counts): what the reference's
                             algorithm would move, not what this kernel moves -- reported, not a fraction.
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference algorithm: AoS rows,
                exhaustive unordered traversal, per-pixel loop) built -O3 -march=native on this box,
                on all host cores over a bounded sample of the same scene/frames ("kind": "port"; the
                reference's own ti.cpu path cannot run: Taichi is not installable here).
"""
import argparse
import json
import os
import sys
import time

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
    ap.add_argument("--tile-size", type=int, default=0, help="pixels per film tile (0: PT_RGB.default_tile_size: 8 columns)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes behind roofline.traffic")
    ap.add_argument("--cpu-target-s", type=float, default=12.0, help="seconds of CPU-oracle work for cpu_baseline")
    ap.add_argument("--save-png", default="")
    ap.add_argument("--emulate-world", type=int, default=0, help="render only rank 0's tiles of an N-rank job (scaling study on one GPU)")
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to tirt_set_option (tuning)")
    return ap.parse_args()


def measure_hbm_traffic(args):
    """HBM-side bytes per k_trace launch, measured now: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not
    fit one pass: TCC has 4 counter slots) over a child run of this script (1 warm-up + 1 step, one render lane).
    FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.  Returns a dict, or {"error": ...}."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return {"error": "this process is itself being profiled: nested rocprofv3 pass skipped"}
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "TIRT_FORCE_DIST")}            # the child is a plain one-process run
    env["TMPDIR"] = "/tmp"
    out = {}
    tmp = tempfile.mkdtemp(prefix="tirt_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--frames-per-step", str(args.frames_per_step),
             "--size", str(args.size), "--ntri", str(args.ntri), "--seed", str(args.seed), "--no-cpu-baseline", "--no-roofline",
             "--opt", "overlap_lanes=1",
             # one batch per step, as in the instrumented pass whose launch duration the bytes are divided by
             "--opt", "batch_paths=%d" % (args.frames_per_step * args.size * args.size),
             "--opt", "merge_paths=%d" % (args.frames_per_step * args.size * args.size)] + sum((["--opt", o] for o in args.opt), [])
    def one_pass(counters):
        d = os.path.join(tmp, counters[0])
        cmd = ["timeout", "-k", "5", "240", exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "--"] + child
        p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            raise RuntimeError("%s pass produced no counter file (rc %d)" % (counters[0], p.returncode))
        tot, ids = {c: 0.0 for c in counters}, set()
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            if "k_trace" in r["Kernel_Name"] and r["Counter_Name"] in tot:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
        if not ids:
            raise RuntimeError("no k_trace dispatch in the %s pass" % counters[0])
        return tot, len(ids)
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            tot, n = one_pass((counter,))
            out[counter + "_KB_per_launch"] = round(tot[counter] / n, 1)
            out["launches_" + counter] = n
        # what the kernel is busy with, same child run: VALU issue and texture-address cycles against the kernel's own clock count
        # (SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD, GRBM_GUI_ACTIVE is summed over the 8 XCDs; MI355X: 1024 SIMDs, 256 CUs)
        try:
            tot, n = one_pass(("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum"))
            cyc = tot["GRBM_GUI_ACTIVE"] / 8.0
            if cyc > 0:
                out["valu_busy"] = round(tot["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0), 3)
                out["ta_busy"] = round(tot["TA_TA_BUSY_sum"] / (cyc * 256.0), 3)
        except Exception as exc:        # noqa: BLE001
            out["busy_error"] = "%s: %s" % (type(exc).__name__, exc)
    except Exception as exc:            # noqa: BLE001 -- a failed profiler pass must not fail the bench line
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["bytes_per_launch"] = round((2.0 * out["FETCH_SIZE_KB_per_launch"] + out["WRITE_SIZE_KB_per_launch"]) * 1024.0)
    out["source"] = "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_ACTIVE_INST_VALU + GRBM_GUI_ACTIVE + TA_TA_BUSY_sum) run by bench.py on a child process; FETCH_SIZE x 2 (gfx950)"
    return out


def host_cpu_info():
    info = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    quota = None
    try:                                                   # cgroup v2, then v1
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        except (OSError, ValueError):
            pass
    info["cgroup_cpu_quota"] = quota
    return info


def cpu_baseline(args, ex, W, H, build_ms):
    """The CPU oracle timed on this box's host cores: -O3 -march=native build made here, dynamic chunk queue, bounded
    sample (~cpu_target_s seconds) of the same scene and frames; 1-thread and N-thread rates, host description."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    host = host_cpu_info()
    native = oracle_api.build_native() is not None
    orc = oracle_api.OracleScene(ex.scene, ex.cam, native=native)
    tb = time.perf_counter()
    orc.lbvh_build()
    cpu_build_s = time.perf_counter() - tb
    if native:      # the native build must compute what the portable one does (same IEEE operations)
        ref = oracle_api.OracleScene(ex.scene, ex.cam); ref.lbvh_build()
        a, _ = orc.render(W, H, 1, 1, seed=args.seed, p_begin=W * H // 2, p_end=W * H // 2 + 2048, nthreads=4)
        b, _ = ref.render(W, H, 1, 1, seed=args.seed, p_begin=W * H // 2, p_end=W * H // 2 + 2048, nthreads=4)
        if not np.array_equal(a, b):
            raise SystemExit("native oracle build differs from the portable build")
    cores = host["affinity"] or host["nproc"] or 1
    if host.get("cgroup_cpu_quota"):
        cores = max(1, min(cores, int(host["cgroup_cpu_quota"] + 0.5)))
    # 1-thread rate on a strip in the middle of the film (~2 s)
    t1 = time.perf_counter()
    _, s1 = orc.render(W, H, 1, 1, seed=args.seed, p_begin=W * H // 2, p_end=W * H // 2 + 16384, nthreads=1)
    one_s = max(time.perf_counter() - t1, 1e-6)
    rate1 = (s1["rays_closest"] + s1["rays_shadow"]) / one_s
    # N threads: whole frames, sized from the 1-thread rate assuming linear scaling
    rays_per_frame = (s1["rays_closest"] + s1["rays_shadow"]) * (W * H / 16384.0)
    nframes = int(min(max(round(args.cpu_target_s * rate1 * cores / rays_per_frame), 1.0), 64.0))
    tc = time.perf_counter()
    _, ost = orc.render(W, H, 1, nframes, seed=args.seed, nthreads=cores)
    cpu_s = time.perf_counter() - tc
    cpu_rays = ost["rays_closest"] + ost["rays_shadow"]
    return {
        "value": round(cpu_rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "build": "gcc -O3 -march=native on this host" if native else "portable -O2 build shipped with the repo (native build failed)",
        "one_thread_Mrays_s": round(rate1 / 1e6, 5), "per_thread_Mrays_s": round(cpu_rays / cpu_s / cores / 1e6, 5),
        "parallel_efficiency": round(cpu_rays / cpu_s / cores / rate1, 3), "host": host,
        "sample": "frames 1..%d of the same scene at the full %dx%d (%d paths, %d rays) in %.2f s on %d threads (dynamic 64-pixel "
                  "chunks); 1-thread rate from 16384 pixels of frame 1 in %.2f s; CPU oracle LBVH build %.3f s (GPU %.3f ms)" %
                  (nframes, W, H, ost["paths"], cpu_rays, cpu_s, cores, one_s, cpu_build_s, build_ms),
    }


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
                          tile_size=args.tile_size or None)
    args.tile_size = ex.integrator.tile_size
    t0 = time.time()
    ctx = ex.scene.ctx
    opts = dict(kv.split("=") for kv in args.opt)
    for k, v in opts.items():
        ctx.set_option(k, float(v))
    ex.build_scene()
    ctx.sync()
    build_wall = time.time() - t0
    # build times (HIP events around the build on the context's stream), second build of each kind: the reference's LBVH alone
    # (Morton, sort, Karras, refit, flatten + the 4-wide collapse of it) and with the binned-SAH traversal tree in between
    tree_opt = int(float(opts.get("traversal_tree", "1")))
    build_detail = {}
    for tree in (1 - tree_opt, tree_opt):
        ctx.set_option("traversal_tree", tree)
        ms = []
        for _ in range(4):
            ctx.lbvh_build(); ms.append(ctx.stats()["ms_build"])
        build_ms = min(ms[1:])                         # (the first one allocates; an idle GPU clocks down between Python calls)
        build_detail["with_sah_traversal_tree_ms" if tree else "lbvh_only_ms"] = round(build_ms, 3)
    fps = args.frames_per_step

    # what the example classes tell the library at set-up (Example.build_scene: their sample count): the frames of the job ahead -- it
    # sizes its wavefront batches and lane buffers by that (tirt_internal.h, plan_batches).  Here: the timed job; the warm-up step
    # before it is a job of its own.
    ctx.set_option("job_frames", args.steps * fps)

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
            "traversal": "ordered+t-culled over 4-wide nodes collapsed from a device-built binned-SAH tree; every hit verified against the reference's LBVH (bit-identical to its exhaustive order)",
        },
        "rays": {"closest": int(rays_closest), "shadow": int(rays_shadow), "paths": int(paths),
                 "rays_per_path": round((rays_closest + rays_shadow) / max(paths, 1.0), 3)},
        "lbvh_build_ms": round(build_ms, 3),
        "build_detail": build_detail,
        "scene_setup_wall_s": round(build_wall, 3),
    }

    if rank == 0 and args.save_png:
        from ti_raytrace_amd.Example import write_png
        ctx.film_import_device(film.data_ptr()) if world > 1 else None
        ctx.tone_map(0.5)
        write_png(ctx.film_download(W, H, want_hdr=False, want_rgb=True)[1], args.save_png)

    # ---- roofline for the dominant kernel (rank 0's shard, untimed extra passes) -----------------
    if not args.no_roofline and rank == 0:
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
        n_launch = max(t["launches_trace_closest"] + t["launches_trace_shadow"], 1)
        avg_ms = (t["ms_trace_closest"] + t["ms_trace_shadow"]) / n_launch
        rays_o = co["rays_closest"] + co["rays_shadow"]
        # what the launch gathers from global memory: 64 B per 4-wide node visit that is not served by the LDS copy of
        # the tree top, 48 B per primitive test, 24 B ray fetch + 16 B hit record per ray
        node_visits = (co["box_closest"] + co["box_shadow"]) / 4.0
        lds_visits = float(co["diag_it_outer"])
        gather_bytes = 64.0 * (node_visits - lds_visits) + 48.0 * (co["leaf_closest"] + co["leaf_shadow"]) + 40.0 * rays_o
        achieved = gather_bytes / n_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        info = ctx.bvh_info()
        working_set = info["node_bytes"] + info["prim_bytes"]
        peak_ws = ctx.micro_gather_rate(working_set, 2000)
        peak_l2 = ctx.micro_gather_rate(2 << 20, 2000)
        # The ceiling for uniformly random records of the whole working set is the right one while that set is of the order of the
        # L2s (the default scene: 7.5 MB); a traversal of a much larger scene (--ntri 1000000: 75 MB) re-reads the top of its tree
        # from cache and EXCEEDS it -- then the L2-resident ceiling is the one that still bounds the kernel.
        peak = peak_ws if achieved <= peak_ws else peak_l2
        # (the profiler passes run the one-GPU workload on this rank's device: only at N = 1, where that is the workload timed)
        traffic = None if (args.no_traffic or world > 1) else measure_hbm_traffic(args)
        tr_bytes = traffic["bytes_per_launch"] if traffic and traffic.get("bytes_per_launch") else None
        alg_trace = alg_closest + alg_shadow
        result["roofline"] = {
            # PMC (traffic_detail.valu_busy / ta_busy, measured by this run): the kernel is bound by VALU issue with the L1 gather path
            # (texture-address unit) second; HBM is far from it.  The byte roofline below is stated on the gather path.
            "bound": "valu_issue+l1_gather", "kernel": "k_trace<ordered> (closest-hit + NEE shadow rays)",
            "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "GB/s",
            "frac": round(achieved / peak, 4) if peak > 0 else None,
            "traffic": tr_bytes,
            "hbm_GBps": (round(tr_bytes / (avg_ms * 1e-3) / 1e9, 1) if (tr_bytes and avg_ms > 0) else None),
            "hbm_frac": (round(tr_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (tr_bytes and avg_ms > 0) else None),
            "traffic_detail": traffic,
            "peak_def": "tirt_micro_gather_rate measured in this run: random 64-byte records (4 x dwordx4 per lane) from an array of "
                        "working_set_bytes = the traversal data of this scene (peak_working_set); peak_l2_resident = the same from 2 MB, used as "
                        "peak when the kernel's non-uniform accesses beat the uniform-random ceiling of a working set far beyond the L2s",
            "peak_working_set": round(peak_ws, 1), "peak_l2_resident": round(peak_l2, 1), "working_set_bytes": int(working_set), "bvh": info,
            "achieved_def": "64 B x 4-wide node visits not served from LDS + 48 B x primitive tests + 40 B x rays, ordered-traversal device "
                            "counters of the same frames, / mean k_trace launch duration (HIP events)",
            "gather_bytes_per_launch": round(gather_bytes / n_launch, 1),
            "gather_bytes_per_ray": round(gather_bytes / max(rays_o, 1), 1),
            "avg_launch_ms": round(avg_ms, 5), "launches": int(n_launch),
            "node_visits_per_ray": round(node_visits / max(rays_o, 1), 2),
            "lds_node_visits_per_ray": round(lds_visits / max(rays_o, 1), 2),
            "prim_tests_per_ray": round((co["leaf_closest"] + co["leaf_shadow"]) / max(rays_o, 1), 2),
            "kernel_ms": {"trace_closest": round(t["ms_trace_closest"], 3), "trace_shadow": round(t["ms_trace_shadow"], 3),
                          "shade": round(t["ms_shade"], 3), "render_total": round(t["ms_render"], 3)},
            "wave_diag_ordered": {"node_iters_per_ray": round(co["diag_it_node"] * 64.0 / max(rays_o, 1), 2),
                                  "node_lane_util": round(co["diag_lanes_node"] / max(co["diag_it_node"] * 64.0, 1), 4),
                                  "leaf_iters_per_ray": round(co["diag_it_leaf"] * 64.0 / max(rays_o, 1), 2),
                                  "leaf_lane_util": round(co["diag_lanes_leaf"] / max(co["diag_it_leaf"] * 64.0, 1), 4),
                                  "refills_per_wave_ray": round(co["diag_refills"] * 64.0 / max(rays_o, 1), 3)},
            # SURVEY.md 8d's algorithmic bytes on the REFERENCE's traversal semantics (exhaustive pop counts): what the
            # reference's algorithm would move for these rays -- the ordered traversal moves far less, so this is NOT a
            # fraction of any roofline of this kernel
            "alg_reference_semantics": {
                "bytes_per_launch": round(alg_trace / n_launch, 1),
                "GBps_equivalent": round((alg_trace / n_launch) / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None,
                "bytes_per_closest_ray": round(alg_closest / max(c["rays_closest"], 1), 1),
                "bytes_per_shadow_ray": round(alg_shadow / max(c["rays_shadow"], 1), 1),
                "n_box_per_closest_ray": round(c["box_closest"] / max(c["rays_closest"], 1), 2),
                "n_leaf_per_closest_ray": round(c["leaf_closest"] / max(c["rays_closest"], 1), 2),
                "whole_job_GBps_equivalent": round((alg_closest + alg_shadow + 260.0 * c["shaded"] + 24.0 * c["paths"]) /
                                                   max(t["ms_render"], 1e-9) / 1e6, 2)},
        }

    # ---- CPU baseline: oracle on the host cores, bounded sample (rank 0, N = 1 only) ---------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, ex, W, H, build_ms)

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
