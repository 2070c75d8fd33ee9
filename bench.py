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
line.  Inputs (scene, BVH) are resident in HBM before the timed region; what belongs to the
RENDER -- the camera rays' per-pixel candidate lists -- is forgotten when the clock starts and
made again by the first timed batch (primary_beams.prepare_ms_in_timed_region).  N = 1 also
prints value_cold_256spp: BASELINE config 3 as written (one camera set, 8 x 32 frames, sync).

Extra objects:
  roofline      dominant kernel = the traversal kernel k_trace (closest hits of bounce b + NEE shadow
                rays of bounce b-1 in one launch).  Three ceilings from MI355X_MICROARCH.md, each with the
                kernel's measured use of it (four rocprofv3 --pmc passes this script runs itself on a child
                process; launch duration from HIP events on the library's stream):
                  fractions.hbm  = HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) / duration / 8 TB/s
                  fractions.l2   = L1 -> L2 read bytes (TCP_TCC_READ_REQ, bytes per request calibrated on k_film
                                   in the same pass) / duration / 34.5 TB/s
                  fractions.valu = share of the kernel's cycles a SIMD's VALU is issuing (SQ_ACTIVE_INST_VALU x 4
                                   / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)); valu.useful_lane_throughput = that x
                                   the active-lane fraction
                  bound          = the largest of the three; achieved / peak / unit / frac state that one
                  traffic        = HBM-side bytes per launch
                  gather         = the node / primitive / ray records the launch gathers from global memory
                                   against the two measured ceilings of that access pattern (working-set sized
                                   and L2-resident array) -- both fractions printed, no switching
                  alg_reference_semantics = SURVEY.md 8d's figure (32 B x N_box + 36 B x N_leaf + 48 B per
                             ray with the REFERENCE's exhaustive pop counts): what the reference's
                             algorithm would move, not what this kernel moves -- reported, not a fraction.
  configs       (N = 1) the other BASELINE.json configs after the timed region -- config 1 as the reference
                committed it (Cornell 512^2 x 512 spp), config 2 (Teapot 1024^2 x 64 spp), config 5
                (veach_bdpt 512^2 x 64 spp, with a roofline of the BDPT kernels), the spectral configs, and the same
                scene at 4 M and 8 M triangles (traversal data beyond L2 / beyond the Infinity Cache: k_trace's
                fabric-side traffic with counters calibrated on gathers) -- each: seconds, rays,
                Mrays/s, NaN pixels, a film sample checked against the CPU oracle; LBVH build at 100k / 1 M.
  distributed   (N > 1) rccl_ranks (an all-reduce of ones), equality of the replicated BVH builds (hash
                all-gather), reduce_ms of the one film reduce.
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference algorithm: AoS rows,
                exhaustive unordered traversal, per-pixel loop) built -O3 -march=native on this box,
                on all host cores over a bounded sample of the same scene/frames ("kind": "port"; the
                reference's own ti.cpu path cannot run: Taichi is not installable here).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one HW queue per render lane; before HIP initialises

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
EA_REQ_CEILING_G = 52.5         # G fabric read requests/s (128-byte lines) that streams and random record gathers alike reach from 0.5-2 GB working sets: profiles/r06_micro_hbm_gather.txt
# configs whose traversal data does not fit on the die: name -> (triangles, vertex spread, pixels of the oracle sample)
BIG_SCENES = {"big_scene_4M_1024x1024_32spp": (4000000, 0.006, 1024), "big_scene_8M_1024x1024_32spp": (8000000, 0.0042, 512)}    # (8 M: the largest this data format takes -- compact_node keeps node indices as f32, exact below 2^24 = 2 x 8.39 M nodes, as the reference does)


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
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child passes behind roofline.traffic / fractions")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (the other BASELINE configs, run after the timed region at N = 1)")
    ap.add_argument("--configs-only", default="", help="internal: run one entry of the configs leg (child of the BDPT profiler passes)")
    ap.add_argument("--cpu-target-s", type=float, default=12.0, help="seconds of CPU-oracle work for cpu_baseline")
    ap.add_argument("--save-png", default="")
    ap.add_argument("--emulate-world", type=int, default=0, help="render only rank 0's tiles of an N-rank job (scaling study on one GPU)")
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to tirt_set_option (tuning)")
    ap.add_argument("--keep-lists", action="store_true", help="internal (profiler children): do not rebuild the camera rays' candidate lists at the start of the timed region -- "
                                                              "their counters are those of a steady-state step, not of one step plus a list build")
    return ap.parse_args()


def _short_kernel(name):
    """'void tirt::k_trace<0, false, 2>(tirt::TraceArgs)' -> 'k_trace<0,false,2>'"""
    n = name.split("(")[0].replace("void ", "").replace("tirt::", "").replace(" ", "")
    return n


def rocprof_passes(child, groups, timeout_s=240):
    """Runs `child` (argv) once per counter group under `rocprofv3 --kernel-trace --pmc <group>` (one group per run, as the
    MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass, no tracing domain besides the kernel trace) and
    returns {kernel: {"launches": n, "dur_ns": total, counter: total, ...}} merged over the passes (durations from the first)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        raise RuntimeError("this process is itself being profiled: nested rocprofv3 pass skipped")
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "TIRT_FORCE_DIST")}            # the child is a plain one-process run
    env["TMPDIR"] = "/tmp"
    tmp = tempfile.mkdtemp(prefix="tirt_pmc_", dir="/tmp")
    out = {}
    try:
        for gi, counters in enumerate(groups):
            d = os.path.join(tmp, "g%d" % gi)
            cmd = ["timeout", "-k", "5", str(timeout_s), exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "--"] + child
            pr = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s + 60)
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            out["__child_tail"] = {"launches": 0, "dur_ns": 0.0, "rc": pr.returncode,
                                   "text": " | ".join(l for l in pr.stdout.decode("utf-8", "replace").splitlines() if "rocprofv3" not in l and "output_stream" not in l)[-700:]}
            if not fs:
                raise RuntimeError("%s pass produced no counter file (rc %d)" % (counters[0], pr.returncode))
            seen = set()
            for r in csv.DictReader(open(max(fs, key=os.path.getsize))):
                k = _short_kernel(r["Kernel_Name"])
                e = out.setdefault(k, {"launches": 0, "dur_ns": 0.0})
                e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                if gi == 0 and (k, r["Dispatch_Id"]) not in seen:
                    seen.add((k, r["Dispatch_Id"]))
                    e["launches"] += 1
                    e["dur_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


# Peaks from /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0          # aggregate L2 bandwidth
VALU_PEAK_GINSTR = 1024 * 2.4 / 2.0    # wave64 VALU instructions per ns: 1024 SIMD-32s x 2.4 GHz, 2 cycles per instruction at best (the 157.3 TFLOP/s FP32 vector peak)


def shade_ceilings(sh):
    """The three ceilings of MI355X_MICROARCH.md for k_shade, from its share of the same profiler passes (durations: the profiled ones)."""
    if not sh or not sh.get("avg_launch_ms_profiled"):
        return None
    dur = sh["avg_launch_ms_profiled"] * 1e-3
    o = {"avg_launch_ms_profiled": sh["avg_launch_ms_profiled"], "launches": sh["launches"],
         "hbm": {"bytes_per_launch": sh["bytes_per_launch"], "GBps": round(sh["bytes_per_launch"] / dur / 1e9, 1), "frac": round(sh["bytes_per_launch"] / dur / 1e9 / HBM_PEAK_GBS, 4)}}
    if sh.get("l2_read_bytes_per_launch"):
        o["l2"] = {"read_bytes_per_launch": sh["l2_read_bytes_per_launch"], "GBps": round(sh["l2_read_bytes_per_launch"] / dur / 1e9, 1),
                   "frac": round(sh["l2_read_bytes_per_launch"] / dur / 1e9 / L2_PEAK_GBS, 4), "hit_rate": sh.get("l2_hit_rate")}
    if sh.get("valu_busy") is not None:
        o["valu"] = {"issue_busy": sh["valu_busy"], "lane_util": sh["valu_lane_util"], "frac": round(sh["valu_busy"] * sh["valu_lane_util"], 4),
                     "cycles_per_inst": sh["valu_cycles_per_inst"], "wave_insts_per_launch": sh["valu_wave_insts_per_launch"], "ta_busy": sh.get("ta_busy")}
    fr = {kk: (vv["issue_busy"] if kk == "valu" else vv["frac"]) for kk, vv in o.items() if isinstance(vv, dict)}
    o["bound"] = max(fr, key=fr.get) if fr else None
    o["note"] = ("measured alone on one lane; in the timed region it runs beside the traversal kernels of other batches, whose VALU issue it shares. "
                 "About 45 % of its VALU instructions are the correctly rounded sqrt / division sequences and the f64 sin / cos polynomials that make "
                 "the film bit-identical to the CPU oracle (DESIGN.md section 4)")
    return o


def measure_trace_counters(args):
    """What the traversal kernel does to the memory system and to the VALUs, measured now on a child run of this script
    (1 warm-up + 1 step, one render lane: one wavefront batch, launches do not overlap).  Four rocprofv3 passes:
      FETCH_SIZE | WRITE_SIZE                     HBM-side bytes (FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md)
      TCP_TCC_READ_REQ_sum, TCC_HIT/MISS_sum      L1 -> L2 requests; bytes per request calibrated IN THE SAME PASS on k_film,
                                                  whose reads are known exactly (3 words per path + the film)
      SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_THREAD_CYCLES_VALU, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE + GRBM_GUI_ACTIVE +
      TA_TA_BUSY_sum                              VALU instructions, busy quad-cycles, active lanes, LDS conflict cycles
    Returns a dict (per k_trace launch), or {"error": ...}."""
    fps, P = args.frames_per_step, args.size * args.size
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--frames-per-step", str(fps),
             "--size", str(args.size), "--ntri", str(args.ntri), "--seed", str(args.seed), "--no-cpu-baseline", "--no-roofline", "--no-configs", "--keep-lists",
             "--opt", "overlap_lanes=1",
             # one batch per step, as in the instrumented pass whose launch duration the bytes are divided by
             "--opt", "batch_paths=%d" % (fps * P), "--opt", "merge_paths=%d" % (fps * P)] + sum((["--opt", o] for o in args.opt), [])
    try:
        k = rocprof_passes(child, [("FETCH_SIZE",), ("WRITE_SIZE",), ("TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"),
                                   ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
                                    "GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum")])
    except Exception as exc:            # noqa: BLE001 -- a failed profiler pass must not fail the bench line
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    def digest(match):
        sel = [v for n, v in k.items() if match(n)]
        if not sel:
            return None, None
        tot = {}
        for v in sel:
            for c, x in v.items():
                tot[c] = tot.get(c, 0.0) + x
        n = max(int(tot["launches"]), 1)
        out = {"launches": n, "avg_launch_ms_profiled": round(tot["dur_ns"] / n / 1e6, 5),
               "FETCH_SIZE_KB_per_launch": round(tot.get("FETCH_SIZE", 0.0) / n, 1), "WRITE_SIZE_KB_per_launch": round(tot.get("WRITE_SIZE", 0.0) / n, 1)}
        out["bytes_per_launch"] = round((2.0 * out["FETCH_SIZE_KB_per_launch"] + out["WRITE_SIZE_KB_per_launch"]) * 1024.0)
        cyc = tot.get("GRBM_GUI_ACTIVE", 0.0) / 8.0           # summed over the 8 XCDs
        if cyc > 0 and tot.get("SQ_INSTS_VALU", 0) > 0:
            out["valu_busy"] = round(tot["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0), 4)        # quad-cycles per SIMD -> share of the kernel's cycles
            out["ta_busy"] = round(tot["TA_TA_BUSY_sum"] / (cyc * 256.0), 4)
            out["valu_lane_util"] = round(tot["SQ_THREAD_CYCLES_VALU"] / (tot["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
            out["valu_cycles_per_inst"] = round(tot["SQ_ACTIVE_INST_VALU"] * 4.0 / tot["SQ_INSTS_VALU"], 3)
            out["valu_wave_insts_per_launch"] = round(tot["SQ_INSTS_VALU"] / n)
            out["clock_GHz_profiled"] = round(cyc / tot["dur_ns"], 3)
            out["lds_bank_conflict_frac_of_lds_cycles"] = round(tot.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(tot.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4)
            out["lds_bank_conflict_cycles_frac_of_kernel"] = round(tot.get("SQ_LDS_BANK_CONFLICT", 0.0) / (cyc * 256.0), 4)   # LDS is per CU
        return out, tot

    out, tot = digest(lambda name: name.startswith("k_trace"))
    if out is None:
        return {"error": "no k_trace dispatch in the profiler passes"}
    n = out["launches"]
    if "valu_busy" in out:
        # every kernel of the wavefront loop, both profiled steps: what one step asks of the VALUs (for `whole_job` in the roofline object)
        per = {}
        for name, v in k.items():
            if name.startswith(("k_trace", "k_shade", "k_generate", "k_film", "k_pvb_cand", "k_pvb_scatter")) and v.get("SQ_INSTS_VALU", 0) > 0:
                per[name] = round(v["SQ_INSTS_VALU"] / 2.0)
        out["valu_wave_insts_per_step_by_kernel"] = per
    bytes_per_req = 64.0
    if tot.get("TCP_TCC_READ_REQ_sum", 0) > 0:
        out["l2_hit_rate"] = round(tot["TCC_HIT_sum"] / max(tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"], 1.0), 4)
        film = k.get("k_film")
        if film and film.get("TCP_TCC_READ_REQ_sum", 0) > 0 and film["launches"] > 0:
            known = film["launches"] * (12.0 * fps * P + 12.0 * P)                  # fr, fg, fb of every path + the film itself
            bytes_per_req = known / film["TCP_TCC_READ_REQ_sum"]
            out["l2_bytes_per_request_calibrated_on_k_film"] = round(bytes_per_req, 2)
        out["l2_read_bytes_per_launch"] = round(tot["TCP_TCC_READ_REQ_sum"] * bytes_per_req / n)
    # the shading kernel, same passes, same definitions (17 % of a step's VALU instructions: on the critical path of a VALU-bound job)
    sh, sh_tot = digest(lambda name: name == "k_shade")
    if sh is not None:
        if sh_tot.get("TCP_TCC_READ_REQ_sum", 0) > 0:
            sh["l2_read_bytes_per_launch"] = round(sh_tot["TCP_TCC_READ_REQ_sum"] * bytes_per_req / sh["launches"])
            sh["l2_hit_rate"] = round(sh_tot["TCC_HIT_sum"] / max(sh_tot["TCC_HIT_sum"] + sh_tot["TCC_MISS_sum"], 1.0), 4)
        out["k_shade"] = sh
    out["source"] = ("rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE | WRITE_SIZE | TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum | SQ_INSTS_VALU "
                     "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE TA_TA_BUSY_sum) run by bench.py on a "
                     "child process; FETCH_SIZE x 2 (gfx950)")
    return out


def _oracle_run_identical(ex, W, H, frames, seed, got_hdr, p0, npx=2048):
    """`npx` pixels of the film starting at linear pixel p0, rendered by the CPU oracle (same frames, same seed), compared bit for
    bit (NaNs in the same places count as equal)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    o = oracle_api.OracleScene(ex.scene, ex.cam)
    o.lbvh_build()
    if getattr(ex.scene, "normals_processed", False):
        o.process_normal(ex.scene.vertex_index_np)
    want, _ = o.render(W, H, 0, frames, seed=seed, p_begin=p0, p_end=p0 + npx)
    a = got_hdr.reshape(-1, 3)[p0:p0 + npx]
    b = want.reshape(-1, 3)[p0:p0 + npx]
    return bool(np.array_equal(a, b, equal_nan=True))


def run_config(name, device_id, seed=1):
    """One entry of the `configs` object: a BASELINE.json config other than the headline one, rendered whole on this GPU after a
    warm-up job of the same size: seconds, rays, Mrays/s, NaN pixels, and a sample of the film checked against the CPU oracle."""
    import numpy as np
    from ti_raytrace_amd import scenes
    from ti_raytrace_amd import UtilsFunc as UF
    gold = os.path.join(ROOT, "tests", "golden")

    def timed(ex, spp, render):
        ctx = ex.scene.ctx
        for kv in filter(None, os.environ.get("TIRT_BENCH_CTX_OPTS", "").split(",")):      # (profiler children: overlap_lanes=1)
            k_, v_ = kv.split("="); ctx.set_option(k_, float(v_))
        render(); ctx.sync()                                   # warm-up: the same job (allocations, clocks), then a fresh film
        ctx.film_clear(); ex.cam.frame = 0; ex.cam.frame_cpu[0] = 0; ctx.sync(); ctx.stats_reset()
        t0 = time.perf_counter(); render(); ctx.sync(); dt = time.perf_counter() - t0
        st = ctx.stats()
        rays = st["rays_closest"] + st["rays_shadow"]
        hdr = ex.integrator.hdr.to_numpy()
        return hdr, {"seconds": round(dt, 4), "rays": int(rays), "Mrays_per_s": round(rays / dt / 1e6, 1), "rays_closest": int(st["rays_closest"]),
                     "rays_shadow": int(st["rays_shadow"]), "nan_pixels": int(np.isnan(hdr).any(axis=2).sum()), "lbvh_build_ms": round(st["ms_build"], 3),
                     "prims": int(ex.scene.primitive_count)}

    def block_pin(ex, hdr, fixture, exposure=0.5):
        """mean colour and 16x16-block rel-L2 of the tone-mapped film against the block means of a reference PNG"""
        ex.scene.ctx.tone_map(exposure)
        rgb = ex.integrator.rgb_film.to_numpy()
        img = np.nan_to_num(np.transpose(rgb, (1, 0, 2))[::-1])
        ref = np.load(os.path.join(gold, fixture)).astype(np.float64)
        s_ = img.shape[0] // 32
        ours = img.reshape(32, s_, 32, s_, 3).mean(axis=(1, 3))
        return {"fixture": fixture, "mean_srgb": [round(float(x), 4) for x in ours.reshape(-1, 3).mean(0)],
                "ref_mean_srgb": [round(float(x), 4) for x in ref.reshape(-1, 3).mean(0)],
                "block_rel_l2": round(float(np.sqrt(((ours - ref) ** 2).sum() / (ref ** 2).sum())), 4)}

    if name == "config1_cornell_512x512_512spp":       # the reference's own committed run (Main.py:14): out.png
        W = H = 512; spp = 512
        ex = scenes.cornell_box(W, H, spp, device_id=device_id, seed=seed); ex.build_scene()
        hdr, r = timed(ex, spp, lambda: ex.integrator.render_frames(spp))
        r["oracle_sample_identical"] = _oracle_run_identical(ex, W, H, spp, seed, hdr, (W // 2) * H + 128)
        r["oracle_sample"] = "2048 pixels x 512 spp, bit for bit"
        r["reference_pin"] = block_pin(ex, hdr, "out_png_blocks.npy")
        return r
    if name == "config2_teapot_1024x1024_64spp":
        W = H = 1024; spp = 64
        ex = scenes.single_model(W, H, spp, device_id=device_id, seed=seed); ex.build_scene()
        hdr, r = timed(ex, spp, lambda: ex.integrator.render_frames(spp))
        r["oracle_sample_identical"] = _oracle_run_identical(ex, W, H, spp, seed, hdr, (W // 2) * H + 300)
        r["oracle_sample"] = "2048 pixels x 64 spp through the teapot, bit for bit (NaN pixels included)"
        return r
    if name == "config5_veach_bdpt_512x512_64spp":
        W = H = 512; spp = 64
        ex = scenes.veach_bdpt(W, H, spp, device_id=device_id, seed=seed); ex.build_scene()
        hdr, r = timed(ex, spp, lambda: ex.scene.ctx.bdpt_rgb_render(0, spp, seed))
        r["reference_pin"] = block_pin(ex, hdr, "veach_bdpt512_blocks.npy")
        # BDPT splats land on any pixel, so the oracle check is a whole (small) film: 48^2 x 4 spp of the same scene
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api
        w = 48
        ex2 = scenes.veach_bdpt(w, w, 4, device_id=device_id, seed=seed); ex2.build_scene()
        ex2.scene.ctx.bdpt_rgb_render(0, 4, seed); got = ex2.integrator.hdr.to_numpy()
        o = oracle_api.OracleScene(ex2.scene, ex2.cam); o.lbvh_build(); o.process_normal(ex2.scene.vertex_index_np)
        want, ost, _ = o.bdpt_render(ex2.cam, w, w, 0, 4, seed=seed)
        rel = float(np.sqrt(((got.astype(np.float64) - want) ** 2).sum() / max((want.astype(np.float64) ** 2).sum(), 1e-30)))
        st2 = ex2.scene.ctx.stats()
        r["oracle_sample_identical"] = bool(rel <= 1e-5 and np.array_equal(np.isnan(got), np.isnan(want)))
        r["oracle_sample"] = "whole 48x48 film x 4 spp: rel-L2 %.2e (float-atomic splats: order-dependent in the last bits), same NaN mask, " \
                             "ray counts equal: %s" % (rel, bool(st2["rays_closest"] + st2["rays_shadow"] >= ost["rays_closest"] + ost["rays_shadow"]))
        return r
    if name == "spectral_cornell_512x512_64spp":       # example/spectral_box.py through PT_Spec (hero wavelengths), rgb2spec table built on this GPU
        W = H = 512; spp = 64
        t0 = time.perf_counter()
        ex = scenes.spectral_box(W, H, spp, device_id=device_id, seed=seed); ex.build_scene(); ex.scene.ctx.sync()
        t_setup = time.perf_counter() - t0
        hdr, r = timed(ex, spp, lambda: ex.integrator.render_frames(spp))
        r["setup_seconds_incl_rgb2spec_table_64cubed"] = round(t_setup, 3)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api
        o = oracle_api.OracleScene(ex.scene, ex.cam); o.lbvh_build()
        if getattr(ex.scene, "normals_processed", False):
            o.process_normal(ex.scene.vertex_index_np)
        o.set_spectral(ex.integrator.tables())
        p0 = (W // 2) * H + 128
        want, _ = o.spec_render(W, H, 0, spp, seed=ex.integrator.seed, p_begin=p0, p_end=p0 + 1024)
        r["oracle_sample_identical"] = bool(np.array_equal(hdr.reshape(-1, 3)[p0:p0 + 1024], want.reshape(-1, 3)[p0:p0 + 1024], equal_nan=True))
        r["oracle_sample"] = "1024 pixels x 64 spp, bit for bit"
        r["reference_pin"] = "structure pin against image/spectral-cornellbox.png: tests/test_spectral.py (the gallery render is of an earlier scene set-up; no radiometric pin)"
        return r
    if name == "prism_rainbow_bdpt_spec_512x512_64spp":       # SURVEY 8f rank 4, second half: example/prism_rainbow.py through BDPT_SPEC
        W = H = 512; spp = 64
        ex = scenes.prism_rainbow(W, H, spp, device_id=device_id, seed=seed); ex.build_scene()
        hdr, r = timed(ex, spp, lambda: ex.integrator.render_frames(spp))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api
        w = 48                                                    # splats land on any pixel: the oracle check is a whole (small) film, as for config 5
        ex2 = scenes.prism_rainbow(w, w, 3, device_id=device_id, seed=seed); ex2.build_scene()
        ex2.integrator.render_frames(3); got = ex2.integrator.hdr.to_numpy()
        o = oracle_api.OracleScene(ex2.scene, ex2.cam); o.lbvh_build(); o.set_spectral(ex2.integrator.tables())
        want, ost, _ = o.bdpt_spec_render(ex2.cam, w, w, 0, 3, seed=seed, stack_size=1024)
        rel = float(np.sqrt(((got.astype(np.float64) - want) ** 2).sum() / max((want.astype(np.float64) ** 2).sum(), 1e-30)))
        st2 = ex2.scene.ctx.stats()
        r["oracle_sample_identical"] = bool(rel <= 1e-5 and st2["rays_closest"] >= ost["rays_closest"])
        r["oracle_sample"] = "whole 48x48 film x 3 spp: rel-L2 %.2e (float-atomic splats)" % rel
        r["reference_pin"] = "structure pin against image/rainbow.png: tests/test_bdpt_spec.py, tests/test_gpu_bdpt_spec.py"
        return r
    if name in BIG_SCENES:       # the regime north_star describes: a tree that does not fit on the die (4 M triangles: 0.30 GB of nodes + primitive records ~ the 256 MiB Infinity Cache; 8 M: 0.6 GB = 2.2 x it)
        ntri, spread, npx = BIG_SCENES[name]
        W = H = 1024; spp = 32
        t0 = time.perf_counter()
        ex = scenes.synthetic(W, H, spp, ntri=ntri, spread=spread, device_id=device_id, seed=seed); ex.build_scene(); ex.scene.ctx.sync()
        t_setup = time.perf_counter() - t0
        hdr, r = timed(ex, spp, lambda: ex.integrator.render_frames(spp))
        info = ex.scene.ctx.bvh_info()
        r["scene"] = "synthetic random mesh, %d triangles (scene seed 1234, s = %g: about the headline scene's covered area per volume), PT_RGB, max_depth 15" % (ntri, spread)
        r["traversal_bytes"] = int(info["node_bytes"] + info["prim_bytes"])
        r["traversal_bytes_over_infinity_cache"] = round((info["node_bytes"] + info["prim_bytes"]) / float(256 << 20), 2)
        r["setup_seconds_host_packing_and_build"] = round(t_setup, 2)
        r["gather_ceiling_GBps_for_this_working_set"] = round(ex.scene.ctx.micro_gather_rate(info["node_bytes"] + info["prim_bytes"], 1000), 1)
        if not os.environ.get("TIRT_BENCH_CTX_OPTS"):      # (not in the profiler children)
            t1 = time.perf_counter()
            r["oracle_sample_identical"] = _oracle_run_identical(ex, W, H, spp, seed, hdr, (W // 2) * H + 128, npx)
            r["oracle_sample"] = "%d pixels x 32 spp, bit for bit (CPU oracle incl. its LBVH build: %.1f s)" % (npx, time.perf_counter() - t1)
        return r
    raise SystemExit("unknown config " + name)


def big_scene_roofline(cfg, name="big_scene_4M_1024x1024_32spp"):
    """Memory-side traffic of k_trace on a scene whose traversal data does not fit on the die (four rocprofv3 passes over a child run of the config on ONE lane): the
    `>= 40 % of the HBM roofline` question of north_star.  What the counters mean was calibrated on k_trace's own access pattern (tools/micro/hbm_gather.hip,
    profiles/r06_micro_hbm_gather.txt: random 64-byte and 128-byte record gathers and a coalesced stream over working sets of 2 MB .. 8 GB): FETCH_SIZE is 64 bytes
    per read request the L2 sends to the fabric (TCC_EA0_RDREQ) for all three patterns -- half the bytes of a stream and of 128-byte records, all the bytes of
    64-byte records -- and all three saturate at the same 52-53 G requests/s, i.e. a request moves a 128-byte line whatever part of it was asked for: bytes moved =
    requests x 128 = FETCH_SIZE x 2, for gathers as for streams.  The same file shows FETCH_SIZE counting requests that hit in the 256 MiB Infinity Cache (a 128 MB
    working set: 0.97 of the known bytes), so `hbm` below is fabric-side traffic, of which `infinity_cache_share_at_most` can have stopped short of HBM."""
    child = [sys.executable, os.path.abspath(__file__), "--configs-only", name + ":overlap_lanes=1"]
    try:
        k = rocprof_passes(child, [("FETCH_SIZE",), ("WRITE_SIZE",), ("TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"), ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum")], timeout_s=600)
    except Exception as exc:            # noqa: BLE001
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    tot = {}
    for n, v in k.items():
        if n.startswith("k_trace"):
            for c_, x in v.items():
                tot[c_] = tot.get(c_, 0.0) + x
    if not tot.get("launches"):
        return {"error": "no k_trace dispatch in the profiler passes", "child": k.get("__child_tail")}
    n = tot["launches"] / 2.0                                   # the child renders the job twice (warm-up + timed)
    by = (2.0 * tot.get("FETCH_SIZE", 0.0) + tot.get("WRITE_SIZE", 0.0)) * 1024.0
    hbm = by / tot["dur_ns"]
    out = {"bound": "hbm", "kernel": "k_trace", "achieved": round(hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 4),
           "traffic": round(by / tot["launches"]), "launches_per_job": int(n), "avg_launch_ms_profiled": round(tot["dur_ns"] / tot["launches"] / 1e6, 4),
           "k_trace_ms_per_job_profiled": round(tot["dur_ns"] / 2e6, 3),
           "fetch_correction": 2.0, "fetch_correction_calibrated_on": "gather (tools/micro/hbm_gather.hip, profiles/r06_micro_hbm_gather.txt: one fabric read request = one 128-byte line, "
                                                                      "tallied as 64 bytes by FETCH_SIZE, for 64-byte record gathers, 128-byte record gathers and streams alike)"}
    if tot.get("TCC_EA0_RDREQ_sum"):
        rq = tot["TCC_EA0_RDREQ_sum"] / tot["dur_ns"]             # G requests/s
        out["fabric_read_requests"] = {"G_per_s": round(rq, 2), "measured_ceiling_G_per_s": EA_REQ_CEILING_G, "frac_of_ceiling": round(rq / EA_REQ_CEILING_G, 4),
                                       "bytes_x128_GBps": round(rq * 128.0, 1), "share_32B": round(tot.get("TCC_EA0_RDREQ_32B_sum", 0.0) / tot["TCC_EA0_RDREQ_sum"], 4),
                                       "fetch_size_bytes_per_request": round(tot.get("FETCH_SIZE", 0.0) * 1024.0 / tot["TCC_EA0_RDREQ_sum"], 2),
                                       "def": "TCC_EA0_RDREQ_sum / profiled k_trace time; ceiling: what random record gathers AND a coalesced stream reach on this device from "
                                              "working sets of 0.5-2 GB (profiles/r06_micro_hbm_gather.txt) = 6.7 TB/s in 128-byte lines"}
    if tot.get("TCC_HIT_sum", 0) + tot.get("TCC_MISS_sum", 0) > 0:
        out["l2_hit_rate"] = round(tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]), 4)
        out["l1_to_l2_read_requests_G_per_s"] = round(tot["TCP_TCC_READ_REQ_sum"] / tot["dur_ns"], 2)
        out["l2_miss_requests_G_per_s"] = round(tot["TCC_MISS_sum"] / tot["dur_ns"], 2)
    if cfg.get("traversal_bytes"):
        share = min(1.0, float(256 << 20) / cfg["traversal_bytes"])
        out["infinity_cache_share_at_most"] = round(share, 3)
        out["hbm_frac_if_the_infinity_cache_held_all_it_can"] = round(hbm / HBM_PEAK_GBS * (1.0 - share), 4)
        out["infinity_cache_note"] = ("the L2's misses go to the deep nodes and the primitive records, spread over the whole working set: the Infinity Cache can hold at most "
                                      "this share of it, and FETCH_SIZE / TCC_EA0_RDREQ count its hits with the HBM reads")
    if "rays" in cfg and cfg.get("seconds"):
        out["hbm_bytes_per_ray"] = round(by / 2.0 / max(cfg["rays"], 1), 1)
    out["source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum over "
                     "`bench.py --configs-only %s:overlap_lanes=1` (one lane, profiled durations)" % name)
    return out


def bdpt_roofline(device_id):
    """roofline of the BDPT wavefront's own kernels (config 5): per kernel the HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE,
    two rocprofv3 passes over a child run of the config on ONE lane, so that launches do not overlap) / its profiled duration against
    8 TB/s; `dominant` = the k_bd_* kernel with the largest share of the batch's time."""
    child = [sys.executable, os.path.abspath(__file__), "--configs-only", "config5_veach_bdpt_512x512_64spp:overlap_lanes=1"]
    try:
        k = rocprof_passes(child, [("FETCH_SIZE",), ("WRITE_SIZE",)])
    except Exception as exc:            # noqa: BLE001
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    rows = {}
    total = sum(v["dur_ns"] for n, v in k.items() if n.startswith("k_bd") or n.startswith("k_trace")) or 1.0
    for n, v in k.items():
        if not (n.startswith("k_bd") or n.startswith("k_trace")) or v["launches"] == 0:
            continue
        by = (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
        rows[n] = {"launches": v["launches"], "avg_ms": round(v["dur_ns"] / v["launches"] / 1e6, 4), "time_share": round(v["dur_ns"] / total, 4),
                   "hbm_bytes_per_launch": round(by / v["launches"]), "hbm_GBps": round(by / v["dur_ns"], 1),
                   "hbm_frac": round(by / v["dur_ns"] / HBM_PEAK_GBS, 4)}
    bd = {n: r for n, r in rows.items() if n.startswith("k_bd")}
    if not bd:
        return {"error": "no k_bd_* dispatch in the profiler passes", "child": k.get("__child_tail")}
    dom = max(bd, key=lambda n: bd[n]["time_share"])
    return {"bound": "hbm", "kernel": dom, "achieved": rows[dom]["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rows[dom]["hbm_frac"],
            "traffic": rows[dom]["hbm_bytes_per_launch"], "kernels": rows,
            "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE over `bench.py --configs-only config5...:overlap_lanes=1` (one lane, profiled durations)"}


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
    one_s = 1e30
    for _ in range(2):                  # (best of two: the first touches the tree cold)
        t1 = time.perf_counter()
        _, s1 = orc.render(W, H, 1, 1, seed=args.seed, p_begin=W * H // 2, p_end=W * H // 2 + 16384, nthreads=1)
        one_s = min(one_s, max(time.perf_counter() - t1, 1e-6))
    rate1 = (s1["rays_closest"] + s1["rays_shadow"]) / one_s
    # N threads: whole frames, sized from the 1-thread rate assuming linear scaling
    rays_per_frame = (s1["rays_closest"] + s1["rays_shadow"]) * (W * H / 16384.0)
    nframes = int(min(max(round(args.cpu_target_s * rate1 * cores / rays_per_frame), 1.0), 64.0))
    tc = time.perf_counter()
    _, ost = orc.render(W, H, 1, nframes, seed=args.seed, nthreads=cores)
    cpu_s = time.perf_counter() - tc
    cpu_rays = ost["rays_closest"] + ost["rays_shadow"]
    # parallel efficiency on IDENTICAL work: the strip and frame of the one-thread probe again on all threads (best of three; 256 chunks of 64 pixels)
    strip_s = 1e30
    for _ in range(3):
        ts = time.perf_counter()
        _, sN = orc.render(W, H, 1, 1, seed=args.seed, p_begin=W * H // 2, p_end=W * H // 2 + 16384, nthreads=cores)
        strip_s = min(strip_s, max(time.perf_counter() - ts, 1e-6))
    assert sN["rays_closest"] == s1["rays_closest"] and sN["rays_shadow"] == s1["rays_shadow"]
    return {
        "value": round(cpu_rays / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "build": "gcc -O3 -march=native on this host" if native else "portable -O2 build shipped with the repo (native build failed)",
        "one_thread_Mrays_s": round(rate1 / 1e6, 5), "per_thread_Mrays_s": round(cpu_rays / cpu_s / cores / 1e6, 5),
        "parallel_efficiency": round(one_s / strip_s / cores, 3),
        "parallel_efficiency_def": "time of the one-thread probe / time of the same 16384 pixels of frame 1 on all %d threads / %d" % (cores, cores), "host": host,
        "sample": "frames 1..%d of the same scene at the full %dx%d (%d paths, %d rays) in %.2f s on %d threads (dynamic 64-pixel "
                  "chunks); 1-thread rate from 16384 pixels of frame 1 in %.2f s; CPU oracle LBVH build %.3f s (GPU %.3f ms)" %
                  (nframes, W, H, ost["paths"], cpu_rays, cpu_s, cores, one_s, cpu_build_s, build_ms),
    }


def _with_timeout(what, fn, seconds):
    """Runs fn() on a helper thread; if it has not returned after `seconds` the process reports why and exits (a stalled RCCL /
    process-group set-up would otherwise hang until the driver's own limit)."""
    import threading
    box = {}

    def run():
        try:
            box["v"] = fn()
        except BaseException as exc:      # noqa: BLE001
            box["e"] = exc
    th = threading.Thread(target=run, daemon=True)
    th.start(); th.join(seconds)
    if th.is_alive():
        sys.stderr.write("bench.py: %s did not finish within %d s (rank %s of %s, MASTER_ADDR=%s MASTER_PORT=%s) -- giving up\n" %
                         (what, seconds, os.environ.get("RANK"), os.environ.get("WORLD_SIZE"), os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")))
        sys.stderr.flush()
        os._exit(3)
    if "e" in box:
        raise box["e"]
    return box.get("v")


def self_launch(args):
    """`python bench.py --gpus N` typed as is (N > 1, no launcher): start one process per GPU through torch.distributed.run on
    127.0.0.1 with a free port and this command line, pass its output through (rank 0 prints the one JSON line) and exit with
    its code.  The parent touches neither torch nor HIP.  The reference is single-device (integrator/PT_RGB.py:44-49): there is
    nothing to match here, only the driver's contract."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(cmd[1:9])))
    sys.stderr.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.configs_only:
        self_launch(args)
    if args.no_roofline:            # the quick form every tool and test uses: headline number only
        args.no_configs = True
    if args.configs_only:                       # child of bdpt_roofline's profiler passes: one config, optional context options
        name, _, optstr = args.configs_only.partition(":")
        if optstr:
            os.environ["TIRT_BENCH_CTX_OPTS"] = optstr
        print(json.dumps({name: run_config(name, 0, args.seed)}), flush=True)
        return
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # a rank renders on device LOCAL_RANK when the process sees all of the node's GPUs, on its only device when the launcher
    # gave every rank one visible device (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES per rank): local_rank modulo what is visible
    n_visible = torch.cuda.device_count()
    if world > 1 and n_visible != 1 and n_visible < int(os.environ.get("LOCAL_WORLD_SIZE", world)) \
            and os.environ.get("TIRT_BENCH_ONE_DEVICE", "0") != "1":
        raise SystemExit("%d ranks on this node but %d visible GPUs (one process per GPU; TIRT_BENCH_ONE_DEVICE=1 + "
                         "TIRT_BENCH_BACKEND=gloo is the self-test that shares one)" % (int(os.environ.get("LOCAL_WORLD_SIZE", world)), n_visible))
    local_rank %= max(n_visible, 1)
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
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("TIRT_BENCH_PG_TIMEOUT_S", "180")))
        if backend == "nccl":
            _with_timeout("init_process_group(nccl)", lambda: dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo), tmo.seconds + 30)
        else:
            _with_timeout("init_process_group(%s)" % backend, lambda: dist.init_process_group(backend, timeout=tmo), tmo.seconds + 30)

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
    # the first collective creates the RCCL communicator: bounded, with a clear message if it stalls
    _with_timeout("the first RCCL collective (communicator set-up)", lambda: tdist.warmup(ctx, W, H, force=force_dist), 240)
    dist_info = None
    if world > 1 or force_dist:
        # how many ranks RCCL really connected: an all-reduce of ones over the communicator the film reduce will use
        ones = torch.ones(1, dtype=torch.float32, device="cuda")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        # replicated builds must be identical: every rank hashes its LBVH (compact_node) and its traversal tree, the hashes are
        # all-gathered, a mismatch aborts the run (each rank would render with a different tree: same film, but it would hide a
        # nondeterministic build)
        import hashlib
        n_prims = ex.scene.primitive_count
        _, _, compact = ctx.lbvh_download(n_prims, want_morton=False, want_bvh=False)
        tree = ctx.traversal_tree_download(n_prims)
        h = hashlib.blake2b(compact.tobytes(), digest_size=8); h.update(tree.tobytes())
        mine = torch.tensor([int.from_bytes(h.digest(), "little", signed=True)], dtype=torch.int64, device="cuda")
        allh = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allh, mine)
        hashes = [int(x.item()) for x in allh]
        if len(set(hashes)) != 1:
            raise SystemExit("replicated BVH builds differ between ranks: %s" % hashes)
        dist_info = {"backend": dist.get_backend(), "rccl_ranks": int(round(float(ones.item()))), "world_size": dist.get_world_size(),
                     "bvh_hash_equal_on_all_ranks": True, "bvh_hash": "%016x" % (hashes[0] & 0xffffffffffffffff)}
    barrier()
    ctx.stats_reset()
    # Everything a render pays for is inside the clock: the camera rays' candidate lists are a per-camera structure -- render work, not scene build --
    # so the ones the warm-up made are forgotten here and the first timed batch makes them again (probe rays + the walk of the pixels' pyramids, on the
    # context's stream: `primary_beams.prepare_ms_in_timed_region`).  The reference's loop has no warm-up either (example/Example.py:38-59).
    if not args.keep_lists:
        ctx.set_option("primary_beams_rebuild", 1)
    t_begin = time.perf_counter()
    run_steps(args.steps)
    t_submitted = time.perf_counter()             # the host has queued everything (the last, deferred batch goes out with the sync below)
    ctx.sync()
    t_reduce = time.perf_counter()
    film = tdist.reduce_film(ctx, W, H, dst=0, force=force_dist)   # one RCCL reduce of the framebuffer (world > 1)
    reduce_ms = (time.perf_counter() - t_reduce) * 1e3            # this rank: export, RCCL reduce, import (after its own rendering finished)
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
        "host_submit_ms": round((t_submitted - t_begin) * 1e3, 3),          # rank 0: time the host spent queueing the timed steps (inside the timed region)
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
            "traversal": "ordered+t-culled over 4-wide nodes collapsed from a device-built binned-SAH tree; every hit verified against the reference's LBVH (bit-identical to its exhaustive order except for rays lying in a triangle's plane to fp32 rounding, DESIGN.md section 2: none in a render); the camera rays (a third of all rays) against per-pixel lists of the leaves they can hit first, the rest of them through the same traversal (csrc/tirt_pvb.hip: same hit records bit for bit)",
        },
        "rays": {"closest": int(rays_closest), "shadow": int(rays_shadow), "paths": int(paths),
                 "rays_per_path": round((rays_closest + rays_shadow) / max(paths, 1.0), 3)},
        "lbvh_build_ms": round(build_ms, 3),
        "build_detail": build_detail,
        "scene_setup_wall_s": round(build_wall, 3),
    }
    try:
        pb = ctx.primary_beam_stats()
        if pb["rays"]:
            result["primary_beams"] = {"pixels_with_list": pb["pixels_with_list"], "leaves_per_listed_pixel": round(pb["leaves_listed"] / max(pb["pixels_with_list"], 1), 2),
                                       "camera_rays_through_lists": pb["rays"], "of_them_to_k_trace": pb["rays_to_k_trace"],
                                       "share_to_k_trace": round(pb["rays_to_k_trace"] / max(pb["rays"], 1), 4),
                                       "list_builds_in_timed_region": pb["list_builds"], "prepare_ms_in_timed_region": round(pb["list_build_ms"], 4),
                                       "list_builds_given_up_for_lack_of_memory": pb["list_builds_skipped"],
                                       "def": "rank 0, the timed region: the lists are forgotten at its start and made again inside it (tirt.h tirt_primary_beam_stats; "
                                              "prepare_ms = HIP-event time of the probe-ray launch and k_pvb_beam on the context's stream)"}
    except Exception as e:                      # (diagnostics only)
        result["primary_beams"] = {"error": str(e)}
    if dist_info is not None:
        red_t = torch.tensor([reduce_ms], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(red_t, op=dist.ReduceOp.MAX)
        dist_info["reduce_ms"] = round(float(red_t.item()), 3)
        dist_info["reduce_def"] = "max over ranks of: device-to-device export of the film, one RCCL reduce(SUM) of W*H*3 f32 onto rank 0, import on rank 0 -- inside the timed region"
        result["distributed"] = dist_info

    if rank == 0 and args.save_png:
        from ti_raytrace_amd.Example import write_png
        ctx.film_import_device(film.data_ptr()) if world > 1 else None
        ctx.tone_map(0.5)
        write_png(ctx.film_download(W, H, want_hdr=False, want_rgb=True)[1], args.save_png)

    # ---- the headline film itself against the CPU oracle: 2048 pixels x every frame rendered so far (warm-up + timed), bit for bit; after the
    #      timed region, on rank 0 (whose context holds the reduced film when N > 1), before the roofline probes add frames -------------------
    if rank == 0 and not args.no_configs:
        n_frames_film = int(ex.cam.frame)
        t_or = time.perf_counter()
        hdr_now = ctx.film_download(W, H)[0]
        p0 = (W // 2) * H + min(128, max(H - 2048, 0))
        npx = min(2048, W * H - p0)
        result["oracle_sample_identical"] = _oracle_run_identical(ex, W, H, n_frames_film, args.seed, hdr_now, p0, npx)
        result["oracle_sample"] = "%d pixels from linear pixel %d x %d frames (warm-up + timed) rendered by the CPU oracle in %.1f s: film words equal bit for bit" % (
            npx, p0, n_frames_film, time.perf_counter() - t_or)

    # ---- BASELINE config 3 literally, cold: ONE camera set, then 8 x 32 frames = 256 spp, then sync -- the job the metric is quoted on, whatever --steps says, with
    #      the lists of that camera made inside the clock (and nothing else of this job done before it: the batch plan is for 256 frames, the film goes on) ----
    if rank == 0 and world == 1 and not force_dist and not args.no_roofline and args.emulate_world == 0:
        try:
            ctx.sync()
            ctx.set_option("job_frames", 8 * 32)
            ex.cam.update()                                   # tirt_camera_set: the one camera set of the job
            ctx.set_option("primary_beams_rebuild", 1)
            ctx.sync(); ctx.stats_reset()
            t0 = time.perf_counter()
            for _ in range(8):
                ex.integrator.render_frames(32); ex.cam.update_frame(32)
            ctx.sync()
            dt = time.perf_counter() - t0
            stc = ctx.stats(); pbc = ctx.primary_beam_stats()
            result["value_cold_256spp"] = round((stc["rays_closest"] + stc["rays_shadow"]) / dt / 1e6, 3)
            result["cold_256spp"] = {"seconds": round(dt, 5), "rays": int(stc["rays_closest"] + stc["rays_shadow"]), "list_builds": pbc["list_builds"],
                                     "prepare_ms": round(pbc["list_build_ms"], 4),
                                     "def": "BASELINE config 3 as written: one tirt_camera_set, 8 x render_frames(32) at %dx%d, sync; candidate lists made inside the clock; "
                                            "runs after the timed region on the same context (buffers allocated, clocks up)" % (W, H)}
        except Exception as e:                  # noqa: BLE001 -- must not hide the headline line
            result["cold_256spp"] = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            ctx.set_option("job_frames", args.steps * fps)

    # ---- the same steps with the camera rays' candidate lists switched off (one GPU only; after the timed region and the oracle sample): what `value` would be
    #      if bounce 0 went through k_trace like every other bounce -- the lists are a per-camera structure made in the warm-up, like the BVH, and `value` uses them
    # (not in the profiler children and tuning runs, which pass --no-roofline: their kernel statistics are of the timed configuration only)
    if rank == 0 and world == 1 and not force_dist and not args.no_roofline and isinstance(result.get("primary_beams"), dict) and "error" not in result["primary_beams"]:
        try:
            ctx.set_option("primary_beams", 0)
            run_steps(1); ctx.sync()
            ctx.stats_reset()
            t0 = time.perf_counter()
            run_steps(args.steps); ctx.sync()
            dt = time.perf_counter() - t0
            st0 = ctx.stats()
            result["primary_beams"]["value_with_the_lists_off"] = round((st0["rays_closest"] + st0["rays_shadow"]) / dt / 1e6, 3)
            result["primary_beams"]["ms_per_step_with_the_lists_off"] = round(dt * 1e3 / max(args.steps, 1), 4)
        finally:
            ctx.set_option("primary_beams", 1)

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
        ctx.set_option("time_kernels", 1)            # per-kernel HIP events; runs the batches on ONE lane (no overlap)
        ctx.stats_reset()
        pb0 = ctx.primary_beam_stats()
        ctx.pt_rgb_render(f0, probe_frames, args.seed, 15, 64, _native.TRAVERSE_ORDERED | _native.COUNT_NODES)
        ctx.sync()
        co = ctx.stats()
        pb1 = ctx.primary_beam_stats()
        # camera rays that went through their pixels' candidate lists (k_pvb_cand) are not k_trace's: its launches of this pass traced the rest
        listed = (pb1["rays"] - pb0["rays"]) - (pb1["rays_to_k_trace"] - pb0["rays_to_k_trace"]) if pb1["rays"] >= pb0["rays"] else 0
        ctx.stats_reset()
        ctx.pt_rgb_render(f0, probe_frames, args.seed, 15, 64, 0)
        ctx.sync()
        t = ctx.stats()
        ctx.set_option("time_kernels", 0)
        n_launch = max(t["launches_trace_closest"] + t["launches_trace_shadow"], 1)
        avg_ms = (t["ms_trace_closest"] + t["ms_trace_shadow"]) / n_launch
        rays_o = co["rays_closest"] + co["rays_shadow"] - listed          # the rays k_trace traced in that pass
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
        # (the profiler passes run the one-GPU workload on this rank's device: only at N = 1, where that is the workload timed)
        pmc = None if (args.no_traffic or world > 1) else measure_trace_counters(args)
        ok = bool(pmc) and "error" not in pmc
        tr_bytes = pmc["bytes_per_launch"] if ok else None
        dur_s = avg_ms * 1e-3
        # Three ceilings from MI355X_MICROARCH.md, each with the kernel's measured use of it; `bound` is the largest fraction.
        #   hbm  : HBM-side bytes (counters) / launch duration / 8 TB/s
        #   l2   : L1 -> L2 read bytes (counters) / launch duration / 34.5 TB/s
        #   valu : share of the kernel's cycles in which a SIMD's VALU is issuing (SQ_ACTIVE_INST_VALU) -- the instruction mix of a node
        #          visit (v_fma_mix, v_min3/v_max3, v_cndmask, v_alignbit) issues at ~4 cycles per wave64 instruction, twice the 2 cycles
        #          behind the 157.3 TFLOP/s FP32 peak, so the same fact reads as `rate_frac_of_fp32_peak` ~ half of `issue_busy`
        fr = {"hbm": None, "l2": None, "valu": None}
        hbm = l2 = valu = None
        if ok and dur_s > 0:
            hbm = {"bytes_per_launch": tr_bytes, "GBps": round(tr_bytes / dur_s / 1e9, 1), "peak_GBps": HBM_PEAK_GBS}
            fr["hbm"] = hbm["frac"] = round(hbm["GBps"] / HBM_PEAK_GBS, 4)
            if pmc.get("l2_read_bytes_per_launch"):
                l2 = {"read_bytes_per_launch": pmc["l2_read_bytes_per_launch"], "GBps": round(pmc["l2_read_bytes_per_launch"] / dur_s / 1e9, 1),
                      "peak_GBps": L2_PEAK_GBS, "hit_rate": pmc.get("l2_hit_rate")}
                fr["l2"] = l2["frac"] = round(l2["GBps"] / L2_PEAK_GBS, 4)
            if pmc.get("valu_busy") is not None:
                ginstr = pmc["valu_wave_insts_per_launch"] / (pmc["avg_launch_ms_profiled"] * 1e-3) / 1e9
                valu = {"issue_busy": pmc["valu_busy"], "lane_util": pmc["valu_lane_util"],
                        "useful_lane_throughput": round(pmc["valu_busy"] * pmc["valu_lane_util"], 4),
                        "wave_insts_per_launch": pmc["valu_wave_insts_per_launch"], "wave_insts_per_ray": round(pmc["valu_wave_insts_per_launch"] * n_launch / max(rays_o, 1), 1),
                        "Ginstr_per_s": round(ginstr, 1), "peak_Ginstr_per_s_at_2_cycles": VALU_PEAK_GINSTR,
                        "rate_frac_of_fp32_peak": round(ginstr / VALU_PEAK_GINSTR, 4), "cycles_per_inst": pmc["valu_cycles_per_inst"],
                        "ta_busy": pmc.get("ta_busy"), "lds_bank_conflict_frac_of_lds_cycles": pmc.get("lds_bank_conflict_frac_of_lds_cycles"),
                        "lds_bank_conflict_cycles_frac_of_kernel": pmc.get("lds_bank_conflict_cycles_frac_of_kernel")}
                fr["valu"] = valu["issue_busy"]
                per = pmc.get("valu_wave_insts_per_step_by_kernel") or {}
                if per and pmc.get("clock_GHz_profiled"):
                    # The timed region runs several batches at once (traversal of one beside the shading of another): is the WHOLE JOB
                    # bound by VALU issue?  All VALU wave-instructions a step needs (profiler pass, every kernel of the loop) x the measured
                    # cycles per instruction / (1024 SIMDs x the clock measured in that pass) = the time the VALUs need for a step if they
                    # never idle, against the step time of the timed region.
                    need_ms = sum(per.values()) * pmc["valu_cycles_per_inst"] / (1024.0 * pmc["clock_GHz_profiled"] * 1e9) * 1e3
                    valu["whole_job"] = {"valu_wave_insts_per_step": int(sum(per.values())), "by_kernel": per,
                                         "valu_issue_ms_per_step": round(need_ms, 3), "ms_per_step_timed": round(result["ms_per_step"], 3),
                                         "issue_busy": round(need_ms / result["ms_per_step"], 4),
                                         # (the clock of the timed region itself is not measured: at the 2.4 GHz peak clock the same instructions need less time)
                                         "clock_GHz_profiled": pmc["clock_GHz_profiled"],
                                         "issue_busy_if_timed_region_ran_at_2.4_GHz": round(need_ms * pmc["clock_GHz_profiled"] / 2.4 / result["ms_per_step"], 4),
                                         "def": "sum over the loop's kernels of SQ_INSTS_VALU per step x measured cycles per instruction / (1024 SIMDs x "
                                                "measured clock), divided by the timed ms_per_step (batches overlapped)"}
        # Round 5: WHICH ceiling binds was asked of the kernel itself (profiles/r05_bound_ladder.txt): 64 extra VALU instructions per node visit (+59 % of a
        # visit's 109) cost the step 5.4 %, ONE extra record gather per visit costs it 27 % -- linear from the first one on.  The kernel is bound by the
        # path its record gathers take, L1 (TCP) -> L2 requests, not by VALU issue (whose busy counter reads 0.89 here and 1.28 on k_generate: uncalibrated).
        # fractions.gather = the records the launch gathers from global memory (ALGORITHMIC bytes: 64 B per node visit that the LDS copy of the tree top does not
        # serve, 48 B per primitive test, 40 B per ray -- device counters of the same frames) / its duration, against the rate at which THIS device gathers random
        # 64-byte records from an L2-resident array with the node fetch's own four loads per record (tirt_micro_gather_rate, 2 MB, measured in this run): like
        # against like.  (Until mid round 5 this compared the launch's L1 -> L2 read BYTES with that rate -- but a request on that path is 128 bytes, so a 64-byte
        # record moves 128: the 0.8 that gave was a unit mismatch.  In requests: `l1_l2_requests` below.)  tools/micro/ta_cost.hip: the rate is set per lane and record.
        if achieved > 0 and peak_l2 > 0:
            fr["gather"] = round(achieved / peak_l2, 4)
        known = {k: v for k, v in fr.items() if v is not None}
        # `bound` is the ceiling the LADDER names (an experiment on this kernel, profiles/r05_bound_ladder.txt: built variants, not this run); what THIS run's
        # counters alone would name -- the largest fraction -- is printed beside it: issue busy is an uncalibrated counter (it reads 1.28 on k_generate)
        largest = max(known, key=known.get) if known else None
        bound = "gather" if fr.get("gather") else (largest or "valu")
        if bound == "gather":
            top = {"achieved": round(achieved, 1), "peak": round(peak_l2, 1), "frac": fr["gather"],
                   "unit": "GB/s of gathered records (achieved: algorithmic bytes of the launch's node / primitive / ray records / launch duration; peak: this device's measured "
                           "rate of random 64-byte record gathers from an L2-resident array, 4 x global_load_dwordx4 per record as a node fetch does)"}
        elif bound == "valu" and valu:
            # VALU issue is the bound; an issue slot whose lanes are masked off is not work, so what is counted is LANE-instructions: the wave
            # instructions the kernel issues x the share of their 64 lanes that are active, against the rate at which this instruction mix would
            # leave the VALUs with every lane active and no idle cycle.  frac = issue_busy x lane_util.
            peak_w = valu["Ginstr_per_s"] / max(valu["issue_busy"], 1e-9)
            top = {"achieved": round(valu["Ginstr_per_s"] * 64.0 * valu["lane_util"], 1), "peak": round(peak_w * 64.0, 1),
                   "unit": "G lane-instructions/s (wave64 VALU instructions x active lanes; peak = 64 lanes x the rate at which this instruction mix saturates "
                           "VALU issue: 1024 SIMDs x clock / measured cycles per instruction)",
                   "frac": round(valu["issue_busy"] * valu["lane_util"], 4)}
        elif bound == "l2" and l2:
            top = {"achieved": l2["GBps"], "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": l2["frac"]}
        elif hbm:
            top = {"achieved": hbm["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"]}
        else:       # no counters (--no-traffic, N > 1): the gathered bytes against the ceiling measured for this working set
            top = {"achieved": round(achieved, 1), "peak": round(peak_ws, 1), "unit": "GB/s", "frac": round(achieved / peak_ws, 4) if peak_ws > 0 else None}
        alg_trace = alg_closest + alg_shadow
        result["roofline"] = {
            "bound": bound, "kernel": "k_trace<ordered> (closest-hit + NEE shadow rays; of the camera rays only those their pixels' candidate lists leave over: k_pvb_cand is not in these launches)",
            "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
            "traffic": tr_bytes,
            "fractions": fr, "hbm": hbm, "l2": l2, "valu": valu,
            "bound_source": "the bound ladder (bound_evidence): extra VALU work per node visit is nearly free, one extra record gather costs 27 %; not derived from this run's counters",
            "bound_by_largest_fraction_of_this_run": {"bound": largest, "frac": known.get(largest) if largest else None,
                                                      "lane_instructions_frac_if_valu": (round(valu["issue_busy"] * valu["lane_util"], 4) if valu else None)},
            "frac_def": "algorithmic gathered record bytes per launch / launch duration / the measured L2-resident record-gather rate of this device (`fractions.gather`); "
                        "`fractions.valu` (issue busy, uncalibrated counter) and `valu.useful_lane_throughput` are printed beside it",
            # the same path in REQUESTS: what the launch sends from L1 to L2 (TCP_TCC_READ_REQ; 128 bytes each, calibrated on k_film) against the request rate of the
            # L2-resident record gather (one request per record there: tools/micro/ta_cost.hip under rocprofv3); neighbouring node records share 128-byte lines
            "l1_l2_requests": ({"G_per_s": round(l2["GBps"] / max(pmc.get("l2_bytes_per_request_calibrated_on_k_film", 128.0), 1.0), 2),
                                "peak_G_per_s": round(peak_l2 / 64.0, 2), "frac": round(l2["GBps"] / max(pmc.get("l2_bytes_per_request_calibrated_on_k_film", 128.0), 1.0) / (peak_l2 / 64.0), 4),
                                "bytes_per_request": pmc.get("l2_bytes_per_request_calibrated_on_k_film"),
                                "per_ray": round(pmc["l2_read_bytes_per_launch"] / max(pmc.get("l2_bytes_per_request_calibrated_on_k_film", 128.0), 1.0) * n_launch / max(rays_o, 1), 2)} if (l2 and ok and peak_l2 > 0) else None),
            "bound_evidence": {"ladder": "profiles/r05_bound_ladder.txt", "valu_pad_64_instructions_per_node_visit": "+5.4 % ms/step", "one_extra_record_gather_per_node_visit": "+27.1 % ms/step",
                               "two": "+51.3 %", "four": "+98.9 %", "eight": "+203 %", "cost_model": "tools/micro/ta_cost.hip: a scattered record costs 0.95 ns per CU whatever "
                               "its width, +0.07 ns per further request to the same line; quad-cooperative fetches do not change it (profiles/r05b)"},
            # the same against the guide's issue peak (a wave64 VALU instruction every 2 cycles: MI355X_MICROARCH.md's 157.3 TFLOP/s FP32); this mix
            # (v_fma_mix, v_min3 / v_max3, v_cndmask, v_alignbit) issues at ~4
            "frac_vs_guide_issue_peak": round(valu["rate_frac_of_fp32_peak"] * valu["lane_util"], 4) if valu else None,
            "why_not_hbm": ("north_star's >= 40 % of the HBM roofline does not apply to this kernel at this scene size: what it walks (2.7 MB of nodes + 4.8 MB of "
                            "primitive records) stays in L2 / LDS (hit rate in `l2`), so its HBM-side traffic is the ray and hit streams only -- counter traffic / "
                            "gathered bytes = `hbm_over_gathered` -- and no re-read is wasted; the ceiling it does sit at is the L1 -> L2 request path of its record gathers "
                            "(`bound_evidence`); configs.big_scene_4M_1024x1024_32spp is the same kernel on a tree that does not fit on the die"),
            "hbm_over_gathered": round(tr_bytes / max(gather_bytes / n_launch, 1.0), 4) if tr_bytes else None,
            "hbm_GBps": hbm["GBps"] if hbm else None, "hbm_frac": fr["hbm"], "l2_frac": fr["l2"], "valu_frac": fr["valu"], "gather_frac": fr.get("gather"),
            # the records the launch gathers from global memory against the two ceilings of that access pattern measured in THIS run
            # (random 64-byte records, 4 x dwordx4 per lane: from an array of the traversal data's size, and from 2 MB = L2-resident);
            # both fractions are printed, neither is chosen after the fact
            "gather": {"GBps": round(achieved, 1), "bytes_per_launch": round(gather_bytes / n_launch, 1), "bytes_per_ray": round(gather_bytes / max(rays_o, 1), 1),
                       "peak_working_set_GBps": round(peak_ws, 1), "frac_of_working_set_peak": round(achieved / peak_ws, 4) if peak_ws > 0 else None,
                       "peak_l2_resident_GBps": round(peak_l2, 1), "frac_of_l2_resident_peak": round(achieved / peak_l2, 4) if peak_l2 > 0 else None,
                       "working_set_bytes": int(working_set),
                       "def": "64 B x 4-wide node visits not served from LDS + 48 B x primitive tests + 40 B x rays (ordered-traversal device counters of the "
                              "same frames) / mean k_trace launch duration (HIP events); peaks: tirt_micro_gather_rate"},
            "k_shade": shade_ceilings(pmc.get("k_shade") if ok else None),
            "traffic_detail": pmc, "bvh": info,
            "avg_launch_ms": round(avg_ms, 5), "launches": int(n_launch),
            "rays_traced_by_k_trace_in_these_launches": int(rays_o), "camera_rays_resolved_by_candidate_lists": int(listed),
            "node_visits_per_ray": round(node_visits / max(rays_o, 1), 2),
            "lds_node_visits_per_ray": round(lds_visits / max(rays_o, 1), 2),
            "prim_tests_per_ray": round((co["leaf_closest"] + co["leaf_shadow"]) / max(rays_o, 1), 2),
            "kernel_ms": {"trace_closest": round(t["ms_trace_closest"], 3), "trace_shadow": round(t["ms_trace_shadow"], 3),
                          "shade": round(t["ms_shade"], 3), "render_total": round(t["ms_render"], 3)},
            "wave_diag_ordered": {"node_iters_per_ray": round(co["diag_it_node"] * 64.0 / max(rays_o, 1), 2),
                                  "node_lane_util": round(co["diag_lanes_node"] / max(co["diag_it_node"] * 64.0, 1), 4),
                                  "leaf_iters_per_ray": round(co["diag_it_leaf"] * 64.0 / max(rays_o, 1), 2),
                                  "leaf_lane_util": round(co["diag_lanes_leaf"] / max(co["diag_it_leaf"] * 64.0, 1), 4),
                                  "refills_per_wave_ray": round(co["diag_refills"] * 64.0 / max(rays_o, 1), 3),
                                  # timeline of the (counting) launches: a wave's mean lifetime against the launch it ran in, and the share of the
                                  # waves' lifetimes spent after the ray queue ran dry (each wave only finishing the rays it holds: the tail)
                                  "wave_life_over_launch": round(co["diag_wave_ticks"] * 1.0e-5 / max(co["diag_waves"], 1) /
                                                                 max((co["ms_trace_closest"] + co["ms_trace_shadow"]) / max(co["launches_trace_closest"] + co["launches_trace_shadow"], 1), 1e-9), 4),
                                  "drain_share_of_wave_life": round(co["diag_drain_ticks"] / max(co["diag_wave_ticks"], 1), 4)},
            # SURVEY.md 8d's algorithmic bytes on the REFERENCE's traversal semantics (exhaustive pop counts): what the
            # reference's algorithm would move for these rays -- the ordered traversal moves far less, so this is NOT a
            # fraction of any roofline of this kernel
            "alg_reference_semantics": {
                "bytes_per_launch": round(alg_trace / n_launch, 1),
                # NOT a bandwidth of this kernel: the bytes the REFERENCE'S algorithm would have moved for these rays, divided by the time this
                # kernel takes for them (it visits ~11 times fewer nodes and is L2-resident)
                "reference_bytes_per_second_of_this_kernel_GBps__not_a_bandwidth": round((alg_trace / n_launch) / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None,
                "times_fewer_bytes_gathered_than_the_reference_semantics": round(alg_trace / max(gather_bytes, 1.0), 1),
                "bytes_per_closest_ray": round(alg_closest / max(c["rays_closest"], 1), 1),
                "bytes_per_shadow_ray": round(alg_shadow / max(c["rays_shadow"], 1), 1),
                "n_box_per_closest_ray": round(c["box_closest"] / max(c["rays_closest"], 1), 2),
                "n_leaf_per_closest_ray": round(c["leaf_closest"] / max(c["rays_closest"], 1), 2),
                "whole_job_reference_bytes_per_second_GBps__not_a_bandwidth": round((alg_closest + alg_shadow + 260.0 * c["shaded"] + 24.0 * c["paths"]) /
                                                   max(t["ms_render"], 1e-9) / 1e6, 2)},
        }

    # ---- CPU baseline: oracle on the host cores, bounded sample (rank 0, N = 1 only) ---------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, ex, W, H, build_ms)

    # ---- the other BASELINE configs and the build at 1 M primitives, driver-observable (rank 0, N = 1, after the timed region) ----
    if rank == 0 and world == 1 and not args.no_configs:
        cfgs = {}
        for name in ("config1_cornell_512x512_512spp", "config2_teapot_1024x1024_64spp", "config5_veach_bdpt_512x512_64spp", "spectral_cornell_512x512_64spp",
                     "prism_rainbow_bdpt_spec_512x512_64spp", "big_scene_4M_1024x1024_32spp", "big_scene_8M_1024x1024_32spp"):
            try:
                cfgs[name] = run_config(name, local_rank, args.seed)
            except Exception as exc:        # noqa: BLE001 -- one failing config must not hide the headline line
                cfgs[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            gc.collect()                    # the config's contexts (tens of GB of wavefront state each) go now, not when the collector gets round to their cycles
        cfgs["config3_headline"] = "this line's `value` (%d steps x %d frames of the 100k scene)" % (args.steps, fps)
        if not args.no_traffic:
            cfgs["config5_veach_bdpt_512x512_64spp"]["roofline"] = bdpt_roofline(local_rank)
            for big in BIG_SCENES:
                if "error" not in cfgs[big]:
                    cfgs[big]["roofline"] = big_scene_roofline(cfgs[big], big)
        # LBVH build (a4..a8) + traversal tree at 1 M primitives, second build of each kind (HIP events on the context's stream)
        try:
            big = scenes.synthetic(64, 64, 4, ntri=1000000, spread=0.012, device_id=local_rank)
            big.build_scene(); bctx = big.scene.ctx
            bd = {}
            for tree in (0, 1):
                bctx.set_option("traversal_tree", tree)
                ms = []
                for _ in range(3):
                    bctx.lbvh_build(); ms.append(bctx.stats()["ms_build"])
                bd["with_sah_traversal_tree_ms" if tree else "lbvh_only_ms"] = round(min(ms[1:]), 3)
            bd["prims"] = int(big.scene.primitive_count)
            cfgs["lbvh_build_1M"] = bd
        except Exception as exc:            # noqa: BLE001
            cfgs["lbvh_build_1M"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        cfgs["lbvh_build_100k"] = dict(build_detail, prims=int(ex.scene.primitive_count))
        result["configs"] = cfgs

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
