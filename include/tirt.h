/*
 * tirt.h -- C-ABI of libtirt.so, the MI355X (gfx950) path-tracing core that sits behind
 * the ti-raytrace Scene / LBvh.Bvh / Camera / PT_RGB.PathTrace Python API.
 *
 * The reference has no FFI boundary for this path: every hot function is a @ti.kernel /
 * @ti.func JIT-compiled by Taichi and reached only from Python.  The
 * boundary is therefore defined here; each entry point cites the reference call it
 * replaces.  Plain pointers and sizes only, no torch types.  Host buffers are
 * C-contiguous f32/i32 (numpy); the caller owns them, the library copies on upload and
 * owns all device memory.  One host thread drives one tirt_ctx (one HIP device, one
 * stream).  Every function returns 0 on success, <0 on error (tirt_last_error() gives
 * the message); nothing throws across the boundary.
 *
 * Embedding.  The library itself touches no process state: no environment variable is read or written, no signal handler or
 * thread is installed (librccl is dlopen()ed by tirt_comm_init only).  A context runs up to `overlap_lanes` (default 4) HIP
 * streams side by side; the ROCm runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), so an embedder that
 * wants each lane on a queue of its own sets GPU_MAX_HW_QUEUES >= 5 before the HIP runtime initialises.  The Python package
 * ti_raytrace_amd does exactly that at import (setdefault to 8) and pre-loads PyTorch's bundled libamdhip64.so when torch is
 * installed, so that libtirt.so and torch share one runtime; TIRT_NO_ENV_TUNING=1 switches both off, TIRT_SYSTEM_HIP=1 only the
 * preload (ti_raytrace_amd/__init__.py, _native.py).  libtirt.so's DT_NEEDED entry is libamdhip64.so.7 by SONAME.
 */
#ifndef TIRT_H
#define TIRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tirt_ctx tirt_ctx;

#define TIRT_OK 0
#define TIRT_ERR_HIP (-1)      /* a HIP runtime call failed (no device, OOM, launch error) */
#define TIRT_ERR_ARG (-2)      /* bad argument / call order */
#define TIRT_ERR_BUILD (-3)    /* LBVH build did not complete (refit did not reach the root) */
#define TIRT_ERR_STACK (-4)    /* traversal stack overflow (reference: "overflow, need larger stack") */

/* tirt_pt_rgb_render / tirt_trace_* flags */
#define TIRT_TRAVERSE_ORDERED 0     /* near-first, t-culled traversal (product default)     */
#define TIRT_TRAVERSE_EXHAUSTIVE 1  /* reference visiting rule: no t-culling (Scene.py:702-744) */
#define TIRT_COUNT_NODES 2          /* accumulate N_box / N_leaf pop counts into tirt_stats  */

typedef struct {
    uint64_t rays_closest;   /* closet_hit calls        (integrator/PT_RGB.py:65)  */
    uint64_t rays_shadow;    /* closet_hit_shadow calls (integrator/PT_RGB.py:104) */
    uint64_t box_closest;    /* compact nodes popped by closest-hit rays (TIRT_COUNT_NODES) */
    uint64_t leaf_closest;   /* ... leaves among them */
    uint64_t box_shadow;
    uint64_t leaf_shadow;
    uint64_t shaded;         /* path vertices that ran the BSDF branch */
    uint64_t paths;          /* pixel-samples started */
    uint64_t stack_overflow; /* rays whose traversal overflowed stack_size */
    double ms_build;         /* last tirt_lbvh_build, HIP-event time on the ctx stream */
    double ms_render;        /* sum of tirt_pt_rgb_render calls since reset */
    double ms_trace_closest; /* sum over closest-hit kernel launches since reset */
    double ms_trace_shadow;
    double ms_shade;
    uint64_t launches_trace_closest;
    uint64_t launches_trace_shadow;
    uint64_t launches_shade;
    /* wave-occupancy diagnostics of the traversal kernels (TIRT_COUNT_NODES only): per-wave
     * loop trips and the number of busy lanes summed over those trips (64 = full wave) */
    uint64_t diag_it_node, diag_lanes_node, diag_it_leaf, diag_lanes_leaf, diag_refills, diag_it_outer;
    /* timeline of the traversal waves (TIRT_COUNT_NODES only), 100 MHz ticks: their lifetimes summed, the part of a lifetime after the wave
     * found the ray queue empty (it only finishes the rays it holds), and how many waves ran */
    uint64_t diag_wave_ticks, diag_drain_ticks, diag_waves;
    uint64_t launches_tail;   /* always 0 since round 6 (rounds 4-5, experiments builds: PT_RGB batches whose last bounces ran as one persistent launch); kept for the layout */
} tirt_stats_t;

const char *tirt_last_error(void);
int tirt_version(void);
int tirt_device_count(int *out);

/* one context = one HIP device + one stream */
int tirt_create(int device_id, tirt_ctx **out);
void tirt_destroy(tirt_ctx *ctx);
int tirt_sync(tirt_ctx *ctx);
/* options: "time_kernels" (0/1) -- bracket every trace/shade launch with HIP events on the
 *            ctx stream so that tirt_stats reports per-kernel time (bench/roofline only)
 *            (also confines the batches to one lane so that kernel times are not overlapped)
 *          "overlap_lanes" (1..8, default 4) -- wavefront batches in flight on separate streams
 *          "trace_lds_depth" / "trace_refill_min" / "trace_node_min" / "trace_grid" / "trace_grid_alone" / "trace_slices" /
 *          "shade_grid" -- kernel tuning
 *          "bdpt_bounded" (0/1, default 1) -- BDPT connection rays are cut off at their target distance (same
 *            visibility answers as the full closest-hit query; 0 = reference-style full query, for cross-checks)
 *          "merge_paths" -- consecutive tirt_pt_rgb_render calls over contiguous frames are merged
 *            until this many pixel-samples are pending (default 32 Mi = one full batch; 0 submits every call at once);
 *            every other entry point submits what is pending first
 *          "job_frames" -- hint: frames the whole job will render (0 = unknown, default).  With it (and no explicit batch_paths)
 *            the job is cut into as few wavefront batches as fit 128 Mi pixel-samples each, at least two, run two at a time
 *            (round 5: a 640 Mi-sample job as 5 x 128 Mi instead of 8 x 80 Mi on four lanes: +1.7 %); lane buffers are sized for that and only as many lanes get one (a 512^2 x 8 spp job does
 *            not allocate 32 Mi-path lanes).  Set it before the first render call of the job
 *          "batch_paths" -- pixel-samples kept in flight per wavefront batch (default 32 Mi, or planned from "job_frames";
 *            204 B of HBM each, lanes hold 1.5 x that)
 *          "split_lone_batch" (0 = off, or the number of parts 2..8; default 0 since round 3) -- a context that owns 1/6 or less of the film
 *            (tile_count >= 6) and whose whole job is one batch runs it as that many smaller batches on as many lanes
 *          "path_order_blocks" (0/1, default 1 since round 5) -- the paths of a wavefront batch numbered pixel-block major (64-path chunk = one 8 x 8 pixel block
 *            of one frame, the frames of a block next to each other) instead of frame major: +1.5 % on the headline scene, +5 % on a 4 M-triangle one; same film
 *          "slices_contiguous" (0/1, default 0) -- k_trace's ray-fetch slices as contiguous stretches of the queue (each XCD one region of the film); no gain measured
 *          "wide_collapse" -- EXPERIMENT (cost-optimal grouping into 4-wide nodes): only in a library built with -DTIRT_EXPERIMENTS (`make experiments`); the product
 *            library answers it with an error unless the value means "off".  (Rounds 4-5 also carried "tail_paths" / "tail_bounce" there: the last bounces of a batch as one
 *            persistent launch -- bit-identical, slower at every switch point; tools/exp/patches/r06_persistent_tail_kernel.patch)
 *          "traversal_tree" (0/1, default 1) -- the tree tirt_lbvh_build collapses into the 4-wide traversal nodes: 1 = a binned-SAH
 *            tree over the same primitives built on the device after the LBVH, 0 = the reference's LBVH itself; results are
 *            bit-identical either way (tirt_traversal_tree_download) -- except for rays that lie, to fp32 rounding, IN the plane of a
 *            triangle: there the reference's Moller-Trumbore divides by a determinant of rounding noise and which "hit" such a ray
 *            gets depends on the visiting order (none in any render); takes effect at the next tirt_lbvh_build
 *          "bdpt_mem_budget" -- bytes a BDPT call may take for its batch state even when more is free (0 = off; tests)
 *          "primary_beams" -- (round 5, default 1) camera rays against per-pixel lists of the leaves they can hit first instead of the bounce-0 traversal launch
 *            (csrc/tirt_pvb.hip: the lists are made once per build / camera / film from five probe rays and a walk of the pixel's pyramid; rays that find no hit
 *            on their list are traced the ordinary way; the same hit records bit for bit: +15 % on the headline scene); 0 = off.  "primary_beams_min_frames"
 *            (default 16): batches of fewer frames keep the ordinary launch (making the lists costs 1.2 ms at 1024^2).  "primary_beams_rebuild" (any value):
 *            the lists are forgotten and made again by the next batch that uses them (bench.py times a build inside its clock).  When the memory for the
 *            lists is not to be had the render goes on without them (tirt_primary_beam_stats out[7]).
 *          "bdpt_batch_items" -- (frame, pixel) items per BDPT wavefront batch (default 16 Mi, ~2.2 KB of HBM each: 38 GB; 5 Mi = 12 GB runs config 5 at 2 900 Mrays/s), shared by the batches in flight on render lanes 0 and 1 ("bdpt_lanes", 1..4, default 2: three or four measured no faster)
 *            (config 5, round 3: 4 Mi 2 890, 8 Mi 2 900, 16 Mi 2 995, 32 Mi 2 980 Mrays/s; 256 frames: 8 Mi 2 912, 16 Mi 2 923, 32 Mi 3 032)
 *            A call never sizes its batches beyond what hipMemGetInfo reports free (less 2 GB): next to other users of the device it renders in smaller batches.
 *          "bdpt_state_fill" (diagnostic) -- what a BDPT batch does to its per-item vertex arrays first: 0 = nothing (default: no read reaches a slot
 *            its item has not written), 1 = zeros (rounds 1-3: +3.5 % time on config 5), 2 = 0xFF poison (the parity tests run under it)
 *          (trace_lds_depth is checked against the LDS a block can have on the device) */
int tirt_set_option(tirt_ctx *ctx, const char *name, double value);

/* Scene.setup_data_gpu field uploads (reference Scene.py:299-308).
 * vertex[nv*9] primitive[n*3] material[nm*10] shape[ns*10] light[nl]; light_count is
 * Scene.light_count (may be 0 while nl == 1, Scene.py:253-261); bmin/bmax = scene AABB
 * (LBvh.Bvh.min/max_boundary, accel/LBvh.py:193-194). */
int tirt_scene_upload(tirt_ctx *ctx, const float *vertex, int nv, const int32_t *primitive, int n,
                      const float *material, int nm, const float *shape, int ns,
                      const int32_t *light, int nl, int light_count,
                      const float bmin[3], const float bmax[3]);
/* Re-upload the material rows only (examples edit material_cpu before setup). */
int tirt_material_upload(tirt_ctx *ctx, const float *material, int nm);

/* Texture.setup_data_gpu (texture/Texture.py:38-39): rgb_packed[w*h] 0xRRGGBB, index x*h + y */
int tirt_env_upload(tirt_ctx *ctx, const int32_t *rgb_packed, int w, int h, float power);

/* LBvh.Bvh.setup_data_gpu (accel/LBvh.py:192-226): Morton codes, stable radix sort, Karras
 * topology, leaf boxes, bottom-up refit, DFS flatten -- all on device. */
int tirt_lbvh_build(tirt_ctx *ctx);
/* any of the three may be NULL: morton_sorted[n*2] (code, prim), bvh_node[(2n-1)*11],
 * compact_node[(2n-1)*9] */
int tirt_lbvh_download(tirt_ctx *ctx, int32_t *morton_sorted, float *bvh_node, float *compact_node);
/* The tree the ordered traversal walks (option "traversal_tree" = 1, the default): a binned-SAH binary tree over the
 * same primitives, built on the device after the LBVH, in the layout of compact_node -- rows [(2n-1)*9], pre-order, row =
 * (1 | prim | box) for a leaf, (0 | index of the right child | box) otherwise, left child = row + 1.  No counterpart
 * in the reference: its LBVH (tirt_lbvh_download) still decides every hit -- a candidate is accepted only if the
 * reference's traversal would have visited that leaf (Scene.py:702-744) -- this tree only finds the candidates with fewer
 * node visits.  With traversal_tree = 0 the rows are those of compact_node. */
int tirt_traversal_tree_download(tirt_ctx *ctx, float *rows);
/* unsorted Morton pairs [n*2] as produced by build_morton_3d (accel/LBvh.py:318-336) */
int tirt_morton_download(tirt_ctx *ctx, int32_t *morton_unsorted);

/* Scene.process_normal (Scene.py:754-798); vertex_index[nv] = owning primitive (Scene.py:128) */
int tirt_process_normal(tirt_ctx *ctx, const int32_t *vertex_index);
int tirt_vertex_download(tirt_ctx *ctx, float *vertex);
/* Scene.total_area (Scene.py:747-750) */
int tirt_total_area(tirt_ctx *ctx, float *out);

/* Camera.update field uploads (Camera.py:91-93) + intrinsics (Camera.py:31-34) */
int tirt_camera_set(tirt_ctx *ctx, const float view[16], const float view_inv[16], const float eye[3],
                    float fx, float fy, float cx, float cy);

/* PathTrace.setup_data_cpu (integrator/PT_RGB.py:34-37): hdr + rgb_film, W*H*3 f32 each,
 * index (i*H + j)*3.  This context renders the pixels whose linear index p = i*H + j lies
 * in a tile (p / tile_size) with tile % tile_count == tile_rank (tile_count 1: all).  A tile_size of a multiple of 8 whole
 * columns (8 * H pixels; H a multiple of 8, W * H a multiple of tile_size) lets the device walk a tile in 8 x 8 pixel blocks
 * -- the camera rays of a wave are then a compact bundle; any other tile_size works too. */
int tirt_film_create(tirt_ctx *ctx, int W, int H, int tile_rank, int tile_count, int tile_size);
int tirt_film_clear(tirt_ctx *ctx);

/* PathTrace.render x frame_count (integrator/PT_RGB.py:44-136), frames frame_begin ..
 * frame_begin+frame_count-1 accumulated into hdr as the running mean of :134-136.
 * Asynchronous, and possibly deferred: see option "merge_paths". */
int tirt_pt_rgb_render(tirt_ctx *ctx, uint32_t frame_begin, int frame_count, uint32_t seed,
                       int max_depth, int stack_size, int flags);

/* BDPT.render x frame_count (integrator/BDPT_RGB.py:595-642): eye + light sub-paths, all
 * connections up to MAX_DEPTH 5 with MIS, light-tracing splats; same film, camera and tiling as
 * PT_RGB (with tiles, every context accumulates splats into its full-size film: sum-reduce). */
int tirt_bdpt_rgb_render(tirt_ctx *ctx, uint32_t frame_begin, int frame_count, uint32_t seed);
/* BDPT.render x frame_count of integrator/BDPT_SPEC.py:660-691: the same bidirectional tracer carrying one wavelength per pixel sample
 * (reflectances and emitters through the RGB -> spectrum table, dispersive glass, AddSplat through the CIE observer).  Needs
 * tirt_spectral_upload.  The example that uses it: example/prism_rainbow.py (a laser through a prism). */
int tirt_bdpt_spec_render(tirt_ctx *ctx, uint32_t frame_begin, int frame_count, uint32_t seed);

/* ---- spectral path: integrator/PT_Spec.py (hero-wavelength path tracer) -----------------------------
 * tirt_spec_table_build: spectrum/JakobSpecTable.py:1-439 on the device -- the RGB -> sigmoid-spectrum coefficient table that
 *   Rgb2Spec.load_table reads from spectrum/spec_table (a file the reference repository lacks).  cie_xyz [n*3] and d65 [n] for
 *   360..830 nm in 1 nm steps (n = 471; spectrum/ciexyz31_1.csv, spectrum/Illuminantd65.csv from 360 nm on);
 *   scale_out [res], coeff_out [3*res^3*3] in the order of the table file (res = 64 there).
 * tirt_spectral_upload: what PathTrace.setup_data_cpu / setup_data_gpu place (PT_Spec.py:56-99): the CIE 1931 observer rows
 *   (sensor [n_sensor*3], wavelengths s_min..s_max, step s_range), four tabulated spectra back to back in `spd` (D65 after
 *   normalize_spec, white, red, green: Spectrum.load_table; sizes spd_n, ranges spd_min / spd_max / spd_range), the Rgb2Spec table
 *   (tbl_scale [res], tbl_data [3*res^3*3]), and the sky (Sky.configs [11*9], Sky.radiances [11], Sky.sun_dir; sky/Sky.py:76-172).
 * tirt_pt_spec_render: PathTrace.render x frame_count (PT_Spec.py:181-279; MAX_DEPTH 10 there); film, camera, tiling, deferred
 *   submission and flags as tirt_pt_rgb_render. */
typedef struct {
    const float *sensor; int n_sensor; float s_min, s_max, s_range;
    const float *spd; int spd_n[4]; float spd_min[4], spd_max[4], spd_range[4];
    const float *tbl_scale, *tbl_data; int tbl_res;
    const float *sky_cfg, *sky_rad; float sun_dir[3];
} tirt_spectral_t;
int tirt_spec_table_build(tirt_ctx *ctx, int res, const float *cie_xyz, const float *d65, int n, float *scale_out, float *coeff_out);
int tirt_spectral_upload(tirt_ctx *ctx, const tirt_spectral_t *tables);
int tirt_pt_spec_render(tirt_ctx *ctx, uint32_t frame_begin, int frame_count, uint32_t seed, int max_depth, int stack_size, int flags);

/* UtilsFunc.tone_map(exposure, hdr, rgb_film) (UtilsFunc.py:583-586) */
int tirt_tone_map(tirt_ctx *ctx, float exposure);
/* field.to_numpy(): either pointer may be NULL */
int tirt_film_download(tirt_ctx *ctx, float *hdr, float *rgb);
/* device-to-device copy of hdr into / from a caller-owned device buffer of W*H*3 f32
 * (e.g. a torch tensor that is then reduced over RCCL) */
int tirt_film_export_device(tirt_ctx *ctx, void *dev_dst);
int tirt_film_import_device(tirt_ctx *ctx, const void *dev_src);

/* Scene.closet_hit / closet_hit_shadow on a batch of rays (Scene.py:702-744, 671-699).  The default (ordered) traversal returns the
 * reference's hit bit for bit for every ray but the in-plane rays named under "traversal_tree" above; rays that start more than 8
 * scene extents away are traced without distance culling (from there the reference's own distances are rounding noise), so they
 * are the reference's too (tests/test_gpu_trace.py::test_ordered_equals_exhaustive_on_two_million_stress_rays).
 * rays[nr*6] = origin, direction.  out_hit[nr*13] = t, pos3, gnormal3, normal3, tex3;
 * out_prim[nr]; counts[nr*2] = N_box, N_leaf per ray (NULL unless TIRT_COUNT_NODES). */
int tirt_trace_closest(tirt_ctx *ctx, const float *rays, int nr, int stack_size, int flags,
                       float *out_hit, int32_t *out_prim, int32_t *counts);
int tirt_trace_shadow(tirt_ctx *ctx, const float *rays, int nr, int stack_size, int flags,
                      float *out_t, int32_t *out_prim, int32_t *counts);

/* Multi-GPU without a Python framework in the loop (the reference has nothing here): ONE host thread drives
 * ndev contexts, one per device of the node, each created with tirt_film_create(..., tile_rank = i, tile_count = ndev, ...).
 * tirt_comm_init builds one RCCL communicator over them (librccl is loaded on first use); tirt_film_reduce sums the films
 * -- zero outside a context's own tiles -- onto ctxs[root] with one ncclReduce per device in a group (xGMI), and returns
 * when every stream has finished.  bench.py's one-process-per-GPU runs use tirt_film_export_device / import_device around
 * torch.distributed's RCCL reduce instead. */
int tirt_comm_init(tirt_ctx **ctxs, int ndev);
int tirt_film_reduce(tirt_ctx **ctxs, int ndev, int root);
int tirt_comm_destroy(tirt_ctx **ctxs, int ndev);

/* Measurement helpers for bench.py's roofline object (no counterpart in the reference).
 * tirt_bvh_info: bytes of the traversal data the ordered traversal walks (out[0] = quantised 4-wide nodes,
 *   out[1] = primitive records, out[2] = node count, out[3] = of those kept in LDS by every block).
 * tirt_micro_gather_rate: the ceiling of the access pattern k_trace is bound by, measured on this device now --
 *   every lane of 1536 x 256 threads gathers `iters` random 64-byte records (4 x 16-byte loads) from an array of
 *   `working_set_bytes`; returns the rate in GB/s (best of 3 launches). */
int tirt_bvh_info(tirt_ctx *ctx, uint64_t out[4]);
int tirt_micro_gather_rate(tirt_ctx *ctx, uint64_t working_set_bytes, int iters, double *gbps_out);

/* Diagnostics of the traversal kernel's schedule.  tirt_set_option(ctx, "trace_timeline", k) arms the k-th counting launch (TIRT_COUNT_NODES)
 * from then on (-1 disarms); that launch records, per wave, four 64-bit words: start, the moment the wave found the ray queue empty (0: never),
 * end -- all in ticks of the device's 100 MHz wall clock -- and the wave's hardware id (HW_ID in the low word, XCC_ID in the high word).
 * tirt_trace_timeline copies up to max_waves records to `out` and reports how many the launch had.  (No reference counterpart: bench / tools.) */
int tirt_trace_timeline(tirt_ctx *ctx, uint64_t *out, int max_waves, int *n_waves);

/* Diagnostics of the camera rays' candidate lists (option "primary_beams", csrc/tirt_pvb.hip): out[0] = local pixels that have a list, out[1] = leaves on
 * all lists, out[2] = pixels whose five probe rays all hit, out[3] = camera rays since the lists were made that found no hit on their pixel's list and were
 * traced by k_trace, out[4] = camera rays that went through the lists; since the last tirt_stats_reset: out[5] = list builds (one per change of build /
 * camera / film seen by a batch that uses lists, or after option "primary_beams_rebuild"), out[6] = their device time in nanoseconds (HIP events on the
 * context's stream: probe rays + the walk of the pixels' pyramids), out[7] = builds given up for lack of memory (the camera rays then take the ordinary
 * launch); with option "primary_beams_diag" (slow: an atomic per wave) the list pass counts since the lists were made out[8] = leaf steps of all camera rays,
 * out[9] = rays that took more than one, out[10] = lane slots of the waves' trips (64 x the leaf steps of each wave's slowest ray), out[11] = rays that took more
 * than two.  Waits for pending work.  (No reference counterpart: bench / tests.) */
int tirt_primary_beam_stats(tirt_ctx *ctx, uint64_t out[12]);

/* Fills *out.  Returns TIRT_ERR_STACK (with *out filled in) when stack_overflow > 0: rays dropped subtrees, what was
 * rendered since the last tirt_stats_reset is wrong -- the reference prints "overflow, need larger stack" (Scene.py:741). */
int tirt_stats(tirt_ctx *ctx, tirt_stats_t *out);
int tirt_stats_reset(tirt_ctx *ctx);

/* Device-side evaluation of the shared scalar functions (known-answer tests):
 * fn 0 sin 1 cos 2 exp 3 log 4 pow(x,y) 5 atan2(x,y) 6 acos 7 sqrt 8 x/y */
int tirt_kat_math(tirt_ctx *ctx, int fn, const float *x, const float *y, float *out, int n);
/* which 0 Disney.evaluate_pdf  in: mat10,N3,V3,L3        out: f, pdf
 *       1 Disney.sample        in: mat10,dir3,N3,rnd3    out: dir3
 *       2 Glass.sample         in: mat10,dir3,N3,prob    out: dir3, f_or_b
 *       3 UF.offset_ray        in: p3,n3                 out: p3
 *       4 UF.CosineSampleHemisphere in: u1,u2  out: dir3     5 UF.mapToDisk in: u1,u2 out: r,phi      6 UF.powerHeuristic in: a,b out: w
 *       7 UF.inverse_transform in: dir3,N3 out: dir3         8 UF.srgb_to_lrgb / 9 UF.lrgb_to_srgb / 10 UF.tone_ACES in: c3 out: c3
 *       11 UF.refract in: I3,N3,eta out: R3,suc              12 UF.schlick in: cos,ior    13 UF.GTR2 in: NDotH,a    14 UF.smithG_GGX in: NDotv,alphaG
 *       15 UF.SchlickFresnel in: u                           16 Glass.sample_lambda in: dir3,N3,lambda,prob out: dir3,f_or_b
 *       17 Camera.get_ray_direction in: view_inv16,fx,fy,cx,cy,i,j,jx,jy out: dir3
 *       18 UF.slabs in: o3,d3,min3,max3 out: hit (0/1), hit by the branch-free form of the traversal kernel
 *       (4..18: the helpers behind 0..3 one by one, for tests/test_gpu_kat.py against tests/golden/refkat.npz -- values computed by the
 *       reference's own source text)
 * in: [n*in_stride] out: [n*out_stride] */
int tirt_kat_brdf(tirt_ctx *ctx, int which, const float *in, int in_stride, float *out, int out_stride, int n);
/* The spectral device functions one by one, on the tables of tirt_spectral_upload (tests/test_gpu_spectral.py against tests/golden/refkat_spec.npz --
 * values computed by the reference's own spectrum modules, sky/Sky.py and integrator/PT_Spec.py text).  which:
 *   0 Spectrum.sample (Spectrum.py:44-52) in: k (0 d65 1 white 2 red 3 green), Lambda out: 1      1 HeroSample.sample (:10-16) in: k, Lambda0 out: 4
 *   2 HeroSample.sample_xyz (:18-29) of PathTrace.sample (PT_Spec.py:131-139) in: Lambda0 out: x4,y4,z4
 *   3 Rgb2Spec.fetch (Rgb2Spec.py:101-137) in: rgb3 out: coff3     4 Rgb2Spec.eval (:139-143) in: coff3, Lambda out: 1
 *   5 HeroSample.srgb_to_spec (:46-58) in: srgb3, Lambda0 out: 4    6 HeroSample.sky_sample (:60-71; Sky.get_solar_radiance, sky/Sky.py:232-264) in: theta, gamma, Lambda0 out: 4
 *   7 PathTrace.emission_to_rad (PT_Spec.py:102-109) in: emission3, Lambda out: 4     8 HeroSample.get_extinction_hero (:37-43) in: Lambda0, t out: 4
 *   9 PathTrace.AddSplat (PT_Spec.py:141-158) in: spec4, Lambda0, coff, hdr3 out: hdr3      10 PathTrace.get_spec_power (:111-127) in: mat10, Lambda out: 4
 *  11 HeroSample.get_rnd_hero (:31-35) in: ti.random(), Lambda0 out: index, Lambda */
int tirt_kat_spec(tirt_ctx *ctx, int which, const float *in, int in_stride, float *out, int out_stride, int n);

/* ---- native Wavefront OBJ/MTL ingest (host only; no device, no context) -----------------------------
 * Replaces the reference's use of the third-party PyWavefront 1.3.3 package in Scene.add_obj
 * (Scene.py:66-127: `pywavefront.Wavefront(filename)`, `scene.materials[name].vertices / vertex_format /
 * diffuse / emissive / transparency / optical_density / shininess`): materials in first-appearance order,
 * ONE interleaved float list per material (fan-triangulated faces in file order), vertex format chosen
 * by the material's first face.
 *   params[19] = diffuse rgba, ambient rgba, specular rgba, emissive rgba, transparency, optical_density, shininess
 *   vertex_format: 4 "V3F", 5 "T2F_V3F", 6 "N3F_V3F", 7 "T2F_N3F_V3F", 0 when the material owns no face */
typedef struct tirt_obj tirt_obj;
int tirt_obj_load(const char *path, tirt_obj **out);
void tirt_obj_free(tirt_obj *obj);
int tirt_obj_material_count(const tirt_obj *obj);
int tirt_obj_material_info(const tirt_obj *obj, int index, char *name, int name_cap, double *params, int *vertex_format,
                           int *is_default, long long *n_floats);
int tirt_obj_material_vertices(const tirt_obj *obj, int index, double *out, long long n_floats);

#ifdef __cplusplus
}
#endif
#endif /* TIRT_H */
