// tirt_internal.h -- context layout and host-side helpers shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/tirt.h"
#include "tirt_device.h"

namespace tirt {

void set_error(const std::string &msg);

#define TIRT_HIP(call)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            tirt::set_error(std::string(#call) + ": " + hipGetErrorString(e__));                \
            return TIRT_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

#define TIRT_REQUIRE(cond, msg)                                                                 \
    do {                                                                                        \
        if (!(cond)) { tirt::set_error(msg); return TIRT_ERR_ARG; }                             \
    } while (0)

// One growable device allocation.
struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    int ensure(size_t want)
    {
        if (want <= bytes && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        if (want == 0) want = 16;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { set_error(std::string("hipMalloc: ") + hipGetErrorString(e)); p = nullptr; return TIRT_ERR_HIP; }
        bytes = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T *as() const { return (T *)p; }
};

// Traversal data handed to the trace kernels by value.
// wnode: one 64-byte record per INTERNAL compact node, indexed by its compact index:
//   q0 = (Lmin.x, Lmin.y, Lmin.z, Lmax.x)  q1 = (Lmax.y, Lmax.z, Rmin.x, Rmin.y)
//   q2 = (Rmin.z, Rmax.x, Rmax.y, Rmax.z)  q3 = (bits codeL, bits codeR, 0, 0)
//   code >= 0: compact index of an internal child (box = the reference's box, bit-exact);
//   code <  0: leaf, prim = (~code) & 0x3fffffff, bit 30 of ~code set for shape primitives
//              (box = leaf box inflated by `pad`, only used for culling in ordered mode).
// tri: one 48-byte record per primitive, indexed by primitive id:
//   triangles: (v0.xyz, bits leaf_compact_index) (v1.xyz, 0) (v2.xyz, 0) -- the three positions themselves: the edges
//              E1 = v1 - v0, E2 = v2 - v0 are formed per test as the reference does (Scene.py:608-609), and the exact
//              leaf box (min / max of the three) is at hand for the hit verification of the ordered traversal
//   spheres  : (centre.xyz, bits leaf_compact_index) (radius, 0, 0, 0) (0,0,0,0)
#ifndef TRI_STRIDE_N
#define TRI_STRIDE_N 3
#endif
constexpr int TRI_STRIDE = TRI_STRIDE_N;                 // float4 per primitive record
constexpr int TR_EMPTY = (int)0x80000001;
#ifndef TR_TOP_LEVELS_N
#define TR_TOP_LEVELS_N 5
#endif
constexpr int TR_TOP_LEVELS = TR_TOP_LEVELS_N;           // levels of 4-wide nodes that get a breadth-first slot
// How many tree-top records: a k_trace block is 256 threads (one wave per SIMD) with 16 KB of stacks, and FIVE of them are to fit a CU's
// 160 KB of LDS: five traversal waves per SIMD at 80 VGPRs leave 112 of the SIMD's 512 registers, which is what lets a k_shade wave (96) of
// the batch running next door live beside them.  LDS is handed out in 2 KB-ish granules: 16 KB + 224 x 64 B = 30 KB per block is safely five per CU.
// (Measured with the wave timeline of tools/timeline.py, round 3: the earlier 512-thread blocks with 32 KB + 336 records = 53 KB were believed to fit
// three to a CU and fitted two -- four waves per SIMD, latency-bound at 0.81 VALU issue; three of them (320 records) traverse 13 % faster on
// their own but take 480 registers and shut the shading kernels out, so the whole job got slower.  Five per SIMD + shading beside: +5 % whole job.)
#ifndef TR_TOP_CAP
#define TR_TOP_CAP 224
#endif
constexpr int TR_TOP_FULL = ((1 << (2 * TR_TOP_LEVELS)) - 1) / 3;
constexpr int TR_TOP_SLOTS = TR_TOP_FULL < TR_TOP_CAP ? TR_TOP_FULL : TR_TOP_CAP;   // the first 224 nodes in breadth-first order: four levels and the start of the fifth
// cnode: the 4-wide nodes again, 64 bytes each, box planes quantised on ONE grid over the root box (k_cnodes):
//   plane = grid_min + h * cell with h an fp16 number of cells measured from the CENTRE of the root box (|h| <= 30000:
//   the spacing of fp16 there is 16 cells = 2.7e-4 of the extent, finer towards the centre), min planes rounded down
//   and max planes up by at least one cell, so a cnode box CONTAINS the reference box it stands for.  fp16 because
//   v_fma_mix_f32 converts and multiplies-adds in one instruction: crossing distance of a plane = h * gA + gB, one
//   VALU op (a 16-bit integer grid needs a conversion first: 24 more instructions per node visit of a kernel that is
//   bound by VALU issue).  k_trace<ordered> walks these (4 dwordx4 loads per visit instead of 7, 3.2 MB instead of
//   6.4 MB at 100k triangles).  A conservative box can only add visits; it may,
//   however, reach a leaf the reference does not (the reference tests its exact fp32 boxes with fp32 arithmetic
//   and a grazing ray can fail an ancestor's box yet pass the triangle test), so a candidate hit is ACCEPTED only
//   after the reference's own condition is re-established: `slabs` on the leaf's exact box -- which implies every
//   ancestor's, `slabs` being monotone in the plane positions --, else `slabs` on every proper ancestor
//   (compact rows, cparent chain; rare).  The closest accepted hit is therefore the reference's, bit for bit.
//   dword 3c+a (c = child slot 0..3, a = axis): half(min plane) | half(max plane) << 16;  dwords 12..15: child codes as
//   in qnode (TR_EMPTY slots hold an inverted box, which no ray passes).
constexpr float TR_GRID_HALF = 30000.0f;      // the padded root box spans cells -30000 .. +30000
constexpr unsigned TR_H_POS = 0x7b53u, TR_H_NEG = 0xfb53u;      // fp16 +-60000
#ifndef TR_BLOCK_SIZE
#define TR_BLOCK_SIZE 256
#endif
constexpr int TR_BLOCK = TR_BLOCK_SIZE;                  // threads per persistent k_trace block
constexpr int TIRT_MAX_DEVICES = 64;
// dynamic LDS of a k_trace block: `depth` stack entries per lane + the LDS copy of the tree top (64-byte records)
#ifdef TR_PADG
constexpr size_t TR_LADDER_LDS = 4 * TR_BLOCK;           // bound-ladder builds only (tirt_render.hip, TR_LADDER_PADS): where the dummy gathers land
#else
constexpr size_t TR_LADDER_LDS = 0;
#endif
inline size_t trace_lds_bytes(int depth) { return sizeof(int) * (size_t)depth * TR_BLOCK + (size_t)TR_TOP_SLOTS * 64 + TR_LADDER_LDS; }
struct BvhView {
    const float4 *wnode;
    const float4 *tri;
    const uint4 *cnode;           // quantised 4-wide nodes in breadth-first order: the first TR_TOP_SLOTS are copied to LDS by k_trace
    int top_count;                // min(number of nodes, TR_TOP_SLOTS)
    const float *compact;         // reference compact_node rows [N*9] (exact boxes: hit verification)
    const int *cparent;           // compact index of the parent of compact node i (-1 for the root)
    float grid_min[3], cell[3], inv_cell[3], inv_extent[3];
    float root_min[3], root_max[3];
    int root_code;                // two-child layout: compact index 0, or the leaf code of a one-primitive scene
    int root_qcode;               // 4-wide layout: node 0, or the same leaf code
    // Analytic spheres whose slot in the 4-wide nodes is their own (padded) box instead of the whole grid (TIRT_SPHERE_PAD, tirt_sah.hip):
    // a ray that starts too far away for that box to be safe (cull_far < 0, k_trace) gets these leaves pushed at its start.
    int far_qcode;                // where a far-origin ray starts: root_qcode, or the first of the chain nodes that hold the padded analytic spheres beside the root (lbvh_build)
};

// ---- what k_trace (ordered and exhaustive), k_pvb_beam and k_pvb_cand must agree on to the last bit: ONE definition each ----
constexpr float TR_FAR_RHO = 8.0f;            // ordered traversal: rays starting further than this many root-box extents from the grid do not cull by distance (and a camera out there gets no candidate lists)
// Conservative margin of the quantised boxes, in cells: 0.25 + 0.25 per root-box extent between the origin and the grid (largest axis).  It has to cover (i) the
// rounding of q * gA + gB (<= 0.016 cells per extent of distance) and (ii) what the reference's primitive tests accept outside a leaf box: Moller-Trumbore works on
// o - v0 and is off by ~5e-7 of the origin's distance IN EVERY DIRECTION (0.03 cells per extent).  k_pvb_beam lists leaves for k_trace's walk and adds TR_MARGIN_BEAM_EXTRA.
constexpr float TR_MARGIN_CELLS = 0.25f, TR_MARGIN_PER_RHO = 0.25f, TR_MARGIN_BEAM_EXTRA = 0.05f;
TD float trace_origin_rho(const BvhView &b, float ox, float oy, float oz)
{ return maxf(maxf(absf(b.grid_min[0] - ox) * b.inv_extent[0], absf(b.grid_min[1] - oy) * b.inv_extent[1]), absf(b.grid_min[2] - oz) * b.inv_extent[2]); }
TD float trace_margin_cells(float rho) { return TR_MARGIN_CELLS + TR_MARGIN_PER_RHO * rho; }

// One leaf of the traversal tree against one ray: the reference's primitive test (Scene.py:529-638 intersect_prim: Moller-Trumbore on E1 = v1 - v0, E2 = v2 - v0, or
// the analytic sphere), its acceptance rule `0 < t < hit_t` (Scene.py:702-744) with the equal-distance rule of the reference's visiting order (the candidate with the
// larger compact-node index wins), and -- VERIFY, the ordered walk and the candidate lists -- the proof that the reference would have REACHED this leaf: `slabs` on the
// leaf's exact box (it implies every ancestor's: `slabs` is monotone in the plane positions), else `slabs` on every proper ancestor (compact_node rows along cparent).
// `code` = ~(child code of the leaf) = record slot | shape << 30.  Returns whether the hit (hit_t .. hit_leaf) was replaced.  Used by k_trace and k_pvb_cand: one
// leaf step, so the two cannot drift apart (they did not in round 5, but only the on / off tests said so).
template <bool VERIFY>
TD bool trace_leaf_step(const BvhView &b, const RayCtx &r, const bool par, const int code, float &hit_t, float &hit_u, float &hit_v, int &hit_prim, int &hit_leaf)
{
    const float4 *tp = b.tri + (size_t)(code & 0x3fffffff) * TRI_STRIDE;          // records in the traversal tree's leaf order
    // (the records themselves must NOT be loaded non-temporally: -25 %, profiles/r05m -- their residency in L2 is what the kernel lives on)
    const float4 ta = tp[0], tb = tp[1], tc = tp[2];
    int prim = __float_as_int(tc.w);                                              // the primitive id rides in the last word
    const v3 o = V(r.ox, r.oy, r.oz), d = V(r.dx, r.dy, r.dz);
    const bool is_tri = ((code >> 30) & 1) == 0;
    const v3 pa = V(ta.x, ta.y, ta.z), pb = V(tb.x, tb.y, tb.z), pc = V(tc.x, tc.y, tc.z);
    float t, u, v;
    bool sph = false;
    if (is_tri) {
        t = intersect_tri_packed(o, d, pa, pb - pa, pc - pa, u, v);          // E1 = v2 - v1, E2 = v3 - v1 (Scene.py:608-609)
    } else {
        // Analytic sphere (Scene.py:565-596).  Its root is t = (-b - sqrt(b^2 - 4ac)) / 2 / a with b = -2 (d . oc), a = d . d > 0: for
        // d . oc <= 0 the numerator is a non-positive number minus a square root -- t <= 0, or NaN -- and such a t is never a
        // candidate (0 < t < hit_t).  Exactly so in fp32 (signs, no rounding involved), so the two square roots and two divisions
        // (~80 instructions, which the whole wave would issue for one lane) are only run for rays that head towards the centre.
        u = 0.0f; v = 0.0f; t = INF_VALUE;
        if ((int)tb.y == SHAPE_SPHERE) { const v3 oc = pa - o; sph = dot(d, oc) > 0.0f; }
    }
    if (__builtin_amdgcn_ballot_w64(sph) != 0ull) { if (sph) { float cc; t = intersect_sphere(o, d, pa, tb.x, cc); } }
    const int leaf = __float_as_int(ta.w);
    // reference: accept iff 0 < t < hit_t (so never t >= INF_VALUE); equal-t candidates: see above
    bool cand = (t > 0.0f) & ((t < hit_t) | ((t == hit_t) & (hit_leaf >= 0) & (leaf > hit_leaf)));
    if (VERIFY && cand) {
        // the leaf's exact box is the min / max of the three positions just loaded (accel/LBvh.py:397-426; spheres: centre -+ r)
        v3 bmn, bmx;
        if (is_tri) {
            // v_min3_f32 / v_max3_f32 (6 instructions, not 24 compare + select): positions are never NaN, and which of
            // -0 / +0 comes out of a tie changes no comparison of `slabs`
            bmn = V(__builtin_fminf(__builtin_fminf(pa.x, pb.x), pc.x), __builtin_fminf(__builtin_fminf(pa.y, pb.y), pc.y), __builtin_fminf(__builtin_fminf(pa.z, pb.z), pc.z));
            bmx = V(__builtin_fmaxf(__builtin_fmaxf(pa.x, pb.x), pc.x), __builtin_fmaxf(__builtin_fmaxf(pa.y, pb.y), pc.y), __builtin_fmaxf(__builtin_fmaxf(pa.z, pb.z), pc.z));
        } else {
            bmn = V(pa.x - tb.x, pa.y - tb.x, pa.z - tb.x); bmx = V(pa.x + tb.x, pa.y + tb.x, pa.z + tb.x);
        }
        float tn_;
        const int inside = par ? slabs(r, bmn.x, bmn.y, bmn.z, bmx.x, bmx.y, bmx.z, tn_) : slabs_fast(r, bmn.x, bmn.y, bmn.z, bmx.x, bmx.y, bmx.z, tn_);
        if (!inside) {
            // a hit within rounding distance of the leaf box's boundary: the ancestors are asked one by one
            for (int an = b.cparent[leaf]; an >= 0; an = b.cparent[an]) {
                const float *ab = b.compact + (size_t)an * CPN_VEC + 2;
                if (!slabs(r, ab[0], ab[1], ab[2], ab[3], ab[4], ab[5], tn_)) { cand = false; break; }
            }
            // the primitive id again, from the leaf's reference row (UtilsFunc.py:get_compact_node_prim): the same number that came with the
            // primitive record -- read here so that NOTHING of that record has to survive the walk above.  ROCm 7.2's register allocator lets
            // the walk's row loads (global_load_dwordx4 v[8:11]) land on the register that holds the record's last word while it is still
            // needed below (tools/dbg/prim_clobber.sh shows the ISA; 156 of 15 000 box-grazing rays on the Cornell box then kept the PREVIOUS
            // hit's primitive id, tests/test_gpu_trace.py::test_quantised_nodes_on_grazing_rays).  Round 3 pinned the value with an empty asm.
            prim = (int)b.compact[(size_t)leaf * CPN_VEC + 1];
        }
    }
    if (cand) { hit_t = t; hit_u = u; hit_v = v; hit_prim = prim; hit_leaf = leaf; }
    return cand;
}

// Wavefront state, struct-of-arrays in HBM.  Live paths are kept DENSE: every bounce the shade
// kernel writes the surviving paths' state into the other PathSoA at consecutive indices
// (wave-ballot compaction), so the trace and shade kernels stream their inputs with fully
// coalesced 4-byte-per-lane accesses instead of gathering through a slot list.
struct PathSoA {
    float *ox, *oy, *oz, *dx, *dy, *dz;          // current ray
    float *tr, *tg, *tb;                         // throughput
    float *rr, *rg, *rb;                         // radiance so far
    float *brdf_pdf; uint32_t *flags;           // bit0 perfect_spec
    int *slot;                                   // path id = frame_in_batch * P + local pixel
};
struct PathState {
    PathSoA st[2];                               // ping-pong: bounce b reads st[b&1], writes st[(b+1)&1]
    float4 *hit;                                 // closest hit of ray q: (t, u, v, bits prim) -- one 16-byte store per ray
    float *sox, *soy, *soz, *sdx, *sdy, *sdz;    // shadow rays (dense, origin on the light)
    float *scr, *scg, *scb; int *sprim;         // contribution if sprim is the closest hit
    float *sdist;                               // distance light point -> shaded point
    int *sdst;                                   // where the contribution goes: >= 0 index in the next
                                                 // PathSoA (path continues), < 0: ~slot in the final radiance
    float *fr, *fg, *fb;                         // final radiance per path id (read by k_film)
    float *scw, *fw;                             // PT_Spec: fourth hero-wavelength component of the shadow-ray contribution / of the final radiance
};

// pixel-tile shard of this context: local pixel k -> linear pixel index (p = i * H + j: a tile is a run of whole or partial columns).
// `blocked` (tiles of a multiple of 8 whole columns, H a multiple of 8, no partial tile): the local order inside a tile walks
// 8 x 8 pixel blocks, so that the 64 camera rays of a wave form a compact bundle instead of a 1 x 64 strip -- the same rays, 5.5 %
// faster through the tree (tools/exp/primary_order.py); which pixels a tile owns does not change.
// `F` > 0: the paths of a wavefront batch of F frames are numbered pixel-block major -- 64-path chunk c = (block of 64 local pixels c / F, frame c % F)
// -- instead of frame major (F == 0: path s = frame * P + pixel): every contiguous stretch of the ray queue then belongs to ONE region of the film in
// all its frames, which is what lets k_trace hand each XCD (its own L2) the rays of one part of the scene (slices_contiguous).  Needs P % 64 == 0.
struct TileMap { int tile_rank, tile_count, tile_size, H, blocked, F; };
TD void slot_to_frame_pixel(const TileMap &m, int P, int slot, int &f, int &k)
{
    if (m.F > 0) { const int c = slot >> 6, kb = c / m.F; f = c - kb * m.F; k = (kb << 6) | (slot & 63); }
    else { f = slot / P; k = slot - f * P; }
}
TD int frame_pixel_to_slot(const TileMap &m, int P, int f, int k)
{ return m.F > 0 ? ((((k >> 6) * m.F + f) << 6) | (k & 63)) : f * P + k; }
// (Measured and dropped in round 5: a wave holding 64 / G pixels in G frames instead of 64 pixels of one frame -- no gain for G <= 8, slower beyond -- and the 8 x 8 blocks of
// 4 .. 128 neighbouring tiles walked row by row, so that what is in flight is a compact patch of the film: no gain, and its index arithmetic cost every kernel 1 %.)
TD int local_to_pixel(const TileMap &m, int k)
{
    int lt = k / m.tile_size, within = k - lt * m.tile_size;
    if (m.blocked) {
        const int rows = m.H >> 3;                       // 8 x 8 blocks per column group
        const int b = within >> 6, l = within & 63;
        const int bc = b / rows, bj = b - bc * rows;
        within = ((bc << 3) + (l >> 3)) * m.H + (bj << 3) + (l & 7);
    }
    return (lt * m.tile_count + m.tile_rank) * m.tile_size + within;
}

struct DevCounters {          // lives in device memory; accumulated by the kernels
    unsigned long long rays_closest, rays_shadow, box_closest, leaf_closest, box_shadow, leaf_shadow;
    unsigned long long shaded, paths, stack_overflow;
    // wave-occupancy diagnostics of k_trace (TIRT_COUNT_NODES only): loop trips and busy lanes
    unsigned long long it_node, lanes_node, it_leaf, lanes_leaf, refills, it_outer;
    // timeline of k_trace's waves in 100 MHz ticks (TIRT_COUNT_NODES only): lifetimes summed, the part of them after the wave found the ray
    // queue empty (the drain) and the number of waves: against the launches' own time these say how much of a launch is its tail
    unsigned long long wave_ticks, drain_ticks, waves;
};

// One in-flight wavefront batch: its own stream, path state, queues and counters.  The lanes
// alternate, so that the latency-bound tail of one batch (a few long rays per bounce) overlaps
// with the bulk of the next one.
#define TIRT_MAX_LANES 8
struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t film_done = nullptr; bool film_recorded = false;
    size_t path_capacity = 0;
    DevBuf path_mem, counters_mem, spill;
    PathState ps;
};

}  // namespace tirt

struct tirt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // scene (Scene.py fields)
    int nv = 0, n = 0, nm = 0, ns = 0, nl = 0, light_count = 0;
    float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
    tirt::DevBuf vertex, primitive, material, shape, light, env, mat_lrgb, shade_rec;
    bool shade_rec_valid = false;                  // shading records follow vertex / primitive / shape uploads and process_normal
    int env_w = 0, env_h = 0; float env_power = 0.0f;

    // LBVH (accel/LBvh.py fields)
    bool built = false;
    tirt::DevBuf morton_unsorted;                 // int2[n] (code, prim) before the sort
    tirt::DevBuf keys_a, keys_b, vals_a, vals_b;  // radix sort ping-pong
    tirt::DevBuf hist;                            // radix digit histograms
    tirt::DevBuf morton_sorted;                   // int2[n]
    tirt::DevBuf bvh_node, compact;               // f32 [N*11], [N*9]
    tirt::DevBuf parent, flag, subtree, build_status, leaf_compact;
    tirt::DevBuf wnode, tri, prim_slot;           // traversal layout; prim_slot[prim] = index of its record in `tri` (leaf order of the traversal tree)
    tirt::DevBuf cnode, cparent, csize, wide_queue, wide_levels;    // quantised 4-wide nodes (ordered traversal) + parent chain of the compact nodes + build scratch
    int wide_nodes = 0;                            // number of 4-wide nodes
    tirt::DevBuf sah_compact, sah_csize, sah_parent, wide_dp, sah_box, sah_idx, sah_tasks, sah_counts;   // traversal tree (tirt_sah.hip): `compact`-layout rows + subtree sizes, build scratch
    int use_sah = 1, sah_levels = 0, built_sah = 0;
    std::vector<int> sphere_prims;                  // primitive ids of the analytic spheres (tirt_scene_upload)
    tirt::DevBuf timeline; int timeline_arm = -1, timeline_waves = 0;      // diagnostics: option "trace_timeline", tirt_trace_timeline
    int n_far_nodes = 0;                      // chain nodes behind the wide tree (cnode[wide_nodes ...]): the entry of far-origin rays when the spheres' slots carry padded boxes (lbvh_build)
    int wide_dp_on = 0;                            // option "wide_collapse": 0 = greedy by surface area (default), 1 = cost-optimal grouping of the binary tree into 4-wide nodes (dynamic programme; 1-9 % fewer visits, same rays/s)               // option "traversal_tree": 1 = binned-SAH tree (default), 0 = the reference's LBVH
    float grid_min[3] = {0, 0, 0}, grid_cell[3] = {1, 1, 1}, grid_inv_cell[3] = {1, 1, 1}, grid_inv_extent[3] = {1, 1, 1};
    size_t lds_optin = 65536;                      // hipDeviceAttributeMaxSharedMemoryPerBlock (opt-in) of this device
    float root_min[3], root_max[3]; int root_code = 0;

    // camera
    tirt::CameraView cam; float view[16]; bool cam_set = false;

    // film
    int W = 0, H = 0, tile_rank = 0, tile_count = 1, tile_size = 4096, tile_blocked = 0;
    long npix_local = 0;
    tirt::DevBuf hdr, rgb;

    // wavefront state
    tirt::Lane lanes[TIRT_MAX_LANES];
    int n_lanes = 4;                               // option "overlap_lanes" (1..TIRT_MAX_LANES)
    unsigned lane_cursor = 0;
    hipEvent_t ev_main = nullptr;
    hipEvent_t last_film = nullptr;               // film_done of the most recent batch (any lane)
    size_t batch_paths = (size_t)32 << 20;         // option "batch_paths"
    // deferred submission: consecutive tirt_pt_rgb_render calls over contiguous frames are merged until
    // merge_paths pixel-samples are pending (small calls -- one frame at a time, or the 1/N-size shards
    // of a multi-GPU job -- then run as one efficient batch); any other API call flushes first
    size_t merge_paths = (size_t)32 << 20;         // option "merge_paths" (0 = submit every call at once)
    unsigned batches_since_sync = 0;               // wavefront batches submitted since the last sync_all
    int split_lone = 0;                            // option "split_lone_batch": a job that is one batch runs as N parts on N lanes (off: with five traversal waves per SIMD and
                                                   // the shading beside them a lone batch fills the GPU better than its halves: 3.15 against 3.27 ms per step at 8 emulated ranks)
    bool grid_user = false;                        // trace_grid / shade_grid were set through tirt_set_option
    int plan_nb = 0, plan_lanes = 0;              // options "plan_batches" / "plan_lanes": a hinted job as this many batches, this many at a time (0: plan_batches' own rule)
    bool batch_user = false, merge_user = false;   // batch_paths / merge_paths were set through tirt_set_option: no automatic sizing
    long job_frames = 0;                           // option "job_frames": expected frames of the whole job (0 = unknown); bounds the head-room
    struct { bool valid = false; uint32_t begin = 0; int count = 0; uint32_t seed = 0; int max_depth = 0, stack_size = 0, flags = 0; bool spectral = false; } pend;
    // traversal tunables (options "trace_lds_depth", "trace_refill_min", "trace_node_min", "trace_grid",
    // "trace_slices" = number of ray-fetch cursors, "shade_grid" = persistent blocks of k_shade)
    int tr_lds_depth = 16, tr_refill_min = 18, tr_node_min = 38, tr_grid = 1280, tr_slice_log2 = 5, sh_grid = 1024;
    int path_order_blocks = 1, slices_contiguous = 0;      // options "path_order_blocks" (TileMap::F) and "slices_contiguous" (k_trace's ray-fetch slices: contiguous ranges of the queue instead of interleaved chunks)
    int tr_grid_alone = 1280;                     // "trace_grid_alone" / "trace_grid": persistent k_trace blocks of a batch submitted to an idle / a busy GPU. Both five per CU
                                                  // (tirt_create scales them by the device's CU count): blocks of the next batch's launch move in as this one's drain
    tirt::DevBuf counters_mem, spill;            // used by the batch trace entry points (main stream)
    int cu_count = 256;

    // BDPT_RGB: persistent per-pixel vertex arrays + per-frame radiance (splat target)
    tirt::DevBuf bdpt_px;                         // bdpt_px: per-pixel memory of the eye vertices' `delta` fields (what persists from frame to frame)
    // wavefront batch state (vertex arrays per (frame, pixel), step state, rays, hits, queue indices, counters, per-frame radiance) of the
    // two BDPT lanes: consecutive batches alternate between them, on the streams of render lanes 0 and 1, so that the traversal
    // launches of one batch (VALU-bound) run next to the vertex / connection kernels of the other (HBM-bound)
    struct BdLane { tirt::DevBuf items, state, rays, hits, qidx, ctr, rad; hipEvent_t delta_done = nullptr, film_done = nullptr; } bd[4];
    int bdpt_lanes = 2;                           // option "bdpt_lanes": BDPT batches in flight (1..4, not more than overlap_lanes)
    int bdpt_state_fill = 0;                       // option "bdpt_state_fill" (diagnostic): 0 = vertex arrays not cleared per batch, 1 = zeros, 2 = 0xFF poison
    size_t bdpt_batch_items = (size_t)16 << 20;   // option "bdpt_batch_items": (frame, pixel) items per wavefront batch
    size_t bdpt_mem_budget = 0;                    // option "bdpt_mem_budget" (bytes, 0 = off): upper bound on what a BDPT call may take for its batch state, as if the device had only that much free (tests)
    int bdpt_stack = 64;                           // option "bdpt_stack_size": traversal stack entries of BDPT's rays (BDPT.__init__'s stack_size; LDS part + paged spill)
    int bdpt_bounded = 1;                          // option "bdpt_bounded": connection rays stop at their target distance

    // primary visibility through pixel beams (tirt_pvb.hip): per local pixel the leaves its camera rays can hit first, made once per (build, camera, film)
    struct PvbKey { tirt::CameraView cam; unsigned long long build; int W, H, tile_rank, tile_count, tile_size, tile_blocked, P; } pvb_key;
    bool pvb_valid = false; int primary_beams = 1, primary_beams_min_frames = 16;      // options "primary_beams" (0 = off) and "primary_beams_min_frames" (batches of fewer frames trace their camera rays the ordinary way)
    unsigned long long build_serial = 0;          // counts lbvh_build calls
    // two sets of lists: a rebuild (camera move) writes the set the batches in flight are NOT reading, so it waits only for the batches of the camera before last
    // (one set made every camera move wait for all earlier batches to drain); `busy` = film_done of the last batch that read the set
    struct PvbSet { tirt::DevBuf count, cand, bound; hipEvent_t busy = nullptr; } pvb_set[2];
    int pvb_cur = 0;                              // the set pvb_key describes
    int pvb_diag = 0;                             // option "primary_beams_diag": k_pvb_cand counts its leaf steps (one atomic per wave: slow) -- tirt_primary_beam_stats out[8..11]
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pvb_ev;      // HIP events around every list build since the last tirt_stats_reset (read by tirt_primary_beam_stats)
    unsigned long long pvb_builds = 0, pvb_build_ns = 0, pvb_skipped = 0;      // list builds since the reset, their device time, builds given up for lack of memory
    tirt::DevBuf pvb_stat, pvb_tmp;               // (pvb_tmp: the probe scratch where the runtime has no stream-ordered allocation)

    // PT_Spec tables (tirt_spectral_upload): CIE observer, spectra, Rgb2Spec table, sky configuration -- one buffer, views in spec_host
    tirt::DevBuf spec_dev;                      // the SpecView again, in device memory (BDPT_SPEC)
    tirt::DevBuf spec_mem; bool spec_set = false; void *spec_view = nullptr;      // spec_view: a heap tirt::SpecView (tirt_spectral.h) with device pointers

    // batch trace scratch
    tirt::DevBuf tr_rays, tr_out, tr_prim, tr_counts;

    // RCCL communicator of tirt_comm_init (single-process multi-GPU film reduce; opaque ncclComm_t)
    void *comm = nullptr; int comm_rank = 0, comm_size = 0;

    // stats
    tirt::DevBuf dev_counters;                    // DevCounters
    double ms_build = 0, ms_render = 0, ms_trace_closest = 0, ms_trace_shadow = 0, ms_shade = 0;
    uint64_t launches_trace_closest = 0, launches_trace_shadow = 0, launches_shade = 0;
    bool time_kernels = false;                    // per-kernel HIP events (bench only)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
};

namespace tirt {
SceneView scene_view(const tirt_ctx *c);
BvhView bvh_view(const tirt_ctx *c);
int lbvh_build(tirt_ctx *c);
// Pixel-samples per wavefront batch and the lanes they run on.  Without a hint: batch_paths (32 Mi) on all lanes.  With the
// "job_frames" hint and no explicit batch_paths, the job is cut into as few batches as fit 128 Mi paths each, at least two (two
// batches of the largest size overlap each other's per-bounce tails as well as four smaller ones, and every launch is
// fuller), two at a time beyond that.  Measured on one GPU with a
// 256 Mi-path job: 8 x 32 Mi 3 965, 4 x 64 Mi 4 037, 2 x 128 Mi 4 071 Mrays/s; rank 0's share of a 2 / 4 / 8-GPU job is
// best as two batches as well (2 x 64, 2 x 32, 2 x 16 Mi).  A 640 Mi-path job (round 5, profiles/r05ad_*): 8 x 80 Mi on four lanes
// (the rule until then: a multiple of the lane count) 4 645, 5 x 128 Mi or 6 x 107 Mi two or three at a time 4 700 Mrays/s.
struct BatchPlan { size_t batch; int lanes; };
inline BatchPlan plan_batches(const tirt_ctx *c)
{
    BatchPlan p = {c->batch_paths, c->n_lanes};
    if (!c->batch_user && c->job_frames > 0 && c->npix_local > 0) {
        const size_t P = (size_t)c->npix_local, J = (size_t)c->job_frames * P, MAXB = (size_t)128 << 20;
        if (J >= ((size_t)24 << 20)) {
            const size_t nb_min = (J + MAXB - 1) / MAXB, L = (size_t)(c->n_lanes > 0 ? c->n_lanes : 1);
            size_t nb = nb_min <= 2 ? 2 : nb_min;
            if (c->plan_nb > 0 && (size_t)c->plan_nb >= nb_min) nb = (size_t)c->plan_nb;
            size_t frames = ((size_t)c->job_frames + nb - 1) / nb;
            p.batch = frames * P;
            p.lanes = (int)(L < 2 ? L : 2);
            if (c->plan_lanes > 0) p.lanes = (int)(L < (size_t)c->plan_lanes ? L : (size_t)c->plan_lanes);
        }
    }
    return p;
}
inline size_t effective_batch_paths(const tirt_ctx *c) { return plan_batches(c).batch; }
inline size_t effective_merge_paths(const tirt_ctx *c) { return (c->merge_user || c->merge_paths == 0) ? c->merge_paths : effective_batch_paths(c); }
int sah_build(tirt_ctx *c, const int *sorted_prims, float sphere_pad_abs);      // tirt_sah.hip
// Box of an analytic sphere in the TRAVERSAL tree: centre -+ (r + pad).  The reference's sphere test (Scene.py:565-596) answers from
// dis_cp = sqrt(|oc|^2 - (d.oc)^2) < r, a difference of squares of the origin's distance: its error is ~2.4e-7 |oc|^2 / r, so a ray may be
// given a "hit" although it passes the sphere at r + that.  pad = 1e-3 r + pad_abs / r with pad_abs = 2.4e-7 R^2, R = 16 scene diagonals: safe
// for every origin within TR_FAR_RHO root-box extents of the grid; rays from further away have the spheres pushed explicitly (BvhView).
inline float sphere_pad_abs(float diag) { const float R = 16.0f * diag; return 2.4e-7f * R * R; }
TD float sphere_pad(float r, float pad_abs) { return 1.0e-3f * r + pad_abs / (r > 1.0e-20f ? r : 1.0e-20f); }
#ifdef TIRT_EXPERIMENTS
int exp_wide_from_tree(tirt_ctx *c, const float *compact_host, const int *csize_host);     // tools/exp/sah_tree.py
#endif
int launch_trace_batch(tirt_ctx *c, const float *rays, int nr, int stack_size, int flags, bool shadow,
                       float *out_f, int32_t *out_prim, int32_t *counts);
struct SpecView;
int pt_render(tirt_ctx *c, uint32_t frame_begin, int frame_count, uint32_t seed, int max_depth, int stack_size, int flags, const SpecView *spec = nullptr);   // spec != nullptr: PT_Spec
int bdpt_render(tirt_ctx *c, uint32_t frame_begin, int frame_count, uint32_t seed, bool spectral = false);      // spectral: BDPT_SPEC
int trace_arrays(tirt_ctx *c, const float *ox, const float *oy, const float *oz, const float *dx, const float *dy, const float *dz,
                 int count, const int *count_ptr, float4 *hit, const int *expect, const float *bound, bool count_rays, int lane = -1,
                 const float4 *ray4 = nullptr, bool query = false, const int *ray_index = nullptr);      // (ray_index: TraceArgs) ray4: the rays as 32-byte records (TraceArgs::ray4); query: bounded queries whose expect / bound ride in the records
int pvb_prepare(tirt_ctx *c);                          // tirt_pvb.hip
void pvb_launch_cand(tirt_ctx *c, hipStream_t st, const BvhView &bv, const float *dx, const float *dy, const float *dz, const TileMap &tm, int P, int S,
                     float4 *hit, int *fb_count, int *fb_slot, float *fb_dx, float *fb_dy, float *fb_dz, DevCounters *ctr);
void pvb_launch_scatter(hipStream_t st, const int *fb_count, const int *fb_slot, const float4 *fb_hit, float4 *hit);
int trace_arrays_prepare(tirt_ctx *c, int lane);      // allocates what trace_arrays needs on that lane (stack spill, fetch cursors)
int ensure_counters(tirt_ctx *c);
int ensure_shade_records(tirt_ctx *c);
int sync_all(tirt_ctx *c);
int flush_pending(tirt_ctx *c);
}  // namespace tirt
