// tirt_render.hip -- BVH traversal + the PT_RGB wavefront pipeline.
//
// Reference path: integrator/PT_RGB.py:44-136 is ONE per-pixel megakernel that inlines
// Scene.closet_hit (Scene.py:702-744, traversal stack in GLOBAL memory, exhaustive visit
// order, no t-culling), the shading branch, Scene.closet_hit_shadow (:671-699) and the film
// update.  Here the same arithmetic is split into a wavefront of kernels over struct-of-arrays
// path state in HBM:
//
//   k_generate        camera rays                                   Camera.py:122-142
//   k_trace<closest>  closest hit of every live path                Scene.py:702-744
//   k_shade           light / glass / disney branch, NEE set-up,    integrator/PT_RGB.py:66-132
//                     next ray, throughput; compacts live paths
//                     into the next dense state (ballots + one
//                     packed atomic per block)
//   k_trace<shadow>   NEE visibility, adds the stored contribution  Scene.py:671-699, PT_RGB.py:104-109
//   k_film            running-mean film update                      integrator/PT_RGB.py:134-136
//
// Traversal: persistent waves, one ray per lane with re-fetch, traversal stack in LDS ([entry][lane]
// layout: conflict-free) with a paged global-memory spill buffer.  Two visiting rules, same closest hit:
//   EXHAUSTIVE  the reference's, on the 64-byte two-child nodes: every internal node whose box
//               passes `slabs` has both children visited, leaves are intersected unconditionally.
//               Used for the N_box / N_leaf "algorithmic bytes" counts and as a parity cross-check.
//   ORDERED     on the 64-byte quantised 4-wide nodes (tirt_internal.h `cnode`; top five levels in LDS):
//               children near to far, children whose entry distance exceeds the current hit (with a
//               1e-4 relative margin) are skipped, leaf children are pre-tested against their (slightly
//               inflated) box.  The quantised boxes contain the reference's; a candidate hit is accepted
//               only if the reference would have reached that leaf (`slabs` on the exact boxes, see
//               BvhView), so the closest hit is the reference's.  Product default.
// Exact-t ties are resolved as the reference's visit order does (it pops the right child
// first and keeps the first-found candidate on `t < hit_t`): the candidate with the larger
// compact-node index wins.
#include "tirt_internal.h"
#include "tirt_spectral.h"

namespace tirt {

constexpr int TR_GRID_MAX = 2048;      // upper bound on persistent blocks (sizes the spill buffer)

enum { KIND_CLOSEST = 0, KIND_SHADOW_ACC = 1, KIND_MIXED = 2, KIND_QUERY = 3 };
// The bound ladder (TR_PAD / TR_PADG) is compiled only with -DTIRT_EXPERIMENTS (`make experiments` -> libtirt_exp.so).  What rounds 3-5 built, measured and did not adopt
// left this file in round 6 and can be put back from tools/exp/patches/: the settled A/B switches (stash, drained-slices count, node-loop threshold, quad-cooperative fetch,
// non-temporal streams, asm record fetch, drain diagnostics, no-verify: r06_settled_ab_switches_of_k_trace.patch) and the persistent tail kernel (KIND_TAIL: the rest of a
// batch in one launch, a lane keeping a PATH through shade / shadow ray / next ray; bit-identical, slower at every switch point: r06_persistent_tail_kernel.patch).
#if !defined(TIRT_EXPERIMENTS) && (defined(TR_PAD) || defined(TR_PADG))
#error "the bound ladder (TR_PAD / TR_PADG) needs an experiments build: add -DTIRT_EXPERIMENTS"
#endif
// MIXED: closest rays of bounce b + shadow rays of bounce b-1 in one launch.  QUERY: connection rays of BDPT -- "is sprim[q] the closest
// hit, about sdist[q] away?" walked like a shadow ray (bounded), answered with the hit record (t, u, v, prim) instead of an accumulation.
struct TraceArgs {
    BvhView bvh;
    const float *ox, *oy, *oz, *dx, *dy, *dz;    // rays, dense: ray q at index q
    float eye[3];                                // KIND_CLOSEST with ox == nullptr (camera rays): the common origin
    const int *count_ptr; int count_fixed;       // number of rays: *count_ptr if non-null
    float4 *hit;                                 // KIND_CLOSEST output, index q: (t, u, v, bits prim)
    // KIND_SHADOW_ACC: contribution of ray q goes to (rr,rg,rb)[sdst[q]] or (fr,fg,fb)[~sdst[q]]
    const int *sprim, *sdst; const float *sdist, *scr, *scg, *scb; float *rr, *rg, *rb, *fr, *fg, *fb;
    const float *scw; float *rw, *fw;            // PT_Spec: the fourth hero wavelength's contribution / radiance (nullptr for the RGB integrators)
    int *spill; int spill_depth;                 // global stack tail: [entry][global thread]
    // KIND_MIXED: the shadow rays live in their own arrays; indices [count, count + scount) are shadow rays
    const float *sox, *soy, *soz, *sdx, *sdy, *sdz; const int *scount_ptr;
    // ray-fetch cursors of this launch (zero on entry).  Same-address device atomics retire at one per
    // ~11 ns on MI355X (tools/micro/atomic_rate.hip), which would cap a single cursor at ~3 Grays/s, so
    // the queue is cut into 2^slice_log2 interleaved slices (64-ray chunks, chunk c belongs to slice
    // c mod S), each with its own cursor in its own 128-byte line; a wave drains its home slice and
    // then moves on to the next one.
    int *fetch; int slice_log2;
    // slices_contig: slice h is the h-th CONTIGUOUS 1/S of the queue (of the closest-hit rays and of the shadow rays each) instead of every S-th chunk.
    // The home slice of a wave is (4 x block + wave) mod 32 and blocks go round-robin over the eight XCDs, so slices 4x .. 4x+3 are served by XCD x
    // first: with paths numbered pixel-block major (TileMap::F) that is one region of the film -- one part of the scene -- per L2.
    int slices_contig;
    int lds_depth;                               // stack entries per lane kept in LDS
    int refill_min;                              // re-fetch rays when this many lanes of a wave are idle
    int node_min;                                // leave the inner-node loop below this many busy lanes
    DevCounters *ctr; int2 *per_ray_counts;
    unsigned long long *timeline;        // diagnostics (option "trace_timeline"): per wave [start, queue found empty, end, hardware id] of this launch
    int no_ray_count;                            // the caller counts its rays itself (queues with dead entries)
    // KIND_CLOSEST / KIND_QUERY: the rays as 32-byte records instead of six arrays -- (o.xyz, d.x), (d.y, d.z, bits expect, bound): two 16-byte
    // loads per ray where the arrays take six to eight (the BDPT ray lists: their kernels are bound by the number of memory instructions)
    const float4 *ray4;
    const int *ray_index;                        // KIND_QUERY: ray q is record ray_index[q] of ray4 (BDPT: the connection rays stay where they were staged; the queue is a list of places)
};

// One path at one bounce: what integrator/PT_RGB.py:66-132 does between the closest hit and the next one -- emission (with MIS), the glass / disney
// branch, the NEE sample (its shadow ray and the contribution it adds IF the ray arrives), the next ray and the throughput, the environment
// for a miss.  The body of k_shade, kept apart from its queue handling.  `radiance`, `throughout`, `brdf_pdf`, `perfect_spec` are the path's state coming in; `radiance` is updated in place.
struct ShadeStep {
    bool want_next = false, want_shadow = false, shaded = false;
    v3 next_o = V(0.0f, 0.0f, 0.0f), next_d = next_o, next_thr = next_o, sh_o = next_o, sh_d = next_o, sh_c = next_o;
    float next_pdf = 0.0f, sh_dist = 0.0f; int next_spec = 0, sh_expect = -2;
};
TD void shade_path(const SceneView &sc, const TileMap &tm, int P, uint32_t frame_begin, uint32_t seed, int bounce, int last_bounce, int slot,
                   const v3 origin, const v3 direction, const float4 hrec, v3 throughout, v3 &radiance, float brdf_pdf, int perfect_spec, ShadeStep &s)
{
    int f, k; slot_to_frame_pixel(tm, P, slot, f, k);
    const uint32_t pixel = (uint32_t)local_to_pixel(tm, k);
    const uint32_t frame = frame_begin + (uint32_t)f;
    const uint32_t dim0 = TM_DIM_BOUNCE0 + TM_DIMS_PER_BOUNCE * (uint32_t)bounce;
    const float t = hrec.x;
    if (t < INF_VALUE) {
        const int prim_id = __float_as_int(hrec.w);
        int mat_id;
        const HitAttr h = hit_attributes_rec(sc.shade_rec, origin, direction, prim_id, t, hrec.y, hrec.z, mat_id);
        const v3 normal = h.nor;
        const v3 fnormal = normal * signf(dot(-direction, h.gnor));            // UtilsFunc.py:465-467
        const float *m = sc.material + (size_t)mat_id * MAT_VEC;
        const v3 mat_color = V(m[2], m[3], m[4]);
        const int mat_type = (int)m[0];
        if (mat_type == MAT_LIGHT) {                                           // PT_RGB.py:72-81
            const float fCosTheta = absf(dot(direction, h.gnor));
            if (perfect_spec == 1) {
                radiance = radiance + throughout * mat_color;
            } else {
                const float area = get_prim_area(sc, prim_id) * (float)sc.light_count;
                const float light_pdf = (t * t) / (area * fCosTheta);
                radiance = radiance + (throughout * power_heuristic(brdf_pdf, light_pdf)) * mat_color;
            }
        } else {
            s.shaded = true;
            // UF.srgb_to_lrgb(material colour) (PT_RGB.py:86): per-material table filled by the same device function
            const v3 reflect_color = V(sc.mat_lrgb[mat_id * 3], sc.mat_lrgb[mat_id * 3 + 1], sc.mat_lrgb[mat_id * 3 + 2]);
            v3 next_dir; float f_or_b = 1.0f, brdf = 1.0f;
            if (mat_type == MAT_GLASS) {                                       // PT_RGB.py:89-92
                perfect_spec = 1;
                next_dir = glass_sample(m, direction, normal, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_GLASS), f_or_b);
                brdf = 1.0f; brdf_pdf = 1.0f;
            } else {
                perfect_spec = 0;
                // Scene.py:477-518 sample_li.  No emitters (env-lit scene): the reference would index light[-1]
                // (Scene.py:423-428, undefined) -- defined here, as in the oracle, as "no NEE sample"
                if (sc.light_count > 0) {
                int lidx = (int)(tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LIGHT) * (float)sc.light_count);
                if (lidx >= sc.light_count) lidx = sc.light_count - 1;
                const int light_prim = sc.light[lidx];
                const float ra = tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LA);
                const float rb = tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LB);
                v3 light_pos, light_normal;
                get_prim_random_point_normal(sc, light_prim, ra, rb, light_pos, light_normal);
                const int lmat = sc.primitive[(size_t)light_prim * PRI_VEC + 2];
                const float *lm = sc.material + (size_t)lmat * MAT_VEC;
                const v3 light_emission0 = V(lm[2], lm[3], lm[4]);
                const float light_area = get_prim_area(sc, light_prim);
                float light_choice_pdf = 1.0f / ((float)sc.light_count * light_area);
                light_normal = normalized(light_normal);
                v3 light_dir = h.pos - light_pos;
                const float light_dist = norm(light_dir);
                light_dir = light_dir / light_dist;
                const v3 light_emission = light_emission0 * light_shape_visible(sc, light_prim, light_dir, light_normal, light_dist, light_choice_pdf);   // spot / laser (Scene.py:491-516)
                const float NdotL_surface = dot(fnormal, light_dir);            // PT_RGB.py:101-109
                const float NdotL_light = dot(light_normal, light_dir);
                if ((NdotL_surface < 0.0f) & (NdotL_light > 0.0f)) {
                    s.want_shadow = true;
                    float e_pdf;
                    const float e_brdf = disney_evaluate_pdf(m, fnormal, -direction, -light_dir, e_pdf);
                    const float light_pdf = light_dist * light_dist * light_choice_pdf / NdotL_light;
                    v3 c = V(0.0f, 0.0f, 0.0f);
                    int expect = -2;                       // never equals a primitive id
                    if (e_pdf > 0.0f) {
                        const float w = power_heuristic(light_pdf, e_pdf) / maxf(0.0001f, light_pdf);
                        c = light_emission * w;
                        c = c * throughout;
                        c = c * reflect_color;
                        c = c * e_brdf;
                        c = c * absf(NdotL_surface);
                        expect = prim_id;
                    }
                    s.sh_o = light_pos; s.sh_d = light_dir; s.sh_c = c; s.sh_expect = expect; s.sh_dist = light_dist;
                }
                }   // light_count > 0
                next_dir = disney_sample(m, direction, fnormal, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LOBE),
                                         tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R1),
                                         tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R2));
                f_or_b = 1.0f;
                brdf = disney_evaluate_pdf(m, fnormal, -direction, next_dir, brdf_pdf);
                brdf *= absf(dot(normal, next_dir));
            }
            const v3 next_origin = offset_ray(h.pos, fnormal * signf(f_or_b));   // PT_RGB.py:115
            if (brdf_pdf > 0.0f) {
                bool alive = true;
                if (f_or_b < 0.0f) {                                             // PT_RGB.py:118-122
                    const float extinction = m[6];
                    const float R = tm_exp(-t / extinction);
                    if (tm_rand(seed, pixel, frame, dim0 + TM_SLOT_EXT) >= R) alive = false;
                }
                if (alive) {
                    throughout = throughout * (reflect_color * (brdf / brdf_pdf));
                    s.want_next = !last_bounce;      // depth reaches MAX_DEPTH after the last bounce: the loop ends
                    s.next_o = next_origin; s.next_d = next_dir; s.next_thr = throughout;
                    s.next_pdf = brdf_pdf; s.next_spec = perfect_spec;
                }
            }
        }
    } else if (sc.env_power == 0.0f && (direction.x - direction.x == 0.0f) && (direction.y - direction.y == 0.0f) &&
               (direction.z - direction.z == 0.0f)) {
        // black environment (PT_RGB.py:127-132 with env_power == 0): for a finite direction the lookup returns a
        // finite e >= 0, so (e * throughput) * 0 is +-0 with throughput's sign, or NaN where throughput is not
        // finite -- exactly throughput * env_power, without the two atan2, four texel fetches and three pow
        radiance = radiance + throughout * sc.env_power;
    } else {                                                                     // PT_RGB.py:127-132
        const float dis = tm_sqrt(direction.x * direction.x + direction.z * direction.z);
        const float tx = (tm_atan2(direction.z, direction.x) + PI_SCENE) / PI_SCENE / 2.0f;
        const float ty = tm_atan2(direction.y, dis) / PI_SCENE + 0.5f;
        const v3 e = srgb_to_lrgb(texture2d(sc, tx, ty));
        radiance = radiance + (e * throughout) * sc.env_power;
    }
}

// Persistent waves with ray re-fetch ("while-while" traversal): every wave keeps pulling rays
// from the queue through one wave-aggregated atomic whenever at least TR_REFILL_MIN of its 64
// lanes are idle, so that a few long rays do not leave the other lanes of the wave parked
// (a one-ray-per-lane loop measured 15 % VALU lane utilisation on this workload).  Inner-node
// steps and triangle tests run in separate loops so that lanes doing the same thing run together.
constexpr int TR_FETCH_STRIDE = 32;           // ints between two slice cursors (one 128-byte line each)
constexpr int TR_SLICES_MAX = 64;
constexpr int TR_FETCH_LINES = TR_SLICES_MAX + 1;   // one cursor per slice, each in a line of its own, + the line of the drained-slices count
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) int lds_int;
constexpr int TR_PAGE = 8;                    // stack entries moved per page-out / page-in
constexpr int TR_SENT = (int)0x80000000;      // "stack empty": never a node index nor a leaf code

TD unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
TD bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

TD unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

constexpr int TR_MIN_WAVES = 6;        // 80 VGPRs: five 256-thread blocks per CU and room for a shading wave per SIMD (tirt_internal.h, TR_TOP_CAP)
// The arguments a ray only needs when it is fetched, written back or paged (40 pointers, the grid constants) are NOT read from the by-value
// argument: held in SGPRs across the whole walk they overflow the scalar register file, and the compiler parks them in VGPR lanes --
// v_writelane / v_readlane, VALU instructions, ~245 of them per refill of a kernel that is bound by VALU issue.  They are read from the
// kernel-argument segment where they are needed instead (scalar loads: no VALU), through a pointer the optimiser cannot see through, so
// that it can neither hoist the loads out of the persistent loop nor keep their results alive across it.
typedef const __attribute__((address_space(4))) TraceArgs *cold_args_t;
#define TR_COLD(ca) cold_args_t ca = (cold_args_t)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(ca))

// ---- bound ladder (tools/ladder.sh, profiles/r05_bound_ladder.txt; experiments only, nothing of it in the product build) ----
// What limits k_trace is asked of the kernel itself: -DTR_PAD=k adds k VALU instructions to every node visit that change nothing (v_fma_f32 x, x, 1.0, 0
// on the nine grid registers of the ray: independent chains, no extra register, films stay bit-identical) -- a kernel bound by VALU issue slows down
// from k = 0 on by k / (instructions of a visit), one with slack does not until the slack is used; -DTR_PAD_LATE puts them behind the sort (in the
// dependent chain of the visit) instead of behind the loads (in the shadow of their latency).  -DTR_PADG=k adds k dummy gathers -- one dword from
// k other 64-byte node records, landed in a scratch kilobyte of LDS by global_load_lds, so that no register and no wait is spent on them: load on
// the texture-address / L1 / L2 path only (+ 3 VALU each for the address).
#if defined(TR_PAD) || defined(TR_PADG)
#ifndef TR_PAD
#define TR_PAD 0
#endif
#ifndef TR_PADG
#define TR_PADG 0
#endif
#ifdef TR_PAD_LATE
#define TR_PAD_WHERE 1
#else
#define TR_PAD_WHERE 0
#endif
#ifdef TR_PAD_MAX          // the pad as an instruction of the node step's own class (v_max / v_min / v_cmp / v_cndmask / v_alignbit / v_fma_mix issue at half the rate of v_fma / v_add: tools/micro/valu_issue.hip)
#define TR_PAD1(reg) asm volatile("v_max_f32 %0, %0, %0" : "+v"(reg))
#else
#define TR_PAD1(reg) asm volatile("v_fma_f32 %0, %0, 1.0, 0" : "+v"(reg))
#endif
#define TR_LADDER_PADS(where)                                                                        \
    do {                                                                                             \
        if ((where) == TR_PAD_WHERE) {                                                               \
            _Pragma("unroll") for (int p__ = 0; p__ < TR_PAD; p__++) {                               \
                switch (p__ % 9) {                                                                   \
                case 0: TR_PAD1(gAx); break; case 1: TR_PAD1(gAy); break; case 2: TR_PAD1(gAz); break;      \
                case 3: TR_PAD1(gBnx); break; case 4: TR_PAD1(gBny); break; case 5: TR_PAD1(gBnz); break;   \
                case 6: TR_PAD1(gBfx); break; case 7: TR_PAD1(gBfy); break; default: TR_PAD1(gBfz); break;  \
                }                                                                                    \
            }                                                                                        \
        }                                                                                            \
        if ((where) == 0) {                                                                          \
            _Pragma("unroll") for (int g__ = 0; g__ < TR_PADG; g__++) {                              \
                const unsigned dn__ = ((unsigned)cur + 977u * (unsigned)(g__ + 1)) & 32767u;         \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)b.cnode + (dn__ << 6)), \
                    (__attribute__((address_space(3))) void *)(size_t)(top_base + (unsigned)TR_TOP_SLOTS * 64u + (unsigned)(tid & ~63) * 4u), 4, 0, 0); \
            }                                                                                        \
        }                                                                                            \
    } while (0)
#else
#define TR_LADDER_PADS(where) do { } while (0)
#endif

template <int MODE, bool COUNT, int KIND>
__global__ __launch_bounds__(TR_BLOCK, TR_MIN_WAVES) void k_trace(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int lds_stack[];      // [lds_depth][TR_BLOCK]
    const int TR_LDS_DEPTH = a.lds_depth;
    constexpr bool MAY_SHADOW = (KIND != KIND_CLOSEST);
    constexpr bool STASH = (MODE != TIRT_TRAVERSE_EXHAUSTIVE);
    constexpr bool BOUNDED = MAY_SHADOW && (MODE != TIRT_TRAVERSE_EXHAUSTIVE);
    const int count_c = (KIND == KIND_SHADOW_ACC || KIND == KIND_QUERY) ? 0 : (a.count_ptr ? *a.count_ptr : a.count_fixed);
    const int count_s = (KIND == KIND_CLOSEST) ? 0 : (KIND == KIND_MIXED ? *a.scount_ptr : (a.count_ptr ? *a.count_ptr : a.count_fixed));
    const int count = count_c + count_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const size_t gstride = (size_t)gridDim.x * TR_BLOCK, gtid = (size_t)blockIdx.x * TR_BLOCK + tid;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const BvhView &b = a.bvh;

    // per-lane ray state
    bool have = false, par = false, is_sh = (KIND == KIND_SHADOW_ACC || KIND == KIND_QUERY);
    int q = 0, cur = TR_SENT, hit_prim = -1, hit_leaf = -1, expect = -3;
    int pend = 0;                               // ordered mode: one stashed leaf code (0 = none; leaf codes are negative)
    unsigned n_overflow = 0;
    float hit_t = INF_VALUE, hit_u = 0.0f, hit_v = 0.0f, cull_far = 3.0e38f, settle = -1.0f;
    float lim = INF_VALUE;                      // ordered mode: min(hit_t * 1.0001, cull_far, INF_VALUE), entry distances beyond it are skipped
    // (ordered mode: a NEGATIVE cull_far marks a ray whose origin is more than TR_FAR_RHO root-box extents away -- no distance culling for it, see the set-up code)
    unsigned nbox = 0, nleaf = 0;
    RayCtx r = {};
    // ordered mode: the ray in the grid of the quantised nodes.  Plane h (fp16, in cells) of axis a is crossed at
    // t = h * gA + gB (gA = cell / d, gB = (grid_min - o) / d); gBn / gBf are gB moved outward by a margin of
    // 0.25 cells + 0.25 cells per root-box extent of distance between the origin and the grid (see the set-up code
    // for what it covers); grot = 16 where gA < 0 rotates a (min | max << 16) plane
    // pair so that the low half is always the near plane.  Axis-parallel components (|d| < 1e-6, where the
    // reference tests the origin against the slab instead, UtilsFunc.py:500-503) do the same in grid units through
    // the same two FMAs: gA = 1e30, gBn / gBf = (-(origin cell) -+ margin) * 1e30, so that the "near distance" is hugely
    // negative or positive and the "far distance" hugely positive or negative according to the side of the planes
    // the origin lies on.  (Ignoring such an axis would be conservative too, but a ray that ignores an axis walks a
    // whole slice of the scene: a few of them per million rays set the duration of every launch.)
    float gAx = 0.0f, gAy = 0.0f, gAz = 0.0f, gBnx = 0.0f, gBny = 0.0f, gBnz = 0.0f, gBfx = 0.0f, gBfy = 0.0f, gBfz = 0.0f;
    int grotx = 0, groty = 0, grotz = 0;
    bool exhausted = false;
    unsigned long long tk_start = 0, tk_exh = 0;
    if (COUNT) tk_start = wall_clock64();
    const int S_LOG = a.slice_log2, S_MASK = (1 << S_LOG) - 1;
    int home = (int)((blockIdx.x * (TR_BLOCK / 64) + (tid >> 6)) & S_MASK), tried = 0;      // wave-uniform
    const int full_chunks = count >> 6;
    unsigned long long sum_box = 0, sum_leaf = 0, sum_box_s = 0, sum_leaf_s = 0, n_over = 0;
    unsigned long long d_it_node = 0, d_lanes_node = 0, d_it_leaf = 0, d_lanes_leaf = 0, d_refills = 0, d_outer = 0;

    // Traversal stack.  `sa` is the LDS address of the TOP entry; entry e of a lane lives at
    // lds_stack + e * TR_BLOCK * 4 + tid * 4.  Entry 0 is a sentinel (TR_SENT, written once), so a pop
    // needs no emptiness test; entries 1 .. TR_LDS_DEPTH-1 hold the stack.  A lane whose LDS part is
    // full pages its oldest TR_PAGE entries out to the global spill buffer ([entry][global thread])
    // and pages them back in when it pops the sentinel -- both on wave-uniform cold paths outside the
    // inner-node loop (which leaves as soon as some lane's LDS part is full), so the hot loop only
    // ever touches LDS.  Addresses are compared as signed ints: after popping the sentinel `sa` is one
    // entry below the bottom.
    constexpr unsigned ENTRY = TR_BLOCK * 4u;
    const unsigned sa_bottom = (unsigned)(size_t)(lds_int *)lds_stack + (unsigned)tid * 4u;
    const unsigned sa_hi = (unsigned)(TR_LDS_DEPTH - 3) * ENTRY + sa_bottom;     // a step pushes up to three entries: page out at this fill level
    unsigned sa = sa_bottom;
    int paged = 0;                              // pages of this lane's stack that live in the spill buffer
#define LDS_AT(addr) (*(lds_int *)(size_t)(addr))
    LDS_AT(sa_bottom) = TR_SENT;
    // the top TR_TOP_LEVELS levels of the 4-wide tree (7 x 16 bytes per record) sit behind the stacks: a
    // visit there costs LDS bandwidth instead of the texture-address path this kernel is bound by
    const unsigned top_base = (unsigned)(size_t)(lds_int *)lds_stack + (unsigned)TR_LDS_DEPTH * ENTRY;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u4v lds_u4;
    if (MODE != TIRT_TRAVERSE_EXHAUSTIVE) {
        for (int k = tid; k < b.top_count * 4; k += TR_BLOCK)
        { const uint4 g = b.cnode[k]; *(lds_u4 *)(size_t)(top_base + (unsigned)k * 16u) = u4v{g.x, g.y, g.z, g.w}; }
        __syncthreads();
    }
#define TR_PAGE_OUT()                                                                                \
    do {                                                                                             \
        TR_COLD(ca);                                                                                 \
        if ((paged + 1) * TR_PAGE <= ca->spill_depth) {                                                \
            for (int k__ = 0; k__ < TR_PAGE; k__++)                                                  \
                ca->spill[(size_t)(paged * TR_PAGE + k__) * gstride + gtid] = LDS_AT(sa_bottom + (unsigned)(k__ + 1) * ENTRY); \
            for (unsigned e__ = sa_bottom + (TR_PAGE + 1) * ENTRY; e__ <= sa; e__ += ENTRY)          \
                LDS_AT(e__ - TR_PAGE * ENTRY) = LDS_AT(e__);                                         \
            sa -= TR_PAGE * ENTRY; paged++;                                                          \
        } else { n_overflow++; sa -= ENTRY; }      /* out of room: the entry on top is lost (counted) */ \
    } while (0)
#define TR_PAGE_IN()                                                                                 \
    do {                                                                                             \
        TR_COLD(ca);                                                                                 \
        paged--;                                                                                     \
        for (int k__ = 0; k__ < TR_PAGE; k__++)                                                      \
            LDS_AT(sa_bottom + (unsigned)(k__ + 1) * ENTRY) = ca->spill[(size_t)(paged * TR_PAGE + k__) * gstride + gtid]; \
        sa = sa_bottom + TR_PAGE * ENTRY;                                                            \
    } while (0)
#define TR_POP(dst) do { dst = LDS_AT(sa); sa -= ENTRY; } while (0)

    for (;;) {
        // ---- refill idle lanes -------------------------------------------------------------
        const unsigned long long idle = ballot64(!have);
        if (idle != 0ull && !exhausted && (__popcll(idle) >= a.refill_min || idle == ~0ull)) {
            TR_COLD(ca);
            // (all of them read here, in one batch of scalar loads with one wait: left to itself the compiler loads each where it is used,
            // a chain of a dozen dependent round trips per refill)
            int *const c_fetch = ca->fetch;
            const float *const c_ox = ca->ox, *const c_oy = ca->oy, *const c_oz = ca->oz, *const c_dx = ca->dx, *const c_dy = ca->dy, *const c_dz = ca->dz;
            const float *const c_sox = MAY_SHADOW ? ca->sox : nullptr, *const c_soy = MAY_SHADOW ? ca->soy : nullptr, *const c_soz = MAY_SHADOW ? ca->soz : nullptr;
            const float *const c_sdx = MAY_SHADOW ? ca->sdx : nullptr, *const c_sdy = MAY_SHADOW ? ca->sdy : nullptr, *const c_sdz = MAY_SHADOW ? ca->sdz : nullptr;
            const int *const c_sprim = MAY_SHADOW ? ca->sprim : nullptr; const float *const c_sdist = MAY_SHADOW ? ca->sdist : nullptr;
            const float4 *const c_ray4 = (KIND == KIND_CLOSEST || KIND == KIND_QUERY) ? ca->ray4 : nullptr;
            const int *const c_rindex = (KIND == KIND_QUERY) ? ca->ray_index : nullptr;
            const float c_gm0 = ca->bvh.grid_min[0], c_gm1 = ca->bvh.grid_min[1], c_gm2 = ca->bvh.grid_min[2];
            const float c_ie0 = ca->bvh.inv_extent[0], c_ie1 = ca->bvh.inv_extent[1], c_ie2 = ca->bvh.inv_extent[2];
            const float c_ic0 = ca->bvh.inv_cell[0], c_ic1 = ca->bvh.inv_cell[1], c_ic2 = ca->bvh.inv_cell[2];
            const float c_ce0 = ca->bvh.cell[0], c_ce1 = ca->bvh.cell[1], c_ce2 = ca->bvh.cell[2];
            const float c_rn0 = ca->bvh.root_min[0], c_rn1 = ca->bvh.root_min[1], c_rn2 = ca->bvh.root_min[2];
            const float c_rx0 = ca->bvh.root_max[0], c_rx1 = ca->bvh.root_max[1], c_rx2 = ca->bvh.root_max[2];
            const int c_root = ca->bvh.root_code, c_rootq = ca->bvh.root_qcode, c_farq = ca->bvh.far_qcode;
            asm volatile("" :: "s"(c_fetch), "s"(c_ox), "s"(c_oy), "s"(c_oz), "s"(c_dx), "s"(c_dy), "s"(c_dz), "s"(c_gm0), "s"(c_gm1), "s"(c_gm2), "s"(c_ie0), "s"(c_ie1), "s"(c_ie2),
                         "s"(c_ic0), "s"(c_ic1), "s"(c_ic2), "s"(c_ce0), "s"(c_ce1), "s"(c_ce2), "s"(c_rn0), "s"(c_rn1), "s"(c_rn2), "s"(c_rx0), "s"(c_rx1), "s"(c_rx2));
            if (MAY_SHADOW) asm volatile("" :: "s"(c_sox), "s"(c_soy), "s"(c_soz), "s"(c_sdx), "s"(c_sdy), "s"(c_sdz), "s"(c_sprim), "s"(c_sdist));
            if (KIND == KIND_CLOSEST || KIND == KIND_QUERY) asm volatile("" :: "s"(c_ray4), "s"(c_rindex));
            const unsigned long long fm = idle;
            int my = count;
            if (fm != 0ull && !exhausted) {
            const int n_idle = __popcll(fm);
            if (COUNT) d_refills++;
            const int leader = __ffsll((long long)fm) - 1;
            // rays of slice `home`: its 64-ray chunks are the global chunks home, home + S, home + 2S, ... (interleaved), or the home-th contiguous
            // stretch of the closest-hit rays followed by the home-th stretch of the shadow rays (slices_contig)
            const bool contig = ca->slices_contig != 0;
            const int Lc__ = ((((count_c + 63) >> 6) + S_MASK) >> S_LOG) << 6, Ls__ = ((((count_s + 63) >> 6) + S_MASK) >> S_LOG) << 6;
            int lenc__ = count_c - home * Lc__; lenc__ = lenc__ < 0 ? 0 : (lenc__ > Lc__ ? Lc__ : lenc__);
            int lens__ = count_s - home * Ls__; lens__ = lens__ < 0 ? 0 : (lens__ > Ls__ ? Ls__ : lens__);
            const int len = contig ? lenc__ + lens__
                                   : (((full_chunks >> S_LOG) + (home < (full_chunks & S_MASK) ? 1 : 0)) << 6) + (home == (full_chunks & S_MASK) ? (count & 63) : 0);
            int base = 0;
            if (lane == leader) base = atomicAdd(c_fetch + home * TR_FETCH_STRIDE, n_idle);
            base = __shfl(base, leader, 64);
            const int v = base + __popcll(fm & lt_mask);                  // index within the slice
            if (contig) my = v < lenc__ ? home * Lc__ + v : (v < len ? count_c + home * Ls__ + (v - lenc__) : count);
            else my = v < len ? (((((v >> 6) << S_LOG) + home) << 6) | (v & 63)) : count;
            if (base + n_idle >= len) {                                    // slice drained: move on
                // The wave whose fetch reached the end of a slice counts the slice as drained, and a wave that moves on looks at that count: once it
                // says "all of them" there is nothing to look for.  Without it every wave of a launch learns that by one atomic round trip per
                // slice, 32 in a row on 32 lines that 5 120 waves are hammering: ~0.15 ms, the better part of what a launch of few rays takes.
                int *const c_drained = c_fetch + TR_SLICES_MAX * TR_FETCH_STRIDE;
                if (lane == leader && base < (len > 0 ? len : 1)) atomicAdd(c_drained, 1);
                if (__hip_atomic_load(c_drained, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > S_MASK) tried = S_MASK;
                home = (home + 1) & S_MASK;
                if (++tried > S_MASK) { exhausted = true; if (COUNT) tk_exh = wall_clock64(); }
            }
            }
            const bool setup_now = !have && my < count;
            if (setup_now) {
                q = my;
                if (KIND == KIND_MIXED) { is_sh = my >= count_c; if (is_sh) q = my - count_c; }
                const bool mixed_sh = (KIND == KIND_MIXED) && is_sh;
                v3 o, d; int rec_expect = -3; float rec_bound = -1.0f;
                const bool from_rec = (KIND == KIND_CLOSEST || KIND == KIND_QUERY) && c_ray4 != nullptr;          // wave-uniform
                if (from_rec) {
                    const size_t ri = (KIND == KIND_QUERY && c_rindex) ? (size_t)c_rindex[q] : (size_t)q;
                    const float4 r0 = c_ray4[2 * ri], r1 = c_ray4[2 * ri + 1];
                    o = V(r0.x, r0.y, r0.z); d = V(r0.w, r1.x, r1.y); rec_expect = __float_as_int(r1.z); rec_bound = r1.w;
                } else {
#define TR_LDS_(p) __builtin_nontemporal_load(&(p))
                    o = mixed_sh ? V(TR_LDS_(c_sox[q]), TR_LDS_(c_soy[q]), TR_LDS_(c_soz[q]))
                                 : ((KIND == KIND_CLOSEST && c_ox == nullptr) ? V(ca->eye[0], ca->eye[1], ca->eye[2]) : V(TR_LDS_(c_ox[q]), TR_LDS_(c_oy[q]), TR_LDS_(c_oz[q])));
                    d = mixed_sh ? V(TR_LDS_(c_sdx[q]), TR_LDS_(c_sdy[q]), TR_LDS_(c_sdz[q])) : V(TR_LDS_(c_dx[q]), TR_LDS_(c_dy[q]), TR_LDS_(c_dz[q]));
                }
                r = make_ray(o, d);
                par = ray_has_parallel_axis(r);
                hit_t = INF_VALUE; hit_u = 0.0f; hit_v = 0.0f; hit_prim = -1; hit_leaf = -1;
                nbox = 1; nleaf = 0; sa = sa_bottom; paged = 0; n_overflow = 0; pend = 0;
                cull_far = 3.0e38f; settle = -1.0f; expect = -3;
                float t_bound = -1.0f;
                if (MAY_SHADOW && is_sh) {
                    expect = from_rec ? rec_expect : TR_LDS_(c_sprim[q]);
                    if (BOUNDED) t_bound = from_rec ? rec_bound : TR_LDS_(c_sdist[q]);
                }
                if (MODE != TIRT_TRAVERSE_EXHAUSTIVE) {
                    // margin in cells, the same on all axes (trace_margin_cells, tirt_internal.h: what it has to cover); rho = trace_origin_rho's expression on the cold arguments
                    const float relx__ = c_gm0 - o.x, rely__ = c_gm1 - o.y, relz__ = c_gm2 - o.z;
                    const float rho__ = maxf(maxf(absf(relx__) * c_ie0, absf(rely__) * c_ie1), absf(relz__) * c_ie2);
                    const float mc__ = trace_margin_cells(rho__);
                    // From far away the reference's Moller-Trumbore returns distances that are rounding noise (its o - v0 carries
                    // |o - v0| * 2^-24, divided by a determinant of the order of the triangle's area): a small or edge-on triangle
                    // seen from hundreds of extents away can "hit" tens of units in FRONT of the surface the ray really meets first,
                    // and the reference, which never culls by distance, takes it.  A ray that starts more than TR_FAR_RHO extents
                    // from the grid therefore does not cull by distance either (no hit-distance cull, no target-distance bound):
                    // it visits every leaf whose boxes it passes, as the reference does, and so finds the same candidates.
                    if (rho__ > TR_FAR_RHO) cull_far = -3.0e38f;
#define TR_GRID_AXIS(rel, dd, idd, k, gA, gBn, gBf, grot)                                            \
                    do {                                                                             \
                        if (absf(dd) < 0.000001f) {      /* the reference's parallel case: origin inside the slab or no hit */ \
                            const float og__ = -(rel) * (k == 0 ? c_ic0 : (k == 1 ? c_ic1 : c_ic2));                               \
                            gA = 1.0e30f; gBn = (-og__ - (mc__ + 1.0f)) * 1.0e30f; gBf = (-og__ + (mc__ + 1.0f)) * 1.0e30f; grot = 0; \
                        } else {                                                                     \
                            gA = (k == 0 ? c_ce0 : (k == 1 ? c_ce1 : c_ce2)) * (idd);                                                  \
                            const float gB__ = (rel) * (idd);                                        \
                            const float m__ = mc__ * absf(gA);                                       \
                            gBn = gB__ - m__; gBf = gB__ + m__; grot = gA < 0.0f ? 16 : 0;           \
                        }                                                                            \
                    } while (0)
                    TR_GRID_AXIS(relx__, d.x, r.idx, 0, gAx, gBnx, gBfx, grotx);
                    TR_GRID_AXIS(rely__, d.y, r.idy, 1, gAy, gBny, gBfy, groty);
                    TR_GRID_AXIS(relz__, d.z, r.idz, 2, gAz, gBnz, gBfz, grotz);
                }
                if (BOUNDED && t_bound > 0.0f && cull_far > 0.0f) { cull_far = t_bound * 1.01f; settle = t_bound * 0.99f; }
                lim = __builtin_fminf(__builtin_fminf(hit_t * 1.0001f, __builtin_fabsf(cull_far)), INF_VALUE);      // (v_min: a canonical value, so the node loop does not re-canonicalise it every step)
                // (a far-origin ray cannot trust the padded boxes of the analytic spheres: it starts at a chain node that holds the root and
                // those spheres with whole-grid boxes, BvhView::far_qcode, and so tests them whatever their boxes say -- as the reference does)
                cur = (MODE == TIRT_TRAVERSE_EXHAUSTIVE) ? c_root : (cull_far < 0.0f ? c_farq : c_rootq);
                // The reference's first step: the root box.  The ordered walk only needs it as a shortcut -- a candidate is accepted after the boxes
                // of ALL its ancestors, the root's included, have been seen to pass `slabs` (leaf phase) --, and it is a shortcut for rays that
                // start outside the scene (camera rays: KIND_CLOSEST).  Bounce and shadow rays start on a surface inside the root box and pass it:
                // for them the test (~30 instructions per refill, at a fifth of the lanes) is skipped and the rare outsider spends one node step instead.
                if ((MODE == TIRT_TRAVERSE_EXHAUSTIVE || KIND == KIND_CLOSEST) && cur >= 0) {
                    float tn;
                    if (!slabs(r, c_rn0, c_rn1, c_rn2, c_rx0, c_rx1, c_rx2, tn)) cur = TR_SENT;
                }
                // A ray with a NaN component passes every `slabs` test (all comparisons are false)
                // and fails every primitive test, in the reference as well: it walks the whole tree
                // and misses.  Same result, without the walk (ordered mode; the exhaustive mode keeps the walk).  (NaN shading normals come from
                // process_normal's acos of a dot product just above 1, Scene.py:377.)
                if (MODE != TIRT_TRAVERSE_EXHAUSTIVE &&
                    !((o.x == o.x) & (o.y == o.y) & (o.z == o.z) & (d.x == d.x) & (d.y == d.y) & (d.z == d.z))) cur = TR_SENT;
                have = true;
            }
        }
        if (ballot64(have) == 0ull) { if (exhausted) break; continue; }

        // ---- inner nodes.  Lanes that reached a leaf (or finished) wait here; the loop goes on
        // while at least node_min lanes still have inner-node work, or nobody is waiting at all.
        // (idle lanes keep cur == TR_SENT, so `cur >= 0` alone means "has inner-node work".)
        // Written to keep the VALU and SALU instruction counts down (59 VALU per step, was 82): the
        // t-cull folded into the far distance, selects on lane masks, stack push/pop without index
        // arithmetic or emptiness tests, no cold code inside the loop.
        // A lane that reaches a leaf does not stop there: the leaf goes into a one-entry stash (`pend`) and the walk
        // goes on with the next stack entry; only a second leaf makes the lane wait.  The primitive tests then run for
        // all stashed leaves of the wave together (lane utilisation of the node loop 0.61 -> 0.73, of the leaf phase 0.38 -> 0.45), at the
        // price of a few node visits the earlier hit would have culled (+2 %).
        if (STASH) { if (have && cur < 0 && cur != TR_SENT && pend == 0) { pend = cur; TR_POP(cur); } }
        const int lim_i = __float_as_int(lim);                  // > 0 always
        // The node loop is left once fewer than node_min lanes have node work and some lane waits at a leaf -- at most: a wave that holds few
        // rays (the end of a launch, no refill any more) would leave it for every single leaf, so the threshold is also bounded by 3/4 of
        // the lanes that hold a ray at all (1/2 and 5/8: no gain, 7/8: half of it; a lone batch -- one rank's share of an 8-GPU job --
        // 3.13 -> 3.06 ms per step, four overlapped batches unchanged: profiles/r04g_node_threshold_ab.log)
        constexpr int TR_NODE_FRAC = 6;
        int node_min_w = (__popcll(ballot64(have)) * TR_NODE_FRAC) >> 3; node_min_w = node_min_w < a.node_min ? node_min_w : a.node_min;
        for (;;) {
            const bool act = cur >= 0;
            const unsigned long long am = ballot64(act);
            if (am == 0ull) break;
            const int n_act = __popcll(am);
            if (n_act < node_min_w && ballot64(have && cur < 0) != 0ull) break;
            if (wave_any((int)sa >= (int)sa_hi)) break;        // a lane's LDS stack is full: page out below (cold)
            if (COUNT) { d_it_node++; d_lanes_node += (unsigned long long)n_act; }
            if (act) {
                if (MODE == TIRT_TRAVERSE_EXHAUSTIVE) {
                    // reference order on the two-child nodes: both children of every box that passes
                    // `slabs`, leaves without a box test, right child first (left is pushed)
                    const float4 *w = (const float4 *)((const char *)b.wnode + ((unsigned)cur << 6));
                    const float4 q0 = w[0], q1 = w[1], q2 = w[2], q3 = w[3];
                    const int cl = __float_as_int(q3.x), cr = __float_as_int(q3.y);
                    if (COUNT) nbox += 2;
                    float tl, tr;
                    int pl, pr;
                    if (!par) {
                        pl = slabs_fast(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tl);
                        pr = slabs_fast(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tr);
                    } else {
                        pl = slabs(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tl);
                        pr = slabs(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tr);
                    }
                    const bool hl = (pl != 0) || (cl < 0), hr = (pr != 0) || (cr < 0);
                    if (hl && hr) { sa += ENTRY; LDS_AT(sa) = cr; }
                    int next = hl ? cl : cr;
                    if (!(hl || hr)) TR_POP(next);
                    cur = next;
                } else {
                    // ordered mode on the quantised 4-wide nodes: four box tests, children visited near to far
                    uint4 q0, q1, q2, q3;
#define TR_U4(v) make_uint4((v).x, (v).y, (v).z, (v).w)
                    const unsigned top_addr = top_base + ((unsigned)cur << 6);
                    // The record comes from LDS for some lanes and from global memory for the others, into the SAME sixteen registers.  Written in C++ the
                    // compiler orders the two (a write-after-write on a register, as it sees it): s_waitcnt vmcnt(0) before the first ds_read, i.e. a wave with
                    // lanes on both sides -- nearly every step -- pays the two latencies in a row.  The lanes are disjoint (complementary exec masks) and a
                    // returning load writes only the lanes it was issued for, so nothing has to be ordered: both sets of loads are issued back to back here, one
                    // wait.  +2 % on the headline (profiles/r05e: 4 525 -> 4 620 Mrays/s), films and hit records unchanged (same loads, same bits).
                    {
                        u4v r0__, r1__, r2__, r3__;
                        unsigned long long sv__;
                        const unsigned long long topm__ = ballot64(cur < TR_TOP_SLOTS);
                        const unsigned goff__ = (unsigned)cur << 6;
                        asm volatile(
                            "s_mov_b64 %4, exec\n\t"
                            "s_and_b64 exec, %4, %8\n\t"
                            "s_cbranch_execz .Ltr_fetch_g%=\n\t"
                            "ds_read_b128 %0, %5\n\t"
                            "ds_read_b128 %1, %5 offset:16\n\t"
                            "ds_read_b128 %2, %5 offset:32\n\t"
                            "ds_read_b128 %3, %5 offset:48\n"
                            ".Ltr_fetch_g%=:\n\t"
                            "s_andn2_b64 exec, %4, %8\n\t"
                            "s_cbranch_execz .Ltr_fetch_e%=\n\t"
                            "global_load_dwordx4 %0, %6, %7\n\t"
                            "global_load_dwordx4 %1, %6, %7 offset:16\n\t"
                            "global_load_dwordx4 %2, %6, %7 offset:32\n\t"
                            "global_load_dwordx4 %3, %6, %7 offset:48\n"
                            ".Ltr_fetch_e%=:\n\t"
                            "s_mov_b64 exec, %4\n\t"
                            "s_waitcnt vmcnt(0) lgkmcnt(0)"
                            : "=&v"(r0__), "=&v"(r1__), "=&v"(r2__), "=&v"(r3__), "=&s"(sv__)
                            : "v"(top_addr), "v"(goff__), "s"(b.cnode), "s"(topm__)
                            : "memory");
                        q0 = TR_U4(r0__); q1 = TR_U4(r1__); q2 = TR_U4(r2__); q3 = TR_U4(r3__);
                        if (COUNT) { if (cur < TR_TOP_SLOTS) d_outer++; }
                    }
                    TR_LADDER_PADS(0);
                    int c0 = (int)q3.x, c1 = (int)q3.y, c2 = (int)q3.z, c3 = (int)q3.w;
                    if (COUNT) nbox += 4;
                    constexpr float MISS = 3.0e38f;
                    float d0, d1, d2, d3;                // entry distance of a hit box, MISS otherwise
                    // near/far crossing of each axis: one rotate, two v_fma_mix_f32 (fp16 plane x f32 + f32); then
                    // box hit and entry not beyond the cull distance: tn <= min(tf, lim)
#define TR_CBOX(X, Y, Z, dist)                                                                       \
                    do {                                                                             \
                        const unsigned ux__ = __builtin_amdgcn_alignbit((X), (X), grotx);            \
                        const unsigned uy__ = __builtin_amdgcn_alignbit((Y), (Y), groty);            \
                        const unsigned uz__ = __builtin_amdgcn_alignbit((Z), (Z), grotz);            \
                        const h2v hx__ = __builtin_bit_cast(h2v, ux__), hy__ = __builtin_bit_cast(h2v, uy__), hz__ = __builtin_bit_cast(h2v, uz__); \
                        const float nx__ = __builtin_fmaf((float)hx__.x, gAx, gBnx), fx__ = __builtin_fmaf((float)hx__.y, gAx, gBfx); \
                        const float ny__ = __builtin_fmaf((float)hy__.x, gAy, gBny), fy__ = __builtin_fmaf((float)hy__.y, gAy, gBfy); \
                        const float nz__ = __builtin_fmaf((float)hz__.x, gAz, gBnz), fz__ = __builtin_fmaf((float)hz__.y, gAz, gBfz); \
                        const float tn__ = __builtin_fmaxf(__builtin_fmaxf(nx__, ny__), __builtin_fmaxf(nz__, 0.0f)); \
                        /* the far distance in the INTEGER domain (v_min_i32 / v_min3_i32: no canonicalisation of `lim` per step):   \
                           exact for non-negative floats, and a negative operand (box behind the ray) gives a negative result */      \
                        const int tfi__ = min(min(__float_as_int(fx__), __float_as_int(fy__)), min(__float_as_int(fz__), lim_i));     \
                        const float tf__ = __int_as_float(tfi__);                                    \
                        dist = (tn__ <= tf__) ? tn__ : MISS;                                         \
                    } while (0)
                    TR_CBOX(q0.x, q0.y, q0.z, d0);
                    TR_CBOX(q0.w, q1.x, q1.y, d1);
                    TR_CBOX(q1.z, q1.w, q2.x, d2);
                    TR_CBOX(q2.y, q2.z, q2.w, d3);
                    // sort the four (distance, child) pairs: misses end up last
#define TR_CE(da, ca, db, cb)                                                                        \
                    do {                                                                             \
                        const bool s__ = (db) < (da);                                                \
                        const float lo__ = s__ ? (db) : (da), hi__ = s__ ? (da) : (db);              \
                        const int clo__ = s__ ? (cb) : (ca), chi__ = s__ ? (ca) : (cb);              \
                        da = lo__; db = hi__; ca = clo__; cb = chi__;                                \
                    } while (0)
                    // (nearest-only selection, 3 exchanges instead of 5, measured 5 % slower: more nodes visited)
                    TR_CE(d0, c0, d1, c1); TR_CE(d2, c2, d3, c3); TR_CE(d0, c0, d2, c2); TR_CE(d1, c1, d3, c3); TR_CE(d1, c1, d2, c2);
                    TR_LADDER_PADS(1);
                    if (d3 < MISS) { sa += ENTRY; LDS_AT(sa) = c3; }
                    if (d2 < MISS) { sa += ENTRY; LDS_AT(sa) = c2; }
                    if (d1 < MISS) { sa += ENTRY; LDS_AT(sa) = c1; }
                    int next = c0;
                    if (!(d0 < MISS)) TR_POP(next);
                    if (STASH) { if ((next < 0) & (next != TR_SENT) & (pend == 0)) { pend = next; TR_POP(next); } }
                    cur = next;
                }
            }
        }

        // ---- cold: lanes whose LDS stack is full move their oldest entries to the spill buffer ------
        if (wave_any(have && (int)sa >= (int)sa_hi)) {
            if (have && (int)sa >= (int)sa_hi) TR_PAGE_OUT();
        }

        // ---- leaf: one primitive test ---------------------------------------------------------
        const bool from_pend = STASH && pend != 0;
        const bool leaf_now = have && (from_pend || (cur < 0 && cur != TR_SENT));
        if (COUNT) {
            const int n_l = __popcll(ballot64(leaf_now));
            if (n_l) { d_it_leaf++; d_lanes_leaf += (unsigned long long)n_l; }
        }
        if (leaf_now) {
            // trace_leaf_step (tirt_internal.h): the reference's primitive test, its acceptance rule with the equal-distance rule, and -- ordered walk -- the proof
            // that the reference would have visited this leaf (`slabs` on its exact box, else on every ancestor); shared with k_pvb_cand
            const int code = ~(from_pend ? pend : cur);
            if (COUNT) nleaf += 1;
            const bool accepted = trace_leaf_step<MODE != TIRT_TRAVERSE_EXHAUSTIVE>(b, r, par, code, hit_t, hit_u, hit_v, hit_prim, hit_leaf);
            if (from_pend) pend = 0; else TR_POP(cur);
            if (accepted) {
                lim = __builtin_fminf(__builtin_fminf(cull_far < 0.0f ? INF_VALUE : hit_t * 1.0001f, __builtin_fabsf(cull_far)), INF_VALUE);      // (v_min: a canonical value, so the node loop does not re-canonicalise it every step)
                if (BOUNDED && hit_prim != expect && hit_t < settle) { cur = TR_SENT; paged = 0; pend = 0; }      // answer settled: "occluded"
            }
        }

        // ---- a lane that popped the sentinel but has paged-out entries gets them back (cold) ---------
        if (ballot64(have && cur == TR_SENT && paged > 0) != 0ull) {
            if (have && cur == TR_SENT && paged > 0) { TR_PAGE_IN(); TR_POP(cur); }
        }

        // ---- finished rays write back and free their lane ---------------------------------------
        if (have && cur == TR_SENT && pend == 0) {
            TR_COLD(ca);
            float4 *const c_hit = ca->hit;
            const int *const c_sdst = MAY_SHADOW ? ca->sdst : nullptr;
            float *const c_rr = MAY_SHADOW ? ca->rr : nullptr, *const c_rg = MAY_SHADOW ? ca->rg : nullptr, *const c_rb = MAY_SHADOW ? ca->rb : nullptr;
            float *const c_fr = MAY_SHADOW ? ca->fr : nullptr, *const c_fg = MAY_SHADOW ? ca->fg : nullptr, *const c_fb = MAY_SHADOW ? ca->fb : nullptr;
            const float *const c_scr = MAY_SHADOW ? ca->scr : nullptr, *const c_scg = MAY_SHADOW ? ca->scg : nullptr, *const c_scb = MAY_SHADOW ? ca->scb : nullptr;
            const float *const c_scw = MAY_SHADOW ? ca->scw : nullptr;
            if (MAY_SHADOW) asm volatile("" :: "s"(c_hit), "s"(c_sdst), "s"(c_rr), "s"(c_rg), "s"(c_rb), "s"(c_fr), "s"(c_fg), "s"(c_fb), "s"(c_scr), "s"(c_scg), "s"(c_scb), "s"(c_scw));      // one batch of scalar loads (as in the refill)
            if (!(MAY_SHADOW && is_sh) || KIND == KIND_QUERY) {
                { typedef float f4n __attribute__((ext_vector_type(4))); f4n hv__ = {hit_t, hit_u, hit_v, __int_as_float(hit_prim)}; __builtin_nontemporal_store(hv__, (f4n *)&c_hit[q]); }
            } else if (hit_prim == expect) {                 // integrator/PT_RGB.py:105-109
                const int dst = TR_LDS_(c_sdst[q]);
                float *pr = dst >= 0 ? c_rr + dst : c_fr + ~dst;
                float *pg = dst >= 0 ? c_rg + dst : c_fg + ~dst;
                float *pb = dst >= 0 ? c_rb + dst : c_fb + ~dst;
                *pr = *pr + TR_LDS_(c_scr[q]); *pg = *pg + TR_LDS_(c_scg[q]); *pb = *pb + TR_LDS_(c_scb[q]);
                if (c_scw) { float *pw = dst >= 0 ? ca->rw + dst : ca->fw + ~dst; *pw = *pw + c_scw[q]; }
            }
            if (COUNT) {
                if (MAY_SHADOW && is_sh) { sum_box_s += nbox; sum_leaf_s += nleaf; }
                else { sum_box += nbox; sum_leaf += nleaf; }
                if (ca->per_ray_counts) ca->per_ray_counts[q] = make_int2((int)nbox, (int)nleaf);
            }
            if (n_overflow) n_over++;
            have = false; sa = sa_bottom;
        }
    }
    TR_COLD(ca);
    if (ca->ctr) {
        if (COUNT) {
            sum_box = wave_sum(sum_box); sum_leaf = wave_sum(sum_leaf);
            sum_box_s = wave_sum(sum_box_s); sum_leaf_s = wave_sum(sum_leaf_s);
            if (lane == 0 && (sum_box | sum_leaf)) { atomicAdd(&ca->ctr->box_closest, sum_box); atomicAdd(&ca->ctr->leaf_closest, sum_leaf); }
            if (lane == 0 && (sum_box_s | sum_leaf_s)) { atomicAdd(&ca->ctr->box_shadow, sum_box_s); atomicAdd(&ca->ctr->leaf_shadow, sum_leaf_s); }
        }
        if (COUNT) { d_outer = wave_sum(d_outer); if (lane == 0) atomicAdd(&ca->ctr->it_outer, d_outer); }   // lane-visits of LDS-resident (top) nodes
        if (COUNT && lane == 0) {
            atomicAdd(&ca->ctr->it_node, d_it_node); atomicAdd(&ca->ctr->lanes_node, d_lanes_node);
            atomicAdd(&ca->ctr->it_leaf, d_it_leaf); atomicAdd(&ca->ctr->lanes_leaf, d_lanes_leaf);
            atomicAdd(&ca->ctr->refills, d_refills);
            const unsigned long long tk_end = wall_clock64();
            atomicAdd(&ca->ctr->wave_ticks, tk_end - tk_start); atomicAdd(&ca->ctr->drain_ticks, tk_end - (tk_exh ? tk_exh : tk_end)); atomicAdd(&ca->ctr->waves, 1ull);
            if (ca->timeline) {
                // HW_ID (wave, SIMD, CU, shader array, shader engine) and XCC_ID of the CU this wave ran on
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                unsigned long long *rec = ca->timeline + (size_t)(gtid >> 6) * 4;
                rec[0] = tk_start; rec[1] = tk_exh; rec[2] = tk_end; rec[3] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            }
        }
        if (n_over) atomicAdd(&ca->ctr->stack_overflow, n_over);
        if (gtid == 0 && !ca->no_ray_count) {
            if (count_c) atomicAdd(&ca->ctr->rays_closest, (unsigned long long)count_c);
            if (count_s) atomicAdd(&ca->ctr->rays_shadow, (unsigned long long)count_s);
        }
    }
}

// diagnostics: option "trace_timeline" = k >= 0 arms the k-th counting launch from now on; it writes [start, queue empty, end, hw id] per wave
static unsigned long long *timeline_for(tirt_ctx *c, int flags, int grid)
{
    if (c->timeline_arm < 0 || !(flags & TIRT_COUNT_NODES)) return nullptr;
    if (c->timeline_arm-- != 0) return nullptr;
    c->timeline_waves = grid * (TR_BLOCK / 64);
    if (c->timeline.ensure(sizeof(unsigned long long) * 4 * (size_t)c->timeline_waves)) { c->timeline_waves = 0; return nullptr; }
    if (hipMemsetAsync(c->timeline.p, 0, sizeof(unsigned long long) * 4 * (size_t)c->timeline_waves, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); c->timeline_waves = 0; return nullptr; }      // no timeline rather than a half-cleared one
    return c->timeline.as<unsigned long long>();
}

template <int MODE, bool COUNT, int KIND>
static int launch_trace_as(tirt_ctx *c, hipStream_t stream, const TraceArgs &a, dim3 g, dim3 b, size_t lds)
{
    // more than 64 KB of dynamic LDS per block (stacks + tree top): the opt-in is per kernel AND per device
    static size_t allowed[TIRT_MAX_DEVICES] = {};
    TIRT_REQUIRE(lds <= c->lds_optin, "k_trace: trace_lds_depth needs more LDS than the device has per block");
    const int dev = (c->device >= 0 && c->device < TIRT_MAX_DEVICES) ? c->device : 0;
    if (lds > allowed[dev] || c->device >= TIRT_MAX_DEVICES) {
        TIRT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace<MODE, COUNT, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed[dev] = lds;
    }
    hipLaunchKernelGGL((k_trace<MODE, COUNT, KIND>), g, b, lds, stream, a);
    return TIRT_OK;
}
template <int KIND>
static int launch_trace(tirt_ctx *c, hipStream_t stream, const TraceArgs &a, int flags, int grid)
{
    const bool exh = (flags & TIRT_TRAVERSE_EXHAUSTIVE) != 0, cnt = (flags & TIRT_COUNT_NODES) != 0;
    dim3 g(grid), b(TR_BLOCK);
    const size_t lds = trace_lds_bytes(a.lds_depth);
    if (exh && cnt) return launch_trace_as<TIRT_TRAVERSE_EXHAUSTIVE, true, KIND>(c, stream, a, g, b, lds);
    if (exh) return launch_trace_as<TIRT_TRAVERSE_EXHAUSTIVE, false, KIND>(c, stream, a, g, b, lds);
    if (cnt) return launch_trace_as<TIRT_TRAVERSE_ORDERED, true, KIND>(c, stream, a, g, b, lds);
    return launch_trace_as<TIRT_TRAVERSE_ORDERED, false, KIND>(c, stream, a, g, b, lds);
}

static int ensure_spill(tirt_ctx *c, DevBuf &spill, int stack_size, int &spill_depth)
{
    int cap = stack_size > 64 ? stack_size : 64;
    spill_depth = cap - (c->tr_lds_depth - 1);      // entry 0 of the LDS part is the sentinel
    if (spill_depth < 0) spill_depth = 0;
    spill_depth = (spill_depth + TR_PAGE - 1) / TR_PAGE * TR_PAGE;
    return spill.ensure(sizeof(int) * (size_t)(spill_depth > 0 ? spill_depth : 1) * TR_GRID_MAX * TR_BLOCK);
}
static void fill_tunables(const tirt_ctx *c, TraceArgs &a)
{ a.lds_depth = c->tr_lds_depth; a.refill_min = c->tr_refill_min; a.node_min = c->tr_node_min; a.slice_log2 = c->tr_slice_log2; a.slices_contig = c->slices_contiguous; }

// ---------------------------------------------------------------------------------------------
// Batch entry points (Debug-integrator style closest hit on caller-supplied rays)
// ---------------------------------------------------------------------------------------------
__global__ void k_split_rays(const float *rays, int nr, float *ox, float *oy, float *oz, float *dx, float *dy, float *dz)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr) return;
    const float *r = rays + (size_t)i * 6;
    ox[i] = r[0]; oy[i] = r[1]; oz[i] = r[2]; dx[i] = r[3]; dy[i] = r[4]; dz[i] = r[5];
}
__global__ void k_hit_attr(SceneView s, int nr, const float *ox, const float *oy, const float *oz, const float *dx, const float *dy,
                           const float *dz, const float4 *hit, float *out, float *out_t, int *out_prim)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr) return;
    float *o = out + (size_t)i * 13;
    const float4 hr = hit[i];
    o[0] = hr.x; out_t[i] = hr.x; out_prim[i] = __float_as_int(hr.w);
    HitAttr h; h.pos = h.gnor = h.nor = h.tex = V(0.0f, 0.0f, 0.0f);
    if (hr.x < INF_VALUE) h = hit_attributes(s, V(ox[i], oy[i], oz[i]), V(dx[i], dy[i], dz[i]), __float_as_int(hr.w), hr.x, hr.y, hr.z);
    else { h.gnor = normalized(h.gnor); h.nor = normalized(h.nor); }     // reference normalises (0,0,0) on a miss
    o[1] = h.pos.x; o[2] = h.pos.y; o[3] = h.pos.z;
    o[4] = h.gnor.x; o[5] = h.gnor.y; o[6] = h.gnor.z;
    o[7] = h.nor.x; o[8] = h.nor.y; o[9] = h.nor.z;
    o[10] = h.tex.x; o[11] = h.tex.y; o[12] = h.tex.z;
}

int launch_trace_batch(tirt_ctx *c, const float *rays, int nr, int stack_size, int flags, bool shadow,
                       float *out_f, int32_t *out_prim, int32_t *counts)
{
    TIRT_REQUIRE(c->built, "trace: LBVH not built");
    TIRT_REQUIRE(nr >= 0, "trace: nr < 0");
    if (nr == 0) return TIRT_OK;
    if (ensure_counters(c)) return TIRT_ERR_HIP;
    hipStream_t st = c->stream;
    if (c->tr_rays.ensure(sizeof(float) * 6 * (size_t)nr)) return TIRT_ERR_HIP;
    if (c->tr_out.ensure(sizeof(float) * (6 + 4 + 1 + 13) * (size_t)nr + 64)) return TIRT_ERR_HIP;
    if (c->tr_prim.ensure(sizeof(int) * (size_t)nr)) return TIRT_ERR_HIP;
    if (c->tr_counts.ensure(sizeof(int2) * (size_t)nr)) return TIRT_ERR_HIP;
    int spill_depth;
    if (ensure_spill(c, c->spill, stack_size, spill_depth)) return TIRT_ERR_HIP;
    float *base = c->tr_out.as<float>();
    float *ox = base, *oy = ox + nr, *oz = oy + nr, *dx = oz + nr, *dy = dx + nr, *dz = dy + nr;
    float4 *hit = (float4 *)(((uintptr_t)(dz + nr) + 15) & ~(uintptr_t)15);
    float *ht = (float *)(hit + nr), *attr = ht + nr;
    TIRT_HIP(hipMemcpyAsync(c->tr_rays.p, rays, sizeof(float) * 6 * (size_t)nr, hipMemcpyHostToDevice, st));
    const int B = 256;
    hipLaunchKernelGGL(k_split_rays, dim3((nr + B - 1) / B), dim3(B), 0, st, c->tr_rays.as<float>(), nr, ox, oy, oz, dx, dy, dz);
    TraceArgs a = {};
    a.bvh = bvh_view(c);
    a.ox = ox; a.oy = oy; a.oz = oz; a.dx = dx; a.dy = dy; a.dz = dz;
    a.count_ptr = nullptr; a.count_fixed = nr;
    a.hit = hit;
    a.spill = c->spill.as<int>(); a.spill_depth = spill_depth;
    a.ctr = c->dev_counters.as<DevCounters>();
    a.per_ray_counts = (flags & TIRT_COUNT_NODES) ? c->tr_counts.as<int2>() : nullptr;
    if (c->counters_mem.ensure(sizeof(int) * TR_FETCH_STRIDE * TR_FETCH_LINES)) return TIRT_ERR_HIP;
    TIRT_HIP(hipMemsetAsync(c->counters_mem.p, 0, sizeof(int) * TR_FETCH_STRIDE * TR_FETCH_LINES, st));
    a.fetch = c->counters_mem.as<int>();
    fill_tunables(c, a);
    int grid = (nr + TR_BLOCK - 1) / TR_BLOCK; if (grid > c->tr_grid) grid = c->tr_grid;
    a.timeline = timeline_for(c, flags, grid);
    if (int rc = launch_trace<KIND_CLOSEST>(c, st, a, flags, grid)) return rc;
    hipLaunchKernelGGL(k_hit_attr, dim3((nr + B - 1) / B), dim3(B), 0, st, scene_view(c), nr, ox, oy, oz, dx, dy, dz, hit, attr, ht,
                       c->tr_prim.as<int>());
    if (!shadow) TIRT_HIP(hipMemcpyAsync(out_f, attr, sizeof(float) * 13 * (size_t)nr, hipMemcpyDeviceToHost, st));
    else TIRT_HIP(hipMemcpyAsync(out_f, ht, sizeof(float) * (size_t)nr, hipMemcpyDeviceToHost, st));
    TIRT_HIP(hipMemcpyAsync(out_prim, c->tr_prim.p, sizeof(int) * (size_t)nr, hipMemcpyDeviceToHost, st));
    if (counts && (flags & TIRT_COUNT_NODES))
        TIRT_HIP(hipMemcpyAsync(counts, c->tr_counts.p, sizeof(int2) * (size_t)nr, hipMemcpyDeviceToHost, st));
    TIRT_HIP(hipStreamSynchronize(st));
    TIRT_HIP(hipGetLastError());
    return TIRT_OK;
}

// The buffers trace_arrays needs on a lane (lane < 0: the main stream) -- the paged tail of the traversal stacks, sized by the integrator's
// stack size (2 GB per lane at bdpt_stack_size 1024), and the ray-fetch cursors -- allocated up front: a caller that sizes its batches to the free
// memory (bdpt_render) calls this BEFORE it measures, and a failed allocation ends the call before anything has been blended into the film.
int trace_arrays_prepare(tirt_ctx *c, int lane)
{
    DevBuf &spill = lane < 0 ? c->spill : c->lanes[lane].spill;
    DevBuf &fetch = lane < 0 ? c->counters_mem : c->lanes[lane].counters_mem;
    int spill_depth;
    if (ensure_spill(c, spill, c->bdpt_stack, spill_depth)) return TIRT_ERR_HIP;
    if (fetch.ensure(sizeof(int) * TR_FETCH_STRIDE * TR_FETCH_LINES)) return TIRT_ERR_HIP;
    return TIRT_OK;
}

// Closest hits (expect == nullptr) or bounded connection queries of `count` rays held in device arrays, hit records to
// `hit` -- the traversal service of the BDPT wavefront (tirt_bdpt.hip).  Ordered traversal, on the main stream (lane < 0) or on
// the stream of a render lane with that lane's ray-fetch cursors and spill buffer (two BDPT batches in flight).
int trace_arrays(tirt_ctx *c, const float *ox, const float *oy, const float *oz, const float *dx, const float *dy, const float *dz,
                 int count, const int *count_ptr, float4 *hit, const int *expect, const float *bound, bool count_rays, int lane,
                 const float4 *ray4, bool query, const int *ray_index)
{
    if (count <= 0) return TIRT_OK;
    hipStream_t st = lane < 0 ? c->stream : c->lanes[lane].stream;
    DevBuf &spill = lane < 0 ? c->spill : c->lanes[lane].spill;
    // (the lanes' counter buffers hold the per-bounce cursors of the path tracer as well: ensure() keeps a larger one)
    DevBuf &fetch = lane < 0 ? c->counters_mem : c->lanes[lane].counters_mem;
    int spill_depth;
    if (ensure_spill(c, spill, c->bdpt_stack, spill_depth)) return TIRT_ERR_HIP;      // (the integrator's stack_size: option "bdpt_stack_size")
    if (fetch.ensure(sizeof(int) * TR_FETCH_STRIDE * TR_FETCH_LINES)) return TIRT_ERR_HIP;
    TIRT_HIP(hipMemsetAsync(fetch.p, 0, sizeof(int) * TR_FETCH_STRIDE * TR_FETCH_LINES, st));
    TraceArgs a = {};
    a.bvh = bvh_view(c);
    a.ox = ox; a.oy = oy; a.oz = oz; a.dx = dx; a.dy = dy; a.dz = dz; a.ray4 = ray4; a.ray_index = ray_index;
    a.count_ptr = count_ptr; a.count_fixed = count; a.hit = hit;      // count: the capacity when count_ptr is given (sizes the grid)
    a.sprim = expect; a.sdist = bound;
    a.spill = spill.as<int>(); a.spill_depth = spill_depth;
    a.ctr = c->dev_counters.as<DevCounters>(); a.per_ray_counts = nullptr; a.no_ray_count = count_rays ? 0 : 1;
    a.fetch = fetch.as<int>();
    fill_tunables(c, a);
    const int grid_cap = c->tr_grid_alone;          // also with two BDPT batches in flight: the other batch mostly runs its vertex / connection kernels (config 5: 384 blocks 2 047, 512 blocks 2 145 Mrays/s)
    int grid = (count + TR_BLOCK - 1) / TR_BLOCK; if (grid > grid_cap) grid = grid_cap;
    return (expect || query) ? launch_trace<KIND_QUERY>(c, st, a, 0, grid) : launch_trace<KIND_CLOSEST>(c, st, a, 0, grid);
}

// ---------------------------------------------------------------------------------------------
// Wavefront PT_RGB
// ---------------------------------------------------------------------------------------------
__global__ void k_generate(PathSoA ps, CameraView cam, TileMap tm, int P, int S, uint32_t frame_begin, uint32_t seed,
                           DevCounters *ctr)
{
    // Camera rays.  Only the direction is stored: the origin is the eye for every path and the rest of the
    // bounce-0 state is constant (throughput 1, radiance 0, pdf 1, specular flag set, path id = index), which
    // k_trace / k_shade of bounce 0 know without reading 48 bytes per path back from HBM.
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    int f, k; slot_to_frame_pixel(tm, P, s, f, k);
    int p = local_to_pixel(tm, k);
    int i = p / tm.H, j = p - i * tm.H;
    uint32_t frame = frame_begin + (uint32_t)f;
    float jx = 0.0f, jy = 0.0f;
    if (frame != 0) {                                    // Camera.py:135-137
        jx = tm_rand(seed, (uint32_t)p, frame, TM_DIM_JX) - 0.5f;
        jy = tm_rand(seed, (uint32_t)p, frame, TM_DIM_JY) - 0.5f;
    }
    v3 d = camera_ray_direction(cam, i, j, jx, jy);
    __builtin_nontemporal_store(d.x, &ps.dx[s]); __builtin_nontemporal_store(d.y, &ps.dy[s]); __builtin_nontemporal_store(d.z, &ps.dz[s]);      // (a stream: k_trace reads it once)
    if (s == 0) atomicAdd(&ctr->paths, (unsigned long long)S);
}

// Block size of k_shade.  The survivors / shadow rays of a block are appended with ONE 64-bit atomic per
// block and round (low word: next-bounce paths, high word: shadow rays): same-address device atomics
// retire at ~11 ns each on MI355X, so one atomic per wave and queue (2 x 524k per 33.5 M paths)
// bounded the kernel at ~6 ms; per 256-thread block it is 131k.  (512-thread blocks halve that again but
// measured 4 % slower end to end: a 4-wave block needs 112 VGPRs per SIMD and fits next to the resident
// k_trace waves of another lane's batch, an 8-wave block does not.)
#ifndef SH_BLOCK
#define SH_BLOCK 256
#endif
#ifndef SH_MIN_WAVES
#define SH_MIN_WAVES 5
#endif
// The 78 array pointers of the path state are the first kernel argument and are never read from it directly: each of the three places
// that needs some of them (a path's state in, the survivor's state out, the shadow ray out) reads those from the kernel-argument segment in
// one batch of scalar loads -- as k_trace does (TR_COLD), and for the same reason: kept in SGPRs across the loop they overflow the scalar
// register file and come back through v_readlane, 700 VALU instructions of the 4 476 in this kernel.
struct ShadeArgs { PathState ps; PathSoA in, out; };
typedef const __attribute__((address_space(4))) ShadeArgs *cold_shade_t;
#define SH_COLD(ca) cold_shade_t ca = (cold_shade_t)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(ca))
#define SH_LD(x) __builtin_nontemporal_load(&(x))
#define SH_ST(x, v) __builtin_nontemporal_store((v), &(x))
__global__ __launch_bounds__(SH_BLOCK, SH_MIN_WAVES) void k_shade(ShadeArgs paths_in_kernarg_segment, SceneView sc, TileMap tm, int P,
                                                   uint32_t frame_begin, uint32_t seed, int bounce, int last_bounce,
                                                   const int *count_ptr, int count_fixed, unsigned long long *append_ctr,
                                                   DevCounters *ctr, v3 eye)
{
    __shared__ unsigned s_wcnt[2][SH_BLOCK / 64];
    __shared__ unsigned long long s_base[2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int count = count_ptr ? *count_ptr : count_fixed;
    const int total = gridDim.x * blockDim.x;
    const int rounds = (count + total - 1) / total;
    unsigned long long n_shaded = 0;
    for (int it = 0; it < rounds; it++) {
        const int q = it * total + blockIdx.x * blockDim.x + threadIdx.x;
        bool live = q < count, want_next = false, want_shadow = false;
        int slot = 0;
        v3 radiance = V(0.0f, 0.0f, 0.0f), next_o = radiance, next_d = radiance, next_thr = radiance;
        v3 sh_o = radiance, sh_d = radiance, sh_c = radiance;
        float next_pdf = 0.0f, sh_dist = 0.0f; int next_spec = 0, sh_expect = -2;
        if (live) {
            const bool first = bounce == 0;          // camera rays: constant state, see k_generate
            SH_COLD(ca);
            const int *const c_slot = ca->in.slot; const uint32_t *const c_flags = ca->in.flags; const float4 *const c_hit = ca->ps.hit;
            const float *const c_ox = ca->in.ox, *const c_oy = ca->in.oy, *const c_oz = ca->in.oz, *const c_dx = ca->in.dx, *const c_dy = ca->in.dy, *const c_dz = ca->in.dz;
            const float *const c_tr = ca->in.tr, *const c_tg = ca->in.tg, *const c_tb = ca->in.tb, *const c_rr = ca->in.rr, *const c_rg = ca->in.rg, *const c_rb = ca->in.rb;
            const float *const c_pdf = ca->in.brdf_pdf;
            asm volatile("" :: "s"(c_slot), "s"(c_flags), "s"(c_hit), "s"(c_ox), "s"(c_oy), "s"(c_oz), "s"(c_dx), "s"(c_dy), "s"(c_dz), "s"(c_tr), "s"(c_tg), "s"(c_tb),
                         "s"(c_rr), "s"(c_rg), "s"(c_rb), "s"(c_pdf));
            slot = first ? q : SH_LD(c_slot[q]);
            const v3 origin = first ? eye : V(SH_LD(c_ox[q]), SH_LD(c_oy[q]), SH_LD(c_oz[q]));
            const v3 direction = V(SH_LD(c_dx[q]), SH_LD(c_dy[q]), SH_LD(c_dz[q]));
            typedef float f4nt__ __attribute__((ext_vector_type(4)));
            const f4nt__ hnt__ = __builtin_nontemporal_load((const f4nt__ *)&c_hit[q]);
            const float4 hrec = make_float4(hnt__.x, hnt__.y, hnt__.z, hnt__.w);
            v3 throughout = first ? V(1.0f, 1.0f, 1.0f) : V(SH_LD(c_tr[q]), SH_LD(c_tg[q]), SH_LD(c_tb[q]));
            radiance = first ? V(0.0f, 0.0f, 0.0f) : V(SH_LD(c_rr[q]), SH_LD(c_rg[q]), SH_LD(c_rb[q]));
            float brdf_pdf = first ? 1.0f : SH_LD(c_pdf[q]);
            int perfect_spec = first ? 1 : (int)(SH_LD(c_flags[q]) & 1u);
            ShadeStep ss;
            shade_path(sc, tm, P, frame_begin, seed, bounce, last_bounce, slot, origin, direction, hrec, throughout, radiance, brdf_pdf, perfect_spec, ss);
            want_next = ss.want_next; want_shadow = ss.want_shadow; if (ss.shaded) n_shaded++;
            next_o = ss.next_o; next_d = ss.next_d; next_thr = ss.next_thr; next_pdf = ss.next_pdf; next_spec = ss.next_spec;
            sh_o = ss.sh_o; sh_d = ss.sh_d; sh_c = ss.sh_c; sh_expect = ss.sh_expect; sh_dist = ss.sh_dist;
        }
        // dense compaction: survivors go to consecutive indices of the other PathSoA, finished
        // paths deposit their radiance in the per-path final array, shadow rays get their own
        // dense list with the address their contribution must be added to
        const int par = it & 1;
        const unsigned long long nmask = __ballot(want_next), smask = __ballot(want_shadow);
        if (lane == 0) s_wcnt[par][wid] = (unsigned)__popcll(nmask) | ((unsigned)__popcll(smask) << 16);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tn = 0, tsd = 0;
#pragma unroll
            for (int w = 0; w < SH_BLOCK / 64; w++) { tn += s_wcnt[par][w] & 0xffffu; tsd += s_wcnt[par][w] >> 16; }
            s_base[par] = (tn | tsd) ? atomicAdd(append_ctr, (unsigned long long)tn | ((unsigned long long)tsd << 32)) : 0ull;
        }
        __syncthreads();
        unsigned pn = 0, psd = 0;
#pragma unroll
        for (int w = 0; w < SH_BLOCK / 64; w++) if (w < wid) { pn += s_wcnt[par][w] & 0xffffu; psd += s_wcnt[par][w] >> 16; }
        const unsigned long long bb = s_base[par];
        const int qn = (int)(unsigned)(bb & 0xffffffffull) + (int)pn + __popcll(nmask & lt_mask);
        const int qs = (int)(unsigned)(bb >> 32) + (int)psd + __popcll(smask & lt_mask);
        if (want_next) {
            SH_COLD(ca);
            float *const o_ox = ca->out.ox, *const o_oy = ca->out.oy, *const o_oz = ca->out.oz, *const o_dx = ca->out.dx, *const o_dy = ca->out.dy, *const o_dz = ca->out.dz;
            float *const o_tr = ca->out.tr, *const o_tg = ca->out.tg, *const o_tb = ca->out.tb, *const o_rr = ca->out.rr, *const o_rg = ca->out.rg, *const o_rb = ca->out.rb;
            float *const o_pdf = ca->out.brdf_pdf; uint32_t *const o_flags = ca->out.flags; int *const o_slot = ca->out.slot;
            asm volatile("" :: "s"(o_ox), "s"(o_oy), "s"(o_oz), "s"(o_dx), "s"(o_dy), "s"(o_dz), "s"(o_tr), "s"(o_tg), "s"(o_tb), "s"(o_rr), "s"(o_rg), "s"(o_rb),
                         "s"(o_pdf), "s"(o_flags), "s"(o_slot));
            SH_ST(o_ox[qn], next_o.x); SH_ST(o_oy[qn], next_o.y); SH_ST(o_oz[qn], next_o.z);
            SH_ST(o_dx[qn], next_d.x); SH_ST(o_dy[qn], next_d.y); SH_ST(o_dz[qn], next_d.z);
            SH_ST(o_tr[qn], next_thr.x); SH_ST(o_tg[qn], next_thr.y); SH_ST(o_tb[qn], next_thr.z);
            SH_ST(o_rr[qn], radiance.x); SH_ST(o_rg[qn], radiance.y); SH_ST(o_rb[qn], radiance.z);
            SH_ST(o_pdf[qn], next_pdf); SH_ST(o_flags[qn], (uint32_t)next_spec); SH_ST(o_slot[qn], slot);
        } else if (live) {
            SH_COLD(ca);
            float *const f_r = ca->ps.fr, *const f_g = ca->ps.fg, *const f_b = ca->ps.fb;
            asm volatile("" :: "s"(f_r), "s"(f_g), "s"(f_b));
            SH_ST(f_r[slot], radiance.x); SH_ST(f_g[slot], radiance.y); SH_ST(f_b[slot], radiance.z);
        }
        if (want_shadow) {
            SH_COLD(ca);
            float *const s_ox = ca->ps.sox, *const s_oy = ca->ps.soy, *const s_oz = ca->ps.soz, *const s_dx = ca->ps.sdx, *const s_dy = ca->ps.sdy, *const s_dz = ca->ps.sdz;
            float *const s_cr = ca->ps.scr, *const s_cg = ca->ps.scg, *const s_cb = ca->ps.scb, *const s_dist = ca->ps.sdist;
            int *const s_prim = ca->ps.sprim, *const s_dst = ca->ps.sdst;
            asm volatile("" :: "s"(s_ox), "s"(s_oy), "s"(s_oz), "s"(s_dx), "s"(s_dy), "s"(s_dz), "s"(s_cr), "s"(s_cg), "s"(s_cb), "s"(s_dist), "s"(s_prim), "s"(s_dst));
            SH_ST(s_ox[qs], sh_o.x); SH_ST(s_oy[qs], sh_o.y); SH_ST(s_oz[qs], sh_o.z);
            SH_ST(s_dx[qs], sh_d.x); SH_ST(s_dy[qs], sh_d.y); SH_ST(s_dz[qs], sh_d.z);
            SH_ST(s_cr[qs], sh_c.x); SH_ST(s_cg[qs], sh_c.y); SH_ST(s_cb[qs], sh_c.z);
            SH_ST(s_prim[qs], sh_expect); SH_ST(s_dist[qs], sh_dist);
            SH_ST(s_dst[qs], want_next ? qn : ~slot);
        }
    }
    // statistics: one global atomic per block (through LDS), not per wave -- same-address atomics retire at ~11 ns
    __shared__ unsigned long long s_shaded;
    if (threadIdx.x == 0) s_shaded = 0ull;
    __syncthreads();
    n_shaded = wave_sum(n_shaded);
    if ((threadIdx.x & 63) == 0 && n_shaded) atomicAdd(&s_shaded, n_shaded);
    __syncthreads();
    if (threadIdx.x == 0 && s_shaded) atomicAdd(&ctr->shaded, s_shaded);
}

// integrator/PT_RGB.py:134-136, frames applied in order
__global__ void k_film(PathState ps, TileMap tm, int P, int F, uint32_t frame_begin, float *hdr)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    int p = local_to_pixel(tm, k);
    float *px = hdr + (size_t)p * 3;
    float r = px[0], g = px[1], b = px[2];
    for (int f = 0; f < F; f++) {
        int s = frame_pixel_to_slot(tm, P, f, k);
        float frame = (float)(int)(frame_begin + (uint32_t)f);
        float coff = 1.0f / (frame + 1.0f);
        r = __builtin_nontemporal_load(&ps.fr[s]) * coff + r * (1.0f - coff);
        g = __builtin_nontemporal_load(&ps.fg[s]) * coff + g * (1.0f - coff);
        b = __builtin_nontemporal_load(&ps.fb[s]) * coff + b * (1.0f - coff);
    }
    px[0] = r; px[1] = g; px[2] = b;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// PT_Spec (integrator/PT_Spec.py:181-279) on the same wavefront: k_generate and the traversal launches are PT_RGB's, the
// shading kernel and the film update are spectral.  Path state in the same PathSoA arrays: throughput[4] in (tr, tg, tb, brdf_pdf),
// radiance[4] in (rr, rg, rb, flags) -- PT_Spec carries no brdf_pdf and no perfect_spec flag from one bounce to the next (its
// `perfect_spec` is reset to 1 at the top of every loop iteration, PT_Spec.py:214, so the emission hit is never MIS-weighted) --;
// the hero wavelength is not stored: Lambda = 360 + 100 * rand(pixel, frame, TM_DIM_SPEC_LAMBDA) is recomputed where it is needed.
// Reference behaviours kept: the NEE sample is tinted with the colour of the surface that was HIT (`light_tint` of :213), not with
// the light's emission; Disney.evaluate_pdf for the continuation is called with (N, V = next_dir, L = -direction) (:251).
__global__ __launch_bounds__(SH_BLOCK, 4) void k_shade_spec(ShadeArgs paths_in_kernarg_segment, SceneView sc, SpecView sp, TileMap tm, int P,
                                                   uint32_t frame_begin, uint32_t seed, int bounce, int last_bounce,
                                                   const int *count_ptr, int count_fixed, unsigned long long *append_ctr,
                                                   DevCounters *ctr, v3 eye)
{
    __shared__ unsigned s_wcnt[2][SH_BLOCK / 64];
    __shared__ unsigned long long s_base[2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int count = count_ptr ? *count_ptr : count_fixed;
    const int total = gridDim.x * blockDim.x;
    const int rounds = (count + total - 1) / total;
    unsigned long long n_shaded = 0;
    for (int it = 0; it < rounds; it++) {
        const int q = it * total + blockIdx.x * blockDim.x + threadIdx.x;
        bool live = q < count, want_next = false, want_shadow = false;
        int slot = 0;
        f4s radiance = f4_set(0.0f), next_thr = radiance, sh_c = radiance;
        v3 next_o = V(0.0f, 0.0f, 0.0f), next_d = next_o, sh_o = next_o, sh_d = next_o;
        float sh_dist = 0.0f; int sh_expect = -2;
        if (live) {
            const bool first = bounce == 0;
            SH_COLD(ca);          // (the path-state pointers: read here, in one batch of scalar loads -- see k_shade)
            const int *const c_slot = ca->in.slot; const float4 *const c_hit = ca->ps.hit;
            const float *const c_ox = ca->in.ox, *const c_oy = ca->in.oy, *const c_oz = ca->in.oz, *const c_dx = ca->in.dx, *const c_dy = ca->in.dy, *const c_dz = ca->in.dz;
            const float *const c_tr = ca->in.tr, *const c_tg = ca->in.tg, *const c_tb = ca->in.tb, *const c_rr = ca->in.rr, *const c_rg = ca->in.rg, *const c_rb = ca->in.rb;
            const float *const c_tw = ca->in.brdf_pdf, *const c_rw = (const float *)ca->in.flags;
            asm volatile("" :: "s"(c_slot), "s"(c_hit), "s"(c_ox), "s"(c_oy), "s"(c_oz), "s"(c_dx), "s"(c_dy), "s"(c_dz), "s"(c_tr), "s"(c_tg), "s"(c_tb),
                         "s"(c_rr), "s"(c_rg), "s"(c_rb), "s"(c_tw), "s"(c_rw));
            slot = first ? q : c_slot[q];
            int f, k; slot_to_frame_pixel(tm, P, slot, f, k);
            const uint32_t pixel = (uint32_t)local_to_pixel(tm, k);
            const uint32_t frame = frame_begin + (uint32_t)f;
            const uint32_t dim0 = TM_DIM_BOUNCE0 + TM_DIMS_PER_BOUNCE * (uint32_t)bounce;
            const float Lambda = HERO_LAMBDA_MIN + HERO_LAMBDA_STEP * tm_rand(seed, pixel, frame, TM_DIM_SPEC_LAMBDA);     // PT_Spec.py:191
            const v3 origin = first ? eye : V(c_ox[q], c_oy[q], c_oz[q]);
            const v3 direction = V(c_dx[q], c_dy[q], c_dz[q]);
            typedef float f4nt__ __attribute__((ext_vector_type(4)));
            const f4nt__ hnt__ = __builtin_nontemporal_load((const f4nt__ *)&c_hit[q]);
            const float4 hrec = make_float4(hnt__.x, hnt__.y, hnt__.z, hnt__.w);
            const float t = hrec.x;
            f4s throughout = f4_set(1.0f);
            if (!first) {
                throughout.v[0] = c_tr[q]; throughout.v[1] = c_tg[q]; throughout.v[2] = c_tb[q]; throughout.v[3] = c_tw[q];
                radiance.v[0] = c_rr[q]; radiance.v[1] = c_rg[q]; radiance.v[2] = c_rb[q]; radiance.v[3] = c_rw[q];
            }
            const f4s light_rad = hero_sample(sp.spd[0], Lambda);                                   // :212
            if (t < INF_VALUE) {
                const int prim_id = __float_as_int(hrec.w);
                int mat_id;
                const HitAttr h = hit_attributes_rec(sc.shade_rec, origin, direction, prim_id, t, hrec.y, hrec.z, mat_id);
                const v3 normal = h.nor;
                const v3 fnormal = normal * signf(dot(-direction, h.gnor));
                const float *m = sc.material + (size_t)mat_id * MAT_VEC;
                const v3 mat_color = V(m[2], m[3], m[4]);
                const int mat_type = (int)m[0];
                const f4s light_tint = emission_to_rad(sp, mat_color, Lambda);                      // :213 (the HIT material's colour)
                if (mat_type == MAT_LIGHT) {                                                        // :216-226
                    const float fCosTheta = dot(direction, normal);
                    if (fCosTheta < 0.0f) radiance = radiance + ((throughout * light_rad) * light_tint);
                } else {
                    n_shaded++;
                    const f4s reflect_spec = get_spec_power(sp, m, Lambda);
                    v3 next_dir; float f_or_b = 1.0f, brdf = 1.0f, brdf_pdf = 1.0f;
                    if (mat_type == MAT_GLASS) {                                                    // :234-238
                        const int index = (int)(tm_rand(seed, pixel, frame, dim0 + TM_SLOT_HERO) * (float)HERO_N);
                        const float rnd_lambda = Lambda + (float)index * HERO_LAMBDA_STEP;
                        next_dir = glass_sample_lambda(direction, normal, rnd_lambda, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_GLASS), f_or_b);
                    } else {
                        if (sc.light_count > 0) {                                                   // :240-249, Scene.sample_li
                            int lidx = (int)(tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LIGHT) * (float)sc.light_count);
                            if (lidx >= sc.light_count) lidx = sc.light_count - 1;
                            const int light_prim = sc.light[lidx];
                            const float ra = tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LA);
                            const float rb = tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LB);
                            v3 light_pos, light_normal;
                            get_prim_random_point_normal(sc, light_prim, ra, rb, light_pos, light_normal);
                            const float light_area = get_prim_area(sc, light_prim);
                            float light_choice_pdf = 1.0f / ((float)sc.light_count * light_area);
                            light_normal = normalized(light_normal);
                            v3 light_dir = h.pos - light_pos;
                            const float light_dist = norm(light_dir);
                            light_dir = light_dir / light_dist;
                            (void)light_shape_visible(sc, light_prim, light_dir, light_normal, light_dist, light_choice_pdf);      // only its pdf is used by :245
                            const float NdotL_surface = dot(fnormal, light_dir);
                            const float NdotL_light = dot(light_normal, light_dir);
                            if ((NdotL_surface < 0.0f) & (NdotL_light > 0.0f)) {
                                want_shadow = true;
                                float e_pdf;
                                const float e_brdf = disney_evaluate_pdf(m, fnormal, -direction, -light_dir, e_pdf);
                                const float light_pdf = light_dist * light_dist * light_choice_pdf / NdotL_light;
                                f4s c = f4_set(0.0f);
                                int expect = -2;
                                if (e_pdf > 0.0f) {
                                    const float w = power_heuristic(light_pdf, e_pdf) / maxf(0.0001f, light_pdf);
                                    c = light_rad * w;
                                    c = c * light_tint;
                                    c = c * throughout;
                                    c = c * reflect_spec;
                                    c = c * e_brdf;
                                    c = c * absf(NdotL_surface);
                                    expect = prim_id;
                                }
                                sh_o = light_pos; sh_d = light_dir; sh_c = c; sh_expect = expect; sh_dist = light_dist;
                            }
                        }
                        next_dir = disney_sample(m, direction, fnormal, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LOBE),
                                                 tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R1), tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R2));
                        f_or_b = 1.0f;
                        brdf = disney_evaluate_pdf(m, fnormal, next_dir, -direction, brdf_pdf);     // (N, V = next_dir, L = -direction): :251
                        brdf *= absf(dot(normal, next_dir));
                    }
                    const v3 next_origin = offset_ray(h.pos, fnormal * signf(f_or_b));             // :254
                    if ((brdf_pdf > 0.0f) & (maxf(throughout.v[2], maxf(throughout.v[0], throughout.v[1])) > 0.0f)) {      // :256-258
                        throughout = throughout * ((reflect_spec * brdf) / brdf_pdf);
                        want_next = !last_bounce;
                        next_o = next_origin; next_d = next_dir; next_thr = throughout;
                    }
                }
            } else {                                                                                // :262-270: the analytic sky
                const float dis = tm_sqrt(direction.x * direction.x + direction.z * direction.z);
                const float beta = tm_atan2(direction.y, dis);
                const float gamma = tm_acos(dot(direction, V(sp.sun_dir[0], sp.sun_dir[1], sp.sun_dir[2])));
                const float theta = clampf(0.5f * PI_SCENE - beta, 0.0f, 0.5f * PI_SCENE);
                f4s ibl;
                for (int w = 0; w < HERO_N; w++) ibl.v[w] = sky_radiance(sp, theta, gamma, Lambda + (float)w * HERO_LAMBDA_STEP);
                radiance = radiance + ((throughout * ibl) * light_rad);
            }
        }
        const int par = it & 1;
        const unsigned long long nmask = __ballot(want_next), smask = __ballot(want_shadow);
        if (lane == 0) s_wcnt[par][wid] = (unsigned)__popcll(nmask) | ((unsigned)__popcll(smask) << 16);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tn = 0, tsd = 0;
#pragma unroll
            for (int w = 0; w < SH_BLOCK / 64; w++) { tn += s_wcnt[par][w] & 0xffffu; tsd += s_wcnt[par][w] >> 16; }
            s_base[par] = (tn | tsd) ? atomicAdd(append_ctr, (unsigned long long)tn | ((unsigned long long)tsd << 32)) : 0ull;
        }
        __syncthreads();
        unsigned pn = 0, psd = 0;
#pragma unroll
        for (int w = 0; w < SH_BLOCK / 64; w++) if (w < wid) { pn += s_wcnt[par][w] & 0xffffu; psd += s_wcnt[par][w] >> 16; }
        const unsigned long long bb = s_base[par];
        const int qn = (int)(unsigned)(bb & 0xffffffffull) + (int)pn + __popcll(nmask & lt_mask);
        const int qs = (int)(unsigned)(bb >> 32) + (int)psd + __popcll(smask & lt_mask);
        if (want_next) {
            SH_COLD(ca);
            float *const o_ox = ca->out.ox, *const o_oy = ca->out.oy, *const o_oz = ca->out.oz, *const o_dx = ca->out.dx, *const o_dy = ca->out.dy, *const o_dz = ca->out.dz;
            float *const o_tr = ca->out.tr, *const o_tg = ca->out.tg, *const o_tb = ca->out.tb, *const o_rr = ca->out.rr, *const o_rg = ca->out.rg, *const o_rb = ca->out.rb;
            float *const o_tw = ca->out.brdf_pdf, *const o_rw = (float *)ca->out.flags; int *const o_slot = ca->out.slot;
            asm volatile("" :: "s"(o_ox), "s"(o_oy), "s"(o_oz), "s"(o_dx), "s"(o_dy), "s"(o_dz), "s"(o_tr), "s"(o_tg), "s"(o_tb), "s"(o_rr), "s"(o_rg), "s"(o_rb),
                         "s"(o_tw), "s"(o_rw), "s"(o_slot));
            o_ox[qn] = next_o.x; o_oy[qn] = next_o.y; o_oz[qn] = next_o.z;
            o_dx[qn] = next_d.x; o_dy[qn] = next_d.y; o_dz[qn] = next_d.z;
            o_tr[qn] = next_thr.v[0]; o_tg[qn] = next_thr.v[1]; o_tb[qn] = next_thr.v[2]; o_tw[qn] = next_thr.v[3];
            o_rr[qn] = radiance.v[0]; o_rg[qn] = radiance.v[1]; o_rb[qn] = radiance.v[2]; o_rw[qn] = radiance.v[3];
            o_slot[qn] = slot;
        } else if (live) {
            SH_COLD(ca);
            float *const f_r = ca->ps.fr, *const f_g = ca->ps.fg, *const f_b = ca->ps.fb, *const f_w = ca->ps.fw;
            asm volatile("" :: "s"(f_r), "s"(f_g), "s"(f_b), "s"(f_w));
            f_r[slot] = radiance.v[0]; f_g[slot] = radiance.v[1]; f_b[slot] = radiance.v[2]; f_w[slot] = radiance.v[3];
        }
        if (want_shadow) {
            SH_COLD(ca);
            float *const s_ox = ca->ps.sox, *const s_oy = ca->ps.soy, *const s_oz = ca->ps.soz, *const s_dx = ca->ps.sdx, *const s_dy = ca->ps.sdy, *const s_dz = ca->ps.sdz;
            float *const s_cr = ca->ps.scr, *const s_cg = ca->ps.scg, *const s_cb = ca->ps.scb, *const s_cw = ca->ps.scw, *const s_dist = ca->ps.sdist;
            int *const s_prim = ca->ps.sprim, *const s_dst = ca->ps.sdst;
            asm volatile("" :: "s"(s_ox), "s"(s_oy), "s"(s_oz), "s"(s_dx), "s"(s_dy), "s"(s_dz), "s"(s_cr), "s"(s_cg), "s"(s_cb), "s"(s_cw), "s"(s_dist), "s"(s_prim), "s"(s_dst));
            s_ox[qs] = sh_o.x; s_oy[qs] = sh_o.y; s_oz[qs] = sh_o.z;
            s_dx[qs] = sh_d.x; s_dy[qs] = sh_d.y; s_dz[qs] = sh_d.z;
            s_cr[qs] = sh_c.v[0]; s_cg[qs] = sh_c.v[1]; s_cb[qs] = sh_c.v[2]; s_cw[qs] = sh_c.v[3];
            s_prim[qs] = sh_expect; s_dist[qs] = sh_dist;
            s_dst[qs] = want_next ? qn : ~slot;
        }
    }
    __shared__ unsigned long long s_shaded;
    if (threadIdx.x == 0) s_shaded = 0ull;
    __syncthreads();
    n_shaded = wave_sum(n_shaded);
    if ((threadIdx.x & 63) == 0 && n_shaded) atomicAdd(&s_shaded, n_shaded);
    __syncthreads();
    if (threadIdx.x == 0 && s_shaded) atomicAdd(&ctr->shaded, s_shaded);
}

// integrator/PT_Spec.py:141-158 (AddSplat) + :273-274, frames applied in order
__global__ void k_film_spec(PathState ps, const float *fw, SpecView sp, TileMap tm, int P, int F, uint32_t frame_begin, uint32_t seed, float *hdr)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const int p = local_to_pixel(tm, k);
    float *px = hdr + (size_t)p * 3;
    float r = px[0], g = px[1], b = px[2];
    for (int f = 0; f < F; f++) {
        const int s = frame_pixel_to_slot(tm, P, f, k);
        const uint32_t frame = frame_begin + (uint32_t)f;
        const float coff = 1.0f / ((float)(int)frame + 1.0f);
        const float Lambda = HERO_LAMBDA_MIN + HERO_LAMBDA_STEP * tm_rand(seed, (uint32_t)p, frame, TM_DIM_SPEC_LAMBDA);
        f4s spec; spec.v[0] = ps.fr[s]; spec.v[1] = ps.fg[s]; spec.v[2] = ps.fb[s]; spec.v[3] = fw[s];
        spec_add_splat(sp, spec, Lambda, coff, r, g, b);
    }
    px[0] = r; px[1] = g; px[2] = b;
}

// Per-batch device counters of a lane: one packed append counter per bounce (low word: paths that go on
// to bounce b+1, high word: shadow rays of bounce b), each in its own 128-byte line, then the sliced
// ray-fetch cursors of the max_depth + 1 traversal launches.
constexpr size_t LINE = 128;
static size_t lane_counter_bytes(int max_depth)
{ return LINE * (size_t)(max_depth + 1) + sizeof(int) * TR_FETCH_STRIDE * TR_FETCH_LINES * (size_t)(max_depth + 1); }

constexpr int PATH_WORDS = 2 * 15 + 4 + 12 + 3 + 2;  // two PathSoA + hit record + shadow ray + final radiance + the two fourth-wavelength words of PT_Spec
static size_t path_state_bytes(size_t S) { return sizeof(float) * PATH_WORDS * ((S + 3) & ~(size_t)3); }
static int ensure_paths(Lane &L, size_t S, int max_depth)
{
    if (S > L.path_capacity || !L.path_mem.p) {
        if (L.path_mem.ensure(path_state_bytes(S))) return TIRT_ERR_HIP;
        const size_t S_user = S;
        S = (S + 3) & ~(size_t)3;                     // array pitch: keeps every array (and the float4 hit records) 16-byte aligned
        float *w = L.path_mem.as<float>();
        PathState &p = L.ps;
        auto nxt = [&]() { float *r = w; w += S; return r; };
        for (int k = 0; k < 2; k++) {
            PathSoA &q = p.st[k];
            q.ox = nxt(); q.oy = nxt(); q.oz = nxt(); q.dx = nxt(); q.dy = nxt(); q.dz = nxt();
            q.tr = nxt(); q.tg = nxt(); q.tb = nxt(); q.rr = nxt(); q.rg = nxt(); q.rb = nxt();
            q.brdf_pdf = nxt(); q.flags = (uint32_t *)nxt(); q.slot = (int *)nxt();
        }
        p.hit = (float4 *)w; w += 4 * S;             // S is a multiple of 4 words from a 256-byte aligned base: 16-byte aligned
        p.sox = nxt(); p.soy = nxt(); p.soz = nxt(); p.sdx = nxt(); p.sdy = nxt(); p.sdz = nxt();
        p.scr = nxt(); p.scg = nxt(); p.scb = nxt(); p.sprim = (int *)nxt(); p.sdist = nxt(); p.sdst = (int *)nxt();
        p.fr = nxt(); p.fg = nxt(); p.fb = nxt();
        p.scw = nxt(); p.fw = nxt();
        L.path_capacity = S_user;
    }
    if (L.counters_mem.ensure(lane_counter_bytes(max_depth))) return TIRT_ERR_HIP;
    return 0;
}

int pt_render(tirt_ctx *c, uint32_t frame_begin, int frame_count, uint32_t seed, int max_depth, int stack_size, int flags, const SpecView *spec)
{
    TIRT_REQUIRE(c->built, "tirt_pt_rgb_render: LBVH not built");
    TIRT_REQUIRE(c->cam_set, "tirt_pt_rgb_render: camera not set");
    TIRT_REQUIRE(c->hdr.p && c->npix_local >= 0, "tirt_pt_rgb_render: film not created");
    TIRT_REQUIRE(frame_count >= 0 && max_depth >= 1 && max_depth <= 4096, "tirt_pt_rgb_render: bad frame_count/max_depth");
    if (frame_count == 0 || c->npix_local == 0) return TIRT_OK;
    if (ensure_counters(c)) return TIRT_ERR_HIP;
    if (ensure_shade_records(c)) return TIRT_ERR_HIP;          // on the main stream: the lanes wait for ev_main below
    const int P = (int)c->npix_local;
    // frames per batch: up to batch_paths pixel-samples in flight (the per-bounce launches of a
    // batch end in a latency-bound tail of a few long rays, so bigger batches amortise it)
    const size_t batch_paths = effective_batch_paths(c), merge_paths = effective_merge_paths(c);
    // batches of equal size, as many as come closest to the planned size (up to 1.5 x it: never a full batch and a sliver)
    int FB;
    { size_t nb = ((size_t)frame_count * P + batch_paths / 2) / batch_paths; if (nb < 1) nb = 1; FB = (int)((frame_count + nb - 1) / nb); }
    // A rank of a wide multi-GPU job renders its whole share as ONE batch, with nothing to overlap the per-bounce tails with:
    // as two half batches on two lanes it is 3.4 % faster at 8 ranks (measured on one GPU rendering rank 0's tiles; at 4 ranks
    // and below, and for a single GPU's stream of full batches, the halves lose 1-9 %: there 32 Mi-path batches win).
    // (Only for the first batch after a synchronisation: a rank with several batches in flight overlaps them anyway.)
    if (c->split_lone >= 2 && c->tile_count >= 6 && c->batches_since_sync == 0 && FB == frame_count && frame_count >= 2 && c->n_lanes >= 2 &&
        !c->time_kernels && (size_t)frame_count * P >= ((size_t)12 << 20))
        { int parts = c->split_lone < c->n_lanes ? c->split_lone : c->n_lanes; if (parts > frame_count) parts = frame_count; FB = (frame_count + parts - 1) / parts; }
    const SceneView sv = scene_view(c);
    const BvhView bv = bvh_view(c);
    TileMap tm = {c->tile_rank, c->tile_count, c->tile_size, c->H, c->tile_blocked, 0};
    DevCounters *ctr = c->dev_counters.as<DevCounters>();
    const int B = 256;
    // the pixels' candidate lists for the camera rays (tirt_pvb.hip), made on the main stream when the scene, the camera or the film changed since
    bool beams_ready = false;
    if (c->primary_beams && FB >= c->primary_beams_min_frames && !(flags & TIRT_TRAVERSE_EXHAUSTIVE)) {
        if (int rc = pvb_prepare(c)) return rc;
        beams_ready = c->pvb_valid;
    }
    // everything queued on the main stream (uploads, film clear, tone map) precedes the lanes' work
    TIRT_HIP(hipEventRecord(c->ev_main, c->stream));
    int n_lanes = c->time_kernels ? 1 : c->n_lanes;
    { const int need = plan_batches(c).lanes + 1; if (need < n_lanes) n_lanes = need; }      // a job of two big batches needs path state on two lanes (+ one for a call the hint did not cover), not on all

    // all lanes get their buffers up front (an allocation inside a later call would stall the pipeline)
    // sized for what deferred submission can merge later (merge_paths + one call), so that a bigger
    // merged batch does not re-allocate in the middle of a job; small films skip the head-room
    int spill_depth = 0;
    size_t cap = (size_t)FB * P;
    if (merge_paths > 0 && P >= 65536) {
        size_t want = batch_paths + batch_paths / 2;        // the largest batch a call can become (the same for every call: a lane never re-allocates mid-job)
        // "job_frames" hint (the example classes pass their sample count): a short job never merges more than itself
        if (c->job_frames > 0 && want > (size_t)c->job_frames * P) want = (size_t)c->job_frames * P;
        want = (want / P) * P;
        if (want > cap) cap = want;
    }
    for (int k = 0; k < n_lanes; k++) {
        // very large films (204 B per path and lane): use fewer lanes rather than run out of HBM
        if (k > 0 && cap > c->lanes[k].path_capacity) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < path_state_bytes(cap) + ((size_t)2 << 30)) { n_lanes = k; break; }
        }
        if (ensure_paths(c->lanes[k], cap, max_depth)) return TIRT_ERR_HIP;
        if (ensure_spill(c, c->lanes[k].spill, stack_size, spill_depth)) return TIRT_ERR_HIP;
    }

    for (int fb = 0; fb < frame_count; fb += FB) {
        Lane &L = c->lanes[n_lanes == 1 ? 0 : (c->lane_cursor++ % (unsigned)n_lanes)];
        c->batches_since_sync++;
        hipStream_t st = L.stream;
        const int F = (frame_count - fb < FB) ? frame_count - fb : FB;
        const int S = F * P;
        tm.F = (c->path_order_blocks && (P & 63) == 0) ? F : 0;
        const bool use_beams = beams_ready && F >= c->primary_beams_min_frames;
        const uint32_t f0 = frame_begin + (uint32_t)fb;
        char *cm = L.counters_mem.as<char>();
        auto append_ctr = [&](int b) { return (unsigned long long *)(cm + LINE * (size_t)b); };
        auto cnt_path = [&](int b) { return (const int *)append_ctr(b - 1); };           // live paths entering bounce b >= 1
        auto cnt_shadow = [&](int b) { return (const int *)append_ctr(b) + 1; };         // shadow rays made by bounce b
        auto fetch = [&](int launch) { return (int *)(cm + LINE * (size_t)(max_depth + 1)) + (size_t)launch * TR_FETCH_STRIDE * TR_FETCH_LINES; };

        TIRT_HIP(hipStreamWaitEvent(st, c->ev_main, 0));
        hipEvent_t r0, r1;
        TIRT_HIP(hipEventCreate(&r0)); TIRT_HIP(hipEventCreate(&r1));
        TIRT_HIP(hipEventRecord(r0, st));
        std::vector<std::pair<hipEvent_t, hipEvent_t>> evc, evs, evh;     // closest / shadow / shade timing pairs
        auto stamp = [&](std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, bool begin) {
            if (!c->time_kernels) return;
            if (begin) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); v.push_back({a, b}); (void)hipEventRecord(a, st); }
            else (void)hipEventRecord(v.back().second, st);
        };

        TIRT_HIP(hipMemsetAsync(L.counters_mem.p, 0, lane_counter_bytes(max_depth), st));
        hipLaunchKernelGGL(k_generate, dim3((S + B - 1) / B), dim3(B), 0, st, L.ps.st[0], c->cam, tm, P, S, f0, seed, ctr);
        // a batch that will run next to the previous one (still in flight on another lane) uses fewer persistent
        // blocks, so that both traversal kernels find LDS on the CUs; a batch submitted to an idle GPU takes them all
        bool busy = false;
        if (c->last_film) { busy = hipEventQuery(c->last_film) == hipErrorNotReady; (void)hipGetLastError(); }
        // (two big batches side by side -- what plan_batches makes of a hinted job -- do best with all persistent blocks each and
        // twice the shading blocks: +2 % for 2 x 128 / 64 / 32 Mi paths; four batches in flight, or two small ones, do not: -2 %)
        const BatchPlan plan = plan_batches(c);
        const bool two_big = plan.lanes <= 2 && plan.batch >= ((size_t)24 << 20) && !c->grid_user;
        const int grid_cap = (busy && n_lanes > 1 && !two_big) ? c->tr_grid : c->tr_grid_alone;
        int grid_full = (S + TR_BLOCK - 1) / TR_BLOCK; if (grid_full > grid_cap) grid_full = grid_cap;
        const int sh_cap = two_big ? 2 * c->sh_grid : c->sh_grid;
        int grid_shade = (S + SH_BLOCK - 1) / SH_BLOCK; if (grid_shade > sh_cap) grid_shade = sh_cap;
        v3 eye_v; eye_v.x = c->cam.eye[0]; eye_v.y = c->cam.eye[1]; eye_v.z = c->cam.eye[2];
        for (int b = 0; b < max_depth; b++) {
            const PathSoA &in = L.ps.st[b & 1], &out = L.ps.st[(b + 1) & 1];
            // closest hits of bounce b, together with the NEE shadow rays of bounce b-1 (they add into
            // `in`'s radiance, which nothing reads before shade(b)): one launch instead of two
            TraceArgs a = {};
            a.bvh = bv;
            a.ox = in.ox; a.oy = in.oy; a.oz = in.oz; a.dx = in.dx; a.dy = in.dy; a.dz = in.dz;
            if (b == 0) { a.ox = a.oy = a.oz = nullptr; for (int k = 0; k < 3; k++) a.eye[k] = c->cam.eye[k]; }   // camera rays share their origin
            a.count_ptr = (b == 0) ? nullptr : cnt_path(b); a.count_fixed = S;
            a.hit = L.ps.hit;
            a.spill = L.spill.as<int>(); a.spill_depth = spill_depth; a.ctr = ctr; a.per_ray_counts = nullptr;
            a.fetch = fetch(b);
            a.sox = L.ps.sox; a.soy = L.ps.soy; a.soz = L.ps.soz; a.sdx = L.ps.sdx; a.sdy = L.ps.sdy; a.sdz = L.ps.sdz;
            a.sprim = L.ps.sprim; a.sdst = L.ps.sdst; a.sdist = L.ps.sdist; a.scr = L.ps.scr; a.scg = L.ps.scg; a.scb = L.ps.scb;
            a.rr = in.rr; a.rg = in.rg; a.rb = in.rb; a.fr = L.ps.fr; a.fg = L.ps.fg; a.fb = L.ps.fb;
            if (spec) { a.scw = L.ps.scw; a.rw = (float *)in.flags; a.fw = L.ps.fw; }
            a.scount_ptr = (b == 0) ? nullptr : cnt_shadow(b - 1);
            fill_tunables(c, a);
            a.timeline = timeline_for(c, flags, grid_full);
            if (!(b == 0 && use_beams)) stamp(evc, true);
            if (b == 0 && use_beams) {
                // camera rays: each against the list of leaves its pixel's rays can hit first (tirt_pvb.hip); those that find nothing there go to k_trace.  Scratch the
                // batch does not touch before shade(0): the `out` arrays (slots, directions of the leftover rays), the shadow-ray arrays (their hit records)
                int *const fb_count = (int *)append_ctr(0) + 4;          // (zeroed with the batch's counters; words 0, 1 of the line: the appends of bounce 0)
                float4 *const fb_hit = (float4 *)L.ps.sox;               // sox, soy, soz, sdx: four consecutive arrays of the lane's pitch
                pvb_launch_cand(c, st, bv, in.dx, in.dy, in.dz, tm, P, S, L.ps.hit, fb_count, (int *)out.ox, out.dx, out.dy, out.dz, ctr);
                TraceArgs a2 = a;
                a2.dx = out.dx; a2.dy = out.dy; a2.dz = out.dz; a2.count_ptr = fb_count; a2.count_fixed = S; a2.hit = fb_hit; a2.no_ray_count = 1;
                stamp(evc, true);                                      // (the traversal timers and counters see k_trace's launch, not the list pass)
                if (int rc = launch_trace<KIND_CLOSEST>(c, st, a2, flags, grid_full)) return rc;
                stamp(evc, false);
                pvb_launch_scatter(st, fb_count, (const int *)out.ox, fb_hit, L.ps.hit);
            } else
            if (int rc = (b == 0) ? launch_trace<KIND_CLOSEST>(c, st, a, flags, grid_full) : launch_trace<KIND_MIXED>(c, st, a, flags, grid_full)) return rc;
            if (!(b == 0 && use_beams)) stamp(evc, false);
            c->launches_trace_closest++;

            stamp(evh, true);
            if (spec)
                hipLaunchKernelGGL(k_shade_spec, dim3(grid_shade), dim3(SH_BLOCK), 0, st, ShadeArgs{L.ps, in, out}, sv, *spec, tm, P, f0, seed, b,
                                   (b == max_depth - 1) ? 1 : 0, (b == 0) ? (const int *)nullptr : cnt_path(b), S,
                                   append_ctr(b), ctr, eye_v);
            else
            hipLaunchKernelGGL(k_shade, dim3(grid_shade), dim3(SH_BLOCK), 0, st, ShadeArgs{L.ps, in, out}, sv, tm, P, f0, seed, b,
                               (b == max_depth - 1) ? 1 : 0, (b == 0) ? (const int *)nullptr : cnt_path(b), S,
                               append_ctr(b), ctr, eye_v);
            stamp(evh, false);
            c->launches_shade++;

            if (b == max_depth - 1) {          // the last bounce's shadow rays have no closest-hit launch to ride on
                TraceArgs sa = {};
                sa.bvh = bv;
                sa.ox = L.ps.sox; sa.oy = L.ps.soy; sa.oz = L.ps.soz; sa.dx = L.ps.sdx; sa.dy = L.ps.sdy; sa.dz = L.ps.sdz;
                sa.count_ptr = cnt_shadow(b); sa.count_fixed = 0;
                sa.sprim = L.ps.sprim; sa.sdst = L.ps.sdst; sa.sdist = L.ps.sdist; sa.scr = L.ps.scr; sa.scg = L.ps.scg; sa.scb = L.ps.scb;
                sa.rr = out.rr; sa.rg = out.rg; sa.rb = out.rb; sa.fr = L.ps.fr; sa.fg = L.ps.fg; sa.fb = L.ps.fb;
                if (spec) { sa.scw = L.ps.scw; sa.rw = (float *)out.flags; sa.fw = L.ps.fw; }
                sa.spill = L.spill.as<int>(); sa.spill_depth = spill_depth; sa.ctr = ctr; sa.per_ray_counts = nullptr;
                sa.fetch = fetch(max_depth);
                fill_tunables(c, sa);
                stamp(evs, true);
                if (int rc = launch_trace<KIND_SHADOW_ACC>(c, st, sa, flags, grid_full)) return rc;
                stamp(evs, false);
                c->launches_trace_shadow++;
            }
        }
        // the running mean is order dependent: this batch's film update follows the previous batch's
        if (c->last_film) TIRT_HIP(hipStreamWaitEvent(st, c->last_film, 0));
        if (spec) hipLaunchKernelGGL(k_film_spec, dim3((P + B - 1) / B), dim3(B), 0, st, L.ps, L.ps.fw, *spec, tm, P, F, f0, seed, c->hdr.as<float>());
        else hipLaunchKernelGGL(k_film, dim3((P + B - 1) / B), dim3(B), 0, st, L.ps, tm, P, F, f0, c->hdr.as<float>());
        TIRT_HIP(hipEventRecord(L.film_done, st));
        L.film_recorded = true;
        c->last_film = L.film_done;
        if (use_beams) c->pvb_set[c->pvb_cur].busy = L.film_done;      // the last batch that reads this set of candidate lists (pvb_prepare)
        TIRT_HIP(hipEventRecord(r1, st));
        if (c->time_kernels) {
            TIRT_HIP(hipStreamSynchronize(st));
            auto drain = [&](std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, double &acc) {
                for (auto &pr : v) { float ms = 0; (void)hipEventElapsedTime(&ms, pr.first, pr.second); acc += ms; (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
                v.clear();
            };
            drain(evc, c->ms_trace_closest); drain(evs, c->ms_trace_shadow); drain(evh, c->ms_shade);
        }
        c->ev_pool.push_back({r0, r1});        // drained (and destroyed) by tirt_stats / tirt_stats_reset
        if (c->ev_pool.size() > 64) {          // long render loops that never ask for stats: retire finished pairs
            size_t keep = 0;
            for (auto &pr : c->ev_pool) {
                if (hipEventQuery(pr.second) == hipSuccess) {
                    float ms = 0.0f;
                    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->ms_render += ms;
                    (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
                } else c->ev_pool[keep++] = pr;
            }
            c->ev_pool.resize(keep);
            (void)hipGetLastError();           // hipEventQuery's hipErrorNotReady is not an error
        }
    }
    // main-stream consumers of the film (tone map, downloads, clear) wait for c->last_film themselves
    TIRT_HIP(hipGetLastError());
    return TIRT_OK;
}

}  // namespace tirt
