// Primary visibility through pixel beams (round 5): the camera rays of a batch are the same few thousand beams over and over.
//
// A wavefront batch holds F frames of the same P pixels: F camera rays per pixel that differ only in their sub-pixel jitter
// (Camera.py:135-137) and walk the same top of the tree to the same handful of leaves -- a third of all the rays of PT_RGB (the
// headline scene: 33.5 M of 97 M per step, 4.3 of 19 ms of traversal).  What the F rays of a pixel can possibly hit is a property of
// the PIXEL (camera, scene), not of the frame:
//
//   pvb_prepare   once per (scene build, camera, film / tiles):
//     k_pvb_probes    five rays per pixel -- its centre and its corners -- traced by k_trace like any other ray;
//     k_pvb_beam      bound(pixel) = the farthest of their hit distances x 1.5 (PVB_REACH); the pyramid of the pixel (its corner
//                     directions moved outward by a twentieth of a pixel) is walked down the quantised 4-wide tree and every LEAF
//                     whose box -- enlarged by k_trace's own margin -- it meets nearer than bound, and whose triangle does not lie
//                     wholly outside one of its faces, goes on the pixel's list, nearest first.  A list holds PVB_CMAX leaves: when
//                     one more turns up the farthest goes and the pixel's bound comes in to just below it (the list stays complete
//                     up to its bound);
//   per batch, instead of the bounce-0 launch of k_trace:
//     k_pvb_cand      one thread per camera ray: the primitive tests of its pixel's list -- k_trace's leaf step (trace_leaf_step, the same function):
//                     Moller-Trumbore / sphere, `0 < t < hit_t` with the equal-distance rule, the `slabs` verification of the
//                     leaf's exact box and, failing that, of its ancestors --, hit record written if the hit lies within bound;
//                     the other rays (no hit within bound: they slipped past what the probes saw) are appended to a list,
//     k_trace         over that list (directions compacted into the idle `out` arrays of the batch, hits into the idle shadow
//                     arrays), k_pvb_scatter puts its hit records where bounce 0 expects them.
//
// Why the hit records are the ones k_trace writes, bit for bit.  k_trace's answer for a ray is the minimum (distance, then larger
// leaf index) over the VERIFIED primitive hits of all leaves it reaches, and it reaches every leaf whose enlarged boxes (its own
// and its ancestors') the ray enters before `hit_t x 1.0001`.  A ray of the pixel lies inside the pyramid; a box it enters at
// distance t is met by the pyramid at a point whose projection on the pyramid's axis is <= t (Cauchy-Schwarz), so every leaf that
// could hold a verified hit at distance <= bound is on the list, with all the leaves in front of it.  If the best verified hit of
// the list lies within bound it is therefore k_trace's; if not, k_trace is asked.  A list made with bound = infinity (some probe
// left the scene) and not cut short is complete: a miss on it is a miss.  Rays that fail `slabs` on the root's exact box miss, as in
// k_trace (KIND_CLOSEST).  A camera further than TR_FAR_RHO extents from the scene (k_trace stops culling by distance there) gets no lists.
#include "tirt_internal.h"
#include <cstring>

namespace tirt {

constexpr int PVB_CMAX = 24;                        // leaves on a pixel's list (12: 3 % of the headline's pixels without a list, -1.3 %; 48: no gain -- profiles/r05an_*)
constexpr int PVB_STACK = 96;                       // node stack of a beam walk (4-wide tree: three entries per level at most)
constexpr float PVB_WIDEN = 0.55f;                  // the pyramid's corners in pixels from the centre (the jitter is [-0.5, 0.5))
constexpr float PVB_REACH = 1.5f;                   // a pixel's list reaches this far beyond the farthest probe hit (1.0001: 1.4 % of the camera rays find nothing on their list, 1.25: 0.10 %, 1.5: 0.08 %; profiles/r05an_*)
constexpr int PVB_BLOCK = 1024;                      // threads of a k_pvb_cand block

struct PvbView { const int *count; const int2 *cand; const float *bound; };     // [P], [PVB_CMAX][P] (leaf code, bits of the nearest distance any ray of the pixel can reach the leaf's box at), [P]; count < 0: no list

__global__ void k_pvb_probes(CameraView cam, TileMap tm, int P, float4 *rays)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    const int p = local_to_pixel(tm, k), i = p / tm.H, j = p - i * tm.H;
    for (int pr = 0; pr < 5; pr++) {
        const float jx = pr == 0 ? 0.0f : (((pr - 1) & 1) ? 0.5f : -0.5f), jy = pr == 0 ? 0.0f : (((pr - 1) & 2) ? 0.5f : -0.5f);
        const v3 d = camera_ray_direction(cam, i, j, jx, jy);
        const size_t r = (size_t)pr * P + k;
        rays[2 * r] = make_float4(cam.eye[0], cam.eye[1], cam.eye[2], d.x);
        rays[2 * r + 1] = make_float4(d.y, d.z, 0.0f, 0.0f);
    }
}

TD float h2f(unsigned h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }

// One WAVE walks the tree for its 64 neighbouring pixels (an 8 x 8 block of the film when the tiles are blocked): one node stack for the wave, a node's record loaded once,
// every lane tests the node's four boxes against ITS pixel's pyramid and a child is pushed if any lane's pyramid meets it (a lane whose pyramid missed the parent misses the
// children too -- the boxes nest --, and a leaf too many on a list would only cost its test).  The first version gave every pixel its own walk and its own stack in scratch
// memory: 5.2 ms for a 1024^2 film against this one's ~1 (`profiles/r05ar_*`).
__global__ __launch_bounds__(64) void k_pvb_beam(BvhView b, CameraView cam, TileMap tm, int P, const float4 *probe_hits, int *count, int2 *cand, float *bound_out,
                                                 unsigned long long *stat)
{
    __shared__ int s_stack[PVB_STACK];
    __shared__ int2 s_list[PVB_CMAX * 64];                 // [entry][lane]
    const int lane = threadIdx.x, k = blockIdx.x * 64 + lane;
    const bool live = k < P;
    const int p = local_to_pixel(tm, live ? k : 0), i = p / tm.H, j = p - i * tm.H;
    float bound = 0.0f;
    if (live) {
        for (int pr = 0; pr < 5; pr++) {
            const float4 h = probe_hits[(size_t)pr * P + k];
            bound = (__float_as_int(h.w) >= 0 && h.x < INF_VALUE) ? maxf(bound, h.x) : INF_VALUE;
            if (!(bound < INF_VALUE)) break;
        }
        if (bound < INF_VALUE) bound = bound * PVB_REACH;
    }
    const v3 eye = V(cam.eye[0], cam.eye[1], cam.eye[2]);
    const v3 cell = V(b.cell[0], b.cell[1], b.cell[2]), gmin = V(b.grid_min[0], b.grid_min[1], b.grid_min[2]);
    // k_trace's margin around a box, in cells (trace_margin_cells of the eye's distance from the grid: tirt_internal.h), and a little more
    const float rho = trace_origin_rho(b, eye.x, eye.y, eye.z);
    const float mc = trace_margin_cells(rho) + TR_MARGIN_BEAM_EXTRA;
    int n = 0;
    const bool near_enough = rho <= TR_FAR_RHO;            // (wave-uniform)
    bool open = live && near_enough;                        // this lane still collects leaves
    bool stack_over = false;
    if (near_enough) {
        v3 dg[4];
        const float W = PVB_WIDEN;
        const float cx[4] = {-W, W, W, -W}, cy[4] = {-W, -W, W, W};
        for (int q = 0; q < 4; q++) { const v3 d = camera_ray_direction(cam, i, j, cx[q], cy[q]); dg[q] = V(d.x / cell.x, d.y / cell.y, d.z / cell.z); }
        const v3 dc = camera_ray_direction(cam, i, j, 0.0f, 0.0f);
        const v3 dcg = V(dc.x / cell.x, dc.y / cell.y, dc.z / cell.z);
        const v3 og = V((eye.x - gmin.x) / cell.x, (eye.y - gmin.y) / cell.y, (eye.z - gmin.z) / cell.z);
        v3 nrm[4]; float eps[4];
        for (int q = 0; q < 4; q++) {
            v3 nn = cross(dg[q], dg[(q + 1) & 3]);
            if (dot(nn, dcg) < 0.0f) nn = -nn;                       // inward
            nrm[q] = nn;
            // rounding of the plane expression: 1e-5 of its terms' magnitude at the far end of the grid (the widening of the pyramid is 5e-5 of the distance)
            eps[q] = 1.0e-5f * (absf(nn.x) + absf(nn.y) + absf(nn.z)) * (absf(og.x) + absf(og.y) + absf(og.z) + 3.0f * (TR_GRID_HALF + 2.0f));
        }
        const v3 ac = V(dc.x * cell.x, dc.y * cell.y, dc.z * cell.z);           // projection on the axis, per cell
        const float abase = dot(gmin - eye, dc);
        int sp = 0;                                         // (wave-uniform: every push below is decided by a ballot)
        const int root = b.root_qcode;
        // a one-primitive scene: the root IS the leaf
        if (root < 0) { if (root != (int)0x80000000 && root != TR_EMPTY && open) { s_list[lane] = make_int2(root, 0); n = 1; } }
        else { s_stack[0] = root; sp = 1; }
        while (sp > 0) {
            if (__ballot(open) == 0ull) break;
            const int node = __builtin_amdgcn_readfirstlane(s_stack[--sp]);
            const uint4 *w = (const uint4 *)((const char *)b.cnode + ((size_t)(unsigned)node << 6));
            const uint4 q0 = w[0], q1 = w[1], q2 = w[2], q3 = w[3];
            const unsigned wx[4] = {q0.x, q0.w, q1.z, q2.y}, wy[4] = {q0.y, q1.x, q1.w, q2.z}, wz[4] = {q0.z, q1.y, q2.x, q2.w};
            const int cc[4] = {(int)q3.x, (int)q3.y, (int)q3.z, (int)q3.w};
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                const int code = cc[ch];
                if (code == TR_EMPTY) continue;
                const float lx = h2f(wx[ch] & 0xffffu) - mc, hx = h2f(wx[ch] >> 16) + mc;
                const float ly = h2f(wy[ch] & 0xffffu) - mc, hy = h2f(wy[ch] >> 16) + mc;
                const float lz = h2f(wz[ch] & 0xffffu) - mc, hz = h2f(wz[ch] >> 16) + mc;
                if (!(lx <= hx && ly <= hy && lz <= hz)) continue;
                bool out = false;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const v3 nn = nrm[q];
                    const float sd = nn.x * ((nn.x >= 0.0f ? hx : lx) - og.x) + nn.y * ((nn.y >= 0.0f ? hy : ly) - og.y) + nn.z * ((nn.z >= 0.0f ? hz : lz) - og.z);
                    out = out || (sd < -eps[q]);
                }
                // nearest any ray of the pyramid can reach the box: the least projection of the box on the axis (<= the distance along any unit direction)
                const float nearp = abase + ac.x * (ac.x >= 0.0f ? lx : hx) + ac.y * (ac.y >= 0.0f ? ly : hy) + ac.z * (ac.z >= 0.0f ? lz : hz);
                bool pass = open && !out && !(nearp > bound);
                if (__ballot(pass) == 0ull) continue;
                if (code >= 0) {
                    if (sp < PVB_STACK) { s_stack[sp] = code; sp++; } else stack_over = true;
                    continue;
                }
                // a triangle leaf: the triangle itself against the pyramid (its box met it; all three corners outside one face: no ray of the pixel
                // comes nearer to it than the widening of the pyramid, a twentieth of a pixel -- the primitive test's own slack is 1e-6 of the
                // distance), and the least projection of its corners on the axis instead of its box's (no point of it is nearer)
                float nr = nearp;
                const int lc = ~code;
                if (((lc >> 30) & 1) == 0) {
                    const float4 *tp = b.tri + (size_t)(lc & 0x3fffffff) * TRI_STRIDE;
                    const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                    const v3 w0 = V(ta.x, ta.y, ta.z), w1 = V(tb.x, tb.y, tb.z), w2 = V(tc.x, tc.y, tc.z);
                    const v3 g0 = V((w0.x - eye.x) / cell.x, (w0.y - eye.y) / cell.y, (w0.z - eye.z) / cell.z);
                    const v3 g1 = V((w1.x - eye.x) / cell.x, (w1.y - eye.y) / cell.y, (w1.z - eye.z) / cell.z);
                    const v3 g2 = V((w2.x - eye.x) / cell.x, (w2.y - eye.y) / cell.y, (w2.z - eye.z) / cell.z);
                    bool off = false;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float e = eps[q] + 0.05f * (absf(nrm[q].x) + absf(nrm[q].y) + absf(nrm[q].z));      // (+ a twentieth of a cell)
                        off = off || (dot(nrm[q], g0) < -e && dot(nrm[q], g1) < -e && dot(nrm[q], g2) < -e);
                    }
                    const float pr = __builtin_fminf(__builtin_fminf(dot(w0 - eye, dc), dot(w1 - eye, dc)), dot(w2 - eye, dc));
                    nr = maxf(nr, pr - 1.0e-4f * absf(pr) - 1.0e-6f);
                    pass = pass && !off && !(nr > bound);
                }
                if (pass) {
                    const float nr0 = maxf(nr, 0.0f);
                    if (n < PVB_CMAX) { s_list[n * 64 + lane] = make_int2(code, __float_as_int(nr0)); n++; }
                    else {
                        // more leaves than a list holds: the farthest one goes (this leaf or one on the list) and the pixel's bound comes in to just below where it
                        // started -- the list stays complete up to its bound, the rays that find nothing that near are k_trace's
                        int far = -1; float fn = nr0;
                        for (int a = 0; a < PVB_CMAX; a++) { const float x = __int_as_float(s_list[a * 64 + lane].y); if (x > fn) { fn = x; far = a; } }
                        if (far >= 0) s_list[far * 64 + lane] = make_int2(code, __float_as_int(nr0));
                        bound = __builtin_fminf(bound, fn * 0.999999f);
                    }
                }
            }
            if (stack_over) break;
        }
    }
    const bool whole = live && near_enough && !stack_over;
    if (whole) {
        // (a bound that came in while the list was made: what lies beyond it is of no use)
        int m = 0;
        for (int a = 0; a < n; a++) { const int2 e = s_list[a * 64 + lane]; if (__int_as_float(e.y) <= bound) { s_list[m * 64 + lane] = e; m++; } }
        n = m;
        // nearest first: a ray that has a hit stops at the first leaf that lies beyond it (k_pvb_cand)
        for (int a = 1; a < n; a++) {
            const int2 e = s_list[a * 64 + lane];
            const float nr = __int_as_float(e.y);
            int z = a - 1;
            while (z >= 0 && __int_as_float(s_list[z * 64 + lane].y) > nr) { s_list[(z + 1) * 64 + lane] = s_list[z * 64 + lane]; z--; }
            s_list[(z + 1) * 64 + lane] = e;
        }
        for (int a = 0; a < n; a++) cand[(size_t)a * P + k] = s_list[a * 64 + lane];
    }
    if (live) { count[k] = whole ? n : -1; bound_out[k] = bound; }
    if (stat) {
        // (diagnostics: pixels with a list, leaves on the lists, pixels whose probes all hit)
        const unsigned long long m = __ballot(whole);
        unsigned long long sm = whole ? (unsigned long long)n : 0ull;
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_down(sm, o, 64);
        const unsigned long long mb = __ballot(whole && bound < INF_VALUE);
        if (lane == 0) { atomicAdd(&stat[0], (unsigned long long)__popcll(m)); atomicAdd(&stat[1], sm); atomicAdd(&stat[2], (unsigned long long)__popcll(mb)); }
    }
}

// One camera ray against its pixel's list (see the header); the unresolved ones are appended to (fb_slot, fb_d*).
__global__ __launch_bounds__(PVB_BLOCK) void k_pvb_cand(BvhView b, PvbView pv, const float *dx, const float *dy, const float *dz, v3 eye, TileMap tm, int P, int S,
                                                  float4 *hit, int *fb_count, int *fb_slot, float *fb_dx, float *fb_dy, float *fb_dz, DevCounters *ctr,
                                                  unsigned long long *stat, unsigned long long *diag)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const bool live = s < S;
    // the ray's local pixel (the frame is not needed).  Pixel-block-major numbering (TileMap::F): the 64-path chunk is the wave's, its division by F scalar
    int k = 0;
    if (tm.F > 0) { const int ch = __builtin_amdgcn_readfirstlane(s >> 6); k = ((ch / tm.F) << 6) | (s & 63); }
    else if (live) { int f; slot_to_frame_pixel(tm, P, s, f, k); }
    const v3 d = live ? V(__builtin_nontemporal_load(&dx[s]), __builtin_nontemporal_load(&dy[s]), __builtin_nontemporal_load(&dz[s])) : V(0.0f, 0.0f, 1.0f);
    const RayCtx r = make_ray(eye, d);
    const bool par = ray_has_parallel_axis(r);
    int n = live ? pv.count[k] : 0;
    const float bound = live ? pv.bound[k] : 0.0f;
    bool resolved = live && n >= 0;
    // k_trace's first step for a camera ray: the root's exact box -- when the root is an inner node: in a one-primitive scene the root IS the leaf, k_trace
    // goes straight to the primitive test and a hit that fails the leaf's box has no ancestor to be refused by -- and a NaN ray misses
    bool dead = false;
    if (live) {
        float tn;
        if (b.root_qcode >= 0 && !slabs(r, b.root_min[0], b.root_min[1], b.root_min[2], b.root_max[0], b.root_max[1], b.root_max[2], tn)) dead = true;
        if (!((d.x == d.x) & (d.y == d.y) & (d.z == d.z))) dead = true;
    }
    if (dead) { n = 0; resolved = true; }
    if (!resolved) n = 0;
    float hit_t = INF_VALUE, hit_u = 0.0f, hit_v = 0.0f; int hit_prim = -1, hit_leaf = -1;
    int steps = 0, trips = 0;          // (diagnostics, option "primary_beams_diag")
    for (int c = 0; c < PVB_CMAX; c++) {
        // (the lists are sorted by the distance at which a ray of the pixel can reach the leaf's box at the earliest: beyond the hit so far, nothing on the rest of the list can win or tie)
        int2 en = make_int2(0, 0x7f800000);
        if (c < n) en = pv.cand[(size_t)c * P + k];
        const bool go = c < n && __int_as_float(en.y) <= hit_t;
        if (!go) n = 0;
        if (__ballot(go) == 0ull) break;
        trips++; if (go) steps++;
        if (go) (void)trace_leaf_step<true>(b, r, par, ~en.x, hit_t, hit_u, hit_v, hit_prim, hit_leaf);      // k_trace's leaf step itself (tirt_internal.h)
    }
    // within bound: k_trace's answer (a list made without a bound is complete: whatever it says holds)
    if (resolved && !dead && !(hit_t <= bound) && bound < INF_VALUE) resolved = false;
    if (resolved) {
        typedef float f4n __attribute__((ext_vector_type(4)));
        f4n hv = {hit_t, hit_u, hit_v, __int_as_float(hit_prim)};
        __builtin_nontemporal_store(hv, (f4n *)&hit[s]);
    }
    // the leftover rays of the BLOCK take their places in the list with one atomic (same-address atomics retire at ~11 ns each: one per wave -- four fifths
    // of the waves have a leftover ray -- made this kernel 2.9 ms long)
    const bool fb = live && !resolved;
    const unsigned long long fm = __ballot(fb);
    __shared__ int s_wn[PVB_BLOCK / 64], s_base;
    const int wid = threadIdx.x >> 6;
    if (lane == 0) s_wn[wid] = __popcll(fm);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < PVB_BLOCK / 64; w++) tot += s_wn[w];
        s_base = tot ? atomicAdd(fb_count, tot) : 0;
        if (stat && tot) atomicAdd(&stat[3], (unsigned long long)tot);
    }
    __syncthreads();
    if (fb) {
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        int at = s_base + __popcll(fm & lt);
        for (int w = 0; w < wid; w++) at += s_wn[w];
        fb_slot[at] = s; fb_dx[at] = d.x; fb_dy[at] = d.y; fb_dz[at] = d.z;
    }
    if (s == 0 && ctr) atomicAdd(&ctr->rays_closest, (unsigned long long)S);
    if (stat && s == 0) atomicAdd(&stat[4], (unsigned long long)S);
    if (diag) {
        // leaf steps of all rays, rays with more than one, wave trips x 64 (lane slots), rays with more than two
        unsigned long long a = (unsigned long long)steps, m1 = steps > 1 ? 1ull : 0ull, m2 = steps > 2 ? 1ull : 0ull;
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); m1 += __shfl_down(m1, o, 64); m2 += __shfl_down(m2, o, 64); }
        if (lane == 0) { atomicAdd(&diag[0], a); atomicAdd(&diag[1], m1); atomicAdd(&diag[2], (unsigned long long)trips * 64ull); atomicAdd(&diag[3], m2); }
    }
}

__global__ void k_pvb_scatter(const int *fb_count, const int *fb_slot, const float4 *fb_hit, float4 *hit)
{
    const int n = *fb_count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) hit[fb_slot[i]] = fb_hit[i];
}

// (re)builds the pixels' lists when the scene, the camera or the film changed since they were made; on the main stream.  The lists are an optimisation: when their
// memory is not to be had (430 B per pixel while they are made, 216 B afterwards) the render goes on without them -- pvb_valid stays false, the camera rays take the
// ordinary bounce-0 launch (a failed allocation here used to fail a render that k_trace alone could do).
int pvb_prepare(tirt_ctx *c)
{
    const int P = (int)c->npix_local;
    tirt_ctx::PvbKey key;
    memset(&key, 0, sizeof(key));
    key.cam = c->cam; key.build = c->build_serial; key.W = c->W; key.H = c->H; key.tile_rank = c->tile_rank; key.tile_count = c->tile_count;
    key.tile_size = c->tile_size; key.tile_blocked = c->tile_blocked; key.P = P;
    if (c->pvb_valid && !memcmp(&key, &c->pvb_key, sizeof(key))) return TIRT_OK;
    c->pvb_valid = false;
    if (P <= 0) return TIRT_OK;
    tirt_ctx::PvbSet &ps = c->pvb_set[c->pvb_cur ^ 1];          // the set no batch submitted since the last rebuild reads
    hipStream_t st = c->stream;
    // the probe rays, their hit records: scratch of this call only (stream-ordered allocation: no synchronisation, gone when the walk is over)
    void *tmp = nullptr; bool tmp_async = true;
    const size_t tmp_bytes = (sizeof(float4) * 2 + sizeof(float4)) * 5 * (size_t)P;
    auto without = [&](const char *what) {
        (void)hipGetLastError();                               // the failure is handled: leave no sticky error behind
        c->pvb_skipped++;
        set_error(std::string("primary_beams: no candidate lists for this camera (") + what + "); the camera rays take the ordinary launch");
        if (tmp && tmp_async) (void)hipFreeAsync(tmp, st);
        return TIRT_OK;
    };
    if (ps.count.ensure(sizeof(int) * (size_t)P) || ps.bound.ensure(sizeof(float) * (size_t)P) || ps.cand.ensure(sizeof(int2) * (size_t)PVB_CMAX * P) || c->pvb_stat.ensure(128))
        return without("list memory");
    if (hipMallocAsync(&tmp, tmp_bytes, st) != hipSuccess) {
        // (a runtime or device without stream-ordered allocation: an ordinary buffer, kept for the next build)
        (void)hipGetLastError(); tmp = nullptr; tmp_async = false;
        if (c->pvb_tmp.ensure(tmp_bytes)) return without("probe scratch");
        tmp = c->pvb_tmp.p;
    }
    if (trace_arrays_prepare(c, -1)) return without("traversal buffers");
    // batches still in flight read the OTHER set (a camera move submits what is pending and does not wait); this one was last read by the batches of the camera
    // before last: the film updates are chained in submission order, so the event of the last of them covers them all
    if (ps.busy && hipStreamWaitEvent(st, ps.busy, 0) != hipSuccess) return without("event");
    TileMap tm = {c->tile_rank, c->tile_count, c->tile_size, c->H, c->tile_blocked, 0};
    float4 *rays = (float4 *)tmp, *hits = rays + 10 * (size_t)P;
    const int B = 256;
    if (hipMemsetAsync(c->pvb_stat.p, 0, 128, st) != hipSuccess) return without("memset");
    // what a list build costs on the device (tirt_primary_beam_stats: builds and their time since the last tirt_stats_reset -- bench.py puts a build inside its clock)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, st);
    else { if (e0) (void)hipEventDestroy(e0); e0 = e1 = nullptr; (void)hipGetLastError(); }
    hipLaunchKernelGGL(k_pvb_probes, dim3((P + B - 1) / B), dim3(B), 0, st, c->cam, tm, P, rays);
    if (int rc = trace_arrays(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 5 * P, nullptr, hits, nullptr, nullptr, false, -1, rays)) { if (tmp_async) (void)hipFreeAsync(tmp, st); if (e0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); } return rc; }
    hipLaunchKernelGGL(k_pvb_beam, dim3((P + 63) / 64), dim3(64), 0, st, bvh_view(c), c->cam, tm, P, hits, ps.count.as<int>(), ps.cand.as<int2>(),
                       ps.bound.as<float>(), c->pvb_stat.as<unsigned long long>());
    if (e0) { (void)hipEventRecord(e1, st); c->pvb_ev.push_back({e0, e1}); }
    if (c->pvb_ev.size() > 64) {           // a caller that moves the camera for hours and never asks for statistics: the oldest pair is long finished
        auto pr = c->pvb_ev.front(); c->pvb_ev.erase(c->pvb_ev.begin());
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->pvb_build_ns += (unsigned long long)((double)ms * 1.0e6);
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); (void)hipGetLastError();
    }
    c->pvb_builds++;
    if (tmp_async) TIRT_HIP(hipFreeAsync(tmp, st));
    TIRT_HIP(hipGetLastError());
    c->pvb_cur ^= 1; ps.busy = nullptr;
    memcpy(&c->pvb_key, &key, sizeof(key)); c->pvb_valid = true;
    return TIRT_OK;
}

// bounce 0 of a batch of S camera rays (directions in `dx ..`, hit records to `hit`) against the pixels' lists; the rays left over are appended to (fb_slot, fb_d*):
// pt_render traces them with k_trace and calls pvb_launch_scatter
void pvb_launch_cand(tirt_ctx *c, hipStream_t st, const BvhView &bv, const float *dx, const float *dy, const float *dz, const TileMap &tm, int P, int S,
                     float4 *hit, int *fb_count, int *fb_slot, float *fb_dx, float *fb_dy, float *fb_dz, DevCounters *ctr)
{
    const tirt_ctx::PvbSet &ps = c->pvb_set[c->pvb_cur];
    PvbView pv = {ps.count.as<int>(), ps.cand.as<int2>(), ps.bound.as<float>()};
    v3 eye; eye.x = c->cam.eye[0]; eye.y = c->cam.eye[1]; eye.z = c->cam.eye[2];
    const int B = PVB_BLOCK;
    hipLaunchKernelGGL(k_pvb_cand, dim3((S + B - 1) / B), dim3(B), 0, st, bv, pv, dx, dy, dz, eye, tm, P, S, hit, fb_count, fb_slot, fb_dx, fb_dy, fb_dz,
                       ctr, c->pvb_stat.as<unsigned long long>(), c->pvb_diag ? c->pvb_stat.as<unsigned long long>() + 8 : nullptr);
}
void pvb_launch_scatter(hipStream_t st, const int *fb_count, const int *fb_slot, const float4 *fb_hit, float4 *hit)
{
    hipLaunchKernelGGL(k_pvb_scatter, dim3(2048), dim3(256), 0, st, fb_count, fb_slot, fb_hit, hit);
}

}  // namespace tirt
