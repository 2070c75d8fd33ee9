// libtirt -- the TRAVERSAL tree: a binned surface-area-heuristic binary tree over the scene's primitives, built on the device,
// which build_wide (tirt_lbvh.hip) collapses into the quantised 4-wide nodes k_trace walks.
//
// Why a second tree.  The reference's structure is its LBVH (accel/LBvh.py:229-467), and its RESULTS are defined by that
// tree: a leaf is visited iff every proper ancestor's box passes `slabs`, ties go to the visiting order.  k_trace
// re-establishes exactly that condition for every candidate hit (leaf box / `cparent` chain of the reference tree, tie rule
// on the reference's leaf index), so the structure that FINDS the candidates may be any hierarchy whose boxes contain their
// leaves' boxes.  A Morton-split tree is a poor one to walk: measured on the same kernel, the SAH tree costs 26.8 node
// visits per ray of the headline scene instead of 32.1 (+10 % rays/s), 9.6 instead of 14.5 on the Teapot (+13 %), 11.0
// instead of 14.5 on the Veach scene (+15 %) -- with bit-identical films (tools/exp/sah_tree.py; PLOC and a two-level
// build over Morton clusters, also tried there, are worse than the plain top-down binned SAH on all three).
//
// Algorithm: top-down, level-synchronous, one primitive per leaf (as the reference).  A node is a contiguous range of an
// index array over the Morton-sorted primitives; its pre-order index is known when it is created (left = self + 1, right =
// self + 2 * n_left: a subtree of k leaves has 2k - 1 nodes), so nodes are written straight into the `compact` layout
// k_wide_level reads, with no numbering pass and no atomics for node ids -- the tree is deterministic.  Per node: (A) box
// and centroid bounds, (B) 32 bins per axis by centroid (integer-keyed min / max / count atomics in LDS), (C) the cheapest
// of the 3 x 31 candidate planes by area(L) * n_L + area(R) * n_R, (D) a stable partition of the range into the other
// index buffer.  All centroids in one bin on every axis (duplicates), or a tree deeper than 64 levels: the range is halved.
// Sizes of node, so that the top of the tree is as parallel as its bottom:
//   subtree (<= SAH_SUB primitives)   the WHOLE subtree is finished by one wave in one launch at the end (k_sah_subtree): its nodes of more
//                                     than SAH_MINI primitives with the binned sweep below, smaller ones 16 lanes per node, everything in
//                                     registers, with an EXACT sweep over the sorted centroids of each axis instead of bins (every lane
//                                     prices the split behind its own primitive)
//   small  (<= SAH_LARGE)             one wave per node, 16 nodes per block                     k_sah_level<1>
//   large  (<= SAH_HUGE)              one 1024-thread block per node                            k_sah_level<16>
//   huge                              a block per SAH_CHUNK primitives, three launches per level: bins of the chunks merged
//                                     into the node's with global atomics (k_sah_huge_bin), plane + per-chunk output offsets
//                                     from the chunks' bin counts (k_sah_huge_eval), partition + the bounds of huge children
//                                     (k_sah_huge_part)
// Where the time goes at 100 001 primitives (rocprofv3 timeline, profiles/r03_build_timeline.txt): six levels with huge nodes 0.47 ms
// (k_sah_huge_eval alone 24 us per level: one wave per node), k_sah_level<16> 0.27 ms, k_sah_level<1> 0.31 ms over six levels (LDS-atomic
// binning of up to 512 primitives by one wave), k_sah_subtree 0.20 ms (a wave's serial chain through six levels), two counter reads 0.19 ms.
// The next level's task lists are appended with one atomic per block and list (same-address atomics: ~11 ns each).
#include "tirt_internal.h"
#include "tirt_device.h"

namespace tirt {

constexpr int SAH_BINS = 32;
constexpr int SAH_BIN_WORDS = 3 * SAH_BINS * 7;      // per node: [axis][bin][min x y z, max x y z, count]
constexpr int SAH_MINI = 16;              // at most this many primitives: a quarter of a wave works on the node, in registers, with an exact sweep
constexpr int SAH_SUB = 64;               // at most this many primitives: the whole subtree is finished by ONE wave in one launch (k_sah_subtree)
constexpr int SAH_LARGE = 512;            // more primitives than this: a whole block works on the node
constexpr int SAH_HUGE = 8192;            // more than this: a block per chunk
constexpr int SAH_CHUNK = 1024;
constexpr int SAH_BLOCK = 1024;
constexpr int SAH_FORCE_HALVING_AFTER = 64;
constexpr int SAH_CNT = 8;                // counters per level: small, large, huge, chunks, mini
constexpr int SAH_MAX_LEVELS = 160;       // halving from level 64 on ends every range within 64 + log2(n) levels

struct SahTask { int start, count, pre, pad; };
struct SahHuge {
    int start, count, pre, first_chunk;
    unsigned bounds[12];                  // keyed: box min[3], max[3], centroid min[3], max[3]
    int axis, plane, nl, pad;
    int child[2], pad2[2];                // slots of huge children in the next level's list, or -1
};

TD unsigned sah_key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }      // unsigned order = float order
TD float sah_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
TD bool sah_is_min(int k) { return (k < 3) || (k >= 6 && k < 9); }
TD float sah_half_area(float dx, float dy, float dz) { return dx * dy + dy * dz + dz * dx; }
TD int sah_bin(float c, float cmin, float scale) { const int b = (int)((c - cmin) * scale); return b < 0 ? 0 : (b >= SAH_BINS ? SAH_BINS - 1 : b); }

// 12 running bounds of one lane -> of the wave (all lanes get the result)
TD void sah_wave_bounds(float r[12])
{
#pragma unroll
    for (int k = 0; k < 12; k++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const float v = __shfl_xor(r[k], o, 64); r[k] = sah_is_min(k) ? fminf(r[k], v) : fmaxf(r[k], v); }
    }
}
TD void sah_bounds_init(float r[12])
{
#pragma unroll
    for (int k = 0; k < 12; k++) r[k] = sah_is_min(k) ? 3.0e38f : -3.0e38f;
}
TD void sah_bounds_add(float r[12], float4 a, float4 b)
{
    r[0] = fminf(r[0], a.x); r[1] = fminf(r[1], a.y); r[2] = fminf(r[2], a.z);
    r[3] = fmaxf(r[3], b.x); r[4] = fmaxf(r[4], b.y); r[5] = fmaxf(r[5], b.z);
    const float cx = 0.5f * (a.x + b.x), cy = 0.5f * (a.y + b.y), cz = 0.5f * (a.z + b.z);
    r[6] = fminf(r[6], cx); r[7] = fminf(r[7], cy); r[8] = fminf(r[8], cz);
    r[9] = fmaxf(r[9], cx); r[10] = fmaxf(r[10], cy); r[11] = fmaxf(r[11], cz);
}

// One primitive into the three axes' bins (LDS).
TD void sah_bin_add(unsigned (*bin)[SAH_BINS][7], float4 a, float4 b, const float cmin[3], const float scale[3])
{
    const unsigned k0 = sah_key(a.x), k1 = sah_key(a.y), k2 = sah_key(a.z), k3 = sah_key(b.x), k4 = sah_key(b.y), k5 = sah_key(b.z);
    const float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
        unsigned *q = bin[ax][sah_bin(c[ax], cmin[ax], scale[ax])];
        atomicMin(&q[0], k0); atomicMin(&q[1], k1); atomicMin(&q[2], k2);
        atomicMax(&q[3], k3); atomicMax(&q[4], k4); atomicMax(&q[5], k5);
        atomicAdd(&q[6], 1u);
    }
}

// One wave: the cheapest of the 3 x 31 planes over the bins in LDS; axis = -1 (halve the range) when no plane separates
// anything.  All lanes return the same values.  Ties: the lowest (axis, plane).
TD void sah_pick(const unsigned (*bin)[SAH_BINS][7], int count, bool halve, int lane, int &axis, int &plane, int &n_left)
{
    float best = 3.0e38f; int best_id = 0x7fffffff, best_nl = 0;
    if (!halve && count >= 2) {
        for (int cand = lane; cand < 3 * SAH_BINS; cand += 64) {
            const int ax = cand / SAH_BINS, sp = cand - ax * SAH_BINS;
            if (sp == SAH_BINS - 1) continue;
            float lmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, lmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, rmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, rmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            unsigned nl = 0, nr = 0;
            for (int b = 0; b < SAH_BINS; b++) {
                const unsigned *q = bin[ax][b];
                const unsigned cnt = q[6];
                if (cnt == 0) continue;
                const float m0 = sah_unkey(q[0]), m1 = sah_unkey(q[1]), m2 = sah_unkey(q[2]), x0 = sah_unkey(q[3]), x1 = sah_unkey(q[4]), x2 = sah_unkey(q[5]);
                if (b <= sp) {
                    lmn[0] = fminf(lmn[0], m0); lmn[1] = fminf(lmn[1], m1); lmn[2] = fminf(lmn[2], m2);
                    lmx[0] = fmaxf(lmx[0], x0); lmx[1] = fmaxf(lmx[1], x1); lmx[2] = fmaxf(lmx[2], x2); nl += cnt;
                } else {
                    rmn[0] = fminf(rmn[0], m0); rmn[1] = fminf(rmn[1], m1); rmn[2] = fminf(rmn[2], m2);
                    rmx[0] = fmaxf(rmx[0], x0); rmx[1] = fmaxf(rmx[1], x1); rmx[2] = fmaxf(rmx[2], x2); nr += cnt;
                }
            }
            if (nl == 0 || nr == 0) continue;
            const float cost = sah_half_area(lmx[0] - lmn[0], lmx[1] - lmn[1], lmx[2] - lmn[2]) * (float)nl +
                               sah_half_area(rmx[0] - rmn[0], rmx[1] - rmn[1], rmx[2] - rmn[2]) * (float)nr;
            if (cost < best || (cost == best && cand < best_id)) { best = cost; best_id = cand; best_nl = (int)nl; }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(best_id, o, 64), on = __shfl_xor(best_nl, o, 64);
        if (ob < best || (ob == best && oi < best_id)) { best = ob; best_id = oi; best_nl = on; }
    }
    if (best_id != 0x7fffffff) { axis = best_id / SAH_BINS; plane = best_id % SAH_BINS; n_left = best_nl; }
    else { axis = -1; plane = 0; n_left = count / 2; }
}

// (slot = the leaf's position in the index array = its rank among the leaves in depth-first order: where its primitive record goes)
TD void sah_write_leaf(float *compact, int *csize, int *prim_slot, int row_index, int slot, int prim, float4 a, float4 b)
{
    prim_slot[prim] = slot;
    float *row = compact + (size_t)row_index * CPN_VEC;
    row[0] = 1.0f; row[1] = (float)prim; row[2] = a.x; row[3] = a.y; row[4] = a.z; row[5] = b.x; row[6] = b.y; row[7] = b.z; row[8] = 0.0f;
    csize[row_index] = 1;
}
TD void sah_write_inner(float *compact, int *csize, int *parent, int row_index, int right, int count, const float box[6])
{
    parent[row_index + 1] = row_index; parent[right] = row_index;
    if (row_index == 0) parent[0] = -1;
    float *row = compact + (size_t)row_index * CPN_VEC;
    row[0] = 0.0f; row[1] = (float)right;
    row[2] = box[0]; row[3] = box[1]; row[4] = box[2]; row[5] = box[3]; row[6] = box[4]; row[7] = box[5]; row[8] = 0.0f;
    csize[row_index] = 2 * count - 1;
}

// Boxes in Morton order (accel/LBvh.py:397-426: triangle min / max, sphere centre -+ r), the identity index array, the
// bounds of everything (root of a huge scene) and the root's empty bins.
__global__ __launch_bounds__(256) void k_sah_prim_boxes(SceneView s, const int *sorted_prims, float4 *sbox, int *idx0, SahHuge *root, unsigned *root_bins, float sph_pad_abs)
{
    __shared__ unsigned s_b[12];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < 12) s_b[threadIdx.x] = sah_is_min(threadIdx.x) ? 0xffffffffu : 0u;
    if (blockIdx.x == 0) for (int k = threadIdx.x; k < SAH_BIN_WORDS; k += blockDim.x) root_bins[k] = (k % 7 < 3) ? 0xffffffffu : 0u;
    __syncthreads();
    float r[12]; sah_bounds_init(r);
    if (i < s.n) {
        const int p = sorted_prims[i];
        const int *pr = s.primitive + (size_t)p * PRI_VEC;
        v3 mn, mx;
        if (pr[0] == PRIMITIVE_TRI) {
            const v3 a = vtx_pos(s, pr[1]), b = vtx_pos(s, pr[1] + 1), c = vtx_pos(s, pr[1] + 2);
            mn = V(fminf(fminf(a.x, b.x), c.x), fminf(fminf(a.y, b.y), c.y), fminf(fminf(a.z, b.z), c.z));
            mx = V(fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z));
        } else {
            const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;
            // analytic sphere: its box in THIS tree is padded by what the reference's sphere test can be off by (sphere_pad, tirt_internal.h;
            // sph_pad_abs < 0: no padding -- the 4-wide slots of shapes then span the whole grid, as in round 2).  Spot / laser shapes are never hit.
            const float rr = ((int)sh[0] == SHAPE_SPHERE && sph_pad_abs >= 0.0f) ? sh[4] + sphere_pad(sh[4], sph_pad_abs) : sh[4];
            mn = V(sh[1] - rr, sh[2] - rr, sh[3] - rr); mx = V(sh[1] + rr, sh[2] + rr, sh[3] + rr);
        }
        const float4 a = make_float4(mn.x, mn.y, mn.z, 0.0f), b = make_float4(mx.x, mx.y, mx.z, 0.0f);
        sbox[2 * (size_t)i] = a; sbox[2 * (size_t)i + 1] = b;
        idx0[i] = i;
        sah_bounds_add(r, a, b);
    }
    sah_wave_bounds(r);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) { if (sah_is_min(k)) atomicMin(&s_b[k], sah_key(r[k])); else atomicMax(&s_b[k], sah_key(r[k])); }
    }
    __syncthreads();
    if (threadIdx.x < 12) { if (sah_is_min(threadIdx.x)) atomicMin(&root->bounds[threadIdx.x], s_b[threadIdx.x]); else atomicMax(&root->bounds[threadIdx.x], s_b[threadIdx.x]); }
}

// Small and large nodes.  WPT waves per task: 1 (16 tasks per block) or 16 (one task per block).  Every thread of a block
// with work reaches every barrier; a task slot without a task has count == 0.
template <int WPT>
__global__ __launch_bounds__(SAH_BLOCK) void k_sah_level(const float4 *__restrict__ sbox, const int *__restrict__ sorted_prims,
                                                         const int *__restrict__ idx_in, int *__restrict__ idx_out,
                                                         const SahTask *__restrict__ tasks, const int *__restrict__ task_count,
                                                         SahTask *next_small, SahTask *next_large, SahTask *sub_list, int *next_count /* [0] small, [1] large */, int *sub_count,
                                                         float *compact, int *csize, int *parent, int *prim_slot, int halve, int level)
{
    constexpr int TPB = 16 / WPT;            // tasks per block
    constexpr int G = 64 * WPT;              // threads per task
    __shared__ unsigned s_bin[TPB][3][SAH_BINS][7];
    __shared__ unsigned s_redk[12];
    __shared__ int s_split[TPB][3];          // axis (-1: halve), plane, n_left
    __shared__ int s_wcount[16][2];
    __shared__ SahTask s_child[TPB][2];
    __shared__ int s_base[3];
    const int ntask = *task_count;
    if ((int)blockIdx.x * TPB >= ntask) return;   // block-uniform
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = wave / WPT;
    const int gt = threadIdx.x - slot * G;   // thread index within the task's group
    const int gwave = wave - slot * WPT;     // wave index within the group
    const int ti = blockIdx.x * TPB + slot;
    SahTask t = {0, 0, 0, 0};
    if (ti < ntask) t = tasks[ti];
    const int start = t.start, count = t.count, pre = t.pre;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // ---- A: node box and centroid bounds ---------------------------------------------------------
    float r[12]; sah_bounds_init(r);
    for (int i = gt; i < count; i += G) {
        const int p = idx_in[start + i];
        sah_bounds_add(r, sbox[2 * (size_t)p], sbox[2 * (size_t)p + 1]);
    }
    sah_wave_bounds(r);
    if (WPT > 1) {                           // across the 16 waves: keyed LDS min / max
        if (threadIdx.x < 12) s_redk[threadIdx.x] = sah_is_min(threadIdx.x) ? 0xffffffffu : 0u;
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 12; k++) { if (sah_is_min(k)) atomicMin(&s_redk[k], sah_key(r[k])); else atomicMax(&s_redk[k], sah_key(r[k])); }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 12; k++) r[k] = sah_unkey(s_redk[k]);
    }
    float scale[3];
    const float cmin[3] = {r[6], r[7], r[8]};
#pragma unroll
    for (int a = 0; a < 3; a++) { const float ext = r[9 + a] - r[6 + a]; scale[a] = ext > 0.0f ? (float)SAH_BINS / ext : 0.0f; }

    // ---- B: bins ---------------------------------------------------------------------------------
    for (int k = gt; k < 3 * SAH_BINS; k += G) {
        unsigned *b = &s_bin[slot][0][0][0] + k * 7;
        b[0] = b[1] = b[2] = 0xffffffffu; b[3] = b[4] = b[5] = 0u; b[6] = 0u;
    }
    __syncthreads();
    if (!halve) {
        for (int i = gt; i < count; i += G) {
            const int p = idx_in[start + i];
            sah_bin_add(s_bin[slot], sbox[2 * (size_t)p], sbox[2 * (size_t)p + 1], cmin, scale);
        }
    }
    __syncthreads();

    // ---- C: the cheapest plane (wave 0 of the group) -----------------------------------------------
    if (gwave == 0) {
        int ax, pl, nl_;
        sah_pick(s_bin[slot], count, halve != 0, lane, ax, pl, nl_);
        if (lane == 0) { s_split[slot][0] = ax; s_split[slot][1] = pl; s_split[slot][2] = nl_; }
    }
    __syncthreads();
    const int axis = s_split[slot][0], plane = s_split[slot][1], nl = s_split[slot][2], nr = count - nl;

    // ---- D: stable partition into the other index buffer; a child of one primitive is written as a leaf right here ----
    int off_l = 0, off_r = 0;
    const float sel_min = axis == 0 ? r[6] : (axis == 1 ? r[7] : r[8]), sel_scale = axis == 0 ? scale[0] : (axis == 1 ? scale[1] : scale[2]);
    for (int base = 0; base < count; base += G) {
        const int i = base + gt;
        const bool active = i < count;
        int p = 0; float4 a = make_float4(0, 0, 0, 0), b = a;
        bool left = false;
        if (active) {
            p = idx_in[start + i];
            a = sbox[2 * (size_t)p]; b = sbox[2 * (size_t)p + 1];
            if (axis < 0) left = i < nl;
            else { const float c = axis == 0 ? 0.5f * (a.x + b.x) : (axis == 1 ? 0.5f * (a.y + b.y) : 0.5f * (a.z + b.z)); left = sah_bin(c, sel_min, sel_scale) <= plane; }
        }
        const unsigned long long ml = __ballot(active && left), mr = __ballot(active && !left);
        int rank_l = __popcll(ml & lt_mask), rank_r = __popcll(mr & lt_mask), tot_l = __popcll(ml), tot_r = __popcll(mr);
        if (WPT > 1) {                       // block-uniform trip count: one task per block
            if (lane == 0) { s_wcount[wave][0] = tot_l; s_wcount[wave][1] = tot_r; }
            __syncthreads();
            int bl = 0, br = 0; tot_l = 0; tot_r = 0;
            for (int w = 0; w < 16; w++) { if (w < wave) { bl += s_wcount[w][0]; br += s_wcount[w][1]; } tot_l += s_wcount[w][0]; tot_r += s_wcount[w][1]; }
            rank_l += bl; rank_r += br;
            __syncthreads();
        }
        if (active) {
            const int dst = left ? start + off_l + rank_l : start + nl + off_r + rank_r;
            idx_out[dst] = p;
            if ((left && nl == 1) || (!left && nr == 1)) sah_write_leaf(compact, csize, prim_slot, left ? pre + 1 : pre + 2 * nl, dst, sorted_prims[p], a, b);
        }
        off_l += tot_l; off_r += tot_r;
    }
    if (gt == 0) {
        s_child[slot][0].count = 0; s_child[slot][1].count = 0;
        if (count >= 2) {
            sah_write_inner(compact, csize, parent, pre, pre + 2 * nl, count, r);
            if (nl >= 2) s_child[slot][0] = SahTask{start, nl, pre + 1, 0};
            if (nr >= 2) s_child[slot][1] = SahTask{start + nl, nr, pre + 2 * nl, 0};
        }
    }
    __syncthreads();
    // ---- next level's lists: one atomic per block and list --------------------------------------------
    if (threadIdx.x == 0) {
        int ns = 0, nlg = 0, nm = 0;
        for (int s = 0; s < TPB; s++) for (int k = 0; k < 2; k++) { const int c = s_child[s][k].count; if (c > SAH_LARGE) nlg++; else if (c > SAH_SUB) ns++; else if (c >= 2) nm++; }
        s_base[0] = ns ? atomicAdd(&next_count[0], ns) : 0;
        s_base[1] = nlg ? atomicAdd(&next_count[1], nlg) : 0;
        s_base[2] = nm ? atomicAdd(sub_count, nm) : 0;
        for (int s = 0; s < TPB; s++) for (int k = 0; k < 2; k++) {
            const int c = s_child[s][k].count;
            if (c > SAH_LARGE) next_large[s_base[1]++] = s_child[s][k];
            else if (c > SAH_SUB) next_small[s_base[0]++] = s_child[s][k];
            else if (c >= 2) { SahTask t = s_child[s][k]; t.pad = level + 1; sub_list[s_base[2]++] = t; }      // its range sits in the buffer level + 1 reads
        }
    }
}

// ---- whole subtrees of at most SAH_SUB primitives: one wave each, one launch -------------------------------------------------
// The bottom of the tree holds most of its nodes and, level by level, most of the launches (round 2: ten levels of k_sah_level<1> +
// k_sah_mini below 512 primitives, 0.7 of the 1.3 ms of a 100k-primitive build, nearly all of it launch latency).  A task of this list
// is finished where it is picked up: the wave keeps the subtree's open nodes of one level in LDS, splits them -- nodes of more
// than SAH_MINI primitives one after the other with the binned sweep of k_sah_level<1> (all 64 lanes on one node), smaller ones
// four at a time with the exact sweep of k_sah_mini (16 lanes each) -- and goes on with their children until none is left.  Same
// per-node decisions as the level-by-level kernels (same bins, same tie rules, same forced halving from level 64 on), hence the
// same tree; a node's range lives in the index buffer of its level's parity, as there.
constexpr int SAH_SUB_WAVES = 4;              // waves (tasks) per block
constexpr int SAH_SUB_LIST = SAH_SUB / 2;     // open nodes of one level of a subtree: each holds two primitives or more
__global__ __launch_bounds__(64 * SAH_SUB_WAVES) void k_sah_subtree(const float4 *__restrict__ sbox, const int *__restrict__ sorted_prims, int *idx_a, int *idx_b,
                                                                   const SahTask *__restrict__ tasks, const int *__restrict__ task_count,
                                                                   float *compact, int *csize, int *parent, int *prim_slot)
{
    __shared__ unsigned s_bin[SAH_SUB_WAVES][3][SAH_BINS][7];
    __shared__ SahTask s_list[SAH_SUB_WAVES][2][SAH_SUB_LIST];
    __shared__ int s_n[SAH_SUB_WAVES][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ti = blockIdx.x * SAH_SUB_WAVES + wave;
    if (ti >= *task_count) return;                               // wave-uniform; no block-wide barrier below
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#define SAH_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
    if (lane == 0) { s_list[wave][0][0] = tasks[ti]; s_n[wave][0] = 1; s_n[wave][1] = 0; }
    SAH_WSYNC();
    int level = tasks[ti].pad;                                   // the level that READS this range: idx[level & 1]
    for (int cur = 0;; cur ^= 1, level++) {
        const int n_open = s_n[wave][cur];
        if (n_open == 0) break;
        const int *idx_in = (level & 1) ? idx_b : idx_a;
        int *idx_out = (level & 1) ? idx_a : idx_b;
        const bool halve = level >= SAH_FORCE_HALVING_AFTER;
        if (lane == 0) s_n[wave][cur ^ 1] = 0;
        SAH_WSYNC();
        // ---- nodes of more than SAH_MINI primitives: the whole wave on one node (k_sah_level<1>'s steps A-D) ----
        for (int k = 0; k < n_open; k++) {
            const SahTask t = s_list[wave][cur][k];
            if (t.count <= SAH_MINI) continue;                   // wave-uniform
            const int start = t.start, count = t.count, pre = t.pre;
            float r[12]; sah_bounds_init(r);
            int p = 0; float4 a = make_float4(0, 0, 0, 0), b = a;
            const bool active = lane < count;                    // count <= 64: one primitive per lane
            if (active) { p = idx_in[start + lane]; a = sbox[2 * (size_t)p]; b = sbox[2 * (size_t)p + 1]; sah_bounds_add(r, a, b); }
            sah_wave_bounds(r);
            float scale[3];
            const float cmin[3] = {r[6], r[7], r[8]};
#pragma unroll
            for (int x = 0; x < 3; x++) { const float ext = r[9 + x] - r[6 + x]; scale[x] = ext > 0.0f ? (float)SAH_BINS / ext : 0.0f; }
            for (int q = lane; q < 3 * SAH_BINS; q += 64) {
                unsigned *bb = &s_bin[wave][0][0][0] + q * 7;
                bb[0] = bb[1] = bb[2] = 0xffffffffu; bb[3] = bb[4] = bb[5] = 0u; bb[6] = 0u;
            }
            SAH_WSYNC();
            if (!halve && active) sah_bin_add(s_bin[wave], a, b, cmin, scale);
            SAH_WSYNC();
            int axis, plane, nl;
            sah_pick(s_bin[wave], count, halve, lane, axis, plane, nl);
            const int nr = count - nl;
            bool left = false;
            if (active) {
                if (axis < 0) left = lane < nl;
                else {
                    const float sel_min = axis == 0 ? r[6] : (axis == 1 ? r[7] : r[8]), sel_scale = axis == 0 ? scale[0] : (axis == 1 ? scale[1] : scale[2]);
                    const float c = axis == 0 ? 0.5f * (a.x + b.x) : (axis == 1 ? 0.5f * (a.y + b.y) : 0.5f * (a.z + b.z));
                    left = sah_bin(c, sel_min, sel_scale) <= plane;
                }
            }
            const unsigned long long ml = __ballot(active && left), mr = __ballot(active && !left);
            if (active) {
                const int dst = left ? start + __popcll(ml & lt_mask) : start + nl + __popcll(mr & lt_mask);
                idx_out[dst] = p;
                if ((left && nl == 1) || (!left && nr == 1)) sah_write_leaf(compact, csize, prim_slot, left ? pre + 1 : pre + 2 * nl, dst, sorted_prims[p], a, b);
            }
            if (lane == 0) {
                sah_write_inner(compact, csize, parent, pre, pre + 2 * nl, count, r);
                int at = s_n[wave][cur ^ 1];
                if (nl >= 2) s_list[wave][cur ^ 1][at++] = SahTask{start, nl, pre + 1, 0};
                if (nr >= 2) s_list[wave][cur ^ 1][at++] = SahTask{start + nl, nr, pre + 2 * nl, 0};
                s_n[wave][cur ^ 1] = at;
            }
            SAH_WSYNC();
        }
        // ---- nodes of at most SAH_MINI primitives: four at a time, 16 lanes each (k_sah_mini's exact sweep) ----
        const int gl = lane & 15, gbase = lane & 48, grp = lane >> 4;
        // the mini nodes of this level, in list order: group g of a round takes the g-th remaining one
        int taken = 0;
        for (;;) {
            // find the next four mini tasks at or after position `taken` (wave-uniform scan of at most SAH_SUB_LIST entries)
            int pos[4] = {-1, -1, -1, -1}; int found = 0, scan = taken;
            for (; scan < n_open && found < 4; scan++) if (s_list[wave][cur][scan].count <= SAH_MINI) pos[found++] = scan;
            if (found == 0) break;
            taken = scan;
            const int my = grp == 0 ? pos[0] : (grp == 1 ? pos[1] : (grp == 2 ? pos[2] : pos[3]));
            SahTask t = {0, 0, 0, 0};
            if (my >= 0) t = s_list[wave][cur][my];
            const int start = t.start, count = t.count, pre = t.pre;
            const bool active = gl < count;
            int p = 0; float4 a = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 0.0f), b = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, 0.0f);
            if (active) { p = idx_in[start + gl]; a = sbox[2 * (size_t)p]; b = sbox[2 * (size_t)p + 1]; }
            const float c0 = 0.5f * (a.x + b.x), c1 = 0.5f * (a.y + b.y), c2 = 0.5f * (a.z + b.z);
            float lmn[3][3], lmx[3][3], rmn[3][3], rmx[3][3]; int nl3[3] = {0, 0, 0};
#pragma unroll
            for (int ax = 0; ax < 3; ax++)
#pragma unroll
                for (int q = 0; q < 3; q++) { lmn[ax][q] = 3.0e38f; lmx[ax][q] = -3.0e38f; rmn[ax][q] = 3.0e38f; rmx[ax][q] = -3.0e38f; }
            float box[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
            int cmax = count;                                    // the longest of the four groups sets the trip count: shuffles are executed by all
#pragma unroll
            for (int o = 32; o >= 16; o >>= 1) { const int v = __shfl_xor(cmax, o, 64); cmax = v > cmax ? v : cmax; }
            for (int j = 0; j < cmax; j++) {
                const int src = gbase + j;
                const float ax_ = __shfl(a.x, src, 64), ay_ = __shfl(a.y, src, 64), az_ = __shfl(a.z, src, 64);
                const float bx_ = __shfl(b.x, src, 64), by_ = __shfl(b.y, src, 64), bz_ = __shfl(b.z, src, 64);
                if (j >= count) continue;                        // group-uniform
                const float cj[3] = {0.5f * (ax_ + bx_), 0.5f * (ay_ + by_), 0.5f * (az_ + bz_)};
                const float ci[3] = {c0, c1, c2};
                const float mn[3] = {ax_, ay_, az_}, mx[3] = {bx_, by_, bz_};
#pragma unroll
                for (int q = 0; q < 3; q++) { box[q] = fminf(box[q], mn[q]); box[3 + q] = fmaxf(box[3 + q], mx[q]); }
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const bool lft = (cj[ax] < ci[ax]) || (cj[ax] == ci[ax] && j <= gl);
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        if (lft) { lmn[ax][q] = fminf(lmn[ax][q], mn[q]); lmx[ax][q] = fmaxf(lmx[ax][q], mx[q]); }
                        else { rmn[ax][q] = fminf(rmn[ax][q], mn[q]); rmx[ax][q] = fmaxf(rmx[ax][q], mx[q]); }
                    }
                    nl3[ax] += lft ? 1 : 0;
                }
            }
            float best = 3.0e38f; int best_id = 0x7fffffff;
            if (active && !halve) {
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const int nl_ = nl3[ax], nr_ = count - nl_;
                    if (nr_ <= 0) continue;
                    const float cost = sah_half_area(lmx[ax][0] - lmn[ax][0], lmx[ax][1] - lmn[ax][1], lmx[ax][2] - lmn[ax][2]) * (float)nl_ +
                                       sah_half_area(rmx[ax][0] - rmn[ax][0], rmx[ax][1] - rmn[ax][1], rmx[ax][2] - rmn[ax][2]) * (float)nr_;
                    const int id = ax * SAH_MINI + gl;
                    if (cost < best || (cost == best && id < best_id)) { best = cost; best_id = id; }
                }
            }
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(best_id, o, 64);
                if (ob < best || (ob == best && oi < best_id)) { best = ob; best_id = oi; }
            }
            // the winner's key, from its lane (every lane executes the shuffle; groups without a winner read lane gbase)
            const int wax = best_id != 0x7fffffff ? best_id / SAH_MINI : 0, wl = best_id != 0x7fffffff ? best_id - wax * SAH_MINI : 0;
            const float mine = wax == 0 ? c0 : (wax == 1 ? c1 : c2);
            const float key = __shfl(mine, gbase + wl, 64);
            bool left;
            if (best_id != 0x7fffffff) left = active && ((mine < key) || (mine == key && gl <= wl));
            else left = active && gl < count / 2;                // forced halving, or no task in this group
            const unsigned long long gm = 0xffffull << gbase;
            const unsigned long long ml = __ballot(left) & gm, mr = __ballot(active && !left) & gm;
            const unsigned long long below = (1ull << lane) - 1ull;
            const int nl = __popcll(ml), nr = __popcll(mr);
            if (active) {
                const int dst = left ? start + __popcll(ml & below) : start + nl + __popcll(mr & below);
                idx_out[dst] = p;
                if ((left && nl == 1) || (!left && nr == 1)) sah_write_leaf(compact, csize, prim_slot, left ? pre + 1 : pre + 2 * nl, dst, sorted_prims[p], a, b);
            }
            if (gl == 0 && count >= 2) sah_write_inner(compact, csize, parent, pre, pre + 2 * nl, count, box);
            // children: the four group leaders append in group order (one lane does it, reading the others' results by shuffle)
            const int g_nl[4] = {__shfl(nl, 0, 64), __shfl(nl, 16, 64), __shfl(nl, 32, 64), __shfl(nl, 48, 64)};
            const int g_cnt[4] = {__shfl(count, 0, 64), __shfl(count, 16, 64), __shfl(count, 32, 64), __shfl(count, 48, 64)};
            const int g_start[4] = {__shfl(start, 0, 64), __shfl(start, 16, 64), __shfl(start, 32, 64), __shfl(start, 48, 64)};
            const int g_pre[4] = {__shfl(pre, 0, 64), __shfl(pre, 16, 64), __shfl(pre, 32, 64), __shfl(pre, 48, 64)};
            if (lane == 0) {
                int at = s_n[wave][cur ^ 1];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    if (g_cnt[g] < 2) continue;
                    const int l_ = g_nl[g], r_ = g_cnt[g] - l_;
                    if (l_ >= 2) s_list[wave][cur ^ 1][at++] = SahTask{g_start[g], l_, g_pre[g] + 1, 0};
                    if (r_ >= 2) s_list[wave][cur ^ 1][at++] = SahTask{g_start[g] + l_, r_, g_pre[g] + 2 * l_, 0};
                }
                s_n[wave][cur ^ 1] = at;
            }
            SAH_WSYNC();
        }
        SAH_WSYNC();
    }
#undef SAH_WSYNC
}

// ---- huge nodes: level counters lc[] = (small, large, huge, chunks) ---------------------------------------------------
// bins of one chunk, merged into the node's
__global__ __launch_bounds__(SAH_CHUNK) void k_sah_huge_bin(const float4 *__restrict__ sbox, const int *__restrict__ idx_in, const SahHuge *__restrict__ huge,
                                                             const int *__restrict__ lc, const int *__restrict__ chunk_task, unsigned *hbins, int *chunk_cnt, int halve)
{
    __shared__ unsigned s_bin[3][SAH_BINS][7];
    const int c = blockIdx.x;
    if (c >= lc[3] || halve) return;
    const int slot = chunk_task[c];
    const SahHuge *h = huge + slot;
    const int i = (c - h->first_chunk) * SAH_CHUNK + (int)threadIdx.x;
    float cmin[3], scale[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { cmin[a] = sah_unkey(h->bounds[6 + a]); const float ext = sah_unkey(h->bounds[9 + a]) - cmin[a]; scale[a] = ext > 0.0f ? (float)SAH_BINS / ext : 0.0f; }
    for (int k = threadIdx.x; k < 3 * SAH_BINS; k += SAH_CHUNK) {
        unsigned *b = &s_bin[0][0][0] + k * 7;
        b[0] = b[1] = b[2] = 0xffffffffu; b[3] = b[4] = b[5] = 0u; b[6] = 0u;
    }
    __syncthreads();
    if (i < h->count) {
        const int p = idx_in[h->start + i];
        sah_bin_add(s_bin, sbox[2 * (size_t)p], sbox[2 * (size_t)p + 1], cmin, scale);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * SAH_BINS; k += SAH_CHUNK) {
        const unsigned *b = &s_bin[0][0][0] + k * 7;
        chunk_cnt[(size_t)c * 3 * SAH_BINS + k] = (int)b[6];
        if (b[6]) {
            unsigned *g = hbins + (size_t)slot * SAH_BIN_WORDS + k * 7;
            atomicMin(&g[0], b[0]); atomicMin(&g[1], b[1]); atomicMin(&g[2], b[2]);
            atomicMax(&g[3], b[3]); atomicMax(&g[4], b[4]); atomicMax(&g[5], b[5]);
            atomicAdd(&g[6], b[6]);
        }
    }
}

// plane of a huge node, output offsets of its chunks, its row, its children
__global__ __launch_bounds__(64) void k_sah_huge_eval(SahHuge *huge, const int *__restrict__ lc, const unsigned *__restrict__ hbins, const int *__restrict__ chunk_cnt,
                                                       int *chunk_base, SahHuge *next_huge, unsigned *next_hbins, int *next_chunk_task,
                                                       SahTask *next_small, SahTask *next_large, SahTask *sub_list, int *nc /* next level's counters */, int *sub_count,
                                                       float *compact, int *csize, int *parent, int halve, int level)
{
    __shared__ unsigned s_bin[3][SAH_BINS][7];
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= lc[2]) return;
    SahHuge *h = huge + slot;
    const int start = h->start, count = h->count, pre = h->pre, first = h->first_chunk;
    for (int k = lane; k < SAH_BIN_WORDS; k += 64) (&s_bin[0][0][0])[k] = hbins[(size_t)slot * SAH_BIN_WORDS + k];
    __syncthreads();
    int axis, plane, nl;
    sah_pick(s_bin, count, halve != 0, lane, axis, plane, nl);
    const int nr = count - nl, nchunks = (count + SAH_CHUNK - 1) / SAH_CHUNK;
    int carry_l = 0, carry_r = 0;
    for (int base = 0; base < nchunks; base += 64) {
        const int k = base + lane;
        int len = 0, left = 0;
        if (k < nchunks) {
            len = count - k * SAH_CHUNK; if (len > SAH_CHUNK) len = SAH_CHUNK;
            if (axis < 0) { left = nl - k * SAH_CHUNK; left = left < 0 ? 0 : (left > len ? len : left); }
            else { const int *cc = chunk_cnt + (size_t)(first + k) * 3 * SAH_BINS + axis * SAH_BINS; for (int b = 0; b <= plane; b++) left += cc[b]; }
        }
        int il = left, ir = len - left;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int vl = __shfl_up(il, o, 64), vr = __shfl_up(ir, o, 64); if (lane >= o) { il += vl; ir += vr; } }
        if (k < nchunks) { chunk_base[2 * (size_t)(first + k)] = carry_l + il - left; chunk_base[2 * (size_t)(first + k) + 1] = carry_r + ir - (len - left); }
        carry_l += __shfl(il, 63, 64); carry_r += __shfl(ir, 63, 64);
    }
    int child_slot[2] = {-1, -1}, child_first[2] = {0, 0};
    if (lane == 0) {
        const float box[6] = {sah_unkey(h->bounds[0]), sah_unkey(h->bounds[1]), sah_unkey(h->bounds[2]), sah_unkey(h->bounds[3]), sah_unkey(h->bounds[4]), sah_unkey(h->bounds[5])};
        sah_write_inner(compact, csize, parent, pre, pre + 2 * nl, count, box);
        for (int side = 0; side < 2; side++) {
            const SahTask t = side == 0 ? SahTask{start, nl, pre + 1, 0} : SahTask{start + nl, nr, pre + 2 * nl, 0};
            if (t.count > SAH_HUGE) {
                const int s = atomicAdd(&nc[2], 1), nch = (t.count + SAH_CHUNK - 1) / SAH_CHUNK, fc = atomicAdd(&nc[3], nch);
                SahHuge n = {};
                n.start = t.start; n.count = t.count; n.pre = t.pre; n.first_chunk = fc;
                for (int k = 0; k < 12; k++) n.bounds[k] = sah_is_min(k) ? 0xffffffffu : 0u;
                n.child[0] = n.child[1] = -1;
                next_huge[s] = n;
                child_slot[side] = s; child_first[side] = fc;
            } else if (t.count > SAH_LARGE) next_large[atomicAdd(&nc[1], 1)] = t;
            else if (t.count > SAH_SUB) next_small[atomicAdd(&nc[0], 1)] = t;
            else if (t.count >= 2) { SahTask u = t; u.pad = level + 1; sub_list[atomicAdd(sub_count, 1)] = u; }
        }
        h->axis = axis; h->plane = plane; h->nl = nl; h->child[0] = child_slot[0]; h->child[1] = child_slot[1];
    }
    for (int side = 0; side < 2; side++) {
        const int s = __shfl(child_slot[side], 0, 64), fc = __shfl(child_first[side], 0, 64);
        if (s < 0) continue;
        const int cnt = side == 0 ? nl : nr, nch = (cnt + SAH_CHUNK - 1) / SAH_CHUNK;
        for (int k = lane; k < nch; k += 64) next_chunk_task[fc + k] = s;
        for (int k = lane; k < SAH_BIN_WORDS; k += 64) next_hbins[(size_t)s * SAH_BIN_WORDS + k] = (k % 7 < 3) ? 0xffffffffu : 0u;
    }
}

// partition of one chunk of a huge node (stable: the chunk's output offsets come from k_sah_huge_eval), the bounds of huge children
__global__ __launch_bounds__(SAH_CHUNK) void k_sah_huge_part(const float4 *__restrict__ sbox, const int *__restrict__ sorted_prims, const int *__restrict__ idx_in, int *__restrict__ idx_out,
                                                              const SahHuge *__restrict__ huge, const int *__restrict__ lc, const int *__restrict__ chunk_task,
                                                              const int *__restrict__ chunk_base, SahHuge *next_huge, float *compact, int *csize, int *prim_slot)
{
    __shared__ int s_wcount[16][2];
    __shared__ unsigned s_b[2][12];
    const int c = blockIdx.x;
    if (c >= lc[3]) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const SahHuge *h = huge + chunk_task[c];
    const int start = h->start, count = h->count, pre = h->pre, axis = h->axis, plane = h->plane, nl = h->nl, nr = count - nl;
    const int i = (c - h->first_chunk) * SAH_CHUNK + (int)threadIdx.x;
    const bool active = i < count;
    if (threadIdx.x < 24) s_b[threadIdx.x / 12][threadIdx.x % 12] = sah_is_min(threadIdx.x % 12) ? 0xffffffffu : 0u;
    int p = 0; float4 a = make_float4(0, 0, 0, 0), b = a;
    bool left = false;
    if (active) {
        p = idx_in[start + i];
        a = sbox[2 * (size_t)p]; b = sbox[2 * (size_t)p + 1];
        if (axis < 0) left = i < nl;
        else {
            const float cmin = sah_unkey(h->bounds[6 + axis]), ext = sah_unkey(h->bounds[9 + axis]) - cmin, scale = ext > 0.0f ? (float)SAH_BINS / ext : 0.0f;
            const float cc = axis == 0 ? 0.5f * (a.x + b.x) : (axis == 1 ? 0.5f * (a.y + b.y) : 0.5f * (a.z + b.z));
            left = sah_bin(cc, cmin, scale) <= plane;
        }
    }
    const unsigned long long ml = __ballot(active && left), mr = __ballot(active && !left);
    int rank_l = __popcll(ml & lt_mask), rank_r = __popcll(mr & lt_mask);
    if (lane == 0) { s_wcount[wave][0] = __popcll(ml); s_wcount[wave][1] = __popcll(mr); }
    __syncthreads();
    for (int w = 0; w < wave; w++) { rank_l += s_wcount[w][0]; rank_r += s_wcount[w][1]; }
    if (active) {
        const int dst = left ? start + chunk_base[2 * (size_t)c] + rank_l : start + nl + chunk_base[2 * (size_t)c + 1] + rank_r;
        idx_out[dst] = p;
        if ((left && nl == 1) || (!left && nr == 1)) sah_write_leaf(compact, csize, prim_slot, left ? pre + 1 : pre + 2 * nl, dst, sorted_prims[p], a, b);
    }
    // bounds of the children that are huge again (the others measure themselves, k_sah_level)
    for (int side = 0; side < 2; side++) {
        const int cs = h->child[side];
        if (cs < 0) continue;                            // block-uniform
        float r[12]; sah_bounds_init(r);
        if (active && (left == (side == 0))) sah_bounds_add(r, a, b);
        sah_wave_bounds(r);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 12; k++) { if (sah_is_min(k)) atomicMin(&s_b[side][k], sah_key(r[k])); else atomicMax(&s_b[side][k], sah_key(r[k])); }
        }
    }
    __syncthreads();
    if (threadIdx.x < 24) {
        const int side = threadIdx.x / 12, k = threadIdx.x % 12, cs = h->child[side];
        if (cs >= 0) { if (sah_is_min(k)) atomicMin(&next_huge[cs].bounds[k], s_b[side][k]); else atomicMax(&next_huge[cs].bounds[k], s_b[side][k]); }
    }
}

// Builds the tree over the primitives in the order `sorted_prims` (Morton order: neighbours in the array are neighbours in
// space, so the box reads of the first levels are streams) into c->sah_compact / c->sah_csize.  Work on c->stream.
int sah_build(tirt_ctx *c, const int *sorted_prims, float sph_pad_abs)
{
    const int n = c->n, N = 2 * n - 1;
    hipStream_t st = c->stream;
    const size_t small_cap = (size_t)n / SAH_SUB + 2, large_cap = (size_t)n / SAH_LARGE + 2, huge_cap = (size_t)n / SAH_HUGE + 2, sub_cap = (size_t)n / 2 + 2;
    const size_t chunk_cap = (size_t)n / SAH_CHUNK + huge_cap + 2;
    // scratch: two task lists of each size, the list of whole subtrees, huge node records + their bins, chunk tables
    const size_t task_bytes = sizeof(SahTask) * (2 * (small_cap + large_cap) + sub_cap);
    const size_t huge_bytes = sizeof(SahHuge) * 2 * huge_cap + sizeof(unsigned) * 2 * huge_cap * SAH_BIN_WORDS;
    const size_t chunk_bytes = sizeof(int) * chunk_cap * (2 /* task, both levels */ + 3 * SAH_BINS + 2);
    if (c->sah_compact.ensure(sizeof(float) * (size_t)N * CPN_VEC) || c->sah_csize.ensure(sizeof(int) * (size_t)N) || c->sah_parent.ensure(sizeof(int) * (size_t)N) ||
        c->sah_box.ensure(sizeof(float4) * 2 * (size_t)n) || c->sah_idx.ensure(sizeof(int) * 2 * (size_t)n) ||
        c->sah_tasks.ensure(task_bytes + huge_bytes + chunk_bytes) || c->sah_counts.ensure(sizeof(int) * SAH_CNT * (SAH_MAX_LEVELS + 2))) return TIRT_ERR_HIP;
    SceneView sv = scene_view(c);
    float4 *sbox = c->sah_box.as<float4>();
    int *idx[2] = {c->sah_idx.as<int>(), c->sah_idx.as<int>() + n};
    SahTask *small[2] = {c->sah_tasks.as<SahTask>(), c->sah_tasks.as<SahTask>() + small_cap};
    SahTask *large[2] = {small[1] + small_cap, small[1] + small_cap + large_cap};
    SahTask *sub = large[1] + large_cap;
    SahHuge *huge[2] = {(SahHuge *)(sub + sub_cap), (SahHuge *)(sub + sub_cap) + huge_cap};
    unsigned *hbins[2] = {(unsigned *)(huge[1] + huge_cap), (unsigned *)(huge[1] + huge_cap) + huge_cap * SAH_BIN_WORDS};
    int *chunk_task[2] = {(int *)(hbins[1] + huge_cap * SAH_BIN_WORDS), (int *)(hbins[1] + huge_cap * SAH_BIN_WORDS) + chunk_cap};
    int *chunk_cnt = chunk_task[1] + chunk_cap, *chunk_base = chunk_cnt + chunk_cap * 3 * SAH_BINS;
    int *counts = c->sah_counts.as<int>();                 // counts[SAH_CNT * level + (0 small | 1 large | 2 huge | 3 chunks)]
    int *sub_count = counts + SAH_CNT * (SAH_MAX_LEVELS + 1);      // (a row of its own behind the levels')
    float *compact = c->sah_compact.as<float>(); int *csize = c->sah_csize.as<int>(), *parent = c->sah_parent.as<int>(), *prim_slot = c->prim_slot.as<int>();

    TIRT_HIP(hipMemsetAsync(counts, 0, sizeof(int) * SAH_CNT * (SAH_MAX_LEVELS + 2), st));
    SahHuge root_huge = {};
    root_huge.count = n; root_huge.child[0] = root_huge.child[1] = -1;
    for (int k = 0; k < 12; k++) root_huge.bounds[k] = ((k < 3) || (k >= 6 && k < 9)) ? 0xffffffffu : 0u;
    TIRT_HIP(hipMemcpyAsync(huge[0], &root_huge, sizeof(root_huge), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_sah_prim_boxes, dim3((n + 255) / 256), dim3(256), 0, st, sv, sorted_prims, sbox, idx[0], huge[0], hbins[0], sph_pad_abs);
    const SahTask root = {0, n, 0, 0};                     // (pad = the level that reads it: 0)
    int first_counts[SAH_CNT] = {0, 0, 0, 0, 0, 0, 0, 0};
    int first_sub = 0;
    if (n > SAH_HUGE) {
        first_counts[2] = 1; first_counts[3] = (n + SAH_CHUNK - 1) / SAH_CHUNK;
        TIRT_HIP(hipMemsetAsync(chunk_task[0], 0, sizeof(int) * (size_t)first_counts[3], st));       // every chunk belongs to slot 0
    } else if (n > SAH_LARGE) { first_counts[1] = 1; TIRT_HIP(hipMemcpyAsync(large[0], &root, sizeof(root), hipMemcpyHostToDevice, st)); }
    else if (n > SAH_SUB) { first_counts[0] = 1; TIRT_HIP(hipMemcpyAsync(small[0], &root, sizeof(root), hipMemcpyHostToDevice, st)); }
    else if (n >= 2) { first_sub = 1; TIRT_HIP(hipMemcpyAsync(sub, &root, sizeof(root), hipMemcpyHostToDevice, st)); }
    TIRT_HIP(hipMemcpyAsync(counts, first_counts, sizeof(first_counts), hipMemcpyHostToDevice, st));
    if (first_sub) TIRT_HIP(hipMemcpyAsync(sub_count, &first_sub, sizeof(int), hipMemcpyHostToDevice, st));

    // Levels are launched in rounds, a blocking read of the counters ends a round (~50 us; an idle level costs its empty launches).
    // Round 1: as many levels as a balanced tree needs to get every node below SAH_HUGE, plus two -- after it the three launches per
    // level of the huge nodes are usually over --; round 2: down to SAH_SUB, plus three; further rounds of 8 levels for what is left
    // of an unbalanced tree.  What falls below SAH_SUB on the way is not split level by level: it waits in the `sub` list for the
    // one launch of k_sah_subtree at the end.
    bool any_huge = n > SAH_HUGE, any_large = n > SAH_LARGE, any_small = n > SAH_SUB;
    int level = 0, host_counts[SAH_CNT * (SAH_MAX_LEVELS + 2)], round_no = 0;
    auto levels_to = [&](long target) { int k = 0; for (long m = n; m > target; m = (m + 1) / 2) k++; return k; };
    while (any_huge || any_large || any_small) {
        int round = 8;
        if (round_no == 0) round = any_huge ? levels_to(SAH_HUGE) + 2 : levels_to(SAH_SUB) + 3;
        else if (round_no == 1) { round = levels_to(SAH_SUB) + 3 - level; if (round < 4) round = 4; }
        round_no++;
        const int until = (level + round < SAH_MAX_LEVELS) ? level + round : SAH_MAX_LEVELS;
        for (; level < until; level++) {
            const int in = level & 1, out = in ^ 1, halve = level >= SAH_FORCE_HALVING_AFTER ? 1 : 0;
            int *lc = counts + SAH_CNT * level, *nc = counts + SAH_CNT * (level + 1);
            long cap = 1; for (int k = 0; k < level && cap < n; k++) cap *= 2;              // a level holds at most 2^level nodes
            const long cap_small = cap < (long)small_cap ? cap : (long)small_cap, cap_large = cap < (long)large_cap ? cap : (long)large_cap;
            const long cap_huge = cap < (long)huge_cap ? cap : (long)huge_cap;
            if (any_huge) {
                hipLaunchKernelGGL(k_sah_huge_bin, dim3((unsigned)chunk_cap), dim3(SAH_CHUNK), 0, st, sbox, idx[in], huge[in], lc, chunk_task[in], hbins[in], chunk_cnt, halve);
                hipLaunchKernelGGL(k_sah_huge_eval, dim3((unsigned)cap_huge), dim3(64), 0, st, huge[in], lc, hbins[in], chunk_cnt, chunk_base, huge[out], hbins[out], chunk_task[out],
                                   small[out], large[out], sub, nc, sub_count, compact, csize, parent, halve, level);
                hipLaunchKernelGGL(k_sah_huge_part, dim3((unsigned)chunk_cap), dim3(SAH_CHUNK), 0, st, sbox, sorted_prims, idx[in], idx[out], huge[in], lc, chunk_task[in], chunk_base,
                                   huge[out], compact, csize, prim_slot);
            }
            if (any_large)
                hipLaunchKernelGGL(k_sah_level<16>, dim3((unsigned)cap_large), dim3(SAH_BLOCK), 0, st, sbox, sorted_prims, idx[in], idx[out], large[in], lc + 1,
                                   small[out], large[out], sub, nc, sub_count, compact, csize, parent, prim_slot, halve, level);
            if (any_small)
                hipLaunchKernelGGL(k_sah_level<1>, dim3((unsigned)((cap_small + 15) / 16)), dim3(SAH_BLOCK), 0, st, sbox, sorted_prims, idx[in], idx[out], small[in], lc,
                                   small[out], large[out], sub, nc, sub_count, compact, csize, parent, prim_slot, halve, level);
        }
        TIRT_HIP(hipMemcpyAsync(host_counts, counts, sizeof(host_counts), hipMemcpyDeviceToHost, st));
        TIRT_HIP(hipStreamSynchronize(st));
        const int *lc = host_counts + SAH_CNT * level;
        any_huge = lc[2] > 0; any_large = any_huge || lc[1] > 0; any_small = any_large || lc[0] > 0;
        TIRT_REQUIRE(level < SAH_MAX_LEVELS || !(any_small), "tirt_lbvh_build: traversal tree deeper than 160 levels");
    }
    int n_sub = first_sub;
    if (round_no > 0) n_sub = host_counts[SAH_CNT * (SAH_MAX_LEVELS + 1)];
    if (n_sub > 0)
        hipLaunchKernelGGL(k_sah_subtree, dim3((unsigned)((n_sub + SAH_SUB_WAVES - 1) / SAH_SUB_WAVES)), dim3(64 * SAH_SUB_WAVES), 0, st, sbox, sorted_prims, idx[0], idx[1],
                           sub, sub_count, compact, csize, parent, prim_slot);
    c->sah_levels = level;
    TIRT_HIP(hipGetLastError());
    return TIRT_OK;
}

}  // namespace tirt
