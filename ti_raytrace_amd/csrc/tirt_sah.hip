// libtirt -- the TRAVERSAL tree: a binned surface-area-heuristic binary tree over the scene's primitives, built on the device,
// which build_wide (tirt_lbvh.hip) collapses into the quantised 4-wide nodes k_trace walks.
//
// Why a second tree.  The reference's structure is its LBVH (accel/LBvh.py:229-467), and its RESULTS are defined by that
// tree: a leaf is visited iff every proper ancestor's box passes `slabs`, ties go to the visiting order.  k_trace
// re-establishes exactly that condition for every candidate hit (leaf box / `cparent` chain of the reference tree, tie rule
// on the reference's leaf index), so the structure that FINDS the candidates may be any hierarchy whose boxes contain their
// leaves' boxes.  A Morton-split tree is a poor one to walk: measured on the same kernel, the SAH tree costs 26.8 node
// visits per ray of the headline scene instead of 32.1 (+9 % rays/s), 9.6 instead of 14.5 on the Teapot (+13 %), 11.0
// instead of 14.5 on the Veach scene (+15 %) -- with bit-identical films (tools/exp/sah_tree.py; PLOC, also tried there,
// is worse than binned SAH on all three).
//
// Algorithm: top-down, level-synchronous, one primitive per leaf (as the reference).  A node is a contiguous range of a
// primitive-index array; its pre-order index is known when it is created (left = self + 1, right = self + 2 * n_left: a
// subtree of k leaves has 2k - 1 nodes), so nodes are written straight into the `compact` layout k_wide_level reads, with
// no numbering pass and no atomics for node ids -- the tree is deterministic.  Per node: (A) box and centroid bounds,
// (B) 32 bins per axis by centroid in LDS (integer-keyed min / max / count atomics), (C) the cheapest of the 3 x 31
// candidate planes by area(L) * n_L + area(R) * n_R, (D) a stable partition of the range into the other index buffer.
// All centroids in one bin on every axis (duplicates), or a tree deeper than 64 levels: the range is halved instead.
// Small nodes (<= SAH_LARGE primitives) take one wave each, 16 to a block; large ones a 1024-thread block each.  The
// next level's task lists are appended with one atomic per block and list (same-address atomics: ~11 ns each).
#include "tirt_internal.h"
#include "tirt_device.h"

namespace tirt {

constexpr int SAH_BINS = 32;
constexpr int SAH_LARGE = 4096;          // more primitives than this: a whole block works on the node
constexpr int SAH_BLOCK = 1024;
constexpr int SAH_FORCE_HALVING_AFTER = 64;

struct SahTask { int start, count, pre, pad; };

TD unsigned sah_key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }      // unsigned order = float order
TD float sah_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void k_sah_prim_boxes(SceneView s, float4 *box)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    const int *pr = s.primitive + (size_t)i * PRI_VEC;
    v3 mn, mx;
    if (pr[0] == PRIMITIVE_TRI) {                       // accel/LBvh.py:397-415
        const v3 a = vtx_pos(s, pr[1]), b = vtx_pos(s, pr[1] + 1), c = vtx_pos(s, pr[1] + 2);
        mn = V(fminf(fminf(a.x, b.x), c.x), fminf(fminf(a.y, b.y), c.y), fminf(fminf(a.z, b.z), c.z));
        mx = V(fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z));
    } else {                                            // accel/LBvh.py:416-426: centre -+ r
        const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;
        mn = V(sh[1] - sh[4], sh[2] - sh[4], sh[3] - sh[4]); mx = V(sh[1] + sh[4], sh[2] + sh[4], sh[3] + sh[4]);
    }
    box[2 * (size_t)i] = make_float4(mn.x, mn.y, mn.z, 0.0f);
    box[2 * (size_t)i + 1] = make_float4(mx.x, mx.y, mx.z, 0.0f);
}

TD float sah_half_area(float dx, float dy, float dz) { return dx * dy + dy * dz + dz * dx; }

// WPT waves per task: 1 (16 tasks per block) or 16 (one task per block).  Every thread of the block reaches every
// barrier; a task slot without a task just has count == 0.
template <int WPT>
__global__ __launch_bounds__(SAH_BLOCK) void k_sah_level(const float4 *__restrict__ box, const int *__restrict__ idx_in, int *__restrict__ idx_out,
                                                         const SahTask *__restrict__ tasks, const int *__restrict__ task_count,
                                                         SahTask *next_small, SahTask *next_large, int *next_count /* [0] small, [1] large */,
                                                         float *compact, int *csize, int halve)
{
    constexpr int TPB = 16 / WPT;            // tasks per block
    constexpr int G = 64 * WPT;              // threads per task
    __shared__ unsigned s_bin[TPB][3][SAH_BINS][7];
    __shared__ unsigned s_redk[12];
    __shared__ int s_split[TPB][3];          // axis (-1: halve), plane, n_left
    __shared__ int s_wcount[16][2];
    __shared__ SahTask s_child[TPB][2];
    __shared__ int s_base[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = wave / WPT;
    const int gt = threadIdx.x - slot * G;   // thread index within the task's group
    const int gwave = wave - slot * WPT;     // wave index within the group
    const int ti = blockIdx.x * TPB + slot;
    const int ntask = *task_count;
    SahTask t = {0, 0, 0, 0};
    if (ti < ntask) t = tasks[ti];
    const int start = t.start, count = t.count, pre = t.pre;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // ---- A: node box and centroid bounds ---------------------------------------------------------
    float r[12] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = gt; i < count; i += G) {
        const int p = idx_in[start + i];
        const float4 a = box[2 * (size_t)p], b = box[2 * (size_t)p + 1];
        r[0] = fminf(r[0], a.x); r[1] = fminf(r[1], a.y); r[2] = fminf(r[2], a.z);
        r[3] = fmaxf(r[3], b.x); r[4] = fmaxf(r[4], b.y); r[5] = fmaxf(r[5], b.z);
        const float cx = 0.5f * (a.x + b.x), cy = 0.5f * (a.y + b.y), cz = 0.5f * (a.z + b.z);
        r[6] = fminf(r[6], cx); r[7] = fminf(r[7], cy); r[8] = fminf(r[8], cz);
        r[9] = fmaxf(r[9], cx); r[10] = fmaxf(r[10], cy); r[11] = fmaxf(r[11], cz);
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const bool is_min = (k < 3) || (k >= 6 && k < 9);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const float v = __shfl_xor(r[k], o, 64); r[k] = is_min ? fminf(r[k], v) : fmaxf(r[k], v); }
    }
    if (WPT > 1) {                           // across the 16 waves: keyed LDS min / max
        if (threadIdx.x < 12) s_redk[threadIdx.x] = ((threadIdx.x < 3) || (threadIdx.x >= 6 && threadIdx.x < 9)) ? 0xffffffffu : 0u;
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 12; k++) { if ((k < 3) || (k >= 6 && k < 9)) atomicMin(&s_redk[k], sah_key(r[k])); else atomicMax(&s_redk[k], sah_key(r[k])); }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 12; k++) r[k] = sah_unkey(s_redk[k]);
    }
    float scale[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { const float ext = r[9 + a] - r[6 + a]; scale[a] = ext > 0.0f ? (float)SAH_BINS / ext : 0.0f; }
#define SAH_BIN(c, a) ({ int b__ = (int)(((c) - r[6 + (a)]) * scale[a]); b__ < 0 ? 0 : (b__ >= SAH_BINS ? SAH_BINS - 1 : b__); })

    // ---- B: bins ---------------------------------------------------------------------------------
    for (int k = gt; k < 3 * SAH_BINS; k += G) {
        unsigned *b = &s_bin[slot][0][0][0] + k * 7;
        b[0] = b[1] = b[2] = 0xffffffffu; b[3] = b[4] = b[5] = 0u; b[6] = 0u;
    }
    __syncthreads();
    if (!halve) {
        for (int i = gt; i < count; i += G) {
            const int p = idx_in[start + i];
            const float4 a = box[2 * (size_t)p], b = box[2 * (size_t)p + 1];
            const unsigned k0 = sah_key(a.x), k1 = sah_key(a.y), k2 = sah_key(a.z), k3 = sah_key(b.x), k4 = sah_key(b.y), k5 = sah_key(b.z);
            const float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                unsigned *q = s_bin[slot][ax][SAH_BIN(c[ax], ax)];
                atomicMin(&q[0], k0); atomicMin(&q[1], k1); atomicMin(&q[2], k2);
                atomicMax(&q[3], k3); atomicMax(&q[4], k4); atomicMax(&q[5], k5);
                atomicAdd(&q[6], 1u);
            }
        }
    }
    __syncthreads();

    // ---- C: the cheapest plane (wave 0 of the group: candidates gt and gt + 64 of 3 x 32) -----------
    if (gwave == 0) {
        float best = 3.0e38f; int best_id = 0x7fffffff, best_nl = 0;
        if (!halve && count >= 2) {
            for (int cand = lane; cand < 3 * SAH_BINS; cand += 64) {
                const int ax = cand / SAH_BINS, sp = cand - ax * SAH_BINS;
                if (sp == SAH_BINS - 1) continue;
                float lmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, lmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, rmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, rmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
                unsigned nl = 0, nr = 0;
                for (int b = 0; b < SAH_BINS; b++) {
                    const unsigned *q = s_bin[slot][ax][b];
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
        if (lane == 0) {
            if (best_id != 0x7fffffff) { s_split[slot][0] = best_id / SAH_BINS; s_split[slot][1] = best_id % SAH_BINS; s_split[slot][2] = best_nl; }
            else { s_split[slot][0] = -1; s_split[slot][1] = 0; s_split[slot][2] = count / 2; }
        }
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
            a = box[2 * (size_t)p]; b = box[2 * (size_t)p + 1];
            if (axis < 0) left = i < nl;
            else { const float c = axis == 0 ? 0.5f * (a.x + b.x) : (axis == 1 ? 0.5f * (a.y + b.y) : 0.5f * (a.z + b.z)); int bb = (int)((c - sel_min) * sel_scale); bb = bb < 0 ? 0 : (bb >= SAH_BINS ? SAH_BINS - 1 : bb); left = bb <= plane; }
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
            if ((left && nl == 1) || (!left && nr == 1)) {
                const int leaf = left ? pre + 1 : pre + 2 * nl;
                float *row = compact + (size_t)leaf * CPN_VEC;
                row[0] = 1.0f; row[1] = (float)p; row[2] = a.x; row[3] = a.y; row[4] = a.z; row[5] = b.x; row[6] = b.y; row[7] = b.z; row[8] = 0.0f;
                csize[leaf] = 1;
            }
        }
        off_l += tot_l; off_r += tot_r;
    }
    if (gt == 0) {
        s_child[slot][0].count = 0; s_child[slot][1].count = 0;
        if (count >= 2) {
            float *row = compact + (size_t)pre * CPN_VEC;
            row[0] = 0.0f; row[1] = (float)(pre + 2 * nl);
            row[2] = r[0]; row[3] = r[1]; row[4] = r[2]; row[5] = r[3]; row[6] = r[4]; row[7] = r[5]; row[8] = 0.0f;
            csize[pre] = 2 * count - 1;
            if (nl >= 2) s_child[slot][0] = SahTask{start, nl, pre + 1, 0};
            if (nr >= 2) s_child[slot][1] = SahTask{start + nl, nr, pre + 2 * nl, 0};
        }
    }
    __syncthreads();
    // ---- next level's lists: one atomic per block and list --------------------------------------------
    if (threadIdx.x == 0) {
        int ns = 0, nlg = 0;
        for (int s = 0; s < TPB; s++) for (int k = 0; k < 2; k++) { const int c = s_child[s][k].count; if (c > SAH_LARGE) nlg++; else if (c >= 2) ns++; }
        s_base[0] = ns ? atomicAdd(&next_count[0], ns) : 0;
        s_base[1] = nlg ? atomicAdd(&next_count[1], nlg) : 0;
        for (int s = 0; s < TPB; s++) for (int k = 0; k < 2; k++) {
            const int c = s_child[s][k].count;
            if (c > SAH_LARGE) next_large[s_base[1]++] = s_child[s][k];
            else if (c >= 2) next_small[s_base[0]++] = s_child[s][k];
        }
    }
#undef SAH_BIN
}

// Builds the tree over the primitives in the order `sorted_prims` (Morton order: neighbours in the array are neighbours in
// space, so the box gathers of the first levels are local) into c->sah_compact / c->sah_csize.  Work on c->stream.
int sah_build(tirt_ctx *c, const int *sorted_prims)
{
    const int n = c->n, N = 2 * n - 1;
    hipStream_t st = c->stream;
    constexpr int MAX_LEVELS = 160;
    const size_t small_cap = (size_t)n / 2 + 2, large_cap = (size_t)n / SAH_LARGE + 2;
    if (c->sah_compact.ensure(sizeof(float) * (size_t)N * CPN_VEC) || c->sah_csize.ensure(sizeof(int) * (size_t)N) ||
        c->sah_box.ensure(sizeof(float4) * 2 * (size_t)n) || c->sah_idx.ensure(sizeof(int) * 2 * (size_t)n) ||
        c->sah_tasks.ensure(sizeof(SahTask) * 2 * (small_cap + large_cap)) || c->sah_counts.ensure(sizeof(int) * 2 * (MAX_LEVELS + 2))) return TIRT_ERR_HIP;
    SceneView sv = scene_view(c);
    float4 *box = c->sah_box.as<float4>();
    int *idx[2] = {c->sah_idx.as<int>(), c->sah_idx.as<int>() + n};
    SahTask *small[2] = {c->sah_tasks.as<SahTask>(), c->sah_tasks.as<SahTask>() + small_cap};
    SahTask *large[2] = {small[1] + small_cap, small[1] + small_cap + large_cap};
    int *counts = c->sah_counts.as<int>();                 // counts[2 * level + (0 small | 1 large)]
    hipLaunchKernelGGL(k_sah_prim_boxes, dim3((n + 255) / 256), dim3(256), 0, st, sv, box);
    TIRT_HIP(hipMemcpyAsync(idx[0], sorted_prims, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));
    TIRT_HIP(hipMemsetAsync(counts, 0, sizeof(int) * 2 * (MAX_LEVELS + 2), st));
    const SahTask root = {0, n, 0, 0};
    const int one = 1;
    const bool root_large = n > SAH_LARGE;
    TIRT_HIP(hipMemcpyAsync(root_large ? large[0] : small[0], &root, sizeof(root), hipMemcpyHostToDevice, st));
    TIRT_HIP(hipMemcpyAsync(counts + (root_large ? 1 : 0), &one, sizeof(int), hipMemcpyHostToDevice, st));
    int level = 0, host_counts[2 * (MAX_LEVELS + 2)];
    for (;;) {
        const int until = (level + 16 < MAX_LEVELS) ? level + 16 : MAX_LEVELS;
        for (; level < until; level++) {
            const int in = level & 1, out = in ^ 1, halve = level >= SAH_FORCE_HALVING_AFTER ? 1 : 0;
            long cap = 1; for (int k = 0; k < level && cap < n; k++) cap *= 2;              // a level holds at most 2^level nodes
            long cap_small = cap < (long)small_cap ? cap : (long)small_cap, cap_large = cap < (long)large_cap ? cap : (long)large_cap;
            hipLaunchKernelGGL(k_sah_level<16>, dim3((unsigned)cap_large), dim3(SAH_BLOCK), 0, st, box, idx[in], idx[out], large[in], counts + 2 * level + 1,
                               small[out], large[out], counts + 2 * (level + 1), c->sah_compact.as<float>(), c->sah_csize.as<int>(), halve);
            hipLaunchKernelGGL(k_sah_level<1>, dim3((unsigned)((cap_small + 15) / 16)), dim3(SAH_BLOCK), 0, st, box, idx[in], idx[out], small[in], counts + 2 * level,
                               small[out], large[out], counts + 2 * (level + 1), c->sah_compact.as<float>(), c->sah_csize.as<int>(), halve);
        }
        TIRT_HIP(hipMemcpyAsync(host_counts, counts, sizeof(host_counts), hipMemcpyDeviceToHost, st));
        TIRT_HIP(hipStreamSynchronize(st));
        if (host_counts[2 * level] == 0 && host_counts[2 * level + 1] == 0) break;
        TIRT_REQUIRE(level < MAX_LEVELS, "tirt_lbvh_build: traversal tree deeper than 160 levels");
    }
    c->sah_levels = level;
    TIRT_HIP(hipGetLastError());
    return TIRT_OK;
}

}  // namespace tirt
