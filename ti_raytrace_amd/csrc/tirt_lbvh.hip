// tirt_lbvh.hip -- LBVH build on the device (replaces accel/LBvh.py:192-226 and its kernels).
//
//   k_morton      build_morton_3d                          accel/LBvh.py:318-336
//   k_rs_*        stable LSD radix sort, 4 x 8-bit digits   (reference: 30 x 1-bit passes with
//                 Blelloch scans, accel/LBvh.py:55-72,339-386 -- any stable sort by the 30-bit
//                 code produces the same permutation)
//   k_karras      build_lbvh: determineRange/findSplit      accel/LBvh.py:229-314,389-450
//   k_refit       gen_aabb + host loop, as one bottom-up pass with per-node arrival counters
//                 (min/max are exact and order-free, so boxes equal the iterative result)
//                                                           accel/LBvh.py:453-467,206-218
//   k_flatten     flatten_tree/build_compact_node (host recursion in the reference): every
//                 node finds its DFS pre-order slot from subtree sizes   accel/LBvh.py:138-173
//   k_wnodes/k_tris  GPU-only traversal layout derived from compact_node (see tirt_internal.h)
//
// All integer work; the only fp32 arithmetic is the centroid/normalisation in k_morton and
// the min/max of the boxes -- compiled with -ffp-contract=off, bit-identical to the oracle.
#include "tirt_internal.h"
#include <hip/hip_fp16.h>

namespace tirt {

// ---------------------------------------------------------------------------------------------
// Morton codes
// ---------------------------------------------------------------------------------------------
TD int expand_bits(int x)          // UtilsFunc.py:538-552
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
TD int morton3d(float x, float y, float z)      // UtilsFunc.py:568-580
{
    x = minf(maxf(x * 1024.0f, 0.0f), 1023.0f);
    y = minf(maxf(y * 1024.0f, 0.0f), 1023.0f);
    z = minf(maxf(z * 1024.0f, 0.0f), 1023.0f);
    return expand_bits((int)x) | (expand_bits((int)y) << 1) | (expand_bits((int)z) << 2);
}

__global__ void k_morton(SceneView s, float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
                         int2 *pairs, int *keys, int *vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    const int *pr = s.primitive + (size_t)i * PRI_VEC;
    int code;
    if (pr[0] == PRIMITIVE_TRI) {
        const float third = (float)(1.0 / 3.0);
        v3 v0 = vtx_pos(s, pr[1]), v1 = vtx_pos(s, pr[1] + 1), v2 = vtx_pos(s, pr[1] + 2);
        v3 c = ((v1 + v2) + v0) * third;
        v3 num = c - V(bminx, bminy, bminz);
        v3 den = V(bmaxx, bmaxy, bmaxz) - V(bminx, bminy, bminz);
        code = morton3d(num.x / den.x, num.y / den.y, num.z / den.z);
    } else {
        const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;     // quirk B7: (type, pos.x, pos.y), un-normalised
        code = morton3d(sh[0], sh[1], sh[2]);
    }
    pairs[i] = make_int2(code, i);
    keys[i] = code; vals[i] = i;
}

// ---------------------------------------------------------------------------------------------
// Stable LSD radix sort, 8-bit digits.  Tile = 256 threads x 8 keys, processed as 8 chunks of
// 256 consecutive keys so that loads/stores stay coalesced and ranks stay stable:
// rank(key) = #equal digits in earlier tiles (scanned histogram) + #in earlier chunks of this
// tile (running[]) + #in earlier waves of this chunk (wcount[]) + #in lower lanes of this wave
// (ballot match + mbcnt-style popcount).
// ---------------------------------------------------------------------------------------------
constexpr int RS_BLOCK = 256, RS_ITEMS = 8, RS_TILE = RS_BLOCK * RS_ITEMS, RS_WAVES = RS_BLOCK / 64;

__global__ __launch_bounds__(RS_BLOCK) void k_rs_hist(const int *keys, int n, int shift, int *hist, int nblocks)
{
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    int base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int c = 0; c < RS_ITEMS; c++) {
        int i = base + c * RS_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255], 1);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];      // digit-major
}

// exclusive scan of the digit-major histogram, in two levels: block d scans row d (the counts of digit d in all tiles) in
// place and leaves the row total in row_total[d]; the 256 row totals are scanned by every scatter block itself.
// (The first version scanned all 256 x tiles entries in ONE block: 2.1 ms of the build at 4 M primitives.)
__global__ __launch_bounds__(256) void k_rs_scan_rows(int *hist, int nblocks, int *row_total)
{
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *row = hist + (size_t)blockIdx.x * nblocks;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 256) {
        const int i = base + tid;
        const int v = (i < nblocks) ? row[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const int carry = carry_s;
        if (i < nblocks) row[i] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 255) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) row_total[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(RS_BLOCK) void k_rs_scatter(const int *keys_in, const int *vals_in, int *keys_out, int *vals_out,
                                                       int n, int shift, const int *hist_scanned, int nblocks, const int *row_total)
{
    __shared__ int wcount[RS_WAVES][256];
    __shared__ int running[256];
    __shared__ int rsum[RS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    {   // exclusive scan of the 256 digit totals: where digit `tid` starts in the output
        const int v = row_total[tid];
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) rsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += rsum[w];
        running[tid] = hist_scanned[(size_t)tid * nblocks + blockIdx.x] + woff + incl - v;
    }
    const int base = blockIdx.x * RS_TILE;
    for (int c = 0; c < RS_ITEMS; c++) {
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) wcount[w][tid] = 0;
        __syncthreads();
        int i = base + c * RS_BLOCK + tid;
        bool valid = i < n;
        int key = valid ? keys_in[i] : 0;
        int val = valid ? vals_in[i] : 0;
        int d = (key >> shift) & 255;
        unsigned long long mask = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            int bit = (d >> b) & 1;
            unsigned long long bal = __ballot(valid && bit);
            mask &= bit ? bal : ~bal;
        }
        int rank_w = __popcll(mask & lt_mask);
        if (valid && rank_w == 0) wcount[wave][d] = __popcll(mask);
        __syncthreads();
        if (valid) {
            int off = running[d];
            for (int w = 0; w < wave; w++) off += wcount[w][d];
            int dst = off + rank_w;
            keys_out[dst] = key; vals_out[dst] = val;
        }
        __syncthreads();
        int add = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) add += wcount[w][tid];
        running[tid] += add;
        __syncthreads();
    }
}

__global__ void k_pack_sorted(const int *keys, const int *vals, int2 *pairs, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pairs[i] = make_int2(keys[i], vals[i]);
}

// ---------------------------------------------------------------------------------------------
// Karras topology with the reference's duplicate-code rule
// ---------------------------------------------------------------------------------------------
TD int common_upper_bits(int a, int b) { int x = a ^ b; return x == 0 ? 32 : __builtin_clz((unsigned)x); }   // UtilsFunc.py:555-566
TD int delta_at(const int *codes, int n, int self_code, int j)
{ return (0 <= j && j < n) ? common_upper_bits(self_code, codes[j]) : -1; }

TD void determine_range(const int *codes, int n, int idx, int &lo, int &hi)     // accel/LBvh.py:229-294
{
    lo = 0; hi = n - 1;
    if (idx == 0) return;
    int self_code = codes[idx], l_code = codes[idx - 1], r_code = codes[idx + 1];
    if (l_code == self_code && r_code == self_code) {
        lo = idx;
        while (idx < n - 1) {
            idx += 1;
            if (idx >= n - 1) break;
            if (codes[idx] != codes[idx + 1]) break;
        }
        hi = idx;
    } else {
        int L_delta = common_upper_bits(self_code, l_code), R_delta = common_upper_bits(self_code, r_code);
        int d = (R_delta > L_delta) ? 1 : -1;
        int delta_min = L_delta < R_delta ? L_delta : R_delta;
        int l_max = 2;
        int delta = delta_at(codes, n, self_code, idx + d * l_max);
        while (delta > delta_min) {
            l_max <<= 1;
            delta = delta_at(codes, n, self_code, idx + d * l_max);
        }
        int l = 0;
        for (int t = l_max >> 1; t > 0; t >>= 1) {
            delta = delta_at(codes, n, self_code, idx + (l + t) * d);
            if (delta > delta_min) l += t;
        }
        lo = idx; hi = idx + l * d;
        if (d < 0) { int tmp = lo; lo = hi; hi = tmp; }
    }
}
TD int find_split(const int *codes, int first, int last)       // accel/LBvh.py:296-314
{
    int first_code = codes[first], last_code = codes[last];
    int split = first;
    if (first_code != last_code) {
        int delta_node = common_upper_bits(first_code, last_code);
        int stride = last - first;
        for (;;) {
            stride = (stride + 1) >> 1;
            int middle = split + stride;
            if (middle < last) {
                int delta = common_upper_bits(first_code, codes[middle]);
                if (delta > delta_node) split = middle;
            }
            if (stride <= 1) break;
        }
    }
    return split;
}

// One thread per bvh node (accel/LBvh.py:389-450).  Thread i writes every word of row i
// except `parent`, which is written by the parent's thread (root: -1), so that the
// reference's separate init offload is not needed.
__global__ void k_karras(SceneView s, const int *codes, const int *prims, float *bvh_node, int *parent, int *flag, int *subtree)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = s.n, N = 2 * n - 1;
    if (i >= N) return;
    float *nd = bvh_node + (size_t)i * NOD_VEC;
    if (i == 0) { nd[3] = -1.0f; parent[0] = -1; }
    flag[i] = 0;
    if (i >= n - 1) {
        nd[0] = 7.0f;            // float(int(float(int(-1) & (0xfffe|1))) & (0x0007|1)), UtilsFunc.py:232-243
        nd[1] = -1.0f; nd[2] = -1.0f;
        int prim = prims[i - (n - 1)];
        nd[4] = (float)prim;
        const int *pr = s.primitive + (size_t)prim * PRI_VEC;
        v3 mn = V(0.0f, 0.0f, 0.0f), mx = mn;
        if (pr[0] == PRIMITIVE_TRI) {
            v3 v1 = vtx_pos(s, pr[1]), v2 = vtx_pos(s, pr[1] + 1), v3_ = vtx_pos(s, pr[1] + 2);
            mn = v1; mx = v1;
            mn.x = minf(minf(mn.x, v2.x), v3_.x); mx.x = maxf(maxf(mx.x, v2.x), v3_.x);
            mn.y = minf(minf(mn.y, v2.y), v3_.y); mx.y = maxf(maxf(mx.y, v2.y), v3_.y);
            mn.z = minf(minf(mn.z, v2.z), v3_.z); mx.z = maxf(maxf(mx.z, v2.z), v3_.z);
        } else {
            const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;
            if ((int)sh[0] == SHAPE_SPHERE) {
                float r = sh[4];
                mn = V(sh[1] + -r, sh[2] + -r, sh[3] + -r);
                mx = V(sh[1] + r, sh[2] + r, sh[3] + r);
            }
        }
        nd[5] = mn.x; nd[6] = mn.y; nd[7] = mn.z; nd[8] = mx.x; nd[9] = mx.y; nd[10] = mx.z;
        subtree[i] = 1;
    } else {
        nd[0] = 65534.0f;        // float(int(-1) & 0xfffe)
        nd[4] = -1.0f;
        nd[5] = nd[6] = nd[7] = INF_VALUE;
        nd[8] = nd[9] = nd[10] = -INF_VALUE;
        int lo, hi;
        determine_range(codes, n, i, lo, hi);
        int split = find_split(codes, lo, hi);
        int left = split, right = split + 1;
        if ((lo < hi ? lo : hi) == split) left += n - 1;
        if ((lo > hi ? lo : hi) == split + 1) right += n - 1;
        nd[1] = (float)left; nd[2] = (float)right;
        bvh_node[(size_t)left * NOD_VEC + 3] = (float)i;  parent[left] = i;
        bvh_node[(size_t)right * NOD_VEC + 3] = (float)i; parent[right] = i;
        subtree[i] = 0;
    }
}

// Bottom-up refit: one thread per leaf climbs; the second thread to arrive at a node owns it.
// Hand-off without cache-wide fences (a release fence per climbing step writes back the whole L2:
// 15 ms of the 19.7 ms build at 4 M primitives): every word another thread will read is written
// with an agent-scope (sc1, write-through) store, the stores are drained with `s_waitcnt vmcnt(0)`
// before the arrival counter is bumped, and the owner reads its children's rows with agent-scope
// (sc1, L1-bypassing) loads -- the guide's "drained sc1 payload -> sc1 flag" recipe
// (MI355X_MICROARCH.md, row handoff-flag).  Leaf rows come from the previous kernel.
constexpr int REFIT_SLOTS = 32;                  // progress counters, one 128-byte line each
__global__ void k_refit(int n, float *bvh_node, const int *parent, int *flag, int *subtree, int *done)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cur = parent[(n - 1) + i];
    int my_done = 0;                                            // nodes this thread finished
    while (cur >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's sc1 stores of the node below have landed
        int old = __hip_atomic_fetch_add(&flag[cur], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // compiler barrier: the (relaxed) child loads below must not be hoisted above the arrival counter; in hardware
        // they are issued after the atomic has returned (the branch depends on its value) and bypass L1 (sc1)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (old == 0) break;
        float *nd = bvh_node + (size_t)cur * NOD_VEC;
        int l = (int)nd[1], r = (int)nd[2];                     // written by k_karras (previous launch)
        const float *ln = bvh_node + (size_t)l * NOD_VEC, *rn = bvh_node + (size_t)r * NOD_VEC;
        float lb[6], rb[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            lb[k] = __hip_atomic_load(&ln[5 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rb[k] = __hip_atomic_load(&rn[5 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int ls = __hip_atomic_load(&subtree[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int rs = __hip_atomic_load(&subtree[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            __hip_atomic_store(&nd[5 + k], minf(lb[k], rb[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nd[8 + k], maxf(lb[3 + k], rb[3 + k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(&subtree[cur], ls + rs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        my_done++;
        cur = parent[cur];
    }
    // progress count for the host's sanity check: once per thread after the climb, spread over REFIT_SLOTS words
    // (one atomicAdd per finished node on a single word serialised the whole kernel at ~11 ns per wave and step:
    // 6.3 of the 11.5 ms build at 4 M primitives)
    if (my_done) atomicAdd(&done[(blockIdx.x & (REFIT_SLOTS - 1)) * 32], my_done);
}

// DFS pre-order slot of node i = depth(i) + sum over ancestors entered through their right
// child of the left sibling's subtree size (accel/LBvh.py:138-161: left first, right's slot
// stored in the parent's word 1, left implicit at slot+1).
__global__ void k_flatten(int n, const float *bvh_node, const int *parent, const int *subtree, float *compact, int *leaf_compact,
                          int *cparent, int *csize)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int N = 2 * n - 1;
    if (i >= N) return;
    int off = 0, cur = i, depth = 0, first_step = 0;
    while (true) {
        int p = parent[cur];
        if (p < 0) break;
        const float *pn = bvh_node + (size_t)p * NOD_VEC;
        int pl = (int)pn[1];
        const int step = 1 + ((cur != pl) ? subtree[pl] : 0);
        if (depth == 0) first_step = step;
        off += step; depth += 1;
        cur = p;
    }
    cparent[off] = (depth == 0) ? -1 : off - first_step;     // the parent's pre-order slot
    csize[off] = subtree[i];                                   // nodes of the subtree (2 x leaves - 1)
    const float *nd = bvh_node + (size_t)i * NOD_VEC;
    float *cn = compact + (size_t)off * CPN_VEC;
    cn[0] = nd[0];
    if ((((int)nd[0]) & 1) == 1) {
        cn[1] = nd[4];
        leaf_compact[(int)nd[4]] = off;
    } else {
        int l = (int)nd[1];
        cn[1] = (float)(off + 1 + subtree[l]);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) cn[2 + k] = nd[5 + k];
    cn[8] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Traversal layout
// ---------------------------------------------------------------------------------------------
// leaf code: ~(index of the primitive's record in `tri` | shape << 30)
TD int child_code(SceneView s, const int *prim_slot, const float *cn, int idx)
{
    if ((((int)cn[0]) & 1) == 1) {
        int prim = (int)cn[1];
        int is_shape = (s.primitive[(size_t)prim * PRI_VEC] == PRIMITIVE_TRI) ? 0 : 1;
        return ~(prim_slot[prim] | (is_shape << 30));
    }
    return idx;
}
// LBVH as the traversal tree: records in Morton order
__global__ void k_slot_sorted(int n, const int *sorted_prims, int *prim_slot)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) prim_slot[sorted_prims[i]] = i;
}
__global__ void k_wnodes(SceneView s, const int *prim_slot, int N, const float *compact, float4 *wnode, float pad)
{
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= N) return;
    const float *cn = compact + (size_t)o * CPN_VEC;
    if ((((int)cn[0]) & 1) == 1) return;
    int li = o + 1, ri = (int)cn[1];
    const float *lc = compact + (size_t)li * CPN_VEC, *rc = compact + (size_t)ri * CPN_VEC;
    int cl = child_code(s, prim_slot, lc, li), cr = child_code(s, prim_slot, rc, ri);
    float lp = cl < 0 ? pad : 0.0f, rp = cr < 0 ? pad : 0.0f;
    float4 *w = wnode + (size_t)o * 4;
    w[0] = make_float4(lc[2] - lp, lc[3] - lp, lc[4] - lp, lc[5] + lp);
    w[1] = make_float4(lc[6] + lp, lc[7] + lp, rc[2] - rp, rc[3] - rp);
    w[2] = make_float4(rc[4] - rp, rc[5] + rp, rc[6] + rp, rc[7] + rp);
    w[3] = make_float4(__int_as_float(cl), __int_as_float(cr), 0.0f, 0.0f);
}
// One 48-byte record per primitive, stored in the leaf order of the traversal tree (prim_slot): rays that walk one part of the
// tree read neighbouring records.  The primitive id rides in the last word.
__global__ void k_tris(SceneView s, const int *leaf_compact, const int *prim_slot, float4 *tri)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    const int *pr = s.primitive + (size_t)i * PRI_VEC;
    float4 *t = tri + (size_t)prim_slot[i] * TRI_STRIDE;
    const float id = __int_as_float(i);
    float lc = __int_as_float(leaf_compact[i]);
    if (pr[0] == PRIMITIVE_TRI) {
        v3 v0 = vtx_pos(s, pr[1]), v1 = vtx_pos(s, pr[1] + 1), v2 = vtx_pos(s, pr[1] + 2);
        t[0] = make_float4(v0.x, v0.y, v0.z, lc);
        t[1] = make_float4(v1.x, v1.y, v1.z, 0.0f);
        t[2] = make_float4(v2.x, v2.y, v2.z, id);
    } else {
        const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;
        t[0] = make_float4(sh[1], sh[2], sh[3], lc);
        t[1] = make_float4(sh[4], sh[0], 0.0f, 0.0f);
        t[2] = make_float4(0.0f, 0.0f, 0.0f, id);
    }
}

// ---------------------------------------------------------------------------------------------
// 4-wide traversal nodes for the ordered (product) traversal (tirt_internal.h, BvhView::cnode).
// The binary LBVH is collapsed top-down, one launch per level of the wide tree: a wide node starts with the two children
// of its binary root and keeps replacing the internal candidate of LARGEST surface area by that node's two children
// until it holds four (or only leaves are left) -- the usual surface-area collapse: 3.9 children per node instead of the
// 3.0 of a fixed "every other level" collapse, i.e. a quarter fewer, fuller nodes and ~20 % fewer visits per ray.  Any
// grouping of the reference's boxes is as good as another for the RESULT: a skipped binary node's box contains its
// children's, `slabs` is monotone in the plane positions, and a hit is only accepted after the reference's own visiting
// condition has been re-established (k_trace).  Nodes are numbered in breadth-first order (a level's nodes take the next
// free indices), so the first TR_TOP_SLOTS records ARE the top of the tree that k_trace keeps in LDS.
// Planes: fp16 cells around the centre of the root box, min planes rounded down and max planes up, one more cell outward
// against the rounding of the mapping itself; leaf slots are padded by `pad` before the mapping (the reference never
// box-tests a leaf: the padding keeps "box missed but Moller-Trumbore hit" impossible).
// Analytic shapes (the sphere light) keep the whole grid as their slot box: the reference's sphere test (Scene.py:565-596:
// a square root of a difference of squares of the distance to the centre) answers "hit" for rays that pass the sphere at
// a distance that grows with the distance of the origin -- no fixed padding of the sphere's box covers that.
// ---------------------------------------------------------------------------------------------
struct GridMap { float g0[3], inv_cell[3]; };
// fp16 bit pattern of the largest half <= x - 1 / the smallest half >= x + 1 (x in grid cells, |x| <= TR_GRID_HALF + a few)
TD unsigned grid_lo(float x, float g0, float inv_cell)
{ const float v = (x - g0) * inv_cell - 1.0f; return (unsigned)__half_as_ushort(__float2half_rd(v < -60000.0f ? -60000.0f : (v > 60000.0f ? 60000.0f : v))); }
TD unsigned grid_hi(float x, float g0, float inv_cell)
{ const float v = (x - g0) * inv_cell + 1.0f; return (unsigned)__half_as_ushort(__float2half_ru(v < -60000.0f ? -60000.0f : (v > 60000.0f ? 60000.0f : v))); }
TD bool cn_leaf(const float *compact, int i) { return (((int)compact[(size_t)i * CPN_VEC]) & 1) == 1; }
TD float cn_area(const float *compact, int i)
{
    const float *c = compact + (size_t)i * CPN_VEC;
    const float dx = c[5] - c[2], dy = c[6] - c[3], dz = c[7] - c[4];
    return dx * dy + dy * dz + dz * dx;
}
#ifdef TIRT_EXPERIMENTS      // (option "wide_collapse": measured 1-9 % fewer visits, the same rays per second -- round 3; not in the product library)
// Cost-optimal grouping of the binary tree into wide nodes (the dynamic programme of Ylitie, Karras, Laine 2017 for a
// visit-count cost): every binary inner node either becomes the root of a wide node (cost: its surface area, the chance of a
// visit) or is dissolved into the wide node above it (cost 0), under the constraint of four children per wide node.
//   T(n, k) = cheapest way to hang the subtree of n below a wide node using at most k of its child slots (k = 1..3)
//   T(n, 1) = area(n) + min_i [T(l, i) + T(r, 4 - i)]                       n becomes a wide node: slots 4 = i + (4 - i)
//   T(n, k) = min(T(n, 1), min_i [T(l, i) + T(r, k - i)])                   or n is dissolved: its children share the k slots
// with T(leaf, k) = 0.  Bottom-up, one thread per leaf climbing, the second arrival owns the node -- the hand-off of k_refit.
// dec[n]: bits 0-1 = i of the root split; bit 2 = choice for k = 2 (0 node, 1 dissolve 1+1); bits 3-4 = choice for k = 3
// (0 node, 1: 1+1, 2: 1+2, 3: 2+1).
__global__ void k_wide_dp(int N, const float *compact, const int *parent, int *flag, float *dp_t, int *dp_dec)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || !cn_leaf(compact, i)) return;
    int cur = parent[i];
    while (cur >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(&flag[cur], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (old == 0) break;
        const int l = cur + 1, r = (int)compact[(size_t)cur * CPN_VEC + 1];
        float tl[4] = {0, 0, 0, 0}, tr[4] = {0, 0, 0, 0};
        if (!cn_leaf(compact, l)) for (int k = 1; k <= 3; k++) tl[k] = __hip_atomic_load(&dp_t[(size_t)l * 3 + k - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!cn_leaf(compact, r)) for (int k = 1; k <= 3; k++) tr[k] = __hip_atomic_load(&dp_t[(size_t)r * 3 + k - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int root_i = 2; float best = tl[2] + tr[2];
        if (tl[1] + tr[3] < best) { best = tl[1] + tr[3]; root_i = 1; }
        if (tl[3] + tr[1] < best) { best = tl[3] + tr[1]; root_i = 3; }
        const float t1 = cn_area(compact, cur) + best;
        float t2 = t1; int d2 = 0;
        if (tl[1] + tr[1] < t2) { t2 = tl[1] + tr[1]; d2 = 1; }
        float t3 = t1; int d3 = 0;
        if (tl[1] + tr[1] < t3) { t3 = tl[1] + tr[1]; d3 = 1; }
        if (tl[1] + tr[2] < t3) { t3 = tl[1] + tr[2]; d3 = 2; }
        if (tl[2] + tr[1] < t3) { t3 = tl[2] + tr[1]; d3 = 3; }
        __hip_atomic_store(&dp_t[(size_t)cur * 3 + 0], t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&dp_t[(size_t)cur * 3 + 1], t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&dp_t[(size_t)cur * 3 + 2], t3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dp_dec[cur] = root_i | (d2 << 2) | (d3 << 3);                    // read by the next kernel only
        cur = parent[cur];
    }
}
#endif

// level_off[L] / level_cnt[L]: first wide index and number of wide nodes of level L; queue holds the binary roots (compact
// indices) of all wide nodes in breadth-first order (queue[w] for wide node w)
__global__ void k_wide_level(SceneView s, const int *prim_slot, const float *compact, const int *csize, const int *dp_dec, int level, int *level_off, int *level_cnt, int *queue, uint4 *cnode, float pad, GridMap gm, int shapes_boxed)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const int n_in = level_cnt[level], off = level_off[level];
    if (t == 0) level_off[level + 1] = off + n_in;           // this level's children are numbered from here on
    const bool live = t < n_in;
    const int w = off + t;
    int cand[4] = {-1, -1, -1, -1}, nc = 0, n_int = 0;
    if (live) {
        const int root = queue[w];
        const float *rc = compact + (size_t)root * CPN_VEC;
        if (dp_dec) {                    // the dynamic programme's choices (k_wide_dp), unfolded from this node's root split
            int st_n[6], st_k[6], sp = 0;
            const int ri = dp_dec[root] & 3;
            st_n[sp] = (int)rc[1]; st_k[sp++] = 4 - ri;
            st_n[sp] = root + 1; st_k[sp++] = ri;
            while (sp > 0) {
                const int c = st_n[--sp], k = st_k[sp];
                int choice = 0;                                    // 0: c itself takes one slot
                if (!cn_leaf(compact, c) && k >= 2) { const int d = dp_dec[c]; choice = (k == 2) ? ((d >> 2) & 1) : ((d >> 3) & 3); }
                if (choice == 0) { cand[nc++] = c; continue; }
                const int li = (choice == 3) ? 2 : 1, rk = (choice == 1) ? 1 : (choice == 2 ? 2 : 1);
                st_n[sp] = (int)compact[(size_t)c * CPN_VEC + 1]; st_k[sp++] = rk;
                st_n[sp] = c + 1; st_k[sp++] = li;
            }
        } else {
        cand[0] = root + 1; cand[1] = (int)rc[1]; nc = 2;
        for (;;) {
            if (nc == 4) break;
            // first choice: an internal candidate small enough to be taken apart completely in the free slots (it then needs no
            // node of its own -- otherwise the bottom of the tree is full of 2- and 3-leaf nodes); else the largest surface area
            int best = -1, best_leaves = 1 << 30; float best_area = -1.0f;
            const int free_slots = 4 - nc;
#ifndef WIDE_PURE_AREA
            for (int k = 0; k < nc; k++) if (!cn_leaf(compact, cand[k])) {
                const int leaves = (csize[cand[k]] + 1) >> 1;
                if (leaves - 1 <= free_slots && leaves < best_leaves) { best_leaves = leaves; best = k; }
            }
#endif
            if (best < 0)
                for (int k = 0; k < nc; k++) if (!cn_leaf(compact, cand[k])) { const float a = cn_area(compact, cand[k]); if (a > best_area) { best_area = a; best = k; } }
            if (best < 0) break;
            const int b = cand[best];
            cand[best] = b + 1; cand[nc++] = (int)compact[(size_t)b * CPN_VEC + 1];
        }
        }
        for (int c = 0; c < nc; c++) if (!cn_leaf(compact, cand[c])) n_int++;
    }
    // queue positions of this wave's internal children: one atomic per wave (one per child serialised the level: same-address
    // device atomics retire at ~11 ns each)
    int incl = n_int;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    const int total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane == 63 && total) base = atomicAdd(&level_cnt[level + 1], total);
    base = __shfl(base, 63, 64);
    if (!live) return;
    int pos = base + incl - n_int;
    unsigned wd[16];
    const int next_off = off + n_in;
    for (int c = 0; c < 4; c++) {
        int code = TR_EMPTY;
        if (c < nc) {
            const float *cn = compact + (size_t)cand[c] * CPN_VEC;
            if ((((int)cn[0]) & 1) == 1) code = child_code(s, prim_slot, cn, cand[c]);
            else { queue[next_off + pos] = cand[c]; code = next_off + pos; pos++; }
            const bool leaf = code < 0;
            const bool shape = !shapes_boxed && leaf && (((~code) >> 30) & 1) != 0;      // (shapes_boxed: the tree's rows hold the spheres' padded boxes, tirt_sah.hip)
            const float p = leaf ? pad : 0.0f;
            for (int a = 0; a < 3; a++)
                wd[3 * c + a] = shape ? (TR_H_NEG | (TR_H_POS << 16))
                                      : (grid_lo(cn[2 + a] - p, gm.g0[a], gm.inv_cell[a]) | (grid_hi(cn[5 + a] + p, gm.g0[a], gm.inv_cell[a]) << 16));
        } else {
            for (int a = 0; a < 3; a++) wd[3 * c + a] = TR_H_POS | (TR_H_NEG << 16);       // inverted box: never hit
        }
        wd[12 + c] = (unsigned)code;
    }
    uint4 *dst = cnode + (size_t)w * 4;
    for (int k = 0; k < 4; k++) dst[k] = make_uint4(wd[4 * k], wd[4 * k + 1], wd[4 * k + 2], wd[4 * k + 3]);
}

// The level loop of the wide build over a binary tree in `compact` layout (pre-order, left child = self + 1) with subtree
// sizes `csize`; ends with c->ev1 recorded after the last level.
static int build_wide(tirt_ctx *c, const float *compact, const int *csize, const int *parent, float pad, const GridMap &gm, int shapes_boxed = 0)
{
    const int n = c->n, N = 2 * n - 1; (void)N;
    hipStream_t st = c->stream;
    SceneView sv = scene_view(c);
    const int *dp_dec = nullptr;
#ifdef TIRT_EXPERIMENTS
    if (c->wide_dp_on && parent) {
        // scratch: arrival counters [N] | choices [N] | costs [3N]
        if (c->wide_dp.ensure(sizeof(int) * 5 * (size_t)N)) return TIRT_ERR_HIP;
        int *flag = c->wide_dp.as<int>(), *dec = flag + N; float *t = (float *)(dec + N);
        TIRT_HIP(hipMemsetAsync(flag, 0, sizeof(int) * (size_t)N, st));
        hipLaunchKernelGGL(k_wide_dp, dim3((N + 255) / 256), dim3(256), 0, st, N, compact, parent, flag, t, dec);
        dp_dec = dec;
    }
#endif
    constexpr int WIDE_LEVELS_MAX = 2048;     // runs of identical Morton codes make chains: a level per three leaves of a chain
    int *lv_off = c->wide_levels.as<int>(), *lv_cnt = lv_off + (WIDE_LEVELS_MAX + 2);
    TIRT_HIP(hipMemsetAsync(c->wide_levels.p, 0, sizeof(int) * 2 * (WIDE_LEVELS_MAX + 2), st));
    const int one = 1;
    TIRT_HIP(hipMemcpyAsync(lv_cnt, &one, sizeof(int), hipMemcpyHostToDevice, st));        // level 0: the root (compact index 0)
    TIRT_HIP(hipMemsetAsync(c->wide_queue.p, 0, sizeof(int), st));
    int level = 0, host_lv[2 * (WIDE_LEVELS_MAX + 2)];
    for (;;) {
        const int until = (level + 16 < WIDE_LEVELS_MAX) ? level + 16 : WIDE_LEVELS_MAX;
        for (; level < until; level++) {
            long cap = 1; for (int k = 0; k < level && cap < n; k++) cap *= 4;             // a level holds at most 4^level nodes
            if (cap > n) cap = n;
            hipLaunchKernelGGL(k_wide_level, dim3((unsigned)((cap + 127) / 128)), dim3(128), 0, st, sv, c->prim_slot.as<int>(), compact, csize, dp_dec, level, lv_off, lv_cnt,
                               c->wide_queue.as<int>(), c->cnode.as<uint4>(), pad, gm, shapes_boxed);
        }
        TIRT_HIP(hipEventRecord(c->ev1, st));
        TIRT_HIP(hipMemcpyAsync(host_lv, c->wide_levels.p, sizeof(host_lv), hipMemcpyDeviceToHost, st));
        TIRT_HIP(hipStreamSynchronize(st));
        if (host_lv[(WIDE_LEVELS_MAX + 2) + level] == 0) { c->wide_nodes = host_lv[level]; break; }       // the next level is empty: done
        TIRT_REQUIRE(level < WIDE_LEVELS_MAX, "tirt_lbvh_build: the 4-wide tree is deeper than 2048 levels");
    }
    return TIRT_OK;
}

#ifdef TIRT_EXPERIMENTS
// tools/exp/sah_tree.py: traverse a 4-wide tree collapsed from ANOTHER binary tree over the same primitives (hits stay the
// reference's: candidates are verified against the reference tree, k_trace)
int exp_wide_from_tree(tirt_ctx *c, const float *compact_host, const int *csize_host)
{
    const int n = c->n, N = 2 * n - 1;
    static DevBuf alt_compact, alt_csize;
    if (alt_compact.ensure(sizeof(float) * (size_t)N * CPN_VEC) || alt_csize.ensure(sizeof(int) * (size_t)N)) return TIRT_ERR_HIP;
    TIRT_HIP(hipMemcpy(alt_compact.p, compact_host, sizeof(float) * (size_t)N * CPN_VEC, hipMemcpyHostToDevice));
    TIRT_HIP(hipMemcpy(alt_csize.p, csize_host, sizeof(int) * (size_t)N, hipMemcpyHostToDevice));
    float ex = c->root_max[0] - c->root_min[0], ey = c->root_max[1] - c->root_min[1], ez = c->root_max[2] - c->root_min[2];
    const float pad = 1.0e-4f * sqrtf(ex * ex + ey * ey + ez * ez);
    GridMap gm;
    for (int k = 0; k < 3; k++) { gm.g0[k] = c->grid_min[k]; gm.inv_cell[k] = c->grid_inv_cell[k]; }
    c->build_serial++;                       // (another traversal tree: the camera rays' candidate lists belong to the old one, tirt_pvb.hip)
    return build_wide(c, alt_compact.as<float>(), alt_csize.as<int>(), nullptr, pad, gm);
}
#endif

// ---------------------------------------------------------------------------------------------
// Host driver
// ---------------------------------------------------------------------------------------------
int lbvh_build(tirt_ctx *c)
{
    TIRT_REQUIRE(c->n >= 1, "tirt_lbvh_build: no primitives uploaded");
    const int n = c->n, N = 2 * n - 1;
    hipStream_t st = c->stream, st0 = st;
    c->built = false; c->built_sah = 0;
    if (c->morton_unsorted.ensure(sizeof(int2) * (size_t)n)) return TIRT_ERR_HIP;
    if (c->morton_sorted.ensure(sizeof(int2) * (size_t)n)) return TIRT_ERR_HIP;
    if (c->keys_a.ensure(sizeof(int) * (size_t)n) || c->keys_b.ensure(sizeof(int) * (size_t)n) ||
        c->vals_a.ensure(sizeof(int) * (size_t)n) || c->vals_b.ensure(sizeof(int) * (size_t)n)) return TIRT_ERR_HIP;
    const int nblocks = (n + RS_TILE - 1) / RS_TILE;
    if (c->hist.ensure(sizeof(int) * 256 * ((size_t)nblocks + 1))) return TIRT_ERR_HIP;      // digit-major tile histograms + the 256 digit totals
    if (c->bvh_node.ensure(sizeof(float) * (size_t)N * NOD_VEC) || c->compact.ensure(sizeof(float) * (size_t)N * CPN_VEC)) return TIRT_ERR_HIP;
    if (c->parent.ensure(sizeof(int) * (size_t)N) || c->flag.ensure(sizeof(int) * (size_t)N) ||
        c->subtree.ensure(sizeof(int) * (size_t)N) || c->build_status.ensure(sizeof(int) * 32 * REFIT_SLOTS) ||
        c->leaf_compact.ensure(sizeof(int) * (size_t)n)) return TIRT_ERR_HIP;
    if (c->wnode.ensure(sizeof(float4) * 4 * (size_t)N) || c->tri.ensure(sizeof(float4) * TRI_STRIDE * (size_t)n) || c->prim_slot.ensure(sizeof(int) * (size_t)n)) return TIRT_ERR_HIP;
    // 4-wide nodes: fewer than n of them; indices are used as 32-bit byte offsets / 64
    TIRT_REQUIRE(n <= (1 << 24), "tirt_lbvh_build: more than 16 Mi primitives");
    constexpr int WIDE_LEVELS_MAX = 2048;
    if (c->cnode.ensure(sizeof(uint4) * 4 * ((size_t)n + 4)) || c->wide_queue.ensure(sizeof(int) * (size_t)n) ||
        c->wide_levels.ensure(sizeof(int) * 2 * (WIDE_LEVELS_MAX + 2)) || c->cparent.ensure(sizeof(int) * (size_t)N) ||
        c->csize.ensure(sizeof(int) * (size_t)N)) return TIRT_ERR_HIP;
    (void)st0;

    SceneView sv = scene_view(c);
    const int B = 256;
    TIRT_HIP(hipEventRecord(c->ev0, st));
    hipLaunchKernelGGL(k_morton, dim3((n + B - 1) / B), dim3(B), 0, st, sv, c->bmin[0], c->bmin[1], c->bmin[2],
                       c->bmax[0], c->bmax[1], c->bmax[2], c->morton_unsorted.as<int2>(), c->keys_a.as<int>(), c->vals_a.as<int>());
    int *ka = c->keys_a.as<int>(), *kb = c->keys_b.as<int>(), *va = c->vals_a.as<int>(), *vb = c->vals_b.as<int>();
    for (int shift = 0; shift < 30; shift += 8) {
        hipLaunchKernelGGL(k_rs_hist, dim3(nblocks), dim3(RS_BLOCK), 0, st, ka, n, shift, c->hist.as<int>(), nblocks);
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(256), dim3(256), 0, st, c->hist.as<int>(), nblocks, c->hist.as<int>() + 256 * (size_t)nblocks);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblocks), dim3(RS_BLOCK), 0, st, ka, va, kb, vb, n, shift, c->hist.as<int>(), nblocks,
                           c->hist.as<int>() + 256 * (size_t)nblocks);
        int *t = ka; ka = kb; kb = t; t = va; va = vb; vb = t;
    }
    // 4 passes: the sorted data is back in keys_a / vals_a
    hipLaunchKernelGGL(k_pack_sorted, dim3((n + B - 1) / B), dim3(B), 0, st, ka, va, c->morton_sorted.as<int2>(), n);
    TIRT_HIP(hipMemsetAsync(c->build_status.p, 0, sizeof(int) * 32 * REFIT_SLOTS, st));
    hipLaunchKernelGGL(k_karras, dim3((N + B - 1) / B), dim3(B), 0, st, sv, ka, va, c->bvh_node.as<float>(),
                       c->parent.as<int>(), c->flag.as<int>(), c->subtree.as<int>());
    hipLaunchKernelGGL(k_refit, dim3((n + B - 1) / B), dim3(B), 0, st, n, c->bvh_node.as<float>(), c->parent.as<int>(),
                       c->flag.as<int>(), c->subtree.as<int>(), c->build_status.as<int>());
    hipLaunchKernelGGL(k_flatten, dim3((N + B - 1) / B), dim3(B), 0, st, n, c->bvh_node.as<float>(), c->parent.as<int>(),
                       c->subtree.as<int>(), c->compact.as<float>(), c->leaf_compact.as<int>(), c->cparent.as<int>(), c->csize.as<int>());
    // root box + refit status back to the host (one small read; the reference does ~depth of them)
    int done = 0, done_slots[32 * REFIT_SLOTS]; float root[11];
    TIRT_HIP(hipMemcpyAsync(done_slots, c->build_status.p, sizeof(done_slots), hipMemcpyDeviceToHost, st));
    TIRT_HIP(hipMemcpyAsync(root, c->bvh_node.p, sizeof(float) * NOD_VEC, hipMemcpyDeviceToHost, st));
    TIRT_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < REFIT_SLOTS; k++) done += done_slots[k * 32];
    if (done != n - 1) {
        set_error("tirt_lbvh_build: refit reached " + std::to_string(done) + " of " + std::to_string(n - 1) + " internal nodes");
        return TIRT_ERR_BUILD;
    }
    for (int k = 0; k < 3; k++) { c->root_min[k] = root[5 + k]; c->root_max[k] = root[8 + k]; }
    float ex = root[8] - root[5], ey = root[9] - root[6], ez = root[10] - root[7];
    float diag = sqrtf(ex * ex + ey * ey + ez * ez);
    float pad = 1.0e-4f * diag;
    // grid of the quantised nodes: the (padded) root box spans cells -TR_GRID_HALF .. +TR_GRID_HALF around its centre
    GridMap gm;
    for (int k = 0; k < 3; k++) {
        const float lo = c->root_min[k] - pad, hi = c->root_max[k] + pad;
        float ext = hi - lo; if (!(ext > 0.0f)) ext = 1.0e-30f;
        const float cell = ext / (2.0f * TR_GRID_HALF);
        c->grid_cell[k] = cell; c->grid_min[k] = lo + 0.5f * ext; c->grid_inv_extent[k] = 1.0f / ext;
        gm.g0[k] = c->grid_min[k]; gm.inv_cell[k] = 1.0f / cell; c->grid_inv_cell[k] = gm.inv_cell[k];
    }
    // surface-area collapse of the binary tree into 4-wide nodes, one launch per level of the wide tree (k_wide_level)
    c->wide_nodes = 0;
    const float *tree = c->compact.as<float>(); const int *tree_size = c->csize.as<int>(), *tree_parent = c->cparent.as<int>();
    // analytic spheres get their own padded box in the traversal tree when there are few of them (far-origin rays enter through chain nodes that
    // hold them with whole-grid boxes: BvhView::far_qcode); otherwise, and on the reference's LBVH, their 4-wide slots span the whole grid as before
    const int shapes_boxed = (n >= 2 && c->use_sah && c->sphere_prims.size() <= 8) ? 1 : 0;
    c->n_far_nodes = 0;
    if (n >= 2 && c->use_sah) {          // walk a better tree than the reference's (tirt_sah.hip); the hits stay the reference's (k_trace)
        if (int rc = sah_build(c, va, shapes_boxed ? sphere_pad_abs(diag) : -1.0f)) return rc;                                  // also fills prim_slot
        tree = c->sah_compact.as<float>(); tree_size = c->sah_csize.as<int>(); tree_parent = c->sah_parent.as<int>();
        c->built_sah = 1;
    } else hipLaunchKernelGGL(k_slot_sorted, dim3((n + B - 1) / B), dim3(B), 0, st, n, va, c->prim_slot.as<int>());
    hipLaunchKernelGGL(k_wnodes, dim3((N + B - 1) / B), dim3(B), 0, st, sv, c->prim_slot.as<int>(), N, c->compact.as<float>(), c->wnode.as<float4>(), pad);
    hipLaunchKernelGGL(k_tris, dim3((n + B - 1) / B), dim3(B), 0, st, sv, c->leaf_compact.as<int>(), c->prim_slot.as<int>(), c->tri.as<float4>());
    if (n >= 2) {
        if (int rc = build_wide(c, tree, tree_size, tree_parent, pad, gm, shapes_boxed)) return rc;
    } else TIRT_HIP(hipEventRecord(c->ev1, st));
    if (n == 1) {
        int prim = 0, is_shape = 0;
        int pr0;
        TIRT_HIP(hipMemcpyAsync(&pr0, c->primitive.p, sizeof(int), hipMemcpyDeviceToHost, st));
        TIRT_HIP(hipStreamSynchronize(st));
        is_shape = (pr0 == PRIMITIVE_TRI) ? 0 : 1;
        c->root_code = ~(prim | (is_shape << 30));
    } else c->root_code = 0;
    if (shapes_boxed && !c->sphere_prims.empty()) {
        // the entry of far-origin rays: up to three chain nodes behind the wide tree, [root, sphere, sphere, sphere | next chain node],
        // every slot with the whole grid as its box (k_trace's refill; a sphere's leaf code is ~(record index | shape bit))
        int slots[8];
        const int ns = (int)c->sphere_prims.size();
        for (int k = 0; k < ns; k++)
            TIRT_HIP(hipMemcpyAsync(&slots[k], c->prim_slot.as<int>() + c->sphere_prims[k], sizeof(int), hipMemcpyDeviceToHost, st));
        TIRT_HIP(hipStreamSynchronize(st));
        int codes[12], nc = 0, nodes = 0;
        unsigned wd[3 * 16];
        codes[nc++] = 0;                                   // the root of the wide tree
        for (int k = 0; k < ns; k++) codes[nc++] = ~(slots[k] | (1 << 30));
        for (int at = 0; at < nc; nodes++) {
            unsigned *w = wd + 16 * nodes;
            const int left = nc - at, take = left <= 4 ? left : 3;      // a full node keeps its last slot for the link
            for (int s = 0; s < 4; s++) {
                const bool used = s < take || (s == 3 && left > 4);
                for (int a = 0; a < 3; a++) w[3 * s + a] = used ? (TR_H_NEG | (TR_H_POS << 16)) : (TR_H_POS | (TR_H_NEG << 16));
                w[12 + s] = s < take ? (unsigned)codes[at + s] : (used ? (unsigned)(c->wide_nodes + nodes + 1) : (unsigned)TR_EMPTY);
            }
            at += take;
        }
        c->n_far_nodes = nodes;
        TIRT_HIP(hipMemcpyAsync(c->cnode.as<uint4>() + (size_t)c->wide_nodes * 4, wd, sizeof(unsigned) * 16 * (size_t)nodes, hipMemcpyHostToDevice, st));
    }
    TIRT_HIP(hipStreamSynchronize(st));
    float ms = 0.0f;
    TIRT_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->ms_build = ms;
    TIRT_HIP(hipGetLastError());
    c->build_serial++;
    c->built = true;
    return TIRT_OK;
}

}  // namespace tirt
