/* tools/exp/sah_tree.py: binned-SAH binary BVH (one primitive per leaf) over primitive boxes, written in the `compact`
 * layout of the reference's flattened LBVH (pre-order; row = flags, prim | right child, min[3], max[3], unused).
 * gcc -O2 -shared -fPIC -o sah_build.so sah_build.c */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifndef BINS
#define BINS 32
#endif
typedef struct { float mn[3], mx[3]; } box_t;
static const float *g_box;      /* [n][6] */
static float *g_out; static int *g_size; static int g_next;
static inline void grow(box_t *b, const float *p) { for (int k = 0; k < 3; k++) { if (p[k] < b->mn[k]) b->mn[k] = p[k]; if (p[3 + k] > b->mx[k]) b->mx[k] = p[3 + k]; } }
static inline float area(const box_t *b) { float dx = b->mx[0] - b->mn[0], dy = b->mx[1] - b->mn[1], dz = b->mx[2] - b->mn[2]; return dx * dy + dy * dz + dz * dx; }
static void empty(box_t *b) { for (int k = 0; k < 3; k++) { b->mn[k] = 1e30f; b->mx[k] = -1e30f; } }
static int build(int *idx, int cnt)
{
    const int me = g_next++;
    float *row = g_out + (size_t)me * 9;
    box_t bb; empty(&bb);
    for (int i = 0; i < cnt; i++) grow(&bb, g_box + (size_t)idx[i] * 6);
    for (int k = 0; k < 3; k++) { row[2 + k] = bb.mn[k]; row[5 + k] = bb.mx[k]; }
    row[8] = 0.0f;
    if (cnt == 1) { row[0] = 1.0f; row[1] = (float)idx[0]; g_size[me] = 1; return me; }
    /* centroid bounds */
    float cmn[3] = {1e30f, 1e30f, 1e30f}, cmx[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = 0; i < cnt; i++) { const float *p = g_box + (size_t)idx[i] * 6; for (int k = 0; k < 3; k++) { float c = 0.5f * (p[k] + p[3 + k]); if (c < cmn[k]) cmn[k] = c; if (c > cmx[k]) cmx[k] = c; } }
    int best_axis = -1, best_split = -1; float best_cost = 1e30f;
    for (int ax = 0; ax < 3; ax++) {
        const float ext = cmx[ax] - cmn[ax];
        if (!(ext > 0.0f)) continue;
        box_t bin[BINS]; int bc[BINS];
        for (int b = 0; b < BINS; b++) { empty(&bin[b]); bc[b] = 0; }
        const float scale = BINS / ext;
        for (int i = 0; i < cnt; i++) {
            const float *p = g_box + (size_t)idx[i] * 6;
            int b = (int)((0.5f * (p[ax] + p[3 + ax]) - cmn[ax]) * scale); if (b >= BINS) b = BINS - 1; if (b < 0) b = 0;
            grow(&bin[b], p); bc[b]++;
        }
        float ra[BINS]; int rc[BINS]; box_t acc; empty(&acc); int c = 0;
        for (int b = BINS - 1; b > 0; b--) { if (bc[b]) { box_t t = bin[b]; float q[6] = {t.mn[0], t.mn[1], t.mn[2], t.mx[0], t.mx[1], t.mx[2]}; grow(&acc, q); } c += bc[b]; ra[b] = c ? area(&acc) : 0.0f; rc[b] = c; }
        empty(&acc); c = 0;
        for (int b = 0; b < BINS - 1; b++) {
            if (bc[b]) { box_t t = bin[b]; float q[6] = {t.mn[0], t.mn[1], t.mn[2], t.mx[0], t.mx[1], t.mx[2]}; grow(&acc, q); }
            c += bc[b];
            if (c == 0 || rc[b + 1] == 0) continue;
#ifdef COST_NODES
            const float cost = area(&acc) * (2 * c - 1) + ra[b + 1] * (2 * rc[b + 1] - 1);      /* nodes of the two sub-trees instead of their leaves */
#elif defined(COST_LOG)
            const float cost = area(&acc) * c * (1.0f + 0.25f * log2f((float)c)) + ra[b + 1] * rc[b + 1] * (1.0f + 0.25f * log2f((float)rc[b + 1]));
#else
            const float cost = area(&acc) * c + ra[b + 1] * rc[b + 1];
#endif
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = b; }
        }
    }
    int mid;
    if (best_axis < 0) mid = cnt / 2;         /* all centroids equal: split the list */
    else {
        const float ext = cmx[best_axis] - cmn[best_axis], scale = BINS / ext;
        int i = 0, j = cnt - 1;
        while (i <= j) {
            const float *p = g_box + (size_t)idx[i] * 6;
            int b = (int)((0.5f * (p[best_axis] + p[3 + best_axis]) - cmn[best_axis]) * scale); if (b >= BINS) b = BINS - 1; if (b < 0) b = 0;
            if (b <= best_split) i++; else { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; j--; }
        }
        mid = i;
        if (mid == 0 || mid == cnt) mid = cnt / 2;
    }
    row[0] = 0.0f;
    build(idx, mid);                          /* left child = me + 1 */
    const int right = build(idx + mid, cnt - mid);
    row[1] = (float)right;
    g_size[me] = g_next - me;
    return me;
}
int sah_build(const float *boxes, int n, float *compact_out, int *csize_out)
{
    int *idx = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) idx[i] = i;
    g_box = boxes; g_out = compact_out; g_size = csize_out; g_next = 0;
    build(idx, n);
    free(idx);
    return g_next;
}
