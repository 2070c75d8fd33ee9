/* tools/exp/sah_tree.py: two-level binned SAH -- the Morton-ordered primitives are cut into clusters of K consecutive
 * ones; a weighted binned-SAH tree over the cluster boxes (weight = primitives in the cluster) forms the top, a plain one
 * over each cluster's primitives the bottom.  Same `compact` output as sah_build.c.
 * gcc -O2 -shared -fPIC -o sah2_build.so sah2_build.c */
#include <stdlib.h>
#include <string.h>
#define BINS 32
typedef struct { float mn[3], mx[3]; } box_t;
typedef struct { box_t b; int w; int first; } item_t;      /* top level: cluster (first prim position, w prims); bottom: w = 1, first = prim id */
static float *g_out; static int *g_size; static int g_next;
static const float *g_box; static const int *g_order;
static inline void grow(box_t *b, const box_t *p) { for (int k = 0; k < 3; k++) { if (p->mn[k] < b->mn[k]) b->mn[k] = p->mn[k]; if (p->mx[k] > b->mx[k]) b->mx[k] = p->mx[k]; } }
static inline float area(const box_t *b) { float dx = b->mx[0] - b->mn[0], dy = b->mx[1] - b->mn[1], dz = b->mx[2] - b->mn[2]; return dx * dy + dy * dz + dz * dx; }
static void empty(box_t *b) { for (int k = 0; k < 3; k++) { b->mn[k] = 1e30f; b->mx[k] = -1e30f; } }
static int build(item_t *it, int cnt, int top);
static int build_cluster(const item_t *cl)
{
    item_t *it = (item_t *)malloc(sizeof(item_t) * cl->w);
    for (int i = 0; i < cl->w; i++) {
        const int p = g_order[cl->first + i]; const float *q = g_box + (size_t)p * 6;
        for (int k = 0; k < 3; k++) { it[i].b.mn[k] = q[k]; it[i].b.mx[k] = q[3 + k]; }
        it[i].w = 1; it[i].first = p;
    }
    const int r = build(it, cl->w, 0);
    free(it);
    return r;
}
static int build(item_t *it, int cnt, int top)
{
    if (cnt == 1 && top) return build_cluster(&it[0]);
    const int me = g_next++;
    float *row = g_out + (size_t)me * 9;
    box_t bb; empty(&bb);
    for (int i = 0; i < cnt; i++) grow(&bb, &it[i].b);
    for (int k = 0; k < 3; k++) { row[2 + k] = bb.mn[k]; row[5 + k] = bb.mx[k]; }
    row[8] = 0.0f;
    if (cnt == 1) { row[0] = 1.0f; row[1] = (float)it[0].first; g_size[me] = 1; return me; }
    float cmn[3] = {1e30f, 1e30f, 1e30f}, cmx[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = 0; i < cnt; i++) for (int k = 0; k < 3; k++) { float c = 0.5f * (it[i].b.mn[k] + it[i].b.mx[k]); if (c < cmn[k]) cmn[k] = c; if (c > cmx[k]) cmx[k] = c; }
    int best_axis = -1, best_split = -1; float best_cost = 1e30f;
    for (int ax = 0; ax < 3; ax++) {
        const float ext = cmx[ax] - cmn[ax];
        if (!(ext > 0.0f)) continue;
        box_t bin[BINS]; int bw[BINS], bc[BINS];
        for (int b = 0; b < BINS; b++) { empty(&bin[b]); bw[b] = 0; bc[b] = 0; }
        const float scale = BINS / ext;
        for (int i = 0; i < cnt; i++) {
            int b = (int)((0.5f * (it[i].b.mn[ax] + it[i].b.mx[ax]) - cmn[ax]) * scale); if (b >= BINS) b = BINS - 1; if (b < 0) b = 0;
            grow(&bin[b], &it[i].b); bw[b] += it[i].w; bc[b]++;
        }
        for (int sp = 0; sp < BINS - 1; sp++) {
            box_t L, R; empty(&L); empty(&R); int wl = 0, wr = 0, cl = 0, cr = 0;
            for (int b = 0; b < BINS; b++) if (bc[b]) { if (b <= sp) { grow(&L, &bin[b]); wl += bw[b]; cl += bc[b]; } else { grow(&R, &bin[b]); wr += bw[b]; cr += bc[b]; } }
            if (cl == 0 || cr == 0) continue;
            const float cost = area(&L) * wl + area(&R) * wr;
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = sp; }
        }
    }
    int mid;
    if (best_axis < 0) mid = cnt / 2;
    else {
        const float ext = cmx[best_axis] - cmn[best_axis], scale = BINS / ext;
        item_t *tmp = (item_t *)malloc(sizeof(item_t) * cnt); int nl = 0, nr = 0;
        for (int i = 0; i < cnt; i++) { int b = (int)((0.5f * (it[i].b.mn[best_axis] + it[i].b.mx[best_axis]) - cmn[best_axis]) * scale); if (b >= BINS) b = BINS - 1; if (b < 0) b = 0; if (b <= best_split) nl++; }
        int l = 0; nr = nl;
        for (int i = 0; i < cnt; i++) { int b = (int)((0.5f * (it[i].b.mn[best_axis] + it[i].b.mx[best_axis]) - cmn[best_axis]) * scale); if (b >= BINS) b = BINS - 1; if (b < 0) b = 0; if (b <= best_split) tmp[l++] = it[i]; else tmp[nr++] = it[i]; }
        memcpy(it, tmp, sizeof(item_t) * cnt); free(tmp);
        mid = nl;
    }
    row[0] = 0.0f;
    build(it, mid, top);
    row[1] = (float)build(it + mid, cnt - mid, top);
    g_size[me] = g_next - me;
    return me;
}
int sah2_build(const float *boxes, const int *order, int n, int K, float *compact_out, int *csize_out)
{
    g_box = boxes; g_order = order; g_out = compact_out; g_size = csize_out; g_next = 0;
    const int nc = (n + K - 1) / K;
    item_t *cl = (item_t *)malloc(sizeof(item_t) * nc);
    for (int c = 0; c < nc; c++) {
        const int first = c * K, w = (first + K <= n) ? K : n - first;
        empty(&cl[c].b);
        for (int i = 0; i < w; i++) { const float *q = boxes + (size_t)order[first + i] * 6; box_t t; for (int k = 0; k < 3; k++) { t.mn[k] = q[k]; t.mx[k] = q[3 + k]; } grow(&cl[c].b, &t); }
        cl[c].w = w; cl[c].first = first;
    }
    build(cl, nc, 1);
    free(cl);
    return g_next;
}
