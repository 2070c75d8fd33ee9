/* tools/exp/sah_tree.py: binned-SAH tree (as sah_build.c) followed by the reinsertion optimisation of Bittner, Hapala,
 * Havran 2013 ("Fast insertion-based optimization of bounding volume hierarchies"), sequential: every inner node's
 * children are removed and reinserted where they enlarge the tree's total surface area least (branch and bound).
 * Output in the `compact` pre-order layout.  gcc -O2 -shared -fPIC -o reinsert_opt.so reinsert_opt.c -lm */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#define BINS 32
typedef struct { float mn[3], mx[3]; } box_t;
typedef struct { box_t b; int parent, left, right, prim; } node_t;
static node_t *nd; static int n_nodes, root;
static const float *g_box;
static inline float area(const box_t *b) { float dx = b->mx[0] - b->mn[0], dy = b->mx[1] - b->mn[1], dz = b->mx[2] - b->mn[2]; return dx * dy + dy * dz + dz * dx; }
static inline void empty(box_t *b) { for (int k = 0; k < 3; k++) { b->mn[k] = 1e30f; b->mx[k] = -1e30f; } }
static inline void grow(box_t *b, const box_t *p) { for (int k = 0; k < 3; k++) { if (p->mn[k] < b->mn[k]) b->mn[k] = p->mn[k]; if (p->mx[k] > b->mx[k]) b->mx[k] = p->mx[k]; } }
static inline box_t join(const box_t *a, const box_t *b) { box_t r = *a; grow(&r, b); return r; }

static int build(int *idx, int cnt, int parent)
{
    const int me = n_nodes++;
    node_t *q = &nd[me];
    q->parent = parent; empty(&q->b);
    for (int i = 0; i < cnt; i++) { box_t t; const float *p = g_box + (size_t)idx[i] * 6; for (int k = 0; k < 3; k++) { t.mn[k] = p[k]; t.mx[k] = p[3 + k]; } grow(&q->b, &t); }
    if (cnt == 1) { q->left = q->right = -1; q->prim = idx[0]; return me; }
    q->prim = -1;
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
            const float *p = g_box + (size_t)idx[i] * 6; box_t t; for (int k = 0; k < 3; k++) { t.mn[k] = p[k]; t.mx[k] = p[3 + k]; }
            int b = (int)((0.5f * (p[ax] + p[3 + ax]) - cmn[ax]) * scale); if (b >= BINS) b = BINS - 1; if (b < 0) b = 0;
            grow(&bin[b], &t); bc[b]++;
        }
        for (int sp = 0; sp < BINS - 1; sp++) {
            box_t L, R; empty(&L); empty(&R); int cl = 0, cr = 0;
            for (int b = 0; b < BINS; b++) if (bc[b]) { if (b <= sp) { grow(&L, &bin[b]); cl += bc[b]; } else { grow(&R, &bin[b]); cr += bc[b]; } }
            if (cl == 0 || cr == 0) continue;
            const float cost = area(&L) * cl + area(&R) * cr;
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = sp; }
        }
    }
    int mid;
    if (best_axis < 0) mid = cnt / 2;
    else {
        const float ext = cmx[best_axis] - cmn[best_axis], scale = BINS / ext;
        int i = 0, j = cnt - 1;
        while (i <= j) {
            const float *p = g_box + (size_t)idx[i] * 6;
            int b = (int)((0.5f * (p[best_axis] + p[3 + best_axis]) - cmn[best_axis]) * scale); if (b >= BINS) b = BINS - 1; if (b < 0) b = 0;
            if (b <= best_split) i++; else { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; j--; }
        }
        mid = i; if (mid == 0 || mid == cnt) mid = cnt / 2;
    }
    const int l = build(idx, mid, me), r = build(idx + mid, cnt - mid, me);
    nd[me].left = l; nd[me].right = r;
    return me;
}
static void refit_up(int i) { while (i >= 0) { nd[i].b = join(&nd[nd[i].left].b, &nd[nd[i].right].b); i = nd[i].parent; } }
static double total_cost(void) { double c = 0; for (int i = 0; i < n_nodes; i++) if (nd[i].left >= 0) c += area(&nd[i].b); return c; }

/* best place for subtree L: the node X such that making a new parent of (X, L) adds the least area (induced cost along the path + direct cost) */
typedef struct { int node; float induced; } cand_t;
static int find_best(int L, float *best_cost_out)
{
    static cand_t *heap = NULL; static int heap_cap = 0;
    if (!heap) { heap_cap = 1 << 16; heap = (cand_t *)malloc(sizeof(cand_t) * heap_cap); }
    int hn = 0; float best = 1e30f; int best_node = -1;
    const float la = area(&nd[L].b);
    heap[hn++] = (cand_t){root, 0.0f};
    while (hn > 0) {
        /* pop the smallest induced cost */
        int bi = 0; for (int i = 1; i < hn; i++) if (heap[i].induced < heap[bi].induced) bi = i;
        cand_t c = heap[bi]; heap[bi] = heap[--hn];
        if (c.induced + la >= best) break;          /* lower bound: induced + area(L) */
        box_t j = join(&nd[c.node].b, &nd[L].b);
        const float direct = area(&j), tot = c.induced + direct;
        if (tot < best) { best = tot; best_node = c.node; }
        const float child_induced = tot - area(&nd[c.node].b);
        if (nd[c.node].left >= 0 && child_induced + la < best) {
            if (hn + 2 > heap_cap) { heap_cap *= 2; heap = (cand_t *)realloc(heap, sizeof(cand_t) * heap_cap); }
            heap[hn++] = (cand_t){nd[c.node].left, child_induced};
            heap[hn++] = (cand_t){nd[c.node].right, child_induced};
        }
    }
    *best_cost_out = best;
    return best_node;
}
/* detach subtree s (and its parent p): p's other child takes p's place.  Returns the freed node p. */
static int detach(int s)
{
    const int p = nd[s].parent, g = nd[p].parent, sib = nd[p].left == s ? nd[p].right : nd[p].left;
    nd[sib].parent = g;
    if (g >= 0) { if (nd[g].left == p) nd[g].left = sib; else nd[g].right = sib; refit_up(g); } else root = sib;
    return p;
}
static void attach(int s, int x, int freed)
{
    const int g = nd[x].parent;
    nd[freed].parent = g; nd[freed].left = x; nd[freed].right = s; nd[freed].prim = -1;
    nd[x].parent = freed; nd[s].parent = freed;
    if (g >= 0) { if (nd[g].left == x) nd[g].left = freed; else nd[g].right = freed; } else root = freed;
    refit_up(freed);
}
static float *g_out; static int *g_size; static int g_next;
static int emit(int i)
{
    const int me = g_next++;
    float *row = g_out + (size_t)me * 9;
    for (int k = 0; k < 3; k++) { row[2 + k] = nd[i].b.mn[k]; row[5 + k] = nd[i].b.mx[k]; }
    row[8] = 0.0f;
    if (nd[i].left < 0) { row[0] = 1.0f; row[1] = (float)nd[i].prim; g_size[me] = 1; return me; }
    row[0] = 0.0f;
    emit(nd[i].left);
    row[1] = (float)emit(nd[i].right);
    g_size[me] = g_next - me;
    return me;
}
static int cmp_area(const void *a, const void *b) { float x = area(&nd[*(const int *)a].b), y = area(&nd[*(const int *)b].b); return x > y ? -1 : (x < y ? 1 : 0); }
int reinsert_build(const float *boxes, int n, int passes, float *compact_out, int *csize_out, double *cost_before, double *cost_after)
{
    g_box = boxes;
    nd = (node_t *)malloc(sizeof(node_t) * (size_t)(2 * n));
    n_nodes = 0;
    int *idx = (int *)malloc(sizeof(int) * n); for (int i = 0; i < n; i++) idx[i] = i;
    root = build(idx, n, -1);
    free(idx);
    *cost_before = total_cost();
    int *order = (int *)malloc(sizeof(int) * n_nodes);
    for (int pass = 0; pass < passes; pass++) {
        int m = 0;
        for (int i = 0; i < n_nodes; i++) if (nd[i].left >= 0 && i != root && nd[i].parent != root) order[m++] = i;
        qsort(order, m, sizeof(int), cmp_area);          /* largest inner nodes first */
        for (int k = 0; k < m; k++) {
            const int v = order[k];
            if (nd[v].left < 0 || v == root || nd[v].parent < 0 || nd[v].parent == root) continue;
            /* remove v: its two children are reinserted one after the other (larger first); v and its parent are the two free nodes */
            int a = nd[v].left, b = nd[v].right;
            if (area(&nd[a].b) < area(&nd[b].b)) { int t = a; a = b; b = t; }
            const int p = nd[v].parent, g = nd[p].parent, sib = nd[p].left == v ? nd[p].right : nd[p].left;
            nd[sib].parent = g;
            if (g >= 0) { if (nd[g].left == p) nd[g].left = sib; else nd[g].right = sib; refit_up(g); } else root = sib;
            float c; int x = find_best(a, &c); attach(a, x, v);
            x = find_best(b, &c); attach(b, x, p);
        }
    }
    *cost_after = total_cost();
    g_out = compact_out; g_size = csize_out; g_next = 0;
    emit(root);
    free(order); free(nd);
    return g_next;
}
