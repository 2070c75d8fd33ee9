/* tools/exp/sah_tree.py: PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) restated sequentially, one
 * primitive per leaf, output in the `compact` pre-order layout.  order[] = primitives in Morton order.
 * gcc -O2 -shared -fPIC -o ploc_build.so ploc_build.c */
#include <stdlib.h>
#include <string.h>
typedef struct { float mn[3], mx[3]; int left, right, prim, size; } node_t;
static node_t *nd;
static float *g_out; static int *g_size; static int g_next;
static inline float merged_area(const node_t *a, const node_t *b)
{
    float d[3];
    for (int k = 0; k < 3; k++) { float lo = a->mn[k] < b->mn[k] ? a->mn[k] : b->mn[k], hi = a->mx[k] > b->mx[k] ? a->mx[k] : b->mx[k]; d[k] = hi - lo; }
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}
static int emit(int i)
{
    const int me = g_next++;
    float *row = g_out + (size_t)me * 9;
    for (int k = 0; k < 3; k++) { row[2 + k] = nd[i].mn[k]; row[5 + k] = nd[i].mx[k]; }
    row[8] = 0.0f;
    if (nd[i].left < 0) { row[0] = 1.0f; row[1] = (float)nd[i].prim; g_size[me] = 1; return me; }
    row[0] = 0.0f;
    emit(nd[i].left);
    row[1] = (float)emit(nd[i].right);
    g_size[me] = g_next - me;
    return me;
}
int ploc_build(const float *boxes, const int *order, int n, int radius, float *compact_out, int *csize_out, int *iters_out)
{
    nd = (node_t *)malloc(sizeof(node_t) * (size_t)(2 * n));
    int *cur = (int *)malloc(sizeof(int) * n), *nxt = (int *)malloc(sizeof(int) * n), *nn = (int *)malloc(sizeof(int) * n);
    int made = 0;
    for (int i = 0; i < n; i++) {
        const float *p = boxes + (size_t)order[i] * 6;
        node_t *q = &nd[made];
        for (int k = 0; k < 3; k++) { q->mn[k] = p[k]; q->mx[k] = p[3 + k]; }
        q->left = q->right = -1; q->prim = order[i]; q->size = 1;
        cur[i] = made++;
    }
    int cnt = n, iters = 0;
    while (cnt > 1) {
        iters++;
        for (int i = 0; i < cnt; i++) {
            int lo = i - radius < 0 ? 0 : i - radius, hi = i + radius >= cnt ? cnt - 1 : i + radius, best = -1; float ba = 1e30f;
            for (int j = lo; j <= hi; j++) if (j != i) { float a = merged_area(&nd[cur[i]], &nd[cur[j]]); if (a < ba) { ba = a; best = j; } }
            nn[i] = best;
        }
        int out = 0;
        for (int i = 0; i < cnt; i++) {
            const int j = nn[i];
            if (nn[j] == i) {
                if (i < j) {
                    node_t *q = &nd[made]; const node_t *a = &nd[cur[i]], *b = &nd[cur[j]];
                    for (int k = 0; k < 3; k++) { q->mn[k] = a->mn[k] < b->mn[k] ? a->mn[k] : b->mn[k]; q->mx[k] = a->mx[k] > b->mx[k] ? a->mx[k] : b->mx[k]; }
                    q->left = cur[i]; q->right = cur[j]; q->prim = -1; q->size = a->size + b->size + 1;
                    nxt[out++] = made++;
                }
            } else nxt[out++] = cur[i];
        }
        int *t = cur; cur = nxt; nxt = t; cnt = out;
    }
    g_out = compact_out; g_size = csize_out; g_next = 0;
    emit(cur[0]);
    *iters_out = iters;
    free(nd); free(cur); free(nxt); free(nn);
    return g_next;
}
