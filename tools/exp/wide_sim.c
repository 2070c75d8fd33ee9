/* tools/exp/wide_sim.py: the binary SAH tree of sah_build.c collapsed to k-wide nodes (greedy: open the child with the largest box until k children),
 * walked like k_trace's ordered mode (children near to far, entry distances beyond the best hit culled), counting per ray the wide nodes visited,
 * the leaves tested and the length of the dependent chain (visits + leaf tests).  Child order: sorted by entry distance, or fixed per ray octant
 * (children ranked by their centre along the octant's diagonal -- what an octant-ordered 8-wide node gives without a sort).
 * gcc -O2 -shared -fPIC -o wide_sim.so wide_sim.c -lm */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#define KMAX 16
typedef struct { int n; int child[KMAX]; } Wide;     /* child >= 0: wide node index; < 0: ~binary leaf row */
static const float *g_compact;
static float area(int row) { const float *b = g_compact + (size_t)row * 9 + 2; float x = b[3] - b[0], y = b[4] - b[1], z = b[5] - b[2]; return x * y + y * z + z * x; }
static int is_leaf(int row) { return g_compact[(size_t)row * 9] != 0.0f; }
static Wide *g_w; static int g_nw;
static int collapse(int row, int k)
{
    int me = g_nw++;
    int c[KMAX], n = 0;
    c[n++] = row + 1; c[n++] = (int)g_compact[(size_t)row * 9 + 1];
    while (n < k) {
        int best = -1; float ba = -1.0f;
        for (int i = 0; i < n; i++) if (!is_leaf(c[i])) { float a = area(c[i]); if (a > ba) { ba = a; best = i; } }
        if (best < 0) break;
        int r = c[best];
        c[best] = r + 1; c[n++] = (int)g_compact[(size_t)r * 9 + 1];
    }
    g_w[me].n = n;
    for (int i = 0; i < n; i++) g_w[me].child[i] = is_leaf(c[i]) ? ~c[i] : -0x40000000;     /* placeholder */
    for (int i = 0; i < n; i++) if (!is_leaf(c[i])) { int id = collapse(c[i], k); g_w[me].child[i] = id; }
    /* remember the binary row of internal children for their boxes: stored in a parallel array */
    return me;
}
static int *g_rowof;      /* wide node -> binary row (its box) */
static void rows(int row, int k, int *cursor)
{
    int me = (*cursor)++;
    g_rowof[me] = row;
    int c[KMAX], n = 0;
    c[n++] = row + 1; c[n++] = (int)g_compact[(size_t)row * 9 + 1];
    while (n < k) {
        int best = -1; float ba = -1.0f;
        for (int i = 0; i < n; i++) if (!is_leaf(c[i])) { float a = area(c[i]); if (a > ba) { ba = a; best = i; } }
        if (best < 0) break;
        int r = c[best];
        c[best] = r + 1; c[n++] = (int)g_compact[(size_t)r * 9 + 1];
    }
    for (int i = 0; i < n; i++) if (!is_leaf(c[i])) rows(c[i], k, cursor);
}
static inline int slab(const float *o, const float *id, const float *b, float tmax, float *tn)
{
    float t0 = 0.0f, t1 = tmax;
    for (int k = 0; k < 3; k++) {
        float a = (b[k] - o[k]) * id[k], c = (b[3 + k] - o[k]) * id[k];
        if (a > c) { float t = a; a = c; c = t; }
        if (a > t0) t0 = a;
        if (c < t1) t1 = c;
    }
    *tn = t0;
    return t0 <= t1;
}
static float tri_hit(const float *o, const float *d, const float *v)
{
    float e1[3], e2[3], p[3], t[3], q[3];
    for (int k = 0; k < 3; k++) { e1[k] = v[3 + k] - v[k]; e2[k] = v[6 + k] - v[k]; }
    p[0] = d[1] * e2[2] - d[2] * e2[1]; p[1] = d[2] * e2[0] - d[0] * e2[2]; p[2] = d[0] * e2[1] - d[1] * e2[0];
    float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (fabsf(det) < 1e-20f) return 1e30f;
    float inv = 1.0f / det;
    for (int k = 0; k < 3; k++) t[k] = o[k] - v[k];
    float u = (t[0] * p[0] + t[1] * p[1] + t[2] * p[2]) * inv;
    if (u < 0.0f || u > 1.0f) return 1e30f;
    q[0] = t[1] * e1[2] - t[2] * e1[1]; q[1] = t[2] * e1[0] - t[0] * e1[2]; q[2] = t[0] * e1[1] - t[1] * e1[0];
    float w = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
    if (w < 0.0f || u + w > 1.0f) return 1e30f;
    float tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    return tt > 1e-4f ? tt : 1e30f;
}
/* mode 0: children sorted by entry distance; 1: fixed octant order.  quant > 0: child boxes snapped outward to a grid of `quant` cells per axis of the
 * PARENT's box (8-bit nodes: 255).  out: [0] wide visits mean, [1] leaf tests mean, [2] child boxes hit per visit, [3] chain p50, [4] p99, [5] p99.9, [6] max, [7] hits;  per_ray[nr] chain */
static float g_margin = 0.0f;      /* every child box grown by this much on every side (absolute units): what a box test of lower precision has to add to stay conservative */
void set_margin(float m) { g_margin = m; }
void simulate_wide(const float *compact, int nrows, int k, int mode, int quant, const int *ref_tri, const float *tris, const float *rays, int nr, double *out, int *per_ray)
{
    g_compact = compact;
    g_w = (Wide *)malloc(sizeof(Wide) * (size_t)nrows); g_nw = 0;
    g_rowof = (int *)malloc(sizeof(int) * (size_t)nrows);
    collapse(0, k);
    int cur = 0; rows(0, k, &cur);
    /* child boxes per wide node (optionally quantised to the parent's box) */
    float *cb = (float *)malloc(sizeof(float) * 6 * KMAX * (size_t)g_nw);
    for (int w = 0; w < g_nw; w++) {
        const float *pb = compact + (size_t)g_rowof[w] * 9 + 2;
        for (int i = 0; i < g_w[w].n; i++) {
            int c = g_w[w].child[i];
            int row = c < 0 ? ~c : g_rowof[c];
            const float *b = compact + (size_t)row * 9 + 2;
            float *o = cb + ((size_t)w * KMAX + i) * 6;
            for (int a = 0; a < 3; a++) {
                if (quant > 0) {
                    float ext = pb[3 + a] - pb[a]; if (ext <= 0) ext = 1e-30f;
                    float cell = ext / quant;
                    o[a] = pb[a] + floorf((b[a] - pb[a]) / cell) * cell;
                    o[3 + a] = pb[a] + ceilf((b[3 + a] - pb[a]) / cell) * cell;
                } else { o[a] = b[a]; o[3 + a] = b[3 + a]; }
                o[a] -= g_margin; o[3 + a] += g_margin;
            }
        }
    }
    double visits = 0, tests = 0, hits = 0, chit = 0;
    for (int r = 0; r < nr; r++) {
        const float *o = rays + (size_t)r * 6, *d = o + 3;
        float id[3]; for (int a = 0; a < 3; a++) id[a] = 1.0f / (fabsf(d[a]) > 1e-12f ? d[a] : 1e-12f);
        float best = 1e30f;
        int stack[512], sp = 0; float sdist[512];
        stack[sp] = 0; sdist[sp++] = 0.0f;
        int chain = 0;
        while (sp) {
            --sp;
            int n = stack[sp];
            if (sdist[sp] > best) continue;                /* culled at pop (the device culls at push and re-checks nothing: close enough) */
            if (n < 0) {
                tests += 1; chain++;
                float t = tri_hit(o, d, tris + (size_t)ref_tri[(int)compact[(size_t)(~n) * 9 + 1]] * 9);
                if (t < best) best = t;
                continue;
            }
            visits += 1; chain++;
            const Wide *w = g_w + n;
            int idx[KMAX]; float dist[KMAX], key[KMAX]; int m = 0;
            for (int i = 0; i < w->n; i++) {
                float tn;
                const float *b = cb + ((size_t)n * KMAX + i) * 6;
                if (slab(o, id, b, best, &tn)) {
                    idx[m] = i; dist[m] = tn;
                    key[m] = mode == 0 ? tn : ((b[0] + b[3]) * (d[0] < 0 ? -1.f : 1.f) + (b[1] + b[4]) * (d[1] < 0 ? -1.f : 1.f) + (b[2] + b[5]) * (d[2] < 0 ? -1.f : 1.f));
                    m++;
                }
            }
            chit += m;
            /* push far to near by key */
            for (int i = 1; i < m; i++) { int ii = idx[i]; float dd = dist[i], kk = key[i]; int j = i - 1; while (j >= 0 && key[j] < kk) { idx[j + 1] = idx[j]; dist[j + 1] = dist[j]; key[j + 1] = key[j]; j--; } idx[j + 1] = ii; dist[j + 1] = dd; key[j + 1] = kk; }
            for (int i = 0; i < m; i++) { stack[sp] = w->child[idx[i]]; sdist[sp++] = dist[i]; }
        }
        if (best < 1e29f) hits += 1;
        per_ray[r] = chain;
    }
    out[0] = visits / nr; out[1] = tests / nr; out[2] = chit / (visits > 0 ? visits : 1); out[7] = hits / nr; out[8] = g_nw;
    free(g_w); free(g_rowof); free(cb);
}
