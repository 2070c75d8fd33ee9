/* tools/exp/sah_tree.py: exact-sweep SAH (every split position of the centroid-sorted order on every axis) for nodes of at
 * most SWEEP_MAX primitives, 32-bin SAH above.  Same `compact` output as sah_build.c.
 * gcc -O2 -shared -fPIC -o sweep_build.so sweep_build.c */
#include <stdlib.h>
#include <string.h>
#ifndef BINS
#define BINS 32
#endif
typedef struct { float mn[3], mx[3]; } box_t;
static const float *g_box; static float *g_out; static int *g_size; static int g_next; static int g_sweep_max;
static inline void grow(box_t *b, const float *p) { for (int k = 0; k < 3; k++) { if (p[k] < b->mn[k]) b->mn[k] = p[k]; if (p[3 + k] > b->mx[k]) b->mx[k] = p[3 + k]; } }
static inline float area(const box_t *b) { float dx = b->mx[0] - b->mn[0], dy = b->mx[1] - b->mn[1], dz = b->mx[2] - b->mn[2]; return dx * dy + dy * dz + dz * dx; }
static void empty(box_t *b) { for (int k = 0; k < 3; k++) { b->mn[k] = 1e30f; b->mx[k] = -1e30f; } }
static int g_axis;
static int cmp(const void *a, const void *b)
{
    const float *p = g_box + (size_t)(*(const int *)a) * 6, *q = g_box + (size_t)(*(const int *)b) * 6;
    const float x = p[g_axis] + p[3 + g_axis], y = q[g_axis] + q[3 + g_axis];
    return x < y ? -1 : (x > y ? 1 : (*(const int *)a - *(const int *)b));
}
static int build(int *idx, int cnt)
{
    const int me = g_next++;
    float *row = g_out + (size_t)me * 9;
    box_t bb; empty(&bb);
    for (int i = 0; i < cnt; i++) grow(&bb, g_box + (size_t)idx[i] * 6);
    for (int k = 0; k < 3; k++) { row[2 + k] = bb.mn[k]; row[5 + k] = bb.mx[k]; }
    row[8] = 0.0f;
    if (cnt == 1) { row[0] = 1.0f; row[1] = (float)idx[0]; g_size[me] = 1; return me; }
    int mid = -1;
    if (cnt <= g_sweep_max) {
        float best = 1e30f; int best_axis = -1, best_pos = -1;
        float *ra = (float *)malloc(sizeof(float) * cnt);
        int *tmp = (int *)malloc(sizeof(int) * cnt);
        for (int ax = 0; ax < 3; ax++) {
            memcpy(tmp, idx, sizeof(int) * cnt);
            g_axis = ax; qsort(tmp, cnt, sizeof(int), cmp);
            box_t acc; empty(&acc);
            for (int i = cnt - 1; i > 0; i--) { grow(&acc, g_box + (size_t)tmp[i] * 6); ra[i] = area(&acc); }
            empty(&acc);
            for (int i = 0; i < cnt - 1; i++) {
                grow(&acc, g_box + (size_t)tmp[i] * 6);
                const float cost = area(&acc) * (i + 1) + ra[i + 1] * (cnt - i - 1);
                if (cost < best) { best = cost; best_axis = ax; best_pos = i + 1; }
            }
        }
        g_axis = best_axis; qsort(idx, cnt, sizeof(int), cmp);
        mid = best_pos;
        free(ra); free(tmp);
    } else {
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
            for (int sp = 0; sp < BINS - 1; sp++) {
                box_t L, R; empty(&L); empty(&R); int cl = 0, cr = 0;
                for (int b = 0; b < BINS; b++) if (bc[b]) { float q[6] = {bin[b].mn[0], bin[b].mn[1], bin[b].mn[2], bin[b].mx[0], bin[b].mx[1], bin[b].mx[2]}; if (b <= sp) { grow(&L, q); cl += bc[b]; } else { grow(&R, q); cr += bc[b]; } }
                if (cl == 0 || cr == 0) continue;
                const float cost = area(&L) * cl + area(&R) * cr;
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = sp; }
            }
        }
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
    }
    row[0] = 0.0f;
    build(idx, mid);
    row[1] = (float)build(idx + mid, cnt - mid);
    g_size[me] = g_next - me;
    return me;
}
int sweep_build(const float *boxes, int n, int sweep_max, float *compact_out, int *csize_out)
{
    int *idx = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) idx[i] = i;
    g_box = boxes; g_out = compact_out; g_size = csize_out; g_next = 0; g_sweep_max = sweep_max;
    build(idx, n);
    free(idx);
    return g_next;
}
