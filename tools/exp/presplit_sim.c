/* tools/exp/presplit_sim.py: ordered, t-culled traversal of a binary BVH in the `compact` layout of sah_build.c over REFERENCES (boxes that each
 * point at a triangle; several may point at the same one), counting what a ray costs: internal nodes visited and triangles tested.
 * gcc -O2 -shared -fPIC -o presplit_sim.so presplit_sim.c */
#include <math.h>
#include <stdlib.h>
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
/* compact rows [N][9]; ref_tri[ref] -> triangle; tris [ntri][9]; rays [nr][6]; out: [0] internal visits, [1] triangle tests, [2] hits */
void simulate(const float *compact, const int *ref_tri, const float *tris, const float *rays, int nr, double *out, float *t_out)
{
    double visits = 0, tests = 0, hits = 0;
    for (int r = 0; r < nr; r++) {
        const float *o = rays + (size_t)r * 6, *d = o + 3;
        float id[3]; for (int k = 0; k < 3; k++) id[k] = 1.0f / (fabsf(d[k]) > 1e-12f ? d[k] : 1e-12f);
        float best = 1e30f, tn;
        int stack[256], sp = 0;
        if (slab(o, id, compact + 2, best, &tn)) stack[sp++] = 0;
        while (sp) {
            int n = stack[--sp];
            const float *row = compact + (size_t)n * 9;
            if (row[0] != 0.0f) {                 /* leaf */
                tests += 1;
                float t = tri_hit(o, d, tris + (size_t)ref_tri[(int)row[1]] * 9);
                if (t < best) best = t;
                continue;
            }
            visits += 1;
            int l = n + 1, rr = (int)row[1];
            float tl, tr;
            int hl = slab(o, id, compact + (size_t)l * 9 + 2, best, &tl), hr = slab(o, id, compact + (size_t)rr * 9 + 2, best, &tr);
            if (hl && hr) { if (tl < tr) { stack[sp++] = rr; stack[sp++] = l; } else { stack[sp++] = l; stack[sp++] = rr; } }
            else if (hl) stack[sp++] = l;
            else if (hr) stack[sp++] = rr;
        }
        if (best < 1e29f) hits += 1;
        if (t_out) t_out[r] = best;
    }
    out[0] = visits / nr; out[1] = tests / nr; out[2] = hits / nr;
}

/* the same tree with every subtree of at most `pk` leaves taken as ONE leaf packet: reaching it costs one packet visit and a test of each of
 * its triangles (no boxes inside).  out: [0] internal visits above the packets, [1] packet visits, [2] triangle tests */
void simulate_packets(const float *compact, const int *csize, int pk, const int *ref_tri, const float *tris, const float *rays, int nr, double *out)
{
    double visits = 0, packets = 0, tests = 0;
    for (int r = 0; r < nr; r++) {
        const float *o = rays + (size_t)r * 6, *d = o + 3;
        float id[3]; for (int k = 0; k < 3; k++) id[k] = 1.0f / (fabsf(d[k]) > 1e-12f ? d[k] : 1e-12f);
        float best = 1e30f, tn;
        int stack[256], sp = 0;
        if (slab(o, id, compact + 2, best, &tn)) stack[sp++] = 0;
        while (sp) {
            int n = stack[--sp];
            const float *row = compact + (size_t)n * 9;
            if (csize[n] <= 2 * pk - 1) {         /* a packet: all leaves of the subtree */
                packets += 1;
                for (int m = n; m < n + csize[n]; m++) {
                    const float *rw = compact + (size_t)m * 9;
                    if (rw[0] != 0.0f) { tests += 1; float t = tri_hit(o, d, tris + (size_t)ref_tri[(int)rw[1]] * 9); if (t < best) best = t; }
                }
                continue;
            }
            visits += 1;
            int l = n + 1, rr = (int)row[1];
            float tl, tr;
            int hl = slab(o, id, compact + (size_t)l * 9 + 2, best, &tl), hr = slab(o, id, compact + (size_t)rr * 9 + 2, best, &tr);
            if (hl && hr) { if (tl < tr) { stack[sp++] = rr; stack[sp++] = l; } else { stack[sp++] = l; stack[sp++] = rr; } }
            else if (hl) stack[sp++] = l;
            else if (hr) stack[sp++] = rr;
        }
    }
    out[0] = visits / nr; out[1] = packets / nr; out[2] = tests / nr;
}
