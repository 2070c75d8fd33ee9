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

/* How many of the triangle tests would a PLANE pre-test spare?  A ray that passes a leaf's box misses its triangle mostly by crossing the triangle's plane
 * outside the box: reject the leaf when the crossing distance t_p lies outside [t_near, min(t_far, best)] of the leaf's own box (widened by rel).
 * out: [0] triangle tests without, [1] with the pre-test, [2] hits (must not change) */
void simulate_plane_pretest(const float *compact, const int *ref_tri, const float *tris, const float *rays, int nr, float rel, double *out)
{
    double tests = 0, tests2 = 0, hits = 0;
    for (int r = 0; r < nr; r++) {
        const float *o = rays + (size_t)r * 6, *d = o + 3;
        float id[3]; for (int k = 0; k < 3; k++) id[k] = 1.0f / (fabsf(d[k]) > 1e-12f ? d[k] : 1e-12f);
        float best = 1e30f, tn;
        int stack[256], sp = 0;
        if (slab(o, id, compact + 2, best, &tn)) stack[sp++] = 0;
        while (sp) {
            int n = stack[--sp];
            const float *row = compact + (size_t)n * 9;
            if (row[0] != 0.0f) {
                tests += 1;
                const float *v = tris + (size_t)ref_tri[(int)row[1]] * 9;
                /* box interval of this leaf */
                float t0 = 0.0f, t1 = best;
                for (int k = 0; k < 3; k++) { float a = (row[2 + k] - o[k]) * id[k], c = (row[5 + k] - o[k]) * id[k]; if (a > c) { float t = a; a = c; c = t; } if (a > t0) t0 = a; if (c < t1) t1 = c; }
                float e1[3], e2[3], nn[3];
                for (int k = 0; k < 3; k++) { e1[k] = v[3 + k] - v[k]; e2[k] = v[6 + k] - v[k]; }
                nn[0] = e1[1] * e2[2] - e1[2] * e2[1]; nn[1] = e1[2] * e2[0] - e1[0] * e2[2]; nn[2] = e1[0] * e2[1] - e1[1] * e2[0];
                float nd = nn[0] * d[0] + nn[1] * d[1] + nn[2] * d[2], no = nn[0] * (v[0] - o[0]) + nn[1] * (v[1] - o[1]) + nn[2] * (v[2] - o[2]);
                int keep = 1;
                if (fabsf(nd) > 1e-20f) { float tp = no / nd; float w = rel * (fabsf(tp) + 1e-6f); if (tp < t0 - w || tp > t1 + w) keep = 0; }
                float t = tri_hit(o, d, v);
                if (keep) tests2 += 1; else if (t < best) { /* a wrongly rejected hit */ tests2 += 1e9; }
                if (t < best) best = t;
                continue;
            }
            int l = n + 1, rr = (int)row[1];
            float tl, tr;
            int hl = slab(o, id, compact + (size_t)l * 9 + 2, best, &tl), hr = slab(o, id, compact + (size_t)rr * 9 + 2, best, &tr);
            if (hl && hr) { if (tl < tr) { stack[sp++] = rr; stack[sp++] = l; } else { stack[sp++] = l; stack[sp++] = rr; } }
            else if (hl) stack[sp++] = l;
            else if (hr) stack[sp++] = rr;
        }
        if (best < 1e29f) hits += 1;
    }
    out[0] = tests / nr; out[1] = tests2 / nr; out[2] = hits / nr;
}

/* Pre-tests that need only a small per-triangle record, not the leaf's box interval: the plane crossing t_p must be in (0, best] and the crossing POINT must lie
 * (mode 0) inside the triangle's bounding sphere (centre = box centre... here: centroid-of-box, radius to the farthest vertex), (mode 1) inside its AABB, both widened by rel. */
void simulate_point_pretest(const float *compact, const int *ref_tri, const float *tris, const float *rays, int nr, float rel, int mode, double *out)
{
    double tests = 0, tests2 = 0, bad = 0;
    for (int r = 0; r < nr; r++) {
        const float *o = rays + (size_t)r * 6, *d = o + 3;
        float id[3]; for (int k = 0; k < 3; k++) id[k] = 1.0f / (fabsf(d[k]) > 1e-12f ? d[k] : 1e-12f);
        float best = 1e30f, tn;
        int stack[256], sp = 0;
        if (slab(o, id, compact + 2, best, &tn)) stack[sp++] = 0;
        while (sp) {
            int n = stack[--sp];
            const float *row = compact + (size_t)n * 9;
            if (row[0] != 0.0f) {
                tests += 1;
                const float *v = tris + (size_t)ref_tri[(int)row[1]] * 9;
                float e1[3], e2[3], nn[3], mn[3], mx[3], ce[3], r2 = 0.0f;
                for (int k = 0; k < 3; k++) { e1[k] = v[3 + k] - v[k]; e2[k] = v[6 + k] - v[k]; mn[k] = fminf(v[k], fminf(v[3 + k], v[6 + k])); mx[k] = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k])); ce[k] = 0.5f * (mn[k] + mx[k]); }
                for (int j = 0; j < 3; j++) { float q = 0; for (int k = 0; k < 3; k++) q += (v[3 * j + k] - ce[k]) * (v[3 * j + k] - ce[k]); if (q > r2) r2 = q; }
                nn[0] = e1[1] * e2[2] - e1[2] * e2[1]; nn[1] = e1[2] * e2[0] - e1[0] * e2[2]; nn[2] = e1[0] * e2[1] - e1[1] * e2[0];
                float nd = nn[0] * d[0] + nn[1] * d[1] + nn[2] * d[2], no = nn[0] * (v[0] - o[0]) + nn[1] * (v[1] - o[1]) + nn[2] * (v[2] - o[2]);
                int keep = 1;
                float nlen = sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
                if (fabsf(nd) > 1e-4f * nlen) {
                    float tp = no / nd;
                    if (tp < -rel * fabsf(tp) - 1e-6f || tp > best * (1.0f + rel)) keep = 0;
                    else {
                        float p[3]; for (int k = 0; k < 3; k++) p[k] = o[k] + tp * d[k];
                        if (mode == 0) { float q = 0; for (int k = 0; k < 3; k++) q += (p[k] - ce[k]) * (p[k] - ce[k]); if (q > r2 * (1.0f + rel) * (1.0f + rel) + 1e-12f) keep = 0; }
                        else { for (int k = 0; k < 3; k++) { float w = rel * (mx[k] - mn[k]) + rel * fabsf(p[k]) * 1e-3f + 1e-7f; if (p[k] < mn[k] - w || p[k] > mx[k] + w) keep = 0; } }
                    }
                }
                float t = tri_hit(o, d, v);
                if (keep) tests2 += 1; else if (t < best) bad += 1;
                if (t < best) best = t;
                continue;
            }
            int l = n + 1, rr = (int)row[1];
            float tl, tr;
            int hl = slab(o, id, compact + (size_t)l * 9 + 2, best, &tl), hr = slab(o, id, compact + (size_t)rr * 9 + 2, best, &tr);
            if (hl && hr) { if (tl < tr) { stack[sp++] = rr; stack[sp++] = l; } else { stack[sp++] = l; stack[sp++] = rr; } }
            else if (hl) stack[sp++] = l;
            else if (hr) stack[sp++] = rr;
        }
    }
    out[0] = tests / nr; out[1] = tests2 / nr; out[2] = bad;
}
