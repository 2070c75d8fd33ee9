/*
 * oracle.c -- CPU restatement of the ti-raytrace hot path (LBVH build + PT_RGB ray loop).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so; the product path
 * (ti_raytrace_amd -> libtirt.so, HIP) never does and fails loudly without its extension.
 *
 * The reference (lyd405121/ti-raytrace) is Python + Taichi 0.7.14 DSL; Taichi is not
 * installable here, so the reference cannot be imported or compiled.
 * Every function below restates one reference function literally -- same data layouts
 * (AoS 9-float vertex rows, 3-int primitive rows, 11-float bvh_node, 9-float compact_node,
 * per-ray global stack), same operation order in fp32, same quirks
 * -- and cites the file:line it follows.  Compiled with -ffp-contract=off.
 *
 * Pins (tests/test_oracle_golden.py):
 *   * nodelist.txt  (reference accel/LBvh.py:164-172 output for model/cornell_box.obj)
 *     reproduced 71/71 lines  -> pins ingest order, Morton, stable sort, Karras topology,
 *     refit, DFS flatten.
 *   * out.png (reference example/Example.py:49, Cornell PT_RGB 512^2 512spp) matched
 *     statistically (block means) -> pins camera, traversal, intersection, shading, film.
 *   Everything else is "parity unpinned" by the reference (it has no tests): SURVEY.md 8c.
 *
 * Scalar transcendental functions and the counter-based RNG come from
 * ti_raytrace_amd/csrc/tirt_math.h (shared, deterministic; validated against libm in
 * tests/test_math.py).  Build with -DORACLE_LIBM to swap in libm instead (used by a test
 * to show the shared math does not bias the image).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>
#include "tirt_math.h"

#ifdef ORACLE_LIBM
#include <math.h>
#define m_sin sinf
#define m_cos cosf
#define m_exp expf
#define m_pow powf
#define m_atan2 atan2f
#define m_acos acosf
#define m_tan tanf
#else
#define m_sin tm_sin
#define m_cos tm_cos
#define m_exp tm_exp
#define m_pow tm_pow
#define m_tan tm_tan
#define m_atan2 tm_atan2
#define m_acos tm_acos
#endif
#define m_sqrt tm_sqrt

/* ---- reference constants ------------------------------------------------------------ */
#define MAT_VEC 10   /* SceneData.py:33 */
#define VER_VEC 9    /* SceneData.py:34 */
#define PRI_VEC 3    /* SceneData.py:35 */
#define SHA_VEC 10   /* SceneData.py:36 */
#define NOD_VEC 11   /* SceneData.py:37 */
#define CPN_VEC 9    /* SceneData.py:38 */
#define SHAPE_SPHERE 1          /* SceneData.py:41 */
#define SHAPE_SPOT 3            /* SceneData.py:43: emitter at a point, cone of half-angles (x1 full, x2 cut-off), never intersected */
#define SHAPE_LASER 4           /* SceneData.py:44: parallel beam of a radius along the shape normal, never intersected */
#define PRIMITIVE_TRI 1         /* SceneData.py:47 */
#define MAT_DISNEY 0            /* SceneData.py:50 */
#define MAT_GLASS 1
#define MAT_LIGHT 2
#define IS_LEAF 1               /* SceneData.py:55 */
static const float INF_VALUE = 1000000.0f;          /* UtilsFunc.py:38 */
static const float M_PIf = (float)3.1415956;        /* UtilsFunc.py:37 (sic) */
static const float PI_SCENE = (float)3.1415926;     /* Scene.py:319,343; PT_RGB.py:129-130 */

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b)
{ return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float vnorm(v3 a) { return m_sqrt(vdot(a, a)); }
/* taichi Vector.normalized(): invlen = 1/(norm+eps), eps=0; invlen * self */
static inline v3 vnormalized(v3 a) { float inv = 1.0f / vnorm(a); return vscale(a, inv); }
static inline float fabs_(float x) { return x < 0.0f ? -x : x; }
static inline float fmin_(float a, float b) { return a < b ? a : b; }
static inline float fmax_(float a, float b) { return a > b ? a : b; }
static inline float clampf(float x, float lo, float hi) { return fmin_(hi, fmax_(lo, x)); }
static inline float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

/* ---- scene handle --------------------------------------------------------------------- */
typedef struct {
    int nv, n, nm, ns, nl;
    float *vertex;      /* [nv*9]  Scene.py:37 */
    int32_t *primitive; /* [n*3]   Scene.py:42 */
    float *material;    /* [nm*10] Scene.py:36 */
    float *shape;       /* [ns*10] Scene.py:43 */
    int32_t *light;     /* [nl]    Scene.py:44 */
    int light_count;    /* Scene.light_count (0 allowed; light[] then holds one dummy) */
    float bmin[3], bmax[3];
    /* env texture, texture/Texture.py */
    int32_t *env; int env_w, env_h; float env_power;
    /* LBVH products */
    int32_t *morton;    /* [n*2] sorted (code, prim) */
    float *bvh_node;    /* [(2n-1)*11] */
    float *compact;     /* [(2n-1)*9]  */
    int node_count;
    int gen_aabb_rounds;
    /* camera, Camera.py */
    float view_inv[16]; float eye[3]; float fx, fy, cx, cy;
} orc_scene;

typedef struct {
    uint64_t rays_closest, rays_shadow;
    uint64_t box_closest, leaf_closest;     /* compact nodes popped / leaves among them */
    uint64_t box_shadow, leaf_shadow;
    uint64_t shaded;                        /* path vertices that ran the BSDF branch */
    uint64_t paths;
    uint64_t max_stack;
    uint64_t overflow;
} orc_stats;

static v3 vtx_pos(const orc_scene *s, int i) { const float *p = s->vertex + (size_t)i * VER_VEC; return V(p[0], p[1], p[2]); }
static v3 vtx_nor(const orc_scene *s, int i) { const float *p = s->vertex + (size_t)i * VER_VEC; return V(p[3], p[4], p[5]); }
static v3 vtx_uv(const orc_scene *s, int i)  { const float *p = s->vertex + (size_t)i * VER_VEC; return V(p[6], p[7], p[8]); }

orc_scene *orc_scene_create(const float *vertex, int nv, const int32_t *primitive, int n,
                            const float *material, int nm, const float *shape, int ns,
                            const int32_t *light, int nl, int light_count,
                            const float *bmin, const float *bmax)
{
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->nv = nv; s->n = n; s->nm = nm; s->ns = ns; s->nl = nl; s->light_count = light_count;
    s->vertex = (float *)malloc(sizeof(float) * (size_t)(nv > 0 ? nv : 1) * VER_VEC);
    memcpy(s->vertex, vertex, sizeof(float) * (size_t)nv * VER_VEC);
    s->primitive = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * PRI_VEC);
    memcpy(s->primitive, primitive, sizeof(int32_t) * (size_t)n * PRI_VEC);
    s->material = (float *)malloc(sizeof(float) * (size_t)nm * MAT_VEC);
    memcpy(s->material, material, sizeof(float) * (size_t)nm * MAT_VEC);
    s->shape = (float *)malloc(sizeof(float) * (size_t)ns * SHA_VEC);
    memcpy(s->shape, shape, sizeof(float) * (size_t)ns * SHA_VEC);
    s->light = (int32_t *)malloc(sizeof(int32_t) * (size_t)nl);
    memcpy(s->light, light, sizeof(int32_t) * (size_t)nl);
    for (int k = 0; k < 3; k++) { s->bmin[k] = bmin[k]; s->bmax[k] = bmax[k]; }
    s->node_count = 2 * n - 1;
    s->morton = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 2);
    s->bvh_node = (float *)malloc(sizeof(float) * (size_t)s->node_count * NOD_VEC);
    s->compact = (float *)calloc((size_t)s->node_count * CPN_VEC, sizeof(float));
    return s;
}

void orc_scene_destroy(orc_scene *s)
{
    if (!s) return;
    free(s->vertex); free(s->primitive); free(s->material); free(s->shape); free(s->light);
    free(s->env); free(s->morton); free(s->bvh_node); free(s->compact); free(s);
}

void orc_env_set(orc_scene *s, const int32_t *rgb, int w, int h, float power)
{
    free(s->env);
    s->env = (int32_t *)malloc(sizeof(int32_t) * (size_t)w * h);
    memcpy(s->env, rgb, sizeof(int32_t) * (size_t)w * h);
    s->env_w = w; s->env_h = h; s->env_power = power;
}

void orc_camera_set(orc_scene *s, const float *view_inv, const float *eye,
                    float fx, float fy, float cx, float cy)
{
    memcpy(s->view_inv, view_inv, sizeof(float) * 16);
    memcpy(s->eye, eye, sizeof(float) * 3);
    s->fx = fx; s->fy = fy; s->cx = cx; s->cy = cy;
}

/* ===================================================================================== */
/* LBVH build                                                                             */
/* ===================================================================================== */

/* UtilsFunc.py:538-552 */
static int32_t expand_bits(int32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
/* UtilsFunc.py:568-580 */
static int32_t morton3d(float x, float y, float z)
{
    x = fmin_(fmax_(x * 1024.0f, 0.0f), 1023.0f);
    y = fmin_(fmax_(y * 1024.0f, 0.0f), 1023.0f);
    z = fmin_(fmax_(z * 1024.0f, 0.0f), 1023.0f);
    int32_t xx = expand_bits((int32_t)x), yy = expand_bits((int32_t)y), zz = expand_bits((int32_t)z);
    return xx | (yy << 1) | (zz << 2);
}
/* UtilsFunc.py:555-566 */
static int common_upper_bits(int32_t lhs, int32_t rhs)
{
    int32_t x = lhs ^ rhs; int ret = 32;
    while (x > 0) { x >>= 1; ret -= 1; }
    return ret;
}

/* accel/LBvh.py:318-336 */
static void build_morton_3d(orc_scene *s)
{
    const float third = (float)(1.0 / 3.0);
    for (int i = 0; i < s->n; i++) {
        const int32_t *pr = s->primitive + (size_t)i * PRI_VEC;
        int32_t code;
        if (pr[0] == PRIMITIVE_TRI) {
            v3 v0 = vtx_pos(s, pr[1]), v1 = vtx_pos(s, pr[1] + 1), v2 = vtx_pos(s, pr[1] + 2);
            v3 c = vscale(vadd(vadd(v1, v2), v0), third);
            v3 mn = V(s->bmin[0], s->bmin[1], s->bmin[2]), mx = V(s->bmax[0], s->bmax[1], s->bmax[2]);
            v3 num = vsub(c, mn), den = vsub(mx, mn);
            code = morton3d(num.x / den.x, num.y / den.y, num.z / den.z);
        } else {
            /* quirk B7: get_vertex_pos(shape, id) = shape row words 0..2 = (type, pos.x, pos.y) */
            const float *sh = s->shape + (size_t)pr[1] * SHA_VEC;
            code = morton3d(sh[0], sh[1], sh[2]);
        }
        s->morton[2 * i] = code; s->morton[2 * i + 1] = i;
    }
}

/* accel/LBvh.py:55-72,339-386: 30 passes of a 1-bit stable split.  The Blelloch
 * up/down-sweep computes an exclusive scan of the (is_zero, is_one) flags; a sequential
 * exclusive scan is the same function. */
static void radix_sort_host(orc_scene *s)
{
    int n = s->n;
    int32_t *d = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 2);
    for (int bit = 0; bit < 30; bit++) {
        int32_t mask = 1 << bit;
        int count_zero = 0;
        for (int i = 0; i < n; i++) if (((s->morton[2 * i] & mask) >> bit) == 0) count_zero++;
        int off0 = 0, off1 = 0;
        for (int i = 0; i < n; i++) {
            int one = (s->morton[2 * i] & mask) >> bit;
            int dst = one ? (off1 + count_zero) : off0;
            d[2 * dst] = s->morton[2 * i]; d[2 * dst + 1] = s->morton[2 * i + 1];
            if (one) off1++; else off0++;
        }
        memcpy(s->morton, d, sizeof(int32_t) * (size_t)n * 2);
    }
    free(d);
}

static inline int32_t mcode(const orc_scene *s, int i) { return s->morton[2 * i]; }

/* accel/LBvh.py:229-294 */
static void determine_range(const orc_scene *s, int idx, int *lo, int *hi)
{
    int n = s->n;
    *lo = 0; *hi = n - 1;
    if (idx != 0) {
        int32_t self_code = mcode(s, idx);
        int32_t l_code = mcode(s, idx - 1), r_code = mcode(s, idx + 1);
        if (l_code == self_code && r_code == self_code) {
            *lo = idx;
            while (idx < n - 1) {
                idx += 1;
                if (idx >= n - 1) break;
                if (mcode(s, idx) != mcode(s, idx + 1)) break;
            }
            *hi = idx;
        } else {
            int L_delta = common_upper_bits(self_code, l_code);
            int R_delta = common_upper_bits(self_code, r_code);
            int d = -1;
            if (R_delta > L_delta) d = 1;
            int delta_min = L_delta < R_delta ? L_delta : R_delta;
            int l_max = 2, delta = -1;
            int i_tmp = idx + d * l_max;
            if (0 <= i_tmp && i_tmp < n) delta = common_upper_bits(self_code, mcode(s, i_tmp));
            while (delta > delta_min) {
                l_max <<= 1;
                i_tmp = idx + d * l_max;
                delta = -1;
                if (0 <= i_tmp && i_tmp < n) delta = common_upper_bits(self_code, mcode(s, i_tmp));
            }
            int l = 0, t = l_max >> 1;
            while (t > 0) {
                i_tmp = idx + (l + t) * d;
                delta = -1;
                if (0 <= i_tmp && i_tmp < n) delta = common_upper_bits(self_code, mcode(s, i_tmp));
                if (delta > delta_min) l += t;
                t >>= 1;
            }
            *lo = idx; *hi = idx + l * d;
            if (d < 0) { int tmp = *lo; *lo = *hi; *hi = tmp; }
        }
    }
}

/* accel/LBvh.py:296-314 */
static int find_split(const orc_scene *s, int first, int last)
{
    int32_t first_code = mcode(s, first), last_code = mcode(s, last);
    int split = first;
    if (first_code != last_code) {
        int delta_node = common_upper_bits(first_code, last_code);
        int stride = last - first;
        for (;;) {
            stride = (stride + 1) >> 1;
            int middle = split + stride;
            if (middle < last) {
                int delta = common_upper_bits(first_code, mcode(s, middle));
                if (delta > delta_node) split = middle;
            }
            if (stride <= 1) break;
        }
    }
    return split;
}

/* UtilsFunc.py:232-243: set_node_type / set_node_prim_size operate on float(int(x) & mask) */
static float node_flag_and(float cur, int mask) { return (float)(((int)cur) & mask); }

/* accel/LBvh.py:389-450 */
static void build_lbvh(orc_scene *s)
{
    int n = s->n, N = s->node_count;
    for (int i = 0; i < N; i++) {               /* UtilsFunc.py:219-231 */
        float *nd = s->bvh_node + (size_t)i * NOD_VEC;
        nd[0] = nd[1] = nd[2] = nd[3] = nd[4] = -1.0f;
        nd[5] = nd[6] = nd[7] = INF_VALUE;
        nd[8] = nd[9] = nd[10] = -INF_VALUE;
    }
    /* pass 1: everything except parent links (the reference writes parents of other nodes
     * from internal-node threads; init ran in a previous offload so order is immaterial) */
    for (int i = 0; i < N; i++) {
        float *nd = s->bvh_node + (size_t)i * NOD_VEC;
        if (i >= n - 1) {
            nd[0] = node_flag_and(nd[0], 0xfffe | IS_LEAF);     /* set_node_type  */
            nd[0] = node_flag_and(nd[0], 0x0007 | 1);           /* set_node_prim_size */
            int prim = s->morton[2 * (i - n + 1) + 1];
            nd[4] = (float)prim;
            const int32_t *pr = s->primitive + (size_t)prim * PRI_VEC;
            v3 mn = V(0, 0, 0), mx = V(0, 0, 0);
            if (pr[0] == PRIMITIVE_TRI) {
                v3 v1 = vtx_pos(s, pr[1]), v2 = vtx_pos(s, pr[1] + 1), v3_ = vtx_pos(s, pr[1] + 2);
                mn = v1; mx = v1;
                mn.x = fmin_(mn.x, v2.x); mn.x = fmin_(mn.x, v3_.x); mx.x = fmax_(mx.x, v2.x); mx.x = fmax_(mx.x, v3_.x);
                mn.y = fmin_(mn.y, v2.y); mn.y = fmin_(mn.y, v3_.y); mx.y = fmax_(mx.y, v2.y); mx.y = fmax_(mx.y, v3_.y);
                mn.z = fmin_(mn.z, v2.z); mn.z = fmin_(mn.z, v3_.z); mx.z = fmax_(mx.z, v2.z); mx.z = fmax_(mx.z, v3_.z);
            } else {
                const float *sh = s->shape + (size_t)pr[1] * SHA_VEC;
                if ((int)sh[0] == SHAPE_SPHERE) {
                    float r = sh[4];
                    mn = V(sh[1] + -r, sh[2] + -r, sh[3] + -r);
                    mx = V(sh[1] + r, sh[2] + r, sh[3] + r);
                }
            }
            nd[5] = mn.x; nd[6] = mn.y; nd[7] = mn.z; nd[8] = mx.x; nd[9] = mx.y; nd[10] = mx.z;
        } else {
            nd[0] = node_flag_and(nd[0], 0xfffe | (1 - IS_LEAF));
            int lo, hi;
            determine_range(s, i, &lo, &hi);
            int split = find_split(s, lo, hi);
            int left = split, right = split + 1;
            if ((lo < hi ? lo : hi) == split) left += n - 1;
            if ((lo > hi ? lo : hi) == split + 1) right += n - 1;
            nd[1] = (float)left; nd[2] = (float)right;
        }
    }
    for (int i = 0; i < n - 1; i++) {
        const float *nd = s->bvh_node + (size_t)i * NOD_VEC;
        int left = (int)nd[1], right = (int)nd[2];
        if (left >= 0 && left < N) s->bvh_node[(size_t)left * NOD_VEC + 3] = (float)i;
        if (right >= 0 && right < N) s->bvh_node[(size_t)right * NOD_VEC + 3] = (float)i;
    }
}

static int node_has_box(const float *nd)   /* UtilsFunc.py:287-289 */
{ return (nd[5] <= nd[8]) & (nd[6] <= nd[9]) & (nd[7] <= nd[10]); }

/* accel/LBvh.py:453-467 + host loop :206-218 */
static int gen_aabb_all(orc_scene *s)
{
    int n = s->n, N = s->node_count;
    int done = 0, done_prev = 0, rounds = 0;
    while (done < n - 1) {
        for (int i = 0; i < N; i++) {
            float *nd = s->bvh_node + (size_t)i * NOD_VEC;
            if (!node_has_box(nd)) {
                int l = (int)nd[1], r = (int)nd[2];
                if (l < 0 || r < 0 || l >= N || r >= N) continue;
                const float *ln = s->bvh_node + (size_t)l * NOD_VEC, *rn = s->bvh_node + (size_t)r * NOD_VEC;
                if (node_has_box(ln) & node_has_box(rn)) {
                    for (int k = 0; k < 3; k++) {
                        nd[5 + k] = fmin_(ln[5 + k], rn[5 + k]);
                        nd[8 + k] = fmax_(ln[8 + k], rn[8 + k]);
                    }
                    done += 1;
                }
            }
        }
        rounds++;
        if (done == done_prev) break;
        done_prev = done;
    }
    s->gen_aabb_rounds = rounds;
    return done;
}

/* accel/LBvh.py:138-173 (recursive DFS, left first; slot1 of an internal node = offset
 * returned by the right child's call).  Iterative to keep the C stack flat. */
static void flatten_tree(orc_scene *s)
{
    int N = s->node_count;
    int *stk_node = (int *)malloc(sizeof(int) * (size_t)(N + 1));
    int *stk_parent_off = (int *)malloc(sizeof(int) * (size_t)(N + 1));
    int sp = 0, offset = 0;
    stk_node[sp] = 0; stk_parent_off[sp] = -1; sp++;
    while (sp > 0) {
        sp--;
        int index = stk_node[sp], parent_off = stk_parent_off[sp];
        int ret_off = offset++;
        const float *nd = s->bvh_node + (size_t)index * NOD_VEC;
        float *cn = s->compact + (size_t)ret_off * CPN_VEC;
        if (parent_off >= 0) s->compact[(size_t)parent_off * CPN_VEC + 1] = (float)ret_off;
        int is_leaf = ((int)nd[0]) & 1;
        cn[0] = nd[0];
        for (int k = 0; k < 6; k++) cn[2 + k] = nd[5 + k];
        if (is_leaf != IS_LEAF) {
            /* visit left now (implicit at ret_off+1), right afterwards; right's offset goes in our slot 1 */
            stk_node[sp] = (int)nd[2]; stk_parent_off[sp] = ret_off; sp++;
            stk_node[sp] = (int)nd[1]; stk_parent_off[sp] = -1; sp++;
        } else {
            cn[1] = nd[4];
        }
    }
    free(stk_node); free(stk_parent_off);
}

/* accel/LBvh.py:192-226; returns number of internal nodes that received a box (== n-1 on success) */
int orc_lbvh_build(orc_scene *s)
{
    build_morton_3d(s);
    radix_sort_host(s);
    build_lbvh(s);
    int done = gen_aabb_all(s);
    memset(s->compact, 0, sizeof(float) * (size_t)s->node_count * CPN_VEC);
    flatten_tree(s);
    return done;
}

void orc_lbvh_get(const orc_scene *s, int32_t *morton_sorted, float *bvh_node, float *compact)
{
    if (morton_sorted) memcpy(morton_sorted, s->morton, sizeof(int32_t) * (size_t)s->n * 2);
    if (bvh_node) memcpy(bvh_node, s->bvh_node, sizeof(float) * (size_t)s->node_count * NOD_VEC);
    if (compact) memcpy(compact, s->compact, sizeof(float) * (size_t)s->node_count * CPN_VEC);
}
/* unsorted Morton codes for kernel-level KATs */
void orc_morton_codes(orc_scene *s, int32_t *out)
{
    int32_t *save = (int32_t *)malloc(sizeof(int32_t) * (size_t)s->n * 2);
    memcpy(save, s->morton, sizeof(int32_t) * (size_t)s->n * 2);
    build_morton_3d(s);
    memcpy(out, s->morton, sizeof(int32_t) * (size_t)s->n * 2);
    memcpy(s->morton, save, sizeof(int32_t) * (size_t)s->n * 2);
    free(save);
}
int orc_gen_aabb_rounds(const orc_scene *s) { return s->gen_aabb_rounds; }
void orc_vertex_get(const orc_scene *s, float *vertex) { memcpy(vertex, s->vertex, sizeof(float) * (size_t)s->nv * VER_VEC); }

/* ===================================================================================== */
/* Traversal + intersection                                                               */
/* ===================================================================================== */

/* UtilsFunc.py:494-523 */
static int slabs(v3 o, v3 d, v3 mn, v3 mx)
{
    int ret = 1;
    float tmin = 0.0f, tmax = INF_VALUE;
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    const float mi[3] = { mn.x, mn.y, mn.z }, ma[3] = { mx.x, mx.y, mx.z };
    for (int i = 0; i < 3; i++) {
        if (fabs_(dd[i]) < 0.000001f) {
            if ((oo[i] < mi[i]) | (oo[i] > ma[i])) ret = 0;
        } else {
            float ood = 1.0f / dd[i];
            float t1 = (mi[i] - oo[i]) * ood;
            float t2 = (ma[i] - oo[i]) * ood;
            if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
            if (t1 > tmin) tmin = t1;
            if (t2 < tmax) tmax = t2;
            if (tmin > tmax) ret = 0;
        }
    }
    return ret;
}

/* Scene.py:603-638 */
static float intersect_tri(const orc_scene *s, v3 origin, v3 direction, int prim, float *uo, float *vo)
{
    float t = INF_VALUE, u = 0.0f, v = 0.0f;
    int vid = s->primitive[(size_t)prim * PRI_VEC + 1];
    v3 v0 = vtx_pos(s, vid), v1 = vtx_pos(s, vid + 1), v2 = vtx_pos(s, vid + 2);
    v3 E1 = vsub(v1, v0), E2 = vsub(v2, v0);
    v3 P = vcross(direction, E2);
    float det = vdot(E1, P);
    v3 T;
    if (det > 0.0f) T = vsub(origin, v0);
    else { T = vsub(v0, origin); det = -det; }
    if (det > 0.0f) {
        u = vdot(T, P);
        if ((u >= 0.0f) & (u <= det)) {
            v3 Q = vcross(T, E1);
            v = vdot(direction, Q);
            if ((v >= 0.0f) & (u + v <= det)) {
                t = vdot(E2, Q);
                float fInvDet = 1.0f / det;
                t *= fInvDet; u *= fInvDet; v *= fInvDet;
            }
        }
    }
    *uo = u; *vo = v;
    return t;
}

typedef struct { float t; v3 pos, gnor, nor, tex; int prim; } hit_t;

/* Scene.py:529-600 */
static float intersect_prim(const orc_scene *s, v3 origin, v3 direction, int prim,
                            v3 *hit_pos, v3 *hit_gnor, v3 *hit_nor, v3 *hit_tex)
{
    const int32_t *pr = s->primitive + (size_t)prim * PRI_VEC;
    float hit_tv = INF_VALUE;
    v3 pos = V(0, 0, 0), nor = V(0, 0, 0), tex = V(0, 0, 0), gnor = V(0, 0, 0);
    if (pr[0] == PRIMITIVE_TRI) {
        float u, v;
        hit_tv = intersect_tri(s, origin, direction, prim, &u, &v);
        if (hit_tv < INF_VALUE) {
            int vi = pr[1];
            float a = 1.0f - u - v, b = u, c = v;
            v3 v1 = vtx_pos(s, vi), v2 = vtx_pos(s, vi + 1), v3_ = vtx_pos(s, vi + 2);
            v3 n1 = vtx_nor(s, vi), n2 = vtx_nor(s, vi + 1), n3 = vtx_nor(s, vi + 2);
            v3 t1 = vtx_uv(s, vi), t2 = vtx_uv(s, vi + 1), t3 = vtx_uv(s, vi + 2);
            v3 v13 = vsub(v3_, v1), v12 = vsub(v2, v1);
            gnor = vcross(v12, v13);
            pos = vadd(vadd(vscale(v1, a), vscale(v2, b)), vscale(v3_, c));
            tex = vadd(vadd(vscale(t1, a), vscale(t2, b)), vscale(t3, c));
            nor = vadd(vadd(vscale(n1, a), vscale(n2, b)), vscale(n3, c));
        }
    } else {
        const float *sh = s->shape + (size_t)pr[1] * SHA_VEC;
        if ((int)sh[0] == SHAPE_SPHERE) {
            float r = sh[4];
            v3 centre = V(sh[1], sh[2], sh[3]);
            v3 oc = vsub(centre, origin);
            float dis_oc_square = vdot(oc, oc);
            float dis_op = vdot(direction, oc);
            float dis_cp = m_sqrt(dis_oc_square - dis_op * dis_op);
            if (dis_cp < r) {
                float a = vdot(direction, direction);
                float b = -2.0f * dis_op;
                float c = dis_oc_square - r * r;
                hit_tv = (-b - m_sqrt(b * b - 4.0f * a * c)) / 2.0f / a;
                pos = vadd(origin, vscale(direction, hit_tv));
                nor = V(pos.x - c, pos.y - c, pos.z - c);      /* quirk B3: scalar c, not centre */
                gnor = nor;
            }
        } else hit_tv = INF_VALUE;
    }
    *hit_pos = pos; *hit_gnor = vnormalized(gnor); *hit_nor = vnormalized(nor); *hit_tex = tex;
    return hit_tv;
}

/* Scene.py:642-669 */
static float intersect_prim_any(const orc_scene *s, v3 origin, v3 direction, int prim)
{
    const int32_t *pr = s->primitive + (size_t)prim * PRI_VEC;
    float hit_tv = INF_VALUE;
    if (pr[0] == PRIMITIVE_TRI) {
        float u, v;
        hit_tv = intersect_tri(s, origin, direction, prim, &u, &v);
    } else {
        const float *sh = s->shape + (size_t)pr[1] * SHA_VEC;
        if ((int)sh[0] == SHAPE_SPHERE) {
            float r = sh[4];
            v3 centre = V(sh[1], sh[2], sh[3]);
            v3 oc = vsub(centre, origin);
            float dis_oc_square = vdot(oc, oc);
            float dis_op = vdot(direction, oc);
            float dis_cp = m_sqrt(dis_oc_square - dis_op * dis_op);
            if (dis_cp < r) {
                float a = vdot(direction, direction);
                float b = -2.0f * dis_op;
                float c = dis_oc_square - r * r;
                hit_tv = (-b - m_sqrt(b * b - 4.0f * a * c)) / 2.0f / a;
            }
        } else hit_tv = INF_VALUE;
    }
    return hit_tv;
}

static inline int cn_is_leaf(const float *cn) { return ((int)cn[0]) & 1; }

/* Scene.py:702-744.  stack has max_size+2 ints (quirk B11: a push may write index MAX). */
static hit_t closet_hit(const orc_scene *s, v3 origin, v3 direction, int32_t *stack, int max_size,
                        orc_stats *st)
{
    hit_t h; h.t = INF_VALUE; h.pos = h.nor = h.gnor = h.tex = V(0, 0, 0); h.prim = -1;
    stack[0] = 0;
    int stack_pos = 0;
    uint64_t nbox = 0, nleaf = 0, maxs = 0;
    while ((stack_pos >= 0) & (stack_pos < max_size)) {
        int node = stack[stack_pos];
        stack_pos -= 1;
        const float *cn = s->compact + (size_t)node * CPN_VEC;
        nbox++;
        if (cn_is_leaf(cn) == IS_LEAF) {
            nleaf++;
            int prim = (int)cn[1];
            v3 pos, gn, nn, tx;
            float t = intersect_prim(s, origin, direction, prim, &pos, &gn, &nn, &tx);
            if ((t < h.t) & (t > 0.0f)) { h.t = t; h.pos = pos; h.nor = nn; h.gnor = gn; h.tex = tx; h.prim = prim; }
        } else {
            if (slabs(origin, direction, V(cn[2], cn[3], cn[4]), V(cn[5], cn[6], cn[7])) == 1) {
                stack_pos += 1; stack[stack_pos] = node + 1;
                stack_pos += 1; stack[stack_pos] = (int)cn[1];
                if ((uint64_t)stack_pos > maxs) maxs = (uint64_t)stack_pos;
            }
        }
    }
    if (st) {
        st->rays_closest++; st->box_closest += nbox; st->leaf_closest += nleaf;
        if (maxs > st->max_stack) st->max_stack = maxs;
        if (stack_pos == max_size) st->overflow++;
    }
    return h;
}

/* Scene.py:671-699 */
static float closet_hit_shadow(const orc_scene *s, v3 origin, v3 direction, int32_t *stack, int max_size,
                               int *hit_prim_out, orc_stats *st)
{
    float hit_tv = INF_VALUE; int hit_prim = -1;
    stack[0] = 0;
    int stack_pos = 0;
    uint64_t nbox = 0, nleaf = 0, maxs = 0;
    while ((stack_pos >= 0) & (stack_pos < max_size)) {
        int node = stack[stack_pos];
        stack_pos -= 1;
        const float *cn = s->compact + (size_t)node * CPN_VEC;
        nbox++;
        if (cn_is_leaf(cn) == IS_LEAF) {
            nleaf++;
            int prim = (int)cn[1];
            float t = intersect_prim_any(s, origin, direction, prim);
            if ((t < hit_tv) & (t > 0.0f)) { hit_tv = t; hit_prim = prim; }
        } else {
            if (slabs(origin, direction, V(cn[2], cn[3], cn[4]), V(cn[5], cn[6], cn[7])) == 1) {
                stack_pos += 1; stack[stack_pos] = node + 1;
                stack_pos += 1; stack[stack_pos] = (int)cn[1];
                if ((uint64_t)stack_pos > maxs) maxs = (uint64_t)stack_pos;
            }
        }
    }
    if (st) {
        st->rays_shadow++; st->box_shadow += nbox; st->leaf_shadow += nleaf;
        if (maxs > st->max_stack) st->max_stack = maxs;
        if (stack_pos == max_size) st->overflow++;
    }
    *hit_prim_out = hit_prim;
    return hit_tv;
}

/* Batch entry points for kernel-level parity tests.
 * rays: [nr*6] (origin, direction); out_f: [nr*13] = t, pos3, gnormal3, normal3, tex3;
 * out_prim: [nr]; counts: [nr*2] = (N_box, N_leaf) per ray (nullable). */
void orc_closest_hit_batch(const orc_scene *s, const float *rays, int nr, int max_size,
                           float *out_f, int32_t *out_prim, int32_t *counts)
{
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_size + 2));
    for (int r = 0; r < nr; r++) {
        const float *q = rays + (size_t)r * 6;
        orc_stats st; memset(&st, 0, sizeof(st));
        hit_t h = closet_hit(s, V(q[0], q[1], q[2]), V(q[3], q[4], q[5]), stack, max_size, &st);
        float *o = out_f + (size_t)r * 13;
        o[0] = h.t; o[1] = h.pos.x; o[2] = h.pos.y; o[3] = h.pos.z;
        o[4] = h.gnor.x; o[5] = h.gnor.y; o[6] = h.gnor.z;
        o[7] = h.nor.x; o[8] = h.nor.y; o[9] = h.nor.z;
        o[10] = h.tex.x; o[11] = h.tex.y; o[12] = h.tex.z;
        out_prim[r] = h.prim;
        if (counts) { counts[2 * r] = (int32_t)st.box_closest; counts[2 * r + 1] = (int32_t)st.leaf_closest; }
    }
    free(stack);
}
void orc_shadow_hit_batch(const orc_scene *s, const float *rays, int nr, int max_size,
                          float *out_t, int32_t *out_prim, int32_t *counts)
{
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_size + 2));
    for (int r = 0; r < nr; r++) {
        const float *q = rays + (size_t)r * 6;
        orc_stats st; memset(&st, 0, sizeof(st));
        int prim;
        out_t[r] = closet_hit_shadow(s, V(q[0], q[1], q[2]), V(q[3], q[4], q[5]), stack, max_size, &prim, &st);
        out_prim[r] = prim;
        if (counts) { counts[2 * r] = (int32_t)st.box_shadow; counts[2 * r + 1] = (int32_t)st.leaf_shadow; }
    }
    free(stack);
}

/* ===================================================================================== */
/* Sampling, materials                                                                    */
/* ===================================================================================== */

/* UtilsFunc.py:76-84 */
static v3 srgb_to_lrgb(v3 c)
{
    float in[3] = { c.x, c.y, c.z }, out[3];
    for (int i = 0; i < 3; i++) {
        if (in[i] < 0.04045f) out[i] = in[i] / 12.92f;
        else out[i] = m_pow((in[i] + 0.055f) / 1.055f, 2.4f);
    }
    return V(out[0], out[1], out[2]);
}
/* UtilsFunc.py:86-94 */
static v3 lrgb_to_srgb(v3 c)
{
    float in[3] = { c.x, c.y, c.z }, out[3];
    const float e = (float)(1.0 / 2.4);
    for (int i = 0; i < 3; i++) {
        if (in[i] < 0.0031308f) out[i] = in[i] * 12.92f;
        else out[i] = 1.055f * m_pow(in[i], e) - 0.055f;
        out[i] = clampf(out[i], 0.0f, 1.0f);
    }
    return V(out[0], out[1], out[2]);
}
/* UtilsFunc.py:113-120 */
static float tone_aces1(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return clampf((x * (a * x + b)) / (x * (c * x + d) + e), 0.0f, 1.0f);
}

/* UtilsFunc.py:352-360 */
static v3 cosine_sample_hemisphere(float u1, float u2)
{
    const float two_pi = (float)(2.0 * 3.1415956);
    float r = m_sqrt(u1);
    float phi = two_pi * u2;
    v3 p;
    p.x = r * m_cos(phi);
    p.y = r * m_sin(phi);
    p.z = m_sqrt(fmax_(0.0f, 1.0f - p.x * p.x - p.y * p.y));
    return vnormalized(p);
}

/* UtilsFunc.py:373-387 */
static v3 inverse_transform(v3 dir, v3 N)
{
    v3 Normal = vnormalized(N);
    v3 Binormal;
    if (fabs_(Normal.x) > fabs_(Normal.z)) Binormal = V(-Normal.y, Normal.x, 0.0f);
    else Binormal = V(0.0f, -Normal.z, Normal.y);
    Binormal = vnormalized(Binormal);
    v3 Tangent = vnormalized(vcross(Binormal, Normal));
    return vadd(vadd(vscale(Tangent, dir.x), vscale(Binormal, dir.y)), vscale(Normal, dir.z));
}

/* UtilsFunc.py:393-415 */
static float schlick_fresnel(float u) { float m = clampf(1.0f - u, 0.0f, 1.0f); float m2 = m * m; return m2 * m2 * m; }
static float gtr2(float NDotH, float a) { float a2 = a * a; float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH; return a2 / (M_PIf * t * t); }
static float smithg_ggx(float NDotv, float alphaG) { float a = alphaG * alphaG, b = NDotv * NDotv; return 1.0f / (NDotv + m_sqrt(a + b - a * b)); }

/* UtilsFunc.py:417-432 */
static v3 refract_(v3 InRay, v3 N, float eta, float *suc)
{
    *suc = -1.0f;
    float N_DOT_I = vdot(N, InRay);
    float k = 1.0f - eta * eta * (1.0f - N_DOT_I * N_DOT_I);
    v3 R = V(0, 0, 0);
    if (k > 0.0f) {
        R = vsub(vscale(InRay, eta), vscale(N, eta * N_DOT_I + m_sqrt(k)));
        *suc = 1.0f;
    }
    return R;
}
static float schlick(float cosine, float ior)
{
    float r0 = (1.0f - ior) / (1.0f + ior);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * m_pow(1.0f - cosine, 5.0f);
}
static v3 reflect_(v3 I, v3 N) { return vsub(I, vscale(N, 2.0f * vdot(N, I))); }   /* taichi_glsl reflect */

/* UtilsFunc.py:435-438 */
static float power_heuristic(float a, float b) { float t = a * a; return t / (b * b + t); }

/* UtilsFunc.py:440-461 */
static v3 offset_ray(v3 p, v3 n)
{
    const float int_scale = 256.0f, float_scale = (float)(1.0 / 2048.0), origin = (float)(1.0 / 256.0);
    float pp[3] = { p.x, p.y, p.z }, nn[3] = { n.x, n.y, n.z }, ret[3];
    for (int k = 0; k < 3; k++) {
        int32_t i_of = (int32_t)(int_scale * nn[k]);
        int32_t i_p = (int32_t)tm_f2u(pp[k]);
        if (pp[k] < 0.0f) i_p = i_p - i_of; else i_p = i_p + i_of;
        float f_p = tm_u2f((uint32_t)i_p);
        if (fabs_(pp[k]) < origin) ret[k] = pp[k] + float_scale * nn[k];
        else ret[k] = f_p;
    }
    return V(ret[0], ret[1], ret[2]);
}

/* brdf/Disney.py:17-40; rnd[3] = (probability, r1, r2) */
static v3 disney_sample(const orc_scene *s, v3 dir, v3 N, int mat_id, const float *rnd)
{
    const float *m = s->material + (size_t)mat_id * MAT_VEC;
    float metal = m[5], rough = m[6];
    float diffuseRatio = 0.5f * (1.0f - metal);
    float specularAlpha = fmax_(0.001f, rough);
    float probability = rnd[0], r1 = rnd[1], r2 = rnd[2];
    v3 next_dir;
    if (probability < diffuseRatio) {
        next_dir = cosine_sample_hemisphere(r1, r2);
        next_dir = inverse_transform(next_dir, N);
    } else {
        float phi = r1 * 2.0f * M_PIf;
        float cosTheta = m_sqrt((1.0f - r2) / (1.0f + (specularAlpha * specularAlpha - 1.0f) * r2));
        float sinTheta = m_sqrt(1.0f - (cosTheta * cosTheta));
        float sinPhi = m_sin(phi), cosPhi = m_cos(phi);
        v3 half = V(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
        half = inverse_transform(half, N);
        next_dir = reflect_(dir, half);
    }
    return next_dir;
}

/* brdf/Disney.py:65-108 */
static float disney_evaluate_pdf(const orc_scene *s, v3 N, v3 Vv, v3 L, int mat_id, float *pdf_out)
{
    float outputC = 0.0f, pdf = -1.0f;
    float NDotL = vdot(N, L), NDotV = vdot(N, Vv);
    if ((NDotL > 0.0f) & (NDotV > 0.0f)) {
        const float inv_pi = (float)(1.0 / 3.1415956);
        const float *m = s->material + (size_t)mat_id * MAT_VEC;
        v3 H = vnormalized(vadd(L, Vv));
        float NDotH = vdot(H, N), LDotH = vdot(H, L);
        float metal = m[5], rough = m[6];
        float Cspec0 = mixf(0.04f, 1.0f, metal);
        float Csheen = 0.5f;
        float FL = schlick_fresnel(NDotL), FV = schlick_fresnel(NDotV);
        float Fd90 = 0.5f + 2.0f * LDotH * LDotH * rough;
        float Fd = mixf(1.0f, Fd90, FL) * mixf(1.0f, Fd90, FV);
        float specularAlpha = fmax_(0.001f, rough);
        float Ds = gtr2(NDotH, specularAlpha);
        float FH = schlick_fresnel(LDotH);
        float Fs = mixf(Cspec0, 1.0f, FH);
        float rg = rough * 0.5f + 0.5f; float roughg = rg * rg;
        float Gs = smithg_ggx(NDotL, roughg) * smithg_ggx(NDotV, roughg);
        float Fsheen = FH * Csheen;
        outputC = (Fsheen + inv_pi) * Fd * (1.0f - metal) + Gs * Fs * Ds;
        float diffuseRatio = 0.5f * (1.0f - metal);
        float specularRatio = 1.0f - diffuseRatio;
        float pdfGTR2 = Ds * NDotH;
        float pdfSpec = pdfGTR2 / (4.0f * fabs_(LDotH));
        float pdfDiff = inv_pi;                       /* quirk B4, brdf/Disney.py:12-15 */
        pdf = diffuseRatio * pdfDiff + specularRatio * pdfSpec;
    }
    *pdf_out = pdf;
    return outputC;
}

/* brdf/Glass.py:9-34 */
static v3 glass_sample(const orc_scene *s, v3 dir, v3 N, int mat_id, float probability, float *f_or_b)
{
    const float *m = s->material + (size_t)mat_id * MAT_VEC;
    v3 w_out = dir;
    float cos_theta_i = vdot(w_out, N);
    float ior = m[5];
    float eta = ior;
    *f_or_b = 1.0f;
    float R = probability + 1.0f;
    if (cos_theta_i > 0.0f) N = vneg(N);
    else { cos_theta_i = -cos_theta_i; eta = 1.0f / ior; }
    float suc;
    v3 next_dir = refract_(w_out, N, eta, &suc);
    if (suc > 0.0f) R = schlick(cos_theta_i, ior);
    if (probability < R) next_dir = reflect_(w_out, N);
    else *f_or_b = -1.0f;
    return next_dir;
}

/* Scene.py:324-350 */
static float get_prim_area(const orc_scene *s, int index)
{
    float ret = 0.0f;
    const int32_t *pr = s->primitive + (size_t)index * PRI_VEC;
    if (pr[0] == PRIMITIVE_TRI) {
        v3 v1 = vtx_pos(s, pr[1]), v2 = vtx_pos(s, pr[1] + 1), v3_ = vtx_pos(s, pr[1] + 2);
        float a = vnorm(vsub(v1, v2)), b = vnorm(vsub(v1, v3_)), c = vnorm(vsub(v3_, v2));
        float sum = (a + b + c) * 0.5f;
        ret = m_sqrt(sum * (sum - a) * (sum - b) * (sum - c));
    } else {
        const float *sh = s->shape + (size_t)pr[1] * SHA_VEC;
        int st = (int)sh[0];
        if (st == SHAPE_SPHERE || st == 3 || st == 4) { float r = sh[4]; ret = r * r * PI_SCENE; }   /* quirk B2 */
    }
    return ret;
}

/* Scene.py:315-322 */
static v3 uniform_sample_sphere(float u1, float u2)
{
    const float two_pi = (float)(2.0 * 3.1415926);
    float z = 1.0f - 2.0f * u1;
    float r = m_sqrt(clampf(1.0f - z * z, 0.0f, 1.0f));
    float phi = two_pi * u2;
    return V(r * m_cos(phi), r * m_sin(phi), z);
}

/* Scene.py:381-420 */
static void get_prim_random_point_normal(const orc_scene *s, int index, float a, float b, v3 *pos_o, v3 *nor_o)
{
    v3 pos = V(0, 0, 0), normal = V(0, 0, 0);
    const int32_t *pr = s->primitive + (size_t)index * PRI_VEC;
    if (pr[0] == PRIMITIVE_TRI) {
        v3 v1 = vtx_pos(s, pr[1]), v2 = vtx_pos(s, pr[1] + 1), v3_ = vtx_pos(s, pr[1] + 2);
        v3 n1 = vtx_nor(s, pr[1]), n2 = vtx_nor(s, pr[1] + 1), n3 = vtx_nor(s, pr[1] + 2);
        if (a + b > 1.0f) { a = 1.0f - a; b = 1.0f - b; }
        pos = vadd(vadd(v1, vscale(vsub(v3_, v1), a)), vscale(vsub(v2, v1), b));
        normal = vnormalized(vadd(vadd(vscale(n1, 1.0f - a - b), vscale(n2, a)), vscale(n3, b)));
    } else {
        const float *sh = s->shape + (size_t)pr[1] * SHA_VEC;
        if ((int)sh[0] == SHAPE_SPHERE) {
            float r = sh[4];
            v3 centre = V(sh[1], sh[2], sh[3]);
            normal = uniform_sample_sphere(a, b);
            pos = vadd(centre, vscale(normal, r));
        } else if ((int)sh[0] == SHAPE_SPOT || (int)sh[0] == SHAPE_LASER) {      /* Scene.py:413-418 */
            normal = V(sh[7], sh[8], sh[9]);
            pos = V(sh[1], sh[2], sh[3]);
        }
    }
    *pos_o = pos; *nor_o = vnormalized(normal);
}

/* Scene.py:491-516: what sample_li adds for the two shape emitters that have no surface -- the factor `visable` on the emission
 * (spot: 1 inside the cone of half-angle x1, falling linearly to 0 at x2, measured between the light's normal and the direction to
 * the shaded point; laser: 1 within `radius` of the beam's axis, else 0) and, for the laser, light_choice_pdf = 1 / light_count. */
static float light_shape_visible(const orc_scene *s, int light_prim, v3 light_dir, v3 light_normal, float light_dist, float *choice_pdf)
{
    float visable = 1.0f;
    const int32_t *pr = s->primitive + (size_t)light_prim * PRI_VEC;
    if (pr[0] != PRIMITIVE_TRI) {
        const float *sh = s->shape + (size_t)pr[1] * SHA_VEC;
        const int st = (int)sh[0];
        if (st == SHAPE_SPOT) {
            const float NdotL = fabs_(vdot(light_dir, light_normal));
            const float x1 = sh[4], x2 = sh[5];
            const float x = m_acos(NdotL);
            if (x > x2) visable = 0.0f;
            else if (x > x1) visable *= 1.0f - (x - x1) / (x2 - x1);
        } else if (st == SHAPE_LASER) {
            *choice_pdf = 1.0f / (float)s->light_count;
            const float proj = vdot(light_dir, light_normal) * light_dist;
            const float r = m_sqrt(light_dist * light_dist - proj * proj);
            if (r > sh[4]) visable = 0.0f;
        }
    }
    return visable;
}

/* UtilsFunc.py:321-345 */
static void map_to_disk(float u1, float u2, float *r_o, float *phi_o)
{
    float phi = 0.0f, r = 0.0f;
    const float a = 2.0f * u1 - 1.0f, b = 2.0f * u2 - 1.0f;
    if (a > -b) {
        if (a > b) { r = a; phi = (M_PIf / 4.0f) * (b / a); }
        else { r = b; phi = (M_PIf / 4.0f) * (2.0f - a / b); }
    } else {
        if (a < b) { r = -a; phi = (M_PIf / 4.0f) * (4.0f + b / a); }
        else { r = -b; phi = (b == 0.0f) ? 0.0f : (M_PIf / 4.0f) * (6.0f - a / b); }
    }
    *r_o = r; *phi_o = phi;
}

/* texture/Texture.py:41-69 */
static v3 tex_sample(const orc_scene *s, float fx, float fy)
{
    int x = (int)fx, y = (int)fy;
    x = x < 0 ? 0 : (x > s->env_w - 1 ? s->env_w - 1 : x);
    y = y < 0 ? 0 : (y > s->env_h - 1 ? s->env_h - 1 : y);
    int32_t RGBA = s->env[(size_t)x * s->env_h + y];
    float R = (float)((RGBA & 0x00FF0000) >> 16) / 255.0f;
    float G = (float)((RGBA & 0x0000FF00) >> 8) / 255.0f;
    float B = (float)(RGBA & 0x000000FF) / 255.0f;
    return V(R, G, B);
}
static v3 vmix(v3 a, v3 b, float t) { return V(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)); }
static v3 texture2d(const orc_scene *s, float u, float v)
{
    float x = clampf(u * (float)s->env_w, 0.0f, (float)s->env_w - 1.0f);
    float y = clampf(v * (float)s->env_h, 0.0f, (float)s->env_h - 1.0f);
    float lx = tm_floor(x), ly = tm_floor(y);
    float wbt = y - tm_floor(y), wlr = x - tm_floor(x);
    v3 lt = tex_sample(s, lx, ly), rt = tex_sample(s, lx + 1.0f, ly);
    v3 lb = tex_sample(s, lx, ly + 1.0f), rb = tex_sample(s, lx + 1.0f, ly + 1.0f);
    return vmix(vmix(lt, rt, wlr), vmix(lb, rb, wlr), wbt);
}

/* ===================================================================================== */
/* PT_RGB.render                                                                          */
/* ===================================================================================== */

#define PATH_MAX_DEPTH_DEFAULT 15   /* integrator/PT_RGB.py:21 */

/* integrator/PT_RGB.py:49-132 for one pixel (i, j) at one frame; returns radiance */
static v3 pt_rgb_pixel(const orc_scene *s, int i, int j, int H, uint32_t frame, uint32_t seed,
                       int max_depth, int32_t *stack, int stack_size, orc_stats *st)
{
    uint32_t pixel = (uint32_t)(i * H + j);
    /* Camera.py:122-142 */
    v3 next_origin = V(s->eye[0], s->eye[1], s->eye[2]);
    float jx = 0.0f, jy = 0.0f;
    if (frame != 0) {
        jx = tm_rand(seed, pixel, frame, TM_DIM_JX) - 0.5f;
        jy = tm_rand(seed, pixel, frame, TM_DIM_JY) - 0.5f;
    }
    v3 next_dir;
    {
        float x = ((float)i + jx - s->cx) / s->fx;
        float y = ((float)j + jy - s->cy) / s->fy;
        float z = -1.0f, w = 0.0f;
        const float *M = s->view_inv;
        float wx = ((M[0] * x + M[1] * y) + M[2] * z) + M[3] * w;
        float wy = ((M[4] * x + M[5] * y) + M[6] * z) + M[7] * w;
        float wz = ((M[8] * x + M[9] * y) + M[10] * z) + M[11] * w;
        next_dir = vnormalized(V(wx, wy, wz));
    }
    int depth = 0;
    float light_pdf = 1.0f, brdf_pdf = 1.0f, f_or_b = 1.0f, brdf = 1.0f;
    int perfect_spec = 1;
    v3 throughout = V(1, 1, 1), radiance = V(0, 0, 0);
    if (st) st->paths++;
    while (depth < max_depth) {
        v3 origin = next_origin, direction = next_dir;
        uint32_t dim0 = TM_DIM_BOUNCE0 + TM_DIMS_PER_BOUNCE * (uint32_t)depth;
        hit_t h = closet_hit(s, origin, direction, stack, stack_size, st);
        if (h.t < INF_VALUE) {
            v3 fnormal = vscale(h.nor, signf(vdot(vneg(direction), h.gnor)));      /* UtilsFunc.py:465-467 */
            int mat_id = s->primitive[(size_t)h.prim * PRI_VEC + 2];
            const float *m = s->material + (size_t)mat_id * MAT_VEC;
            v3 mat_color = V(m[2], m[3], m[4]);
            int mat_type = (int)m[0];
            if (mat_type == MAT_LIGHT) {
                float fCosTheta = fabs_(vdot(direction, h.gnor));
                if (perfect_spec == 1) {
                    radiance = vadd(radiance, vmul(throughout, mat_color));
                } else {
                    float area = get_prim_area(s, h.prim) * (float)s->light_count;
                    light_pdf = (h.t * h.t) / (area * fCosTheta);
                    radiance = vadd(radiance, vmul(vscale(throughout, power_heuristic(brdf_pdf, light_pdf)), mat_color));
                }
                break;
            } else {
                v3 reflect_color = srgb_to_lrgb(mat_color);
                v3 normal = h.nor;
                if (st) st->shaded++;
                if (mat_type == MAT_GLASS) {
                    perfect_spec = 1;
                    next_dir = glass_sample(s, direction, normal, mat_id,
                                            tm_rand(seed, pixel, frame, dim0 + TM_SLOT_GLASS), &f_or_b);
                    brdf = 1.0f; brdf_pdf = 1.0f;                                   /* brdf/Glass.py:72-74 */
                } else {
                    perfect_spec = 0;
                    /* Scene.py:477-518 sample_li.  A scene without emitters (light_count == 0, env-lit): the reference
                     * would index light[-1] (Scene.py:423-428, undefined); defined here as "no NEE sample". */
                    if (s->light_count > 0) {
                    int lidx = (int)(tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LIGHT) * (float)s->light_count);
                    if (lidx >= s->light_count) lidx = s->light_count - 1;
                    int light_prim = s->light[lidx];
                    float ra = tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LA);
                    float rb = tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LB);
                    v3 light_pos, light_normal;
                    get_prim_random_point_normal(s, light_prim, ra, rb, &light_pos, &light_normal);
                    int lmat = s->primitive[(size_t)light_prim * PRI_VEC + 2];
                    const float *lm = s->material + (size_t)lmat * MAT_VEC;
                    v3 light_emission = V(lm[2], lm[3], lm[4]);
                    float light_area = get_prim_area(s, light_prim);
                    float light_choice_pdf = 1.0f / ((float)s->light_count * light_area);
                    light_normal = vnormalized(light_normal);
                    v3 light_dir = vsub(h.pos, light_pos);
                    float light_dist = vnorm(light_dir);
                    light_dir = vdivs(light_dir, light_dist);
                    light_emission = vscale(light_emission, light_shape_visible(s, light_prim, light_dir, light_normal, light_dist, &light_choice_pdf));
                    /* PT_RGB.py:101-109 */
                    float NdotL_surface = vdot(fnormal, light_dir);
                    float NdotL_light = vdot(light_normal, light_dir);
                    if ((NdotL_surface < 0.0f) & (NdotL_light > 0.0f)) {
                        int shadow_prim;
                        (void)closet_hit_shadow(s, light_pos, light_dir, stack, stack_size, &shadow_prim, st);
                        if (shadow_prim == h.prim) {
                            brdf = disney_evaluate_pdf(s, fnormal, vneg(direction), vneg(light_dir), mat_id, &brdf_pdf);
                            light_pdf = light_dist * light_dist * light_choice_pdf / NdotL_light;
                            if (brdf_pdf > 0.0f) {
                                float w = power_heuristic(light_pdf, brdf_pdf) / fmax_(0.0001f, light_pdf);
                                v3 c = vscale(light_emission, w);
                                c = vmul(c, throughout);
                                c = vmul(c, reflect_color);
                                c = vscale(c, brdf);
                                c = vscale(c, fabs_(NdotL_surface));
                                radiance = vadd(radiance, c);
                            }
                        }
                    }
                    }   /* light_count > 0 */
                    float rnd[3] = { tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LOBE),
                                     tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R1),
                                     tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R2) };
                    next_dir = disney_sample(s, direction, fnormal, mat_id, rnd);
                    f_or_b = 1.0f;
                    brdf = disney_evaluate_pdf(s, fnormal, vneg(direction), next_dir, mat_id, &brdf_pdf);
                    brdf *= fabs_(vdot(normal, next_dir));
                }
                next_origin = offset_ray(h.pos, vscale(fnormal, signf(f_or_b)));
                if (brdf_pdf > 0.0f) {
                    if (f_or_b < 0.0f) {
                        float extinction = m[6];
                        float R = m_exp(-h.t / extinction);
                        if (tm_rand(seed, pixel, frame, dim0 + TM_SLOT_EXT) >= R) break;
                    }
                    throughout = vmul(throughout, vscale(reflect_color, brdf / brdf_pdf));
                    depth += 1;
                } else break;
            }
        } else {
            /* PT_RGB.py:127-132 */
            float dis = m_sqrt(direction.x * direction.x + direction.z * direction.z);
            float tx = (m_atan2(direction.z, direction.x) + PI_SCENE) / PI_SCENE / 2.0f;
            float ty = m_atan2(direction.y, dis) / PI_SCENE + 0.5f;
            v3 e = srgb_to_lrgb(texture2d(s, tx, ty));
            radiance = vadd(radiance, vscale(vmul(e, throughout), s->env_power));
            break;
        }
    }
    return radiance;
}

/* =====================================================================================================================
 * PT_Spec: the hero-wavelength spectral path tracer, integrator/PT_Spec.py:44-279, with
 * spectrum/Spectrum.py (tabulated spectra), spectrum/HeroSample.py (four wavelengths 100 nm apart per path),
 * spectrum/Rgb2Spec.py (Jakob-Hanika sigmoid spectra from a 3 x 64^3 coefficient table), sky/Sky.py (the analytic sky
 * dome that is PT_Spec's environment) and brdf/Glass.py:36-59 (dispersion).  Tables arrive as the host prepared them
 * (ti_raytrace_amd/Spectrum.py, Rgb2Spec.py, Sky.py: the same arrays go to the device).
 * ===================================================================================================================== */
#define MAT_SPECTRAL 10         /* SceneData.py:53 */
#define HERO_N 4                /* spectrum/HeroSample.py:5-8 */
static const float HERO_LAMBDA_MIN = 360.0f, HERO_LAMBDA_STEP = (760.0f - 360.0f) / 4.0f;
#define TM_DIM_SPEC_LAMBDA 4000u         /* the path's hero wavelength (PT_Spec.py:191); bounce dims as PT_RGB, slot 7 = get_rnd_hero */
#define TM_SLOT_HERO 7

typedef struct { float v[HERO_N]; } v4s;
typedef struct { const float *data; int n; float lmin, lmax, lrange; } orc_spd;
typedef struct {
    float *sensor; int n_sensor; float s_min, s_max, s_range;       /* CIE 1931 observer, [n][3] (PT_Spec.py:56-77) */
    orc_spd spd[4]; float *spd_mem;                                  /* d65 (normalised), white, red, green */
    float *tbl_scale, *tbl_data; int tbl_res;                       /* Rgb2Spec */
    float sky_cfg[11 * 9], sky_rad[11], sun_dir[3];                 /* Sky.configs / radiances / sun_dir */
} orc_spec;

orc_spec *orc_spec_create(const float *sensor, int n_sensor, float s_min, float s_max, float s_range,
                          const float *spd_concat, const int *spd_n, const float *spd_lmin, const float *spd_lmax, const float *spd_lrange,
                          const float *tbl_scale, const float *tbl_data, int tbl_res,
                          const float *sky_cfg, const float *sky_rad, const float *sun_dir)
{
    orc_spec *sp = (orc_spec *)calloc(1, sizeof(orc_spec));
    sp->sensor = (float *)malloc(sizeof(float) * 3 * (size_t)n_sensor); memcpy(sp->sensor, sensor, sizeof(float) * 3 * (size_t)n_sensor);
    sp->n_sensor = n_sensor; sp->s_min = s_min; sp->s_max = s_max; sp->s_range = s_range;
    int tot = 0; for (int k = 0; k < 4; k++) tot += spd_n[k];
    sp->spd_mem = (float *)malloc(sizeof(float) * (size_t)tot); memcpy(sp->spd_mem, spd_concat, sizeof(float) * (size_t)tot);
    int off = 0;
    for (int k = 0; k < 4; k++) { sp->spd[k].data = sp->spd_mem + off; sp->spd[k].n = spd_n[k]; sp->spd[k].lmin = spd_lmin[k]; sp->spd[k].lmax = spd_lmax[k]; sp->spd[k].lrange = spd_lrange[k]; off += spd_n[k]; }
    sp->tbl_res = tbl_res;
    sp->tbl_scale = (float *)malloc(sizeof(float) * (size_t)tbl_res); memcpy(sp->tbl_scale, tbl_scale, sizeof(float) * (size_t)tbl_res);
    size_t nt = (size_t)9 * tbl_res * tbl_res * tbl_res;
    sp->tbl_data = (float *)malloc(sizeof(float) * nt); memcpy(sp->tbl_data, tbl_data, sizeof(float) * nt);
    memcpy(sp->sky_cfg, sky_cfg, sizeof(sp->sky_cfg)); memcpy(sp->sky_rad, sky_rad, sizeof(sp->sky_rad)); memcpy(sp->sun_dir, sun_dir, sizeof(sp->sun_dir));
    return sp;
}
void orc_spec_destroy(orc_spec *sp) { if (sp) { free(sp->sensor); free(sp->spd_mem); free(sp->tbl_scale); free(sp->tbl_data); free(sp); } }

static float fractf_(float x) { return x - tm_floor(x); }                     /* taichi_glsl fract */
/* spectrum/Spectrum.py:44-52 (the weight is fract(offset), not fract(offset / range): kept; data[idx + 1] at Lambda == lambda_max
 * would be one past the table -- read as the last entry here) */
static float spd_sample(const orc_spd *d, float Lambda)
{
    float ret = 0.0f;
    if ((Lambda >= d->lmin) & (Lambda <= d->lmax)) {
        const float offset = Lambda - d->lmin;
        const int idx = (int)(offset / d->lrange);
        const float w = fractf_(offset);
        const int i1 = idx + 1 < d->n ? idx + 1 : d->n - 1;
        ret = mixf(d->data[idx], d->data[i1], w);
    }
    return ret;
}
/* spectrum/HeroSample.py:10-16 */
static v4s hero_sample(const orc_spd *d, float Lambda0)
{ v4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = spd_sample(d, Lambda0 + (float)i * HERO_LAMBDA_STEP); return r; }
/* integrator/PT_Spec.py:131-139 */
static v3 sensor_sample(const orc_spec *sp, float Lambda)
{
    v3 ret = V(0, 0, 0);
    if ((Lambda >= sp->s_min) & (Lambda <= sp->s_max)) {
        const float offset = Lambda - sp->s_min;
        const int idx = (int)(offset / sp->s_range);
        const float w = fractf_(offset);
        const int i1 = idx + 1 < sp->n_sensor ? idx + 1 : sp->n_sensor - 1;
        const float *a = sp->sensor + 3 * (size_t)idx, *b = sp->sensor + 3 * (size_t)i1;
        ret = V(mixf(a[0], b[0], w), mixf(a[1], b[1], w), mixf(a[2], b[2], w));
    }
    return ret;
}
/* spectrum/Rgb2Spec.py:83-99 */
static int r2s_find_interval(const orc_spec *sp, int size, float x)
{
    int left = 0, last_interval = size - 2;
    size = last_interval;
    while (size > 0) {
        const int half = size >> 1, middle = left + half + 1;
        if (sp->tbl_scale[middle] <= x) { left = middle; size -= half + 1; }
        else size = half;
    }
    return left < last_interval ? left : last_interval;
}
static float r2s_tri(const orc_spec *sp, int i, float x0, float y0, float z0)     /* :77-80 */
{
    const int dx = 3, dy = 3 * sp->tbl_res, dz = 3 * sp->tbl_res * sp->tbl_res;
    const float *t = sp->tbl_data;
    return mixf(mixf(mixf(t[i], t[i + dx], x0), mixf(t[i + dy], t[i + dy + dx], x0), y0),
                mixf(mixf(t[i + dz], t[i + dz + dx], x0), mixf(t[i + dy + dz], t[i + dx + dy + dz], x0), y0), z0);
}
/* spectrum/Rgb2Spec.py:101-137 (fetch) with get_max_component (:50-74) */
static v3 r2s_fetch(const orc_spec *sp, v3 rgb)
{
    rgb = V(clampf(rgb.x, 0.0f, 1.0f), clampf(rgb.y, 0.0f, 1.0f), clampf(rgb.z, 0.0f, 1.0f));
    int index = 0;
    float x = rgb.x, y = rgb.y, z = rgb.z;
    if (rgb.y > rgb.x) {
        if (rgb.z > rgb.y) index = 2;
        else { index = 1; x = rgb.z; y = rgb.x; z = rgb.y; }
    } else {
        if (rgb.z > rgb.x) index = 2;
        else { index = 0; x = rgb.y; y = rgb.z; z = rgb.x; }
    }
    z = fmax_(0.00001f, z);
    const float scale = (float)(sp->tbl_res - 1) / z;
    x *= scale; y *= scale;
    const int res = sp->tbl_res;
    const int xi = (int)fmin_(x, (float)(res - 2)), yi = (int)fmin_(y, (float)(res - 2));
    const int zi = r2s_find_interval(sp, res, z);
    const int offset = (((index * res + zi) * res + yi) * res + xi) * 3;
    const float x0 = x - (float)xi, y0 = y - (float)yi;
    const float z0 = (z - sp->tbl_scale[zi]) / (sp->tbl_scale[zi + 1] - sp->tbl_scale[zi]);
    return V(r2s_tri(sp, offset, x0, y0, z0), r2s_tri(sp, offset + 1, x0, y0, z0), r2s_tri(sp, offset + 2, x0, y0, z0));
}
static float r2s_eval(v3 c, float Lambda)                                       /* :139-143, fma(a,b,c) = a*b + c */
{
    const float x = (c.x * Lambda + c.y) * Lambda + c.z;
    const float y = 1.0f / m_sqrt(x * x + 1.0f);
    return (0.5f * x) * y + 0.5f;
}
/* spectrum/HeroSample.py:46-58 */
static v4s srgb_to_spec(const orc_spec *sp, v3 srgb, float Lambda0)
{
    const v3 coff = r2s_fetch(sp, srgb_to_lrgb(srgb));
    v4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = r2s_eval(coff, Lambda0 + (float)i * HERO_LAMBDA_STEP);
    return r;
}
static v4s v4_mul(v4s a, v4s b) { v4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = a.v[i] * b.v[i]; return r; }
static v4s v4_scale(v4s a, float k) { v4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = a.v[i] * k; return r; }
static v4s v4_add(v4s a, v4s b) { v4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
static v4s v4_divs(v4s a, float k) { v4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = a.v[i] / k; return r; }
/* integrator/PT_Spec.py:102-109 */
static v4s emission_to_rad(const orc_spec *sp, v3 emission, float Lambda)
{
    const float scale = vnorm(emission);
    v4s ret; for (int i = 0; i < HERO_N; i++) ret.v[i] = 0.0f;
    if (scale > 0.0f) ret = srgb_to_spec(sp, vdivs(emission, scale), Lambda);
    return v4_scale(ret, scale);
}
/* integrator/PT_Spec.py:111-127 */
static v4s get_spec_power(const orc_scene *s, const orc_spec *sp, int mat_id, float Lambda)
{
    const float *m = s->material + (size_t)mat_id * MAT_VEC;
    const int mat_type = (int)m[0], mat_tex = (int)m[1];
    v4s ret; for (int i = 0; i < HERO_N; i++) ret.v[i] = 0.0f;
    if (mat_type == MAT_SPECTRAL) {
        if (mat_tex == 0) ret = hero_sample(&sp->spd[1], Lambda);
        if (mat_tex == 1) ret = hero_sample(&sp->spd[2], Lambda);
        if (mat_tex == 2) ret = hero_sample(&sp->spd[3], Lambda);
    } else ret = srgb_to_spec(sp, V(m[2], m[3], m[4]), Lambda);
    return ret;
}
/* sky/Sky.py:176-186, 232-246, 248-255: the sky dome without the sun's disc (solar_radiance_internal2 is commented out there) */
static float sky_internal(const orc_spec *sp, int wl, float theta, float gamma)
{
    const float *c = sp->sky_cfg + 9 * wl;
    const float expM = m_exp(c[4] * gamma);
    const float rayM = m_cos(gamma) * m_cos(gamma);
    const float mieM = (1.0f + m_cos(gamma) * m_cos(gamma)) / m_pow((1.0f + c[8] * c[8]) - 2.0f * c[8] * m_cos(gamma), 1.5f);
    const float zenith = m_sqrt(m_cos(theta));
    return (1.0f + c[0] * m_exp(c[1] / (m_cos(theta) + 0.01f))) *
           ((((c[2] + c[3] * expM) + c[5] * rayM) + c[6] * mieM) + c[7] * zenith);
}
static float sky_radiance(const orc_spec *sp, float theta, float gamma, float wavelength)
{
    float ret = 0.0f;
    if ((wavelength >= 320.0f) & (wavelength <= 720.0f)) {
        const int low_wl = (int)((wavelength - 320.0f) / 40.0f);
        float result = 0.0f;
        if ((low_wl >= 0) & (low_wl < 11)) {
            const float interp = fractf_((wavelength - 320.0f) / 40.0f);
            const float val_low = sky_internal(sp, low_wl, theta, gamma) * sp->sky_rad[low_wl];
            if (interp < 1e-6f) result = val_low;
            else {
                result = (1.0f - interp) * val_low;
                if (low_wl + 1 < 11) result += interp * sky_internal(sp, low_wl + 1, theta, gamma) * sp->sky_rad[low_wl + 1];
            }
        }
        ret = result;
    }
    return ret;
}
/* brdf/Glass.py:36-59 with UF.get_glass_ior (UtilsFunc.py:481-484, BK7 Sellmeier) */
static float get_glass_ior(float Lambda)
{
    Lambda = Lambda / 1000.0f;
    const float L2 = Lambda * Lambda;
    return m_sqrt(((1.0f + 1.03961212f * L2 / (L2 - 0.00600069867f)) + 0.231792344f * L2 / (L2 - 0.0200179144f)) + 1.01046945f * L2 / (L2 - 103.560653f));
}
static v3 glass_sample_lambda(v3 dir, v3 N, float Lambda, float probability, float *f_or_b)
{
    v3 w_out = dir;
    float cos_theta_i = vdot(w_out, N);
    const float ior = get_glass_ior(Lambda);
    float eta = ior;
    *f_or_b = 1.0f;
    float R = probability + 1.0f;
    if (cos_theta_i > 0.0f) N = vneg(N);
    else { cos_theta_i = -cos_theta_i; eta = 1.0f / ior; }
    float suc;
    v3 next_dir = refract_(w_out, N, eta, &suc);
    if (suc > 0.0f) R = schlick(cos_theta_i, ior);
    if (probability < R) next_dir = reflect_(w_out, N);
    else *f_or_b = -1.0f;
    return next_dir;
}

/* integrator/PT_Spec.py:181-279: one pixel-sample; returns the four hero radiances and the hero wavelength */
static v4s pt_spec_pixel(const orc_scene *s, const orc_spec *sp, int i, int j, int H, uint32_t frame, uint32_t seed,
                         int max_depth, int32_t *stack, int stack_size, orc_stats *st, float *Lambda_out)
{
    uint32_t pixel = (uint32_t)(i * H + j);
    v3 next_origin = V(s->eye[0], s->eye[1], s->eye[2]);
    float jx = 0.0f, jy = 0.0f;
    if (frame != 0) {
        jx = tm_rand(seed, pixel, frame, TM_DIM_JX) - 0.5f;
        jy = tm_rand(seed, pixel, frame, TM_DIM_JY) - 0.5f;
    }
    v3 next_dir;
    {
        float x = ((float)i + jx - s->cx) / s->fx;
        float y = ((float)j + jy - s->cy) / s->fy;
        float z = -1.0f, w = 0.0f;
        const float *M = s->view_inv;
        float wx = ((M[0] * x + M[1] * y) + M[2] * z) + M[3] * w;
        float wy = ((M[4] * x + M[5] * y) + M[6] * z) + M[7] * w;
        float wz = ((M[8] * x + M[9] * y) + M[10] * z) + M[11] * w;
        next_dir = vnormalized(V(wx, wy, wz));
    }
    const float Lambda = HERO_LAMBDA_MIN + HERO_LAMBDA_STEP * tm_rand(seed, pixel, frame, TM_DIM_SPEC_LAMBDA);
    *Lambda_out = Lambda;
    int depth = 0;
    float light_pdf = 1.0f, brdf_pdf = 1.0f, f_or_b = 1.0f, brdf = 1.0f;
    v4s throughout, radiance;
    for (int k = 0; k < HERO_N; k++) { throughout.v[k] = 1.0f; radiance.v[k] = 0.0f; }
    const v4s light_rad = hero_sample(&sp->spd[0], Lambda);          /* the same for every vertex of the path (:212) */
    if (st) st->paths++;
    while (depth < max_depth) {
        v3 origin = next_origin, direction = next_dir;
        uint32_t dim0 = TM_DIM_BOUNCE0 + TM_DIMS_PER_BOUNCE * (uint32_t)depth;
        hit_t h = closet_hit(s, origin, direction, stack, stack_size, st);
        if (h.t < INF_VALUE) {
            v3 normal = h.nor;
            v3 fnormal = vscale(h.nor, signf(vdot(vneg(direction), h.gnor)));
            int mat_id = s->primitive[(size_t)h.prim * PRI_VEC + 2];
            const float *m = s->material + (size_t)mat_id * MAT_VEC;
            v3 mat_color = V(m[2], m[3], m[4]);
            int mat_type = (int)m[0];
            /* :213 -- the "tint" is the colour of the material that was HIT, also where the NEE sample below uses it */
            const v4s light_tint = emission_to_rad(sp, mat_color, Lambda);
            if (mat_type == MAT_LIGHT) {
                const float fCosTheta = vdot(direction, normal);
                if (fCosTheta < 0.0f) {
                    const float area = get_prim_area(s, h.prim);
                    light_pdf = (h.t * h.t) / (area * fCosTheta);
                    (void)light_pdf;       /* perfect_spec is reset to 1 at the top of every iteration (:214): the MIS branch is dead */
                    radiance = v4_add(radiance, v4_mul(v4_mul(throughout, light_rad), light_tint));
                }
                break;
            } else {
                if (st) st->shaded++;
                const v4s reflect_spec = get_spec_power(s, sp, mat_id, Lambda);
                if (mat_type == MAT_GLASS) {
                    int index = (int)(tm_rand(seed, pixel, frame, dim0 + TM_SLOT_HERO) * (float)HERO_N);     /* HeroSample.py:33-35 */
                    const float rnd_lambda = Lambda + (float)index * HERO_LAMBDA_STEP;
                    next_dir = glass_sample_lambda(direction, normal, rnd_lambda, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_GLASS), &f_or_b);
                    brdf = 1.0f; brdf_pdf = 1.0f;
                } else {
                    if (s->light_count > 0) {
                    int lidx = (int)(tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LIGHT) * (float)s->light_count);
                    if (lidx >= s->light_count) lidx = s->light_count - 1;
                    int light_prim = s->light[lidx];
                    float ra = tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LA);
                    float rb = tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LB);
                    v3 light_pos, light_normal;
                    get_prim_random_point_normal(s, light_prim, ra, rb, &light_pos, &light_normal);
                    float light_area = get_prim_area(s, light_prim);
                    float light_choice_pdf = 1.0f / ((float)s->light_count * light_area);
                    light_normal = vnormalized(light_normal);
                    v3 light_dir = vsub(h.pos, light_pos);
                    float light_dist = vnorm(light_dir);
                    light_dir = vdivs(light_dir, light_dist);
                    (void)light_shape_visible(s, light_prim, light_dir, light_normal, light_dist, &light_choice_pdf);   /* only its pdf reaches :245 */
                    float NdotL_surface = vdot(fnormal, light_dir);
                    float NdotL_light = vdot(light_normal, light_dir);
                    if ((NdotL_surface < 0.0f) & (NdotL_light > 0.0f)) {
                        int shadow_prim;
                        (void)closet_hit_shadow(s, light_pos, light_dir, stack, stack_size, &shadow_prim, st);
                        if (shadow_prim == h.prim) {
                            brdf = disney_evaluate_pdf(s, fnormal, vneg(direction), vneg(light_dir), mat_id, &brdf_pdf);
                            light_pdf = light_dist * light_dist * light_choice_pdf / NdotL_light;
                            if (brdf_pdf > 0.0f) {
                                const float w = power_heuristic(light_pdf, brdf_pdf) / fmax_(0.0001f, light_pdf);
                                v4s c = v4_scale(light_rad, w);
                                c = v4_mul(c, light_tint);
                                c = v4_mul(c, throughout);
                                c = v4_mul(c, reflect_spec);
                                c = v4_scale(c, brdf);
                                c = v4_scale(c, fabs_(NdotL_surface));
                                radiance = v4_add(radiance, c);
                            }
                        }
                    }
                    }
                    float rnd[3] = { tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LOBE),
                                     tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R1),
                                     tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R2) };
                    next_dir = disney_sample(s, direction, fnormal, mat_id, rnd);
                    f_or_b = 1.0f;
                    brdf = disney_evaluate_pdf(s, fnormal, next_dir, vneg(direction), mat_id, &brdf_pdf);      /* (N, V = next_dir, L = -direction): :251 */
                    brdf *= fabs_(vdot(normal, next_dir));
                }
                next_origin = offset_ray(h.pos, vscale(fnormal, signf(f_or_b)));
                if ((brdf_pdf > 0.0f) & (fmax_(throughout.v[2], fmax_(throughout.v[0], throughout.v[1])) > 0.0f)) {
                    throughout = v4_mul(throughout, v4_divs(v4_scale(reflect_spec, brdf), brdf_pdf));
                    depth += 1;
                } else break;
            }
        } else {
            /* :267-274: the analytic sky */
            const float dis = m_sqrt(direction.x * direction.x + direction.z * direction.z);
            const float beta = m_atan2(direction.y, dis);
            const float gamma = m_acos(vdot(direction, V(sp->sun_dir[0], sp->sun_dir[1], sp->sun_dir[2])));
            const float theta = clampf(0.5f * PI_SCENE - beta, 0.0f, 0.5f * PI_SCENE);
            v4s ibl;
            for (int k = 0; k < HERO_N; k++) ibl.v[k] = sky_radiance(sp, theta, gamma, Lambda + (float)k * HERO_LAMBDA_STEP);
            radiance = v4_add(radiance, v4_mul(v4_mul(throughout, ibl), light_rad));
            break;
        }
    }
    return radiance;
}
/* integrator/PT_Spec.py:141-158 AddSplat: hero radiances -> CIE XYZ -> linear sRGB, mixed into hdr */
static void spec_add_splat(const orc_spec *sp, v4s spec, float Lambda0, float coff, float *px)
{
    float xf[HERO_N], yf[HERO_N], zf[HERO_N];
    for (int k = 0; k < HERO_N; k++) {
        const v3 xyz = sensor_sample(sp, Lambda0 + (float)k * HERO_LAMBDA_STEP);
        xf[k] = xyz.x * spec.v[k]; yf[k] = xyz.y * spec.v[k]; zf[k] = xyz.z * spec.v[k];
    }
    const float range = sp->s_max - sp->s_min;
    float X = 0.0f, Y = 0.0f, Z = 0.0f;
    for (int k = 0; k < HERO_N; k++) { X += xf[k] * range / (float)HERO_N; Y += yf[k] * range / (float)HERO_N; Z += zf[k] * range / (float)HERO_N; }
    /* UtilsFunc.py:42 xyz_to_srgb @ xyz */
    const float r = ((float)3.240479 * X + (float)-1.537150 * Y) + (float)-0.498535 * Z;
    const float g = ((float)-0.969256 * X + (float)1.875991 * Y) + (float)0.041556 * Z;
    const float b = ((float)0.055648 * X + (float)-0.204043 * Y) + (float)1.057311 * Z;
    px[0] = mixf(px[0], r, coff); px[1] = mixf(px[1], g, coff); px[2] = mixf(px[2], b, coff);
}

typedef struct {
    const orc_scene *s; int W, H; uint32_t frame_begin; int frame_count; uint32_t seed;
    int max_depth, stack_size; float *hdr;
    const orc_spec *spec;           /* non-NULL: PT_Spec instead of PT_RGB */
    long p_begin, p_end;            /* linear pixel range [p_begin, p_end), p = i*H + j */
    int tile_rank, tile_count, tile_size;
    orc_stats st;
} render_job;

static void *render_worker(void *arg)
{
    render_job *jb = (render_job *)arg;
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(jb->stack_size + 2));
    for (long p = jb->p_begin; p < jb->p_end; p++) {
        if (jb->tile_count > 1 && (int)((p / jb->tile_size) % jb->tile_count) != jb->tile_rank) continue;
        int i = (int)(p / jb->H), j = (int)(p % jb->H);
        float *px = jb->hdr + (size_t)p * 3;
        for (int f = 0; f < jb->frame_count; f++) {
            uint32_t frame = jb->frame_begin + (uint32_t)f;
            float ff = (float)(int32_t)frame;
            float coff = 1.0f / (ff + 1.0f);
            if (jb->spec) {
                float Lambda;
                v4s sr = pt_spec_pixel(jb->s, jb->spec, i, j, jb->H, frame, jb->seed, jb->max_depth, stack, jb->stack_size, &jb->st, &Lambda);
                spec_add_splat(jb->spec, sr, Lambda, coff, px);
                continue;
            }
            v3 rad = pt_rgb_pixel(jb->s, i, j, jb->H, frame, jb->seed, jb->max_depth, stack, jb->stack_size, &jb->st);
            /* PT_RGB.py:134-136 */
            px[0] = rad.x * coff + px[0] * (1.0f - coff);
            px[1] = rad.y * coff + px[1] * (1.0f - coff);
            px[2] = rad.z * coff + px[2] * (1.0f - coff);
        }
    }
    free(stack);
    return NULL;
}

/* Work distribution: the pixel range is cut into small chunks handed out through one atomic counter
 * (dynamic queue: path lengths vary a lot between pixels), every thread accumulates its statistics in a
 * thread-local struct (the per-chunk structs of the first version sat next to each other in memory and
 * were incremented per ray: false sharing across 256 threads). */
typedef struct { render_job proto; long p_begin, total; int chunks; int *next; orc_stats st; char pad[128]; } thread_ctx;
static void *thread_main(void *arg)
{
    thread_ctx *tc = (thread_ctx *)arg;
    render_job jb = tc->proto;
    memset(&jb.st, 0, sizeof(jb.st));
    for (;;) {
        int c = __atomic_fetch_add(tc->next, 1, __ATOMIC_RELAXED);
        if (c >= tc->chunks) break;
        jb.p_begin = tc->p_begin + tc->total * c / tc->chunks;
        jb.p_end = tc->p_begin + tc->total * (c + 1) / tc->chunks;
        render_worker(&jb);
    }
    tc->st = jb.st;
    return NULL;
}

/* hdr: [W*H*3], index (i*H + j)*3, read-modify-written (running mean, PT_RGB.py:134-136).
 * Pixels are visited for linear index p in [p_begin, p_end) whose tile (p / tile_size) %
 * tile_count == tile_rank (tile_count <= 1: all).  stats may be NULL. */
static int render_common(const orc_scene *s, const orc_spec *spec, int W, int H, uint32_t frame_begin, int frame_count,
                         uint32_t seed, int max_depth, int stack_size, float *hdr,
                         long p_begin, long p_end, int tile_rank, int tile_count, int tile_size,
                         int nthreads, orc_stats *stats)
{
    if (p_end > (long)W * H) p_end = (long)W * H;
    if (p_begin < 0) p_begin = 0;
    if (nthreads < 1) nthreads = 1;
    long total = p_end - p_begin;
    if (total <= 0) return 0;
    long chunks_l = (total + 63) / 64;                 /* ~64 pixels per chunk */
    if (chunks_l > (1l << 30)) chunks_l = 1l << 30;
    int chunks = (int)chunks_l;
    if (nthreads > chunks) nthreads = chunks;
    int next = 0;
    thread_ctx *tc = (thread_ctx *)calloc((size_t)nthreads, sizeof(thread_ctx));
    for (int t = 0; t < nthreads; t++) {
        render_job *jb = &tc[t].proto;
        jb->s = s; jb->W = W; jb->H = H; jb->frame_begin = frame_begin; jb->frame_count = frame_count;
        jb->seed = seed; jb->max_depth = max_depth; jb->stack_size = stack_size; jb->hdr = hdr; jb->spec = spec;
        jb->tile_rank = tile_rank; jb->tile_count = tile_count; jb->tile_size = tile_size > 0 ? tile_size : 1;
        tc[t].p_begin = p_begin; tc[t].total = total; tc[t].chunks = chunks; tc[t].next = &next;
    }
    if (nthreads == 1) {
        thread_main(&tc[0]);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, thread_main, &tc[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (int t = 0; t < nthreads; t++) {
            const orc_stats *a = &tc[t].st;
            stats->rays_closest += a->rays_closest; stats->rays_shadow += a->rays_shadow;
            stats->box_closest += a->box_closest; stats->leaf_closest += a->leaf_closest;
            stats->box_shadow += a->box_shadow; stats->leaf_shadow += a->leaf_shadow;
            stats->shaded += a->shaded; stats->paths += a->paths; stats->overflow += a->overflow;
            if (a->max_stack > stats->max_stack) stats->max_stack = a->max_stack;
        }
    }
    free(tc);
    return 0;
}
int orc_pt_rgb_render(const orc_scene *s, int W, int H, uint32_t frame_begin, int frame_count,
                      uint32_t seed, int max_depth, int stack_size, float *hdr,
                      long p_begin, long p_end, int tile_rank, int tile_count, int tile_size,
                      int nthreads, orc_stats *stats)
{ return render_common(s, NULL, W, H, frame_begin, frame_count, seed, max_depth, stack_size, hdr, p_begin, p_end, tile_rank, tile_count, tile_size, nthreads, stats); }
/* PT_Spec.render x frame_count (integrator/PT_Spec.py:181-279, MAX_DEPTH 10): same film conventions as orc_pt_rgb_render */
int orc_pt_spec_render(const orc_scene *s, const orc_spec *spec, int W, int H, uint32_t frame_begin, int frame_count,
                       uint32_t seed, int max_depth, int stack_size, float *hdr,
                       long p_begin, long p_end, int tile_rank, int tile_count, int tile_size,
                       int nthreads, orc_stats *stats)
{ return render_common(s, spec, W, H, frame_begin, frame_count, seed, max_depth, stack_size, hdr, p_begin, p_end, tile_rank, tile_count, tile_size, nthreads, stats); }
/* =====================================================================================================================
 * spectrum/JakobSpecTable.py:1-439 -- the offline optimiser behind spectrum/spec_table (the reference repository lacks the
 * table itself: .MISSING_LARGE_BLOBS).  For every cell of a 3 x res^3 grid over RGB (largest component l, its value through
 * scale[k], the two others as fractions i, j of it) three sigmoid-polynomial coefficients are fitted by Gauss-Newton in CIE
 * Lab under D65, warm-started along k.  Double precision throughout, as the reference (ti.init(default_fp=ti.f64)).
 * cie_xyz: [n][3] and d65: [n] for 360..830 nm in 1 nm steps (n = 471), as float32 (the reference loads them through float32
 * numpy arrays, :386-388).  scale_out: [res], coeff_out: [3*res^3*3] as float32 (the table file holds %.9g of the doubles).
 * ===================================================================================================================== */
typedef struct { int n; double rgb_tbl[471][3], lam[471], wp[3]; } spt_ctx;
static double spt_sqr(double x) { return x * x; }
static double spt_smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }
static double spt_sigmoid(double x) { return 0.5 * x / tm_sqrtd(1.0 + x * x) + 0.5; }
static double spt_f(double t)                                     /* :88-97; pow(t, 1/3) = exp(log(t) / 3) with the shared double kernels */
{
    const double delta = 6.0 / 29.0;
    return (t > delta * delta * delta) ? tm_expd(tm_logd(t) * (1.0 / 3.0)) : t / (delta * delta * 3.0) + (4.0 / 29.0);
}
static void spt_cie_lab(const spt_ctx *c, const double p[3], double out[3])   /* :99-105 */
{
    const double X = (0.412453 * p[0] + 0.357580 * p[1]) + 0.180423 * p[2];
    const double Y = (0.212671 * p[0] + 0.715160 * p[1]) + 0.072169 * p[2];
    const double Z = (0.019334 * p[0] + 0.119193 * p[1]) + 0.950227 * p[2];
    out[0] = 116.0 * spt_f(Y / c->wp[1]) - 16.0;
    out[1] = 500.0 * (spt_f(X / c->wp[0]) - spt_f(Y / c->wp[1]));
    out[2] = 200.0 * (spt_f(Y / c->wp[1]) - spt_f(Z / c->wp[2]));
}
static void spt_residual(const spt_ctx *c, const double co[3], const double rgb[3], double out[3])   /* :260-277 */
{
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < c->n; i++) {
        const double L = (c->lam[i] - 360.0) / (830.0 - 360.0);
        double x = co[0];
        x = x * L + co[1];
        x = x * L + co[2];
        const double sg = spt_sigmoid(x);
        acc[0] += c->rgb_tbl[i][0] * sg; acc[1] += c->rgb_tbl[i][1] * sg; acc[2] += c->rgb_tbl[i][2] * sg;
    }
    double a[3], b[3];
    spt_cie_lab(c, rgb, a); spt_cie_lab(c, acc, b);
    out[0] = a[0] - b[0]; out[1] = a[1] - b[1]; out[2] = a[2] - b[2];
}
/* :107-209 LUPDecompose, as written there: the pivot search of the SECOND column looks at column 0 again (A[k,0], :170) -- kept */
static int spt_lup(double A[3][3], int P[4])
{
    const double Tol = 1e-15;
    int ret = 1;
    P[0] = 0; P[1] = 1; P[2] = 2; P[3] = 3;
    double maxA = 0.0; int imax = 0;
    for (int k = 0; k < 3; k++) { const double a = A[k][0] < 0 ? -A[k][0] : A[k][0]; if (a > maxA) { maxA = a; imax = k; } }
    if (maxA < Tol) ret = 0;
    if (imax != 0) {
        const int o = imax;           /* swap rows 0 and imax, and the permutation */
        for (int q = 0; q < 3; q++) { const double t = A[0][q]; A[0][q] = A[o][q]; A[o][q] = t; }
        { const int t = P[0]; P[0] = P[o]; P[o] = t; }
        P[3] += 1;
    }
    A[1][0] /= A[0][0]; A[1][1] -= A[1][0] * A[0][1]; A[1][2] -= A[1][0] * A[0][2];
    A[2][0] /= A[0][0]; A[2][1] -= A[2][0] * A[0][1]; A[2][2] -= A[2][0] * A[0][2];
    maxA = 0.0; imax = 1;
    for (int k = 1; k < 3; k++) { const double a = A[k][0] < 0 ? -A[k][0] : A[k][0]; if (a > maxA) { maxA = a; imax = k; } }
    if (maxA < Tol) ret = 0;
    if (imax != 1) {
        for (int q = 0; q < 3; q++) { const double t = A[1][q]; A[1][q] = A[2][q]; A[2][q] = t; }
        { const int t = P[1]; P[1] = P[2]; P[2] = t; }
        P[3] += 1;
    }
    A[2][1] /= A[1][1]; A[2][2] -= A[2][1] * A[1][2];
    if ((A[2][2] < 0 ? -A[2][2] : A[2][2]) < Tol) ret = 0;
    return ret;
}
static void spt_solve(double A[3][3], const int P[4], const double b[3], double x[3])     /* :212-257 */
{
    x[0] = b[P[0]];
    x[1] = b[P[1]]; x[1] -= A[1][0] * x[0];
    x[2] = b[P[2]]; x[2] -= A[2][0] * x[0]; x[2] -= A[2][1] * x[1];
    x[2] = x[2] / A[2][2];
    x[1] -= A[1][2] * x[2]; x[1] = x[1] / A[1][1];
    x[0] -= A[0][1] * x[1]; x[0] -= A[0][2] * x[2]; x[0] = x[0] / A[0][0];
}
static int spt_gauss_newton(const spt_ctx *c, const double rgb[3], double co[3])         /* :301-332 */
{
    const double EPS = 1e-4;
    int rv = 1;
    for (int it = 0; it < 15; it++) {
        double res[3], J[3][3];
        spt_residual(c, co, rgb, res);
        for (int i = 0; i < 3; i++) {
            double tmp[3] = {co[0], co[1], co[2]}, r0[3], r1[3];
            tmp[i] -= EPS; spt_residual(c, tmp, rgb, r0);
            tmp[0] = co[0]; tmp[1] = co[1]; tmp[2] = co[2];
            tmp[i] += EPS; spt_residual(c, tmp, rgb, r1);
            for (int q = 0; q < 3; q++) J[q][i] = (r1[q] - r0[q]) / (2.0 * EPS);
        }
        int P[4];
        rv = spt_lup(J, P);
        if (rv != 1) break;
        double x[3];
        spt_solve(J, P, res, x);
        co[0] -= x[0]; co[1] -= x[1]; co[2] -= x[2];
        const double r = (res[0] * res[0] + res[1] * res[1]) + res[2] * res[2];
        if (r < 0.000001) break;
        const double m01 = co[0] > co[1] ? co[0] : co[1], cm = m01 > co[2] ? m01 : co[2];
        if (cm > 200.0) { const double k = 200.0 / cm; co[0] *= k; co[1] *= k; co[2] *= k; }
    }
    return rv;
}
static void spt_get_rgb(const double *scale, int k, double x, double y, int l, double rgb[3])   /* :51-67 */
{
    const double b = scale[k];
    rgb[0] = rgb[1] = rgb[2] = 0.0;
    if (l == 0) { rgb[0] = b; rgb[1] = x * b; rgb[2] = y * b; }
    else if (l == 1) { rgb[1] = b; rgb[2] = x * b; rgb[0] = y * b; }
    else { rgb[2] = b; rgb[0] = x * b; rgb[1] = y * b; }
}
static void spt_write(float *out, long idx, const double co[3])                               /* :69-77 */
{
    const double c0 = 360.0, c1 = 1.0 / (830.0 - 360.0);
    out[3 * idx + 0] = (float)(co[0] * spt_sqr(c1));
    out[3 * idx + 1] = (float)(co[1] * c1 - 2 * co[0] * c0 * spt_sqr(c1));
    out[3 * idx + 2] = (float)(co[2] - co[1] * c0 * c1 + co[0] * spt_sqr(c0 * c1));
}
typedef struct { const spt_ctx *c; const double *scale; int res, l, j0, j1; float *out; } spt_job;
static void *spt_worker(void *arg)
{
    spt_job *jb = (spt_job *)arg;
    const int res = jb->res;
    for (int j = jb->j0; j < jb->j1; j++) for (int i = 0; i < res; i++) {                    /* sovle(l), :345-375 */
        const double x = (double)i / (double)(res - 1), y = (double)j / (double)(res - 1);
        double co[3] = {0.0, 0.0, 0.0}, rgb[3];
        for (int k = res / 5; k < res; k++) {
            spt_get_rgb(jb->scale, k, x, y, jb->l, rgb);
            if (spt_gauss_newton(jb->c, rgb, co) != 1) break;
            spt_write(jb->out, (((long)jb->l * res + k) * res + j) * res + i, co);
        }
        co[0] = co[1] = co[2] = 0.0;
        for (int k = res / 5; k >= 0; k--) {
            spt_get_rgb(jb->scale, k, x, y, jb->l, rgb);
            if (spt_gauss_newton(jb->c, rgb, co) != 1) break;
            spt_write(jb->out, (((long)jb->l * res + k) * res + j) * res + i, co);
        }
    }
    return NULL;
}
int orc_spec_table_build(int res, const float *cie_xyz, const float *d65, int n, float *scale_out, float *coeff_out, int nthreads)
{
    if (n != 471 || res < 5 || res > 64) return -1;
    spt_ctx *c = (spt_ctx *)calloc(1, sizeof(spt_ctx));
    c->n = n;
    const double h = (830.0 - 360.0) / (double)(n - 1);
    double wp[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < n; i++) {                                                             /* pre_compute, :334-343 */
        double weight = 3.0 / 8.0 * h;
        if ((i == 0) | (i == n - 1)) weight = weight;
        else if ((i - 1) % 3 == 2) weight = weight * 2.0;
        else weight = weight * 3.0;
        const double X = (double)cie_xyz[3 * i], Y = (double)cie_xyz[3 * i + 1], Z = (double)cie_xyz[3 * i + 2], D = (double)d65[i];
        c->lam[i] = 360.0 + (double)i;
        c->rgb_tbl[i][0] = (((3.240479 * X + -1.537150 * Y) + -0.498535 * Z) * D) * weight;
        c->rgb_tbl[i][1] = (((-0.969256 * X + 1.875991 * Y) + 0.041556 * Z) * D) * weight;
        c->rgb_tbl[i][2] = (((0.055648 * X + -0.204043 * Y) + 1.057311 * Z) * D) * weight;
        wp[0] += (X * D) * weight; wp[1] += (Y * D) * weight; wp[2] += (Z * D) * weight;
    }
    double scale[64];
    for (int i = 0; i < res; i++) { scale[i] = spt_smoothstep(spt_smoothstep((double)i / (double)(res - 1))); scale_out[i] = (float)scale[i]; }
    /* :413-417: white point and rgb table normalised by the white point's Y */
    for (int i = 0; i < n; i++) { c->rgb_tbl[i][0] /= wp[1]; c->rgb_tbl[i][1] /= wp[1]; c->rgb_tbl[i][2] /= wp[1]; }
    c->wp[0] = wp[0] / wp[1]; c->wp[2] = wp[2] / wp[1]; c->wp[1] = wp[1] / wp[1];
    memset(coeff_out, 0, sizeof(float) * 9 * (size_t)res * res * res);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > res) nthreads = res;
    for (int l = 0; l < 3; l++) {
        pthread_t th[64]; spt_job jb[64];
        for (int t = 0; t < nthreads; t++) {
            jb[t].c = c; jb[t].scale = scale; jb[t].res = res; jb[t].l = l; jb[t].out = coeff_out;
            jb[t].j0 = res * t / nthreads; jb[t].j1 = res * (t + 1) / nthreads;
            pthread_create(&th[t], NULL, spt_worker, &jb[t]);
        }
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    free(c);
    return 0;
}

/* UtilsFunc.py:583-586; in/out: [npix*3] */
void orc_tone_map(float exposure, const float *in, float *out, long npix)
{
    for (long p = 0; p < npix; p++) {
        v3 x = V(in[3 * p] * exposure, in[3 * p + 1] * exposure, in[3 * p + 2] * exposure);
        v3 y = lrgb_to_srgb(V(tone_aces1(x.x), tone_aces1(x.y), tone_aces1(x.z)));
        out[3 * p] = y.x; out[3 * p + 1] = y.y; out[3 * p + 2] = y.z;
    }
}

/* Scene.py:747-750 */
float orc_total_area(const orc_scene *s)
{
    float a = 0.0f;
    int cnt = s->light_count > 0 ? s->light_count : 1;     /* light field has >= 1 entry (Scene.py:258-261) */
    for (int i = 0; i < cnt; i++) a += get_prim_area(s, s->light[i]);
    return a;
}

/* Scene.py:353-377 */
static float get_prim_angle(const orc_scene *s, int index, v3 v)
{
    float ret = 0.0f;
    const int32_t *pr = s->primitive + (size_t)index * PRI_VEC;
    if (pr[0] == PRIMITIVE_TRI) {
        v3 v1 = vtx_pos(s, pr[1]), v2 = vtx_pos(s, pr[1] + 1), v3_ = vtx_pos(s, pr[1] + 2);
        if (vnorm(vsub(v1, v)) < 0.00001f) ret = vdot(vnormalized(vsub(v2, v1)), vnormalized(vsub(v3_, v1)));
        else if (vnorm(vsub(v2, v)) < 0.00001f) ret = vdot(vnormalized(vsub(v1, v2)), vnormalized(vsub(v3_, v2)));
        else ret = vdot(vnormalized(vsub(v1, v3_)), vnormalized(vsub(v2, v3_)));
    }
    return m_acos(ret);
}

/* Scene.py:754-798.  vertex_index[i] = primitive that owns vertex i (Scene.py:128).
 * Rewrites the normals in s->vertex. */
void orc_process_normal(orc_scene *s, const int32_t *vertex_index)
{
    const int MAX_STACK_SIZE = 32;    /* Scene.py:19 */
    float *smooth = (float *)calloc((size_t)s->nv * 3, sizeof(float));
    int32_t stack[34];
    for (int i = 0; i < s->nv; i++) {
        v3 v = vtx_pos(s, i);
        v3 n = vnormalized(vtx_nor(s, i));
        int f = vertex_index[i];
        v3 sm = vscale(vscale(n, get_prim_angle(s, f, v)), get_prim_area(s, f));
        stack[0] = 0;
        int stack_pos = 0;
        while ((stack_pos >= 0) & (stack_pos < MAX_STACK_SIZE)) {
            int node = stack[stack_pos];
            stack_pos -= 1;
            const float *cn = s->compact + (size_t)node * CPN_VEC;
            if (cn_is_leaf(cn) == IS_LEAF) {
                int prim = (int)cn[1];
                const int32_t *pr = s->primitive + (size_t)prim * PRI_VEC;
                if (pr[0] == PRIMITIVE_TRI) {
                    for (int j = 0; j < 3; j++) {
                        int nb = j + pr[1];
                        if (i != nb) {
                            v3 nv = vtx_pos(s, nb);
                            v3 nn = vnormalized(vtx_nor(s, nb));
                            if ((vnorm(vsub(v, nv)) < 0.000001f) & (vdot(nn, n) > 0.5f)) {
                                float angle = get_prim_angle(s, prim, nv);
                                sm = vadd(sm, vscale(vscale(nn, angle), get_prim_area(s, prim)));
                            }
                        }
                    }
                }
            } else {
                if ((v.x >= cn[2]) & (v.y >= cn[3]) & (v.z >= cn[4]) & (v.x <= cn[5]) & (v.y <= cn[6]) & (v.z <= cn[7])) {
                    stack_pos += 1; stack[stack_pos] = node + 1;
                    stack_pos += 1; stack[stack_pos] = (int)cn[1];
                }
            }
        }
        smooth[3 * i] = sm.x; smooth[3 * i + 1] = sm.y; smooth[3 * i + 2] = sm.z;
    }
    for (int i = 0; i < s->nv; i++) {
        v3 nn = vnormalized(V(smooth[3 * i], smooth[3 * i + 1], smooth[3 * i + 2]));
        float *p = s->vertex + (size_t)i * VER_VEC;
        p[3] = nn.x; p[4] = nn.y; p[5] = nn.z;
    }
    free(smooth);
}

/* ===================================================================================== */
/* BDPT_RGB (BASELINE config 5)                                      */
/* ===================================================================================== */
/* Restates integrator/BDPT_RGB.py + BDPT_Vertex.py.  "Parity unpinned": the reference holds
 * no golden output for it (only the gallery image image/veach-bdpt512.png).  Reference
 * behaviours kept on purpose:
 *  - the per-pixel vertex arrays (eye[7], light[6], sample, 4 temporaries) persist across
 *    frames and only beta/type/fpdf/rpdf are cleared per frame (BDPT_RGB.py:597-614), so
 *    delta/prim/mat/normal of a slot can be stale (e.g. `delta` of a light-type eye vertex);
 *  - `mat_id == SCD.MAT_DISNEY` compares a material INDEX with the type constant 0
 *    (:364,379,432);
 *  - mis_weight restores light[l-1], light[l-2], eye[e-2] with indices that can be -1
 *    (:472-477).  Taichi pads the depth axis (6, 7) to 8, so index -1 lands in padding; the
 *    restatement skips those copies.
 * RNG dimensions (tirt_math.h generator, same pixel/frame keys as PT_RGB):
 *   0,1 jitter | 16+8d+slot eye bounce d | 80..84 light start (index, a, b, u1, u2)
 *   | 96+8d+slot light bounce d | 176+4e+{0,1,2} sample_li of the l==1 connection at eye vertex e. */
#define BD_MAX_DEPTH 5            /* BDPT_RGB.py:23 */
#define BD_EYE_MAX (BD_MAX_DEPTH + 2)
#define BD_LIGHT_MAX (BD_MAX_DEPTH + 1)
#define VERTEX_NONE 0
#define VERTEX_LIGHT 1
#define VERTEX_LENS 2
#define VERTEX_SURFACE 3
#define BD_DIM_EYE 16
#define BD_DIM_LSTART 80
#define BD_DIM_LIGHT 96
#define BD_DIM_CONNECT 176
static const float EPS_UF = 0.00001f;      /* UtilsFunc.py:36 */

typedef struct { v3 pos, normal, snormal, beta, wo; float fpdf, rpdf; int32_t type, prim, mat, delta; } bvert;   /* BDPT_Vertex.py:10-21 */
typedef struct { bvert eye[BD_EYE_MAX], light[BD_LIGHT_MAX], sample, ltemp, etemp, lminustemp, eminustemp; } bpixel;

typedef struct { float view[16]; } orc_view;
typedef struct { int W, H; bpixel *px; float view[16]; } orc_bdpt;

orc_bdpt *orc_bdpt_create(int W, int H, const float *view16)
{
    orc_bdpt *b = (orc_bdpt *)calloc(1, sizeof(orc_bdpt));
    b->W = W; b->H = H;
    b->px = (bpixel *)calloc((size_t)W * H, sizeof(bpixel));
    memcpy(b->view, view16, sizeof(float) * 16);
    return b;
}
void orc_bdpt_destroy(orc_bdpt *b) { if (b) { free(b->px); free(b); } }

static float cosine_hemisphere_pdf(float c) { return fmax_(0.01f, c / M_PIf); }      /* UtilsFunc.py:348-350 */
static float remap0(float f) { return f == 0.0f ? 1.0f : f; }                        /* BDPT_RGB.py:89-93 */

/* brdf/Disney.py:43-63 */
static float disney_pdf(const orc_scene *s, v3 N, v3 Vv, v3 L, int mat_id)
{
    float pdf = 0.0f;
    float NDotL = vdot(N, L), NDotV = vdot(N, Vv);
    if ((NDotL > 0.0f) & (NDotV > 0.0f)) {
        const float inv_pi = (float)(1.0 / 3.1415956);
        const float *m = s->material + (size_t)mat_id * MAT_VEC;
        v3 H = vnormalized(vadd(L, Vv));
        float NDotH = vdot(H, N), LDotH = vdot(H, L);
        float metal = m[5], rough = m[6];
        float specularAlpha = fmax_(0.001f, rough);
        float Ds = gtr2(NDotH, specularAlpha);
        float diffuseRatio = 0.5f * (1.0f - metal);
        float specularRatio = 1.0f - diffuseRatio;
        float pdfGTR2 = Ds * NDotH;
        float pdfSpec = pdfGTR2 / (4.0f * fabs_(LDotH));
        pdf = diffuseRatio * inv_pi + specularRatio * pdfSpec;
    }
    return pdf;
}

static v3 camera_dir(const orc_scene *s, int i, int j, float jx, float jy)            /* Camera.py:130-142 */
{
    float x = ((float)i + jx - s->cx) / s->fx, y = ((float)j + jy - s->cy) / s->fy, z = -1.0f, w = 0.0f;
    const float *M = s->view_inv;
    float wx = ((M[0] * x + M[1] * y) + M[2] * z) + M[3] * w;
    float wy = ((M[4] * x + M[5] * y) + M[6] * z) + M[7] * w;
    float wz = ((M[8] * x + M[9] * y) + M[10] * z) + M[11] * w;
    return vnormalized(V(wx, wy, wz));
}

/* Camera.py:144-158 */
static v3 get_image_point(const orc_scene *s, const orc_bdpt *b, v3 p, int *u_o, int *v_o)
{
    const float *M = b->view;
    float px = ((M[0] * p.x + M[1] * p.y) + M[2] * p.z) + M[3] * 1.0f;
    float py = ((M[4] * p.x + M[5] * p.y) + M[6] * p.z) + M[7] * 1.0f;
    float pz = ((M[8] * p.x + M[9] * p.y) + M[10] * p.z) + M[11] * 1.0f;
    float fu = -px / pz * s->fx + s->cx, fv = -py / pz * s->fy + s->cy;
    /* int() of a NaN / out-of-range float is undefined in the reference too: treated as off-screen */
    int u = (fu > -2.0e9f && fu < 2.0e9f) ? (int)fu : -1;
    int v = (fv > -2.0e9f && fv < 2.0e9f) ? (int)fv : -1;
    v3 wi = V(0, 0, 0);
    if ((u < 0) | (u >= b->W) | (v < 0) | (v >= b->H) | (pz > 0.0f)) { u = -1; v = -1; }
    else wi = vsub(p, V(s->eye[0], s->eye[1], s->eye[2]));
    *u_o = u; *v_o = v;
    return vnormalized(wi);
}

/* shared tail of eye_path / light_path: BSDF sample at a surface vertex (BDPT_RGB.py:159-193, 255-289) */
typedef struct { v3 next_dir; float f_or_b, brdf, pdfFwd; } bsample;
/* ---- BDPT_SPEC (integrator/BDPT_SPEC.py): the same bidirectional tracer carrying ONE wavelength per pixel sample.  `spc` == NULL is
 * BDPT_RGB; otherwise the colours of BDPT_RGB become powers at spc->Lambda (replicated into the three components of the v3 the RGB code
 * carries, so that every product below is the reference's scalar product), and the places where BDPT_SPEC.py differs from BDPT_RGB.py in
 * more than that are marked "SPEC". */
typedef struct { const orc_spec *sp; float Lambda; } bd_spec;
#define BD_DIM_CONNECT_SPEC 256u          /* + 8 e: the seven numbers of sample_light() for the l == 1 connection at eye vertex e */
#define BD_DIM_LAMBDA 2u                   /* the sample's wavelength (BDPT_SPEC.py:668); dims 0, 1 are the camera jitter */
/* BDPT_SPEC.py:146-155 */
static float bd_light_power(const bd_spec *spc, v3 emission)
{
    float ret = 0.0f;
    const float scale = vnorm(emission);
    if (scale > 0.0f) {
        const v3 coff = r2s_fetch(spc->sp, vdivs(emission, scale));
        ret = spd_sample(&spc->sp->spd[0], spc->Lambda) * r2s_eval(coff, spc->Lambda) * scale;
    }
    return ret;
}
/* BDPT_SPEC.py:134-144 */
static float bd_reflect_power(const orc_scene *s, const bd_spec *spc, int mat_id)
{
    const float *m = s->material + (size_t)mat_id * MAT_VEC;
    const v3 mat_color = V(m[2], m[3], m[4]);
    if ((int)m[0] == MAT_LIGHT) return bd_light_power(spc, mat_color);
    return r2s_eval(r2s_fetch(spc->sp, srgb_to_lrgb(mat_color)), spc->Lambda);
}
/* what BDPT_RGB multiplies a path's throughput with at a surface of material mat_id: its linear colour / its power at the wavelength */
static v3 bd_reflect(const orc_scene *s, const bd_spec *spc, int mat_id)
{
    if (spc) { const float p = bd_reflect_power(s, spc, mat_id); return V(p, p, p); }
    const float *m = s->material + (size_t)mat_id * MAT_VEC;
    return srgb_to_lrgb(V(m[2], m[3], m[4]));
}

static bsample bd_sample(const orc_scene *s, const bd_spec *spc, v3 dir, v3 normal, v3 fnormal, int mat_id, int mat_type,
                         uint32_t seed, uint32_t pixel, uint32_t frame, uint32_t dim0, int32_t *delta)
{
    bsample r; r.next_dir = dir; r.f_or_b = 1.0f; r.brdf = 0.0f; r.pdfFwd = 0.0f;
    if (mat_type == MAT_GLASS) {
        if (spc) r.next_dir = glass_sample_lambda(dir, normal, spc->Lambda, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_GLASS), &r.f_or_b);      /* SPEC: BDPT_SPEC.py:241, 335 */
        else r.next_dir = glass_sample(s, dir, normal, mat_id, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_GLASS), &r.f_or_b);
        r.brdf = 1.0f; r.pdfFwd = 1.0f;
        *delta = 1;
    } else {
        float rnd[3] = { tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LOBE), tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R1),
                         tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R2) };
        r.next_dir = disney_sample(s, dir, fnormal, mat_id, rnd);
        r.f_or_b = 1.0f;
        r.brdf = disney_evaluate_pdf(s, fnormal, vneg(dir), r.next_dir, mat_id, &r.pdfFwd);
        *delta = 0;
    }
    return r;
}

/* BDPT_RGB.py:103-198 */
static int bd_eye_path(const orc_scene *s, const bd_spec *spc, bpixel *P, int i, int j, int H, uint32_t frame, uint32_t seed,
                       int32_t *stack, int stack_size, orc_stats *st)
{
    uint32_t pixel = (uint32_t)(i * H + j);
    bvert *eye = P->eye;
    v3 origin = V(s->eye[0], s->eye[1], s->eye[2]);
    float jx = 0.0f, jy = 0.0f;
    if (frame != 0) { jx = tm_rand(seed, pixel, frame, TM_DIM_JX) - 0.5f; jy = tm_rand(seed, pixel, frame, TM_DIM_JY) - 0.5f; }
    v3 dir = camera_dir(s, i, j, jx, jy);
    eye[0].pos = origin; eye[0].normal = dir; eye[0].beta = V(1, 1, 1); eye[0].fpdf = 1.0f; eye[0].type = VERTEX_LENS;
    int pre_depth = 0, depth = 1;
    float pdfFwd = 1.0f, pdfRev = 0.0f;
    v3 beta = V(1, 1, 1);
    while (depth < BD_EYE_MAX) {
        hit_t h = closet_hit(s, origin, dir, stack, stack_size, st);
        if (h.t < INF_VALUE) {
            v3 normal = h.nor, pos = h.pos;
            v3 fnormal = vscale(normal, signf(vdot(vneg(dir), h.gnor)));
            int mat_id = s->primitive[(size_t)h.prim * PRI_VEC + 2];
            const float *m = s->material + (size_t)mat_id * MAT_VEC;
            v3 mat_color = V(m[2], m[3], m[4]);
            int mat_type = (int)m[0];
            v3 to = vsub(pos, origin);
            float dist = fmax_(vnorm(to), 0.01f);
            float inv_dist2 = 1.0f / (dist * dist);
            to = vdivs(to, dist);
            bvert *e = &eye[depth];
            e->pos = pos; e->normal = normal; e->snormal = fnormal; e->wo = dir; e->rpdf = 0.0f; e->prim = h.prim; e->mat = mat_id;
            e->fpdf = pdfFwd * fabs_(vdot(to, eye[pre_depth].normal)) * inv_dist2;
            if (mat_type == MAT_LIGHT) {
                if (spc) e->beta = vmul(beta, bd_reflect(s, spc, mat_id));          /* SPEC: beta * reflect_power, no cosine (BDPT_SPEC.py:228) */
                else e->beta = vscale(vmul(beta, mat_color), fabs_(vdot(normal, dir)));
                e->type = VERTEX_LIGHT;
                depth += 1;
                break;
            } else {
                e->beta = vscale(beta, fabs_(vdot(dir, normal)));
                e->type = VERTEX_SURFACE;
            }
            v3 reflect_color = bd_reflect(s, spc, mat_id);
            bsample bs = bd_sample(s, spc, dir, normal, fnormal, mat_id, mat_type, seed, pixel, frame,
                                   BD_DIM_EYE + 8u * (uint32_t)depth, &e->delta);
            pdfFwd = bs.pdfFwd;
            if (pdfFwd > 0.0f) {
                if (mat_type == MAT_GLASS) {
                    pdfRev = 0.0f; pdfFwd = 0.0f;
                    beta = vmul(beta, vscale(reflect_color, bs.brdf));
                } else {
                    beta = vmul(beta, vdivs(vscale(vscale(reflect_color, bs.brdf), fabs_(vdot(normal, bs.next_dir))), pdfFwd));
                    pdfRev = disney_pdf(s, fnormal, bs.next_dir, vneg(dir), mat_id);
                }
                eye[pre_depth].rpdf = pdfRev * fabs_(vdot(to, e->normal)) * inv_dist2;
                if (!spc && bs.f_or_b < 0.0f) {            /* (SPEC has no extinction roulette) */
                    float R = m_exp(-h.t / m[6]);
                    if (tm_rand(seed, pixel, frame, BD_DIM_EYE + 8u * (uint32_t)depth + TM_SLOT_EXT) >= R) break;
                }
                depth += 1; pre_depth += 1;
                origin = offset_ray(pos, vscale(fnormal, signf(bs.f_or_b)));
                dir = bs.next_dir;
            } else break;
        } else break;
    }
    return depth;
}

/* Scene.py:430-474 */
static void bd_sample_light(const orc_scene *s, uint32_t seed, uint32_t pixel, uint32_t frame, uint32_t BD_DIM_LBASE,
                            v3 *pos, v3 *nor, v3 *dir, v3 *emission, int *prim, float *choice_pdf, float *dir_pdf)
{
    int lidx = (int)(tm_rand(seed, pixel, frame, BD_DIM_LBASE + 0) * (float)s->light_count);
    if (lidx >= s->light_count) lidx = s->light_count - 1;
    int lp = s->light[lidx];
    float a = tm_rand(seed, pixel, frame, BD_DIM_LBASE + 1), b = tm_rand(seed, pixel, frame, BD_DIM_LBASE + 2);
    v3 lpos, lnor;
    get_prim_random_point_normal(s, lp, a, b, &lpos, &lnor);
    int lmat = s->primitive[(size_t)lp * PRI_VEC + 2];
    const float *lm = s->material + (size_t)lmat * MAT_VEC;
    float area = get_prim_area(s, lp);
    *choice_pdf = 1.0f / ((float)s->light_count * area);
    lnor = vnormalized(lnor);
    v3 ld = cosine_sample_hemisphere(tm_rand(seed, pixel, frame, BD_DIM_LBASE + 3), tm_rand(seed, pixel, frame, BD_DIM_LBASE + 4));
    *dir_pdf = cosine_hemisphere_pdf(ld.z);
    *dir = inverse_transform(ld, lnor);
    *pos = lpos; *nor = lnor; *emission = V(lm[2], lm[3], lm[4]); *prim = lp;
    const int32_t *pr = s->primitive + (size_t)lp * PRI_VEC;
    if (pr[0] != PRIMITIVE_TRI) {                                   /* Scene.py:449-472 */
        const float *sh = s->shape + (size_t)pr[1] * SHA_VEC;
        const int st = (int)sh[0];
        if (st == SHAPE_SPOT) {
            const float scale = sh[6];
            *dir_pdf = 1.0f;
            float r, phi;
            map_to_disk(tm_rand(seed, pixel, frame, BD_DIM_LBASE + 5), tm_rand(seed, pixel, frame, BD_DIM_LBASE + 6), &r, &phi);
            const float r1 = scale * m_tan(sh[4]), r2 = scale * m_tan(sh[5]);
            r *= r2;
            if (r > r1) *emission = vscale(*emission, 1.0f - (r - r1) / (r2 - r1));
            const v3 sp = V(r * m_cos(phi), r * m_sin(phi), m_sqrt(fmax_(0.0f, scale * scale - r * r)));
            *dir = inverse_transform(sp, lnor);
        } else if (st == SHAPE_LASER) {
            *choice_pdf = 1.0f / (float)s->light_count;
            const float r = sh[4];
            const float phi = tm_rand(seed, pixel, frame, BD_DIM_LBASE + 5) * M_PIf * 2.0f;
            v3 sp = V(r * m_cos(phi), r * m_sin(phi), 0.0f);
            sp = inverse_transform(sp, lnor);
            *dir = lnor;
            *dir_pdf = 1.0f;
            *pos = vadd(lpos, sp);
        }
    }
}

/* BDPT_RGB.py:200-294 */
static int bd_light_path(const orc_scene *s, const bd_spec *spc, bpixel *P, int i, int j, int H, uint32_t frame, uint32_t seed,
                         int32_t *stack, int stack_size, orc_stats *st)
{
    uint32_t pixel = (uint32_t)(i * H + j);
    bvert *light = P->light;
    v3 lpos, lnor, ldir, emission; int lprim; float choice_pdf, dir_pdf;
    bd_sample_light(s, seed, pixel, frame, BD_DIM_LSTART, &lpos, &lnor, &ldir, &emission, &lprim, &choice_pdf, &dir_pdf);
    float light_pdf = choice_pdf;
    if (spc) { const float pw = bd_light_power(spc, emission) / light_pdf; light[0].beta = V(pw, pw, pw); }      /* SPEC: BDPT_SPEC.py:284 */
    else light[0].beta = vdivs(emission, light_pdf);
    light[0].pos = lpos; light[0].normal = lnor;
    light[0].fpdf = light_pdf; light[0].rpdf = 0.0f; light[0].wo = ldir; light[0].type = VERTEX_LIGHT;
    int pre_depth = 0, depth = 1;
    float pdfFwd = dir_pdf, pdfRev = 0.0f;
    v3 beta = spc ? light[0].beta                                               /* SPEC: beta = power[0], no cosine (BDPT_SPEC.py:294) */
                  : vscale(vdivs(emission, light_pdf), fabs_(vdot(lnor, ldir)));
    v3 origin = lpos, dir = ldir;
    while (depth < BD_LIGHT_MAX) {
        hit_t h = closet_hit(s, origin, dir, stack, stack_size, st);
        if (h.t < INF_VALUE) {
            v3 normal = h.nor, pos = h.pos;
            v3 fnormal = vscale(normal, signf(vdot(vneg(dir), h.gnor)));
            int mat_id = s->primitive[(size_t)h.prim * PRI_VEC + 2];
            const float *m = s->material + (size_t)mat_id * MAT_VEC;
            (void)m;
            int mat_type = (int)m[0];
            if (mat_type == MAT_LIGHT) break;
            bvert *L = &light[depth];
            L->pos = pos; L->normal = normal; L->snormal = fnormal; L->beta = vscale(beta, fabs_(vdot(dir, normal)));
            L->wo = dir; L->fpdf = pdfFwd; L->rpdf = 0.0f; L->type = VERTEX_SURFACE; L->prim = h.prim; L->mat = mat_id;
            v3 to = vsub(pos, light[pre_depth].pos);
            float dist = vnorm(to);
            float inv_dist2 = 1.0f / (dist * dist);
            to = vdivs(to, dist);
            L->fpdf *= fabs_(vdot(to, light[pre_depth].normal)) * inv_dist2;
            v3 reflect_color = bd_reflect(s, spc, mat_id);
            bsample bs = bd_sample(s, spc, dir, normal, fnormal, mat_id, mat_type, seed, pixel, frame,
                                   BD_DIM_LIGHT + 8u * (uint32_t)depth, &L->delta);
            pdfFwd = bs.pdfFwd;
            if (pdfFwd > 0.0f) {
                if (mat_type == MAT_GLASS) {
                    pdfRev = 0.0f; pdfFwd = 0.0f;
                    beta = vmul(beta, vscale(reflect_color, bs.brdf));
                } else {
                    beta = vmul(beta, vdivs(vscale(vscale(reflect_color, bs.brdf), fabs_(vdot(normal, bs.next_dir))), pdfFwd));
                    pdfRev = disney_pdf(s, fnormal, bs.next_dir, vneg(dir), mat_id);
                }
                light[pre_depth].rpdf = pdfRev * fabs_(vdot(to, L->normal)) * inv_dist2;
                if (!spc && bs.f_or_b < 0.0f) {
                    float R = m_exp(-h.t / m[6]);
                    if (tm_rand(seed, pixel, frame, BD_DIM_LIGHT + 8u * (uint32_t)depth + TM_SLOT_EXT) >= R) break;
                }
                origin = offset_ray(pos, vscale(fnormal, signf(bs.f_or_b)));
                dir = bs.next_dir;
                depth += 1; pre_depth += 1;
            } else break;
        } else break;
    }
    return depth;
}

/* BDPT_RGB.py:300-479 */
static float bd_mis_weight(const orc_scene *s, const orc_bdpt *B, bpixel *P, int e, int l)
{
    bvert *light = P->light, *eye = P->eye;
    float weight_sum = 0.0f;
    if (l + e != 2) {
        if (l > 0) P->ltemp = light[l - 1];
        if (e > 0) P->etemp = eye[e - 1];
        if (l > 1) P->lminustemp = light[l - 2];
        if (e > 1) P->eminustemp = eye[e - 2];
        if (l == 1) light[0] = P->sample;
        else if (e == 1) eye[0] = P->sample;
        if (l > 0) light[l - 1].delta = 0;
        if (e > 0) eye[e - 1].delta = 0;

        if (e > 0) {
            if (l == 0) {
                float pdfPos = 1.0f / get_prim_area(s, eye[e - 1].prim);
                float pdfChoice = 1.0f / (float)s->light_count;
                eye[e - 1].rpdf = pdfPos * pdfChoice;
            } else if (l == 1) {
                if (eye[e - 1].type == VERTEX_SURFACE) {
                    v3 to = vsub(eye[e - 1].pos, light[0].pos);
                    float dist = vnorm(to);
                    to = vdivs(to, dist);
                    float pdfDir = cosine_hemisphere_pdf(fabs_(vdot(to, light[0].normal)));
                    float LdotN = fabs_(vdot(to, light[0].normal));
                    eye[e - 1].rpdf = pdfDir * LdotN / (dist * dist);
                } else eye[e - 1].rpdf = 1.0f;
            } else {
                v3 wi = vsub(light[l - 2].pos, light[l - 1].pos);
                v3 wo = vsub(eye[e - 1].pos, light[l - 1].pos);
                float dist = vnorm(wo);
                wi = vnormalized(wi); wo = vnormalized(wo);
                float pdf = 1.0f;
                int mat_id = light[l - 1].mat;
                if (mat_id == MAT_DISNEY) pdf = disney_pdf(s, light[l - 1].snormal, wi, wo, mat_id);
                eye[e - 1].rpdf = pdf * fabs_(vdot(light[l - 1].normal, wo)) / (dist * dist);
            }
        }
        if (l > 0) {
            if (e > 1) {
                if (eye[e - 1].type == VERTEX_SURFACE) {
                    v3 wi = vsub(eye[e - 2].pos, eye[e - 1].pos);
                    v3 wo = vsub(light[l - 1].pos, eye[e - 1].pos);
                    float dist = vnorm(wo);
                    wi = vnormalized(wi); wo = vnormalized(wo);
                    float pdf = 1.0f;
                    int mat_id = eye[e - 1].mat;
                    if (mat_id == MAT_DISNEY) pdf = disney_pdf(s, eye[e - 1].snormal, wi, wo, mat_id);
                    light[l - 1].rpdf = pdf * fabs_(vdot(eye[e - 1].normal, wo)) / (dist * dist);
                } else light[l - 1].rpdf = 1.0f;
            } else {
                v3 to = vsub(eye[0].pos, light[l - 1].pos);
                float dist = vnorm(to);
                to = vdivs(to, dist);
                v3 axis = V(B->view[8], B->view[9], B->view[10]);       /* Camera.py:126-127 */
                float LdotN = vdot(to, axis);
                light[l - 1].rpdf = LdotN / (dist * dist);
            }
        }
        if (e > 1) {
            if (l == 0) {
                v3 to = vsub(eye[e - 2].pos, eye[e - 1].pos);
                float dist = vnorm(to);
                to = vdivs(to, dist);
                float pdfDir = cosine_hemisphere_pdf(fabs_(vdot(to, eye[e - 1].normal)));
                float LdotN = vdot(to, eye[e - 1].normal);
                eye[e - 2].rpdf = fabs_(pdfDir * LdotN) / (dist * dist);
            } else {
                if (eye[e - 1].type == VERTEX_SURFACE) {
                    v3 wi = vsub(light[l - 1].pos, eye[e - 1].pos);
                    v3 wo = vsub(eye[e - 2].pos, eye[e - 1].pos);
                    float dist = vnorm(wo);
                    wi = vnormalized(wi); wo = vnormalized(wo);
                    int mat_id = eye[e - 1].mat;
                    float pdf = disney_pdf(s, eye[e - 1].snormal, wi, wo, mat_id);
                    eye[e - 2].rpdf = pdf / (dist * dist);
                    if (eye[e - 2].type == VERTEX_SURFACE) eye[e - 2].rpdf *= fabs_(vdot(eye[e - 1].normal, wo));
                } else eye[e - 2].rpdf = 1.0f;
            }
        }
        if (l > 1) {
            if (eye[e - 1].type != VERTEX_LIGHT) {
                v3 wi = vsub(eye[e - 1].pos, light[l - 1].pos);
                v3 wo = vsub(light[l - 2].pos, light[l - 1].pos);
                float dist = vnorm(wo);
                wi = vnormalized(wi); wo = vnormalized(wo);
                float pdf = 1.0f;
                int mat_id = light[l - 1].mat;
                if (mat_id == MAT_DISNEY) pdf = disney_pdf(s, light[l - 1].normal, wi, wo, mat_id);
                light[l - 2].rpdf = pdf / (dist * dist);
                if (light[l - 2].type == VERTEX_SURFACE) light[l - 2].rpdf *= fabs_(vdot(light[l - 1].normal, wo));
            } else light[l - 2].rpdf = 1.0f;
        }

        float weight = 1.0f;
        for (int k = e - 1; k > 0; k--) {
            weight *= remap0(eye[k].rpdf) / remap0(eye[k].fpdf);
            if ((eye[k].delta == 0) & (eye[k - 1].delta == 0)) weight_sum += weight;
        }
        weight = 1.0f;
        for (int k = l - 1; k >= 0; k--) {
            weight *= remap0(light[k].rpdf) / remap0(light[k].fpdf);
            if (k == 0) { if (light[k].delta == 0) weight_sum += weight; }
            else if ((light[k].delta == 0) & (light[k - 1].delta == 0)) weight_sum += weight;
        }

        /* give back the original data; copies to index -1 (Taichi: padding) are skipped */
        if (l - 1 >= 0) light[l - 1] = P->ltemp;
        eye[e - 1] = P->etemp;
        if (l > 0 && l - 2 >= 0) light[l - 2] = P->lminustemp;
        if (e > 0 && e - 2 >= 0) eye[e - 2] = P->eminustemp;
    }
    return 1.0f / (1.0f + weight_sum);
}

/* BDPT_RGB.py:481-592; returns radiance * misweight and the pixel it belongs to (-1: none) */
static v3 bd_connect_path(const orc_scene *s, const bd_spec *spc, const orc_bdpt *B, bpixel *P, int i, int j, int e, int l, uint32_t frame,
                          uint32_t seed, int32_t *stack, int stack_size, orc_stats *st, int *nu, int *nv)
{
    bvert *eye = P->eye, *light = P->light;
    uint32_t pixel = (uint32_t)(i * B->H + j);
    v3 radiance = V(0, 0, 0);
    *nu = i; *nv = j;
    if (l == 0) {
        if (eye[e - 1].type == VERTEX_LIGHT) radiance = eye[e - 1].beta;
    } else if (e == 1) {
        int prim = light[l - 1].prim;
        v3 surface = light[l - 1].pos;
        v3 wi = get_image_point(s, B, surface, nu, nv);
        v3 origin = V(s->eye[0], s->eye[1], s->eye[2]);
        int mat_id = light[l - 1].mat;
        v3 snormal = light[l - 1].snormal;
        float NdotL = vdot(wi, snormal);
        if ((*nu >= 0) & (light[l - 1].delta != 1) & (NdotL < 0.0f) & (light[l - 1].type == VERTEX_SURFACE)) {
            int hit_prim;
            float t = closet_hit_shadow(s, origin, wi, stack, stack_size, &hit_prim, st);
            if (hit_prim == prim) {
                float pdf;
                float brdf = disney_evaluate_pdf(s, snormal, vneg(light[l - 1].wo), vneg(wi), mat_id, &pdf);
                if (pdf > 0.0f) {
                    float G = fabs_(NdotL) / (t * t);
                    radiance = vdivs(vscale(vmul(vscale(light[l - 1].beta, G), bd_reflect(s, spc, mat_id)), brdf), pdf);
                    P->sample.pos = origin; P->sample.wo = wi; P->sample.type = VERTEX_LENS; P->sample.fpdf = 1.0f;
                }
            }
        }
    } else if (l == 1) {
        v3 surface = offset_ray(eye[e - 1].pos, eye[e - 1].snormal);
        int mat_id = eye[e - 1].mat;
        if (spc && eye[e - 1].delta != 1) {
            /* SPEC: BDPT_SPEC.py:605-630 -- sample_light() (Scene.py:430-474: its direction sample is drawn and dropped) instead of
             * sample_li(surface); an emitter without a surface (spot, laser) can never be the shadow ray's hit, so it contributes
             * through the light path only */
            v3 light_pos, light_normal, light_dir_unused, light_emission; int light_prim; float light_choice_pdf, light_dir_pdf;
            bd_sample_light(s, seed, pixel, frame, BD_DIM_CONNECT_SPEC + 8u * (uint32_t)e, &light_pos, &light_normal, &light_dir_unused,
                            &light_emission, &light_prim, &light_choice_pdf, &light_dir_pdf);
            const v3 wi = vnormalized(vsub(surface, light_pos));
            const float NdotLl = vdot(wi, light_normal), NdotLe = vdot(wi, eye[e - 1].snormal);
            int shadow_prim;
            const float t = closet_hit_shadow(s, surface, vneg(wi), stack, stack_size, &shadow_prim, st);
            if ((shadow_prim == light_prim) & (t > EPS_UF)) {
                const float light_pdf = light_choice_pdf;
                float pdf;
                const float brdf = disney_evaluate_pdf(s, eye[e - 1].snormal, vneg(eye[e - 1].wo), vneg(wi), mat_id, &pdf);
                if (pdf > 0.0f) {
                    const float G = fabs_(NdotLe * NdotLl) / (t * t);
                    v3 c = vdivs(vscale(vscale(eye[e - 1].beta, G), brdf), pdf);
                    c = vmul(c, bd_reflect(s, spc, mat_id));
                    c = vscale(c, bd_light_power(spc, light_emission));
                    radiance = vdivs(c, light_pdf);
                }
                P->sample.pos = light_pos; P->sample.wo = wi; P->sample.type = VERTEX_LIGHT; P->sample.fpdf = light_pdf;
                P->sample.prim = light_prim; P->sample.normal = light_normal; P->sample.snormal = light_normal;
            }
        } else if (eye[e - 1].delta != 1) {
            /* Scene.py:477-518 sample_li(surface) */
            uint32_t d0 = BD_DIM_CONNECT + 4u * (uint32_t)e;
            int lidx = (int)(tm_rand(seed, pixel, frame, d0) * (float)s->light_count);
            if (lidx >= s->light_count) lidx = s->light_count - 1;
            int light_prim = s->light[lidx];
            v3 light_pos, light_normal;
            get_prim_random_point_normal(s, light_prim, tm_rand(seed, pixel, frame, d0 + 1), tm_rand(seed, pixel, frame, d0 + 2),
                                         &light_pos, &light_normal);
            int lmat = s->primitive[(size_t)light_prim * PRI_VEC + 2];
            const float *lm = s->material + (size_t)lmat * MAT_VEC;
            v3 light_emission = V(lm[2], lm[3], lm[4]);
            float light_choice_pdf = 1.0f / ((float)s->light_count * get_prim_area(s, light_prim));
            light_normal = vnormalized(light_normal);
            v3 wi = vsub(surface, light_pos);
            float light_dist = vnorm(wi);
            wi = vdivs(wi, light_dist);
            light_emission = vscale(light_emission, light_shape_visible(s, light_prim, wi, light_normal, light_dist, &light_choice_pdf));
            float NdotLl = vdot(wi, light_normal);
            float NdotLe = vdot(wi, eye[e - 1].snormal);
            int shadow_prim;
            float t = closet_hit_shadow(s, surface, vneg(wi), stack, stack_size, &shadow_prim, st);
            if ((shadow_prim == light_prim) & (t > EPS_UF)) {
                float light_pdf = light_choice_pdf;
                float pdf;
                float brdf = disney_evaluate_pdf(s, eye[e - 1].snormal, vneg(eye[e - 1].wo), vneg(wi), mat_id, &pdf);
                if (pdf > 0.0f) {
                    float G = fabs_(NdotLe * NdotLl) / (t * t);
                    const float *m = s->material + (size_t)mat_id * MAT_VEC;
                    v3 c = vdivs(vscale(vscale(eye[e - 1].beta, G), brdf), pdf);
                    c = vmul(c, srgb_to_lrgb(V(m[2], m[3], m[4])));
                    c = vmul(c, light_emission);
                    radiance = vdivs(c, light_pdf);
                }
                P->sample.pos = light_pos; P->sample.wo = wi; P->sample.type = VERTEX_LIGHT; P->sample.fpdf = light_pdf;
                P->sample.prim = light_prim; P->sample.normal = light_normal; P->sample.snormal = light_normal;
            }
        }
    } else {
        if ((light[l - 1].delta != 1) & (eye[e - 1].delta != 1) & (eye[e - 1].type == VERTEX_SURFACE) & (light[l - 1].type == VERTEX_SURFACE)) {
            int primE = eye[e - 1].prim, mat_idE = eye[e - 1].mat, mat_idL = light[l - 1].mat;
            v3 surfaceE = eye[e - 1].pos, surfaceL = light[l - 1].pos;
            v3 dir = vsub(surfaceE, surfaceL);
            float dist = vnorm(dir);
            dir = vdivs(dir, dist);
            float NdotLl = vdot(dir, light[l - 1].snormal), NdotLe = vdot(dir, eye[e - 1].snormal);
            int shadow_prim;
            float t = closet_hit_shadow(s, surfaceL, dir, stack, stack_size, &shadow_prim, st);
            if ((shadow_prim == primE) & (t > EPS_UF)) {
                float lpdf, epdf;
                float brdfL = disney_evaluate_pdf(s, light[l - 1].snormal, vneg(light[l - 1].wo), dir, mat_idL, &lpdf);
                float brdfE = disney_evaluate_pdf(s, eye[e - 1].snormal, vneg(eye[e - 1].wo), vneg(dir), mat_idE, &epdf);
                if ((brdfL > 0.0f) & (brdfE > 0.0f)) {
                    float G = fabs_(NdotLe * NdotLl) / (dist * dist);
                    const float *mE = s->material + (size_t)mat_idE * MAT_VEC, *mL = s->material + (size_t)mat_idL * MAT_VEC;
                    v3 c = vmul(vscale(eye[e - 1].beta, G), light[l - 1].beta);
                    c = vdivs(vscale(c, brdfL), lpdf);
                    c = vdivs(vscale(c, brdfE), epdf);
                    (void)mE; (void)mL;
                    c = vmul(c, bd_reflect(s, spc, mat_idE));
                    radiance = vmul(c, bd_reflect(s, spc, mat_idL));
                }
            }
        }
    }
    float misweight = 1.0f;
    if ((radiance.x > 0.0f) & (radiance.y > 0.0f) & (radiance.z > 0.0f)) misweight = bd_mis_weight(s, B, P, e, l);
    return vscale(radiance, misweight);
}

/* BDPT_RGB.py:595-642.  radiance: [W*H*3] scratch (cleared here); hdr: running mean.  Single thread:
 * light-tracing contributions land on other pixels (`radiance[eye_new_pos] += r_path`). */
static int bdpt_render_common(const orc_scene *s, const orc_spec *spec, orc_bdpt *B, uint32_t frame_begin, int frame_count, uint32_t seed,
                              int stack_size, float *radiance, float *hdr, orc_stats *stats)
{
    int W = B->W, H = B->H;
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(stack_size + 2));
    orc_stats st; memset(&st, 0, sizeof(st));
    /* debugging aids (tools/dbg/bdpt_counts.py): ORC_BDPT_DUMP = file for the connection rays per pixel of the last frame,
     * ORC_BDPT_PIXEL = pixel whose connections are printed */
    const char *dbg_dump = getenv("ORC_BDPT_DUMP"), *dbg_pixel_s = getenv("ORC_BDPT_PIXEL");
    const long dbg_pixel = dbg_pixel_s ? atol(dbg_pixel_s) : -1;
    const char *dbg_big_s = getenv("ORC_BDPT_BIG"); const double dbg_big = dbg_big_s ? atof(dbg_big_s) : 0.0;      /* print every connection above this value */
    for (int f = 0; f < frame_count; f++) {
        uint32_t frame = frame_begin + (uint32_t)f;
        memset(radiance, 0, sizeof(float) * (size_t)W * H * 3);
        for (long p = 0; p < (long)W * H; p++) {
            bpixel *P = &B->px[p];
            for (int e = 0; e < BD_EYE_MAX; e++) { P->eye[e].beta = V(0, 0, 0); P->eye[e].type = VERTEX_NONE; P->eye[e].fpdf = 0.0f; P->eye[e].rpdf = 0.0f; }
            for (int l = 0; l < BD_LIGHT_MAX; l++) { P->light[l].beta = V(0, 0, 0); P->light[l].type = VERTEX_NONE; P->light[l].fpdf = 0.0f; P->light[l].rpdf = 0.0f; }
        }
        for (long p = 0; p < (long)W * H; p++) {
            int i = (int)(p / H), j = (int)(p % H);
            bpixel *P = &B->px[p];
            st.paths++;
            /* SPEC: one wavelength per pixel sample, lambda_min + lambda_range * size * rand (BDPT_SPEC.py:668: up to one step beyond lambda_max,
             * where the sensor reads zero) */
            bd_spec spc_v; const bd_spec *spc = NULL;
            if (spec) { spc_v.sp = spec; spc_v.Lambda = spec->s_min + (spec->s_range * (float)spec->n_sensor) * tm_rand(seed, (uint32_t)p, frame, BD_DIM_LAMBDA); spc = &spc_v; }
            int eye_depth = bd_eye_path(s, spc, P, i, j, H, frame, seed, stack, stack_size, &st);
            int light_depth = bd_light_path(s, spc, P, i, j, H, frame, seed, stack, stack_size, &st);
            const uint64_t shadow_before = st.rays_shadow;
            for (int e = 1; e <= eye_depth; e++) {
                for (int l = 0; l <= light_depth; l++) {
                    int depth = l + e - 2;
                    if (((l == 1) & (e == 1)) | (depth < 0) | (depth > BD_MAX_DEPTH)) continue;
                    int nu, nv;
                    const uint64_t sh0__ = st.rays_shadow;
                    v3 r = bd_connect_path(s, spc, B, P, i, j, e, l, frame, seed, stack, stack_size, &st, &nu, &nv);
                    if (spc) {            /* SPEC: AddSplat (BDPT_SPEC.py:178-181): the sensor's response at the wavelength, as clamped sRGB, times the range */
                        const v3 xyz = sensor_sample(spec, spc->Lambda);
                        const float range = spec->s_max - spec->s_min;
                        const float cr = ((float)3.240479 * xyz.x + (float)-1.537150 * xyz.y) + (float)-0.498535 * xyz.z;
                        const float cg = ((float)-0.969256 * xyz.x + (float)1.875991 * xyz.y) + (float)0.041556 * xyz.z;
                        const float cb = ((float)0.055648 * xyz.x + (float)-0.204043 * xyz.y) + (float)1.057311 * xyz.z;
                        r = V((clampf(cr, 0.0f, 1000.0f) * range) * r.x, (clampf(cg, 0.0f, 1000.0f) * range) * r.x, (clampf(cb, 0.0f, 1000.0f) * range) * r.x);
                    }
                    if (dbg_big > 0.0 && (r.x > dbg_big || r.y > dbg_big || r.z > dbg_big)) {
                        fprintf(stderr, "BIG pixel %ld frame %u e %d l %d r = %g %g %g eye_depth %d light_depth %d\n", p, frame, e, l, r.x, r.y, r.z, eye_depth, light_depth);
                        for (int k = 0; k < eye_depth; k++) fprintf(stderr, "   eye[%d] type %d prim %d mat %d delta %d beta %g fpdf %g rpdf %g pos %g %g %g\n", k, P->eye[k].type, P->eye[k].prim, P->eye[k].mat, P->eye[k].delta, P->eye[k].beta.x, P->eye[k].fpdf, P->eye[k].rpdf, P->eye[k].pos.x, P->eye[k].pos.y, P->eye[k].pos.z);
                        for (int k = 0; k < light_depth; k++) fprintf(stderr, "   light[%d] type %d prim %d mat %d delta %d beta %g fpdf %g rpdf %g pos %g %g %g\n", k, P->light[k].type, P->light[k].prim, P->light[k].mat, P->light[k].delta, P->light[k].beta.x, P->light[k].fpdf, P->light[k].rpdf, P->light[k].pos.x, P->light[k].pos.y, P->light[k].pos.z);
                    }
                    if (dbg_pixel == p && f == frame_count - 1) {
                        fprintf(stderr, "pixel %ld frame %u e %d l %d: shadow rays %d, r = %g %g %g\n", p, frame, e, l, (int)(st.rays_shadow - sh0__), r.x, r.y, r.z);
                        if ((e == 1 && l == 2) || (e == 2 && l == 0 && light_depth < 2)) {      /* (the first connection a pixel sample makes) */
                            for (int k = 0; k < BD_EYE_MAX; k++) fprintf(stderr, "   eye[%d] type %d prim %d mat %d delta %d beta %g %g %g fpdf %g rpdf %g pos %g %g %g\n", k, P->eye[k].type, P->eye[k].prim, P->eye[k].mat, P->eye[k].delta, P->eye[k].beta.x, P->eye[k].beta.y, P->eye[k].beta.z, P->eye[k].fpdf, P->eye[k].rpdf, P->eye[k].pos.x, P->eye[k].pos.y, P->eye[k].pos.z);
                            for (int k = 0; k < BD_LIGHT_MAX; k++) fprintf(stderr, "   light[%d] type %d prim %d mat %d delta %d beta %g %g %g fpdf %g rpdf %g pos %g %g %g\n", k, P->light[k].type, P->light[k].prim, P->light[k].mat, P->light[k].delta, P->light[k].beta.x, P->light[k].beta.y, P->light[k].beta.z, P->light[k].fpdf, P->light[k].rpdf, P->light[k].pos.x, P->light[k].pos.y, P->light[k].pos.z);
                        }
                    }
                    long q = (e == 1) ? ((nu >= 0) ? (long)nu * H + nv : -1) : p;
                    if (q >= 0) { radiance[3 * q] += r.x; radiance[3 * q + 1] += r.y; radiance[3 * q + 2] += r.z; }
                }
            }
            if (dbg_dump && f == frame_count - 1) {
                static FILE *fp = NULL;
                if (p == 0) { if (fp) fclose(fp); fp = fopen(dbg_dump, "w"); }
                if (fp) { fprintf(fp, "%ld %d %d %d\n", p, (int)(st.rays_shadow - shadow_before), eye_depth, light_depth); if (p == (long)W * H - 1) { fclose(fp); fp = NULL; } }
            }
        }
        float ff = (float)(int32_t)frame, coff = 1.0f / (ff + 1.0f);
        for (long k = 0; k < (long)W * H * 3; k++) hdr[k] = radiance[k] * coff + hdr[k] * (1.0f - coff);
    }
    free(stack);
    if (stats) *stats = st;
    return 0;
}
int orc_bdpt_render(const orc_scene *s, orc_bdpt *B, uint32_t frame_begin, int frame_count, uint32_t seed,
                    int stack_size, float *radiance, float *hdr, orc_stats *stats)
{ return bdpt_render_common(s, NULL, B, frame_begin, frame_count, seed, stack_size, radiance, hdr, stats); }
/* integrator/BDPT_SPEC.py:660-691 */
int orc_bdpt_spec_render(const orc_scene *s, const orc_spec *spec, orc_bdpt *B, uint32_t frame_begin, int frame_count, uint32_t seed,
                         int stack_size, float *radiance, float *hdr, orc_stats *stats)
{ return bdpt_render_common(s, spec, B, frame_begin, frame_count, seed, stack_size, radiance, hdr, stats); }

/* ---- scalar KAT entry points (tests compare the HIP kernels' device functions) -------- */
void orc_kat_disney(const float *mat10, const float *N, const float *Vv, const float *L, float *out2)
{
    orc_scene s; memset(&s, 0, sizeof(s)); s.material = (float *)mat10;
    float pdf; float f = disney_evaluate_pdf(&s, V(N[0], N[1], N[2]), V(Vv[0], Vv[1], Vv[2]), V(L[0], L[1], L[2]), 0, &pdf);
    out2[0] = f; out2[1] = pdf;
}
void orc_kat_disney_sample(const float *mat10, const float *dir, const float *N, const float *rnd3, float *out3)
{
    orc_scene s; memset(&s, 0, sizeof(s)); s.material = (float *)mat10;
    v3 r = disney_sample(&s, V(dir[0], dir[1], dir[2]), V(N[0], N[1], N[2]), 0, rnd3);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void orc_kat_glass_sample(const float *mat10, const float *dir, const float *N, float prob, float *out4)
{
    orc_scene s; memset(&s, 0, sizeof(s)); s.material = (float *)mat10;
    float fb; v3 r = glass_sample(&s, V(dir[0], dir[1], dir[2]), V(N[0], N[1], N[2]), 0, prob, &fb);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = fb;
}
void orc_kat_offset_ray(const float *p, const float *n, float *out3)
{
    v3 r = offset_ray(V(p[0], p[1], p[2]), V(n[0], n[1], n[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
int orc_kat_slabs(const float *o, const float *d, const float *mn, const float *mx)
{ return slabs(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), V(mn[0], mn[1], mn[2]), V(mx[0], mx[1], mx[2])); }
int32_t orc_kat_morton3d(float x, float y, float z) { return morton3d(x, y, z); }
float orc_kat_rand(uint32_t seed, uint32_t pixel, uint32_t frame, uint32_t dim) { return tm_rand(seed, pixel, frame, dim); }
/* math KATs: evaluate the shared header on the host for bit-comparison with the device */
void orc_kat_math(int fn, const float *x, const float *y, float *out, int n)
{
    for (int i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = tm_sin(x[i]); break;
            case 1: out[i] = tm_cos(x[i]); break;
            case 2: out[i] = tm_exp(x[i]); break;
            case 3: out[i] = tm_log(x[i]); break;
            case 4: out[i] = tm_pow(x[i], y[i]); break;
            case 5: out[i] = tm_atan2(x[i], y[i]); break;
            case 6: out[i] = tm_acos(x[i]); break;
            case 7: out[i] = tm_sqrt(x[i]); break;
            case 8: out[i] = x[i] / y[i]; break;
            default: out[i] = 0.0f;
        }
    }
}
/* Further per-function entry points for tests/test_refkat.py: the values the REFERENCE'S OWN SOURCE TEXT computes (executed through the
 * stand-in of tools/refkat, build container only) are compared with these.  `in` / `out` are plain float rows, see the test for the layouts. */
void orc_kat_util(int which, const float *in, float *out)
{
    switch (which) {
        case 0: { v3 r = cosine_sample_hemisphere(in[0], in[1]); out[0] = r.x; out[1] = r.y; out[2] = r.z; break; }
        case 1: map_to_disk(in[0], in[1], &out[0], &out[1]); break;
        case 2: out[0] = power_heuristic(in[0], in[1]); break;
        case 3: { v3 r = inverse_transform(V(in[0], in[1], in[2]), V(in[3], in[4], in[5])); out[0] = r.x; out[1] = r.y; out[2] = r.z; break; }
        case 4: { v3 r = srgb_to_lrgb(V(in[0], in[1], in[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; break; }
        case 5: { v3 r = lrgb_to_srgb(V(in[0], in[1], in[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; break; }
        case 6: out[0] = tone_aces1(in[0]); out[1] = tone_aces1(in[1]); out[2] = tone_aces1(in[2]); break;
        case 7: { float suc; v3 r = refract_(V(in[0], in[1], in[2]), V(in[3], in[4], in[5]), in[6], &suc); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = suc; break; }
        case 8: out[0] = schlick(in[0], in[1]); break;
        case 9: out[0] = gtr2(in[0], in[1]); break;
        case 10: out[0] = smithg_ggx(in[0], in[1]); break;
        case 11: out[0] = schlick_fresnel(in[0]); break;
        case 12: { orc_scene s; memset(&s, 0, sizeof(s)); s.material = (float *)in;
                   out[0] = disney_pdf(&s, V(in[10], in[11], in[12]), V(in[13], in[14], in[15]), V(in[16], in[17], in[18]), 0); break; }
        case 13: { float fb; v3 r = glass_sample_lambda(V(in[0], in[1], in[2]), V(in[3], in[4], in[5]), in[6], in[7], &fb);
                   out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = fb; break; }
        case 14: { int32_t a, b; memcpy(&a, in, 4); memcpy(&b, in + 1, 4); out[0] = (float)common_upper_bits(a, b); break; }
        case 15: { orc_scene s; memset(&s, 0, sizeof(s)); memcpy(s.view_inv, in, sizeof(float) * 16); s.fx = in[16]; s.fy = in[17]; s.cx = in[18]; s.cy = in[19];
                   v3 r = camera_dir(&s, (int)in[20], (int)in[21], in[22], in[23]); out[0] = r.x; out[1] = r.y; out[2] = r.z; break; }
        default: break;
    }
}
/* Spectral per-function entry points (tests/test_refkat.py against tests/golden/refkat_spec.npz: values computed by the reference's own
 * spectrum modules, sky/Sky.py and integrator/PT_Spec.py text through tools/refkat).  which:
 *   0 Spectrum.sample            in: k (0 d65 1 white 2 red 3 green), Lambda          out: 1
 *   1 HeroSample.sample          in: k, Lambda0                                        out: 4
 *   2 HeroSample.sample_xyz      in: Lambda0  (sensor = PathTrace.sample, :131-139)    out: x[4], y[4], z[4]
 *   3 Rgb2Spec.fetch             in: rgb3                                              out: coff3
 *   4 Rgb2Spec.eval              in: coff3, Lambda                                     out: 1
 *   5 HeroSample.srgb_to_spec    in: srgb3, Lambda0                                    out: 4
 *   6 HeroSample.sky_sample      in: theta, gamma, Lambda0                             out: 4
 *   7 PathTrace.emission_to_rad  in: emission3, Lambda                                 out: 4
 *   8 HeroSample.get_extinction_hero in: Lambda0, t                                    out: 4
 *   9 PathTrace.AddSplat         in: spec4, Lambda0, coff, hdr3                        out: hdr3
 *  10 PathTrace.get_spec_power   in: mat10, Lambda                                     out: 4
 *  11 HeroSample.get_rnd_hero    in: ti.random(), Lambda0                              out: index, Lambda */
void orc_kat_spec(const orc_spec *sp, int which, const float *in, float *out)
{
    v4s r; int nout4 = 0;
    switch (which) {
        case 0: out[0] = spd_sample(&sp->spd[(int)in[0]], in[1]); break;
        case 1: r = hero_sample(&sp->spd[(int)in[0]], in[1]); nout4 = 1; break;
        case 2: for (int k = 0; k < HERO_N; k++) { v3 c = sensor_sample(sp, in[0] + (float)k * HERO_LAMBDA_STEP); out[k] = c.x; out[4 + k] = c.y; out[8 + k] = c.z; } break;
        case 3: { v3 c = r2s_fetch(sp, V(in[0], in[1], in[2])); out[0] = c.x; out[1] = c.y; out[2] = c.z; break; }
        case 4: out[0] = r2s_eval(V(in[0], in[1], in[2]), in[3]); break;
        case 5: r = srgb_to_spec(sp, V(in[0], in[1], in[2]), in[3]); nout4 = 1; break;
        case 6: for (int k = 0; k < HERO_N; k++) out[k] = sky_radiance(sp, in[0], in[1], in[2] + (float)k * HERO_LAMBDA_STEP); break;
        case 7: r = emission_to_rad(sp, V(in[0], in[1], in[2]), in[3]); nout4 = 1; break;
        case 8: for (int k = 0; k < HERO_N; k++) out[k] = m_exp(-in[1] / (in[0] + (float)k * HERO_LAMBDA_STEP)); break;
        case 9: { for (int k = 0; k < HERO_N; k++) r.v[k] = in[k]; out[0] = in[6]; out[1] = in[7]; out[2] = in[8]; spec_add_splat(sp, r, in[4], in[5], out); break; }
        case 10: { orc_scene s; memset(&s, 0, sizeof(s)); s.material = (float *)in; r = get_spec_power(&s, sp, 0, in[10]); nout4 = 1; break; }
        case 11: { const int index = (int)(in[0] * (float)HERO_N); out[0] = (float)index; out[1] = in[1] + (float)index * HERO_LAMBDA_STEP; break; }
        default: break;
    }
    if (nout4) for (int k = 0; k < HERO_N; k++) out[k] = r.v[k];
}
int orc_uses_libm(void)
{
#ifdef ORACLE_LIBM
    return 1;
#else
    return 0;
#endif
}
