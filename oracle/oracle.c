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

typedef struct {
    const orc_scene *s; int W, H; uint32_t frame_begin; int frame_count; uint32_t seed;
    int max_depth, stack_size; float *hdr;
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
            v3 rad = pt_rgb_pixel(jb->s, i, j, jb->H, frame, jb->seed, jb->max_depth, stack, jb->stack_size, &jb->st);
            /* PT_RGB.py:134-136 */
            float ff = (float)(int32_t)frame;
            float coff = 1.0f / (ff + 1.0f);
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
int orc_pt_rgb_render(const orc_scene *s, int W, int H, uint32_t frame_begin, int frame_count,
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
        jb->seed = seed; jb->max_depth = max_depth; jb->stack_size = stack_size; jb->hdr = hdr;
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
static bsample bd_sample(const orc_scene *s, v3 dir, v3 normal, v3 fnormal, int mat_id, int mat_type,
                         uint32_t seed, uint32_t pixel, uint32_t frame, uint32_t dim0, int32_t *delta)
{
    bsample r; r.next_dir = dir; r.f_or_b = 1.0f; r.brdf = 0.0f; r.pdfFwd = 0.0f;
    if (mat_type == MAT_GLASS) {
        r.next_dir = glass_sample(s, dir, normal, mat_id, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_GLASS), &r.f_or_b);
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
static int bd_eye_path(const orc_scene *s, bpixel *P, int i, int j, int H, uint32_t frame, uint32_t seed,
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
                e->beta = vscale(vmul(beta, mat_color), fabs_(vdot(normal, dir)));
                e->type = VERTEX_LIGHT;
                depth += 1;
                break;
            } else {
                e->beta = vscale(beta, fabs_(vdot(dir, normal)));
                e->type = VERTEX_SURFACE;
            }
            v3 reflect_color = srgb_to_lrgb(mat_color);
            bsample bs = bd_sample(s, dir, normal, fnormal, mat_id, mat_type, seed, pixel, frame,
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
                if (bs.f_or_b < 0.0f) {
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
static void bd_sample_light(const orc_scene *s, uint32_t seed, uint32_t pixel, uint32_t frame,
                            v3 *pos, v3 *nor, v3 *dir, v3 *emission, int *prim, float *choice_pdf, float *dir_pdf)
{
    int lidx = (int)(tm_rand(seed, pixel, frame, BD_DIM_LSTART + 0) * (float)s->light_count);
    if (lidx >= s->light_count) lidx = s->light_count - 1;
    int lp = s->light[lidx];
    float a = tm_rand(seed, pixel, frame, BD_DIM_LSTART + 1), b = tm_rand(seed, pixel, frame, BD_DIM_LSTART + 2);
    v3 lpos, lnor;
    get_prim_random_point_normal(s, lp, a, b, &lpos, &lnor);
    int lmat = s->primitive[(size_t)lp * PRI_VEC + 2];
    const float *lm = s->material + (size_t)lmat * MAT_VEC;
    float area = get_prim_area(s, lp);
    *choice_pdf = 1.0f / ((float)s->light_count * area);
    lnor = vnormalized(lnor);
    v3 ld = cosine_sample_hemisphere(tm_rand(seed, pixel, frame, BD_DIM_LSTART + 3), tm_rand(seed, pixel, frame, BD_DIM_LSTART + 4));
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
            map_to_disk(tm_rand(seed, pixel, frame, BD_DIM_LSTART + 5), tm_rand(seed, pixel, frame, BD_DIM_LSTART + 6), &r, &phi);
            const float r1 = scale * m_tan(sh[4]), r2 = scale * m_tan(sh[5]);
            r *= r2;
            if (r > r1) *emission = vscale(*emission, 1.0f - (r - r1) / (r2 - r1));
            const v3 sp = V(r * m_cos(phi), r * m_sin(phi), m_sqrt(fmax_(0.0f, scale * scale - r * r)));
            *dir = inverse_transform(sp, lnor);
        } else if (st == SHAPE_LASER) {
            *choice_pdf = 1.0f / (float)s->light_count;
            const float r = sh[4];
            const float phi = tm_rand(seed, pixel, frame, BD_DIM_LSTART + 5) * M_PIf * 2.0f;
            v3 sp = V(r * m_cos(phi), r * m_sin(phi), 0.0f);
            sp = inverse_transform(sp, lnor);
            *dir = lnor;
            *dir_pdf = 1.0f;
            *pos = vadd(lpos, sp);
        }
    }
}

/* BDPT_RGB.py:200-294 */
static int bd_light_path(const orc_scene *s, bpixel *P, int i, int j, int H, uint32_t frame, uint32_t seed,
                         int32_t *stack, int stack_size, orc_stats *st)
{
    uint32_t pixel = (uint32_t)(i * H + j);
    bvert *light = P->light;
    v3 lpos, lnor, ldir, emission; int lprim; float choice_pdf, dir_pdf;
    bd_sample_light(s, seed, pixel, frame, &lpos, &lnor, &ldir, &emission, &lprim, &choice_pdf, &dir_pdf);
    float light_pdf = choice_pdf;
    light[0].pos = lpos; light[0].normal = lnor; light[0].beta = vdivs(emission, light_pdf);
    light[0].fpdf = light_pdf; light[0].rpdf = 0.0f; light[0].wo = ldir; light[0].type = VERTEX_LIGHT;
    int pre_depth = 0, depth = 1;
    float pdfFwd = dir_pdf, pdfRev = 0.0f;
    v3 beta = vscale(vdivs(emission, light_pdf), fabs_(vdot(lnor, ldir)));
    v3 origin = lpos, dir = ldir;
    while (depth < BD_LIGHT_MAX) {
        hit_t h = closet_hit(s, origin, dir, stack, stack_size, st);
        if (h.t < INF_VALUE) {
            v3 normal = h.nor, pos = h.pos;
            v3 fnormal = vscale(normal, signf(vdot(vneg(dir), h.gnor)));
            int mat_id = s->primitive[(size_t)h.prim * PRI_VEC + 2];
            const float *m = s->material + (size_t)mat_id * MAT_VEC;
            v3 mat_color = V(m[2], m[3], m[4]);
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
            v3 reflect_color = srgb_to_lrgb(mat_color);
            bsample bs = bd_sample(s, dir, normal, fnormal, mat_id, mat_type, seed, pixel, frame,
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
                if (bs.f_or_b < 0.0f) {
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
static v3 bd_connect_path(const orc_scene *s, const orc_bdpt *B, bpixel *P, int i, int j, int e, int l, uint32_t frame,
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
                    const float *m = s->material + (size_t)mat_id * MAT_VEC;
                    radiance = vdivs(vscale(vmul(vscale(light[l - 1].beta, G), srgb_to_lrgb(V(m[2], m[3], m[4]))), brdf), pdf);
                    P->sample.pos = origin; P->sample.wo = wi; P->sample.type = VERTEX_LENS; P->sample.fpdf = 1.0f;
                }
            }
        }
    } else if (l == 1) {
        v3 surface = offset_ray(eye[e - 1].pos, eye[e - 1].snormal);
        int mat_id = eye[e - 1].mat;
        if (eye[e - 1].delta != 1) {
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
                    c = vmul(c, srgb_to_lrgb(V(mE[2], mE[3], mE[4])));
                    radiance = vmul(c, srgb_to_lrgb(V(mL[2], mL[3], mL[4])));
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
int orc_bdpt_render(const orc_scene *s, orc_bdpt *B, uint32_t frame_begin, int frame_count, uint32_t seed,
                    int stack_size, float *radiance, float *hdr, orc_stats *stats)
{
    int W = B->W, H = B->H;
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(stack_size + 2));
    orc_stats st; memset(&st, 0, sizeof(st));
    /* debugging aids (tools/dbg/bdpt_counts.py): ORC_BDPT_DUMP = file for the connection rays per pixel of the last frame,
     * ORC_BDPT_PIXEL = pixel whose connections are printed */
    const char *dbg_dump = getenv("ORC_BDPT_DUMP"), *dbg_pixel_s = getenv("ORC_BDPT_PIXEL");
    const long dbg_pixel = dbg_pixel_s ? atol(dbg_pixel_s) : -1;
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
            int eye_depth = bd_eye_path(s, P, i, j, H, frame, seed, stack, stack_size, &st);
            int light_depth = bd_light_path(s, P, i, j, H, frame, seed, stack, stack_size, &st);
            const uint64_t shadow_before = st.rays_shadow;
            for (int e = 1; e <= eye_depth; e++) {
                for (int l = 0; l <= light_depth; l++) {
                    int depth = l + e - 2;
                    if (((l == 1) & (e == 1)) | (depth < 0) | (depth > BD_MAX_DEPTH)) continue;
                    int nu, nv;
                    const uint64_t sh0__ = st.rays_shadow;
                    v3 r = bd_connect_path(s, B, P, i, j, e, l, frame, seed, stack, stack_size, &st, &nu, &nv);
                    if (dbg_pixel == p && f == frame_count - 1) {
                        fprintf(stderr, "pixel %ld frame %u e %d l %d: shadow rays %d, r = %g %g %g\n", p, frame, e, l, (int)(st.rays_shadow - sh0__), r.x, r.y, r.z);
                        if (e == 1 && l == 0) {
                            for (int k = 0; k < BD_EYE_MAX; k++) fprintf(stderr, "   eye[%d] type %d prim %d mat %d delta %d beta %g %g %g fpdf %g rpdf %g pos %g %g %g\n", k, P->eye[k].type, P->eye[k].prim, P->eye[k].mat, P->eye[k].delta, P->eye[k].beta.x, P->eye[k].beta.y, P->eye[k].beta.z, P->eye[k].fpdf, P->eye[k].rpdf, P->eye[k].pos.x, P->eye[k].pos.y, P->eye[k].pos.z);
                            for (int k = 0; k < BD_LIGHT_MAX; k++) fprintf(stderr, "   light[%d] type %d prim %d mat %d delta %d beta %g %g %g fpdf %g rpdf %g\n", k, P->light[k].type, P->light[k].prim, P->light[k].mat, P->light[k].delta, P->light[k].beta.x, P->light[k].beta.y, P->light[k].beta.z, P->light[k].fpdf, P->light[k].rpdf);
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
int orc_uses_libm(void)
{
#ifdef ORACLE_LIBM
    return 1;
#else
    return 0;
#endif
}
