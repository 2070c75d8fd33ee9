// tirt_device.h -- device-side restatement (HIP, gfx950) of the reference's @ti.func helpers:
// UtilsFunc.py (slabs, sampling, ONB, GTR/Smith/Schlick, refract, offset_ray, colour),
// Scene.py (intersect_tri/prim, areas, light sampling), brdf/Disney.py, brdf/Glass.py,
// texture/Texture.py.  fp32 operation order follows the reference expression by
// expression (this whole library is compiled with -ffp-contract=off); transcendental
// functions come from tirt_math.h so that results are bit-reproducible against the CPU
// oracle.  Each function cites the reference lines it follows.
#pragma once
#include <hip/hip_runtime.h>
#include "tirt_math.h"

#define TD __device__ __forceinline__

namespace tirt {

// ---- reference constants ---------------------------------------------------------------
constexpr int MAT_VEC = 10, VER_VEC = 9, PRI_VEC = 3, SHA_VEC = 10, NOD_VEC = 11, CPN_VEC = 9;  // SceneData.py:33-38
constexpr int SHAPE_SPHERE = 1, SHAPE_SPOT = 3, SHAPE_LASER = 4;                                 // SceneData.py:40-44
constexpr int PRIMITIVE_TRI = 1;                                                                // SceneData.py:47
constexpr int MAT_DISNEY = 0, MAT_GLASS = 1, MAT_LIGHT = 2;                                      // SceneData.py:50-52
constexpr float INF_VALUE = 1000000.0f;             // UtilsFunc.py:38
constexpr float PI_UF = (float)3.1415956;           // UtilsFunc.py:37 (sic, quirk B1)
constexpr float PI_SCENE = (float)3.1415926;        // Scene.py:319,343; integrator/PT_RGB.py:129-130

struct v3 { float x, y, z; };
TD v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
TD v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
TD v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
TD v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
TD v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
TD v3 operator/(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
TD v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
TD float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
TD v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
TD float norm(v3 a) { return tm_sqrt(dot(a, a)); }
// taichi Vector.normalized(): invlen = 1 / norm; invlen * self
TD v3 normalized(v3 a) { float inv = 1.0f / norm(a); return a * inv; }
TD float absf(float x) { return x < 0.0f ? -x : x; }
TD float minf(float a, float b) { return a < b ? a : b; }
TD float maxf(float a, float b) { return a > b ? a : b; }
TD float clampf(float x, float lo, float hi) { return minf(hi, maxf(lo, x)); }
TD float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
TD float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
TD v3 mix3(v3 a, v3 b, float t) { return V(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)); }

// ---- scene view passed to kernels by value ---------------------------------------------------
struct SceneView {
    const float *vertex;     // [nv*9]
    const int *primitive;    // [n*3]
    const float *material;   // [nm*10]
    const float *shape;      // [ns*10]
    const int *light;        // [nl]
    const int *env;          // [w*h]
    const float *mat_lrgb;   // [nm*3] srgb_to_lrgb(material colour), filled on device at upload
    const float4 *shade_rec; // [n*8] one 128-byte line per primitive: what k_shade needs to shade a hit on it (k_shade_records)
    int n, light_count, env_w, env_h;
    float env_power;
};
struct CameraView { float view_inv[12]; float eye[3]; float fx, fy, cx, cy; };

TD v3 vtx_pos(const SceneView &s, int i) { const float *p = s.vertex + (size_t)i * VER_VEC; return V(p[0], p[1], p[2]); }
TD v3 vtx_nor(const SceneView &s, int i) { const float *p = s.vertex + (size_t)i * VER_VEC; return V(p[3], p[4], p[5]); }
TD v3 vtx_uv(const SceneView &s, int i)  { const float *p = s.vertex + (size_t)i * VER_VEC; return V(p[6], p[7], p[8]); }

// ---- UtilsFunc.py:494-523 slabs, with 1/d hoisted out (same quotient, computed once per ray) ----
struct RayCtx { float ox, oy, oz, dx, dy, dz, idx, idy, idz; };
TD RayCtx make_ray(v3 o, v3 d)
{
    RayCtx r; r.ox = o.x; r.oy = o.y; r.oz = o.z; r.dx = d.x; r.dy = d.y; r.dz = d.z;
    r.idx = 1.0f / d.x; r.idy = 1.0f / d.y; r.idz = 1.0f / d.z;
    return r;
}
TD void slab_axis(float o, float d, float ood, float mn, float mx, float &tmin, float &tmax, int &ret)
{
    if (absf(d) < 0.000001f) {
        if ((o < mn) | (o > mx)) ret = 0;
    } else {
        float t1 = (mn - o) * ood;
        float t2 = (mx - o) * ood;
        if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
        if (t1 > tmin) tmin = t1;
        if (t2 < tmax) tmax = t2;
        if (tmin > tmax) ret = 0;
    }
}
// returns the reference's 0/1 and, through tnear, the entry distance it computed
TD int slabs(const RayCtx &r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float &tnear)
{
    int ret = 1; float tmin = 0.0f, tmax = INF_VALUE;
    slab_axis(r.ox, r.dx, r.idx, mnx, mxx, tmin, tmax, ret);
    slab_axis(r.oy, r.dy, r.idy, mny, mxy, tmin, tmax, ret);
    slab_axis(r.oz, r.dz, r.idz, mnz, mxz, tmin, tmax, ret);
    tnear = tmin;
    return ret;
}
// Branch-free form for rays with |d| >= 1e-6 on every axis (all but axis-parallel rays).
// Same products and the same min/max selections as slab_axis, so tmin/tmax are the same
// floats; tmin only grows and tmax only shrinks across the axes, so the reference's
// per-axis `tmin > tmax` checks are equivalent to the single check at the end (no NaN can
// arise: 1/d is finite).
TD bool ray_has_parallel_axis(const RayCtx &r)
{ return (absf(r.dx) < 0.000001f) || (absf(r.dy) < 0.000001f) || (absf(r.dz) < 0.000001f); }
TD int slabs_fast(const RayCtx &r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float &tnear)
{
    const float ax = (mnx - r.ox) * r.idx, bx = (mxx - r.ox) * r.idx;
    const float ay = (mny - r.oy) * r.idy, by = (mxy - r.oy) * r.idy;
    const float az = (mnz - r.oz) * r.idz, bz = (mxz - r.oz) * r.idz;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)),
                                       __builtin_fmaxf(__builtin_fminf(az, bz), 0.0f));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)),
                                       __builtin_fminf(__builtin_fmaxf(az, bz), INF_VALUE));
    tnear = tmin;
    return (tmin > tmax) ? 0 : 1;
}

// ---- Scene.py:603-638 intersect_tri on a packed (v0, E1 = v1-v0, E2 = v2-v0) triangle ------------
TD float intersect_tri_packed(v3 origin, v3 direction, v3 v0, v3 E1, v3 E2, float &u, float &v)
{
    float t = INF_VALUE; u = 0.0f; v = 0.0f;
    v3 P = cross(direction, E2);
    float det = dot(E1, P);
    v3 T;
    if (det > 0.0f) T = origin - v0;
    else { T = v0 - origin; det = -det; }
    if (det > 0.0f) {
        u = dot(T, P);
        if ((u >= 0.0f) & (u <= det)) {
            v3 Q = cross(T, E1);
            v = dot(direction, Q);
            if ((v >= 0.0f) & (u + v <= det)) {
                t = dot(E2, Q);
                float fInvDet = 1.0f / det;
                t *= fInvDet; u *= fInvDet; v *= fInvDet;
            }
        }
    }
    return t;
}

// ---- Scene.py:565-596 / 653-665 sphere branch of intersect_prim(_any) ----------------------------
TD float intersect_sphere(v3 origin, v3 direction, v3 centre, float r, float &c_out)
{
    float hit_t = INF_VALUE;
    v3 oc = centre - origin;
    float dis_oc_square = dot(oc, oc);
    float dis_op = dot(direction, oc);
    float dis_cp = tm_sqrt(dis_oc_square - dis_op * dis_op);
    c_out = 0.0f;
    if (dis_cp < r) {
        float a = dot(direction, direction);
        float b = -2.0f * dis_op;
        float c = dis_oc_square - r * r;
        hit_t = (-b - tm_sqrt(b * b - 4.0f * a * c)) / 2.0f / a;
        c_out = c;
    }
    return hit_t;
}

// ---- hit attributes of the winning candidate (Scene.py:537-561, 565-596) -----------------------
struct HitAttr { v3 pos, gnor, nor, tex; };
TD HitAttr hit_attributes(const SceneView &s, v3 origin, v3 direction, int prim, float t, float u, float v)
{
    HitAttr h; h.pos = h.gnor = h.nor = h.tex = V(0.0f, 0.0f, 0.0f);
    const int *pr = s.primitive + (size_t)prim * PRI_VEC;
    v3 gn = V(0.0f, 0.0f, 0.0f), nn = gn;
    if (pr[0] == PRIMITIVE_TRI) {
        int vi = pr[1];
        float a = 1.0f - u - v, b = u, c = v;
        v3 v1 = vtx_pos(s, vi), v2 = vtx_pos(s, vi + 1), v3_ = vtx_pos(s, vi + 2);
        v3 n1 = vtx_nor(s, vi), n2 = vtx_nor(s, vi + 1), n3 = vtx_nor(s, vi + 2);
        v3 t1 = vtx_uv(s, vi), t2 = vtx_uv(s, vi + 1), t3 = vtx_uv(s, vi + 2);
        v3 v13 = v3_ - v1, v12 = v2 - v1;
        gn = cross(v12, v13);
        h.pos = (v1 * a + v2 * b) + v3_ * c;
        h.tex = (t1 * a + t2 * b) + t3 * c;
        nn = (n1 * a + n2 * b) + n3 * c;
    } else {
        const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;
        if ((int)sh[0] == SHAPE_SPHERE) {
            float c;
            (void)intersect_sphere(origin, direction, V(sh[1], sh[2], sh[3]), sh[4], c);
            h.pos = origin + direction * t;
            nn = V(h.pos.x - c, h.pos.y - c, h.pos.z - c);          // quirk B3
            gn = nn;
        }
    }
    h.gnor = normalized(gn); h.nor = normalized(nn);
    return h;
}

// The same from the 128-byte shading record of the primitive (one cache line instead of a 12-byte primitive row plus three
// 36-byte vertex rows in three more lines; exact copies of the same floats, so the same results):
//   triangles: (v1.xyz, bits mat) (v2.xyz, bits PRIMITIVE_TRI) (v3.xyz, -) (n1.xyz, -) (n2.xyz, -) (n3.xyz, -) - -
//   shapes   : (centre.xyz, bits mat) (radius, shape type, -, bits 2)
// uv is not carried: the path tracer does not use it (the reference's albedo textures are unused, PT_RGB.py:86).
TD HitAttr hit_attributes_rec(const float4 *rec, v3 origin, v3 direction, int prim, float t, float u, float v, int &mat_id)
{
    HitAttr h; h.pos = h.gnor = h.nor = h.tex = V(0.0f, 0.0f, 0.0f);
    const float4 *r = rec + (size_t)prim * 8;
    const float4 r0 = r[0], r1 = r[1];
    mat_id = __float_as_int(r0.w);
    v3 gn = V(0.0f, 0.0f, 0.0f), nn = gn;
    if (__float_as_int(r1.w) == PRIMITIVE_TRI) {
        const float4 r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5];
        float a = 1.0f - u - v, b = u, c = v;
        v3 v1 = V(r0.x, r0.y, r0.z), v2 = V(r1.x, r1.y, r1.z), v3_ = V(r2.x, r2.y, r2.z);
        v3 n1 = V(r3.x, r3.y, r3.z), n2 = V(r4.x, r4.y, r4.z), n3 = V(r5.x, r5.y, r5.z);
        v3 v13 = v3_ - v1, v12 = v2 - v1;
        gn = cross(v12, v13);
        h.pos = (v1 * a + v2 * b) + v3_ * c;
        nn = (n1 * a + n2 * b) + n3 * c;
    } else if ((int)r1.y == SHAPE_SPHERE) {
        float c;
        (void)intersect_sphere(origin, direction, V(r0.x, r0.y, r0.z), r1.x, c);
        h.pos = origin + direction * t;
        nn = V(h.pos.x - c, h.pos.y - c, h.pos.z - c);          // quirk B3
        gn = nn;
    }
    h.gnor = normalized(gn); h.nor = normalized(nn);
    return h;
}

// ---- colour (UtilsFunc.py:76-94, 113-120) ----------------------------------------------------------
TD float srgb_to_lrgb1(float c) { return (c < 0.04045f) ? c / 12.92f : tm_pow((c + 0.055f) / 1.055f, 2.4f); }
TD v3 srgb_to_lrgb(v3 c) { return V(srgb_to_lrgb1(c.x), srgb_to_lrgb1(c.y), srgb_to_lrgb1(c.z)); }
TD float lrgb_to_srgb1(float c)
{
    const float e = (float)(1.0 / 2.4);
    float o = (c < 0.0031308f) ? c * 12.92f : 1.055f * tm_pow(c, e) - 0.055f;
    return clampf(o, 0.0f, 1.0f);
}
TD float tone_aces1(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return clampf((x * (a * x + b)) / (x * (c * x + d) + e), 0.0f, 1.0f);
}

// ---- sampling helpers (UtilsFunc.py:352-387) ---------------------------------------------------------
TD v3 cosine_sample_hemisphere(float u1, float u2)
{
    const float two_pi = (float)(2.0 * 3.1415956);
    float r = tm_sqrt(u1);
    float phi = two_pi * u2;
    v3 p;
    float sn, cs; tm_sincos(phi, &sn, &cs);
    p.x = r * cs;
    p.y = r * sn;
    p.z = tm_sqrt(maxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
    return normalized(p);
}
TD v3 inverse_transform(v3 dir, v3 N)
{
    v3 Normal = normalized(N);
    v3 Binormal;
    if (absf(Normal.x) > absf(Normal.z)) Binormal = V(-Normal.y, Normal.x, 0.0f);
    else Binormal = V(0.0f, -Normal.z, Normal.y);
    Binormal = normalized(Binormal);
    v3 Tangent = normalized(cross(Binormal, Normal));
    return (Tangent * dir.x + Binormal * dir.y) + Normal * dir.z;
}

// ---- microfacet terms (UtilsFunc.py:393-438) -------------------------------------------------------------
TD float schlick_fresnel(float u) { float m = clampf(1.0f - u, 0.0f, 1.0f); float m2 = m * m; return m2 * m2 * m; }
TD float gtr2(float NDotH, float a) { float a2 = a * a; float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH; return a2 / (PI_UF * t * t); }
TD float smithg_ggx(float NDotv, float alphaG) { float a = alphaG * alphaG, b = NDotv * NDotv; return 1.0f / (NDotv + tm_sqrt(a + b - a * b)); }
TD v3 refract_(v3 InRay, v3 N, float eta, float &suc)
{
    suc = -1.0f;
    float N_DOT_I = dot(N, InRay);
    float k = 1.0f - eta * eta * (1.0f - N_DOT_I * N_DOT_I);
    v3 R = V(0.0f, 0.0f, 0.0f);
    if (k > 0.0f) { R = InRay * eta - N * (eta * N_DOT_I + tm_sqrt(k)); suc = 1.0f; }
    return R;
}
TD float schlick(float cosine, float ior)
{
    float r0 = (1.0f - ior) / (1.0f + ior);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * tm_pow(1.0f - cosine, 5.0f);
}
TD v3 reflect_(v3 I, v3 N) { return I - N * (2.0f * dot(N, I)); }
TD float power_heuristic(float a, float b) { float t = a * a; return t / (b * b + t); }

// ---- UtilsFunc.py:440-461 ------------------------------------------------------------------------------
TD float offset_ray1(float p, float n)
{
    const float int_scale = 256.0f, float_scale = (float)(1.0 / 2048.0), origin = (float)(1.0 / 256.0);
    int i_of = (int)(int_scale * n);
    int i_p = (int)tm_f2u(p);
    if (p < 0.0f) i_p = i_p - i_of; else i_p = i_p + i_of;
    float f_p = tm_u2f((uint32_t)i_p);
    return (absf(p) < origin) ? p + float_scale * n : f_p;
}
TD v3 offset_ray(v3 p, v3 n) { return V(offset_ray1(p.x, n.x), offset_ray1(p.y, n.y), offset_ray1(p.z, n.z)); }

// ---- brdf/Disney.py:17-40 -----------------------------------------------------------------------------
TD v3 disney_sample(const float *m, v3 dir, v3 N, float probability, float r1, float r2)
{
    // Both lobes of the reference (diffuse: cosine_sample_hemisphere + inverse_transform, specular: GTR2
    // half vector + inverse_transform + reflect) take one sin/cos pair of a lobe-specific angle and one change
    // of basis around N.  Written so that a wave with lanes in both lobes evaluates the expensive shared
    // pieces (sincos, the three normalisations of the basis) once; every lane still performs exactly the
    // reference's operations for its lobe.
    float metal = m[5], rough = m[6];
    float diffuseRatio = 0.5f * (1.0f - metal);
    float specularAlpha = maxf(0.001f, rough);
    const bool diffuse = probability < diffuseRatio;
    const float two_pi = (float)(2.0 * 3.1415956);
    const float phi = diffuse ? two_pi * r2 : r1 * 2.0f * PI_UF;       // UtilsFunc.py:352-361 / Disney.py:28
    float sinPhi, cosPhi; tm_sincos(phi, &sinPhi, &cosPhi);
    v3 local;
    if (diffuse) {
        float r = tm_sqrt(r1);
        v3 p;
        p.x = r * cosPhi;
        p.y = r * sinPhi;
        p.z = tm_sqrt(maxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
        local = normalized(p);
    } else {
        float cosTheta = tm_sqrt((1.0f - r2) / (1.0f + (specularAlpha * specularAlpha - 1.0f) * r2));
        float sinTheta = tm_sqrt(1.0f - (cosTheta * cosTheta));
        local = V(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
    }
    const v3 world = inverse_transform(local, N);
    return diffuse ? world : reflect_(dir, world);
}
// ---- brdf/Disney.py:65-108 -----------------------------------------------------------------------------
TD float disney_evaluate_pdf(const float *m, v3 N, v3 Vv, v3 L, float &pdf)
{
    float outputC = 0.0f; pdf = -1.0f;
    float NDotL = dot(N, L), NDotV = dot(N, Vv);
    if ((NDotL > 0.0f) & (NDotV > 0.0f)) {
        const float inv_pi = (float)(1.0 / 3.1415956);
        v3 H = normalized(L + Vv);
        float NDotH = dot(H, N), LDotH = dot(H, L);
        float metal = m[5], rough = m[6];
        float Cspec0 = mixf(0.04f, 1.0f, metal);
        float Csheen = 0.5f;
        float FL = schlick_fresnel(NDotL), FV = schlick_fresnel(NDotV);
        float Fd90 = 0.5f + 2.0f * LDotH * LDotH * rough;
        float Fd = mixf(1.0f, Fd90, FL) * mixf(1.0f, Fd90, FV);
        float specularAlpha = maxf(0.001f, rough);
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
        float pdfSpec = pdfGTR2 / (4.0f * absf(LDotH));
        float pdfDiff = inv_pi;                               // quirk B4
        pdf = diffuseRatio * pdfDiff + specularRatio * pdfSpec;
    }
    return outputC;
}
// ---- brdf/Glass.py:9-34 ----------------------------------------------------------------------------------
TD v3 glass_sample(const float *m, v3 dir, v3 N, float probability, float &f_or_b)
{
    v3 w_out = dir;
    float cos_theta_i = dot(w_out, N);
    float ior = m[5];
    float eta = ior;
    f_or_b = 1.0f;
    float R = probability + 1.0f;
    if (cos_theta_i > 0.0f) N = -N;
    else { cos_theta_i = -cos_theta_i; eta = 1.0f / ior; }
    float suc;
    v3 next_dir = refract_(w_out, N, eta, suc);
    if (suc > 0.0f) R = schlick(cos_theta_i, ior);
    if (probability < R) next_dir = reflect_(w_out, N);
    else f_or_b = -1.0f;
    return next_dir;
}

// ---- Scene.py:324-350 ------------------------------------------------------------------------------------
TD float get_prim_area(const SceneView &s, int index)
{
    float ret = 0.0f;
    const int *pr = s.primitive + (size_t)index * PRI_VEC;
    if (pr[0] == PRIMITIVE_TRI) {
        v3 v1 = vtx_pos(s, pr[1]), v2 = vtx_pos(s, pr[1] + 1), v3_ = vtx_pos(s, pr[1] + 2);
        float a = norm(v1 - v2), b = norm(v1 - v3_), c = norm(v3_ - v2);
        float sum = (a + b + c) * 0.5f;
        ret = tm_sqrt(sum * (sum - a) * (sum - b) * (sum - c));
    } else {
        const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;
        int st = (int)sh[0];
        if (st == SHAPE_SPHERE || st == SHAPE_SPOT || st == SHAPE_LASER) { float r = sh[4]; ret = r * r * PI_SCENE; }   // quirk B2
    }
    return ret;
}
// ---- Scene.py:315-322 ------------------------------------------------------------------------------------
TD v3 uniform_sample_sphere(float u1, float u2)
{
    const float two_pi = (float)(2.0 * 3.1415926);
    float z = 1.0f - 2.0f * u1;
    float r = tm_sqrt(clampf(1.0f - z * z, 0.0f, 1.0f));
    float phi = two_pi * u2;
    float sn, cs; tm_sincos(phi, &sn, &cs);
    return V(r * cs, r * sn, z);
}
// ---- Scene.py:381-420 ------------------------------------------------------------------------------------
TD void get_prim_random_point_normal(const SceneView &s, int index, float a, float b, v3 &pos, v3 &nor)
{
    pos = V(0.0f, 0.0f, 0.0f); v3 normal = pos;
    const int *pr = s.primitive + (size_t)index * PRI_VEC;
    if (pr[0] == PRIMITIVE_TRI) {
        v3 v1 = vtx_pos(s, pr[1]), v2 = vtx_pos(s, pr[1] + 1), v3_ = vtx_pos(s, pr[1] + 2);
        v3 n1 = vtx_nor(s, pr[1]), n2 = vtx_nor(s, pr[1] + 1), n3 = vtx_nor(s, pr[1] + 2);
        if (a + b > 1.0f) { a = 1.0f - a; b = 1.0f - b; }
        pos = (v1 + (v3_ - v1) * a) + (v2 - v1) * b;
        normal = normalized((n1 * (1.0f - a - b) + n2 * a) + n3 * b);
    } else {
        const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;
        if ((int)sh[0] == SHAPE_SPHERE) {
            float r = sh[4];
            normal = uniform_sample_sphere(a, b);
            pos = V(sh[1], sh[2], sh[3]) + normal * r;
        } else if ((int)sh[0] == SHAPE_SPOT || (int)sh[0] == SHAPE_LASER) {      // Scene.py:413-418: the shape's own point and normal
            normal = V(sh[7], sh[8], sh[9]);
            pos = V(sh[1], sh[2], sh[3]);
        }
    }
    nor = normalized(normal);
}
// ---- Scene.py:491-516: what sample_li adds for the two shape emitters that have no surface -- the factor `visable` on the
// emission (spot: 1 inside the cone of half-angle x1, falling linearly to 0 at x2, measured between the light's normal and the
// direction to the shaded point; laser: 1 within `radius` of the beam's axis, else 0) and, for the laser, light_choice_pdf =
// 1 / light_count ----
TD float light_shape_visible(const SceneView &s, int light_prim, v3 light_dir, v3 light_normal, float light_dist, float &choice_pdf)
{
    float visable = 1.0f;
    const int *pr = s.primitive + (size_t)light_prim * PRI_VEC;
    if (pr[0] != PRIMITIVE_TRI) {
        const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;
        const int st = (int)sh[0];
        if (st == SHAPE_SPOT) {
            const float NdotL = absf(dot(light_dir, light_normal));
            const float x1 = sh[4], x2 = sh[5];
            const float x = tm_acos(NdotL);
            if (x > x2) visable = 0.0f;
            else if (x > x1) visable *= 1.0f - (x - x1) / (x2 - x1);
        } else if (st == SHAPE_LASER) {
            choice_pdf = 1.0f / (float)s.light_count;
            const float proj = dot(light_dir, light_normal) * light_dist;
            const float r = tm_sqrt(light_dist * light_dist - proj * proj);
            if (r > sh[4]) visable = 0.0f;
        }
    }
    return visable;
}
// ---- UtilsFunc.py:321-345 ----
TD void map_to_disk(float u1, float u2, float &r, float &phi)
{
    phi = 0.0f; r = 0.0f;
    const float a = 2.0f * u1 - 1.0f, b = 2.0f * u2 - 1.0f;
    if (a > -b) {
        if (a > b) { r = a; phi = (PI_UF / 4.0f) * (b / a); }
        else { r = b; phi = (PI_UF / 4.0f) * (2.0f - a / b); }
    } else {
        if (a < b) { r = -a; phi = (PI_UF / 4.0f) * (4.0f + b / a); }
        else { r = -b; phi = (b == 0.0f) ? 0.0f : (PI_UF / 4.0f) * (6.0f - a / b); }
    }
}
// ---- Scene.py:353-377 ------------------------------------------------------------------------------------
TD float get_prim_angle(const SceneView &s, int index, v3 v)
{
    float ret = 0.0f;
    const int *pr = s.primitive + (size_t)index * PRI_VEC;
    if (pr[0] == PRIMITIVE_TRI) {
        v3 v1 = vtx_pos(s, pr[1]), v2 = vtx_pos(s, pr[1] + 1), v3_ = vtx_pos(s, pr[1] + 2);
        if (norm(v1 - v) < 0.00001f) ret = dot(normalized(v2 - v1), normalized(v3_ - v1));
        else if (norm(v2 - v) < 0.00001f) ret = dot(normalized(v1 - v2), normalized(v3_ - v2));
        else ret = dot(normalized(v1 - v3_), normalized(v2 - v3_));
    }
    return tm_acos(ret);
}

// ---- texture/Texture.py:41-69 -------------------------------------------------------------------------------
TD v3 tex_sample(const SceneView &s, float fx, float fy)
{
    int x = (int)fx, y = (int)fy;
    x = x < 0 ? 0 : (x > s.env_w - 1 ? s.env_w - 1 : x);
    y = y < 0 ? 0 : (y > s.env_h - 1 ? s.env_h - 1 : y);
    int RGBA = s.env[(size_t)x * s.env_h + y];
    float R = (float)((RGBA & 0x00FF0000) >> 16) / 255.0f;
    float G = (float)((RGBA & 0x0000FF00) >> 8) / 255.0f;
    float B = (float)(RGBA & 0x000000FF) / 255.0f;
    return V(R, G, B);
}
TD v3 texture2d(const SceneView &s, float u, float v)
{
    float x = clampf(u * (float)s.env_w, 0.0f, (float)s.env_w - 1.0f);
    float y = clampf(v * (float)s.env_h, 0.0f, (float)s.env_h - 1.0f);
    float lx = tm_floor(x), ly = tm_floor(y);
    float wbt = y - tm_floor(y), wlr = x - tm_floor(x);
    v3 lt = tex_sample(s, lx, ly), rt = tex_sample(s, lx + 1.0f, ly);
    v3 lb = tex_sample(s, lx, ly + 1.0f), rb = tex_sample(s, lx + 1.0f, ly + 1.0f);
    return mix3(mix3(lt, rt, wlr), mix3(lb, rb, wlr), wbt);
}

// ---- Camera.py:122-142 ----------------------------------------------------------------------------------------
TD v3 camera_ray_direction(const CameraView &c, int i, int j, float jx, float jy)
{
    float x = ((float)i + jx - c.cx) / c.fx;
    float y = ((float)j + jy - c.cy) / c.fy;
    float z = -1.0f, w = 0.0f;
    const float *M = c.view_inv;
    float wx = ((M[0] * x + M[1] * y) + M[2] * z) + M[3] * w;
    float wy = ((M[4] * x + M[5] * y) + M[6] * z) + M[7] * w;
    float wz = ((M[8] * x + M[9] * y) + M[10] * z) + M[11] * w;
    return normalized(V(wx, wy, wz));
}

}  // namespace tirt
