// tirt_bdpt.hip -- BDPT_RGB (BASELINE config 5): bidirectional path tracing
// with all (eye, light) sub-path connections, MIS and light-tracing splats, as a WAVEFRONT.
//
// The reference (integrator/BDPT_RGB.py:595-642) is one per-pixel kernel: eye_path, light_path, then every (e, l)
// connection, each with its traversal inlined.  Here every ray goes through the traversal kernel of PT_RGB
// (tirt_render.hip, k_trace) in large batches, and the arithmetic around the rays runs in between:
//
//   k_bd_init          lens vertex + camera ray, light vertex + first light ray       BDPT_RGB.py:104-125, 201-228
//   6 x { trace the live rays ; k_bd_step(d) }   vertex d of the eye and of the light sub-path; the rays of a depth are a
//                      dense list, survivors append to the next depth's list               BDPT_RGB.py:126-198, 229-294
//   k_bd_delta         the one field that survives from frame to frame (see below)
//   k_bd_connect       per item: geometry of every (e, l) connection, connection rays staged   BDPT_RGB.py:481-592
//   k_bd_compact       the dense queue: per place, where the ray is staged and whose it is (item, pair slot) -- the rays are not copied
//   trace queries      "is the expected primitive the closest hit?"  (k_trace<KIND_QUERY>, bounded)
//   k_bd_emitted       the l == 0 pairs (eye sub-path ended on an emitter): no ray
//   k_bd_resolve       per QUEUED CONNECTION: contribution + MIS weight (BDPT_RGB.py:300-479), splats with float atomics
//   k_bdpt_film        running mean, frames in order                                   BDPT_RGB.py:639-642
//
// An item is one (frame, pixel) pair with its own vertex arrays (`bpixel`, 1.0 KB); a batch holds up to
// `bdpt_batch_items` of them (frames x owned pixels), so many frames are in flight at once.  The reference keeps ONE
// set of vertex arrays per pixel and clears only beta/type/fpdf/rpdf between frames; walking through every read of a
// field that the current frame has not written shows a single one that matters: the `delta` flag of an eye vertex
// that ended on a light (eye_path breaks before it stores delta there, BDPT_RGB.py:152-158) is the delta some earlier
// frame left in that slot, and connect_path's l == 1 branch reads it.  k_bd_delta replays exactly that: per pixel, in
// frame order, over a small persistent per-pixel memory of the seven delta fields (`bdpt_px`) -- every surface vertex's store
// counts, including the one of a vertex whose sampling ended the path and which the returned depth therefore leaves out.  Everything else
// (sample/temp vertices of mis_weight, geometry of slots beyond the current depth) is written before it is read.
// Other reference behaviours kept on purpose (see oracle/oracle.c for the same list): material INDEX compared with
// MAT_DISNEY in mis_weight; restores to index -1 skipped.  Light-tracing contributions are added to other pixels with
// float atomics, so a frame is reproducible only up to the order of those additions (the parity test uses the north
// star's 1e-3 relative-L2 tolerance; measured 1e-7).
#include "tirt_internal.h"
#include "tirt_spectral.h"

#if (defined(BDX_NO_ATOMIC) || defined(BDX_NO_MIS) || defined(BDX_NO_PRESUM) || defined(BD_RESOLVE_WAVES)) && !defined(TIRT_EXPERIMENTS)
#error "the BDX_* / BD_RESOLVE_* switches are ablations of k_bd_resolve: build them with -DTIRT_EXPERIMENTS (make experiments EXTRA=...)"
#endif

namespace tirt {

constexpr int BD_MAX_DEPTH = 5;                    // BDPT_RGB.py:23
constexpr int BD_EYE_MAX = BD_MAX_DEPTH + 2, BD_LIGHT_MAX = BD_MAX_DEPTH + 1;
[[maybe_unused]] constexpr int VERTEX_NONE = 0, VERTEX_LIGHT = 1, VERTEX_LENS = 2, VERTEX_SURFACE = 3;
constexpr uint32_t BD_DIM_EYE = 16, BD_DIM_LSTART = 80, BD_DIM_LIGHT = 96, BD_DIM_CONNECT = 176;
constexpr float EPS_UF = 0.00001f;                 // UtilsFunc.py:36

// BDPT_Vertex.py:10-21 in 80 bytes, 16-byte aligned, every vector with a scalar that is read with it: a vertex moves as five dwordx4
// accesses instead of 21 scalar ones, and a connection's geometry pass reads (pos, prim), (snormal, mat) and the type/delta word.
struct alignas(16) bvert { v3 pos; int prim; v3 snormal; int mat; v3 normal; float fpdf; v3 beta; float rpdf; v3 wo; short type, delta; };
static_assert(sizeof(bvert) == 80 && offsetof(bvert, rpdf) == 60 && offsetof(bvert, delta) == 78, "bvert is five quads: ... (beta, rpdf) (wo, type | delta)");
// The vertex arrays of a batch's items in HBM: quad k (16 bytes) of vertex slot s of item `it` lies at q[(s * 5 + k) * stride + it] (eye vertices: slots 0..6, light
// vertices: 7..12).  Lanes that hold neighbouring items read neighbouring quads: a wave's vertex load is five requests of 1 KB.  (Rounds 2-3 kept an item's 13
// vertices together, 1 040 bytes apart from the next item's: every lane of every vertex access in a cache line of its own -- what made k_bd_step, k_bd_connect
// and k_bd_resolve texture-address bound.)
constexpr int BD_QUADS = 5, BD_SLOTS = BD_EYE_MAX + BD_LIGHT_MAX;
struct BdItems { float4 *q; size_t stride; };
constexpr size_t BD_ITEM_BYTES = sizeof(bvert) * BD_SLOTS;
struct varr {                              // one of an item's two vertex arrays: reads of a whole vertex or of one field of it (the quad that holds it), whole-vertex and single-field stores
    float4 *q; size_t stride;
    TD float4 *at(int s, int k) const { return q + (size_t)(s * BD_QUADS + k) * stride; }
    TD bvert operator[](int s) const
    {
        bvert v;
#pragma unroll
        for (int k = 0; k < BD_QUADS; k++) { const float4 x = *at(s, k); __builtin_memcpy((char *)&v + 16 * k, &x, 16); }
        return v;
    }
    TD void put(int s, const bvert &v) const
    {
#pragma unroll
        for (int k = 0; k < BD_QUADS; k++) { float4 x; __builtin_memcpy(&x, (const char *)&v + 16 * k, 16); *at(s, k) = x; }
    }
    TD float rpdf(int s) const { return ((const float *)at(s, 3))[3]; }
    TD float fpdf(int s) const { return ((const float *)at(s, 2))[3]; }
    TD int delta(int s) const { return (int)((const short *)at(s, 4))[7]; }
    TD void set_rpdf(int s, float f) const { ((float *)at(s, 3))[3] = f; }                    // bvert::rpdf: last word of quad 3
    TD void set_delta(int s, int d) const { ((short *)at(s, 4))[7] = (short)d; }              // bvert::delta: last half-word of quad 4
};
struct bpixel { varr eye, light; };        // (a view of one item's vertex arrays; the reference's sample / temp vertices (BDPT_RGB.py:60-64) are locals of the connection code)
TD bpixel bd_item(const BdItems &I, size_t it) { float4 *b = I.q + it; bpixel B; B.eye.q = b; B.eye.stride = I.stride; B.light.q = b + (size_t)BD_EYE_MAX * BD_QUADS * I.stride; B.light.stride = I.stride; return B; }

struct BdView { float view[12]; int W, H; };
struct SimpleHit { float t, u, v; int prim; };

// A connection needs at most one ray.  Pass 0 of k_bd_connect records it (and sees a miss, so nothing downstream
// runs and nothing is written); pass 1 is handed the traced result.
struct Tracer { int phase; bool want; v3 o, d; int expect; float bound; SimpleHit res; };
TD SimpleHit bd_trace(Tracer &T, v3 o, v3 d, int expect, float bound)
{
    if (T.phase == 0) { T.want = true; T.o = o; T.d = d; T.expect = expect; T.bound = bound; SimpleHit m; m.t = INF_VALUE; m.u = 0.0f; m.v = 0.0f; m.prim = -1; return m; }
    return T.res;
}

TD float cosine_hemisphere_pdf(float c) { return maxf(0.01f, c / PI_UF); }       // UtilsFunc.py:348-350
TD float remap0(float f) { return f == 0.0f ? 1.0f : f; }                       // BDPT_RGB.py:89-93

// brdf/Disney.py:43-63
TD float disney_pdf(const float *m, v3 N, v3 Vv, v3 L)
{
    float pdf = 0.0f;
    float NDotL = dot(N, L), NDotV = dot(N, Vv);
    if ((NDotL > 0.0f) & (NDotV > 0.0f)) {
        const float inv_pi = (float)(1.0 / 3.1415956);
        v3 H = normalized(L + Vv);
        float NDotH = dot(H, N), LDotH = dot(H, L);
        float metal = m[5], rough = m[6];
        float specularAlpha = maxf(0.001f, rough);
        float Ds = gtr2(NDotH, specularAlpha);
        float diffuseRatio = 0.5f * (1.0f - metal);
        float specularRatio = 1.0f - diffuseRatio;
        float pdfGTR2 = Ds * NDotH;
        float pdfSpec = pdfGTR2 / (4.0f * absf(LDotH));
        pdf = diffuseRatio * inv_pi + specularRatio * pdfSpec;
    }
    return pdf;
}
TD const float *mat_row(const SceneView &s, int mat_id) { return s.material + (size_t)mat_id * MAT_VEC; }
// UF.srgb_to_lrgb(material colour): the per-material table the upload fills with that very function (tirt_api.hip, k_material_lrgb) -- inline it is three f64 pow
// polynomials per call, and connect_path alone has four call sites (round 5)
TD v3 mat_lrgb(const SceneView &s, int mat_id) { const float *t = s.mat_lrgb + (size_t)mat_id * 3; return V(t[0], t[1], t[2]); }

// Camera.py:144-158
TD v3 get_image_point(const CameraView &cam, const BdView &bv, v3 p, int &u, int &v)
{
    const float *M = bv.view;
    float px = ((M[0] * p.x + M[1] * p.y) + M[2] * p.z) + M[3] * 1.0f;
    float py = ((M[4] * p.x + M[5] * p.y) + M[6] * p.z) + M[7] * 1.0f;
    float pz = ((M[8] * p.x + M[9] * p.y) + M[10] * p.z) + M[11] * 1.0f;
    float fu = -px / pz * cam.fx + cam.cx, fv = -py / pz * cam.fy + cam.cy;
    u = (fu > -2.0e9f && fu < 2.0e9f) ? (int)fu : -1;
    v = (fv > -2.0e9f && fv < 2.0e9f) ? (int)fv : -1;
    v3 wi = V(0.0f, 0.0f, 0.0f);
    if ((u < 0) | (u >= bv.W) | (v < 0) | (v >= bv.H) | (pz > 0.0f)) { u = -1; v = -1; }
    else wi = p - V(cam.eye[0], cam.eye[1], cam.eye[2]);
    return normalized(wi);
}

struct bsample { v3 next_dir; float f_or_b, brdf, pdfFwd; };
TD bsample bd_sample(const SceneView &s, v3 dir, v3 normal, v3 fnormal, int mat_id, int mat_type, uint32_t seed, uint32_t pixel,
                     uint32_t frame, uint32_t dim0, int &delta, bool spectral = false, float Lambda = 0.0f)
{
    bsample r; r.next_dir = dir; r.f_or_b = 1.0f; r.brdf = 0.0f; r.pdfFwd = 0.0f;
    const float *m = mat_row(s, mat_id);
    if (mat_type == MAT_GLASS) {
        if (spectral) r.next_dir = glass_sample_lambda(dir, normal, Lambda, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_GLASS), r.f_or_b);      // SPEC: BDPT_SPEC.py:241, 335
        else r.next_dir = glass_sample(m, dir, normal, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_GLASS), r.f_or_b);
        r.brdf = 1.0f; r.pdfFwd = 1.0f;
        delta = 1;
    } else {
        r.next_dir = disney_sample(m, dir, fnormal, tm_rand(seed, pixel, frame, dim0 + TM_SLOT_LOBE),
                                   tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R1), tm_rand(seed, pixel, frame, dim0 + TM_SLOT_R2));
        r.f_or_b = 1.0f;
        r.brdf = disney_evaluate_pdf(m, fnormal, -dir, r.next_dir, r.pdfFwd);
        delta = 0;
    }
    return r;
}

// `spec` != nullptr: BDPT_SPEC (integrator/BDPT_SPEC.py) -- the same tracer carrying ONE wavelength per pixel sample.  The colours of BDPT_RGB
// become powers at that wavelength, replicated into the three components of the v3 the RGB code carries (so every product below is the
// reference's scalar product); where BDPT_SPEC.py differs from BDPT_RGB.py in more than that, the code says "SPEC".  The SpecView lives in
// device memory (tirt_spectral_upload) and is read through the pointer where it is needed.
struct BdCtx { SceneView sc; CameraView cam; BdView bv; uint32_t seed; int bounded; const SpecView *spec; };
constexpr uint32_t BD_DIM_LAMBDA = 2, BD_DIM_CONNECT_SPEC = 256;      // the sample's wavelength (dims 0, 1: camera jitter); + 8 e: sample_light() of the l == 1 connection
// BDPT_SPEC.py:668: lambda_min + lambda_range * size * rand -- up to one step beyond lambda_max, where the sensor reads zero
TD float bd_lambda(const BdCtx &c, uint32_t pixel, uint32_t frame)
{ const SpecView &sp = *c.spec; return sp.s_min + (sp.s_range * (float)sp.n_sensor) * tm_rand(c.seed, pixel, frame, BD_DIM_LAMBDA); }
TD float bd_light_power(const SpecView &sp, v3 emission, float Lambda)             // BDPT_SPEC.py:146-155
{
    float ret = 0.0f;
    const float scale = norm(emission);
    if (scale > 0.0f) {
        const v3 coff = r2s_fetch(sp, emission / scale);
        ret = spd_sample(sp.spd[0], Lambda) * r2s_eval(coff, Lambda) * scale;
    }
    return ret;
}
TD float bd_reflect_power(const SceneView &s, const SpecView &sp, int mat_id, float Lambda)      // BDPT_SPEC.py:134-144
{
    const float *m = s.material + (size_t)mat_id * MAT_VEC;
    const v3 mat_color = V(m[2], m[3], m[4]);
    if ((int)m[0] == MAT_LIGHT) return bd_light_power(sp, mat_color, Lambda);
    return r2s_eval(r2s_fetch(sp, mat_lrgb(s, mat_id)), Lambda);
}
// what BDPT_RGB multiplies a path's throughput with at a surface of material mat_id: its linear colour / its power at the wavelength
template <bool SPEC>
TD v3 bd_reflect(const BdCtx &c, int mat_id, float Lambda)
{
    if (SPEC) { const float p = bd_reflect_power(c.sc, *c.spec, mat_id, Lambda); return V(p, p, p); }
    return mat_lrgb(c.sc, mat_id);
}


// BDPT_RGB.py:300-479
TD float bd_mis_weight(const BdCtx &c, const bpixel &P, const bvert &sample, int e, int l)
{
    const SceneView &s = c.sc;
    const varr light = P.light, eye = P.eye;
    float weight_sum = 0.0f;
    if (l + e != 2) {
        // The reference saves the four vertices next to the connection (ltemp, etemp, lminustemp, eminustemp), overwrites
        // them in place, reads the arrays, and restores them (BDPT_RGB.py:303-321, 472-477).  Same values here without
        // the round trip through memory: private copies E1 = eye[e-1], E2 = eye[e-2], L1 = light[l-1], L2 = light[l-2]
        // take the modifications, every other vertex is read where it lies.
        bvert E1 = bvert(), E2 = bvert(), L1 = bvert(), L2 = bvert();
        if (l > 0) L1 = light[l - 1];
        if (e > 0) E1 = eye[e - 1];
        if (l > 1) L2 = light[l - 2];
        if (e > 1) E2 = eye[e - 2];
        if (l == 1) L1 = sample;
        else if (e == 1) E1 = sample;
        if (l > 0) L1.delta = 0;
        if (e > 0) E1.delta = 0;

        if (e > 0) {
            if (l == 0) {
                const float pdfPos = 1.0f / get_prim_area(s, E1.prim);
                const float pdfChoice = 1.0f / (float)s.light_count;
                E1.rpdf = pdfPos * pdfChoice;
            } else if (l == 1) {
                if (E1.type == VERTEX_SURFACE) {
                    v3 to = E1.pos - L1.pos;
                    const float dist = norm(to);
                    to = to / dist;
                    const float pdfDir = cosine_hemisphere_pdf(absf(dot(to, L1.normal)));
                    const float LdotN = absf(dot(to, L1.normal));
                    E1.rpdf = pdfDir * LdotN / (dist * dist);
                } else E1.rpdf = 1.0f;
            } else {
                v3 wi = L2.pos - L1.pos;
                v3 wo = E1.pos - L1.pos;
                const float dist = norm(wo);
                wi = normalized(wi); wo = normalized(wo);
                float pdf = 1.0f;
                const int mat_id = L1.mat;
                if (mat_id == MAT_DISNEY) pdf = disney_pdf(mat_row(s, mat_id), L1.snormal, wi, wo);
                E1.rpdf = pdf * absf(dot(L1.normal, wo)) / (dist * dist);
            }
        }
        if (l > 0) {
            if (e > 1) {
                if (E1.type == VERTEX_SURFACE) {
                    v3 wi = E2.pos - E1.pos;
                    v3 wo = L1.pos - E1.pos;
                    const float dist = norm(wo);
                    wi = normalized(wi); wo = normalized(wo);
                    float pdf = 1.0f;
                    const int mat_id = E1.mat;
                    if (mat_id == MAT_DISNEY) pdf = disney_pdf(mat_row(s, mat_id), E1.snormal, wi, wo);
                    L1.rpdf = pdf * absf(dot(E1.normal, wo)) / (dist * dist);
                } else L1.rpdf = 1.0f;
            } else {
                v3 to = E1.pos - L1.pos;
                const float dist = norm(to);
                to = to / dist;
                const v3 axis = V(c.bv.view[8], c.bv.view[9], c.bv.view[10]);      // Camera.py:126-127
                const float LdotN = dot(to, axis);
                L1.rpdf = LdotN / (dist * dist);
            }
        }
        if (e > 1) {
            if (l == 0) {
                v3 to = E2.pos - E1.pos;
                const float dist = norm(to);
                to = to / dist;
                const float pdfDir = cosine_hemisphere_pdf(absf(dot(to, E1.normal)));
                const float LdotN = dot(to, E1.normal);
                E2.rpdf = absf(pdfDir * LdotN) / (dist * dist);
            } else {
                if (E1.type == VERTEX_SURFACE) {
                    v3 wi = L1.pos - E1.pos;
                    v3 wo = E2.pos - E1.pos;
                    const float dist = norm(wo);
                    wi = normalized(wi); wo = normalized(wo);
                    const int mat_id = E1.mat;
                    const float pdf = disney_pdf(mat_row(s, mat_id), E1.snormal, wi, wo);
                    E2.rpdf = pdf / (dist * dist);
                    if (E2.type == VERTEX_SURFACE) E2.rpdf *= absf(dot(E1.normal, wo));
                } else E2.rpdf = 1.0f;
            }
        }
        if (l > 1) {
            if (E1.type != VERTEX_LIGHT) {
                v3 wi = E1.pos - L1.pos;
                v3 wo = L2.pos - L1.pos;
                const float dist = norm(wo);
                wi = normalized(wi); wo = normalized(wo);
                float pdf = 1.0f;
                const int mat_id = L1.mat;
                if (mat_id == MAT_DISNEY) pdf = disney_pdf(mat_row(s, mat_id), L1.normal, wi, wo);
                L2.rpdf = pdf / (dist * dist);
                if (L2.type == VERTEX_SURFACE) L2.rpdf *= absf(dot(L1.normal, wo));
            } else L2.rpdf = 1.0f;
        }

        float weight = 1.0f;
        for (int k = e - 1; k > 0; k--) {
            const float rp = (k == e - 1) ? E1.rpdf : ((k == e - 2) ? E2.rpdf : eye.rpdf(k));
            const float fp = (k == e - 1) ? E1.fpdf : ((k == e - 2) ? E2.fpdf : eye.fpdf(k));
            const int dk = (k == e - 1) ? E1.delta : ((k == e - 2) ? E2.delta : eye.delta(k));
            const int dk1 = (k - 1 == e - 2) ? E2.delta : eye.delta(k - 1);
            weight *= remap0(rp) / remap0(fp);
            if ((dk == 0) & (dk1 == 0)) weight_sum += weight;
        }
        weight = 1.0f;
        for (int k = l - 1; k >= 0; k--) {
            const float rp = (k == l - 1) ? L1.rpdf : ((k == l - 2) ? L2.rpdf : light.rpdf(k));
            const float fp = (k == l - 1) ? L1.fpdf : ((k == l - 2) ? L2.fpdf : light.fpdf(k));
            const int dk = (k == l - 1) ? L1.delta : ((k == l - 2) ? L2.delta : light.delta(k));
            weight *= remap0(rp) / remap0(fp);
            if (k == 0) { if (dk == 0) weight_sum += weight; }
            else {
                const int dk1 = (k - 1 == l - 2) ? L2.delta : light.delta(k - 1);
                if ((dk == 0) & (dk1 == 0)) weight_sum += weight;
            }
        }
    }
    return 1.0f / (1.0f + weight_sum);
}

// Scene.sample_light (Scene.py:430-474) with its random numbers at dimensions dim_base .. dim_base + 6
TD void bd_sample_light(const BdCtx &c, uint32_t pixel, uint32_t frame, uint32_t dim_base, v3 &lpos_o, v3 &lnor_o, v3 &ldir_o, v3 &emission_o, int &lp_o,
                        float &choice_pdf_o, float &dir_pdf_o)
{
    const SceneView &s = c.sc;
    int lidx = (int)(tm_rand(c.seed, pixel, frame, dim_base + 0) * (float)s.light_count);
    if (lidx >= s.light_count) lidx = s.light_count - 1;
    const int lp = s.light[lidx];
    v3 lpos, lnor;
    get_prim_random_point_normal(s, lp, tm_rand(c.seed, pixel, frame, dim_base + 1), tm_rand(c.seed, pixel, frame, dim_base + 2), lpos, lnor);
    const float *lm = mat_row(s, s.primitive[(size_t)lp * PRI_VEC + 2]);
    v3 emission = V(lm[2], lm[3], lm[4]);
    float choice_pdf = 1.0f / ((float)s.light_count * get_prim_area(s, lp));
    lnor = normalized(lnor);
    const v3 ld = cosine_sample_hemisphere(tm_rand(c.seed, pixel, frame, dim_base + 3), tm_rand(c.seed, pixel, frame, dim_base + 4));
    float dir_pdf = cosine_hemisphere_pdf(ld.z);
    v3 ldir = inverse_transform(ld, lnor);
    const int *lpr = s.primitive + (size_t)lp * PRI_VEC;
    if (lpr[0] != PRIMITIVE_TRI) {                                  // Scene.py:449-472: the two shape emitters without a surface
        const float *sh = s.shape + (size_t)lpr[1] * SHA_VEC;
        const int st = (int)sh[0];
        if (st == SHAPE_SPOT) {
            const float scale = sh[6];
            dir_pdf = 1.0f;
            float r, phi;
            map_to_disk(tm_rand(c.seed, pixel, frame, dim_base + 5), tm_rand(c.seed, pixel, frame, dim_base + 6), r, phi);
            const float r1 = scale * tm_tan(sh[4]), r2 = scale * tm_tan(sh[5]);
            r *= r2;
            if (r > r1) emission = emission * (1.0f - (r - r1) / (r2 - r1));
            const v3 sp = V(r * tm_cos(phi), r * tm_sin(phi), tm_sqrt(maxf(0.0f, scale * scale - r * r)));
            ldir = inverse_transform(sp, lnor);
        } else if (st == SHAPE_LASER) {
            choice_pdf = 1.0f / (float)s.light_count;
            const float r = sh[4];
            const float phi = tm_rand(c.seed, pixel, frame, dim_base + 5) * PI_UF * 2.0f;
            v3 sp = V(r * tm_cos(phi), r * tm_sin(phi), 0.0f);
            sp = inverse_transform(sp, lnor);
            ldir = lnor;
            dir_pdf = 1.0f;
            lpos = lpos + sp;
        }
    }
    lpos_o = lpos; lnor_o = lnor; ldir_o = ldir; emission_o = emission; lp_o = lp; choice_pdf_o = choice_pdf; dir_pdf_o = dir_pdf;
}

// BDPT_RGB.py:481-592
template <bool SPEC>
TD v3 bd_connect_path(const BdCtx &c, const bpixel &P, const bvert &EV, bvert &sample, int i, int j, int e, int l, uint32_t frame, int &nu, int &nv, Tracer &T)
{
    const SceneView &s = c.sc;
    // private copies of the two vertices being connected (one wide load each; nothing is written to the arrays): EV = eye[e - 1] is the
    // caller's (the per-item loop keeps it across the l loop), LV is loaded here
    const bvert LV = (l > 0) ? P.light[l - 1] : bvert();
    const uint32_t pixel = (uint32_t)(i * c.bv.H + j);
    constexpr bool spectral = SPEC;
    const float Lambda = spectral ? bd_lambda(c, pixel, frame) : 0.0f;
    v3 radiance = V(0.0f, 0.0f, 0.0f);
    nu = i; nv = j;
    if (l == 0) {
        if (EV.type == VERTEX_LIGHT) radiance = EV.beta;
    } else if (e == 1) {
        const int prim = LV.prim;
        const v3 surface = LV.pos;
        const v3 wi = get_image_point(c.cam, c.bv, surface, nu, nv);
        const v3 origin = V(c.cam.eye[0], c.cam.eye[1], c.cam.eye[2]);
        const int mat_id = LV.mat;
        const v3 snormal = LV.snormal;
        const float NdotL = dot(wi, snormal);
        if ((nu >= 0) & (LV.delta != 1) & (NdotL < 0.0f) & (LV.type == VERTEX_SURFACE)) {
            const SimpleHit sh = bd_trace(T, origin, wi, prim, c.bounded ? norm(surface - origin) : -1.0f);
            if (sh.prim == prim) {
                float pdf;
                const float brdf = disney_evaluate_pdf(mat_row(s, mat_id), snormal, -LV.wo, -wi, pdf);
                if (pdf > 0.0f) {
                    const float G = absf(NdotL) / (sh.t * sh.t);
                    radiance = ((((LV.beta * G) * bd_reflect<SPEC>(c, mat_id, Lambda)) * brdf) / pdf);
                    sample.pos = origin; sample.wo = wi; sample.type = VERTEX_LENS; sample.fpdf = 1.0f;
                }
            }
        }
    } else if (l == 1) {
        const v3 surface = offset_ray(EV.pos, EV.snormal);
        const int mat_id = EV.mat;
        if (spectral && EV.delta != 1) {
            // SPEC: BDPT_SPEC.py:605-630 -- sample_light() (its direction sample is drawn and dropped) instead of sample_li(surface); an emitter without
            // a surface (spot, laser) can never be the shadow ray's hit, so it contributes through the light path only
            v3 light_pos, light_normal, light_dir_unused, light_emission; int light_prim; float light_choice_pdf, light_dir_pdf;
            bd_sample_light(c, pixel, frame, BD_DIM_CONNECT_SPEC + 8u * (uint32_t)e, light_pos, light_normal, light_dir_unused, light_emission, light_prim,
                            light_choice_pdf, light_dir_pdf);
            const v3 wi = normalized(surface - light_pos);
            const float NdotLl = dot(wi, light_normal), NdotLe = dot(wi, EV.snormal);
            const SimpleHit sh = bd_trace(T, surface, -wi, light_prim, c.bounded ? norm(surface - light_pos) : -1.0f);
            if ((sh.prim == light_prim) & (sh.t > EPS_UF)) {
                const float light_pdf = light_choice_pdf;
                float pdf;
                const float brdf = disney_evaluate_pdf(mat_row(s, mat_id), EV.snormal, -EV.wo, -wi, pdf);
                if (pdf > 0.0f) {
                    const float G = absf(NdotLe * NdotLl) / (sh.t * sh.t);
                    v3 cc = ((EV.beta * G) * brdf) / pdf;
                    cc = cc * bd_reflect<SPEC>(c, mat_id, Lambda);
                    cc = cc * bd_light_power(*c.spec, light_emission, Lambda);
                    radiance = cc / light_pdf;
                }
                sample.pos = light_pos; sample.wo = wi; sample.type = VERTEX_LIGHT; sample.fpdf = light_pdf;
                sample.prim = light_prim; sample.normal = light_normal; sample.snormal = light_normal;
            }
        } else if (EV.delta != 1) {
            const uint32_t d0 = BD_DIM_CONNECT + 4u * (uint32_t)e;              // Scene.py:477-518 sample_li(surface)
            int lidx = (int)(tm_rand(c.seed, pixel, frame, d0) * (float)s.light_count);
            if (lidx >= s.light_count) lidx = s.light_count - 1;
            const int light_prim = s.light[lidx];
            v3 light_pos, light_normal;
            get_prim_random_point_normal(s, light_prim, tm_rand(c.seed, pixel, frame, d0 + 1), tm_rand(c.seed, pixel, frame, d0 + 2), light_pos, light_normal);
            const float *lm = mat_row(s, s.primitive[(size_t)light_prim * PRI_VEC + 2]);
            v3 light_emission = V(lm[2], lm[3], lm[4]);
            float light_choice_pdf = 1.0f / ((float)s.light_count * get_prim_area(s, light_prim));
            light_normal = normalized(light_normal);
            v3 wi = surface - light_pos;
            const float light_dist = norm(wi);
            wi = wi / light_dist;
            light_emission = light_emission * light_shape_visible(s, light_prim, wi, light_normal, light_dist, light_choice_pdf);   // spot / laser (Scene.py:491-516)
            const float NdotLl = dot(wi, light_normal);
            const float NdotLe = dot(wi, EV.snormal);
            const SimpleHit sh = bd_trace(T, surface, -wi, light_prim, c.bounded ? light_dist : -1.0f);
            if ((sh.prim == light_prim) & (sh.t > EPS_UF)) {
                const float light_pdf = light_choice_pdf;
                float pdf;
                const float brdf = disney_evaluate_pdf(mat_row(s, mat_id), EV.snormal, -EV.wo, -wi, pdf);
                if (pdf > 0.0f) {
                    const float G = absf(NdotLe * NdotLl) / (sh.t * sh.t);
                    v3 cc = ((EV.beta * G) * brdf) / pdf;
                    cc = cc * mat_lrgb(s, mat_id);
                    cc = cc * light_emission;
                    radiance = cc / light_pdf;
                }
                sample.pos = light_pos; sample.wo = wi; sample.type = VERTEX_LIGHT; sample.fpdf = light_pdf;
                sample.prim = light_prim; sample.normal = light_normal; sample.snormal = light_normal;
            }
        }
    } else {
        if ((LV.delta != 1) & (EV.delta != 1) & (EV.type == VERTEX_SURFACE) & (LV.type == VERTEX_SURFACE)) {
            const int primE = EV.prim, mat_idE = EV.mat, mat_idL = LV.mat;
            const v3 surfaceE = EV.pos, surfaceL = LV.pos;
            v3 dir = surfaceE - surfaceL;
            const float dist = norm(dir);
            dir = dir / dist;
            const float NdotLl = dot(dir, LV.snormal), NdotLe = dot(dir, EV.snormal);
            const SimpleHit sh = bd_trace(T, surfaceL, dir, primE, c.bounded ? dist : -1.0f);
            if ((sh.prim == primE) & (sh.t > EPS_UF)) {
                float lpdf, epdf;
                const float brdfL = disney_evaluate_pdf(mat_row(s, mat_idL), LV.snormal, -LV.wo, dir, lpdf);
                const float brdfE = disney_evaluate_pdf(mat_row(s, mat_idE), EV.snormal, -EV.wo, -dir, epdf);
                if ((brdfL > 0.0f) & (brdfE > 0.0f)) {
                    const float G = absf(NdotLe * NdotLl) / (dist * dist);
                    v3 cc = (EV.beta * G) * LV.beta;
                    cc = (cc * brdfL) / lpdf;
                    cc = (cc * brdfE) / epdf;
                    cc = cc * bd_reflect<SPEC>(c, mat_idE, Lambda);
                    radiance = cc * bd_reflect<SPEC>(c, mat_idL, Lambda);
                }
            }
        }
    }
    float misweight = 1.0f;
#ifdef BDX_NO_MIS
    if (radiance.x == -12345.0f) misweight =
#else
    if ((radiance.x > 0.0f) & (radiance.y > 0.0f) & (radiance.z > 0.0f)) misweight =
#endif
        bd_mis_weight(c, P, sample, e, l);
    return radiance * misweight;
}

// ---- wavefront state ------------------------------------------------------------------------------------------
// e_tail: slot eye[eye_depth] holds a SURFACE vertex of THIS item (its sampling ended the path, so the depth leaves it out) -- k_bd_delta
// In HBM as four arrays over the items (a sub-path's thread reads and writes ONE quad and one word, next to its neighbours'; as a 48-byte record per item
// every field access of every lane was a request of its own): (e_beta, e_pdfFwd), (l_beta, l_pdfFwd), eye_depth | e_tail << 16, light_depth.
struct BdSteps { float4 *eb, *lb; int *ed, *ld; };
constexpr size_t BD_STEP_BYTES = 2 * sizeof(float4) + 2 * sizeof(int);
inline BdSteps bd_steps(void *base, size_t n) { BdSteps S; S.eb = (float4 *)base; S.lb = S.eb + n; S.ed = (int *)(S.lb + n); S.ld = S.ed + n; return S; }
struct BdRays { float4 *r; };             // a ray list: 32-byte records (o.xyz, d.x), (d.y, d.z, bits expect, bound) -- TraceArgs::ray4: two memory instructions per ray and side
constexpr int BD_PAIRS = BD_EYE_MAX * (BD_LIGHT_MAX + 1);          // (e - 1) * 7 + l
// The pairs that can carry a connection ray: l >= 1, (e, l) != (1, 1), 0 <= e + l - 2 <= BD_MAX_DEPTH -- for e = 1..6: 5, 5, 4, 3, 2, 1
// (the seven l == 0 pairs never do).  The staging area, the dense queue and its hit records are sized by it.
constexpr int BD_RAY_PAIRS = 20;
constexpr int bd_count_ray_pairs() { int n = 0; for (int e = 1; e <= BD_EYE_MAX; e++) for (int l = 1; l <= BD_LIGHT_MAX; l++) if (!(l == 1 && e == 1) && l + e - 2 >= 0 && l + e - 2 <= BD_MAX_DEPTH) n++; return n; }
static_assert(bd_count_ray_pairs() == BD_RAY_PAIRS, "BD_RAY_PAIRS");
TD void put_ray(const BdRays &r, size_t k, v3 o, v3 d) { r.r[2 * k] = make_float4(o.x, o.y, o.z, d.x); r.r[2 * k + 1] = make_float4(d.y, d.z, 0.0f, 0.0f); }

// BDPT_RGB.py:104-125 (lens vertex, camera ray) and :201-228 with Scene.sample_light (Scene.py:430-474)
template <bool SPEC>
__global__ void k_bd_init(BdCtx c, BdItems items, BdSteps steps, BdRays rays, int *owner, int *alive_cnt, TileMap tm, int P, int N, uint32_t frame_begin, unsigned long long *paths)
{
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= N) return;
    int f, k; slot_to_frame_pixel(tm, P, it, f, k);
    const int p = local_to_pixel(tm, k);
    const int i = p / c.bv.H, j = p - i * c.bv.H;
    const uint32_t pixel = (uint32_t)p, frame = frame_begin + (uint32_t)f;
    const bpixel B = bd_item(items, (size_t)it);
    // eye
    {
        const varr eye = B.eye;
        const v3 origin = V(c.cam.eye[0], c.cam.eye[1], c.cam.eye[2]);
        float jx = 0.0f, jy = 0.0f;
        if (frame != 0) { jx = tm_rand(c.seed, pixel, frame, TM_DIM_JX) - 0.5f; jy = tm_rand(c.seed, pixel, frame, TM_DIM_JY) - 0.5f; }
        const v3 dir = camera_ray_direction(c.cam, i, j, jx, jy);
        bvert ev = bvert();                    // whole 80-byte stores: the fields the reference does not set are the zeros its field starts with
        ev.pos = origin; ev.normal = dir; ev.beta = V(1.0f, 1.0f, 1.0f); ev.fpdf = 1.0f; ev.type = VERTEX_LENS;
        eye.put(0, ev);
        steps.eb[it] = make_float4(1.0f, 1.0f, 1.0f, 1.0f); steps.ed[it] = 1;        // beta, pdfFwd; depth 1, no tail
        put_ray(rays, (size_t)it, origin, dir); owner[it] = it;
    }
    // light
    {
        const varr light = B.light;
        v3 lpos, lnor, ldir, emission; int lp; float choice_pdf, dir_pdf;
        bd_sample_light(c, pixel, frame, BD_DIM_LSTART, lpos, lnor, ldir, emission, lp, choice_pdf, dir_pdf);
        (void)lp;
        const float light_pdf = choice_pdf;
        v3 beta0 = emission / light_pdf, beta1 = (emission / light_pdf) * absf(dot(lnor, ldir));
        if (SPEC) {          // SPEC: power[0] = light power at the wavelength / pdf, and beta = power[0] without the cosine (BDPT_SPEC.py:284, 294)
            const float pw = bd_light_power(*c.spec, emission, bd_lambda(c, pixel, frame)) / light_pdf;
            beta0 = V(pw, pw, pw); beta1 = beta0;
        }
        bvert lv = bvert();
        lv.pos = lpos; lv.normal = lnor; lv.beta = beta0;
        lv.fpdf = light_pdf; lv.rpdf = 0.0f; lv.wo = ldir; lv.type = VERTEX_LIGHT;
        light.put(0, lv);
        steps.lb[it] = make_float4(beta1.x, beta1.y, beta1.z, dir_pdf); steps.ld[it] = 1;
        put_ray(rays, (size_t)N + it, lpos, ldir); owner[N + it] = N + it;
    }
    if (it == 0) { atomicAdd(paths, (unsigned long long)N); alive_cnt[1] = 2 * N; }        // depth 1: every sub-path has its first ray
}

// One iteration of the while loops of eye_path (BDPT_RGB.py:126-198; threads [0, N)) and light_path (:229-294; threads
// [N, 2N)): the hit of the ray traced for vertex `depth`, the vertex, the next ray.
// The rays of a depth are a dense list (`rays`, `owner`: which sub-path -- t < N eye of item t, else light of item t - N); the
// sub-paths that go on append their next ray to the list of the next depth (one atomic per wave), so the later depths launch
// work only for what is still alive.
constexpr int BD_STEP_BLOCK = 512;
template <bool SPEC>
__global__ __launch_bounds__(BD_STEP_BLOCK) void k_bd_step(BdCtx c, BdItems items, BdSteps steps, BdRays rays, const int *owner, BdRays rays_out, int *owner_out, int *alive_cnt,
                          const float4 *hits, TileMap tm, int P, int N, uint32_t frame_begin, int depth, unsigned long long *rays_closest)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool go_on = false;
    v3 next_o = V(0.0f, 0.0f, 0.0f), next_d = next_o;
    int t = 0;
    if (qi < alive_cnt[depth]) {
        t = owner[qi];
        const bool is_eye = t < N;
        const int it = is_eye ? t : t - N;
        {
            int f, k; slot_to_frame_pixel(tm, P, it, f, k);
            const uint32_t pixel = (uint32_t)local_to_pixel(tm, k), frame = frame_begin + (uint32_t)f;
            const SceneView &s = c.sc;
            const bpixel B = bd_item(items, (size_t)it);
            constexpr bool spectral = SPEC;
            const float Lambda = spectral ? bd_lambda(c, pixel, frame) : 0.0f;
            const float4 rq0 = rays.r[2 * (size_t)qi], rq1 = rays.r[2 * (size_t)qi + 1];
            const v3 origin = V(rq0.x, rq0.y, rq0.z), dir = V(rq0.w, rq1.x, rq1.y);
            const float4 hr = hits[qi];
            SimpleHit sh; sh.t = hr.x; sh.u = hr.y; sh.v = hr.z; sh.prim = __float_as_int(hr.w);
            const int pre_depth = depth - 1;
            int final_depth = depth;              // what the reference's function returns if the loop ends here
            next_o = origin; next_d = dir;
            if (is_eye) {
                const varr eye = B.eye;
                const float4 sq = steps.eb[it];
                float pdfFwd = sq.w, pdfRev = 0.0f;
                v3 beta = V(sq.x, sq.y, sq.z);
                bool stored_surface = false;
                if (sh.t < INF_VALUE) {
                    int mat_id;
                    const HitAttr h = hit_attributes_rec(s.shade_rec, origin, dir, sh.prim, sh.t, sh.u, sh.v, mat_id);
                    const v3 normal = h.nor, pos = h.pos;
                    const v3 fnormal = normal * signf(dot(-dir, h.gnor));
                    const float *m = mat_row(s, mat_id);
                    const v3 mat_color = V(m[2], m[3], m[4]);
                    const int mat_type = (int)m[0];
                    v3 to = pos - origin;
                    const float dist = maxf(norm(to), 0.01f);
                    const float inv_dist2 = 1.0f / (dist * dist);
                    to = to / dist;
                    bvert ev = bvert(); bvert *e = &ev;
                    e->pos = pos; e->normal = normal; e->snormal = fnormal; e->wo = dir; e->rpdf = 0.0f; e->prim = sh.prim; e->mat = mat_id;
                    e->fpdf = pdfFwd * absf(dot(to, eye[pre_depth].normal)) * inv_dist2;
                    if (mat_type == MAT_LIGHT) {
                        if (spectral) e->beta = beta * bd_reflect<SPEC>(c, mat_id, Lambda);          // SPEC: beta * reflect_power, no cosine (BDPT_SPEC.py:228)
                        else e->beta = (beta * mat_color) * absf(dot(normal, dir));
                        e->type = VERTEX_LIGHT;
                        final_depth = depth + 1;
                    } else {
                        e->beta = beta * absf(dot(dir, normal));
                        e->type = VERTEX_SURFACE; stored_surface = true;
                        const v3 reflect_color = spectral ? bd_reflect<SPEC>(c, mat_id, Lambda) : mat_lrgb(s, mat_id);
                        int delta = 0;
                        const bsample bs = bd_sample(s, dir, normal, fnormal, mat_id, mat_type, c.seed, pixel, frame, BD_DIM_EYE + 8u * (uint32_t)depth, delta, spectral, Lambda);
                        e->delta = delta;
                        pdfFwd = bs.pdfFwd;
                        if (pdfFwd > 0.0f) {
                            if (mat_type == MAT_GLASS) {
                                pdfRev = 0.0f; pdfFwd = 0.0f;
                                beta = beta * (reflect_color * bs.brdf);
                            } else {
                                beta = beta * (((reflect_color * bs.brdf) * absf(dot(normal, bs.next_dir))) / pdfFwd);
                                pdfRev = disney_pdf(m, fnormal, bs.next_dir, -dir);
                            }
                            eye.set_rpdf(pre_depth, pdfRev * absf(dot(to, e->normal)) * inv_dist2);
                            bool killed = false;
                            if (!spectral && bs.f_or_b < 0.0f) {          // (SPEC has no extinction roulette)
                                const float R = tm_exp(-sh.t / m[6]);
                                if (tm_rand(c.seed, pixel, frame, BD_DIM_EYE + 8u * (uint32_t)depth + TM_SLOT_EXT) >= R) killed = true;
                            }
                            if (!killed) {
                                final_depth = depth + 1;
                                go_on = (depth + 1 < BD_EYE_MAX);
                                next_o = offset_ray(pos, fnormal * signf(bs.f_or_b));
                                next_d = bs.next_dir;
                            }
                        }
                    }
                    eye.put(depth, ev);                   // five 16-byte stores
                }
                steps.eb[it] = make_float4(beta.x, beta.y, beta.z, pdfFwd); steps.ed[it] = final_depth | ((stored_surface && final_depth == depth) ? 1 << 16 : 0);
            } else {
                const varr light = B.light;
                const float4 sq = steps.lb[it];
                float pdfFwd = sq.w, pdfRev = 0.0f;
                v3 beta = V(sq.x, sq.y, sq.z);
                if (sh.t < INF_VALUE) {
                    int mat_id;
                    const HitAttr h = hit_attributes_rec(s.shade_rec, origin, dir, sh.prim, sh.t, sh.u, sh.v, mat_id);
                    const v3 normal = h.nor, pos = h.pos;
                    const v3 fnormal = normal * signf(dot(-dir, h.gnor));
                    const float *m = mat_row(s, mat_id);
                    const int mat_type = (int)m[0];
                    if (mat_type != MAT_LIGHT) {
                        bvert lv = bvert(); bvert *L = &lv;
                        L->pos = pos; L->normal = normal; L->snormal = fnormal; L->beta = beta * absf(dot(dir, normal));
                        L->wo = dir; L->fpdf = pdfFwd; L->rpdf = 0.0f; L->type = VERTEX_SURFACE; L->prim = sh.prim; L->mat = mat_id;
                        v3 to = pos - light[pre_depth].pos;
                        const float dist = norm(to);
                        const float inv_dist2 = 1.0f / (dist * dist);
                        to = to / dist;
                        L->fpdf *= absf(dot(to, light[pre_depth].normal)) * inv_dist2;
                        const v3 reflect_color = spectral ? bd_reflect<SPEC>(c, mat_id, Lambda) : mat_lrgb(s, mat_id);
                        int delta = 0;
                        const bsample bs = bd_sample(s, dir, normal, fnormal, mat_id, mat_type, c.seed, pixel, frame, BD_DIM_LIGHT + 8u * (uint32_t)depth, delta, spectral, Lambda);
                        L->delta = delta;
                        pdfFwd = bs.pdfFwd;
                        if (pdfFwd > 0.0f) {
                            if (mat_type == MAT_GLASS) {
                                pdfRev = 0.0f; pdfFwd = 0.0f;
                                beta = beta * (reflect_color * bs.brdf);
                            } else {
                                beta = beta * (((reflect_color * bs.brdf) * absf(dot(normal, bs.next_dir))) / pdfFwd);
                                pdfRev = disney_pdf(m, fnormal, bs.next_dir, -dir);
                            }
                            light.set_rpdf(pre_depth, pdfRev * absf(dot(to, L->normal)) * inv_dist2);
                            bool killed = false;
                            if (!spectral && bs.f_or_b < 0.0f) {
                                const float R = tm_exp(-sh.t / m[6]);
                                if (tm_rand(c.seed, pixel, frame, BD_DIM_LIGHT + 8u * (uint32_t)depth + TM_SLOT_EXT) >= R) killed = true;
                            }
                            if (!killed) {
                                final_depth = depth + 1;
                                go_on = (depth + 1 < BD_LIGHT_MAX);
                                next_o = offset_ray(pos, fnormal * signf(bs.f_or_b));
                                next_d = bs.next_dir;
                            }
                        }
                        light.put(depth, lv);
                    }
                }
                steps.lb[it] = make_float4(beta.x, beta.y, beta.z, pdfFwd); steps.ld[it] = final_depth;
            }
        }
    }
    // the survivors' places in the next depth's list: ONE atomic per block (same-address atomics retire at ~11 ns each on MI355X: one per wave -- 262 k of them
    // at depth 1 of a 8 Mi-item batch -- bounded this kernel at 2.9 ms of its 6.4; the ray count, which was a second such stream, is the list's length)
    __shared__ int s_wc[BD_STEP_BLOCK / 64], s_base;
    const unsigned long long gm = __ballot(go_on);
    const int wid = threadIdx.x >> 6;
    if (lane == 0) s_wc[wid] = __popcll(gm);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < BD_STEP_BLOCK / 64; w++) tot += s_wc[w];
        s_base = tot ? atomicAdd(&alive_cnt[depth + 1], tot) : 0;
    }
    __syncthreads();
    if (go_on) {
        int base = s_base;
#pragma unroll
        for (int w = 0; w < BD_STEP_BLOCK / 64; w++) if (w < wid) base += s_wc[w];
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const size_t qo = (size_t)(base + __popcll(gm & lt));
        put_ray(rays_out, qo, next_o, next_d); owner_out[qo] = t;
    }
    if (qi == 0) atomicAdd(rays_closest, (unsigned long long)alive_cnt[depth]);          // every entry of this depth's list was a traced ray
}

// The `delta` field of an eye vertex that ended on a light is whatever an earlier frame of the same pixel left in that
// slot (file header): replayed per pixel, in frame order, over the persistent per-pixel memory.
__global__ void k_bd_delta(BdItems items, BdSteps steps, TileMap tm, int P, int F, int *delta_mem)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    int *mem = delta_mem + (size_t)local_to_pixel(tm, k) * 8;
    for (int f = 0; f < F; f++) {
        const size_t it = (size_t)frame_pixel_to_slot(tm, P, f, k);
        const varr eye = bd_item(items, it).eye;
        const int edt = steps.ed[it], ed = edt & 0xffff;
        for (int v = 1; v < ed; v++) {
            const float4 td = *eye.at(v, 4);                     // (wo, type | delta << 16)
            const int type = (int)(short)(__float_as_uint(td.w) & 0xffffu), delta = (int)(short)(__float_as_uint(td.w) >> 16);
            if (type == VERTEX_SURFACE) mem[v] = delta;
            else if (type == VERTEX_LIGHT) eye.set_delta(v, mem[v]);
        }
        // v == ed: a surface vertex whose sampling ended the path (pdf 0, or the extinction roulette of a refraction) is not counted
        // in the depth, but its delta has been stored (BDPT_RGB.py:160-187); the tail bit of BdSteps::ed says that slot is this item's
        // (the vertex arrays are not cleared between batches)
        if (ed < BD_EYE_MAX && (edt >> 16)) mem[ed] = eye.delta(ed);
    }
}

// AddSplat (BDPT_RGB.py:594-613; SPEC: BDPT_SPEC.py:178-181): a contribution goes to the film pixel of its sample, or -- a light sub-path
// vertex seen through the lens (e == 1) -- to the pixel it projects to, with float atomics.
// where a contribution goes (pixel index, -1: nowhere) and, SPEC, what arrives there: the sensor's response at the wavelength, as clamped sRGB, times the range
template <bool SPEC>
TD long bd_splat_target(const BdCtx &c, int e, int nu, int nv, int p, uint32_t frame, v3 &r)
{
    const long q = (e == 1) ? ((nu >= 0) ? (long)nu * c.bv.H + nv : -1) : (long)p;
    if (SPEC) {
        const SpecView &sp = *c.spec;
        const v3 xyz = sensor_sample(sp, bd_lambda(c, (uint32_t)p, frame));
        const float range = sp.s_max - sp.s_min;
        const float cr = (3.240479f * xyz.x + -1.537150f * xyz.y) + -0.498535f * xyz.z;
        const float cg = (-0.969256f * xyz.x + 1.875991f * xyz.y) + 0.041556f * xyz.z;
        const float cb = (0.055648f * xyz.x + -0.204043f * xyz.y) + 1.057311f * xyz.z;
        r = V((clampf(cr, 0.0f, 1000.0f) * range) * r.x, (clampf(cg, 0.0f, 1000.0f) * range) * r.x, (clampf(cb, 0.0f, 1000.0f) * range) * r.x);
    }
    return q;
}
TD void bd_splat_add(float *rad, long q, v3 r)
{
#ifdef BDX_NO_ATOMIC
    if (q >= 0 && r.x == -12345.0f && r.y == -2.0f && r.z == -3.0f) {
#else
    if (q >= 0 && (r.x != 0.0f || r.y != 0.0f || r.z != 0.0f)) {
#endif
        atomicAdd(&rad[3 * q], r.x); atomicAdd(&rad[3 * q + 1], r.y); atomicAdd(&rad[3 * q + 2], r.z);
    }
}
template <bool SPEC>
TD void bd_splat(const BdCtx &c, float *rad, int e, int nu, int nv, int p, uint32_t frame, v3 r)
{
    const long q = bd_splat_target<SPEC>(c, e, nu, nv, p, frame, r);
    bd_splat_add(rad, q, r);
}

constexpr int BD_OWNER_BITS = 26;                  // a queued connection's owner word: item | pair slot << 26 (49 slots; a batch holds < 2^26 items)

// BDPT_RGB.py:615-637, the double loop over (e, l), per item: the geometry of every connection.  A pair that needs a visibility ray
// stages it (k_bd_compact makes the queue dense, k_trace answers it, k_bd_resolve -- one thread per QUEUED CONNECTION -- adds the
// contribution); the pairs that need none (l == 0: the eye sub-path ended on an emitter) are k_bd_emitted's.
constexpr int BD_CONNECT_BLOCK = 256;
template <bool SPEC>
__global__ __launch_bounds__(BD_CONNECT_BLOCK) void k_bd_connect(BdCtx c, BdItems items, BdSteps steps, TileMap tm, int P, int N, uint32_t frame_begin,
                             float4 *stage, unsigned long long *qmask, int *ibase, int *icount, int *scount)
{
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = it < N;
    const int lane = threadIdx.x & 63;
    int eye_depth = 0, light_depth = -1, i = 0, j = 0, p = 0, f = 0;
    const bpixel B = bd_item(items, (size_t)(live ? it : 0));
    if (live) {
        int k; slot_to_frame_pixel(tm, P, it, f, k);
        p = local_to_pixel(tm, k); i = p / c.bv.H; j = p - i * c.bv.H;
        eye_depth = steps.ed[it] & 0xffff; light_depth = steps.ld[it];
    }
    const uint32_t frame = frame_begin + (uint32_t)f;
    unsigned emitted = 0;
    // which (e, l) pairs of this item carry a connection ray: one bit per pair slot (49 of them).  The rays are staged in slot order, so
    // k_bd_compact finds the slot of the item's j-th ray as the j-th set bit
    unsigned long long pairs = 0ull;
    bvert sample = bvert();              // BDPT_RGB.py:60 `sample`: written by a connection, read by its MIS weight
    for (int e = 1; e <= BD_EYE_MAX; e++) {
        if (__ballot(live && e <= eye_depth) == 0ull) break;
        bvert EV = bvert();
        if (live && e <= eye_depth) EV = B.eye[e - 1];                 // once per e, not once per pair (the kernel is bound by its vector-memory instructions)
        for (int l = 0; l <= BD_LIGHT_MAX; l++) {
            const int depth = l + e - 2;
            if (((l == 1) & (e == 1)) | (depth < 0) | (depth > BD_MAX_DEPTH)) continue;       // wave-uniform
            const bool valid = live && e <= eye_depth && l <= light_depth;
            if (!valid || l == 0) continue;
            const int slot = (e - 1) * (BD_LIGHT_MAX + 1) + l;
            Tracer T; T.phase = 0; T.want = false; T.res.t = INF_VALUE; T.res.u = 0.0f; T.res.v = 0.0f; T.res.prim = -1;
            T.o = V(0.0f, 0.0f, 0.0f); T.d = T.o; T.expect = -3; T.bound = -1.0f;
            int nu = 0, nv = 0;
            (void)bd_connect_path<SPEC>(c, B, EV, sample, i, j, e, l, frame, nu, nv, T);       // pass 0: a traced pair sees a miss and returns zero
            if (T.want) {
                // the item's j-th connection ray goes to staging slot [j][item] (k_bd_compact makes the queue dense: one atomic
                // per wave at the end of this kernel instead of one per wave and pair -- same-address atomics retire at ~11 ns)
                const int local = (int)emitted++;
                pairs |= 1ull << slot;
                const size_t k = (size_t)local * (size_t)N + it;
                // one 32-byte record per staged ray (two 16-byte stores): as six + two scattered words this kernel was bound by its
                // store REQUESTS (texture-address unit 0.89 busy, 34 write requests per item)
                stage[2 * k] = make_float4(T.o.x, T.o.y, T.o.z, T.d.x);
                stage[2 * k + 1] = make_float4(T.d.y, T.d.z, __int_as_float(T.expect), T.bound);
            }
        }
    }
    // dense queue positions of this wave's rays: [base, base + total), lane by lane
    int incl = (int)emitted;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    const int total = __shfl(incl, 63, 64);
    // one atomic per block for the waves' places (k_bd_compact adds the queue's length to the ray counter)
    __shared__ int s_wt[BD_CONNECT_BLOCK / 64], s_cb;
    const int wid = threadIdx.x >> 6;
    if (lane == 0) s_wt[wid] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < BD_CONNECT_BLOCK / 64; w++) tot += s_wt[w];
        s_cb = tot ? atomicAdd(scount, tot) : 0;
    }
    __syncthreads();
    int base = s_cb;
#pragma unroll
    for (int w = 0; w < BD_CONNECT_BLOCK / 64; w++) if (w < wid) base += s_wt[w];
    if (live) { ibase[it] = base; icount[it] = (int)emitted; qmask[it] = pairs; }          // the wave's place in the dense queue; k_bd_compact orders it slot by slot
}

// The l == 0 pairs (BDPT_RGB.py:489-491): an eye vertex that lies on an emitter contributes its beta, MIS-weighted.  A sub-path ends on
// the emitter it meets (:152-158), so per item only e = eye_depth can be one; the light sub-path's depth does not matter (l = 0 <= any).
template <bool SPEC>
__global__ void k_bd_emitted(BdCtx c, BdItems items, BdSteps steps, TileMap tm, int P, int N, uint32_t frame_begin, float *radiance, long frame_stride)
{
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= N) return;
    const int e = steps.ed[it] & 0xffff;
    if (e < 2 || e > BD_EYE_MAX || e - 2 > BD_MAX_DEPTH) return;
    const bpixel B = bd_item(items, (size_t)it);
    if (B.eye[e - 1].type != VERTEX_LIGHT) return;          // depth = e - 2 in 0..BD_MAX_DEPTH
    int f, k; slot_to_frame_pixel(tm, P, it, f, k);
    const int p = local_to_pixel(tm, k), i = p / c.bv.H, j = p - i * c.bv.H;
    const uint32_t frame = frame_begin + (uint32_t)f;
    Tracer T; T.phase = 1; T.want = false; T.res.t = INF_VALUE; T.res.u = 0.0f; T.res.v = 0.0f; T.res.prim = -1;
    T.o = V(0.0f, 0.0f, 0.0f); T.d = T.o; T.expect = -3; T.bound = -1.0f;
    bvert sample = bvert();
    int nu = 0, nv = 0;
    const bvert EV = B.eye[e - 1];
    const v3 r = bd_connect_path<SPEC>(c, B, EV, sample, i, j, e, 0, frame, nu, nv, T);
    bd_splat<SPEC>(c, radiance + (size_t)f * (size_t)frame_stride, e, nu, nv, p, frame, r);
}

// Pass 1 of a connection, one thread per queued connection ray (per item, 27 pair slots of which at most 20 and on average ~7 carry a ray and ~4 of those are
// unoccluded, the VALU ran at 19 % of its lanes): the traced answer, then -- only if the expected primitive is what the ray met --
// contribution and MIS weight (BDPT_RGB.py:300-479), splatted with float atomics.
//
// Every branch of connect_path wants the ray's closest hit to be the expected primitive before anything else (and about half of the connections
// are occluded), so a block first lists the unoccluded ones of BD_RESOLVE_ITEMS items in LDS and then works through the list with all its lanes.
// The list is ITEM by item (round 5): the four or so connections of an item sit in neighbouring lanes, a wave of 64 connections belongs to ~16
// neighbouring items, and the ~22 vertex quads a connection reads are lines that the same wave-instruction's other lanes want too -- ~130 lines
// per 64 connections, each fetched once.  Rounds 3-4 listed a 2 048-entry chunk of the queue in QUEUE order (the j-th rays of 64 items side by side):
// a wave's lanes then belonged to ~110 items and every round over the same items fetched their lines again (43.5 GB per 8 Mi-item launch for 8.7 GB
// of vertices: the working set of the blocks in flight is four times an XCD's L2).
constexpr int BD_RESOLVE_ITEMS = 128;
static_assert(BD_RESOLVE_ITEMS == 128, "k_bd_resolve's list entries hold the item in seven bits");             // items a block lists at a time: two of k_bd_connect's waves, one per wave of the block
#ifndef BD_RESOLVE_WAVES
#define BD_RESOLVE_WAVES 4
#endif
#ifdef BD_RESOLVE_VGPR
#define BD_RESOLVE_BOUNDS __launch_bounds__(BD_RESOLVE_ITEMS) __attribute__((amdgpu_waves_per_eu(BD_RESOLVE_VGPR, BD_RESOLVE_VGPR)))
#else
#define BD_RESOLVE_BOUNDS __launch_bounds__(BD_RESOLVE_ITEMS, BD_RESOLVE_WAVES)
#endif      // (said so, the compiler schedules the kernel within 128 VGPRs a little better: config 5 + 2 %, three A/B pairs)
template <bool SPEC>
__global__ BD_RESOLVE_BOUNDS void k_bd_resolve(BdCtx c, BdItems items, TileMap tm, int P, int N, uint32_t frame_begin, const int *ibase, const int *icount,
                             const unsigned long long *qmask, const float4 *shits, const float4 *stage, float *radiance, long frame_stride)
{
    // the unoccluded connections, item by item: item of the group | pair slot << 7 | queue place from the base of the item's wave on << 13 (< 64 x 20).  One word each: at
    // eight bytes the list alone (20 KB) held a CU to seven blocks
    __shared__ unsigned s_list[BD_RESOLVE_ITEMS * BD_RAY_PAIRS];
    __shared__ int s_wn[BD_RESOLVE_ITEMS / 64], s_off0[BD_RESOLVE_ITEMS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int groups = (N + BD_RESOLVE_ITEMS - 1) / BD_RESOLVE_ITEMS;
    for (int g = blockIdx.x; g < groups; g += (int)gridDim.x) {
        // a wave takes the 64 items of one of k_bd_connect's waves, lane = item, and walks their rays the way k_bd_compact numbered them: the
        // j-th rays of the items that have one are consecutive queue places from the wave's base on
        const int it = g * BD_RESOLVE_ITEMS + (int)threadIdx.x;
        const bool live = it < N;
        const int n = live ? icount[it] : 0;
        const unsigned long long pairs = (n > 0) ? qmask[it] : 0ull;
        int off0 = live ? ibase[it] : 0;
        off0 = __shfl(off0, 0, 64);
        unsigned ok = 0u;                                  // bit j: my j-th ray met the primitive it was aimed at
        int off = off0;
        for (int j = 0; j < BD_RAY_PAIRS; j++) {
            const unsigned long long m = __ballot(n > j);
            if (m == 0ull) break;
            if (n > j) {
                const size_t q = (size_t)(off + __popcll(m & lt)), k = (size_t)j * (size_t)N + (size_t)it;
                if (__float_as_int(shits[q].w) == __float_as_int(stage[2 * k + 1].z)) ok |= 1u << j;
            }
            off += __popcll(m);
        }
        const int mine = __popc(ok);
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) { s_wn[wid] = incl; s_off0[wid] = off0; }
        __syncthreads();
        int w = incl - mine, total = 0;
#pragma unroll
        for (int x = 0; x < BD_RESOLVE_ITEMS / 64; x++) { if (x < wid) w += s_wn[x]; total += s_wn[x]; }
        unsigned long long rest = pairs;
        off = off0;
        for (int j = 0; j < BD_RAY_PAIRS; j++) {
            const unsigned long long m = __ballot(n > j);
            if (m == 0ull) break;
            if (n > j) {
                const int slot = __ffsll((long long)rest) - 1;
                rest &= rest - 1ull;
                if ((ok >> j) & 1u) s_list[w++] = threadIdx.x | ((unsigned)slot << 7) | ((unsigned)(off - off0 + __popcll(m & lt)) << 13);
            }
            off += __popcll(m);
        }
        __syncthreads();
        for (int k0 = 0; k0 < total; k0 += (int)blockDim.x) {
          const int k = k0 + (int)threadIdx.x;
          long long key = -1; v3 r = V(0.0f, 0.0f, 0.0f);           // the word of the batch's radiance planes the contribution goes to / 3
          if (k < total) {
            const unsigned en = s_list[k];
            const int il = (int)(en & 127u), slot = (int)((en >> 7) & 63u), ci = g * BD_RESOLVE_ITEMS + il;
            const float4 hr = shits[s_off0[il >> 6] + (int)(en >> 13)];
            const int e = slot / (BD_LIGHT_MAX + 1) + 1, l = slot - (e - 1) * (BD_LIGHT_MAX + 1);
            int f, kk; slot_to_frame_pixel(tm, P, ci, f, kk);
            const int p = local_to_pixel(tm, kk), i = p / c.bv.H, j = p - i * c.bv.H;
            const uint32_t frame = frame_begin + (uint32_t)f;
            Tracer T; T.phase = 1; T.want = false; T.res.t = hr.x; T.res.u = hr.y; T.res.v = hr.z; T.res.prim = __float_as_int(hr.w);
            T.o = V(0.0f, 0.0f, 0.0f); T.d = T.o; T.expect = -3; T.bound = -1.0f;
            bvert sample = bvert();
            int nu = 0, nv = 0;
            const bpixel B = bd_item(items, (size_t)ci);
            const bvert EV = B.eye[e - 1];
            r = bd_connect_path<SPEC>(c, B, EV, sample, i, j, e, l, frame, nu, nv, T);
            const long q = bd_splat_target<SPEC>(c, e, nu, nv, p, frame, r);
            if (q >= 0) key = (long long)f * (long long)(frame_stride / 3) + q;
          }
          // The e >= 2 connections of an item all add to the item's own pixel and sit in neighbouring lanes: one atomic per run of equal targets instead of one per
          // connection (a third of the atomics; the kernel is 19 % faster without any).  A run's sum is formed pairwise, so the film differs from per-connection
          // atomics in the last bits -- as it does from run to run anyway (header).
#ifndef BDX_NO_PRESUM
          {
            const long long kp = __shfl_up(key, 1, 64);
            const unsigned long long heads = __ballot(lane == 0 || kp != key);
            const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
            const int run = above ? (__ffsll((long long)above) - 1) : (63 - lane);        // lanes after mine with my target
#pragma unroll
            for (int d = 1; d <= 32; d <<= 1) {            // a run is usually the <= 20 connections of one item, but e == 1 connections of the NEXT items can project onto the same (frame, pixel) and extend it: d = 32 sums any run a wave can hold (with d <= 16 a run of more than 32 lanes would have lost its tail)
                const float vx = __shfl_down(r.x, d, 64), vy = __shfl_down(r.y, d, 64), vz = __shfl_down(r.z, d, 64);
                if (d <= run) { r.x += vx; r.y += vy; r.z += vz; }
            }
            if (!((heads >> lane) & 1ull)) key = -1;
          }
#endif
          bd_splat_add(radiance, key, r);
        }
        __syncthreads();
    }
}

// staging slots [j][item] -> dense connection-ray queue.  The queue is a list of PLACES: a wave numbers the rays of its 64 items slot by slot
// (the j-th rays of the items that have one take consecutive queue places, so that k_trace's lanes read neighbouring staging records), and
// writes per place where the ray is staged (`qlist`, read by k_trace through TraceArgs::ray_index; k_bd_resolve walks the items' rays the same
// way and needs no list).  The rays themselves stay where they are (rounds 2-3a copied them: 128 B of traffic and 32 B of state per ray).
__global__ void k_bd_compact(int N, const int *ibase, const int *icount, int *qlist, const int *scount, unsigned long long *rays_shadow)
{
    const int it = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    if (it == 0 && *scount) atomicAdd(rays_shadow, (unsigned long long)*scount);          // the queue's length: every staged connection is a traced ray
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const bool live = it < N;
    const int n = live ? icount[it] : 0;
    int off = live ? ibase[it] : 0;
    off = __shfl(off, 0, 64);                           // lane 0 of a wave is live whenever any lane is
    for (int j = 0; j < BD_RAY_PAIRS; j++) {
        const unsigned long long m = __ballot(n > j);
        if (m == 0ull) break;
        if (n > j) {
            const size_t k = (size_t)j * (size_t)N + it, q = (size_t)(off + __popcll(m & lt_mask));
            qlist[q] = (int)k;                                             // the ray stays where it was staged: the queue is a list of places (k_trace reads through it)
        }
        off += __popcll(m);
    }
}

// BDPT_RGB.py:639-642
__global__ void k_bdpt_film(const float *radiance, float *hdr, long nvals, float coff)
{
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nvals) return;
    hdr[k] = radiance[k] * coff + hdr[k] * (1.0f - coff);
}

int bdpt_render(tirt_ctx *c, uint32_t frame_begin, int frame_count, uint32_t seed, bool spectral)
{
    TIRT_REQUIRE(c->built && c->cam_set && c->hdr.p, "tirt_bdpt_rgb_render: scene, camera and film must be set up");
    TIRT_REQUIRE(frame_count >= 0, "tirt_bdpt_rgb_render: bad frame_count");
    // the light sub-path starts on an emitter (Scene.sample_light, Scene.py:430-474): a scene without one would index light[-1]
    TIRT_REQUIRE(c->light_count >= 1, "tirt_bdpt_rgb_render: the scene has no emitter (BDPT_RGB samples its light sub-path from one)");
    if (frame_count == 0) return TIRT_OK;
    if (ensure_counters(c)) return TIRT_ERR_HIP;
    if (sync_all(c)) return TIRT_ERR_HIP;
    if (ensure_shade_records(c)) return TIRT_ERR_HIP;
    const long NP = (long)c->W * c->H;
    const int P = (int)c->npix_local;
    hipStream_t st = c->stream;
    if (c->bdpt_px.bytes < sizeof(int) * 8 * (size_t)NP) {         // the per-pixel delta memory (zero = what a fresh film starts from)
        if (c->bdpt_px.ensure(sizeof(int) * 8 * (size_t)NP)) return TIRT_ERR_HIP;
        TIRT_HIP(hipMemsetAsync(c->bdpt_px.p, 0, sizeof(int) * 8 * (size_t)NP, st));
    }
    if (P == 0) return TIRT_OK;
    // Two batches in flight on the streams of render lanes 0 and 1 (when the job has more than one batch): the traversal launches
    // of one (VALU-bound) run next to the vertex / connection kernels of the other (HBM-bound).  What is order dependent --
    // k_bd_delta's per-pixel memory and the running mean of the film -- is chained through events, batch after batch.
    int FB = (int)(c->bdpt_batch_items / (size_t)P); if (FB < 1) FB = 1; if (FB > frame_count) FB = frame_count;
    int NL = 1;
    if (c->n_lanes >= 2 && !c->time_kernels && frame_count >= 2 && (size_t)frame_count * P >= ((size_t)1 << 20)) {
        NL = c->bdpt_lanes < c->n_lanes ? c->bdpt_lanes : c->n_lanes;
        if (NL > frame_count) NL = frame_count;
        int share = (int)(c->bdpt_batch_items / (size_t)NL / (size_t)P); if (share < 1) share = 1;
        if (FB > share) FB = share;                                 // the lanes share the batch budget
        if (FB > (frame_count + NL - 1) / NL) FB = (frame_count + NL - 1) / NL;
    }
    // the traversal's own buffers first (stack spill: ~2 GB per lane at bdpt_stack_size 1024), so that the measurement below sees them and a
    // failed allocation leaves the film untouched (they used to be allocated inside the batch loop, after the guard)
    for (int l = 0; l < NL; l++) if (int rc = trace_arrays_prepare(c, NL > 1 ? l : -1)) return rc;
    {
        // not more than the device has free right now (plus what this context's BDPT buffers hold already: growing them frees them first), less
        // 2 GB: a batch half the size is a few per cent slower, a failed hipMalloc ends the render (bench.py's profiler child, next to the
        // contexts of the other configs, ran into exactly that with 16 Mi-item batches)
        const size_t per_item = BD_ITEM_BYTES + BD_STEP_BYTES + sizeof(float) * (16 + 8 * BD_RAY_PAIRS) + sizeof(float4) * (2 + BD_RAY_PAIRS) + sizeof(int) * (4 + BD_RAY_PAIRS);       // = the ensure() calls below
        size_t free_b = 0, total_b = 0, held = 0;
        for (int l = 0; l < 4; l++) held += c->bd[l].items.bytes + c->bd[l].state.bytes + c->bd[l].rays.bytes + c->bd[l].hits.bytes + c->bd[l].qidx.bytes + c->bd[l].rad.bytes;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            size_t avail = free_b + held > ((size_t)2 << 30) ? free_b + held - ((size_t)2 << 30) : 0;
            if (c->bdpt_mem_budget && avail > c->bdpt_mem_budget) avail = c->bdpt_mem_budget;      // option "bdpt_mem_budget": pretend the device has only this much left
            while (FB > 1 && (size_t)NL * (size_t)FB * ((size_t)P * per_item + sizeof(float) * 3 * (size_t)NP) > avail) FB = (FB + 1) / 2;
        } else (void)hipGetLastError();
    }
    { const int nb = (frame_count + FB - 1) / FB; FB = (frame_count + nb - 1) / nb; }       // batches of equal size
    // buffers for the batches of THIS call (a frame-at-a-time caller gets one-frame buffers; DevBuf::ensure only ever grows, so a caller
    // whose calls grow re-allocates a few times at most).  Round 2 reserved for the whole job from the "job_frames" hint -- 31 GB for
    // an example loop that renders one frame per call and never merges calls.
    const int FB_alloc = FB;
    const size_t NMAX = (size_t)FB_alloc * P;
    TIRT_REQUIRE(NMAX < ((size_t)1 << BD_OWNER_BITS), "tirt_bdpt_rgb_render: bdpt_batch_items must stay below 2^26");
    TIRT_REQUIRE(NMAX * BD_PAIRS < ((size_t)1 << 31), "tirt_bdpt_rgb_render: film too large for one frame per batch");
    const size_t SCAP = NMAX * BD_RAY_PAIRS;                     // at most 20 (e, l) pairs per item carry a connection ray (staging: [20][N])
    for (int l = 0; l < NL; l++) {
        auto &bl = c->bd[l];
        if (bl.items.ensure(BD_ITEM_BYTES * NMAX) || bl.state.ensure(BD_STEP_BYTES * NMAX) ||
            bl.rays.ensure(sizeof(float) * (8 * 2 * NMAX + 8 * SCAP)) || bl.hits.ensure(sizeof(float4) * (2 * NMAX + SCAP)) ||
            bl.qidx.ensure(sizeof(int) * (NMAX * 4 + SCAP)) || bl.ctr.ensure(256) ||
            bl.rad.ensure(sizeof(float) * 3 * (size_t)NP * (size_t)FB_alloc)) return TIRT_ERR_HIP;
        if (!bl.delta_done) TIRT_HIP(hipEventCreateWithFlags(&bl.delta_done, hipEventDisableTiming));
        if (!bl.film_done) TIRT_HIP(hipEventCreateWithFlags(&bl.film_done, hipEventDisableTiming));
    }
    BdCtx bc;
    bc.sc = scene_view(c); bc.cam = c->cam; bc.seed = seed; bc.bounded = c->bdpt_bounded;
    bc.spec = spectral ? c->spec_dev.as<SpecView>() : nullptr;
    for (int k = 0; k < 12; k++) bc.bv.view[k] = c->view[k];
    bc.bv.W = c->W; bc.bv.H = c->H;
    DevCounters *ctr = c->dev_counters.as<DevCounters>();
    TileMap tm = {c->tile_rank, c->tile_count, c->tile_size, c->H, c->tile_blocked, 0};
    const int B = 128;
    // everything queued on the main stream so far precedes the lanes' work
    if (NL > 1) { TIRT_HIP(hipEventRecord(c->ev_main, c->stream)); for (int l = 0; l < NL; l++) TIRT_HIP(hipStreamWaitEvent(c->lanes[l].stream, c->ev_main, 0)); }
    hipEvent_t last_delta = nullptr, last_film = nullptr;
    int batch = 0;
    for (int f0 = 0; f0 < frame_count; f0 += FB, batch++) {
        const int lane = NL > 1 ? (batch % NL) : -1;
        auto &bl = c->bd[NL > 1 ? (batch % NL) : 0];
        hipStream_t st = lane < 0 ? c->stream : c->lanes[lane].stream;
        float *rf = bl.rays.as<float>();
        BdRays er = {(float4 *)rf};                                  // 2 N records
        float *gf = rf + 16 * NMAX;                                  // staging area, [slot j][item]: 8 SCAP = 160 NMAX words
        float4 *stage = (float4 *)gf;                               // [slot j][item]: 32-byte records (o, d, expect, bound); k_trace reads them where they lie
        // during the sub-path phase nothing is staged yet: the area holds the second ray list (2 N records) and the two owner lists (2 N ints each)
        BdRays sr = {(float4 *)gf};
        int *sexpect = (int *)(gf + 16 * NMAX), *gexpect = (int *)(gf + 18 * NMAX);
        unsigned long long *qmask = bl.qidx.as<unsigned long long>();          // [item]: the pairs that have a connection ray
        int *ibase = bl.qidx.as<int>() + NMAX * 2, *icount = ibase + NMAX;
        int *qlist = icount + NMAX;                                   // [queue place]: where the ray is staged (j * N + item)
        float4 *ehits = bl.hits.as<float4>(), *shits = ehits + 2 * NMAX;
        int *scount = bl.ctr.as<int>();
        const int F = frame_count - f0 < FB ? frame_count - f0 : FB;
        const int N = F * P;
        tm.F = (c->path_order_blocks && (P & 63) == 0) ? F : 0;          // items numbered pixel-block major (tirt_internal.h, TileMap::F): the frames of a pixel block are neighbours in every ray list
        const BdItems items = {bl.items.as<float4>(), (size_t)N};          // (the batch's items side by side: stride = their number)
        const BdSteps state = bd_steps(bl.state.p, (size_t)N);
        const uint32_t frame0 = frame_begin + (uint32_t)f0;
        TIRT_HIP(hipMemsetAsync(bl.rad.p, 0, sizeof(float) * 3 * (size_t)NP * (size_t)F, st));
        // No read of a vertex field goes to a slot this item has not written (header; k_bd_delta supplies the one exception), so the
        // 1.0 KB per item need no clearing.  Option "bdpt_state_fill": 1 = zeros (the round-1..3 behaviour), 2 = 0xFF poison -- the
        // parity tests render under poison and must not see a bit change.
        if (c->bdpt_state_fill) TIRT_HIP(hipMemsetAsync(bl.items.p, c->bdpt_state_fill == 2 ? 0xFF : 0, BD_ITEM_BYTES * (size_t)N, st));
        TIRT_HIP(hipMemsetAsync(scount, 0, 64, st));                 // the connection-ray count and the alive counts of the depths
        // during the sub-path phase the dense connection-ray arrays and the two `expect` arrays are free: they hold the second ray list and the owners
        BdRays rset[2] = {sr, er};                                  // depth d reads rset[d & 1]
        int *oset[2] = {sexpect, gexpect};
        int *alive_cnt = scount + 4;
        if (spectral) hipLaunchKernelGGL(k_bd_init<true>, dim3((N + B - 1) / B), dim3(B), 0, st, bc, items, state, rset[1], oset[1], alive_cnt, tm, P, N, frame0, &ctr->paths);
        else hipLaunchKernelGGL(k_bd_init<false>, dim3((N + B - 1) / B), dim3(B), 0, st, bc, items, state, rset[1], oset[1], alive_cnt, tm, P, N, frame0, &ctr->paths);
        // rays of the two sub-paths share the launches
        for (int d = 1; d < BD_EYE_MAX; d++) {
            const BdRays &ri = rset[d & 1], &ro = rset[(d + 1) & 1];
            if (int rc = trace_arrays(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 2 * N, alive_cnt + d, ehits, nullptr, nullptr, false, lane, ri.r, false)) return rc;
            if (spectral) hipLaunchKernelGGL(k_bd_step<true>, dim3((2 * N + BD_STEP_BLOCK - 1) / BD_STEP_BLOCK), dim3(BD_STEP_BLOCK), 0, st, bc, items, state, ri, oset[d & 1], ro,
                               oset[(d + 1) & 1], alive_cnt, ehits, tm, P, N, frame0, d, &ctr->rays_closest);
            else hipLaunchKernelGGL(k_bd_step<false>, dim3((2 * N + BD_STEP_BLOCK - 1) / BD_STEP_BLOCK), dim3(BD_STEP_BLOCK), 0, st, bc, items, state, ri, oset[d & 1], ro,
                               oset[(d + 1) & 1], alive_cnt, ehits, tm, P, N, frame0, d, &ctr->rays_closest);
        }
        if (last_delta) TIRT_HIP(hipStreamWaitEvent(st, last_delta, 0));      // the per-pixel memory is replayed in frame order
        hipLaunchKernelGGL(k_bd_delta, dim3((P + B - 1) / B), dim3(B), 0, st, items, state, tm, P, F, c->bdpt_px.as<int>());
        if (NL > 1) { TIRT_HIP(hipEventRecord(bl.delta_done, st)); last_delta = bl.delta_done; }
        if (spectral) hipLaunchKernelGGL(k_bd_connect<true>, dim3((N + BD_CONNECT_BLOCK - 1) / BD_CONNECT_BLOCK), dim3(BD_CONNECT_BLOCK), 0, st, bc, items, state, tm, P, N, frame0,
                           stage, qmask, ibase, icount, scount);
        else hipLaunchKernelGGL(k_bd_connect<false>, dim3((N + BD_CONNECT_BLOCK - 1) / BD_CONNECT_BLOCK), dim3(BD_CONNECT_BLOCK), 0, st, bc, items, state, tm, P, N, frame0,
                           stage, qmask, ibase, icount, scount);
        hipLaunchKernelGGL(k_bd_compact, dim3((N + B - 1) / B), dim3(B), 0, st, N, ibase, icount, qlist, scount, &ctr->rays_shadow);
        if (int rc = trace_arrays(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (int)(SCAP < (size_t)N * BD_RAY_PAIRS ? SCAP : (size_t)N * BD_RAY_PAIRS), scount, shits, nullptr, nullptr, false, lane, stage, true, qlist)) return rc;
        if (spectral) hipLaunchKernelGGL(k_bd_emitted<true>, dim3((N + B - 1) / B), dim3(B), 0, st, bc, items, state, tm, P, N, frame0, bl.rad.as<float>(), 3 * NP);
        else hipLaunchKernelGGL(k_bd_emitted<false>, dim3((N + B - 1) / B), dim3(B), 0, st, bc, items, state, tm, P, N, frame0, bl.rad.as<float>(), 3 * NP);
        {
            size_t rg = ((size_t)N + BD_RESOLVE_ITEMS - 1) / BD_RESOLVE_ITEMS;            // grid-stride over groups of BD_RESOLVE_ITEMS items
            if (rg > 8192) rg = 8192;
            if (spectral) hipLaunchKernelGGL(k_bd_resolve<true>, dim3((unsigned)rg), dim3(BD_RESOLVE_ITEMS), 0, st, bc, items, tm, P, N, frame0, ibase, icount, qmask, shits, stage, bl.rad.as<float>(), 3 * NP);
            else hipLaunchKernelGGL(k_bd_resolve<false>, dim3((unsigned)rg), dim3(BD_RESOLVE_ITEMS), 0, st, bc, items, tm, P, N, frame0, ibase, icount, qmask, shits, stage, bl.rad.as<float>(), 3 * NP);
        }
        if (last_film) TIRT_HIP(hipStreamWaitEvent(st, last_film, 0));        // the running mean applies the frames in order
        for (int f = 0; f < F; f++) {
            const float coff = 1.0f / ((float)(int)(frame0 + (uint32_t)f) + 1.0f);
            hipLaunchKernelGGL(k_bdpt_film, dim3((unsigned)((3 * NP + 255) / 256)), dim3(256), 0, st, bl.rad.as<float>() + (size_t)f * 3 * NP,
                               c->hdr.as<float>(), 3 * NP, coff);
        }
        if (NL > 1) { TIRT_HIP(hipEventRecord(bl.film_done, st)); last_film = bl.film_done; }
    }
    // what follows on the main stream (tone map, download, the next call) comes after both lanes
    if (NL > 1) for (int l = 0; l < NL; l++) {
        TIRT_HIP(hipEventRecord(c->bd[l].delta_done, c->lanes[l].stream));
        TIRT_HIP(hipStreamWaitEvent(c->stream, c->bd[l].delta_done, 0));
    }
    TIRT_HIP(hipGetLastError());
    return TIRT_OK;
}

}  // namespace tirt
