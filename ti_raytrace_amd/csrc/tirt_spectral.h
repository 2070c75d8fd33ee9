// tirt_spectral.h -- device-side restatement (HIP, gfx950) of what the hero-wavelength integrator PT_Spec needs beyond PT_RGB
// (SURVEY.md 8f rank 4): spectrum/Spectrum.py (tabulated spectra), spectrum/HeroSample.py (four wavelengths 100 nm apart per
// path), spectrum/Rgb2Spec.py (Jakob-Hanika sigmoid spectra from a 3 x res^3 coefficient table), sky/Sky.py (the analytic sky dome
// that is PT_Spec's environment), brdf/Glass.py:36-59 + UF.get_glass_ior (dispersion) and PT_Spec.AddSplat (CIE XYZ -> linear
// sRGB).  fp32 operation order follows the reference expression by expression, like tirt_device.h.
#pragma once
#include "tirt_device.h"

namespace tirt {

constexpr int MAT_SPECTRAL = 10;                        // SceneData.py:53
constexpr int HERO_N = 4;                               // spectrum/HeroSample.py:5
constexpr float HERO_LAMBDA_MIN = 360.0f, HERO_LAMBDA_STEP = (760.0f - 360.0f) / 4.0f;      // :6-8
constexpr uint32_t TM_DIM_SPEC_LAMBDA = 4000u;          // the path's hero wavelength (PT_Spec.py:191); bounce dimensions as PT_RGB
constexpr uint32_t TM_SLOT_HERO = 7u;                   // Hero.get_rnd_hero (HeroSample.py:33-35)
constexpr int SPEC_MAX_DEPTH = 10;                      // integrator/PT_Spec.py:26

struct Spd { const float *data; int n; float lmin, lmax, lrange; };
struct SpecView {
    const float *sensor; int n_sensor; float s_min, s_max, s_range;     // CIE 1931 observer rows (x, y, z), PT_Spec.py:56-77
    Spd spd[4];                                                          // d65 (normalised to Y = 1), white, red, green
    const float *tbl_scale, *tbl_data; int tbl_res;                     // Rgb2Spec.table_scale / table_data
    const float *sky_cfg, *sky_rad; float sun_dir[3];                   // Sky.configs [11][9], Sky.radiances [11], Sky.sun_dir
};
struct f4s { float v[HERO_N]; };
TD f4s f4_set(float x) { f4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = x; return r; }
TD f4s operator*(f4s a, f4s b) { f4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = a.v[i] * b.v[i]; return r; }
TD f4s operator*(f4s a, float k) { f4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = a.v[i] * k; return r; }
TD f4s operator/(f4s a, float k) { f4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = a.v[i] / k; return r; }
TD f4s operator+(f4s a, f4s b) { f4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
TD float fractf(float x) { return x - tm_floor(x); }                   // taichi_glsl.fract

// spectrum/Spectrum.py:44-52.  The interpolation weight is fract(offset), not fract(offset / range) -- kept; data[idx + 1] at
// Lambda == lambda_max would lie one past the table: read as the last entry.
TD float spd_sample(const Spd &d, float Lambda)
{
    float ret = 0.0f;
    if ((Lambda >= d.lmin) & (Lambda <= d.lmax)) {
        const float offset = Lambda - d.lmin;
        const int idx = (int)(offset / d.lrange);
        const float w = fractf(offset);
        const int i1 = idx + 1 < d.n ? idx + 1 : d.n - 1;
        ret = mixf(d.data[idx], d.data[i1], w);
    }
    return ret;
}
TD f4s hero_sample(const Spd &d, float Lambda0)                         // spectrum/HeroSample.py:10-16
{ f4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = spd_sample(d, Lambda0 + (float)i * HERO_LAMBDA_STEP); return r; }
TD v3 sensor_sample(const SpecView &sp, float Lambda)                   // integrator/PT_Spec.py:131-139
{
    v3 ret = V(0.0f, 0.0f, 0.0f);
    if ((Lambda >= sp.s_min) & (Lambda <= sp.s_max)) {
        const float offset = Lambda - sp.s_min;
        const int idx = (int)(offset / sp.s_range);
        const float w = fractf(offset);
        const int i1 = idx + 1 < sp.n_sensor ? idx + 1 : sp.n_sensor - 1;
        const float *a = sp.sensor + 3 * (size_t)idx, *b = sp.sensor + 3 * (size_t)i1;
        ret = V(mixf(a[0], b[0], w), mixf(a[1], b[1], w), mixf(a[2], b[2], w));
    }
    return ret;
}

// ---- spectrum/Rgb2Spec.py ------------------------------------------------------------------------------------------------
TD int r2s_find_interval(const SpecView &sp, int size, float x)        // :83-99
{
    int left = 0;
    const int last_interval = size - 2;
    size = last_interval;
    while (size > 0) {
        const int half = size >> 1, middle = left + half + 1;
        if (sp.tbl_scale[middle] <= x) { left = middle; size -= half + 1; }
        else size = half;
    }
    return left < last_interval ? left : last_interval;
}
TD float r2s_tri(const SpecView &sp, int i, float x0, float y0, float z0)      // :77-80
{
    const int dx = 3, dy = 3 * sp.tbl_res, dz = 3 * sp.tbl_res * sp.tbl_res;
    const float *t = sp.tbl_data;
    return mixf(mixf(mixf(t[i], t[i + dx], x0), mixf(t[i + dy], t[i + dy + dx], x0), y0),
                mixf(mixf(t[i + dz], t[i + dz + dx], x0), mixf(t[i + dy + dz], t[i + dx + dy + dz], x0), y0), z0);
}
TD v3 r2s_fetch(const SpecView &sp, v3 rgb)                             // :101-137 with get_max_component (:50-74)
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
    z = maxf(0.00001f, z);
    const float scale = (float)(sp.tbl_res - 1) / z;
    x *= scale; y *= scale;
    const int res = sp.tbl_res;
    const int xi = (int)minf(x, (float)(res - 2)), yi = (int)minf(y, (float)(res - 2));
    const int zi = r2s_find_interval(sp, res, z);
    const int offset = (((index * res + zi) * res + yi) * res + xi) * 3;
    const float x0 = x - (float)xi, y0 = y - (float)yi;
    const float z0 = (z - sp.tbl_scale[zi]) / (sp.tbl_scale[zi + 1] - sp.tbl_scale[zi]);
    return V(r2s_tri(sp, offset, x0, y0, z0), r2s_tri(sp, offset + 1, x0, y0, z0), r2s_tri(sp, offset + 2, x0, y0, z0));
}
TD float r2s_eval(v3 c, float Lambda)                                   // :139-143; its fma(a, b, c) is a * b + c
{
    const float x = (c.x * Lambda + c.y) * Lambda + c.z;
    const float y = 1.0f / tm_sqrt(x * x + 1.0f);
    return (0.5f * x) * y + 0.5f;
}
TD f4s srgb_to_spec(const SpecView &sp, v3 srgb, float Lambda0)         // spectrum/HeroSample.py:46-58
{
    const v3 coff = r2s_fetch(sp, srgb_to_lrgb(srgb));
    f4s r; for (int i = 0; i < HERO_N; i++) r.v[i] = r2s_eval(coff, Lambda0 + (float)i * HERO_LAMBDA_STEP);
    return r;
}
TD f4s emission_to_rad(const SpecView &sp, v3 emission, float Lambda)   // integrator/PT_Spec.py:102-109
{
    const float scale = norm(emission);
    f4s ret = f4_set(0.0f);
    if (scale > 0.0f) ret = srgb_to_spec(sp, emission / scale, Lambda);
    return ret * scale;
}
TD f4s get_spec_power(const SpecView &sp, const float *m, float Lambda)  // integrator/PT_Spec.py:111-127 (m: the material row)
{
    const int mat_type = (int)m[0], mat_tex = (int)m[1];
    f4s ret = f4_set(0.0f);
    if (mat_type == MAT_SPECTRAL) {
        if (mat_tex == 0) ret = hero_sample(sp.spd[1], Lambda);
        if (mat_tex == 1) ret = hero_sample(sp.spd[2], Lambda);
        if (mat_tex == 2) ret = hero_sample(sp.spd[3], Lambda);
    } else ret = srgb_to_spec(sp, V(m[2], m[3], m[4]), Lambda);
    return ret;
}

// ---- sky/Sky.py:176-186, 232-255: the sky dome without the sun's disc (solar_radiance_internal2 is commented out there) -------
TD float sky_internal(const SpecView &sp, int wl, float theta, float gamma)
{
    const float *c = sp.sky_cfg + 9 * wl;
    const float cg = tm_cos(gamma), ct = tm_cos(theta);
    const float expM = tm_exp(c[4] * gamma);
    const float rayM = cg * cg;
    const float mieM = (1.0f + cg * cg) / tm_pow((1.0f + c[8] * c[8]) - 2.0f * c[8] * cg, 1.5f);
    const float zenith = tm_sqrt(ct);
    return (1.0f + c[0] * tm_exp(c[1] / (ct + 0.01f))) * ((((c[2] + c[3] * expM) + c[5] * rayM) + c[6] * mieM) + c[7] * zenith);
}
TD float sky_radiance(const SpecView &sp, float theta, float gamma, float wavelength)
{
    float ret = 0.0f;
    if ((wavelength >= 320.0f) & (wavelength <= 720.0f)) {
        const int low_wl = (int)((wavelength - 320.0f) / 40.0f);
        float result = 0.0f;
        if ((low_wl >= 0) & (low_wl < 11)) {
            const float interp = fractf((wavelength - 320.0f) / 40.0f);
            const float val_low = sky_internal(sp, low_wl, theta, gamma) * sp.sky_rad[low_wl];
            if (interp < 1e-6f) result = val_low;
            else {
                result = (1.0f - interp) * val_low;
                if (low_wl + 1 < 11) result += interp * sky_internal(sp, low_wl + 1, theta, gamma) * sp.sky_rad[low_wl + 1];
            }
        }
        ret = result;
    }
    return ret;
}

// ---- brdf/Glass.py:36-59 with UF.get_glass_ior (UtilsFunc.py:481-484: BK7, Sellmeier) -----------------------------------------
TD float get_glass_ior(float Lambda)
{
    Lambda = Lambda / 1000.0f;
    const float L2 = Lambda * Lambda;
    return tm_sqrt(((1.0f + 1.03961212f * L2 / (L2 - 0.00600069867f)) + 0.231792344f * L2 / (L2 - 0.0200179144f)) + 1.01046945f * L2 / (L2 - 103.560653f));
}
TD v3 glass_sample_lambda(v3 dir, v3 N, float Lambda, float probability, float &f_or_b)
{
    const v3 w_out = dir;
    float cos_theta_i = dot(w_out, N);
    const float ior = get_glass_ior(Lambda);
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

// ---- integrator/PT_Spec.py:141-158 AddSplat: the four hero radiances -> CIE XYZ -> linear sRGB, mixed into the film word ----------
TD void spec_add_splat(const SpecView &sp, f4s spec, float Lambda0, float coff, float &r_io, float &g_io, float &b_io)
{
    const float range = sp.s_max - sp.s_min;
    float X = 0.0f, Y = 0.0f, Z = 0.0f;
    for (int k = 0; k < HERO_N; k++) {
        const v3 xyz = sensor_sample(sp, Lambda0 + (float)k * HERO_LAMBDA_STEP);
        X += (xyz.x * spec.v[k]) * range / (float)HERO_N;
        Y += (xyz.y * spec.v[k]) * range / (float)HERO_N;
        Z += (xyz.z * spec.v[k]) * range / (float)HERO_N;
    }
    // UtilsFunc.py:42 xyz_to_srgb @ xyz
    const float r = ((float)3.240479 * X + (float)-1.537150 * Y) + (float)-0.498535 * Z;
    const float g = ((float)-0.969256 * X + (float)1.875991 * Y) + (float)0.041556 * Z;
    const float b = ((float)0.055648 * X + (float)-0.204043 * Y) + (float)1.057311 * Z;
    r_io = mixf(r_io, r, coff); g_io = mixf(g_io, g, coff); b_io = mixf(b_io, b, coff);
}

}  // namespace tirt
