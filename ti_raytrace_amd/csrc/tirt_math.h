/*
 * tirt_math.h -- deterministic scalar math + counter-based RNG shared by the HIP
 * device code (hipcc, gfx950) and by host C (gcc).
 *
 * Why this exists: the render parity bar is "images within 1e-3 relative L2 at a
 * fixed seed".  A path tracer amplifies 1-ulp differences in sin/cos/pow into
 * different paths, so the transcendental functions are defined HERE, once, as plain
 * IEEE-754 double-precision polynomial evaluations (no libm, no device-libs, no
 * FMA contraction: every translation unit that includes this file is compiled with
 * -ffp-contract=off).  Given the same input bits they return the same output bits on
 * an x86-64 host and on a CDNA4 device.  Accuracy (checked in tests/test_math.py
 * against libm in double): <= 1 ulp(f32) for every function.
 *
 * The functions mirror what the reference gets from Taichi/taichi_glsl
 * (ti.sin/cos/exp/log/pow/sqrt, ts.atan(y,x), ts.acos) -- see SURVEY.md 8c table of
 * un-vendored third-party arithmetic.
 *
 * RNG: the reference uses Taichi's per-thread xorshift (`ti.random()`), which is not
 * reproducible across runs.  Both our oracle and the HIP path use
 * the counter-based generator tm_rand(seed, pixel, frame, dim) below with the
 * dimension schedule of SURVEY.md Appendix A.6.
 */
#ifndef TIRT_MATH_H
#define TIRT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define TM_HD __host__ __device__ static inline
#else
#define TM_HD static inline
#endif

/* ---- bit casts --------------------------------------------------------------- */
TM_HD uint32_t tm_f2u(float f)  { union { float f; uint32_t u; } c; c.f = f; return c.u; }
TM_HD float    tm_u2f(uint32_t u){ union { float f; uint32_t u; } c; c.u = u; return c.f; }
TM_HD uint64_t tm_d2u(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }
TM_HD double   tm_u2d(uint64_t u){ union { double d; uint64_t u; } c; c.u = u; return c.d; }

TM_HD float tm_nan(void) { return tm_u2f(0x7fc00000u); }

/* ---- sqrt: IEEE correctly rounded on both sides ------------------------------- */
TM_HD float  tm_sqrt(float x)   { return __builtin_sqrtf(x); }
TM_HD double tm_sqrtd(double x) { return __builtin_sqrt(x); }

/* ---- 2^k as a double, k in [-1022, 1023] --------------------------------------- */
TM_HD double tm_pow2i(int k) { return tm_u2d((uint64_t)(k + 1023) << 52); }

/* ---- exp (double core) ---------------------------------------------------------- */
TM_HD double tm_expd(double x)
{
    if (x != x) return x;
    if (x > 709.0)  return tm_u2d(0x7ff0000000000000ull);
    if (x < -745.0) return 0.0;
    double kf = x * 1.4426950408889634;
    int k = (int)(kf + (kf >= 0.0 ? 0.5 : -0.5));
    double r = (x - (double)k * 0.6931471803691238) - (double)k * 1.9082149292705877e-10;
    /* Taylor to r^13, |r| <= 0.3466 -> truncation < 1e-18 */
    double p = 1.6059043836821613e-10;
    p = p * r + 2.08767569878681e-09;
    p = p * r + 2.505210838544172e-08;
    p = p * r + 2.755731922398589e-07;
    p = p * r + 2.7557319223985893e-06;
    p = p * r + 2.48015873015873e-05;
    p = p * r + 0.0001984126984126984;
    p = p * r + 0.001388888888888889;
    p = p * r + 0.008333333333333333;
    p = p * r + 0.041666666666666664;
    p = p * r + 0.16666666666666666;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    /* scale in two steps so that k down to -1074 stays representable */
    if (k < -1000) return (p * tm_pow2i(k + 1000)) * tm_pow2i(-1000);
    if (k > 1000)  return (p * tm_pow2i(k - 1000)) * tm_pow2i(1000);
    return p * tm_pow2i(k);
}

/* ---- log (double core), x > 0 finite and normal ---------------------------------- */
TM_HD double tm_logd(double x)
{
    uint64_t u = tm_d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = tm_u2d((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);   /* [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }                   /* [0.707,1.414] */
    double s = (m - 1.0) / (m + 1.0);                                         /* |s| <= 0.1716 */
    double s2 = s * s;
    double p = 0.05263157894736842;       /* 1/19 */
    p = p * s2 + 0.058823529411764705;    /* 1/17 */
    p = p * s2 + 0.06666666666666667;
    p = p * s2 + 0.07692307692307693;
    p = p * s2 + 0.09090909090909091;
    p = p * s2 + 0.1111111111111111;
    p = p * s2 + 0.14285714285714285;
    p = p * s2 + 0.2;
    p = p * s2 + 0.3333333333333333;
    p = p * s2 + 1.0;
    double lm = 2.0 * s * p;
    return ((double)e * 0.6931471803691238 + lm) + (double)e * 1.9082149292705877e-10;
}

TM_HD float tm_exp(float x) { return (float)tm_expd((double)x); }

TM_HD float tm_log(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return tm_nan();
    if (x == 0.0f) return tm_u2f(0xff800000u);
    if (x > 3.4028234e38f) return x;
    return (float)tm_logd((double)x);   /* every positive float is a normal double */
}

/* powf semantics for the cases the renderer produces: x >= 0 any y; x < 0 with
 * integral y (Schlick's pow(1-cos, 5.0) when cos creeps above 1). */
TM_HD float tm_pow(float x, float y)
{
    if (x != x || y != y) return tm_nan();
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : tm_u2f(0x7f800000u);
    float ax = x < 0.0f ? -x : x;
    double r;
    if (ax > 3.4028234e38f) r = (y > 0.0f) ? (double)ax : 0.0;
    else r = tm_expd((double)y * tm_logd((double)ax));
    if (x < 0.0f) {
        float ay = y < 0.0f ? -y : y;
        if (ay >= 16777216.0f) return (float)r;               /* even integer */
        int iy = (int)ay;
        if ((float)iy != ay) return tm_nan();
        if (iy & 1) r = -r;
    }
    return (float)r;
}

/* ---- sin / cos (double core, Cody-Waite on pi/2) ---------------------------------- */
TM_HD double tm_ksin(double r)
{
    double r2 = r * r;
    double p = 2.8114572543455206e-15;
    p = p * r2 + -7.647163731819816e-13;
    p = p * r2 + 1.6059043836821613e-10;
    p = p * r2 + -2.505210838544172e-08;
    p = p * r2 + 2.7557319223985893e-06;
    p = p * r2 + -0.0001984126984126984;
    p = p * r2 + 0.008333333333333333;
    p = p * r2 + -0.16666666666666666;
    return r + r * (r2 * p);
}
TM_HD double tm_kcos(double r)
{
    double r2 = r * r;
    double p = -1.5619206968586225e-16;
    p = p * r2 + 4.779477332387385e-14;
    p = p * r2 + -1.1470745597729725e-11;
    p = p * r2 + 2.08767569878681e-09;
    p = p * r2 + -2.755731922398589e-07;
    p = p * r2 + 2.48015873015873e-05;
    p = p * r2 + -0.001388888888888889;
    p = p * r2 + 0.041666666666666664;
    p = p * r2 + -0.5;
    return 1.0 + r2 * p;
}
/* valid for |x| < ~1e6 (the renderer only passes angles in [0, 2pi]) */
TM_HD int tm_reduce(double x, double *r)
{
    double kf = x * 0.6366197723675814;
    int k = (int)(kf + (kf >= 0.0 ? 0.5 : -0.5));
    *r = (x - (double)k * 1.5707963267341256) - (double)k * 6.077100506506192e-11;
    return k;
}
TM_HD float tm_sin(float x)
{
    if (x != x || x > 1.0e6f || x < -1.0e6f) return tm_nan();
    double r; int q = tm_reduce((double)x, &r) & 3;
    double v = (q & 1) ? tm_kcos(r) : tm_ksin(r);
    return (float)((q & 2) ? -v : v);
}
TM_HD float tm_cos(float x)
{
    if (x != x || x > 1.0e6f || x < -1.0e6f) return tm_nan();
    double r; int q = tm_reduce((double)x, &r) & 3;
    double v = (q & 1) ? tm_ksin(r) : tm_kcos(r);
    return (float)(((q + 1) & 2) ? -v : v);
}

/* sin and cos of the same angle from ONE range reduction; bit-identical to tm_sin / tm_cos */
/* tan as the quotient of the two correctly rounded values above (<= 1.5 ulp; the reference's ts.tan is LLVM's tanf) -- only
 * Scene.sample_light's spot branch uses it (Scene.py:458-459) */
TM_HD float tm_tan(float x) { return tm_sin(x) / tm_cos(x); }
TM_HD void tm_sincos(float x, float *s, float *c)
{
    if (x != x || x > 1.0e6f || x < -1.0e6f) { *s = tm_nan(); *c = tm_nan(); return; }
    double r; int q = tm_reduce((double)x, &r) & 3;
    double ks = tm_ksin(r), kc = tm_kcos(r);
    double sv = (q & 1) ? kc : ks;
    double cv = (q & 1) ? ks : kc;
    *s = (float)((q & 2) ? -sv : sv);
    *c = (float)(((q + 1) & 2) ? -cv : cv);
}

/* ---- atan / atan2 / acos ------------------------------------------------------------ */
/* atan of z in [0,1]: table of atan(k/8) + odd series on the residual */
TM_HD double tm_atan01(double z)
{
    int k = (int)(z * 8.0 + 0.5);
    double c = (double)k * 0.125;
    double t = (z - c) / (1.0 + z * c);           /* |t| <= 1/16 */
    double t2 = t * t;
    double p = 0.06666666666666667;               /* 1/15 */
    p = p * t2 + -0.07692307692307693;
    p = p * t2 + 0.09090909090909091;
    p = p * t2 + -0.1111111111111111;
    p = p * t2 + 0.14285714285714285;
    p = p * t2 + -0.2;
    p = p * t2 + 0.3333333333333333;
    double at = t - t * (t2 * p);
    double base;
    switch (k) {
        case 0: base = 0.0; break;
        case 1: base = 0.12435499454676144; break;
        case 2: base = 0.24497866312686414; break;
        case 3: base = 0.35877067027057225; break;
        case 4: base = 0.4636476090008061; break;
        case 5: base = 0.5585993153435624; break;
        case 6: base = 0.6435011087932844; break;
        case 7: base = 0.7188299996216245; break;
        default: base = 0.7853981633974483; break;
    }
    return base + at;
}
TM_HD double tm_atan2d(double y, double x)
{
    double ay = y < 0.0 ? -y : y, ax = x < 0.0 ? -x : x;
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ay <= ax) a = tm_atan01(ay / ax);
    else a = 1.5707963267948966 - tm_atan01(ax / ay);
    if (x < 0.0) a = 3.141592653589793 - a;
    return y < 0.0 ? -a : a;
}
/* GLSL atan(y, x); signed zeros are not distinguished (atan2(-0, -1) = +pi) */
TM_HD float tm_atan2(float y, float x)
{
    if (x != x || y != y) return tm_nan();
    return (float)tm_atan2d((double)y, (double)x);
}
TM_HD float tm_acos(float x)
{
    if (x != x || x > 1.0f || x < -1.0f) return tm_nan();
    double xd = (double)x;
    return (float)tm_atan2d(tm_sqrtd((1.0 - xd) * (1.0 + xd)), xd);
}

/* ---- floor for |x| < 2^31 (texture coordinates) ------------------------------------- */
TM_HD float tm_floor(float x)
{
    float t = (float)(int)x;
    return (t > x) ? t - 1.0f : t;
}

/* ---- counter-based RNG ----------------------------------------------------------------
 * u = tm_rand(seed, pixel, frame, dim) in [0,1), 24 random bits.
 * Dimension schedule:
 *   0,1                     camera jitter jx, jy (drawn only when frame != 0)
 *   2 + 8*bounce + slot     slot 0 light index | glass Fresnel choice
 *                           slot 1,2 light point (a, b)
 *                           slot 3,4,5 Disney lobe, r1, r2
 *                           slot 6 glass extinction roulette
 */
#define TM_DIM_JX 0
#define TM_DIM_JY 1
#define TM_DIM_BOUNCE0 2
#define TM_DIMS_PER_BOUNCE 8
#define TM_SLOT_LIGHT 0
#define TM_SLOT_GLASS 0
#define TM_SLOT_LA 1
#define TM_SLOT_LB 2
#define TM_SLOT_LOBE 3
#define TM_SLOT_R1 4
#define TM_SLOT_R2 5
#define TM_SLOT_EXT 6

TM_HD uint32_t tm_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
TM_HD uint32_t tm_rand_u32(uint32_t seed, uint32_t pixel, uint32_t frame, uint32_t dim)
{
    uint32_t h = tm_mix32(seed ^ 0x9e3779b9u);
    h = tm_mix32(h ^ pixel);
    h = tm_mix32(h + frame * 0x85ebca6bu + 0x68bc21ebu);
    h = tm_mix32(h ^ (dim * 0xc2b2ae35u + 0x02e5be93u));
    return h;
}
TM_HD float tm_rand(uint32_t seed, uint32_t pixel, uint32_t frame, uint32_t dim)
{
    return (float)(tm_rand_u32(seed, pixel, frame, dim) >> 8) * 5.9604644775390625e-08f;
}

#endif /* TIRT_MATH_H */
