// tirt_spectral.hip -- host entry points of the spectral path: upload of PT_Spec's tables and the
// device version of the reference's offline optimiser spectrum/JakobSpecTable.py, whose output (spectrum/spec_table) the reference
// repository does not carry (.MISSING_LARGE_BLOBS).  The integrator itself (k_shade_spec, k_film_spec) sits with the wavefront
// in tirt_render.hip; the device functions are in tirt_spectral.h.
#include "tirt_internal.h"
#include "tirt_spectral.h"

namespace tirt {

// ---- spectrum/JakobSpecTable.py -------------------------------------------------------------------------------------------------
// For every cell of a 3 x res^3 grid over RGB -- l: which component is the largest; its value is scale[k]; the two others are
// the fractions i / (res-1), j / (res-1) of it -- three coefficients of a sigmoid-of-polynomial spectrum are fitted by Gauss-Newton
// in CIE Lab under D65 (:301-332), warm-started along k from k = res / 5 upwards and then downwards (:345-375).  One thread per
// (l, j, i) walks its k chain: 3 x res^2 = 12 288 independent chains of 64 + 13 solves, each up to 15 iterations of seven residual
// evaluations over the 471 wavelengths -- double precision like the reference (ti.init(default_fp=ti.f64)), the wavelength tables
// in LDS.  0.3 s on an MI355X (the reference notes "a long time"; the CPU oracle takes 8 s on eight threads).
constexpr int SPT_N = 471;               // 360 .. 830 nm in 1 nm steps
struct SptArgs { const double *rgb_tbl; double wp[3]; const double *scale; int res; float *out; };

__device__ __forceinline__ double spt_sigmoid(double x) { return 0.5 * x / tm_sqrtd(1.0 + x * x) + 0.5; }
__device__ __forceinline__ double spt_f(double t)                    // :88-97; pow(t, 1/3) as exp(log(t) / 3) on the shared double kernels
{
    const double delta = 6.0 / 29.0;
    return (t > delta * delta * delta) ? tm_expd(tm_logd(t) * (1.0 / 3.0)) : t / (delta * delta * 3.0) + (4.0 / 29.0);
}
__device__ __forceinline__ void spt_cie_lab(const double *wp, double r, double g, double b, double &L, double &A, double &B)    // :99-105
{
    const double X = (0.412453 * r + 0.357580 * g) + 0.180423 * b;
    const double Y = (0.212671 * r + 0.715160 * g) + 0.072169 * b;
    const double Z = (0.019334 * r + 0.119193 * g) + 0.950227 * b;
    L = 116.0 * spt_f(Y / wp[1]) - 16.0;
    A = 500.0 * (spt_f(X / wp[0]) - spt_f(Y / wp[1]));
    B = 200.0 * (spt_f(Y / wp[1]) - spt_f(Z / wp[2]));
}
__device__ void spt_residual(const double *tbl, const double *wp, double c0, double c1, double c2, const double *rgb, double *out)   // :260-277
{
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i = 0; i < SPT_N; i++) {
        const double L = ((360.0 + (double)i) - 360.0) / (830.0 - 360.0);
        double x = c0;
        x = x * L + c1;
        x = x * L + c2;
        const double sg = spt_sigmoid(x);
        a0 += tbl[3 * i] * sg; a1 += tbl[3 * i + 1] * sg; a2 += tbl[3 * i + 2] * sg;
    }
    double l0, l1, l2, m0, m1, m2;
    spt_cie_lab(wp, rgb[0], rgb[1], rgb[2], l0, l1, l2);
    spt_cie_lab(wp, a0, a1, a2, m0, m1, m2);
    out[0] = l0 - m0; out[1] = l1 - m1; out[2] = l2 - m2;
}
// :107-209 and :212-257 -- LU with partial pivoting as written there: the pivot search of the second column looks at column 0 again (:170)
__device__ int spt_lup(double A[3][3], int P[3])
{
    const double Tol = 1e-15;
    int ret = 1;
    P[0] = 0; P[1] = 1; P[2] = 2;
    double maxA = 0.0; int imax = 0;
    for (int k = 0; k < 3; k++) { const double a = fabs(A[k][0]); if (a > maxA) { maxA = a; imax = k; } }
    if (maxA < Tol) ret = 0;
    if (imax != 0) {
        for (int q = 0; q < 3; q++) { const double t = A[0][q]; A[0][q] = A[imax][q]; A[imax][q] = t; }
        const int t = P[0]; P[0] = P[imax]; P[imax] = t;
    }
    A[1][0] /= A[0][0]; A[1][1] -= A[1][0] * A[0][1]; A[1][2] -= A[1][0] * A[0][2];
    A[2][0] /= A[0][0]; A[2][1] -= A[2][0] * A[0][1]; A[2][2] -= A[2][0] * A[0][2];
    maxA = 0.0; imax = 1;
    for (int k = 1; k < 3; k++) { const double a = fabs(A[k][0]); if (a > maxA) { maxA = a; imax = k; } }
    if (maxA < Tol) ret = 0;
    if (imax != 1) {
        for (int q = 0; q < 3; q++) { const double t = A[1][q]; A[1][q] = A[2][q]; A[2][q] = t; }
        const int t = P[1]; P[1] = P[2]; P[2] = t;
    }
    A[2][1] /= A[1][1]; A[2][2] -= A[2][1] * A[1][2];
    if (fabs(A[2][2]) < Tol) ret = 0;
    return ret;
}
__device__ int spt_gauss_newton(const double *tbl, const double *wp, const double *rgb, double *co)     // :301-332
{
    const double EPS = 1e-4;
    int rv = 1;
    for (int it = 0; it < 15; it++) {
        double res[3], J[3][3];
        spt_residual(tbl, wp, co[0], co[1], co[2], rgb, res);
        for (int i = 0; i < 3; i++) {
            double r0[3], r1[3];
            spt_residual(tbl, wp, i == 0 ? co[0] - EPS : co[0], i == 1 ? co[1] - EPS : co[1], i == 2 ? co[2] - EPS : co[2], rgb, r0);
            spt_residual(tbl, wp, i == 0 ? co[0] + EPS : co[0], i == 1 ? co[1] + EPS : co[1], i == 2 ? co[2] + EPS : co[2], rgb, r1);
            for (int q = 0; q < 3; q++) J[q][i] = (r1[q] - r0[q]) / (2.0 * EPS);
        }
        int P[3];
        rv = spt_lup(J, P);
        if (rv != 1) break;
        double x[3];
        x[0] = res[P[0]];
        x[1] = res[P[1]]; x[1] -= J[1][0] * x[0];
        x[2] = res[P[2]]; x[2] -= J[2][0] * x[0]; x[2] -= J[2][1] * x[1];
        x[2] = x[2] / J[2][2];
        x[1] -= J[1][2] * x[2]; x[1] = x[1] / J[1][1];
        x[0] -= J[0][1] * x[1]; x[0] -= J[0][2] * x[2]; x[0] = x[0] / J[0][0];
        co[0] -= x[0]; co[1] -= x[1]; co[2] -= x[2];
        const double r = (res[0] * res[0] + res[1] * res[1]) + res[2] * res[2];
        if (r < 0.000001) break;
        const double m01 = co[0] > co[1] ? co[0] : co[1], cm = m01 > co[2] ? m01 : co[2];
        if (cm > 200.0) { const double k = 200.0 / cm; co[0] *= k; co[1] *= k; co[2] *= k; }
    }
    return rv;
}
__global__ __launch_bounds__(64) void k_spec_table(SptArgs a)
{
    __shared__ double s_tbl[3 * SPT_N];
    for (int k = threadIdx.x; k < 3 * SPT_N; k += blockDim.x) s_tbl[k] = a.rgb_tbl[k];
    __syncthreads();
    const int res = a.res;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= 3 * res * res) return;
    const int l = cell / (res * res), j = (cell / res) % res, i = cell % res;
    const double x = (double)i / (double)(res - 1), y = (double)j / (double)(res - 1);
    for (int pass = 0; pass < 2; pass++) {                                              // sovle(l), :345-375
        double co[3] = {0.0, 0.0, 0.0};
        for (int k = res / 5; pass == 0 ? k < res : k >= 0; k += (pass == 0 ? 1 : -1)) {
            const double b = a.scale[k];
            double rgb[3];
            if (l == 0) { rgb[0] = b; rgb[1] = x * b; rgb[2] = y * b; }
            else if (l == 1) { rgb[1] = b; rgb[2] = x * b; rgb[0] = y * b; }
            else { rgb[2] = b; rgb[0] = x * b; rgb[1] = y * b; }
            if (spt_gauss_newton(s_tbl, a.wp, rgb, co) != 1) break;
            const double c0 = 360.0, c1 = 1.0 / (830.0 - 360.0);                         // write_to_result, :69-77
            const long idx = (((long)l * res + k) * res + j) * res + i;
            a.out[3 * idx + 0] = (float)(co[0] * (c1 * c1));
            a.out[3 * idx + 1] = (float)(co[1] * c1 - 2 * co[0] * c0 * (c1 * c1));
            a.out[3 * idx + 2] = (float)(co[2] - co[1] * c0 * c1 + co[0] * ((c0 * c1) * (c0 * c1)));
        }
    }
}

// known-answer evaluation of the spectral device functions (tirt_kat_spec; `which` as in include/tirt.h)
__global__ void k_kat_spec(SpecView sp, int which, const float *in, int in_stride, float *out, int out_stride, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *a = in + (size_t)i * in_stride;
    float *o = out + (size_t)i * out_stride;
    f4s r = f4_set(0.0f); bool out4 = false;
    if (which == 0) o[0] = spd_sample(sp.spd[(int)a[0]], a[1]);
    else if (which == 1) { r = hero_sample(sp.spd[(int)a[0]], a[1]); out4 = true; }
    else if (which == 2) { for (int k = 0; k < HERO_N; k++) { const v3 c = sensor_sample(sp, a[0] + (float)k * HERO_LAMBDA_STEP); o[k] = c.x; o[4 + k] = c.y; o[8 + k] = c.z; } }
    else if (which == 3) { const v3 c = r2s_fetch(sp, V(a[0], a[1], a[2])); o[0] = c.x; o[1] = c.y; o[2] = c.z; }
    else if (which == 4) o[0] = r2s_eval(V(a[0], a[1], a[2]), a[3]);
    else if (which == 5) { r = srgb_to_spec(sp, V(a[0], a[1], a[2]), a[3]); out4 = true; }
    else if (which == 6) { for (int k = 0; k < HERO_N; k++) o[k] = sky_radiance(sp, a[0], a[1], a[2] + (float)k * HERO_LAMBDA_STEP); }
    else if (which == 7) { r = emission_to_rad(sp, V(a[0], a[1], a[2]), a[3]); out4 = true; }
    else if (which == 8) { for (int k = 0; k < HERO_N; k++) o[k] = tm_exp(-a[1] / (a[0] + (float)k * HERO_LAMBDA_STEP)); }
    else if (which == 9) { for (int k = 0; k < HERO_N; k++) r.v[k] = a[k]; float x = a[6], y = a[7], z = a[8]; spec_add_splat(sp, r, a[4], a[5], x, y, z); o[0] = x; o[1] = y; o[2] = z; }
    else if (which == 10) { r = get_spec_power(sp, a, a[10]); out4 = true; }
    else if (which == 11) { const int index = (int)(a[0] * (float)HERO_N); o[0] = (float)index; o[1] = a[1] + (float)index * HERO_LAMBDA_STEP; }
    if (out4) for (int k = 0; k < HERO_N; k++) o[k] = r.v[k];
}

}  // namespace tirt


using namespace tirt;

extern "C" {

int tirt_spec_table_build(tirt_ctx *c, int res, const float *cie_xyz, const float *d65, int n, float *scale_out, float *coeff_out)
{
    TIRT_REQUIRE(c && cie_xyz && d65 && scale_out && coeff_out, "tirt_spec_table_build: null");
    TIRT_REQUIRE(n == SPT_N && res >= 5 && res <= 64, "tirt_spec_table_build: 471 wavelengths (360..830 nm), 5 <= res <= 64");
    TIRT_HIP(hipSetDevice(c->device));
    // pre_compute (:334-343) and the normalisation by the white point's Y (:413-417): 471 terms, on the host, in index order
    std::vector<double> tbl(3 * SPT_N), scale(res);
    double wp[3] = {0.0, 0.0, 0.0};
    const double h = (830.0 - 360.0) / (double)(n - 1);
    for (int i = 0; i < n; i++) {
        double weight = 3.0 / 8.0 * h;
        if ((i == 0) || (i == n - 1)) { }                              // (JakobSpecTable.py:338: the end points keep 3/8 h)
        else if ((i - 1) % 3 == 2) weight = weight * 2.0;
        else weight = weight * 3.0;
        const double X = (double)cie_xyz[3 * i], Y = (double)cie_xyz[3 * i + 1], Z = (double)cie_xyz[3 * i + 2], D = (double)d65[i];
        tbl[3 * i + 0] = (((3.240479 * X + -1.537150 * Y) + -0.498535 * Z) * D) * weight;
        tbl[3 * i + 1] = (((-0.969256 * X + 1.875991 * Y) + 0.041556 * Z) * D) * weight;
        tbl[3 * i + 2] = (((0.055648 * X + -0.204043 * Y) + 1.057311 * Z) * D) * weight;
        wp[0] += (X * D) * weight; wp[1] += (Y * D) * weight; wp[2] += (Z * D) * weight;
    }
    for (int i = 0; i < res; i++) {
        const double t = (double)i / (double)(res - 1), s1 = t * t * (3.0 - 2.0 * t);
        scale[i] = s1 * s1 * (3.0 - 2.0 * s1);
        scale_out[i] = (float)scale[i];
    }
    for (int i = 0; i < 3 * n; i++) tbl[i] /= wp[1];
    SptArgs a;
    a.wp[0] = wp[0] / wp[1]; a.wp[2] = wp[2] / wp[1]; a.wp[1] = wp[1] / wp[1];
    a.res = res;
    DevBuf d_tbl, d_scale, d_out;
    const size_t nout = sizeof(float) * 9 * (size_t)res * res * res;
    int rc = TIRT_OK;
    if (d_tbl.ensure(sizeof(double) * tbl.size()) || d_scale.ensure(sizeof(double) * scale.size()) || d_out.ensure(nout)) rc = TIRT_ERR_HIP;
    if (rc == TIRT_OK) {
        hipError_t e = hipMemcpyAsync(d_tbl.p, tbl.data(), sizeof(double) * tbl.size(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_scale.p, scale.data(), sizeof(double) * scale.size(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_out.p, 0, nout, c->stream);
        a.rgb_tbl = d_tbl.as<double>(); a.scale = d_scale.as<double>(); a.out = d_out.as<float>();
        if (e == hipSuccess) {
            const int cells = 3 * res * res;
            hipLaunchKernelGGL(k_spec_table, dim3((cells + 63) / 64), dim3(64), 0, c->stream, a);
            e = hipMemcpyAsync(coeff_out, d_out.p, nout, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error(std::string("tirt_spec_table_build: ") + hipGetErrorString(e)); rc = TIRT_ERR_HIP; }
    }
    d_tbl.release(); d_scale.release(); d_out.release();
    return rc;
}

int tirt_kat_spec(tirt_ctx *c, int which, const float *in, int in_stride, float *out, int out_stride, int n)
{
    TIRT_REQUIRE(c && in && out && n >= 0 && which >= 0 && which <= 11, "tirt_kat_spec: bad arguments");
    TIRT_REQUIRE(c->spec_set && c->spec_view, "tirt_kat_spec: tirt_spectral_upload first");
    if (n == 0) return TIRT_OK;
    TIRT_HIP(hipSetDevice(c->device));
    DevBuf din, dout;
    int rc = TIRT_OK;
    if (din.ensure(sizeof(float) * (size_t)n * in_stride) || dout.ensure(sizeof(float) * (size_t)n * out_stride)) rc = TIRT_ERR_HIP;
    if (rc == TIRT_OK) {
        hipError_t e = hipMemcpyAsync(din.p, in, sizeof(float) * (size_t)n * in_stride, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(dout.p, 0, sizeof(float) * (size_t)n * out_stride, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_kat_spec, dim3((n + 255) / 256), dim3(256), 0, c->stream, *(const SpecView *)c->spec_view, which, din.as<float>(), in_stride, dout.as<float>(), out_stride, n);
            e = hipMemcpyAsync(out, dout.p, sizeof(float) * (size_t)n * out_stride, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error(std::string("tirt_kat_spec: ") + hipGetErrorString(e)); rc = TIRT_ERR_HIP; }
    }
    din.release(); dout.release();
    return rc;
}

int tirt_spectral_upload(tirt_ctx *c, const tirt_spectral_t *t)
{
    TIRT_REQUIRE(c && t && t->sensor && t->spd && t->tbl_scale && t->tbl_data && t->sky_cfg && t->sky_rad, "tirt_spectral_upload: null");
    TIRT_REQUIRE(t->n_sensor >= 2 && t->tbl_res >= 2 && t->tbl_res <= 256, "tirt_spectral_upload: bad table sizes");
    TIRT_HIP(hipSetDevice(c->device));
    if (int rc = flush_pending(c)) return rc;
    if (sync_all(c)) return TIRT_ERR_HIP;
    size_t n_spd = 0;
    // the samplers index with (int)((lambda - min) / range) and clamp only the upper neighbour (Spectrum.py:sample): a table shorter than its
    // declared wavelength span would be read past its end
    auto spans = [](float lo, float hi, float step, int n) { return step > 0.0f && hi >= lo && (double)(hi - lo) / (double)step <= (double)(n - 1) * (1.0 + 1.0e-5) + 1.0e-3; };      // (relative: `step` arrives rounded to fp32)
    TIRT_REQUIRE(spans(t->s_min, t->s_max, t->s_range, t->n_sensor), "tirt_spectral_upload: the sensor table is shorter than (s_max - s_min) / s_range + 1");
    for (int k = 0; k < 4; k++) {
        TIRT_REQUIRE(t->spd_n[k] >= 2, "tirt_spectral_upload: a spectrum needs two samples");
        TIRT_REQUIRE(spans(t->spd_min[k], t->spd_max[k], t->spd_range[k], t->spd_n[k]), "tirt_spectral_upload: a spectrum is shorter than (max - min) / range + 1");
        n_spd += (size_t)t->spd_n[k];
    }
    const size_t n_tbl = (size_t)9 * t->tbl_res * t->tbl_res * t->tbl_res;
    // one buffer: sensor | spectra | table scale | table data | sky configs | sky radiances (all f32, 16-byte aligned pieces)
    auto al = [](size_t n) { return (n + 3) & ~(size_t)3; };
    const size_t o_sensor = 0, o_spd = o_sensor + al(3 * (size_t)t->n_sensor), o_scale = o_spd + al(n_spd), o_data = o_scale + al((size_t)t->tbl_res),
                 o_cfg = o_data + al(n_tbl), o_rad = o_cfg + al(99), total = o_rad + al(11);
    if (c->spec_mem.ensure(sizeof(float) * total)) return TIRT_ERR_HIP;
    float *base = c->spec_mem.as<float>();
    hipStream_t st = c->stream;
    TIRT_HIP(hipMemcpyAsync(base + o_sensor, t->sensor, sizeof(float) * 3 * (size_t)t->n_sensor, hipMemcpyHostToDevice, st));
    TIRT_HIP(hipMemcpyAsync(base + o_spd, t->spd, sizeof(float) * n_spd, hipMemcpyHostToDevice, st));
    TIRT_HIP(hipMemcpyAsync(base + o_scale, t->tbl_scale, sizeof(float) * (size_t)t->tbl_res, hipMemcpyHostToDevice, st));
    TIRT_HIP(hipMemcpyAsync(base + o_data, t->tbl_data, sizeof(float) * n_tbl, hipMemcpyHostToDevice, st));
    TIRT_HIP(hipMemcpyAsync(base + o_cfg, t->sky_cfg, sizeof(float) * 99, hipMemcpyHostToDevice, st));
    TIRT_HIP(hipMemcpyAsync(base + o_rad, t->sky_rad, sizeof(float) * 11, hipMemcpyHostToDevice, st));
    TIRT_HIP(hipStreamSynchronize(st));
    SpecView *v = c->spec_view ? (SpecView *)c->spec_view : new SpecView();
    v->sensor = base + o_sensor; v->n_sensor = t->n_sensor; v->s_min = t->s_min; v->s_max = t->s_max; v->s_range = t->s_range;
    size_t off = o_spd;
    for (int k = 0; k < 4; k++) {
        v->spd[k].data = base + off; v->spd[k].n = t->spd_n[k]; v->spd[k].lmin = t->spd_min[k]; v->spd[k].lmax = t->spd_max[k]; v->spd[k].lrange = t->spd_range[k];
        off += (size_t)t->spd_n[k];
    }
    v->tbl_scale = base + o_scale; v->tbl_data = base + o_data; v->tbl_res = t->tbl_res;
    v->sky_cfg = base + o_cfg; v->sky_rad = base + o_rad;
    for (int k = 0; k < 3; k++) v->sun_dir[k] = t->sun_dir[k];
    c->spec_view = v; c->spec_set = true;
    // a copy of the view in device memory: the BDPT kernels read it through a pointer (BdCtx::spec, tirt_bdpt.hip)
    if (c->spec_dev.ensure(sizeof(SpecView))) return TIRT_ERR_HIP;
    TIRT_HIP(hipMemcpy(c->spec_dev.p, v, sizeof(SpecView), hipMemcpyHostToDevice));
    return TIRT_OK;
}

}  // extern "C"
